"""`SpacedDiffusion` / `space_timesteps` -- diffusion/respace.py:9-134 on the MI355X path.

The reference wraps the model in `_WrappedModel` and rebuilds a `map_tensor` on the device every step
(respace.py:125-130); here the map is a host int array handed to the native loop once.
"""
import numpy as np

from .gaussian_diffusion import GaussianDiffusion


def space_timesteps(num_timesteps, section_counts):
    """respace.py:9-62: which original timesteps a respaced process keeps."""
    if isinstance(section_counts, str):
        if section_counts.startswith("ddim"):
            desired = int(section_counts[len("ddim"):])
            for stride in range(1, num_timesteps):
                if len(range(0, num_timesteps, stride)) == desired:
                    return set(range(0, num_timesteps, stride))
            raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
        section_counts = [int(x) for x in section_counts.split(",")]
    size_per = num_timesteps // len(section_counts)
    extra = num_timesteps % len(section_counts)
    start, taken = 0, []
    for i, count in enumerate(section_counts):
        size = size_per + (1 if i < extra else 0)
        if size < count:
            raise ValueError(f"cannot divide section of {size} steps into {count}")
        frac = 1 if count <= 1 else (size - 1) / (count - 1)
        cur = 0.0
        for _ in range(count):
            taken.append(start + round(cur))
            cur += frac
        start += size
    return set(taken)


class SpacedDiffusion(GaussianDiffusion):
    """respace.py:65-110: a diffusion process over a subset of the base process' timesteps."""

    def __init__(self, use_timesteps, **kwargs):
        self.use_timesteps = set(use_timesteps)
        self.original_num_steps = len(kwargs["betas"])
        base_ac = np.cumprod(1.0 - np.array(kwargs["betas"], dtype=np.float64))
        last, new_betas, tmap = 1.0, [], []
        for i, ac in enumerate(base_ac):
            if i in self.use_timesteps:
                new_betas.append(1 - ac / last)
                last = ac
                tmap.append(i)
        kwargs["betas"] = np.array(new_betas)
        super().__init__(**kwargs)
        self.timestep_map = tmap

    def _wrap_model(self, model):
        return model     # the timestep map is applied inside the native loop

    def _scale_timesteps(self, t):
        return t

"""Batch-sharded sampling across the GPUs of one node -- new design: the reference has no live
distributed code (utils/dist_util.py:18-41 is a stub).

Every sample of a batch is an independent Markov chain, so the data path needs NO collective: rank r runs the
fused loop on its contiguous shard with Philox streams keyed by the GLOBAL sample index (so the gathered batch
is bit-identical to the unsharded one), and one all-gather (RCCL over xGMI; `nccl` backend == RCCL on ROCm)
collects the final `[B/G, J, F, T]` fp32 shards.  One process per GPU, launched by torchrun.
"""
import os

import torch
import torch.distributed as dist

_SHARDED_KEYS = ("mask", "lengths", "scale", "inpainting_mask", "inpainted_motion", "prefix", "action", "target_cond")
_SHARDED_LISTS = ("text", "action_text", "target_joint_names", "is_heading")     # per-sample python lists (data_loaders/tensors.py)


def init_from_env(backend=None, force=False):
    """Join the process group torchrun describes (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*); no-op single-process unless
    `force` (a one-rank group: exercises the RCCL bring-up and the collective on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if force and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def shard_bounds(B, rank, world):
    """Contiguous shard [lo, hi) of a batch of B; the first B % world ranks take one extra sample."""
    q, r = divmod(B, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_y(y, lo, hi):
    """Slice the per-sample entries of model_kwargs['y'] (sample/generate.py:107-132 layout)."""
    out = {}
    for k, v in y.items():
        if k == "text_embed" and torch.is_tensor(v):
            out[k] = v[:, lo:hi]                         # [1, B, clip_dim]
        elif k == "text_embed" and isinstance(v, tuple):  # DiP: (tokens [Ntok, B, 768], pad mask [B or 1, Ntok]) (model/mdm.py:180-187)
            tok, pad = v
            if tok.dim() == 4:       # dynamic text (generate.py:139-140): (enc [B, Ntok, P, 768], pad [B, P, Ntok]) -- sample-major
                out[k] = (tok[lo:hi], pad[lo:hi])
            else:
                out[k] = (tok[:, lo:hi], pad if pad.shape[0] == 1 else pad[lo:hi])
        elif k in _SHARDED_LISTS and isinstance(v, (list, tuple)):
            out[k] = list(v[lo:hi])
        elif k == "is_heading" and torch.is_tensor(v):
            out[k] = v[lo:hi]
        elif k in _SHARDED_KEYS and torch.is_tensor(v) and v.dim() >= 1:
            out[k] = v[lo:hi]
        else:
            out[k] = v
    return out


def all_gather_samples(local, B, world):
    """Gather the per-rank shards into the full batch on every rank (shards may be ragged by one sample)."""
    if world == 1 and not dist.is_initialized():
        return local
    sizes = [shard_bounds(B, r, world)[1] - shard_bounds(B, r, world)[0] for r in range(world)]
    if len(set(sizes)) == 1:
        out = torch.empty((B,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    # ragged by one sample: pad every shard to the largest, gather once, drop the pad rows
    n_max = max(sizes)
    padded = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * n_max: r * n_max + n] for r, n in enumerate(sizes)])


def sample_sharded(diffusion, model, shape, model_kwargs, *, seed, ddim=False, gather=True, **loop_kw):
    """`diffusion.p_sample_loop(model, shape, ...)` for the GLOBAL batch `shape[0]`, computed as one shard per rank.

    `seed` must be the same on every rank (it keys the Philox stream together with the global sample index).
    Returns the full [B, J, F, T] batch on every rank (or this rank's shard if gather=False)."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    B = int(shape[0])
    lo, hi = shard_bounds(B, rank, world)
    if hi == lo:
        raise ValueError(f"batch {B} is smaller than the world size {world}")
    kw = dict(model_kwargs or {})
    kw["y"] = shard_y(kw.get("y", {}), lo, hi)
    prev = diffusion.sample_base
    diffusion.sample_base = lo
    try:
        fn = diffusion.ddim_sample_loop if ddim else diffusion.p_sample_loop
        local = fn(model, (hi - lo,) + tuple(shape[1:]), model_kwargs=kw, seed=seed, **loop_kw)
    finally:
        diffusion.sample_base = prev
    return all_gather_samples(local, B, world) if gather else local


def autoregressive_sharded(sampler, diffusion, model, shape, model_kwargs, *, gather=True, **sample_kw):
    """DiP across ranks (BASELINE.json configs[4]: 256 motions over 8 GPUs): `sampler.sample(model, shape, ...)` -- an
    `AutoRegressiveSampler` whose `sample_fn` draws its per-window seeds identically on every rank -- for the GLOBAL batch
    `shape[0]`, one contiguous shard per rank.  As in `sample_sharded` the data path has no collective: `diffusion.sample_base`
    carries the shard's first global sample index into every window's Philox streams, so that the gathered `[B, J, F, frames]`
    batch is bit-identical to the one-rank run; one all-gather at the end."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    B = int(shape[0])
    lo, hi = shard_bounds(B, rank, world)
    if hi == lo:
        raise ValueError(f"batch {B} is smaller than the world size {world}")
    kw = dict(model_kwargs or {})
    kw["y"] = shard_y(kw.get("y", {}), lo, hi)
    prev = diffusion.sample_base
    diffusion.sample_base = lo
    try:
        local = sampler.sample(model, (hi - lo,) + tuple(shape[1:]), model_kwargs=kw, **sample_kw)
    finally:
        diffusion.sample_base = prev
    return all_gather_samples(local, B, world) if gather else local

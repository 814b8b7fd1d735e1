"""TEST INFRASTRUCTURE ONLY -- imports the *upstream reference* (read-only, /root/reference).

This module only works inside the build container where /root/reference is mounted.
It is used by `oracle/make_golden.py` to pin `oracle/mdm_oracle.py` against the real
reference and to generate the fixtures under tests/golden/.  Nothing on the GPU box
may import it at run time (the reference tree does not exist there).

The reference `model/mdm.py` imports `clip` (mdm.py:5) and builds `Rotation2xyz` -> SMPL
(mdm.py:135), neither of which is installed/available offline.  Both are irrelevant to
the denoiser arithmetic, so they are replaced by inert stubs *in sys.modules* -- the
reference sources are not modified or copied.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REFERENCE_ROOT = os.environ.get("MDM_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "model", "mdm.py"))


def _install_stubs():
    sys.dont_write_bytecode = True  # reference tree is read-only
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if "clip" not in sys.modules:
        clip = types.ModuleType("clip")

        class _NoClip(nn.Module):
            def encode_text(self, *_a, **_k):
                raise RuntimeError("CLIP is stubbed out in the oracle harness; pass y['text_embed']")

        clip.load = lambda *a, **k: (_NoClip(), None)
        clip.tokenize = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("clip stub"))
        clip.model = types.SimpleNamespace(convert_weights=lambda m: None)
        sys.modules["clip"] = clip
    if "model.rotation2xyz" not in sys.modules:
        import model  # noqa: F401  (reference package)
        r2x = types.ModuleType("model.rotation2xyz")

        class Rotation2xyz:  # mdm.py:135, :286-293 need .smpl_model to be an nn.Module
            def __init__(self, device="cpu", dataset="amass"):
                self.smpl_model = nn.Module()

            def __call__(self, x, *a, **k):
                return x

        r2x.Rotation2xyz = Rotation2xyz
        sys.modules["model.rotation2xyz"] = r2x


def mdm_kwargs(arch="trans_enc", latent_dim=512, num_layers=8, mask_frames=True, **over):
    """The kwargs `utils/model_util.py:24-71 get_model_args` produces for HumanML3D text2motion."""
    kw = dict(modeltype="", njoints=263, nfeats=1, num_actions=1, translation=True, pose_rep="rot6d",
              glob=True, glob_rot=True, latent_dim=latent_dim, ff_size=1024, num_layers=num_layers,
              num_heads=4, dropout=0.1, activation="gelu", data_rep="hml_vec", cond_mode="text",
              cond_mask_prob=0.1, action_emb="tensor", arch=arch, emb_trans_dec=False,
              clip_version="ViT-B/32", dataset="humanml", text_encoder_type="clip",
              pos_embed_max_len=5000, mask_frames=mask_frames, pred_len=0, context_len=0,
              emb_policy="add", all_goal_joint_names=[], multi_target_cond=False,
              multi_encoder_type="multi", target_enc_layers=1)
    kw.update(over)
    return kw


def build_reference_model(seed=0, **over):
    """Reference MDM with PyTorch-default-init weights under torch.manual_seed(seed)."""
    _install_stubs()
    import io
    import contextlib
    from model.mdm import MDM
    torch.manual_seed(seed)
    with contextlib.redirect_stdout(io.StringIO()):
        m = MDM(**mdm_kwargs(**over))
    m.eval()
    return m


def load_reference_weights(model, state_dict):
    """Load a CLIP-less state dict (ours or the reference's) into the reference model the way
    `utils/model_util.py:8-15 load_model_wo_clip` does: positional tables are recomputed, CLIP is absent."""
    sd = {k: v for k, v in state_dict.items() if "sequence_pos_encoder" not in k and not k.startswith("clip_model.")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert len(unexpected) == 0, unexpected
    assert all(k.startswith("clip_model.") or "sequence_pos_encoder" in k for k in missing), missing
    return model


def reference_state_dict(model):
    """state_dict as `train/training_loop.py:404-410` saves it: CLIP keys stripped."""
    return {k: v.detach().clone() for k, v in model.state_dict().items() if not k.startswith("clip_model.")}


def build_reference_diffusion(steps=50, noise_schedule="cosine", sigma_small=True, respacing=""):
    """`utils/model_util.py:75-116 create_gaussian_diffusion` with the HumanML3D defaults."""
    _install_stubs()
    from diffusion import gaussian_diffusion as gd
    from diffusion.respace import SpacedDiffusion, space_timesteps
    betas = gd.get_named_beta_schedule(noise_schedule, steps, 1.0)
    return SpacedDiffusion(
        use_timesteps=space_timesteps(steps, respacing if respacing else [steps]),
        betas=betas,
        model_mean_type=gd.ModelMeanType.START_X,
        model_var_type=gd.ModelVarType.FIXED_SMALL if sigma_small else gd.ModelVarType.FIXED_LARGE,
        loss_type=gd.LossType.MSE,
        rescale_timesteps=False,
        lambda_vel=0.0, lambda_rcxyz=0.0, lambda_fc=0.0, lambda_target_loc=0.0,
    )


def reference_cfg(model):
    _install_stubs()
    from utils.sampler_util import ClassifierFreeSampleModel
    return ClassifierFreeSampleModel(model)


def make_y(B, T, seed, lengths=None, scale=2.5):
    """Synthetic `model_kwargs['y']` exactly as SURVEY 8d / generate.py:107-132 shapes it (no 'text' key)."""
    g = torch.Generator().manual_seed(seed)
    if lengths is None:
        lengths = [T] * B
    lengths = torch.as_tensor(lengths, dtype=torch.long)
    mask = (torch.arange(T)[None, :] < lengths[:, None]).view(B, 1, 1, T)
    return {
        "mask": mask,
        "lengths": lengths,
        "text_embed": torch.randn(1, B, 512, generator=g),
        "scale": torch.ones(B) * scale,
    }

"""Round-6 GPU parity tests (`-m gpu`): DiP's dynamic-text mode (`--dynamic_text_path`, a prompt per prediction window) against the
reference's own run, and DiP's shard invariance bit for bit in BOTH arithmetic modes."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import dip, make_pair, maxabs, memo, orc, synth_dip_state_dict, synth_dip_y, to_dev

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL_DIP_AR = 2e-4          # CFG 7.5 over 10-step windows; the reference-vs-oracle floor of these fixtures is 2.0e-5 (PIN_REPORT.json)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from mdm_amd import _native
    assert _native.load_native().path.endswith("libmdm_hip.so")


def _sd():
    return memo("sd_dip0", lambda: synth_dip_state_dict(seed=0))


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
@pytest.mark.parametrize("name", ["dip_dynamic_text_B2_P3", "dip_dynamic_text_B4_P2"])
def test_dip_dynamic_text_matches_reference_golden(golden_dir, name, prec):
    """sample/generate.py:63-65, :134-142 + utils/sampler_util.py:52, :66-71 + diffusion/gaussian_diffusion.py:633-635: the fixture is
    the UPSTREAM run (its AutoRegressiveSampler over its p_sample_loop, functional BERT stand-in, CFG 7.5, CPU noise stream).  `y` is
    built exactly as generate.py leaves it -- y['text'] = the prompt list per sample, y['text_embed'] = (enc [B, Ntok, P, 768],
    pad [B, P, Ntok]) -- and goes through this repository's AutoRegressiveSampler.  B != Ntok (round 5: AssertionError) and
    B == Ntok == 4 (round 5: a silently different motion, 3.46 max-abs)."""
    from mdm_amd.sampler_util import AutoRegressiveSampler
    from oracle.synth import synth_dip_dynamic_y
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    steps, B, frames, seed = int(g["steps"]), int(g["B"]), int(g["frames"]), int(g["seed"])
    prompts = [str(p) for p in g["prompts"]]
    model, diffusion = make_pair(_sd(), steps, DEV, guided=True, context_len=20, pred_len=40, precision=prec)
    y = to_dev(synth_dip_dynamic_y(B, 40, 20, seed=int(g["y_seed"]), prompts=prompts, scale=float(g["scale"])), DEV)
    chunks = iter(dip.make_noise_chunks((B, 263, 1, 40), steps, seed, len(prompts)))

    def sample_fn(mdl, shape, **kw):
        x_T, eps = next(chunks)
        return diffusion.p_sample_loop(mdl, shape, noise_sequence=[x_T] + [e.contiguous() for e in eps], **kw)

    args = SimpleNamespace(pred_len=40, context_len=20, autoregressive_include_prefix=False)
    out = AutoRegressiveSampler(args, sample_fn, frames).sample(
        model, (B, 263, 1, frames), clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0, init_image=None,
        progress=False, dump_steps=None, noise=None, const_noise=False)
    err = maxabs(out.cpu(), g["final"])
    print(f"[parity] {name} {prec}: max-abs vs reference = {err:.3e}")
    assert out.shape == (B, 263, 1, frames) and err < TOL_DIP_AR


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_dip_shard_invariance_is_bitwise_in_both_modes(prec):
    """SURVEY 8e for BASELINE.json configs[4]'s per-GPU shape: B = 32 motions x 196 frames (5 windows x 10 steps, CFG 7.5, ragged
    prompts, frame masks, production Philox streams) run as ONE batch and as the shards 12 + 20 with `sample_base` (what
    mdm_amd.dist.autoregressive_sharded does per rank) -- equal BIT FOR BIT in both arithmetic modes (round 5: 2e-5 in f16x3, the hoisted
    memory projection re-associated with its row count); plus run-to-run identity and seed sensitivity."""
    from mdm_amd.dist import shard_y
    from mdm_amd.sampler_util import AutoRegressiveSampler
    B, frames, steps, C, P = 32, 196, 10, 20, 40
    g = torch.Generator().manual_seed(77)
    tl = [int(v) for v in torch.randint(2, 25, (B,), generator=g)]
    y = to_dev(synth_dip_y(B, P, C, seed=43, text_lengths=tl, lengths=[40 - (3 * i) % 17 for i in range(B)], scale=7.5), DEV)
    model, diffusion = make_pair(_sd(), steps, DEV, guided=True, context_len=C, pred_len=P, mask_frames=True, precision=prec)
    args = SimpleNamespace(pred_len=P, context_len=C, autoregressive_include_prefix=False)

    def run(lo, hi, base_seed=600):
        it = iter(range(base_seed, base_seed + 10))
        fn = lambda mdl, shape, **kw: diffusion.p_sample_loop(mdl, shape, seed=next(it), **kw)   # noqa: E731
        diffusion.sample_base = lo
        try:
            return AutoRegressiveSampler(args, fn, frames).sample(model, (hi - lo, 263, 1, frames), clip_denoised=False,
                                                                  model_kwargs={"y": shard_y(y, lo, hi)})
        finally:
            diffusion.sample_base = 0

    whole = run(0, B)
    assert torch.isfinite(whole).all() and torch.equal(whole, run(0, B))
    assert not torch.equal(whole, run(0, B, base_seed=700))
    parts = torch.cat([run(0, 12), run(12, B)])
    d = maxabs(parts.cpu(), whole.cpu())
    print(f"[parity] DiP B=32 as 12 + 20 shards vs one batch, {prec}: max-abs = {d:.3e}")
    assert torch.equal(parts, whole)

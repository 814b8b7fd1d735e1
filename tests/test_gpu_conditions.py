"""GPU parity tests (`-m gpu`) of seam B1's other conditions against the UPSTREAM REFERENCE's own outputs (tests/golden/a2m_*,
target_*, dip_target_*.npz; oracle/make_golden_r6b.py): cond_mode='action' (model/mdm.py:224-226, :389-397 -- the action-to-motion
checkpoints, 25 joints x 6 rot6d features) and --multi_target_cond (model/mdm.py:197-199, :399-479 -- the target-conditioned DiP of
DiP.md:105), full-size models (latent_dim 512, 8 layers), both arithmetic modes, through the Python seams -> C ABI -> HIP kernels."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import dip, make_pair, maxabs, memo, orc, synth_dip_state_dict, synth_dip_y, synth_state_dict, synth_y, to_dev
from oracle.synth import synth_a2m_state_dict, synth_target_params, synth_target_y

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL_FWD, TOL_LOOP, TOL_DIP_AR = 1e-4, 1e-4, 2e-4     # the tolerances of the text-conditioned suites (north_star: 1e-3)
A2M = dict(dataset="humanact12", num_actions=12)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from mdm_amd import _native
    assert _native.load_native().path.endswith("libmdm_hip.so")


def a2m_y(B, T, seed, lengths, action, scale=2.5):
    y = synth_y(B, T, seed, lengths=lengths, scale=scale)
    del y["text_embed"]
    y["action"] = torch.as_tensor(action)
    assert torch.equal(y["action"], torch.randint(0, 12, (B, 1), generator=torch.Generator().manual_seed(seed + 5)))
    return y


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_action_conditioned_forward_matches_reference_golden(golden_dir, gemm_path, prec):
    if prec == "f32" and gemm_path != "small":
        pytest.skip("the f32 mode has one GEMM kernel")
    g = np.load(os.path.join(golden_dir, "a2m_fwd_B3_T60.npz"))
    B, T = 3, 60
    sd = memo("sd_a2m", lambda: synth_a2m_state_dict(seed=0))
    model, _ = make_pair(sd, 10, DEV, guided=True, precision=prec, **A2M)
    y = to_dev(a2m_y(B, T, int(g["y_seed"]), list(g["lengths"]), g["action"], float(g["scale"])), DEV)
    x = torch.randn(B, 25, 6, T, generator=torch.Generator().manual_seed(int(g["x_seed"]))).to(DEV)
    t = torch.as_tensor(g["t"]).to(DEV)
    errs = {"cond": maxabs(model.model(x, t, y=dict(y)).cpu(), g["out_cond"]),
            "uncond": maxabs(model.model(x, t, y={**y, "uncond": True}).cpu(), g["out_uncond"]),
            "cfg": maxabs(model(x, t, y=dict(y)).cpu(), g["out_cfg"])}
    print(f"[parity] a2m_fwd_B3_T60 {prec} {gemm_path}: max-abs vs reference = {errs}")
    assert max(errs.values()) < TOL_FWD, errs


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_action_conditioned_loop_matches_reference_golden(golden_dir, prec):
    """10-step p_sample_loop over the reference's CPU noise stream: under the guidance wrapper and as the bare model
    (sample/generate.py:93-94 wraps only when guidance_param != 1)."""
    g = np.load(os.path.join(golden_dir, "a2m_loop10_B2_T60.npz"))
    steps, B, T, seed = int(g["steps"]), int(g["B"]), int(g["T"]), int(g["seed"])
    sd = memo("sd_a2m", lambda: synth_a2m_state_dict(seed=0))
    shape = (B, 25, 6, T)
    x_T, noises = orc.make_noise(shape, steps, seed)
    seq = [x_T] + [n.contiguous() for n in noises]
    y = to_dev(a2m_y(B, T, seed + 1000, list(g["lengths"]), g["action"], float(g["scale"])), DEV)
    model, diffusion = make_pair(sd, steps, DEV, guided=True, precision=prec, **A2M)
    for name, mdl in (("cfg", model), ("nocfg", model.model)):
        out = diffusion.p_sample_loop(mdl, shape, clip_denoised=False, model_kwargs={"y": dict(y)}, noise_sequence=seq)
        err = maxabs(out.cpu(), g["final_" + name])
        print(f"[parity] a2m_loop10_B2_T60 {name} {prec}: max-abs vs reference = {err:.3e}")
        assert err < TOL_LOOP, (name, err)


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_target_condition_on_the_encoder_matches_reference_golden(golden_dir, gemm_path, prec):
    if prec == "f32" and gemm_path != "small":
        pytest.skip("the f32 mode has one GEMM kernel")
    g = np.load(os.path.join(golden_dir, "target_enc_fwd_B4_T48.npz"))
    B, T = 4, 48
    sd = memo("sd_tgt_enc", lambda: {**synth_state_dict(seed=0), **synth_target_params("single", seed=0)})
    model, _ = make_pair(sd, 10, DEV, guided=True, precision=prec, multi_target_cond=True, multi_encoder_type="single")
    ys = int(g["y_seed"])
    y = to_dev({**synth_y(B, T, seed=ys, lengths=list(g["lengths"]), scale=float(g["scale"])), **synth_target_y(B, seed=ys)}, DEV)
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(int(g["x_seed"]))).to(DEV)
    t = torch.as_tensor(g["t"]).to(DEV)
    errs = {"cond": maxabs(model.model(x, t, y=dict(y)).cpu(), g["out_cond"]),
            "cfg": maxabs(model(x, t, y=dict(y)).cpu(), g["out_cfg"]),
            "target_uncond": maxabs(model.model(x, t, y={**y, "target_uncond": True}).cpu(), g["out_target_uncond"])}
    print(f"[parity] target_enc_fwd_B4_T48 {prec} {gemm_path}: max-abs vs reference = {errs}")
    assert max(errs.values()) < TOL_FWD, errs
    assert maxabs(g["out_cond"], g["out_target_uncond"]) > 100 * TOL_FWD       # (the fixture's target moves the output: 2.6e-2)


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
@pytest.mark.parametrize("kind", ["single", "split", "multi"])
def test_target_conditioned_dip_forward_matches_reference_golden(golden_dir, kind, prec):
    g = np.load(os.path.join(golden_dir, "dip_target_fwd_B4.npz"))
    B = 4
    sd = memo("sd_tgt_dip_" + kind, lambda: {**synth_dip_state_dict(seed=0), **synth_target_params(kind, seed=0)})
    model, _ = make_pair(sd, 10, DEV, guided=True, context_len=20, pred_len=40, mask_frames=False, precision=prec,
                         multi_target_cond=True, multi_encoder_type=kind)
    ys = int(g["y_seed"])
    y = to_dev({**synth_dip_y(B, 40, 20, seed=ys, text_lengths=list(g["text_lengths"])), **synth_target_y(B, seed=ys)}, DEV)
    x = torch.randn(B, 263, 1, 40, generator=torch.Generator().manual_seed(int(g["x_seed"]))).to(DEV)
    t = torch.as_tensor(g["t"]).to(DEV)
    errs = {"cond": maxabs(model.model(x, t, y=dict(y)).cpu(), g["out_cond_" + kind]),
            "uncond": maxabs(model.model(x, t, y={**y, "uncond": True}).cpu(), g["out_uncond_" + kind])}
    print(f"[parity] dip_target_fwd_B4 {kind} {prec}: max-abs vs reference = {errs}")
    assert max(errs.values()) < TOL_FWD, errs


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_target_conditioned_dip_autoregressive_matches_reference_golden(golden_dir, prec):
    """The reference's AutoRegressiveSampler over two 40-frame windows of the target-conditioned DiP, CFG 7.5: here through this
    repository's sampler and mdm_sample_loop_dec (target folded into the hoisted text memory of each window)."""
    from mdm_amd.sampler_util import AutoRegressiveSampler
    g = np.load(os.path.join(golden_dir, "dip_target_ar10_B2_F80.npz"))
    steps, B, frames, seed = int(g["steps"]), int(g["B"]), int(g["frames"]), int(g["seed"])
    sd = memo("sd_tgt_dip_single", lambda: {**synth_dip_state_dict(seed=0), **synth_target_params("single", seed=0)})
    model, diffusion = make_pair(sd, steps, DEV, guided=True, context_len=20, pred_len=40, mask_frames=False, precision=prec,
                                 multi_target_cond=True, multi_encoder_type="single")
    y = {**synth_dip_y(B, 40, 20, seed=int(g["y_seed"]), text_lengths=list(g["text_lengths"]), scale=float(g["scale"])),
         **synth_target_y(B, seed=seed, first=1)}
    y = to_dev({k: v for k, v in y.items() if k != "text"}, DEV)
    chunks = iter(dip.make_noise_chunks((B, 263, 1, 40), steps, seed, 2))

    def sample_fn(mdl, shape, **kw):
        x_T, eps = next(chunks)
        return diffusion.p_sample_loop(mdl, shape, noise_sequence=[x_T] + [e.contiguous() for e in eps], **kw)

    args = SimpleNamespace(pred_len=40, context_len=20, autoregressive_include_prefix=False)
    out = AutoRegressiveSampler(args, sample_fn, frames).sample(
        model, (B, 263, 1, frames), clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0, init_image=None,
        progress=False, dump_steps=None, noise=None, const_noise=False)
    err = maxabs(out.cpu(), g["final"])
    print(f"[parity] dip_target_ar10_B2_F80 {prec}: max-abs vs reference = {err:.3e}")
    assert out.shape == (B, 263, 1, frames) and err < TOL_DIP_AR


def test_target_condition_is_shard_invariant_and_fills_the_batch():
    """SURVEY 8e with a target: B = 32 windows of the target-conditioned DiP as one batch and as the shards 12 + 20 (`shard_y` slices
    the per-sample target entries with the batch) -- equal bit for bit."""
    from mdm_amd.dist import shard_y
    B, steps, C, P = 32, 10, 20, 40
    sd = memo("sd_tgt_dip_single", lambda: {**synth_dip_state_dict(seed=0), **synth_target_params("single", seed=0)})
    model, diffusion = make_pair(sd, steps, DEV, guided=True, context_len=C, pred_len=P, mask_frames=True,
                                 multi_target_cond=True, multi_encoder_type="single")
    tl = [2 + (5 * i) % 23 for i in range(B)]
    y = to_dev({**synth_dip_y(B, P, C, seed=43, text_lengths=tl, lengths=[40 - (3 * i) % 17 for i in range(B)], scale=7.5),
                **synth_target_y(B, seed=43)}, DEV)

    def run(lo, hi):
        diffusion.sample_base = lo
        try:
            return diffusion.p_sample_loop(model, (hi - lo, 263, 1, P), clip_denoised=False, model_kwargs={"y": shard_y(y, lo, hi)},
                                           seed=600)
        finally:
            diffusion.sample_base = 0
    whole = run(0, B)
    assert torch.isfinite(whole).all()
    assert torch.equal(torch.cat([run(0, 12), run(12, B)]), whole)
    y_off = {**y, "target_uncond": True}
    diffusion.sample_base = 0
    other = diffusion.p_sample_loop(model, (B, 263, 1, P), clip_denoised=False, model_kwargs={"y": y_off}, seed=600)
    assert maxabs(other.cpu(), whole.cpu()) > 1e-3


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
@pytest.mark.parametrize("route", ["row_tiles", "sequence_tiles"])
def test_full_length_trans_dec_matches_reference_golden(golden_dir, engine_options, route, prec):
    """README.md:254 `humanml_trans_dec_512_bert-50steps` -- the trans_dec + DistilBERT denoiser OUTSIDE DiP's windows (no prefix, 196
    frames, sample/generate.py's plain p_sample_loop): forward (cond / CFG) and a 10-step guided loop against the reference's own run.
    Both GEMM routes of csrc/decoder.h: gemm_x3s.h's row tiles (up to 80 sequences) and gemm_x3.h's sequence-sized tiles (what a large
    batch runs; forced at 4 sequences with small_gemm_max_seqs = 1)."""
    if prec == "f32" and route != "row_tiles":
        pytest.skip("the f32 mode has one GEMM kernel")
    engine_options(**({"small_gemm_max_seqs": 1} if route == "sequence_tiles" else {}))
    g = np.load(os.path.join(golden_dir, "transdec_B2_T196.npz"))
    B, T, steps, seed = 2, 196, int(g["steps"]), int(g["seed"])
    model, diffusion = make_pair(memo("sd_dip0", lambda: synth_dip_state_dict(seed=0)), steps, DEV, guided=True, context_len=0,
                                 pred_len=0, mask_frames=True, precision=prec)
    y = synth_dip_y(B, T, 1, seed=int(g["y_seed"]), text_lengths=list(g["text_lengths"]), lengths=list(g["lengths"]), scale=float(g["scale"]))
    y.pop("prefix")
    y = to_dev(y, DEV)
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(int(g["x_seed"]))).to(DEV)
    t = torch.as_tensor(g["t"]).to(DEV)
    errs = {"cond": maxabs(model.model(x, t, y=dict(y)).cpu(), g["out_cond"]), "cfg": maxabs(model(x, t, y=dict(y)).cpu(), g["out_cfg"])}
    shape = (B, 263, 1, T)
    x_T, noises = orc.make_noise(shape, steps, seed)
    out = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)},
                                  noise_sequence=[x_T] + [n.contiguous() for n in noises])
    errs["loop10"] = maxabs(out.cpu(), g["final"])
    print(f"[parity] transdec_B2_T196 {route} {prec}: max-abs vs reference = {errs}")
    assert max(errs.values()) < TOL_LOOP, errs


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_decoder_with_class_token_matches_reference_golden(golden_dir, prec):
    """README `humanml-decoder-with-emb-512` (`--arch trans_dec --emb_trans_dec`, CLIP memory): forward (cond / uncond / CFG, ragged frame
    masks, per-sample timesteps) and a 10-step guided loop against the reference's own run (tests/golden/decemb_B3_T60.npz)."""
    g = np.load(os.path.join(golden_dir, "decemb_B3_T60.npz"))
    B, T, steps, seed = 3, 60, int(g["steps"]), int(g["seed"])
    sd = memo("sd_dip_clip", lambda: synth_dip_state_dict(seed=0, bert_dim=512))
    model, diffusion = make_pair(sd, steps, DEV, guided=True, precision=prec, text_encoder_type="clip", emb_trans_dec=True, mask_frames=True)
    y = to_dev(synth_y(B, T, seed=int(g["y_seed"]), lengths=list(g["lengths"]), scale=float(g["scale"])), DEV)
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(int(g["x_seed"]))).to(DEV)
    t = torch.as_tensor(g["t"]).to(DEV)
    errs = {"cond": maxabs(model.model(x, t, y=dict(y)).cpu(), g["out_cond"]),
            "uncond": maxabs(model.model(x, t, y={**y, "uncond": True}).cpu(), g["out_uncond"]),
            "cfg": maxabs(model(x, t, y=dict(y)).cpu(), g["out_cfg"])}
    shape = (B, 263, 1, T)
    x_T, noises = orc.make_noise(shape, steps, seed)
    out = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)},
                                  noise_sequence=[x_T] + [n.contiguous() for n in noises])
    errs["loop10"] = maxabs(out.cpu(), g["final"])
    print(f"[parity] decemb_B3_T60 {prec}: max-abs vs reference = {errs}")
    assert max(errs.values()) < TOL_LOOP, errs


def test_full_length_trans_dec_routes_agree_at_the_headline_batch(engine_options):
    """The full-length trans_dec denoiser at B = 128 under guidance (256 sequences of 196 tokens, ragged frame masks and prompts,
    production Philox noise, 10 steps): the sequence-tile route -- with layer 0's self-attention block shared between the branches and
    no cross-attention work for the unconditional half (csrc/decoder.h) -- against the row-tile route, which takes neither shortcut:
    two different arithmetic paths over the whole batch agree to the loop tolerance; and the sequence-tile run is reproducible bit for bit."""
    B, T, steps = 128, 196, 10
    sd = memo("sd_dip0", lambda: synth_dip_state_dict(seed=0))
    g = torch.Generator().manual_seed(5)
    tl = [int(v) for v in torch.randint(3, 25, (B,), generator=g)]
    y = synth_dip_y(B, T, 1, seed=77, text_lengths=tl, lengths=[196 - (11 * i) % 157 for i in range(B)], scale=2.5)
    y.pop("prefix")
    y = to_dev(y, DEV)
    outs = {}
    for route, opts in (("sequence_tiles", {}), ("row_tiles", {"small_gemm_max_seqs": 100000})):
        engine_options(**opts)
        model, diffusion = make_pair(sd, steps, DEV, guided=True, context_len=0, pred_len=0, mask_frames=True)
        run = lambda: diffusion.p_sample_loop(model, (B, 263, 1, T), clip_denoised=False, model_kwargs={"y": dict(y)}, seed=31)  # noqa: E731
        outs[route] = run()
        if route == "sequence_tiles":
            assert torch.equal(run(), outs[route])
    assert torch.isfinite(outs["row_tiles"]).all()
    err = maxabs(outs["sequence_tiles"].cpu(), outs["row_tiles"].cpu())
    print(f"[parity] transdec B=128 T=196 10 steps, sequence tiles (shared layer-0 block, no uncond cross-attention) vs row tiles: {err:.3e}")
    assert err < TOL_LOOP and not torch.equal(outs["sequence_tiles"], outs["row_tiles"])

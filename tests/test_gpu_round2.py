"""Round-2 GPU parity tests (`-m gpu`): the BASELINE.json configurations at their REAL sizes with multi-sample oracle replay,
non-identity timestep maps, the progressive generator and const_noise against the reference's own fixtures
(oracle/make_golden_r2.py), "trained-like" hostile weights, and the numeric envelope of the fp16 operand planes."""
import os

import numpy as np
import pytest
import torch

from helpers import make_pair, maxabs, memo, orc, synth_state_dict, synth_y
from oracle.synth import synth_state_dict_hostile, synth_y_hostile

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
PRECISIONS = ["f16x3", "f32"]
TOL_LOOP = {"f32": 1e-4, "f16x3": 1e-4}      # BASELINE's bar: 1e-3
TOL_FWD = {"f32": 2e-5, "f16x3": 3e-5}


@pytest.fixture(scope="module")
def sd():
    return synth_state_dict(seed=0)


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def _replay(sd, model, y, idx, seed, steps, T, tab=None, **kw):
    """Samples `idx` of a Philox-driven batch, recomputed by the oracle on the library's own noise for those samples."""
    eng = model.model.engine() if hasattr(model, "model") else model.engine()
    n = len(idx)
    seq = [torch.cat([eng.randn((1, 263, 1, T), DEV, seed, i, k).cpu() for i in idx]) for k in range(steps + 1)]
    ys = {"mask": y["mask"][idx], "lengths": y["lengths"][idx], "text_embed": y["text_embed"][:, idx], "scale": y["scale"][idx]}
    tab = tab or orc.Tables(orc.named_betas("cosine", steps))
    return orc.sample_loop(sd, tab, (n, 263, 1, T), ys, seq[0], seq[1:], cfg=True, **kw)


# ---------------------------------------------------------------------------------------------------
# BASELINE.json configs[1] and configs[2] at their real sizes
# ---------------------------------------------------------------------------------------------------
def test_config1_B128_T196_50_steps_eight_samples_replayed_through_the_oracle(sd):
    """configs[1] exactly: B=128, T=196, 50-step p_sample_loop, CFG 2.5, mixed lengths, the production Philox stream; eight
    samples spread over the batch (first / last rows of the GEMM tile grid, short and full sequences) are recomputed by the
    oracle on the same noise.  Both arithmetic modes against ONE oracle run (the noise does not depend on the mode)."""
    steps, B, T, seed = 50, 128, 196, 4242
    shape = (B, 263, 1, T)
    lengths = [196 - (11 * i) % 157 for i in range(B)]
    y = synth_y(B, T, seed=19, lengths=lengths)
    idx = [0, 17, 31, 50, 64, 89, 101, 127]
    outs = {}
    for prec in PRECISIONS:
        model, diffusion = make_pair(sd, steps, DEV, guided=True, precision=prec)
        outs[prec] = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)}, seed=seed).cpu()
        assert torch.isfinite(outs[prec]).all()
    want = _replay(sd, model, y, idx, seed, steps, T)
    for prec in PRECISIONS:
        err = maxabs(outs[prec][idx], want)
        print(f"[parity] configs[1] B=128 T=196 50 steps, 8 samples replayed, {prec}: max-abs vs oracle = {err:.3e}")
        assert err < TOL_LOOP[prec]
    # the other 120 samples: the two independent arithmetic modes (exact-fp32 MFMA vs the fp16 split) agree over the WHOLE batch
    cross = maxabs(outs["f16x3"], outs["f32"])
    print(f"[parity] configs[1] all 128 samples, f16x3 vs f32 mode: max-abs = {cross:.3e}")
    assert cross < TOL_LOOP["f16x3"]


def test_config2_B64_T196_1000_steps_replayed_through_the_oracle(sd):
    """configs[2]: 1000-step DDPM, B=64, T=196 (the step-fusion stress).  One sample is recomputed by the oracle over all 1000
    steps (~0.2 s of CPU per step); the other 63 are covered by the agreement of the two independent arithmetic modes."""
    steps, B, T, seed = 1000, 64, 196, 99
    shape = (B, 263, 1, T)
    y = synth_y(B, T, seed=23, lengths=[196 - (13 * i) % 120 for i in range(B)])
    idx = [40]
    outs = {}
    for prec in PRECISIONS:
        model, diffusion = make_pair(sd, steps, DEV, guided=True, precision=prec)
        outs[prec] = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)}, seed=seed).cpu()
    want = _replay(sd, model, y, idx, seed, steps, T)
    for prec in PRECISIONS:
        err = maxabs(outs[prec][idx], want)
        print(f"[parity] configs[2] B=64 T=196 1000 steps, sample 40 replayed, {prec}: max-abs vs oracle = {err:.3e}")
        assert err < TOL_LOOP[prec]
    cross = maxabs(outs["f16x3"], outs["f32"])
    print(f"[parity] configs[2] all 64 samples, f16x3 vs f32 mode: max-abs = {cross:.3e}")
    assert cross < TOL_LOOP["f16x3"]


# ---------------------------------------------------------------------------------------------------
# rows a7 / a4 of SURVEY 8a against the reference's own fixtures
# ---------------------------------------------------------------------------------------------------
def _respaced_pair(sd, spacing, precision):
    from mdm_amd import gaussian_diffusion as gd
    from mdm_amd.respace import SpacedDiffusion, space_timesteps
    model, _ = make_pair(sd, 50, DEV, guided=True, precision=precision)
    diffusion = SpacedDiffusion(use_timesteps=space_timesteps(1000, spacing),
                                betas=gd.get_named_beta_schedule("cosine", 1000, 1.0),
                                model_mean_type=gd.ModelMeanType.START_X, model_var_type=gd.ModelVarType.FIXED_SMALL,
                                loss_type=gd.LossType.MSE, rescale_timesteps=False)
    return model, diffusion


@pytest.mark.parametrize("name", ["respaced_ddim50of1000_B2_T64", "respaced_p50of1000_B2_T64"])
@pytest.mark.parametrize("prec", PRECISIONS)
def test_respaced_timestep_map_matches_reference(golden_dir, sd, name, prec):
    """A non-identity `timestep_map` (respace.py:125-130) handed to the native loop: 50 of 1000 steps, DDIM and DDPM."""
    g = _g(golden_dir, name)
    B, T, seed = int(g["B"]), int(g["T"]), int(g["seed"])
    spacing = str(g["spacing"])
    model, diffusion = _respaced_pair(sd, spacing if spacing.startswith("ddim") else [int(spacing)], prec)
    assert diffusion.timestep_map == [int(v) for v in g["timestep_map"]] and diffusion.num_timesteps == 50
    shape = (B, 263, 1, T)
    y = synth_y(B, T, seed=seed + 1000, lengths=list(g["lengths"]))
    x_T, noises = orc.make_noise(shape, 50, seed)
    fn = diffusion.ddim_sample_loop if bool(g["ddim"]) else diffusion.p_sample_loop
    out = fn(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)}, noise_sequence=[x_T] + [n.contiguous() for n in noises])
    err = maxabs(out.cpu(), g["final"])
    print(f"[parity] {name} {prec}: max-abs vs reference = {err:.3e}")
    assert err < TOL_LOOP[prec]


@pytest.mark.parametrize("prec", PRECISIONS)
def test_progressive_generator_matches_every_reference_yield(golden_dir, sd, prec):
    """p_sample_loop_progressive (gaussian_diffusion.py:660-727) with the reference's injected noise stream: every yielded
    `sample` and `pred_xstart` against the reference's own yields."""
    g = _g(golden_dir, "progressive8_B2_T24")
    steps, B, T, seed = int(g["steps"]), int(g["B"]), int(g["T"]), int(g["seed"])
    shape = (B, 263, 1, T)
    y = synth_y(B, T, seed=seed + 1000, lengths=list(g["lengths"]))
    x_T, noises = orc.make_noise(shape, steps, seed)
    model, diffusion = make_pair(sd, steps, DEV, guided=True, precision=prec)
    outs = list(diffusion.p_sample_loop_progressive(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)},
                                                    noise_sequence=[x_T] + [n.contiguous() for n in noises]))
    assert len(outs) == steps
    for k, o in enumerate(outs):
        assert maxabs(o["sample"].cpu(), g["samples"][k]) < TOL_LOOP[prec]
        assert maxabs(o["pred_xstart"].cpu(), g["pred_xstart"][k]) < TOL_LOOP[prec]


def test_const_noise_matches_reference_and_broadcasts_sample_zero(golden_dir, sd):
    """p_sample_loop(const_noise=True) (gaussian_diffusion.py:527-528): the reference's fixture with the injected stream, and
    on the Philox stream every sample must receive the step noise of sample 0."""
    g = _g(golden_dir, "const_noise50_B3_T32")
    steps, B, T, seed = int(g["steps"]), int(g["B"]), int(g["T"]), int(g["seed"])
    shape = (B, 263, 1, T)
    y = synth_y(B, T, seed=seed + 1000, lengths=list(g["lengths"]))
    x_T, noises = orc.make_noise(shape, steps, seed)
    model, diffusion = make_pair(sd, steps, DEV, guided=True)
    out = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)}, const_noise=True,
                                  noise_sequence=[x_T] + [n.contiguous() for n in noises])
    assert maxabs(out.cpu(), g["final"]) < TOL_LOOP["f16x3"]
    # Philox: eps = (x_prev - a_x0 x0 - a_xt x_t) / sigma is the same for every sample
    x = torch.randn(shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    t = torch.full((B,), 30, device=DEV, dtype=torch.long)
    a_x0, a_xt, sigma = diffusion.ddpm_coefficients()
    for const in (True, False):
        o = diffusion.p_sample(model, x, t, clip_denoised=False, model_kwargs={"y": dict(y)}, const_noise=const)
        eps = (o["sample"] - float(a_x0[30]) * o["pred_xstart"] - float(a_xt[30]) * x) / float(sigma[30])
        same = float((eps[1:] - eps[:1]).abs().max())
        assert (same < 1e-4) if const else (same > 1.0)
    with pytest.raises(NotImplementedError):
        diffusion.ddim_sample_loop(model, shape, const_noise=True, model_kwargs={"y": dict(y)})    # as the reference (:902-903)


# ---------------------------------------------------------------------------------------------------
# "trained-like" hostile weights
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", PRECISIONS)
def test_hostile_weights_forward_and_loop(gemm_path, golden_dir, prec):
    """Outlier channels (|beta|, bias 50-300x), LayerNorm gamma in [0.05, 8], 10x weight rows, 20x text embedding
    (oracle/synth.py synth_state_dict_hostile).  These weights amplify rounding noise, so the fixtures record the reference
    arithmetic's OWN noise `floor` = |reference fp32 - fp64 oracle|; the bars are the unchanged tolerances or a multiple of
    that floor, whichever is larger -- measured against fp64 truth for the loop: 3 floors for the exact-fp32 mode, 6 for the
    split mode (22-bit products against fp32's 24: a factor 4 by construction).  The round-1 bf16 split is ~100 floors here
    (tools/precision_probe.py --hostile); the folded LayerNorm's statistics are merged Chan-style (gemm_x3.h)."""
    if prec == "f32" and gemm_path != "small":
        pytest.skip("the f32 mode has one GEMM kernel")
    K = {"f32": 3.0, "f16x3": 6.0}[prec]
    sdh = synth_state_dict_hostile(0)
    g = _g(golden_dir, "hostile_fwd_B2_T196")
    B, T = 2, 196
    y = synth_y_hostile(B, T, seed=int(g["y_seed"]), lengths=list(g["lengths"]))
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    t = torch.from_numpy(g["t"])
    model, _ = make_pair(sdh, 50, DEV, guided=True, precision=prec)
    oc = model.model(x.to(DEV), t.to(DEV), y=dict(y)).cpu()
    og = model(x.to(DEV), t.to(DEV), y=dict(y)).cpu()
    e_c, e_g = maxabs(oc, g["out_cond"]), maxabs(og, g["out_cfg"])
    print(f"[parity] hostile fwd {prec}: cond {e_c:.3e} (floor {float(g['floor_cond']):.1e}), cfg {e_g:.3e} (floor {float(g['floor_cfg']):.1e})")
    assert e_c < max(TOL_FWD[prec], K * float(g["floor_cond"]))
    assert e_g < max(4 * TOL_FWD[prec], K * float(g["floor_cfg"]))
    assert e_c < 1e-3 and e_g < 1e-3                             # the bar that matters (BASELINE.json: 1e-3 max-abs), in absolute terms
    g = _g(golden_dir, "hostile_loop50_B2_T196")
    steps, seed = int(g["steps"]), int(g["seed"])
    shape = (B, 263, 1, T)
    y = synth_y_hostile(B, T, seed=seed + 1000, lengths=list(g["lengths"]))
    x_T, noises = orc.make_noise(shape, steps, seed)
    model, diffusion = make_pair(sdh, steps, DEV, guided=True, precision=prec)
    out = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)},
                                  noise_sequence=[x_T] + [n.contiguous() for n in noises]).cpu()
    truth = memo("hostile_truth", lambda: orc.sample_loop(sdh, orc.Tables(orc.named_betas("cosine", steps)), shape, y, x_T, noises,
                                                          cfg=True, dtype=torch.float64))
    floor = float(g["floor"])
    e_ref, e_64 = maxabs(out, g["final"]), maxabs(out, truth)
    print(f"[parity] hostile loop50 {prec}: vs reference {e_ref:.3e}, vs fp64 {e_64:.3e} (reference's own {floor:.1e}; |x0| max "
          f"{float(np.abs(g['final']).max()):.1f})")
    assert e_64 < max(TOL_LOOP[prec], K * floor) and e_ref < max(TOL_LOOP[prec], (K + 1) * floor)
    assert e_64 < 1e-3 and e_ref < 1e-3                          # BASELINE's bar in absolute terms, even on these weights


# ---------------------------------------------------------------------------------------------------
# the fp16 operand planes: subnormals, range
# ---------------------------------------------------------------------------------------------------
def test_fp16_planes_keep_subnormals_and_fail_loudly_out_of_range(sd):
    """mdm_linear_x3 against fp64 on operands far outside the comfortable range: tiny activations (the lo plane is entirely
    fp16-subnormal: the MFMA must not flush it), large ones inside fp16's range -- and beyond it (|x| > 65504) the result
    must be non-finite, not a plausible wrong number; the sampler seam turns that into an actionable error."""
    from mdm_amd import _native
    lib = _native.load_native()
    M, N, K = 197 * 4, 512, 512
    g = torch.Generator().manual_seed(0)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.zeros(N)
    nb = lib.mdm_linear_x3_scratch_bytes(M, N, K)
    scratch = torch.empty(nb, dtype=torch.uint8, device=DEV)

    wd, bd = w.to(DEV), b.to(DEV)

    def run(a):
        ad, out = a.to(DEV), torch.empty(M, N, device=DEV)        # (kept alive across the asynchronous call)
        lib.check(lib.mdm_linear_x3(ad.data_ptr(), wd.data_ptr(), bd.data_ptr(), None, out.data_ptr(),
                                    M, N, K, 0, scratch.data_ptr(), nb, torch.cuda.current_stream().cuda_stream), "mdm_linear_x3")
        torch.cuda.synchronize()
        return out.cpu()

    for scale, rel in ((1.0, 2e-6), (1e-3, 3e-5), (3e-5, 1.5e-3), (1e4, 2e-6)):   # (flushed subnormals would give 3e-5 / 1)
        a = torch.randn(M, K, generator=g) * scale
        out = run(a)
        ref = a.double() @ w.double().t()
        err = float((out.double() - ref).abs().max() / ref.abs().max())
        print(f"[parity] mdm_linear_x3 operand scale {scale:g}: max error / max|out| = {err:.2e}")
        assert torch.isfinite(out).all() and err < rel
    a = torch.randn(M, K, generator=g)
    a[5, 7] = 1.0e5                                           # beyond fp16's largest finite value
    assert not torch.isfinite(run(a)[5]).all()
    # ... and through the seams: a checkpoint whose activations overflow is reported, with the way out
    bad = {k: v.clone() for k, v in sd.items()}
    bad["input_process.poseEmbedding.bias"][3] = 2.0e5
    model, diffusion = make_pair(bad, 4, DEV, guided=True)
    y = synth_y(2, 16, seed=1)
    with pytest.raises(FloatingPointError, match="precision='f32'"):
        diffusion.p_sample_loop(model, (2, 263, 1, 16), clip_denoised=False, model_kwargs={"y": dict(y)}, seed=1)
    model, diffusion = make_pair(bad, 4, DEV, guided=True, precision="f32")
    assert torch.isfinite(diffusion.p_sample_loop(model, (2, 263, 1, 16), clip_denoised=False, model_kwargs={"y": dict(y)}, seed=1)).all()


# ---------------------------------------------------------------------------------------------------
# seams: masks
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", PRECISIONS)
def test_frame_masks_with_holes_are_honoured(gemm_path, sd, prec):
    """model/mdm.py:241-247 hands ANY `~y['mask']` to src_key_padding_mask; masks that are not prefix masks reach the attention
    kernels as per-sample bitmaps (include/mdm_hip.h lengths_dev, ABI 7).  T = 196 (seven key tiles), three samples: a prefix
    mask, a mask with holes incl. frame 0 and a tile boundary, a mask with only scattered frames; cond, uncond and guided."""
    if prec == "f32" and gemm_path != "small":
        pytest.skip("the f32 mode has one GEMM kernel")
    B, T = 3, 196
    y = synth_y(B, T, seed=2, lengths=[196, 150, 196])
    y["mask"] = y["mask"].clone()
    y["mask"][1, 0, 0, [0, 3, 31, 32, 63, 64, 95, 100, 149]] = False
    y["mask"][2] = False
    y["mask"][2, 0, 0, [1, 30, 33, 127, 128, 190, 195]] = True
    model, _ = make_pair(sd, 50, DEV, guided=True, precision=prec)
    g = torch.Generator().manual_seed(4)
    x, t = torch.randn(B, 263, 1, T, generator=g), torch.tensor([49, 7, 0])
    yd = {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in y.items()}
    want_c = memo("holes_c", lambda: orc.mdm_forward(sd, x, t, y))
    want_u = memo("holes_u", lambda: orc.mdm_forward(sd, x, t, {**y, "uncond": True}))
    assert maxabs(model.model(x.to(DEV), t.to(DEV), y=dict(yd)).cpu(), want_c) < TOL_FWD[prec]
    assert maxabs(model.model(x.to(DEV), t.to(DEV), y={**yd, "uncond": True}).cpu(), want_u) < TOL_FWD[prec]
    want_g = want_u + y["scale"].view(-1, 1, 1, 1) * (want_c - want_u)
    assert maxabs(model(x.to(DEV), t.to(DEV), y=dict(yd)).cpu(), want_g) < 3 * TOL_FWD[prec]


def test_broadcast_inpainting_mask_works(sd):
    B, T = 2, 16
    y = synth_y(B, T, seed=2, lengths=[16, 9])
    model, diffusion = make_pair(sd, 50, DEV, guided=True)
    x = torch.randn(B, 263, 1, T).to(DEV)
    t = torch.tensor([10, 10], device=DEV)
    yi = dict(y)
    m = torch.zeros(1, 263, 1, T, dtype=torch.bool)
    m[:, :4] = True
    yi["inpainting_mask"] = m                                 # broadcastable over the batch
    yi["inpainted_motion"] = torch.randn(B, 263, 1, T)
    o = diffusion.p_sample(model, x, t, clip_denoised=False, model_kwargs={"y": yi})
    assert torch.equal(o["pred_xstart"][:, :4].cpu(), yi["inpainted_motion"][:, :4])


def test_dip_dump_steps_are_loop_indices():
    """p_sample_loop(dump_steps=...) on the trans_dec path must snapshot by the loop's enumerate index k
    (gaussian_diffusion.py:637-655), like the fused loop -- not by the descending timestep.  The native window loop
    (mdm_sample_loop_dec: text K / V hoisted, time row added in the attention kernel) is compared with the step-at-a-time
    composition of the same library calls (the progressive generator, and diffusion.dip_stepwise): same values up to the
    re-association of the hoisted projection."""
    from helpers import synth_dip_state_dict, synth_dip_y, to_dev
    sdd = synth_dip_state_dict(seed=0)
    steps, B = 10, 2
    model, diffusion = make_pair(sdd, steps, DEV, guided=True, context_len=20, pred_len=40)
    y = to_dev(synth_dip_y(B, 40, 20, seed=4, text_lengths=[7, 12]), DEV)
    shape = (B, 263, 1, 40)
    x_T, noises = orc.make_noise(shape, steps, 5)
    seq = [x_T] + [n.contiguous() for n in noises]
    traj = [o["sample"].clone() for o in diffusion.p_sample_loop_progressive(model, shape, clip_denoised=False,
                                                                              model_kwargs={"y": y}, noise_sequence=seq)]
    dumps = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, noise_sequence=seq,
                                    dump_steps=[0, 3, 9])
    assert len(dumps) == 3
    for d, k in zip(dumps, (0, 3, 9)):
        assert maxabs(d.cpu(), traj[k].cpu()) < 2e-5
    assert maxabs(dumps[0].cpu(), traj[-1].cpu()) > 1e-2
    diffusion.dip_stepwise = True
    step = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, noise_sequence=seq,
                                   dump_steps=[0, 3, 9])
    for d, k in zip(step, (0, 3, 9)):
        assert torch.equal(d, traj[k])


@pytest.mark.parametrize("prec", PRECISIONS)
def test_kit_shape_251_features(prec):
    """dataset='kit' (utils/model_util.py:47-49): 251 pose features instead of 263 -- other K / N paddings of the 263-wide
    projections, other tail tiles in the transposing kernels.  Forward and a short guided loop against the oracle."""
    sdk = synth_state_dict(seed=0, input_feats=251)
    B, T, steps = 3, 50, 6
    model, diffusion = make_pair(sdk, steps, DEV, guided=True, precision=prec, dataset="kit")
    assert model.njoints == 251
    y = synth_y(B, T, seed=8, lengths=[50, 17, 33])
    x = torch.randn(B, 251, 1, T, generator=torch.Generator().manual_seed(2))
    t = torch.tensor([5, 0, 3])
    assert maxabs(model(x.to(DEV), t.to(DEV), y=dict(y)).cpu(), memo("kit_fwd", lambda: orc.cfg_forward(sdk, x, t, y))) < 4 * TOL_FWD[prec]
    shape = (B, 251, 1, T)
    g = torch.Generator().manual_seed(4)
    seq = [torch.randn(shape, generator=g) for _ in range(steps + 1)]
    out = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)}, noise_sequence=seq)
    want = memo("kit_loop", lambda: orc.sample_loop(sdk, orc.Tables(orc.named_betas("cosine", steps)), shape, y, seq[0], seq[1:], cfg=True))
    assert maxabs(out.cpu(), want) < TOL_LOOP[prec]

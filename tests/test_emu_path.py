"""The SAME kernel sources compiled for the CPU lock-step emulator (tests/emu/hip_emu.h -- test infrastructure:
there is no GPU in the build container) driven through the product's Python seams, against the oracle.  This
validates index arithmetic, masking, the MFMA fragment maps and the fused epilogues before code goes to the GPU;
it says nothing about speed and is never a product path."""
import sys
import os

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))

from emu_lib import emu, f32, ptr  # noqa: E402
from helpers import (dip, dip_small_state_dict, make_pair, maxabs, orc, small_state_dict, synth_dip_y,  # noqa: E402
                     synth_y)


@pytest.fixture(scope="module")
def lib():
    return emu()


def _skip_redundant(gemm_path, prec, T=0):
    """`gemm_path` (tests/conftest.py) picks one of the two split-precision encoder GEMM kernels: it does not exist in the f32 mode,
    and the 207 / 208-frame cases are edge shapes of gemm_x3.h's 208- / 224-row tiles only (the CPU suite's time is emulator time)."""
    if prec == "f32" and gemm_path != "small":
        pytest.skip("the f32 mode has one GEMM kernel")
    if T >= 207 and gemm_path == "small":
        pytest.skip("an edge shape of the sequence-tile kernel")


@pytest.mark.parametrize("prec,tol", [("f16x3", 2e-5), ("f32", 2e-5)])
def test_emulated_cfg_loop_matches_oracle(lib, gemm_path, prec, tol):
    _skip_redundant(gemm_path, prec)
    steps, B, T = 2, 2, 9
    sd = small_state_dict(num_layers=1)
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, precision=prec,
                                 pos_embed_max_len=5000)     # (the reference's table length; the other emulator cases bind 512 rows)
    y = synth_y(B, T, seed=5, lengths=[T, 4])
    shape = (B, 263, 1, T)
    x_T, noises = orc.make_noise(shape, steps, 11)
    got = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)},
                                  noise_sequence=[x_T] + [n.contiguous() for n in noises])
    want = orc.sample_loop(sd, orc.Tables(orc.named_betas("cosine", steps)), shape, y, x_T, noises, cfg=True,
                           num_heads=2)
    assert maxabs(got, want) < tol


@pytest.mark.parametrize("prec,tol,layers,B,T,lengths", [
    ("f32", 1e-5, 1, 2, 33, [33, 5]),
    ("f16x3", 2e-5, 2, 2, 33, [33, 5]),        # S = 34: two key tiles, ragged tail; six sequences per 208-row GEMM tile
    ("f16x3", 2e-5, 2, 1, 196, [150]),         # S = 197 (headline): one sequence per 208-row tile, 16-row last sub-tile
    ("f16x3", 2e-5, 2, 2, 100, [100, 100]),    # S = 101, D = 256: in_proj's second tile starts on an odd row, i.e. 8-byte aligned in
                                               # the one-partial-per-row statistics array (round 4: its last row read row M-2's)
    ("f16x3", 2e-5, 1, 1, 207, [207]),         # S = 208: the 16-row sub-tile completely used
    ("f16x3", 2e-5, 1, 1, 208, [208]),         # S = 209: does not fit 208 rows -> the 224-row (7 x 32) form
])
def test_emulated_forward_branches(lib, gemm_path, prec, tol, layers, B, T, lengths):
    """layers = 2 reaches the GEMM kinds only a second layer uses: in_proj with the previous LayerNorm folded in, and
    out_proj whose residual is a LayerNorm rebuilt from the pre-norm planes + row statistics."""
    _skip_redundant(gemm_path, prec, T)
    sd = small_state_dict(num_layers=layers)
    model, _ = make_pair(sd, 50, "cpu", guided=False, native_lib=lib, precision=prec)
    y = synth_y(B, T, seed=2, lengths=lengths)
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(B, 263, 1, T, generator=g), torch.tensor([49, 0][:B])
    assert maxabs(model(x, t, y=dict(y)), orc.mdm_forward(sd, x, t, y, num_heads=2)) < tol
    yu = {**y, "uncond": True}
    assert maxabs(model(x, t, y=yu), orc.mdm_forward(sd, x, t, yu, num_heads=2)) < tol


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_emulated_frame_masks_with_holes(lib, gemm_path, prec):
    """Arbitrary key-padding masks (model/mdm.py:241-247) through the bitmap form of `lengths` (include/mdm_hip.h, ABI 7):
    S = 41 (two key tiles): a prefix-mask sample, a sample with holes at frame 0 / across the tile boundary, a sparse sample."""
    _skip_redundant(gemm_path, prec)
    B, T = 3, 40
    sd = small_state_dict(num_layers=1)
    model, _ = make_pair(sd, 50, "cpu", guided=False, native_lib=lib, precision=prec)
    y = synth_y(B, T, seed=2, lengths=[T, 33, T])
    y["mask"] = y["mask"].clone()
    y["mask"][1, 0, 0, [0, 5, 30, 31, 32]] = False
    y["mask"][2] = False
    y["mask"][2, 0, 0, [2, 31, 39]] = True
    from mdm_amd.mdm import MDM
    ext = MDM.frame_mask_lengths(y["mask"].reshape(B, T))
    assert ext.numel() == 9 * B and ext[:B].tolist() == [T, -1, -1]
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(B, 263, 1, T, generator=g), torch.tensor([49, 0, 13])
    assert maxabs(model(x, t, y=dict(y)), orc.mdm_forward(sd, x, t, y, num_heads=2)) < 2e-5


def test_p_sample_takes_per_sample_timesteps(lib):
    """gaussian_diffusion.py:489-541: p_sample gathers its coefficients per sample, so `t` may differ inside a batch (a loop
    never does that; the public method may be called so).  One guided denoiser evaluation of the mixed-t batch + the fused step
    per sample, against the oracle with the same injected noise; and the t = 0 sample receives no noise."""
    steps, B, T = 50, 3, 9
    sd = small_state_dict(num_layers=1)
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib)
    y = synth_y(B, T, seed=3, lengths=[T, 5, 7])
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 263, 1, T, generator=g)
    eps = torch.randn(B, 263, 1, T, generator=g)
    t = torch.tensor([49, 0, 17])
    got = diffusion.p_sample(model, x, t, clip_denoised=False, model_kwargs={"y": dict(y)}, noise=eps)
    tab = orc.Tables(orc.named_betas("cosine", steps))
    x0 = orc.predict_x0(lambda a, b, c: orc.cfg_forward(sd, a, b, c, num_heads=2), x, t, y)
    want = orc.ddpm_step(tab, x, x0, t, eps)
    assert maxabs(got["pred_xstart"], x0) < 2e-5
    assert maxabs(got["sample"], want) < 2e-5
    assert maxabs(got["sample"][1], x0[1]) < 2e-5          # t = 0: the sample IS the predicted x0 (coef1 = 1, no noise)


def test_weight_beyond_the_fp16_planes_is_refused_at_bind_time(lib):
    """include/mdm_hip.h mdm_weights_in_range: the f16x3 planes hold w * 2^8 as fp16, so a weight (or a LayerNorm-gamma-folded
    weight) of magnitude >= 255.9 cannot be carried.  The pack kernels of mdm_prepare flag it; the Python seam raises at bind
    time in the default mode -- not NaN samples later -- and the same checkpoint runs in the exact-fp32 mode."""
    from mdm_amd._native import MdmError
    T = 9
    y = synth_y(1, T, seed=5, lengths=[T])
    x, t = torch.randn(1, 263, 1, T, generator=torch.Generator().manual_seed(0)), torch.tensor([3])
    # (a) a raw weight entry out of range
    sd = small_state_dict(num_layers=1)
    sd["seqTransEncoder.layers.0.linear2.weight"] = sd["seqTransEncoder.layers.0.linear2.weight"].clone()
    sd["seqTransEncoder.layers.0.linear2.weight"][3, 7] = 300.0
    model, _ = make_pair(sd, 50, "cpu", guided=False, native_lib=lib, precision="f16x3")
    with pytest.raises(MdmError, match="precision='f32'"):
        model(x, t, y=dict(y))
    model32, _ = make_pair(sd, 50, "cpu", guided=False, native_lib=lib, precision="f32")
    assert maxabs(model32(x, t, y=dict(y)), orc.mdm_forward(sd, x, t, y, num_heads=2)) < 2e-4
    model32.precision = "f16x3"                                  # switching the mode afterwards is refused as well
    with pytest.raises(MdmError, match="precision='f32'"):
        model32(x, t, y=dict(y))
    # (b) only the gamma-FOLDED weight leaves the range: |W| ~ 0.1 but the LayerNorm gamma it is folded with is 5000
    sd = small_state_dict(num_layers=1)
    sd["seqTransEncoder.layers.0.norm1.weight"] = torch.full_like(sd["seqTransEncoder.layers.0.norm1.weight"], 5000.0)
    model, _ = make_pair(sd, 50, "cpu", guided=False, native_lib=lib, precision="f16x3")
    with pytest.raises(MdmError, match="precision='f32'"):
        model(x, t, y=dict(y))


@pytest.mark.parametrize("M,N,K,act,res", [(70, 130, 36, 0, True), (129, 64, 8, 1, False), (3, 5, 4, 2, False)])
def test_emulated_linear(lib, M, N, K, act, res):
    rng = np.random.default_rng(M)
    a, w, b = f32(rng.standard_normal((M, K))), f32(rng.standard_normal((N, K))), f32(rng.standard_normal(N))
    r = f32(rng.standard_normal((M, N))) if res else None
    out = np.zeros((M, N), np.float32)
    lib.check(lib.mdm_linear(ptr(a), ptr(w), ptr(b), ptr(r) if res else None, ptr(out), M, N, K, act, None), "linear")
    ref = torch.from_numpy(a).double() @ torch.from_numpy(w).double().t() + torch.from_numpy(b).double()
    ref = torch.nn.functional.gelu(ref) if act == 1 else torch.nn.functional.silu(ref) if act == 2 else ref
    if res:
        ref = ref + torch.from_numpy(r).double()
    assert maxabs(out, ref.numpy()) < 1e-5


@pytest.mark.parametrize("M,N,K,act,res", [(130, 72, 64, 0, True), (5, 132, 32, 1, False), (260, 256, 96, 0, False)])
def test_emulated_linear_f16x3(lib, M, N, K, act, res):
    """LDS-DMA source swizzle, fragment reads and the 3-product accumulation of gemm_x3.h, incl. ragged M/N tiles."""
    rng = np.random.default_rng(M)
    a, w = f32(rng.standard_normal((M, K))), f32(rng.standard_normal((N, K)) / np.sqrt(K))
    b = f32(rng.standard_normal(N))
    r = f32(rng.standard_normal((M, N))) if res else None
    out = np.full((M, N), np.nan, np.float32)
    nb = lib.mdm_linear_x3_scratch_bytes(M, N, K)
    scratch = np.zeros(nb, np.uint8)
    lib.check(lib.mdm_linear_x3(ptr(a), ptr(w), ptr(b), ptr(r) if res else None, ptr(out), M, N, K, act,
                                    ptr(scratch), nb, None), "linear_f16x3")
    ref = torch.from_numpy(a).double() @ torch.from_numpy(w).double().t() + torch.from_numpy(b).double()
    ref = torch.nn.functional.gelu(ref) if act == 1 else ref
    if res:
        ref = ref + torch.from_numpy(r).double()
    assert maxabs(out, ref.numpy()) < 6e-5


def _f16f6_reference(a, w):
    """numpy restatement of lab/csrc_probe/gemm_f16f6.h's operand decomposition: fp16 hi + MX-FP6 (E2M3, power-of-two scale per 32 k)
    of hi and lo; returns sum_k ah*wh + q6(ah)*q6(wl) + q6(al)*q6(wh) in float64."""
    def q6(x):
        R, K = x.shape
        b = x.reshape(R, K // 32, 32).astype(np.float64)
        amax = np.abs(b).max(-1, keepdims=True)
        e = np.where(amax > 0, np.floor(np.log2(np.where(amax > 0, amax, 1.0))) - 2, 0.0)
        v = np.clip(b / 2.0 ** e, -7.5, 7.5)
        av = np.abs(v)
        step = np.where(av >= 4, 0.5, np.where(av >= 2, 0.25, 0.125))
        q = np.clip(np.round(v / step) * step, -7.5, 7.5)         # np.round = round-half-even, as rintf
        return (q * 2.0 ** e).reshape(R, K)
    ah, wh = a.astype(np.float16).astype(np.float64), w.astype(np.float16).astype(np.float64)
    al, wl = a.astype(np.float64) - ah, w.astype(np.float64) - wh
    return ah @ wh.T + q6(ah) @ q6(wl).T + q6(al) @ q6(wh).T


@pytest.mark.parametrize("M,N,K,act,res,ref_kernel", [(70, 40, 96, 0, True, True), (33, 96, 128, 1, False, True),
                                                      (230, 264, 96, 0, True, False), (70, 40, 64, 1, False, False)])
def test_emulated_linear_f16f6(lib, M, N, K, act, res, ref_kernel):
    """Seed of the next GEMM (fp16 pass + two MX-FP6 cross terms): device quantiser, plane records, fragment mapping of both
    instructions (emulated) against the numpy restatement of the decomposition and against the exact product."""
    rng = np.random.default_rng(M)
    a, w = f32(rng.standard_normal((M, K))), f32(rng.standard_normal((N, K)) / np.sqrt(K))
    a[::7, ::5] *= 9.0                                            # outliers inside MX blocks
    a[3, 32:64] = 0.0                                             # an all-zero block
    b = f32(rng.standard_normal(N))
    r = f32(rng.standard_normal((M, N))) if res else None
    out = np.full((M, N), np.nan, np.float32)
    nb = lib.mdm_linear_f16f6_scratch_bytes(M, N, K)
    scratch = np.zeros(nb, np.uint8)
    lib.mdm_debug_set(4, 1 if ref_kernel else 0)       # reference kernel / the production skeleton with the f16f6 k-loop
    try:
        lib.check(lib.mdm_linear_f16f6(ptr(a), ptr(w), ptr(b), ptr(r) if res else None, ptr(out), M, N, K, act,
                                       ptr(scratch), nb, None), "linear_f16f6")
    finally:
        lib.mdm_debug_set(4, 0)

    def finish(c):
        c = torch.from_numpy(c) + torch.from_numpy(b).double()
        c = torch.nn.functional.gelu(c) if act == 1 else c
        return (c + torch.from_numpy(r).double() if res else c).numpy()
    assert maxabs(out, finish(_f16f6_reference(a, w))) < 2e-5      # the scheme itself: fp32 accumulation order only
    assert maxabs(out, finish(a.astype(np.float64) @ w.astype(np.float64).T)) < 6e-4   # vs the exact product (rows with 9x outliers: ~3e-5 of the output range)


def test_emulated_attention_mask(lib):
    nseq, B, S, D, H, hd = 2, 2, 37, 256, 2, 128
    rng = np.random.default_rng(0)
    qkv = f32(rng.standard_normal((nseq * S, 3 * D)))
    qkv[:, :D] /= np.sqrt(hd)
    lengths = np.array([36, 3], np.int32)
    out = np.full((nseq * S, D), np.nan, np.float32)
    lib.check(lib.mdm_attention(ptr(qkv), ptr(out), ptr(lengths), nseq, B, S, D, H, None), "attention")
    t = torch.from_numpy(qkv).double()
    q, k, v = (u.view(nseq, S, H, hd).transpose(1, 2) for u in t.split(D, -1))
    sc = q @ k.transpose(-1, -2)
    for s in range(nseq):
        sc[s, :, :, 1 + int(lengths[s]):] = float("-inf")
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(nseq * S, D)
    assert maxabs(out, ref.numpy()) < 1e-5


@pytest.mark.parametrize("S,lengths", [(37, [36, 3]), (70, [69, 40]), (197, [196, 100])])
def test_emulated_attention_f16x3(lib, S, lengths):
    """Split-precision attention: plane layouts (swizzled K rows, MFMA-ordered V^T), masking, deferred normalisation."""
    nseq, B, D, H, hd = 2, 2, 256, 2, 128
    rng = np.random.default_rng(S)
    qkv = f32(rng.standard_normal((nseq * S, 3 * D)))
    qkv[:, :D] /= np.sqrt(hd)
    lengths = np.array(lengths, np.int32)
    out = np.full((nseq * S, D), np.nan, np.float32)
    nb = lib.mdm_attention_x3_scratch_bytes(nseq, S, D)
    scratch = np.zeros(nb, np.uint8)
    lib.check(lib.mdm_attention_x3(ptr(qkv), ptr(out), ptr(lengths), nseq, B, S, D, H, ptr(scratch), nb, None),
              "attention_f16x3")
    t = torch.from_numpy(qkv).double()
    q, k, v = (u.view(nseq, S, H, hd).transpose(1, 2) for u in t.split(D, -1))
    sc = q @ k.transpose(-1, -2)
    for s in range(nseq):
        sc[s, :, :, 1 + int(lengths[s]):] = float("-inf")
    ref = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(nseq * S, D)
    assert maxabs(out, ref.numpy()) < 5e-5


def test_emulated_recover_from_ric(lib):
    """csrc/motion_recover.h (inv_transform + recover_from_ric + permute) against the oracle, ragged T and KIT width."""
    from mdm_amd.motion_process import recover_from_ric
    from oracle import motion_oracle as mo
    from oracle.make_golden_motion import motion_inputs
    for B, T, JF, J in [(2, 37, 263, 22), (1, 300, 251, 21)]:
        sample, mean, std = motion_inputs(B, T, seed=B * 100 + T, JF=JF)
        got = recover_from_ric(sample, mean, std, _native_lib=lib)
        want = mo.recover_from_ric(sample.numpy(), mean.numpy(), std.numpy(), J)
        assert got.shape == (B, J, 3, T)
        assert maxabs(got, want) < 2e-6 * float(np.abs(want).max())


@pytest.mark.parametrize("masked,prec", [(False, "f16x3"), (True, "f32")])
def test_emulated_dip_decoder_forward(lib, masked, prec):
    """trans_dec denoiser: prefix completion, token-level text memory with ragged lengths, cross-attention
    with a different key count than queries, both CFG branches, through MDM.forward / ClassifierFreeSampleModel."""
    B, C, P = 2, 5, 12
    sd = dip_small_state_dict(num_layers=1 if masked else 2)
    model, _ = make_pair(sd, 10, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, mask_frames=masked,
                         precision=prec)
    y = synth_dip_y(B, P, C, seed=3, text_lengths=[6, 3], lengths=[12, 7] if masked else None, scale=2.5)
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([9, 0])
    kw = dict(context_len=C, num_heads=2, mask_frames=masked)
    assert maxabs(model.model(x, t, y=dict(y)), dip.dip_forward(sd, x, t, y, **kw)) < 2e-5
    assert maxabs(model.model(x, t, y={**y, "uncond": True}), dip.dip_forward(sd, x, t, {**y, "uncond": True}, **kw)) < 2e-5
    assert maxabs(model(x, t, y=dict(y)), dip.dip_cfg_forward(sd, x, t, y, **kw)) < 5e-5


def test_emulated_dip_decoder_planes_and_fp32_skeleton(lib, engine_options):
    """The f16x3 trans_dec stack has two routes (csrc/decoder.h dec_on_planes): operand planes through gemm_x3s.h /
    attention_x3.h (what DiP's callers run) and the fp32 skeleton of gemm_f32.h (small_gemm_max_seqs = 0 forces it).  Both
    against the oracle, on 32- and 64-row tiles; the two are different arithmetic, so agreeing bit for bit would mean the switch
    did nothing."""
    B, C, P = 3, 5, 12
    sd = dip_small_state_dict(num_layers=2)
    y = synth_dip_y(B, P, C, seed=3, text_lengths=[6, 3, 2], lengths=None, scale=2.5)
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([9, 0, 4])
    want = dip.dip_cfg_forward(sd, x, t, y, context_len=C, num_heads=2, mask_frames=False)
    outs = {}
    for tag, opts in (("planes32", {"small_gemm_row_tiles": 1}), ("planes64", {"small_gemm_row_tiles": 2}),
                      ("skeleton", {"small_gemm_max_seqs": 0})):
        engine_options(**opts)
        model, _ = make_pair(sd, 10, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, precision="f16x3")
        outs[tag] = model(x, t, y=dict(y))
        assert maxabs(outs[tag], want) < 5e-5, tag
    assert not torch.equal(outs["planes32"], outs["skeleton"])
    assert maxabs(outs["planes32"], outs["planes64"]) < 1e-5


@pytest.mark.parametrize("B,C,P,text_lengths,rt", [(2, 3, 70, [40, 3], 2), (3, 0, 33, [5, 9, 1], 1)])
def test_emulated_dip_decoder_planes_other_windows(lib, engine_options, B, C, P, text_lengths, rt):
    """The plane route of the trans_dec stack at other window shapes than 20 + 40: S = 73 tokens (a 64-row tile + 9 rows per sequence
    for in_proj, contiguous tiles elsewhere), a 40-token memory (two key tiles in the cross-attention), no prefix at all."""
    engine_options(small_gemm_row_tiles=rt)
    sd = dip_small_state_dict(num_layers=2)
    y = synth_dip_y(B, P, max(C, 1), seed=3, text_lengths=text_lengths, lengths=None, scale=2.5)
    if C == 0:
        y.pop("prefix")
    else:
        y["prefix"] = y["prefix"][..., :C].contiguous()
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(1))
    t = torch.arange(B) % 10
    model, _ = make_pair(sd, 10, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, precision="f16x3")
    want = dip.dip_cfg_forward(sd, x, t, y, context_len=C, num_heads=2, mask_frames=False)
    assert maxabs(model(x, t, y=dict(y)), want) < 5e-5


@pytest.mark.parametrize("guided,prec", [(True, "f16x3"), (False, "f32")])
def test_emulated_dip_window_loop(lib, guided, prec):
    """mdm_sample_loop_dec (one p_sample_loop over a DiP prediction window: text projections hoisted out of the steps, the
    step's projected time row added while the attention kernel stages K / V) against the oracle's loop, and against the
    same loop composed step by step from mdm_forward_dec + mdm_sampler_step (diffusion.dip_stepwise); dump_steps included."""
    B, C, P, steps = 2, 5, 12, 2
    sd = dip_small_state_dict(num_layers=1)
    model, diffusion = make_pair(sd, steps, "cpu", guided=guided, native_lib=lib, context_len=C, pred_len=P, precision=prec)
    y = synth_dip_y(B, P, C, seed=4, text_lengths=[6, 3], scale=2.5)
    g = torch.Generator().manual_seed(8)
    seq = [torch.randn(B, 263, 1, P, generator=g) for _ in range(1 + steps)]
    tab = orc.Tables(orc.named_betas("cosine", steps))
    want = dip.dip_sample_loop(sd, tab, (B, 263, 1, P), y, seq[0], seq[1:], context_len=C, cfg=guided, num_heads=2)
    run = lambda **kw: diffusion.p_sample_loop(model, (B, 263, 1, P), clip_denoised=False,   # noqa: E731
                                               model_kwargs={"y": dict(y)}, noise_sequence=seq, **kw)
    got = run()
    assert maxabs(got, want) < 5e-5
    dumps = run(dump_steps=[0, 1])
    assert len(dumps) == 2 and torch.equal(dumps[1], got)
    diffusion.dip_stepwise = True
    step = run(dump_steps=[0, 1])
    assert maxabs(got, step[1]) < 2e-5 and maxabs(step[0], dumps[0]) < 2e-5


@pytest.mark.parametrize("prec", ["f32"])
def test_emulated_dip_window_loop_sample_groups(lib, monkeypatch, prec):
    """mdm_sample_loop_dec cuts the batch into sample groups that run on concurrent streams (sequentially on the emulator):
    per-group slices of every buffer, the hoisted text K / V of the WHOLE batch read through (branch, sample) remapping, per-group
    Philox bases.  The result must not depend on the number of groups, with injected and with Philox noise: bit for bit in the
    exact-fp32 mode; in f16x3 up to the re-association of a GEMM whose tile shape follows the row count (gemm_f32.h)."""
    B, C, P, steps = 4, 5, 12, 2
    sd = dip_small_state_dict(num_layers=1)
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, precision=prec)
    y = synth_dip_y(B, P, C, seed=6, text_lengths=[6, 3, 1, 5], lengths=None, scale=2.5)
    g = torch.Generator().manual_seed(9)
    seq = [torch.randn(B, 263, 1, P, generator=g) for _ in range(1 + steps)]
    other = "2" if prec == "f32" else "4"
    run = lambda **kw: diffusion.p_sample_loop(model, (B, 263, 1, P), clip_denoised=False,   # noqa: E731
                                               model_kwargs={"y": dict(y)}, **kw)
    kw = dict(noise_sequence=seq) if prec == "f32" else dict(seed=3)
    monkeypatch.setenv("MDM_DIP_GROUPS", "1")
    one = run(**kw)
    monkeypatch.setenv("MDM_DIP_GROUPS", other)
    many = run(**kw)
    assert torch.equal(many, one) if prec == "f32" else maxabs(many, one) < 2e-5
    if prec == "f32":
        tab = orc.Tables(orc.named_betas("cosine", steps))
        want = dip.dip_sample_loop(sd, tab, (B, 263, 1, P), y, seq[0], seq[1:], context_len=C, cfg=True, num_heads=2)
        assert maxabs(many, want) < 5e-5


@pytest.mark.parametrize("ddim,clip,fixed_large", [(False, True, False), (True, True, False), (False, False, True)])
def test_emulated_loop_clip_denoised_and_fixed_large(lib, ddim, clip, fixed_large):
    """`clip_denoised=True` -- the reference's signature default, gaussian_diffusion.py:591-608, :347-353 -- and FIXED_LARGE
    variances (:325-333) through the fused loop, on the emulator (the GPU suite holds the same branches against fixtures of the
    upstream reference: tests/test_gpu_round4.py); the clamp must be live on the case."""
    steps, B, T = 3, 2, 9
    sd = small_state_dict(num_layers=1)
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, sigma_small=not fixed_large)
    y = synth_y(B, T, seed=5, lengths=[T, 4])
    shape = (B, 263, 1, T)
    x_T, noises = orc.make_noise(shape, steps, 11)
    fn = diffusion.ddim_sample_loop if ddim else diffusion.p_sample_loop
    seq = [x_T] + [n.contiguous() for n in noises]
    got = fn(model, shape, clip_denoised=clip, model_kwargs={"y": dict(y)}, noise_sequence=seq)
    tab = orc.Tables(orc.named_betas("cosine", steps))
    kw = dict(cfg=True, num_heads=2, ddim=ddim, fixed_large=fixed_large)
    want = orc.sample_loop(sd, tab, shape, y, x_T, noises, clip_denoised=clip, **kw)
    assert maxabs(got, want) < 2e-5
    if clip:
        assert maxabs(want, orc.sample_loop(sd, tab, shape, y, x_T, noises, clip_denoised=False, **kw)) > 1e-2
        default = fn(model, shape, model_kwargs={"y": dict(y)}, noise_sequence=seq)       # the signature default IS the clamp
        assert torch.equal(default, got)


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_emulated_trans_dec_with_clip_memory(lib, prec):
    """model/mdm.py:85-93, :261-262: the decoder with `text_encoder_type='clip'` -- ONE memory token per sample, no memory pad
    mask -- forwards (cond / uncond / guided) and a window loop against the oracle."""
    B, C, P, steps = 2, 5, 12, 2
    from oracle.synth import synth_dip_state_dict
    sd = synth_dip_state_dict(seed=1, latent_dim=256, num_layers=1, bert_dim=512)
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, precision=prec,
                                 text_encoder_type="clip")
    assert model.model.clip_dim == 512
    g = torch.Generator().manual_seed(3)
    y = {"mask": torch.ones(B, 1, 1, P, dtype=torch.bool), "lengths": torch.full((B,), P),
         "text_embed": torch.randn(1, B, 512, generator=g), "prefix": torch.randn(B, 263, 1, C, generator=g),
         "scale": torch.ones(B) * 2.5}
    x = torch.randn(B, 263, 1, P, generator=g)
    t = torch.tensor([1, 0])
    kw = dict(context_len=C, num_heads=2)
    assert maxabs(model.model(x, t, y=dict(y)), dip.dip_forward(sd, x, t, y, **kw)) < 2e-5
    assert maxabs(model.model(x, t, y={**y, "uncond": True}), dip.dip_forward(sd, x, t, {**y, "uncond": True}, **kw)) < 2e-5
    assert maxabs(model(x, t, y=dict(y)), dip.dip_cfg_forward(sd, x, t, y, **kw)) < 5e-5
    seq = [torch.randn(B, 263, 1, P, generator=g) for _ in range(1 + steps)]
    got = diffusion.p_sample_loop(model, (B, 263, 1, P), clip_denoised=False, model_kwargs={"y": dict(y)}, noise_sequence=seq)
    want = dip.dip_sample_loop(sd, orc.Tables(orc.named_betas("cosine", steps)), (B, 263, 1, P), y, seq[0], seq[1:],
                               context_len=C, cfg=True, num_heads=2)
    assert maxabs(got, want) < 5e-5


def test_emulated_dip_sample_groups_with_hole_masks(lib, monkeypatch):
    """with sample groups (MDM_DIP_GROUPS > 1, probe / emulator builds) the ABI-7 bitmap form of `lengths`
    ([9B]: counts, then eight words per sample) was sliced as `lengths + b0` with B = the group's size, i.e. a group read the
    WRONG samples' bitmaps.  The whole-batch array + (len_B, len_b0) now travel to the attention kernel: groups must reproduce
    the ungrouped loop bit for bit in the exact-fp32 mode on masks with holes."""
    B, C, P, steps = 4, 5, 12, 2
    sd = dip_small_state_dict(num_layers=1)
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, precision="f32",
                                 mask_frames=True)
    y = synth_dip_y(B, P, C, seed=6, text_lengths=[6, 3, 1, 5], lengths=[12, 9, 12, 7], scale=2.5)
    y["mask"] = y["mask"].clone()
    y["mask"][1, 0, 0, [0, 4]] = False          # holes: samples 1 and 3 travel as bitmaps, 0 and 2 as counts
    y["mask"][3, 0, 0, [2, 3, 5]] = False
    g = torch.Generator().manual_seed(9)
    seq = [torch.randn(B, 263, 1, P, generator=g) for _ in range(1 + steps)]
    run = lambda: diffusion.p_sample_loop(model, (B, 263, 1, P), clip_denoised=False, model_kwargs={"y": dict(y)},   # noqa: E731
                                          noise_sequence=seq)
    monkeypatch.setenv("MDM_DIP_GROUPS", "1")
    one = run()
    monkeypatch.setenv("MDM_DIP_GROUPS", "2")
    two = run()
    assert torch.equal(one, two)
    want = dip.dip_sample_loop(sd, orc.Tables(orc.named_betas("cosine", steps)), (B, 263, 1, P), y, seq[0], seq[1:],
                               context_len=C, cfg=True, num_heads=2, mask_frames=True)
    assert maxabs(one, want) < 5e-5


def test_integration_md_snippet_is_a_program_and_its_structs_match_the_binding():
    """INTEGRATION.md section 2 is executed verbatim on the GPU (tests/test_gpu_round4.py); here: it parses, and the structs it
    declares by hand have exactly the fields, order and ctypes of the binding the seams use (which tests/test_abi.py ties to
    include/mdm_hip.h)."""
    import ast
    import re
    from helpers import ROOT
    from mdm_amd import _native as nat
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- ctypes-snippet-begin -->\s*```python\n(.*?)```\s*<!-- ctypes-snippet-end -->", text, flags=re.S)
    assert m
    tree = ast.parse(m.group(1))
    ns = {}
    head = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom, ast.Assign, ast.ClassDef))][:4]
    # import ctypes ...; the i32/u32/... aliases; the two struct classes
    exec(compile(ast.Module(body=[n for n in tree.body[:1]] + [n for n in tree.body if isinstance(n, ast.Assign)][:1] +
                            [n for n in tree.body if isinstance(n, ast.ClassDef)], type_ignores=[]), "snippet", "exec"), ns)
    for mine, theirs in ((ns["MdmConfig"], nat.MdmConfig), (ns["MdmSampleParams"], nat.MdmSampleParams)):
        assert [(n, t) for n, t in mine._fields_] == [(n, t) for n, t in theirs._fields_]
    assert "mdm_abi_version() == %d" % nat.ABI_VERSION in m.group(1)


@pytest.mark.parametrize("rt", [1, 2])
def test_emulated_small_gemm_at_the_headline_width(lib, engine_options, rt):
    """csrc/gemm_x3s.h at latent_dim = 512 (the emulator cases above run 256: one K-chunk): two chunks for in_proj / out_proj /
    linear1 (four 128-k chunks), eight for linear2, the 18-sub-step single chunk of InputProcess (K = 288), N = 264 of OutputProcess
    (a column tile whose last three waves lie past the packed weight rows), 32- and 64-row tiles, a 5-row last tile (S = 37)."""
    engine_options(small_gemm_row_tiles=rt)
    B, T = 1, 36
    sd = small_state_dict(latent_dim=512, num_layers=2)
    model, _ = make_pair(sd, 50, "cpu", guided=True, native_lib=lib, precision="f16x3")
    y = synth_y(B, T, seed=2, lengths=[23])
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(B, 263, 1, T, generator=g), torch.tensor([49])
    assert maxabs(model(x, t, y=dict(y)), orc.cfg_forward(sd, x, t, y, num_heads=4)) < 5e-5


def test_emulated_wide_form_of_the_pipelined_gemm(lib, monkeypatch, engine_options):
    """gemm_x3.h NCB = 2 (round 4, probe / emulator builds: MDM_X3_WIDE=1): the pipelined k-loop as four waves x 64 columns -- W slots,
    counted waits, 16-row sub-tile and every epilogue over two column blocks per wave.  Measured slower on the MI355X and not a
    product path (profiles/r04d_wide.md); kept correct."""
    monkeypatch.setenv("MDM_X3_WIDE", "1")       # (an experiment switch of the probe / emulator builds; not in the product library)
    engine_options(small_gemm_max_seqs=0)
    B, T = 2, 33
    sd = small_state_dict(num_layers=2)
    model, _ = make_pair(sd, 50, "cpu", guided=True, native_lib=lib, precision="f16x3")
    y = synth_y(B, T, seed=2, lengths=[33, 5])
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(B, 263, 1, T, generator=g), torch.tensor([49, 0])
    assert maxabs(model(x, t, y=dict(y)), orc.cfg_forward(sd, x, t, y, num_heads=2)) < 5e-5


# ---- round 5 ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("route", ["planes32"])
def test_emulated_dip_plane_route_takes_frame_masks(lib, engine_options, route):
    """VERDICT r04, What's missing 1: DiP's recipe trains with --mask_frames (DiP.md:181), so every forward of a real checkpoint
    carries a tgt_key_padding_mask (model/mdm.py:241-247, :263-265) -- and round 4's operand-plane route refused any.  Now the
    counts / bitmaps reach attention_x3.h with lead = 0 (no condition token; the context_len prefix frames are counted as frames):
    prefix masks (counts), masks with holes (bitmaps) and the all-valid mask of sample/generate.py:107, forward and window loop,
    against the oracle -- and the route really is the plane route (differs from the fp32 skeleton in the last bits)."""
    B, C, P, steps = 4, 5, 12, 2
    engine_options(small_gemm_row_tiles=1 if route == "planes32" else 2)
    sd = dip_small_state_dict(num_layers=2)
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, mask_frames=True)
    y = synth_dip_y(B, P, C, seed=6, text_lengths=[6, 3, 1, 5], lengths=[12, 9, 12, 7], scale=2.5)
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([1, 0, 1, 0])
    kw = dict(context_len=C, num_heads=2, mask_frames=True)
    got = model(x, t, y=dict(y))
    assert maxabs(got, dip.dip_cfg_forward(sd, x, t, y, **kw)) < 5e-5
    assert maxabs(got, dip.dip_cfg_forward(sd, x, t, y, context_len=C, num_heads=2, mask_frames=False)) > 1e-3   # the mask matters
    yh = dict(y)
    yh["mask"] = y["mask"].clone()
    yh["mask"][1, 0, 0, [0, 4]] = False          # holes: samples 1 and 3 travel as bitmaps, 0 and 2 as counts
    yh["mask"][3, 0, 0, [2, 3, 5]] = False
    got_h = model(x, t, y=dict(yh))
    assert maxabs(got_h, dip.dip_cfg_forward(sd, x, t, yh, **kw)) < 5e-5
    assert maxabs(got_h, got) > 1e-4
    g = torch.Generator().manual_seed(9)
    seq = [torch.randn(B, 263, 1, P, generator=g) for _ in range(1 + steps)]
    loop = diffusion.p_sample_loop(model, (B, 263, 1, P), clip_denoised=False, model_kwargs={"y": dict(yh)}, noise_sequence=seq)
    want = dip.dip_sample_loop(sd, orc.Tables(orc.named_betas("cosine", steps)), (B, 263, 1, P), yh, seq[0], seq[1:],
                               context_len=C, cfg=True, num_heads=2, mask_frames=True)
    assert maxabs(loop, want) < 5e-5
    # generate.py's mask: ones, wider than the window -> all valid, equal to the unmasked model's output on this route bit for bit
    ya = dict(y)
    ya["mask"] = torch.ones(B, 1, 1, 196, dtype=torch.bool)
    plain, _ = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, mask_frames=False)
    assert torch.equal(model(x, t, y=dict(ya)), plain(x, t, y=dict(ya)))
    engine_options(small_gemm_max_seqs=0)
    skel, _ = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, mask_frames=True)
    got_s = skel(x, t, y=dict(yh))
    assert maxabs(got_s, got_h) < 5e-5 and not torch.equal(got_s, got_h)
    engine_options(small_gemm_row_tiles=2)          # the 64-row tiles of the same route: one holed forward
    m64, _ = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, mask_frames=True)
    assert maxabs(m64(x, t, y=dict(yh)), dip.dip_cfg_forward(sd, x, t, yh, **kw)) < 5e-5


@pytest.mark.parametrize("latent_dim", [768, 1024])
def test_emulated_wide_latent_dims_on_both_gemm_kernels(lib, gemm_path, latent_dim):
    if gemm_path == "big" and latent_dim == 1024:
        pytest.skip("four 256-column partials on the sequence tiles: round 4's own path (the emulator's time goes to the 208-row tiles)")
    """ADVICE r04 (high): mdm_create accepts latent_dim 768 and 1024, the small-tile kernel leaves D / 128 = 6 / 8 partial
    statistics per row, and its consumer only knew 1, 2, 4 and "else = 3" (round 4: launch refused with parts > 4 -- no small batch
    ran at those widths at all).  Both GEMM kernels, two layers (every folded-LayerNorm kind), against the oracle."""
    B, T = 1, 13
    sd = small_state_dict(latent_dim=latent_dim, num_layers=2)
    model, _ = make_pair(sd, 50, "cpu", guided=True, native_lib=lib, precision="f16x3")
    y = synth_y(B, T, seed=2, lengths=[11])
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(B, 263, 1, T, generator=g), torch.tensor([31])
    assert maxabs(model(x, t, y=dict(y)), orc.cfg_forward(sd, x, t, y, num_heads=latent_dim // 128)) < 5e-5


def test_emulated_dip_plane_route_at_latent_dim_768(lib, engine_options):
    """The same six-partial statistics on the trans_dec stack (three folded LayerNorms per layer), with a frame mask."""
    B, C, P = 1, 5, 12
    engine_options(small_gemm_row_tiles=1)
    sd = dip_small_state_dict(latent_dim=768, num_layers=2)
    model, _ = make_pair(sd, 10, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, mask_frames=True)
    y = synth_dip_y(B, P, C, seed=3, text_lengths=[6], lengths=[7], scale=2.5)
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([9])
    assert maxabs(model(x, t, y=dict(y)), dip.dip_cfg_forward(sd, x, t, y, context_len=C, num_heads=6, mask_frames=True)) < 5e-5


def test_engine_options_are_set_through_the_abi_not_the_environment(lib, monkeypatch):
    """the kernel route of a handle is an option of the handle (include/mdm_hip.h mdm_set_option, ABI 9); the
    environment variables rounds 3-4 read on the launch path do nothing any more."""
    from mdm_amd._engine import Engine
    cfg = dict(njoints=263, nfeats=1, latent_dim=256, ff_size=1024, num_layers=1, num_heads=2, clip_dim=512, max_len=64,
               mask_frames=1, arch=0, context_len=0)
    monkeypatch.setenv("MDM_X3S_MAX_SEQS", "0")
    monkeypatch.setenv("MDM_X3S_RT", "2")
    if not lib.has_probes:      # (the probe / emulator builds preset new handles from these two variables for tools/' A/B scripts)
        e = Engine(cfg, lib=lib)
        assert e.get_option("small_gemm_max_seqs") == 80 and e.get_option("small_gemm_row_tiles") == 0
    monkeypatch.delenv("MDM_X3S_MAX_SEQS")
    monkeypatch.delenv("MDM_X3S_RT")
    e = Engine(cfg, lib=lib, options={"small_gemm_row_tiles": 2})
    assert e.get_option("small_gemm_max_seqs") == 80 and e.get_option("small_gemm_row_tiles") == 2
    e.set_option("small_gemm_max_seqs", 0)
    assert e.get_option("small_gemm_max_seqs") == 0
    with pytest.raises(Exception, match="ROW_TILES"):
        e.set_option("small_gemm_row_tiles", 3)
    with pytest.raises(ValueError):
        e.set_option("no_such_option", 1)


@pytest.mark.parametrize("B,C,P,text_lengths,lengths", [(3, 5, 12, [6, 3, 2], [12, 7, 12]),      # S = 17: one 17-row tile per sequence
                                                         (2, 20, 40, [24, 64], None),            # DiP's window; a 64-token memory (2 key tiles)
                                                         (2, 3, 70, [70, 3], None),              # S = 73: 32 + 32 + 9 rows; 3 key tiles
                                                         (2, 0, 33, [40, 9], [33, 20])])         # no prefix; S = 33: 32 + 1; 2 key tiles
def test_emulated_dip_fused_cross_attention_block(lib, engine_options, B, C, P, text_lengths, lengths):
    """csrc/xattn_block.h (round 5): query projection with norm1 folded -> attention over the text memory -> out_proj + norm1
    residual + row statistics as ONE kernel per decoder layer, against the oracle and against the three-launch form it replaces
    (dec_fused_xattn = 0: the small GEMM twice around the exact-fp32 attention kernel): same values to the last few bits, not the
    same bits (the attention contractions are split-precision here, exact fp32 there).  Forward (memory projected per call) and
    window loop (text K / V hoisted, the step's time row added while the fragments are built)."""
    steps = 2
    sd = dip_small_state_dict(num_layers=2)
    masked = lengths is not None
    y = synth_dip_y(B, P, max(C, 1), seed=3, text_lengths=text_lengths, lengths=lengths, scale=2.5)
    if C == 0:
        y.pop("prefix")
    else:
        y["prefix"] = y["prefix"][..., :C].contiguous()
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(1))
    t = torch.arange(B) % steps
    g = torch.Generator().manual_seed(9)
    seq = [torch.randn(B, 263, 1, P, generator=g) for _ in range(1 + steps)]
    kw = dict(context_len=C, num_heads=2, mask_frames=masked)
    want_f = dip.dip_cfg_forward(sd, x, t, y, **kw)
    want_l = dip.dip_sample_loop(sd, orc.Tables(orc.named_betas("cosine", steps)), (B, 263, 1, P), y, seq[0], seq[1:],
                                 context_len=C, cfg=True, num_heads=2, mask_frames=masked) if (B, C, P) == (3, 5, 12) else None
    outs = {}
    full = (B, C, P) == (3, 5, 12)      # the window loop (hoisted memory, time row per step) and the three-launch form: first case only
    # 2: projection + attention per (sequence, head) (selfattn_block.h CROSS; S = 73 / 70 memory tokens: falls to 1); 1: xattn_block.h
    for fused in ((2, 1, 0) if full else (2, 1)):
        engine_options(dec_fused_xattn=fused, small_gemm_row_tiles=1)
        model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, mask_frames=masked)
        assert model.model.engine().get_option("dec_fused_xattn") == fused
        f = model(x, t, y=dict(y))
        assert maxabs(f, want_f) < 5e-5, fused
        lo = None
        if full:
            lo = diffusion.p_sample_loop(model, (B, 263, 1, P), clip_denoised=False, model_kwargs={"y": dict(y)}, noise_sequence=seq)
            assert maxabs(lo, want_l) < 5e-5, fused
        outs[fused] = (f, lo)
    assert maxabs(outs[1][0], outs[2][0]) < 2e-5
    if full:
        assert not torch.equal(outs[0][0], outs[1][0]) and maxabs(outs[0][0], outs[1][0]) < 2e-5
        assert maxabs(outs[0][1], outs[2][1]) < 5e-5 and maxabs(outs[0][1], outs[1][1]) < 5e-5


@pytest.mark.parametrize("B,C,P,text_lengths,holes", [(3, 5, 12, [6, 3, 2], True),       # S = 17: one sub-tile, 47 pad rows
                                                       (2, 20, 40, [9, 24], False),      # S = 60: DiP's window
                                                       (2, 0, 64, [5, 9], True)])        # S = 64: the full tile, no prefix
def test_emulated_dip_fused_self_attention_block(lib, engine_options, B, C, P, text_lengths, holes):
    """csrc/selfattn_block.h (round 5): in_proj (norm3 of the previous layer folded for l >= 1) + self-attention of a (sequence, head)
    as ONE kernel for sequences of at most 64 tokens -- Q, K, V^T live in LDS fragment images instead of 35 MB of planes per layer.
    Against the oracle and against the two-launch form (dec_fused_selfattn = 0: gemm_x3s kind 0 / 6 + attention_x3_kernel), with
    frame masks as counts and as bitmaps (lead = 0), two layers (plain and folded in_proj)."""
    sd = dip_small_state_dict(num_layers=2)
    lengths = [P - 3 * i for i in range(B)]
    y = synth_dip_y(B, P, max(C, 1), seed=3, text_lengths=text_lengths, lengths=lengths, scale=2.5)
    if C == 0:
        y.pop("prefix")
    else:
        y["prefix"] = y["prefix"][..., :C].contiguous()
    if holes:
        y["mask"] = y["mask"].clone()
        y["mask"][0, 0, 0, [1, 4]] = False
        y["mask"][B - 1, 0, 0, [0, 2, 3]] = False
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(1))
    t = torch.arange(B) % 10
    want = dip.dip_cfg_forward(sd, x, t, y, context_len=C, num_heads=2, mask_frames=True)
    outs = {}
    for fused in (1, 0):
        engine_options(dec_fused_selfattn=fused, small_gemm_row_tiles=1)
        model, _ = make_pair(sd, 10, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, mask_frames=True)
        assert model.model.engine().get_option("dec_fused_selfattn") == fused
        outs[fused] = model(x, t, y=dict(y))
        assert maxabs(outs[fused], want) < 5e-5, fused
    assert maxabs(outs[0], outs[1]) < 2e-5


@pytest.mark.parametrize("T,lengths", [(40, [40, 13, 33]), (9, [9, 4, 9])])
def test_emulated_attention_direct_output_is_bit_identical(lib, engine_options, T, lengths):
    """csrc/attention_x3.h DIRECT (round 5, the item-boundary drain of VERDICT r04 item 4): planes straight from the accumulators, the
    next item's key tiles 1 and 2 requested in front of those stores, no queue drain at the item start (counted waits that include the
    stores in flight).  Same arithmetic: the encoder forward must be bit-identical to the staged form; 6 sequences x 2 heads = 12
    items on the emulator's 16 persistent workgroups, i.e. every workgroup of the first group of 8 carries a second item."""
    B = 3
    sd = small_state_dict(num_layers=2)
    y = synth_y(B, T, seed=2, lengths=lengths)
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(B, 263, 1, T, generator=g), torch.tensor([49, 0, 7])
    outs = []
    for direct in (1, 0):
        engine_options(attn_direct_out=direct, small_gemm_max_seqs=0)       # (the sequence-tile route: what the headline batch runs)
        model, _ = make_pair(sd, 50, "cpu", guided=True, native_lib=lib, precision="f16x3")
        assert model.model.engine().get_option("attn_direct_out") == direct
        outs.append(model(x, t, y=dict(y)))
    assert torch.equal(outs[0], outs[1])
    assert maxabs(outs[0], orc.cfg_forward(sd, x, t, y, num_heads=2)) < 5e-5


def test_default_route_runs_row_tiles_up_to_80_sequences(lib, engine_options):
    """MDM_OPT_SMALL_GEMM_MAX_SEQS defaults to 80 since the cross-over was re-measured (profiles/r05k_crossovers.md): a guided forward of
    B = 24 (48 sequences -- the sequence-tile kernel under round 4's threshold of 40) takes csrc/gemm_x3s.h's row tiles by default.  The
    two kernels round differently (row statistics per 128 / per 256 columns; the small-batch tests hold them against each other), so
    the route shows in the bits: default == pinned row tiles.  (The other side -- 82 sequences on the sequence tiles -- costs the
    emulator minutes and is what every large-batch GPU test runs.)"""
    sd = small_state_dict(num_layers=1)
    B, T = 24, 3
    y = synth_y(B, T, seed=4, lengths=[3, 2] * 12)
    g = torch.Generator().manual_seed(1)
    x, t = torch.randn(B, 263, 1, T, generator=g), torch.arange(B) % 50
    outs = {}
    for name, opts in (("default", {}), ("rows", {"small_gemm_max_seqs": 128})):
        engine_options(**opts)
        model, _ = make_pair(sd, 50, "cpu", guided=True, native_lib=lib, precision="f16x3")
        outs[name] = model(x, t, y=dict(y))
    assert maxabs(outs["rows"], orc.cfg_forward(sd, x, t, y, num_heads=2)) < 5e-5
    assert torch.equal(outs["default"], outs["rows"])


# ---- round 6 ------------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,prompts", [(2, ["walks forward slowly", "turns", "sits down on a chair now"]),   # Ntok 5 / 3 / 8, B = 2
                                       (4, ["jumps high", "runs"])])                                            # B == Ntok == 4
def test_emulated_dip_dynamic_text_is_a_prompt_per_window(lib, B, prompts):
    """`--dynamic_text_path` (sample/generate.py:63-65, :134-142; utils/sampler_util.py:52, :66-71).  Upstream
    computes "window i <- prompt i of every sample" because p_sample_loop re-encodes y['text'] (gaussian_diffusion.py:633-635);
    round 5 handed the decoder upstream's SAMPLE-major slice of the cached embedding: an assert for B != Ntok and a silently
    different motion for B == Ntok.  Both shapes against the oracle's restatement (pinned to the reference run itself by
    tests/golden/dip_dynamic_text_*.npz), with y exactly as generate.py:130-142 leaves it; and the result is NOT what any fixed
    prompt gives."""
    from types import SimpleNamespace
    from mdm_amd.sampler_util import AutoRegressiveSampler
    from oracle.synth import synth_bert_encode_text, synth_dip_dynamic_y
    C, P, steps = 5, 12, 2
    frames = P * len(prompts)
    sd = dip_small_state_dict(num_layers=1)
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P)
    y = synth_dip_dynamic_y(B, P, C, seed=11, prompts=prompts, scale=2.5)
    chunks = dip.make_noise_chunks((B, 263, 1, P), steps, 5, len(prompts))
    it = iter(chunks)

    def sample_fn(mdl, shape, **kw):
        x_T, eps = next(it)
        return diffusion.p_sample_loop(mdl, shape, noise_sequence=[x_T] + [e.contiguous() for e in eps], **kw)

    args = SimpleNamespace(pred_len=P, context_len=C, autoregressive_include_prefix=False)
    got = AutoRegressiveSampler(args, sample_fn, frames).sample(model, (B, 263, 1, frames), clip_denoised=False,
                                                                model_kwargs={"y": y})
    tab = orc.Tables(orc.named_betas("cosine", steps))
    kw = dict(context_len=C, pred_len=P, required_frames=frames, cfg=True, num_heads=2)
    want = dip.autoregressive_sample(sd, tab, (B, 263, 1, frames), y, chunks, encode_text=synth_bert_encode_text, **kw)
    assert maxabs(got, want) < 5e-5
    y0 = {**y, "text": [prompts[0]] * B, "text_embed": synth_bert_encode_text([prompts[0]] * B)}
    assert maxabs(got, dip.autoregressive_sample(sd, tab, (B, 263, 1, frames), y0, chunks, **kw)) > 1e-2
    assert isinstance(y["text"][0], list) and y["text_embed"][0].dim() == 4         # the caller's dict is not rewritten


def test_dec_inputs_refuse_a_sample_major_text_embedding(lib):
    """model/mdm.py:185: bert_encode_text returns the embedding TOKEN-major [Ntok, B, 768].  A [B, Ntok, 768] block (upstream's
    dynamic-text slice) is a ValueError that names the layout, not a bare assert."""
    B, C, P = 2, 5, 12
    sd = dip_small_state_dict(num_layers=1)
    model, _ = make_pair(sd, 2, "cpu", guided=False, native_lib=lib, context_len=C, pred_len=P)
    y = synth_dip_y(B, P, C, seed=4, text_lengths=[6, 3])
    enc, pad = y["text_embed"]
    y["text_embed"] = (enc.permute(1, 0, 2).contiguous(), pad)
    x = torch.randn(B, 263, 1, P)
    with pytest.raises(ValueError, match="TOKEN-major"):
        model(x, torch.tensor([1, 0]), y=y)


def test_bert_encode_text_runs_an_attached_encoder(lib):
    """model/mdm.py:119, :180-187: `clip_model` is the DistilBERT wrapper; encode_text = permute + inverted mask.  None is attached
    offline (RuntimeError that says so); with one attached, a loop given only y['text'] encodes once and caches into the caller's
    dict (gaussian_diffusion.py:633-635) and equals the loop given the cached embedding."""
    from oracle.synth import synth_bert, synth_bert_encode_text
    B, C, P, steps = 2, 5, 12, 2
    sd = dip_small_state_dict(num_layers=1)
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P)
    texts = ["walks forward", "a person sits down"]
    with pytest.raises(RuntimeError, match="DistilBERT"):
        model.encode_text(texts)
    model.model.clip_model = synth_bert
    enc, pad = model.encode_text(texts)
    ref = synth_bert_encode_text(texts)
    assert torch.equal(enc, ref[0]) and torch.equal(pad, ref[1])
    y = synth_dip_y(B, P, C, seed=4, text_lengths=[6, 3], scale=2.5)
    seq = [torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(i)) for i in range(1 + steps)]
    y1 = {k: v for k, v in y.items() if k != "text_embed"}
    y1["text"] = texts
    a = diffusion.p_sample_loop(model, (B, 263, 1, P), clip_denoised=False, model_kwargs={"y": y1}, noise_sequence=seq)
    assert "text_embed" in y1
    y2 = {**y, "text": texts, "text_embed": ref}
    b = diffusion.p_sample_loop(model, (B, 263, 1, P), clip_denoised=False, model_kwargs={"y": y2}, noise_sequence=seq)
    assert torch.equal(a, b)


@pytest.mark.parametrize("prec,T,lengths,guided,loop", [("f16x3", 230, [230, 37], True, True), ("f32", 226, [100, 226], False, False),
                                                         ("f16x3", 259, [259, 224], False, False),
                                                         ("f16x3", 225, [200], False, False)])     # (the case tests/test_emu_late.py re-runs)
def test_emulated_sequences_longer_than_224_tokens(lib, prec, T, lengths, guided, loop):
    """VERDICT r05 "What's missing" 5: the reference is bounded by its positional table only (model/mdm.py:55, :251-253); this seam
    stopped at 224 tokens (exact-softmax attention kernels, sequence-sized GEMM tiles).  Longer sequences now run the GEMMs on row
    tiles and the attention with a streaming softmax over 32-key tiles (csrc/attention_long.h) in both arithmetic modes: ragged
    lengths (a count inside the first tiles: the tiles past it are not walked; a count in the last tile; a prefix mask of more than
    256 frames), a forward or a guided 2-step loop against the oracle, at the tolerances of the short-sequence cases."""
    B = len(lengths)
    sd = small_state_dict(num_layers=1)
    model, diffusion = make_pair(sd, 2, "cpu", guided=guided, native_lib=lib, precision=prec)
    y = synth_y(B, T, seed=3, lengths=lengths)
    g = torch.Generator().manual_seed(1)
    if loop:
        seq = [torch.randn(B, 263, 1, T, generator=g) for _ in range(3)]
        got = diffusion.p_sample_loop(model, (B, 263, 1, T), clip_denoised=False, model_kwargs={"y": dict(y)}, noise_sequence=seq)
        want = orc.sample_loop(sd, orc.Tables(orc.named_betas("cosine", 2)), (B, 263, 1, T), y, seq[0], seq[1:], cfg=True, num_heads=2)
        assert maxabs(got, want) < 5e-5
    else:
        x, t = torch.randn(B, 263, 1, T, generator=g), torch.tensor([1, 0][:B])
        assert maxabs(model(x, t, y=dict(y)), orc.mdm_forward(sd, x, t, y, num_heads=2)) < 5e-5


def test_emulated_dip_window_and_memory_longer_than_224_tokens(lib):
    """trans_dec: a 20 + 210-frame window (self-attention over 230 tokens: in_proj planes + the streaming kernel, lead = 0 with a
    frame count) and a 230-token text memory (cross-attention through the exact-fp32 streaming kernel) against the oracle."""
    B, C, P = 2, 20, 210
    sd = dip_small_state_dict(num_layers=1)
    model, _ = make_pair(sd, 2, "cpu", guided=False, native_lib=lib, context_len=C, pred_len=P, mask_frames=True)
    y = synth_dip_y(B, P, C, seed=4, text_lengths=[230, 9], lengths=[210, 77])
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(2))
    t = torch.tensor([1, 0])
    want = dip.dip_forward(sd, x, t, y, context_len=C, num_heads=2, mask_frames=True)
    assert maxabs(model(x, t, y=dict(y)), want) < 5e-5


def test_emulated_full_length_trans_dec_on_sequence_tiles(lib, engine_options):
    """The reference's full-length trans_dec checkpoint (README.md:254 humanml_trans_dec_512_bert-50steps: no prefix, 196 frames, plain
    p_sample_loop) at large batch: sequences of 129 .. 224 tokens, more of them than small_gemm_max_seqs, run the decoder's GEMMs on
    gemm_x3.h's sequence-sized tiles like the encoder (csrc/decoder.h dec_sequence_tiles; row statistics per 256 columns, cross-attention
    as q projection + exact-fp32 attention + out_proj).  Forced here with small_gemm_max_seqs = 1 on two sequences of 130 tokens (frame
    mask, ragged prompt): the guided forward and a two-step window loop (hoisted memory projections) against the oracle.  (The GPU suite
    holds both routes against the reference's own run at T = 196: tests/test_gpu_conditions.py.)"""
    B, C, P, steps = 1, 0, 130, 2
    sd = dip_small_state_dict(num_layers=2)
    y = synth_dip_y(B, P, 1, seed=3, text_lengths=[6], lengths=[97], scale=2.5)
    y.pop("prefix")
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([1])
    kw = dict(context_len=C, num_heads=2, mask_frames=True)
    engine_options(small_gemm_max_seqs=1)
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=0, mask_frames=True)
    assert maxabs(model(x, t, y=dict(y)), dip.dip_cfg_forward(sd, x, t, y, **kw)) < 5e-5
    g = torch.Generator().manual_seed(8)
    seq = [torch.randn(B, 263, 1, P, generator=g) for _ in range(1 + steps)]
    tab = orc.Tables(orc.named_betas("cosine", steps))
    got = diffusion.p_sample_loop(model, (B, 263, 1, P), clip_denoised=False, model_kwargs={"y": dict(y)}, noise_sequence=seq)
    assert maxabs(got, dip.dip_sample_loop(sd, tab, (B, 263, 1, P), y, seq[0], seq[1:], cfg=True, **kw)) < 5e-5

"""Round-5 GPU parity tests (`-m gpu`): the DiP operand-plane route under frame masks (what a `--mask_frames` checkpoint -- DiP.md:181,
the published DiP recipe -- hands over on every forward: model/mdm.py:241-247, :263-265), latent_dim 768 / 1024 on both encoder GEMM
kernels, a sampling loop captured into a hipGraph on a side stream after a warm-up on another one (ADVICE r04
medium), and the handle options that replaced the environment variables."""
import os

import numpy as np
import pytest
import torch

from helpers import (dip, make_pair, maxabs, memo, orc, small_state_dict, synth_dip_state_dict, synth_dip_y, synth_state_dict,
                     synth_y, to_dev)

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL_DIP_FWD, TOL_DIP_AR = 2e-5, 2e-4

ROUTES = {"planes32": {"small_gemm_row_tiles": 1}, "planes64": {"small_gemm_row_tiles": 2}, "skeleton": {"small_gemm_max_seqs": 0}}


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from mdm_amd import _native
    assert _native.load_native().path.endswith("libmdm_hip.so")


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("route", ["planes32", "planes64", "skeleton"])
def test_dip_masked_forward_on_every_route_matches_reference_golden(golden_dir, engine_options, route):
    """`dip_fwd_masked_B3` is the UPSTREAM reference's own output of a trans_dec forward with mask_frames=True and ragged lengths.
    Round 4 could only run it on the fp32 skeleton; now the operand-plane route (both tile heights) takes the mask too.  Plus a
    mask with holes (bitmap form of `lengths`) against the oracle on the same route."""
    engine_options(**ROUTES[route])
    sd_dip = memo("sd_dip0", lambda: synth_dip_state_dict(seed=0))
    g = _g(golden_dir, "dip_fwd_masked_B3")
    B = 3
    model, _ = make_pair(sd_dip, 10, DEV, guided=True, context_len=20, pred_len=40, mask_frames=True)
    assert model.model.engine().get_option("small_gemm_max_seqs") == ROUTES[route].get("small_gemm_max_seqs", 80)
    y_cpu = synth_dip_y(B, 40, 20, seed=int(g["y_seed"]), text_lengths=list(g["text_lengths"]), lengths=list(g["lengths"]))
    y = to_dev(y_cpu, DEV)
    x_cpu = torch.randn(B, 263, 1, 40, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    x, t = x_cpu.to(DEV), torch.from_numpy(g["t"]).to(DEV)
    e_ref = maxabs(model.model(x, t, y=dict(y)).cpu(), g["out_cond"])
    yh_cpu = dict(y_cpu)
    yh_cpu["mask"] = y_cpu["mask"].clone()
    yh_cpu["mask"][0, 0, 0, [3, 17, 18]] = False
    yh_cpu["mask"][2, 0, 0, [0]] = False
    want = dip.dip_cfg_forward(sd_dip, x_cpu, torch.from_numpy(g["t"]), yh_cpu, context_len=20, mask_frames=True)
    e_holes = maxabs(model(x, t, y=to_dev(yh_cpu, DEV)).cpu(), want)
    print(f"[parity] DiP masked forward B=3 ({route}) f16x3: max-abs vs reference = {e_ref:.3e}; holes (guided) vs oracle = {e_holes:.3e}")
    assert e_ref < TOL_DIP_FWD and e_holes < 5e-5


@pytest.mark.parametrize("B,C,P,text_lengths,lengths,rt", [(2, 0, 64, [70, 3], [64, 31], 1), (3, 8, 100, [5, 5, 12], [100, 2, 57], 2),
                                                            (5, 20, 40, [1, 9, 33, 17, 40], [40, 1, 40, 22, 39], 1)])
def test_dip_masked_plane_route_matches_oracle_shapes(engine_options, B, C, P, text_lengths, lengths, rt):
    """Ragged frame masks on the plane route at other window / memory shapes than 20 + 40 (no prefix; three key tiles of text; a
    108-token window = four 32-key tiles in the self-attention), both tile heights."""
    engine_options(small_gemm_row_tiles=rt)
    sd_dip = memo("sd_dip0", lambda: synth_dip_state_dict(seed=0))
    model, _ = make_pair(sd_dip, 10, DEV, guided=False, context_len=C, pred_len=P, mask_frames=True)
    y = synth_dip_y(B, P, max(C, 1), seed=B, text_lengths=text_lengths, lengths=lengths)
    if C == 0:
        y.pop("prefix")
    else:
        y["prefix"] = y["prefix"][..., :C].contiguous()
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(P))
    t = torch.arange(B) % 10
    want = dip.dip_forward(sd_dip, x, t, y, context_len=C, mask_frames=True)
    err = maxabs(model(x.to(DEV), t.to(DEV), y=to_dev(y, DEV)).cpu(), want)
    print(f"[parity] DiP masked forward B={B} C={C} P={P} rows x{32 * rt}: max-abs vs oracle = {err:.3e}")
    assert err < TOL_DIP_FWD


def test_dip_generate_mask_is_the_unmasked_result_bit_for_bit():
    """bench_dip.py's configuration (BASELINE.json configs[4] per GPU: B = 32, 196 frames = 5 windows x 10 steps, CFG 7.5) with the
    mask sample/generate.py:107 builds -- ones [B, 1, 1, 196] -- on a mask_frames=True model: a non-NULL, all-valid `lengths` on every
    forward.  Must reproduce the mask_frames=False model (NULL `lengths`) bit for bit on the same Philox streams: the masked plane
    route is the same arithmetic, and the benched number is the recipe's number."""
    from types import SimpleNamespace
    from mdm_amd.sampler_util import AutoRegressiveSampler
    sd_dip = memo("sd_dip0", lambda: synth_dip_state_dict(seed=0))
    B, frames, steps, C, P = 32, 196, 10, 20, 40
    g = torch.Generator().manual_seed(11)
    tl = [int(v) for v in torch.randint(3, 25, (B,), generator=g)]
    y = synth_dip_y(B, P, C, seed=43, text_lengths=tl, scale=7.5)
    y["mask"] = torch.ones(B, 1, 1, frames, dtype=torch.bool)
    y["lengths"] = torch.full((B,), frames)
    y = to_dev(y, DEV)
    args = SimpleNamespace(pred_len=P, context_len=C, autoregressive_include_prefix=False)
    outs = []
    for masked in (True, False):
        model, diffusion = make_pair(sd_dip, steps, DEV, guided=True, context_len=C, pred_len=P, mask_frames=masked)
        it = iter(range(7000, 7005))
        fn = lambda mdl, shape, **kw: diffusion.p_sample_loop(mdl, shape, seed=next(it), **kw)   # noqa: E731
        outs.append(AutoRegressiveSampler(args, fn, frames).sample(model, (B, 263, 1, frames), clip_denoised=False,
                                                                   model_kwargs={"y": y}).cpu())
        lengths = model.model._dec_inputs(torch.empty(B, 263, 1, P, device=DEV), y)[3]
        assert (lengths is not None) == masked and (not masked or bool((lengths == C + P).all()))
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("latent_dim", [768, 1024])
def test_wide_latent_dims_on_both_gemm_kernels(gemm_path, latent_dim):
    """latent_dim 768 / 1024 leave 6 / 8 partial LayerNorm statistics per row on the small-tile kernel (3 / 4 on
    the sequence tiles).  Forward + a short guided loop against the oracle on both kernels."""
    B, T, steps = 2, 60, 4
    sd = memo(("sd_wide", latent_dim), lambda: small_state_dict(latent_dim=latent_dim, num_layers=3))
    model, diffusion = make_pair(sd, steps, DEV, guided=True)
    y = synth_y(B, T, seed=2, lengths=[60, 17])
    g = torch.Generator().manual_seed(0)
    x, t = torch.randn(B, 263, 1, T, generator=g), torch.tensor([3, 0])
    H = latent_dim // 128
    e_f = maxabs(model(x.to(DEV), t.to(DEV), y=dict(y)).cpu(), orc.cfg_forward(sd, x, t, y, num_heads=H))
    shape = (B, 263, 1, T)
    x_T, noises = orc.make_noise(shape, steps, 5)
    out = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)},
                                  noise_sequence=[x_T] + [n.contiguous() for n in noises])
    want = orc.sample_loop(sd, orc.Tables(orc.named_betas("cosine", steps)), shape, y, x_T, noises, cfg=True, num_heads=H)
    e_l = maxabs(out.cpu(), want)
    print(f"[parity] latent_dim {latent_dim}, GEMM kernel {gemm_path}: forward {e_f:.3e}, {steps}-step loop {e_l:.3e} (max-abs vs oracle)")
    assert e_f < 5e-5 and e_l < 1e-4


@pytest.mark.parametrize("arch", ["trans_enc", "trans_dec"])
def test_sample_loop_is_captured_into_a_hip_graph_on_a_side_stream(arch):
    """include/mdm_hip.h "hipGraph CAPTURE" (ADVICE r04 medium): warm-up on the default stream, then the whole loop -- one
    mdm_sample_loop / mdm_sample_loop_dec call -- captured by torch.cuda.graph on ITS side stream (the per-device chain guard sees a
    stream change and must not wait on an event recorded outside the capture), replayed twice on fresh inputs, compared with the
    eager call bit for bit."""
    steps = 4
    if arch == "trans_enc":
        sd = memo("sd_enc0", lambda: synth_state_dict(seed=0))
        B, T = 2, 64
        model, diffusion = make_pair(sd, steps, DEV, guided=True)
        y = synth_y(B, T, seed=9, lengths=[64, 40])
    else:
        sd = memo("sd_dip0", lambda: synth_dip_state_dict(seed=0))
        B, T = 4, 40
        model, diffusion = make_pair(sd, steps, DEV, guided=True, context_len=20, pred_len=40, mask_frames=True)
        y = to_dev(synth_dip_y(B, 40, 20, seed=9, text_lengths=[5, 11, 3, 24], lengths=[40, 33, 40, 8]), DEV)
    y = to_dev(y, DEV) if arch == "trans_enc" else y
    shape = (B, 263, 1, T)
    diffusion.check_finite = False            # (the finite check is a host synchronisation: not capturable, not part of the loop)
    xs = [torch.randn(shape, generator=torch.Generator().manual_seed(s)).to(DEV) for s in (1, 2)]
    eager = [diffusion.p_sample_loop(model, shape, noise=xi, clip_denoised=False, model_kwargs={"y": dict(y)}, seed=77).clone()
             for xi in xs]                    # also the warm-up (default stream): workspace allocated, kernel attributes set
    torch.cuda.synchronize()
    static_in = torch.empty(shape, device=DEV)
    graph = torch.cuda.CUDAGraph()
    static_in.copy_(xs[0])
    with torch.cuda.graph(graph):
        static_out = diffusion.p_sample_loop(model, shape, noise=static_in, clip_denoised=False, model_kwargs={"y": dict(y)}, seed=77)
    for xi, want in zip(xs, eager):
        static_in.copy_(xi)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.isfinite(static_out).all() and torch.equal(static_out, want)
    # and the eager path still works afterwards (the guard's "previous stream" is the capture stream, which may be gone)
    again = diffusion.p_sample_loop(model, shape, noise=xs[1], clip_denoised=False, model_kwargs={"y": dict(y)}, seed=77)
    assert torch.equal(again, eager[1])


@pytest.mark.parametrize("fused,fused_sa", [(2, 1), (1, 1), (0, 1), (2, 0), (0, 0)])
def test_dip_fused_cross_attention_block_matches_reference_goldens(golden_dir, engine_options, fused, fused_sa):
    """csrc/xattn_block.h (the cross-attention block of a decoder layer as one kernel) and csrc/selfattn_block.h (in_proj +
    self-attention of a (sequence, head) as one kernel) against the UPSTREAM reference's own outputs: the B = 3 forwards (plain and
    frame-masked), the 100-frame autoregressive generation (3 windows x 10 steps, CFG 7.5) -- and the multi-launch forms they replace
    (dec_fused_xattn = 0 / dec_fused_selfattn = 0) on the same fixtures."""
    from types import SimpleNamespace
    from mdm_amd.sampler_util import AutoRegressiveSampler
    engine_options(dec_fused_xattn=fused, dec_fused_selfattn=fused_sa)
    sd_dip = memo("sd_dip0", lambda: synth_dip_state_dict(seed=0))
    errs = []
    for name, masked in (("dip_fwd_B3", False), ("dip_fwd_masked_B3", True)):
        g = _g(golden_dir, name)
        model, _ = make_pair(sd_dip, 10, DEV, guided=True, context_len=20, pred_len=40, mask_frames=masked)
        assert model.model.engine().get_option("dec_fused_xattn") == fused
        y = to_dev(synth_dip_y(3, 40, 20, seed=int(g["y_seed"]), text_lengths=list(g["text_lengths"]),
                               lengths=list(g["lengths"]) if masked else None), DEV)
        x = torch.randn(3, 263, 1, 40, generator=torch.Generator().manual_seed(int(g["x_seed"]))).to(DEV)
        t = torch.from_numpy(g["t"]).to(DEV)
        errs.append(maxabs(model.model(x, t, y=dict(y)).cpu(), g["out_cond"]))
        if not masked:
            errs.append(maxabs(model.model(x, t, y={**y, "uncond": True}).cpu(), g["out_uncond"]))
            e_g = maxabs(model(x, t, y=dict(y)).cpu(), g["out_cfg"])
    g = _g(golden_dir, "dip_ar10_B2_F100")
    steps, B, frames, seed = int(g["steps"]), int(g["B"]), int(g["frames"]), int(g["seed"])
    model, diffusion = make_pair(sd_dip, steps, DEV, guided=True, context_len=20, pred_len=40)
    y = to_dev(synth_dip_y(B, 40, 20, seed=int(g["y_seed"]), text_lengths=list(g["text_lengths"]), scale=float(g["scale"])), DEV)
    chunks = iter(dip.make_noise_chunks((B, 263, 1, 40), steps, seed, 3))

    def sample_fn(mdl, shape, **kw):
        x_T, eps = next(chunks)
        return diffusion.p_sample_loop(mdl, shape, noise_sequence=[x_T] + [e.contiguous() for e in eps], **kw)

    args = SimpleNamespace(pred_len=40, context_len=20, autoregressive_include_prefix=False)
    out = AutoRegressiveSampler(args, sample_fn, frames).sample(
        model, (B, 263, 1, frames), clip_denoised=False, model_kwargs={"y": y}, skip_timesteps=0, init_image=None,
        progress=False, dump_steps=None, noise=None, const_noise=False)
    e_ar = maxabs(out.cpu(), g["final"])
    print(f"[parity] DiP cross-attention block {['as three launches', 'as one kernel (xattn_block_kernel)', 'per (sequence, head) + out_proj GEMM'][fused]}, self-attention "
          f"{'fused (selfattn_block_kernel)' if fused_sa else 'as two launches'}: forwards vs reference "
          f"{errs[0]:.3e} / {errs[1]:.3e} / masked {errs[2]:.3e}, guided {e_g:.3e}; dip_ar10_B2_F100 {e_ar:.3e}")
    assert max(errs) < TOL_DIP_FWD and e_g < 5e-5 and e_ar < TOL_DIP_AR


@pytest.mark.parametrize("B,C,P,text_lengths,lengths", [(2, 0, 64, [70, 3], [64, 31]), (3, 8, 100, [5, 5, 12], [100, 2, 57]),
                                                         (4, 20, 40, [33, 40, 1, 17], None), (3, 0, 64, [64, 2, 31], [64, 64, 9])])
def test_dip_fused_cross_attention_block_other_shapes(engine_options, B, C, P, text_lengths, lengths):
    """The fused block at other shapes than DiP's 20 + 40 / 24 tokens: 70 and 40 memory tokens (3 / 2 key tiles), a 108-token window
    (32 + 32 + 32 + 12 rows), no prefix, against the oracle and against the three-launch form."""
    sd_dip = memo("sd_dip0", lambda: synth_dip_state_dict(seed=0))
    masked = lengths is not None
    y = synth_dip_y(B, P, max(C, 1), seed=B, text_lengths=text_lengths, lengths=lengths)
    if C == 0:
        y.pop("prefix")
    else:
        y["prefix"] = y["prefix"][..., :C].contiguous()
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(P))
    t = torch.arange(B) % 10
    want = dip.dip_forward(sd_dip, x, t, y, context_len=C, mask_frames=masked)
    outs = []
    for fused in (2, 1, 0):
        engine_options(dec_fused_xattn=fused)
        model, _ = make_pair(sd_dip, 10, DEV, guided=False, context_len=C, pred_len=P, mask_frames=masked)
        outs.append(model(x.to(DEV), t.to(DEV), y=to_dev(y, DEV)).cpu())
    e2, e1, e0 = (maxabs(o, want) for o in outs)
    print(f"[parity] DiP forward B={B} C={C} P={P} ntok={max(text_lengths)}: per (sequence, head) {e2:.3e}, one kernel {e1:.3e}, three launches {e0:.3e} (max-abs vs oracle)")
    assert max(e2, e1, e0) < TOL_DIP_FWD and not torch.equal(outs[1], outs[2])


@pytest.mark.parametrize("B,C,P,text_lengths,holes", [(5, 20, 40, [1, 9, 24, 17, 12], True), (3, 0, 64, [5, 9, 70], True), (4, 5, 12, [6, 3, 2, 9], False)])
def test_dip_fused_self_attention_block_other_shapes(engine_options, B, C, P, text_lengths, holes):
    """selfattn_block_kernel at DiP's window with ragged / holed frame masks, at the full 64-token tile without a prefix, at a 17-token
    window, against the oracle and against the two-launch form."""
    sd_dip = memo("sd_dip0", lambda: synth_dip_state_dict(seed=0))
    lengths = [max(P - 7 * i, 1) for i in range(B)]
    y = synth_dip_y(B, P, max(C, 1), seed=B, text_lengths=text_lengths, lengths=lengths)
    if C == 0:
        y.pop("prefix")
    else:
        y["prefix"] = y["prefix"][..., :C].contiguous()
    if holes:
        y["mask"] = y["mask"].clone()
        y["mask"][0, 0, 0, [1, 4, 9]] = False
        y["mask"][B - 1, 0, 0, [0]] = False
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(P))
    t = torch.arange(B) % 10
    want = dip.dip_forward(sd_dip, x, t, y, context_len=C, mask_frames=True)
    outs = []
    for fused in (1, 0):
        engine_options(dec_fused_selfattn=fused)
        model, _ = make_pair(sd_dip, 10, DEV, guided=False, context_len=C, pred_len=P, mask_frames=True)
        outs.append(model(x.to(DEV), t.to(DEV), y=to_dev(y, DEV)).cpu())
    e1, e0 = maxabs(outs[0], want), maxabs(outs[1], want)
    print(f"[parity] DiP forward B={B} C={C} P={P} masked{' + holes' if holes else ''}: fused self-attention {e1:.3e}, two launches {e0:.3e} (max-abs vs oracle)")
    assert e1 < TOL_DIP_FWD and e0 < TOL_DIP_FWD


_DIRECT_BIG = {}


@pytest.mark.parametrize("direct", [1, 0])
def test_attention_direct_output_matches_reference_goldens(golden_dir, engine_options, direct):
    """attention_x3.h DIRECT (planes straight from the accumulators; the next item's key tiles requested in front of the stores; counted
    waits that include the stores in flight) against the UPSTREAM reference's forward and 50-step loop goldens at T = 196 on the
    sequence-tile route, and -- same arithmetic -- bit-identical to the staged form at the headline shape's kernel (B = 24 guided:
    192 items over 512 persistent workgroups is one item each; B = 80: 640 items, every workgroup carries a second one)."""
    from helpers import golden_loop_inputs, run_product_loop
    engine_options(attn_direct_out=direct, small_gemm_max_seqs=0)
    sd = memo("sd_enc0", lambda: synth_state_dict(seed=0))
    g = _g(golden_dir, "fwd_B3_T196")
    B, T = 3, 196
    y = synth_y(B, T, seed=int(g["y_seed"]), lengths=list(g["lengths"]))
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    t = torch.from_numpy(g["t"])
    model, _ = make_pair(sd, 50, DEV, guided=True)
    assert model.model.engine().get_option("attn_direct_out") == direct
    e_c = maxabs(model.model(x.to(DEV), t.to(DEV), y=dict(y)).cpu(), g["out_cond"])
    e_g = maxabs(model(x.to(DEV), t.to(DEV), y=dict(y)).cpu(), g["out_cfg"])
    gl = _g(golden_dir, "loop50_B2_T196")
    out = run_product_loop(sd, golden_loop_inputs(gl), DEV)
    e_l = maxabs(out.cpu(), gl["final"])
    # a batch whose (sequence, head) items outnumber the persistent workgroups: the carried-item path
    B2 = 80
    y2 = synth_y(B2, T, seed=5, lengths=[196 - (7 * i) % 150 for i in range(B2)])
    x2 = torch.randn(B2, 263, 1, T, generator=torch.Generator().manual_seed(3)).to(DEV)
    t2 = (torch.arange(B2) % 50).to(DEV)
    big = model(x2, t2, y=dict(y2)).cpu()
    other = _DIRECT_BIG.get(1 - direct)
    _DIRECT_BIG[direct] = big
    print(f"[parity] attention {'DIRECT' if direct else 'staged'} output: forward vs reference {e_c:.3e} / guided {e_g:.3e}; loop50_B2_T196 {e_l:.3e}")
    assert e_c < 3e-5 and e_g < 1.2e-4 and e_l < 1e-4 and torch.isfinite(big).all()
    if other is not None:
        assert torch.equal(big, other)

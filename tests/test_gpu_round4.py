"""Round-4 GPU parity tests (`-m gpu`): the branches of the path no test had reached: `clip_denoised=True`
(the reference's signature default), FIXED_LARGE variances, per-sample timesteps in `p_sample`, the `trans_dec` denoiser with a
single CLIP token as its memory -- each against fixtures the UPSTREAM REFERENCE produced (oracle/make_golden_r4.py,
tests/golden/PIN_REPORT_r4.json) -- the stand-alone ctypes binding of INTEGRATION.md section 2 executed as written, the RCCL
bring-up of bench.py as a one-rank group, and the small-batch latency path against the oracle."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from helpers import (ROOT, dip, make_pair, maxabs, memo, orc, synth_dip_state_dict, synth_state_dict, synth_y, to_dev)

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
PRECISIONS = ["f16x3", "f32"]
TOL_LOOP = 1e-4          # stated bar (BASELINE.json): 1e-3


@pytest.fixture(scope="module")
def sd():
    return synth_state_dict(seed=0)


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from mdm_amd import _native
    assert _native.load_native().path.endswith("libmdm_hip.so")


def _g(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


@pytest.mark.parametrize("name", ["clip50_B2_T64", "ddimclip50_B2_T64", "fixedlarge50_B2_T64"])
@pytest.mark.parametrize("prec", PRECISIONS)
def test_clip_denoised_and_fixed_large_loops_match_reference(golden_dir, sd, name, prec):
    """gaussian_diffusion.py:347-353 (the clamp of pred_xstart inside p_mean_variance, feeding the posterior mean AND DDIM's eps)
    and :325-333 (FIXED_LARGE: sigma_t^2 = beta_t, posterior variance at t = 1).  The clamp is live on these trajectories:
    PIN_REPORT_r4.json `clamp_effect` is the distance to the unclipped loop on the same noise."""
    g = _g(golden_dir, name)
    steps, B, T, seed = int(g["steps"]), int(g["B"]), int(g["T"]), int(g["seed"])
    shape = (B, 263, 1, T)
    y = synth_y(B, T, seed=seed + 1000, lengths=list(g["lengths"]), scale=float(g["scale"]))
    model, diffusion = make_pair(sd, steps, DEV, guided=True, precision=prec, sigma_small=not bool(g["fixed_large"]))
    assert diffusion.model_var_type.name == ("FIXED_LARGE" if bool(g["fixed_large"]) else "FIXED_SMALL")
    x_T, noises = orc.make_noise(shape, steps, seed)
    kw = dict(clip_denoised=bool(g["clip"]), model_kwargs={"y": dict(y)},
              noise_sequence=[x_T] + [n.contiguous() for n in noises])
    out = (diffusion.ddim_sample_loop if bool(g["ddim"]) else diffusion.p_sample_loop)(model, shape, **kw)
    err = maxabs(out.cpu(), g["final"])
    print(f"[parity] {name} {prec}: max-abs vs reference = {err:.3e}")
    assert err < TOL_LOOP
    if bool(g["clip"]):   # ... and the unclipped product loop on the same noise is far from it: the branch was taken
        kw["clip_denoised"] = False
        free = (diffusion.ddim_sample_loop if bool(g["ddim"]) else diffusion.p_sample_loop)(model, shape, **kw)
        assert maxabs(free.cpu(), g["final"]) > 100 * TOL_LOOP


def test_clip_denoised_default_is_true_like_the_reference(golden_dir, sd):
    """p_sample_loop's signature default (gaussian_diffusion.py:596): leaving the argument out must give the clipped loop."""
    g = _g(golden_dir, "clip50_B2_T64")
    steps, B, T, seed = int(g["steps"]), int(g["B"]), int(g["T"]), int(g["seed"])
    shape = (B, 263, 1, T)
    y = synth_y(B, T, seed=seed + 1000, lengths=list(g["lengths"]), scale=float(g["scale"]))
    model, diffusion = make_pair(sd, steps, DEV, guided=True)
    x_T, noises = orc.make_noise(shape, steps, seed)
    out = diffusion.p_sample_loop(model, shape, model_kwargs={"y": dict(y)}, noise_sequence=[x_T] + [n.contiguous() for n in noises])
    assert maxabs(out.cpu(), g["final"]) < TOL_LOOP


@pytest.mark.parametrize("prec", PRECISIONS)
def test_p_sample_with_per_sample_timesteps_matches_reference(golden_dir, sd, prec):
    """gaussian_diffusion.py:489-541 with t = [49, 25, 0]: coefficients gathered per sample, no noise at t = 0, with and without
    the clamp -- against the reference's own p_sample on the same generator state (round 3: emulator only)."""
    g = _g(golden_dir, "psample_mixed_t_B3_T32")
    steps, B, T, seed = int(g["steps"]), int(g["B"]), int(g["T"]), int(g["seed"])
    shape = (B, 263, 1, T)
    y = to_dev(synth_y(B, T, seed=seed + 1000, lengths=list(g["lengths"]), scale=float(g["scale"])), DEV)
    x = (torch.randn(*shape, generator=torch.Generator().manual_seed(seed + 1)) * 1.3).to(DEV)
    t = torch.from_numpy(g["t"]).to(DEV)
    eps = torch.from_numpy(g["noise"]).to(DEV)
    model, diffusion = make_pair(sd, steps, DEV, guided=True, precision=prec)
    for clip, ks, kx in ((False, "sample", "pred_xstart"), (True, "sample_clip", "pred_xstart_clip")):
        out = diffusion.p_sample(model, x, t, clip_denoised=clip, model_kwargs={"y": dict(y)}, noise=eps)
        e1, e2 = maxabs(out["sample"].cpu(), g[ks]), maxabs(out["pred_xstart"].cpu(), g[kx])
        print(f"[parity] p_sample mixed t, clip={clip}, {prec}: sample {e1:.2e}, pred_xstart {e2:.2e}")
        assert e1 < 1e-4 and e2 < 1e-4
        assert maxabs(out["sample"][2].cpu(), out["pred_xstart"][2].cpu()) == 0.0       # t = 0: no noise, coef1 = 1
    assert float(np.abs(g["pred_xstart_clip"]).max()) == 1.0 and float(np.abs(g["pred_xstart"]).max()) > 1.0


def _dip_clip_y(B, seed, scale=7.5):
    g = torch.Generator().manual_seed(seed)
    return {"mask": torch.ones(B, 1, 1, 40, dtype=torch.bool), "lengths": torch.full((B,), 40),
            "text_embed": torch.randn(1, B, 512, generator=g), "prefix": torch.randn(B, 263, 1, 20, generator=g),
            "scale": torch.ones(B) * scale}


@pytest.mark.parametrize("prec", PRECISIONS)
def test_trans_dec_with_a_clip_memory_token_matches_reference(golden_dir, prec):
    """model/mdm.py:85-93, :261-262: `arch='trans_dec'` with `text_encoder_type='clip'` -- the decoder's memory is ONE token per
    sample (embed_text(clip feature) + time embedding), no memory padding mask.  Forwards and a 10-step window loop against
    the reference's outputs (mdm.py's docstring claimed this configuration since round 2; nothing constructed it)."""
    g = _g(golden_dir, "dip_clip_fwd_B3")
    sdc = synth_dip_state_dict(seed=int(g["sd_seed"]), bert_dim=512)
    model, _ = make_pair(sdc, 10, DEV, guided=True, precision=prec, text_encoder_type="clip", context_len=20, pred_len=40)
    assert model.model.arch == "trans_dec" and model.model.clip_dim == 512
    B = 3
    y = to_dev(_dip_clip_y(B, int(g["y_seed"])), DEV)
    x = torch.randn(B, 263, 1, 40, generator=torch.Generator().manual_seed(int(g["x_seed"]))).to(DEV)
    t = torch.from_numpy(g["t"]).to(DEV)
    oc = model.model(x, t, y=dict(y))
    ou = model.model(x, t, y={**y, "uncond": True})
    og = model(x, t, y=dict(y))
    errs = (maxabs(oc.cpu(), g["out_cond"]), maxabs(ou.cpu(), g["out_uncond"]), maxabs(og.cpu(), g["out_cfg"]))
    print(f"[parity] trans_dec + CLIP memory {prec}: cond {errs[0]:.2e} uncond {errs[1]:.2e} cfg {errs[2]:.2e}")
    assert errs[0] < 3e-5 and errs[1] < 3e-5 and errs[2] < 4e-4          # CFG 7.5: (2s - 1) = 14x the branch errors
    gl = _g(golden_dir, "dip_clip_loop10_B2")
    B, steps, seed = int(gl["B"]), int(gl["steps"]), int(gl["seed"])
    model, diffusion = make_pair(sdc, steps, DEV, guided=True, precision=prec, text_encoder_type="clip", context_len=20, pred_len=40)
    y = to_dev(_dip_clip_y(B, int(gl["y_seed"])), DEV)
    shape = (B, 263, 1, 40)
    x_T, noises = orc.make_noise(shape, steps, seed)
    out = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)},
                                  noise_sequence=[x_T] + [n.contiguous() for n in noises])
    err = maxabs(out.cpu(), gl["final"])
    print(f"[parity] trans_dec + CLIP memory, 10-step window loop {prec}: {err:.2e}")
    assert err < 2e-4


@pytest.mark.parametrize("route", ["planes32", "planes64", "skeleton"])
def test_dip_decoder_routes_match_reference_goldens(golden_dir, engine_options, route):
    """The f16x3 trans_dec stack on its two routes (csrc/decoder.h dec_on_planes): operand planes through gemm_x3s.h +
    attention_x3.h -- what the DiP callers' sizes take; 32- and 64-row tiles -- and the fp32 skeleton of gemm_f32.h (forced here by
    small_gemm_max_seqs = 0).  Each against the UPSTREAM reference's own outputs: the
    B = 3 forward (both branches, guided) and the 100-frame autoregressive generation (3 windows x 10 steps, CFG 7.5)."""
    from types import SimpleNamespace
    from helpers import synth_dip_y
    from mdm_amd.sampler_util import AutoRegressiveSampler
    engine_options(**{"planes32": {"small_gemm_row_tiles": 1}, "planes64": {"small_gemm_row_tiles": 2},
                      "skeleton": {"small_gemm_max_seqs": 0}}[route])
    sd_dip = memo("sd_dip0", lambda: synth_dip_state_dict(seed=0))
    g = _g(golden_dir, "dip_fwd_B3")
    model, _ = make_pair(sd_dip, 10, DEV, guided=True, context_len=20, pred_len=40, mask_frames=False)
    y = to_dev(synth_dip_y(3, 40, 20, seed=int(g["y_seed"]), text_lengths=list(g["text_lengths"]), lengths=None), DEV)
    x = torch.randn(3, 263, 1, 40, generator=torch.Generator().manual_seed(int(g["x_seed"]))).to(DEV)
    t = torch.from_numpy(g["t"]).to(DEV)
    e_c = maxabs(model.model(x, t, y=dict(y)).cpu(), g["out_cond"])
    e_u = maxabs(model.model(x, t, y={**y, "uncond": True}).cpu(), g["out_uncond"])
    e_g = maxabs(model(x, t, y=dict(y)).cpu(), g["out_cfg"])
    print(f"[parity] DiP forward B=3 ({route}) f16x3: max-abs vs reference = {e_c:.3e} / {e_u:.3e} / {e_g:.3e} (cond / uncond / guided)")
    assert e_c < 2e-5 and e_u < 2e-5 and e_g < 5e-5
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
    err = maxabs(out.cpu(), g["final"])
    print(f"[parity] dip_ar10_B2_F100 ({route}) f16x3: max-abs vs reference = {err:.3e}")
    assert err < 2e-4


def test_integration_md_ctypes_snippet_runs_as_written(tmp_path):
    """the stand-alone ctypes binding of `mdm_sample_loop` -- the fenced python block is extracted and
    executed verbatim in a fresh interpreter (only torch for device memory); it must print its own `max-abs vs oracle` line."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    m = re.search(r"<!-- ctypes-snippet-begin -->\s*```python\n(.*?)```\s*<!-- ctypes-snippet-end -->", text, flags=re.S)
    assert m, "INTEGRATION.md lost its marked ctypes snippet"
    script = tmp_path / "snippet.py"
    script.write_text(m.group(1))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, str(script)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    mm = re.search(r"max-abs vs oracle = ([0-9.eE+-]+)", r.stdout)
    assert mm and float(mm.group(1)) < 1e-4, r.stdout[-2000:]


def test_bench_force_pg_brings_up_rccl_as_a_one_rank_group():
    """SURVEY 8e / VERDICT r03 item 7: the multi-GPU leg of bench.py (process group on backend 'nccl' == RCCL,
    all_gather_into_tensor of the shards, barriers) exercised on the one GPU this box has, every round: `--force-pg` joins a
    ONE-rank group and runs the same code the N > 1 path runs.  The line must say so and carry per-rank loop times."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-pg", "--steps", "1", "--warmup", "1", "--batch", "16",
                        "--quick"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["ranks"]["backend"].startswith("nccl") and line["ranks"]["world_size"] == 1 and line["n_gpus"] == 1
    assert line["ranks"]["gathered_shape"][0] == 16
    assert len(line["ranks"]["loop_ms_per_rank"]) == 1 and line["ranks"]["loop_ms_min"] <= line["ranks"]["loop_ms_max"]
    assert line["value"] > 0


@pytest.mark.parametrize("B,lengths", [(1, [196]), (6, [196, 120, 57, 196, 33, 180])])
def test_small_batch_loop_matches_oracle(sd, gemm_path, B, lengths):
    """The latency regime (sample/generate.py's default `--num_samples 6`; README.md:13's per-call latency; BASELINE.json
    configs[0]'s batch 1): the 50-step CFG loop at T = 196, B = 1 and B = 6, on BOTH encoder GEMM kernels -- csrc/gemm_x3s.h's
    32 / 64-row tiles (what runs up to 80 sequences) and csrc/gemm_x3.h's sequence tiles -- against the oracle on the same
    injected noise; and the two kernels against each other."""
    steps, T = 50, 196
    shape = (B, 263, 1, T)
    y = synth_y(B, T, seed=400 + B, lengths=lengths)
    x_T, noises = orc.make_noise(shape, steps, 40 + B)
    model, diffusion = make_pair(sd, steps, DEV, guided=True)
    out = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)},
                                  noise_sequence=[x_T] + [n.contiguous() for n in noises])
    want = memo(("small_batch", B), lambda: orc.sample_loop(sd, orc.Tables(orc.named_betas("cosine", steps)), shape, y, x_T, noises, cfg=True))
    err = maxabs(out.cpu(), want)
    print(f"[parity] small-batch loop B={B} T=196 50 steps, GEMM kernel {gemm_path}: max-abs vs oracle = {err:.3e}")
    assert err < TOL_LOOP


def test_operand_split_is_bit_exact_fp16_hi_plus_lo():
    """common.h split2_p16 (round 4: lo = fp16(x - float(hi)) in one v_fma_mix{lo,hi}_f16 per value): the planes a kernel writes
    must be EXACTLY hi = rne16(x), lo = rne16(x - hi) -- checked bit for bit against numpy over normal, tiny (fp16-subnormal
    hi and lo), large and signed values, through the planes mdm_linear_x3 leaves in its scratch (hi at 0, lo at M*K)."""
    from mdm_amd import _native
    lib = _native.load_native()
    M, N, K = 64, 32, 64
    rng = np.random.default_rng(0)
    x = rng.standard_normal((M, K)).astype(np.float32)
    x[0] *= 1e-3; x[1] *= 1e-5; x[2] *= 3e-8; x[3] *= 1e3; x[4] *= 6e4 / np.abs(x[4]).max(); x[5] = 0.0
    x[6, :8] = [65504.0, -65504.0, 6.1e-5, -6.1e-5, 5.96e-8, 1.0, -1.0, 0.33333334]
    xt = torch.from_numpy(x).to(DEV)
    w = torch.zeros(N, K, device=DEV); b = torch.zeros(N, device=DEV); out = torch.empty(M, N, device=DEV)
    nb = lib.mdm_linear_x3_scratch_bytes(M, N, K)
    scratch = torch.zeros(nb, dtype=torch.uint8, device=DEV)
    lib.check(lib.mdm_linear_x3(xt.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), M, N, K, 0, scratch.data_ptr(), nb,
                                torch.cuda.current_stream().cuda_stream), "mdm_linear_x3")
    torch.cuda.synchronize()
    planes = scratch.cpu().numpy()[: M * K * 4].view(np.uint16)
    hi, lo = planes[: M * K].reshape(M, K), planes[M * K: 2 * M * K].reshape(M, K)
    want_hi = x.astype(np.float16)
    want_lo = (x - want_hi.astype(np.float32)).astype(np.float16)
    assert np.array_equal(hi, want_hi.view(np.uint16))
    # (-0.0 vs +0.0 in lo is value-identical: compare values where both are zero, bits elsewhere)
    z = (want_lo == 0) & (lo.view(np.float16) == 0)
    assert np.array_equal(lo[~z], want_lo.view(np.uint16)[~z])


def test_wide_form_of_the_pipelined_gemm_matches_reference(golden_dir, sd, monkeypatch, engine_options):
    """gemm_x3.h NCB = 2 (probe library, MDM_X3_WIDE=1): four waves x 64 columns, one wave per SIMD, accumulators in AGPRs -- the
    arrangement VERDICT r03 item 1a asked for.  12 % slower over the loop (profiles/r04d_wide.md), so not a product path; this
    keeps it correct against the reference's forward goldens on the sequence-tile kernel."""
    from mdm_amd import _native
    monkeypatch.setenv("MDM_X3_WIDE", "1")       # (an experiment switch of the probe library; the product reads no environment)
    engine_options(small_gemm_max_seqs=0)
    g = _g(golden_dir, "fwd_B3_T196")
    B, T = 3, 196
    y = synth_y(B, T, seed=int(g["y_seed"]), lengths=list(g["lengths"]))
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    t = torch.from_numpy(g["t"])
    model, _ = make_pair(sd, 50, DEV, guided=True, native_lib=_native.load_probe())
    og = model(x.to(DEV), t.to(DEV), y=dict(y))
    oc = model.model(x.to(DEV), t.to(DEV), y=dict(y))
    assert maxabs(oc.cpu(), g["out_cond"]) < 3e-5 and maxabs(og.cpu(), g["out_cfg"]) < 1.2e-4

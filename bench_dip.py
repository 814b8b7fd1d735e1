"""DiP benchmark (SURVEY.md 8f row 1 / BASELINE.json configs[4]): motions/sec of autoregressive text-to-motion with the
trans_dec denoiser -- 196 frames = 5 prediction windows of 40 frames (20-frame prefix), 10 diffusion steps per window,
classifier-free guidance (2 denoiser forwards per step), B motions per GPU (256 over 8 GPUs = 32 per GPU).

    python bench_dip.py [--gpus N] [--steps K] [--warmup W] [--batch B]

Same contract as bench.py (one JSON line on rank 0; a "step" is one whole 196-frame generation of one batch); it is a
separate file because bench.py is the driver's headline-metric entry point (bench.py embeds `measure()`'s record as its
`dip` sub-line).  `roofline` prices the dominant kernel class (the decoder GEMMs); `launches_per_motion_batch` / `kernel_ms`
show where the wall time goes.
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import mdm_amd  # noqa: E402,F401
from mdm_amd import dist as mdist  # noqa: E402
from mdm_amd import model_util  # noqa: E402
from mdm_amd.sampler_util import AutoRegressiveSampler, ClassifierFreeSampleModel  # noqa: E402

CONTEXT, PRED, FRAMES, DSTEPS, NTOK = 20, 40, 196, 10, 24


def synthetic_y(B, device, seed):
    g = torch.Generator().manual_seed(seed)
    tl = torch.randint(6, NTOK + 1, (B,), generator=g)
    tl[0] = NTOK
    # sample/generate.py:107 collates `lengths = n_frames` for every sample: y['mask'] = ones [B, 1, 1, 196].  A model trained with
    # `--mask_frames` (DiP.md:181, the published DiP recipe) turns that into an (all-valid) tgt_key_padding_mask on EVERY forward
    # (model/mdm.py:241-247): the window loop below runs with a non-NULL `lengths` array.
    return {"mask": torch.ones(B, 1, 1, FRAMES, dtype=torch.bool, device=device),
            "lengths": torch.full((B,), FRAMES, dtype=torch.long, device=device), "text": ["synthetic prompt"] * B,
            "text_embed": (torch.randn(NTOK, B, 768, generator=g).to(device),
                           (torch.arange(NTOK)[None, :] >= tl[:, None]).to(device)),
            "prefix": torch.randn(B, 263, 1, CONTEXT, generator=g).to(device),
            "scale": torch.full((B,), 7.5, device=device)}


def cpu_baseline(state, budget_s=12.0):
    """oracle/dip_oracle.py (CPU restatement pinned to the reference) on the host cores: CFG denoiser steps of one
    60-token window at B=4, scaled to whole 196-frame motions (5 windows x 10 steps)."""
    from oracle import dip_oracle as dip
    from oracle import mdm_oracle as orc
    from oracle.synth import synth_dip_y
    B = 4
    sd = {k: v.detach().cpu().float() for k, v in state.items() if "pos_encoder" not in k}
    y = synth_dip_y(B, PRED, CONTEXT, seed=1, text_lengths=[NTOK, 9, 15, 12])
    tab = orc.Tables(orc.named_betas("cosine", DSTEPS))
    x = torch.randn(B, 263, 1, PRED)
    all_thr = torch.get_num_threads()
    torch.set_num_threads(min(16, all_thr))     # every hardware thread of the GPU box (128+) is 5-10x SLOWER at these sizes (bench.py cpu_baseline)
    x = orc.ddpm_step(tab, x, dip.dip_cfg_forward(sd, x, torch.full((B,), DSTEPS - 1, dtype=torch.long), y, context_len=CONTEXT),
                      torch.full((B,), DSTEPS - 1, dtype=torch.long), torch.randn_like(x))      # warm-up step
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        t = torch.full((B,), DSTEPS - 1 - n % DSTEPS, dtype=torch.long)
        x = orc.ddpm_step(tab, x, dip.dip_cfg_forward(sd, x, t, y, context_len=CONTEXT), t, torch.randn_like(x))
        n += 1
    per_step = (time.perf_counter() - t0) / n
    used = torch.get_num_threads()
    torch.set_num_threads(all_thr)
    return {"value": B / (per_step * DSTEPS * 5), "unit": "motions/s", "cores": used, "kind": "port",
            "sample": f"oracle CFG p_sample of one 60-token window: {n} diffusion steps at B={B}, scaled to 5 windows x "
                      f"{DSTEPS} steps per motion ({per_step * 1e3:.0f} ms per batch-step)"}


DIP_PMC_PROFILE = os.path.join("profiles", "r06_dip_pmc.json")


def pmc_traffic_per_launch():
    """Fabric-side bytes per launch of the decoder's GEMM-class kernels (the `linear` profiling class: gemm_x3s_kernel, the
    (sequence, head) attention blocks, xattn_block_kernel) from the committed rocprofv3 PMC passes of THIS command at B = 32
    (tools/gpu_dip_pmc.sh + tools/dip_pmc_to_json.py), call-weighted; quoted only while the kernel sources are the ones the
    passes were taken on (bench.csrc_sha256)."""
    path = os.path.join(ROOT, DIP_PMC_PROFILE)
    if not os.path.isfile(path):
        return None, f"{DIP_PMC_PROFILE} absent"
    try:
        import bench
        with open(path) as f:
            d = json.load(f)
        if d.get("csrc_sha256") != bench.csrc_sha256():
            return None, f"{DIP_PMC_PROFILE}: taken on OTHER kernel sources / build flags, not quoted"
        rows = [v for k, v in d["kernels"].items() if ("gemm_x3s" in k or "seqhead" in k or "cross_attention_block" in k)
                and "fabric_bytes" in v and v.get("calls")]
        calls = sum(v["calls"] for v in rows)
        return int(sum(v["fabric_bytes"] * v["calls"] for v in rows) / calls), f"{DIP_PMC_PROFILE} (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE of this command)"
    except (KeyError, ValueError, OSError, ZeroDivisionError) as e:
        return None, f"{DIP_PMC_PROFILE} unreadable ({type(e).__name__})"


def measure_small_batch(model, diffusion, mdm, dev, sync, state, batches=(1, 6), passes=5, cpu=True):
    """The reference's OWN published DiP metric (DiP.md:15 -> assets/dip_spec.png: 11 ms per 40-frame prediction call and ~3,500
    frames/s on an RTX 3090 -- other hardware, quoted as context, never a target or a `vs_baseline`): milliseconds per
    window call (ONE `p_sample_loop` of a 40-frame window behind a 20-frame prefix: 10 DDPM steps, CFG 7.5 = 20 denoiser
    forwards; what `AutoRegressiveSampler` calls once per 2 s of motion) and per 196-frame motion (5 such calls) at batch 1 and 6
    (sample/generate.py:76 `--num_samples`' default), mean of `passes` after one warm-up; and the oracle port's host time for the
    same window call at batch 1 (checker code timed as a reported baseline only)."""
    out = {}
    for B in batches:
        y = synthetic_y(B, dev, 4000 + B)
        sampler = AutoRegressiveSampler(SimpleNamespace(pred_len=PRED, context_len=CONTEXT, autoregressive_include_prefix=False),
                                        diffusion.p_sample_loop, FRAMES)
        win = lambda: diffusion.p_sample_loop(model, (B, 263, 1, PRED), clip_denoised=False, model_kwargs={"y": y})   # noqa: E731
        gen = lambda: sampler.sample(model, (B, 263, 1, FRAMES), clip_denoised=False, model_kwargs={"y": y})          # noqa: E731
        rec = {}
        diffusion.check_finite = False        # (the seam's finite check syncs per call: asserted behind the clock instead)
        try:
            for name, fn, n in (("window_call_ms", win, passes * 5), ("motion_196_frames_ms", gen, passes)):
                x = fn()
                sync()
                t0 = time.perf_counter()
                for _ in range(n):
                    x = fn()
                sync()
                rec[name] = round((time.perf_counter() - t0) / n * 1e3, 3)
                assert bool(torch.isfinite(x).all())
        finally:
            diffusion.check_finite = True
        rec["frames_per_s"] = round(B * FRAMES / (rec["motion_196_frames_ms"] * 1e-3), 1)
        out[f"B{B}"] = rec
    if cpu:
        from oracle import dip_oracle as dip
        from oracle import mdm_oracle as orc
        from oracle.synth import synth_dip_y
        sd = {k: v.detach().cpu().float() for k, v in state.items() if "pos_encoder" not in k}
        y1 = synth_dip_y(1, PRED, CONTEXT, seed=1, text_lengths=[NTOK])
        tab = orc.Tables(orc.named_betas("cosine", DSTEPS))
        g = torch.Generator().manual_seed(0)
        seq = [torch.randn(1, 263, 1, PRED, generator=g) for _ in range(1 + DSTEPS)]
        call = lambda: dip.dip_sample_loop(sd, tab, (1, 263, 1, PRED), y1, seq[0], seq[1:], context_len=CONTEXT, cfg=True)  # noqa: E731
        # torch's default intra-op thread count is every hardware thread of the box; at batch 1 (60-row GEMMs) that over-subscription
        # is SLOWER than a few cores (bench.py cpu_baseline measured the same): 8 and 16 threads, the better one is the value
        all_thr, best = torch.get_num_threads(), None
        for nthr in sorted({min(8, all_thr), min(16, all_thr)}):
            torch.set_num_threads(nthr)
            call()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                call()
                ts.append(time.perf_counter() - t0)
            if best is None or min(ts) < best[0]:
                best = (min(ts), nthr)
        torch.set_num_threads(all_thr)
        out["cpu_window_call_B1"] = {"ms": round(best[0] * 1e3, 1), "cores": best[1], "kind": "port",
                                     "sample": "oracle/dip_oracle.dip_sample_loop, one 40-frame window call at batch 1 (10 steps, CFG), best of 3 "
                                               "at the better of 8 / 16 intra-op threads"}
    out["config"] = {"workload": f"DiP window call = p_sample_loop of a {PRED}-frame window behind a {CONTEXT}-frame prefix: {DSTEPS} DDPM steps, "
                                 f"CFG 7.5, {NTOK}-token DistilBERT memory (cached), mask_frames=True; motion = {FRAMES} frames = 5 calls",
                     "published_context": "DiP.md:15 / assets/dip_spec.png (upstream, RTX 3090): 11 ms per 40-frame call, ~3,500 frames/s "
                                          "-- other hardware, not a target"}
    return out


def measure(dev, rank, world, B, steps, warmup, cpu=True, mask_frames=True, engine_options=None, native_lib=None, tiny=False,
            small_batch=False):
    """Time `steps` whole 196-frame generations of B motions per rank on `dev`; returns the JSON record (rank 0) or None.
    bench.py embeds this record as its `dip` sub-line so that the driver's BENCH file carries it.

    `tiny` + `native_lib` (test infrastructure: `bench.py --emulate`): the same launcher / sharding / gather / record code on gloo
    ranks with the kernels in the CPU emulator and a one-layer d = 256 model, 12 frames = 1 window x 2 steps.  Never a measurement.

    N > 1: a failure on ONE rank must not leave the others blocked in this leg's collectives.  The first generation
    runs WITHOUT a collective inside try / except, the ranks then agree on an `ok` flag (one MIN all-reduce every rank reaches), and
    only if every rank is fine do the timed passes -- with their all-gathers -- start; else every rank returns the error record."""
    global CONTEXT, PRED, FRAMES, DSTEPS, NTOK
    saved = (CONTEXT, PRED, FRAMES, DSTEPS, NTOK)
    if tiny:
        CONTEXT, PRED, FRAMES, DSTEPS, NTOK = 5, 12, 12, 2, 6
    try:
        return _measure(dev, rank, world, B, steps, warmup, cpu, mask_frames, engine_options, native_lib, tiny, small_batch)
    finally:
        CONTEXT, PRED, FRAMES, DSTEPS, NTOK = saved


def _measure(dev, rank, world, B, steps, warmup, cpu, mask_frames, engine_options, native_lib, tiny, small_batch):
    torch.manual_seed(0)
    over = dict(layers=1, latent_dim=256, pos_embed_max_len=64) if tiny else {}
    args = model_util.default_args(diffusion_steps=DSTEPS, arch="trans_dec", text_encoder_type="bert", context_len=CONTEXT,
                                   pred_len=PRED, mask_frames=mask_frames, guidance_param=7.5, **over)      # DiP.md:181: `--mask_frames`
    extra = dict(_native_lib=native_lib, num_heads=2) if tiny else {}
    mdm, diffusion = model_util.create_model_and_diffusion(args, engine_options=engine_options, **extra)
    state = {k: v.clone() for k, v in mdm.state_dict().items()}
    model = ClassifierFreeSampleModel(mdm).to(dev).eval()
    y = synthetic_y(B, dev, 1000 + rank)
    diffusion.sample_base = rank * B
    GB = B * world
    sampler = AutoRegressiveSampler(SimpleNamespace(pred_len=PRED, context_len=CONTEXT, autoregressive_include_prefix=False),
                                    diffusion.p_sample_loop, FRAMES)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def local_pass():
        return sampler.sample(model, (B, 263, 1, FRAMES), clip_denoised=False, model_kwargs={"y": y})

    def one_pass():
        return mdist.all_gather_samples(local_pass(), GB, world)

    def fence():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    err = None
    try:                                   # stage 1: no collective in here
        out = local_pass()
        sync()
        if not bool(torch.isfinite(out).all()):
            raise FloatingPointError("non-finite DiP sample")
    except Exception as e:                 # noqa: BLE001
        err = f"{type(e).__name__}: {e}"
    if world > 1:
        ok = torch.tensor([0 if err else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            return {"error": err or "the DiP leg failed on another rank"} if rank == 0 else None
    elif err:
        raise RuntimeError(err)

    for _ in range(warmup):
        one_pass()
    eng = mdm.engine()
    if tiny:                 # the emulator dry run reads the per-class record off its one timed pass (no extra generation)
        eng.profile(True)
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one_pass()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert out.shape == (GB, 263, 1, FRAMES) and bool(torch.isfinite(out).all())
    if not tiny:
        eng.profile(True)
        one_pass()
        sync()
    prof = eng.profile_read()
    eng.profile(False)
    if rank != 0:
        return None
    lin = prof["linear"]
    ach = lin["flops"] / (lin["ms"] * 1e-3) / 1e12 if lin["ms"] > 0 else 0.0
    prec = mdm.precision
    peak = 157.3 if prec == "f32" else 2500.0
    nwin = (FRAMES + PRED - 1) // PRED
    line = {"metric": f"motions/sec (DiP: {FRAMES} frames = {nwin} windows x {DSTEPS} steps, CFG, B={B} per GPU)",
            "value": round(GB * steps / dt, 3), "unit": "motions/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": prec, "data": "synthetic",
            "config": {"workload": f"DiP autoregressive text2motion: trans_dec 8 layers d=512, prefix 20 + window 40 "
                                   f"frames, {NTOK}-token DistilBERT memory (cached), {DSTEPS} DDPM steps per window, "
                                   f"CFG 7.5, batch={B} per GPU, random-init weights, mask_frames={mask_frames} with y['mask'] = "
                                   f"ones[B,1,1,{FRAMES}] as sample/generate.py:107 builds it" +
                                   (" (the DiP.md:181 recipe: a frame mask on every forward)" if mask_frames else " (no frame mask reaches the kernels: A/B only)"),
                       "global_batch": GB, "mask_frames": bool(mask_frames),
                       "parallelism": f"dp{world}: batch shards, all_gather of final samples"},
            "roofline": {"bound": "mfma", "kernel": "decoder GEMMs (" + ("gemm_f32_kernel" if prec == "f32" else "gemm_x3s_kernel on operand planes + the (sequence, head) attention blocks selfattn_block_kernel<0|1|2> (in_proj + self-attention; cross q projection + memory attention) -- from 144 row tiles on xattn_block_kernel (the whole cross-attention block in one launch); K / V of the text memory: gemm_f32_kernel<X3>") + ")",
                         "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         "traffic": pmc_traffic_per_launch()[0] if B == 32 else None, "traffic_source": pmc_traffic_per_launch()[1],
                         "launches": lin["launches"],
                         "avg_launch_us": round(lin["ms"] * 1e3 / max(lin["launches"], 1), 2)},
            "kernel_ms": {k: round(v["ms"], 3) for k, v in prof.items()},
            "launches_per_motion_batch": int(sum(v["launches"] for v in prof.values()))}
    if tiny:
        line["data"] = "synthetic (CPU emulator dry run of the launcher / sharding / gather code: NOT a measurement)"
    if world == 1 and small_batch and not tiny:
        line["small_batch"] = measure_small_batch(model, diffusion, mdm, dev, sync, state, cpu=cpu)
    if world == 1 and cpu and not tiny:
        line["cpu_baseline"] = cpu_baseline(state)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="motions per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the per-call latency sub-record (B = 1 / 6)")
    ap.add_argument("--no-mask-frames", action="store_true", help="A/B: a model built without --mask_frames (NULL lengths)")
    ap.add_argument("--no-fused-xattn", action="store_true", help="A/B: the cross-attention block as three launches (round 4's form)")
    ap.add_argument("--xattn", type=int, default=3, help="A/B: MDM_OPT_DEC_FUSED_XATTN (3 by size, 2 per (sequence, head) + GEMM, 1 one kernel, 0 three launches)")
    ap.add_argument("--row-tiles", type=int, default=0, help="A/B: MDM_OPT_SMALL_GEMM_ROW_TILES (0 by size, 1 = 32-row tiles, 2 = 64-row tiles)")
    ap.add_argument("--no-fused-selfattn", action="store_true", help="A/B: in_proj + self-attention as two launches (round 4's form)")
    a = ap.parse_args()
    rank, world, local = mdist.init_from_env("nccl")
    assert world == a.gpus and torch.cuda.is_available()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    line = measure(dev, rank, world, a.batch, a.steps, a.warmup, cpu=not a.no_cpu_baseline, mask_frames=not a.no_mask_frames,
                   small_batch=not a.no_small_batch, engine_options={"dec_fused_xattn": 0 if a.no_fused_xattn else a.xattn, "dec_fused_selfattn": 0 if a.no_fused_selfattn else 1,
                                   **({"small_gemm_row_tiles": a.row_tiles} if a.row_tiles else {})})
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

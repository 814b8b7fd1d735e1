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
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        t = torch.full((B,), DSTEPS - 1 - n % DSTEPS, dtype=torch.long)
        x = orc.ddpm_step(tab, x, dip.dip_cfg_forward(sd, x, t, y, context_len=CONTEXT), t, torch.randn_like(x))
        n += 1
    per_step = (time.perf_counter() - t0) / n
    return {"value": B / (per_step * DSTEPS * 5), "unit": "motions/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle CFG p_sample of one 60-token window: {n} diffusion steps at B={B}, scaled to 5 windows x "
                      f"{DSTEPS} steps per motion ({per_step * 1e3:.0f} ms per batch-step)"}


DIP_PMC_PROFILE = os.path.join("profiles", "r05_dip_pmc.json")


def pmc_traffic_per_launch():
    """Fabric-side bytes per launch of the decoder's GEMM-class kernels (the `linear` profiling class: gemm_x3s_kernel, the
    (sequence, head) attention blocks, xattn_block_kernel) from the committed rocprofv3 PMC passes of THIS command at B = 32
    (tools/gpu_r5_dip_pmc.sh + tools/dip_pmc_to_json.py), call-weighted; quoted only while the kernel sources are the ones the
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


def measure(dev, rank, world, B, steps, warmup, cpu=True, mask_frames=True, engine_options=None):
    """Time `steps` whole 196-frame generations of B motions per rank on `dev`; returns the JSON record (rank 0) or None.
    bench.py embeds this record as its `dip` sub-line so that the driver's BENCH file carries it."""
    torch.manual_seed(0)
    args = model_util.default_args(diffusion_steps=DSTEPS, arch="trans_dec", text_encoder_type="bert", context_len=CONTEXT,
                                   pred_len=PRED, mask_frames=mask_frames, guidance_param=7.5)      # DiP.md:181: `--mask_frames`
    mdm, diffusion = model_util.create_model_and_diffusion(args, engine_options=engine_options)
    state = {k: v.clone() for k, v in mdm.state_dict().items()}
    model = ClassifierFreeSampleModel(mdm).to(dev).eval()
    y = synthetic_y(B, dev, 1000 + rank)
    diffusion.sample_base = rank * B
    GB = B * world
    sampler = AutoRegressiveSampler(SimpleNamespace(pred_len=PRED, context_len=CONTEXT, autoregressive_include_prefix=False),
                                    diffusion.p_sample_loop, FRAMES)

    def one_pass():
        out = sampler.sample(model, (B, 263, 1, FRAMES), clip_denoised=False, model_kwargs={"y": y})
        return mdist.all_gather_samples(out, GB, world)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        one_pass()
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
    eng = mdm.engine()
    eng.profile(True)
    one_pass()
    torch.cuda.synchronize(dev)
    prof = eng.profile_read()
    eng.profile(False)
    if rank != 0:
        return None
    lin = prof["linear"]
    ach = lin["flops"] / (lin["ms"] * 1e-3) / 1e12 if lin["ms"] > 0 else 0.0
    prec = mdm.precision
    peak = 157.3 if prec == "f32" else 2500.0
    line = {"metric": "motions/sec (DiP: 196 frames = 5 windows x 10 steps, CFG, B=32 per GPU)",
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
    if world == 1 and cpu:
        line["cpu_baseline"] = cpu_baseline(state)
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="motions per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
                   engine_options={"dec_fused_xattn": 0 if a.no_fused_xattn else a.xattn, "dec_fused_selfattn": 0 if a.no_fused_selfattn else 1,
                                   **({"small_gemm_row_tiles": a.row_tiles} if a.row_tiles else {})})
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Headline benchmark: motions/sec of the DDPM sampling hot path (BASELINE.json) on N MI355X of one node.

One "step" = one full `p_sample_loop` (x_T -> x_0: 50 diffusion steps, classifier-free guidance 2.5, i.e. 100
denoiser forwards) over one batch of 128 synthetic HumanML3D-shaped motions [128, 263, 1, 196] per GPU, with the
text embedding pre-cached and every input already resident in HBM (configs[1] of BASELINE.json).  N > 1: one
process per GPU (torchrun), each rank samples its own 128-motion shard of a 128*N batch (weak scaling) and the
final samples are all-gathered with RCCL inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Prints ONE JSON line on rank 0.  Besides the contract fields it carries
  roofline      the dominant kernel class (encoder GEMMs) against the MFMA peak it runs on, from hipEvent pairs the
                library records around every launch of one extra, untimed-for-throughput loop (mdm_profile_*),
  kernel_ms     per-kernel-class totals of that loop,
  cpu_baseline  the oracle (CPU restatement of the reference, pinned to it: tests/golden) timed on this box's host
                cores on a bounded sample of the same workload (N=1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import mdm_amd  # noqa: E402,F401
from mdm_amd import dist as mdist  # noqa: E402
from mdm_amd import model_util  # noqa: E402
from mdm_amd.cfg_sampler import ClassifierFreeSampleModel  # noqa: E402

# algorithmic flops of one MDM forward of one sample (S=197, d=512, ff=1024, L=8, J=263)
PEAKS_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0}      # MI355X_MICROARCH.md: fp32 MFMA / dense bf16 MFMA


def algorithmic_flops_per_forward(T, d=512, ff=1024, L=8, J=263):
    S = T + 1
    per_layer = 2 * S * d * 3 * d + 2 * 2 * S * S * d + 2 * S * d * d + 4 * S * d * ff
    return 2 * T * J * d + L * per_layer + 2 * T * d * J + 3 * 2 * d * d


def synthetic_y(B, T, device, seed):
    """model_kwargs['y'] as sample/generate.py:107-132 builds it for a text prompt batch (all frames valid)."""
    g = torch.Generator().manual_seed(seed)
    return {"mask": torch.ones(B, 1, 1, T, dtype=torch.bool, device=device),
            "lengths": torch.full((B,), T, dtype=torch.long, device=device),
            "text_embed": torch.randn(1, B, 512, generator=g).to(device),
            "scale": torch.full((B,), 2.5, device=device)}


def cpu_baseline(state, T, dsteps, budget_s=15.0):
    """The oracle's p_sample_loop (CFG on) on the host cores: a bounded number of diffusion steps of a small batch,
    scaled to whole 50-step motions.  Checker code used as a reported baseline only."""
    from oracle import mdm_oracle as orc
    B = 4
    sd = {k: v.detach().cpu().float() for k, v in state.items()}
    tab = orc.Tables(orc.named_betas("cosine", dsteps))
    g = torch.Generator().manual_seed(0)
    y = {"mask": torch.ones(B, 1, 1, T, dtype=torch.bool), "lengths": torch.full((B,), T),
         "text_embed": torch.randn(1, B, 512, generator=g), "scale": torch.full((B,), 2.5)}
    x = torch.randn(B, 263, 1, T, generator=g)
    pe = orc.positional_table(5000, 512)

    def one_step(i, x):
        t = torch.full((B,), i, dtype=torch.long)
        x0 = orc.predict_x0(lambda a, b, c: orc.cfg_forward(sd, a, b, c, pe=pe), x, t, y)
        return orc.ddpm_step(tab, x, x0, t, torch.randn(x.shape, generator=g))

    with torch.no_grad():
        x = one_step(dsteps - 1, x)                       # warm-up (thread pool, allocator)
        t0 = time.perf_counter()
        x = one_step(dsteps - 2, x)
        per = time.perf_counter() - t0
        n = int(max(2, min(dsteps - 2, budget_s / max(per, 1e-3))))
        t0 = time.perf_counter()
        for k in range(n):
            x = one_step(dsteps - 3 - k, x)
        per = (time.perf_counter() - t0) / n
    return {"value": B / (per * dsteps), "unit": "motions/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle CFG p_sample: {n} of {dsteps} diffusion steps at B={B}, T={T}, scaled to {dsteps}-step "
                      f"motions ({per * 1e3:.0f} ms per batch-step)"}


# Mean ALGORITHMIC HBM bytes of one encoder-GEMM launch at the headline shape (256 sequences x 197 tokens, D=512, FF=1024;
# DESIGN.md section 4): operand planes read once + weights once + output written once (+ residual planes), averaged over
# the four GEMMs of a layer.  in_proj 103+3+352 MB, out_proj 103+1+103+103, linear1 103+2+207, linear2 207+2+103+103.
ALGORITHMIC_GEMM_BYTES_PER_LAUNCH = int((458.9e6 + 310.9e6 + 311.9e6 + 415.2e6) / 4)


def pmc_traffic_per_gemm_launch():
    """HBM-side bytes per encoder-GEMM launch from the committed rocprofv3 PMC passes of THIS command
    (profiles/r01_pmc.json, produced by tools/gpu_prof.sh + tools/pmc_to_json.py: FETCH_SIZE x2 per the gfx950 correction,
    WRITE_SIZE as is; separate --pmc passes).  PMC collection needs rocprofv3 around the process, so the live run can
    only quote the profile of the same build; None if the profile is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc.json")
    if not os.path.isfile(path):
        return None, None
    try:
        with open(path) as f:
            d = json.load(f)
        w = {"gemm_bf16x3<in_proj (LayerNorm folded) -> Q/K/V^T planes>": 1,
             "gemm_bf16x3<out_proj | linear2, LayerNorm residual, planes + row stats>": 2,
             "gemm_bf16x3<linear1 (LayerNorm folded) + GELU -> planes>": 1}
        tot = sum(d[k]["hbm_bytes"] * n for k, n in w.items())
        return int(tot / sum(w.values())), "profiles/r01_pmc.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE of this command)"
    except (KeyError, ValueError, OSError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed p_sample_loop passes")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=128, help="motions per GPU")
    ap.add_argument("--frames", type=int, default=196)
    ap.add_argument("--diffusion-steps", type=int, default=50)
    ap.add_argument("--precision", default="bf16x3", choices=["bf16x3", "f32"],
                    help="arithmetic of the encoder GEMMs (include/mdm_hip.h mdm_set_precision)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    rank, world, local = mdist.init_from_env("nccl")
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    B, T, DS = a.batch, a.frames, a.diffusion_steps
    torch.manual_seed(0)                                   # random-init weights of the named architecture
    args = model_util.default_args(diffusion_steps=DS)
    mdm, diffusion = model_util.create_model_and_diffusion(args, precision=a.precision)
    state = {k: v.clone() for k, v in mdm.state_dict().items()}
    model = ClassifierFreeSampleModel(mdm).to(dev).eval()
    y = synthetic_y(B, T, dev, seed=1000 + rank)
    diffusion.sample_base = rank * B
    shape = (B, 263, 1, T)
    GB = B * world

    def one_pass(seed):
        out = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, seed=seed)
        return mdist.all_gather_samples(out, GB, world)

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for w in range(a.warmup):
        one_pass(w)
    fence()
    t0 = time.perf_counter()
    for k in range(a.steps):
        out = one_pass(100 + k)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert out.shape[0] == GB and bool(torch.isfinite(out).all())

    # ---- per-kernel-class timing of one more pass (rank 0's GPU), hipEvents on the launch stream
    eng = mdm.engine()
    eng.profile(True)
    one_pass(999)
    torch.cuda.synchronize(dev)
    prof = eng.profile_read()
    eng.profile(False)

    if rank == 0:
        traffic, traffic_src = pmc_traffic_per_gemm_launch()
        motions_s = GB * a.steps / dt
        lin = prof["linear"]
        ach = lin["flops"] / (lin["ms"] * 1e-3) / 1e12 if lin["ms"] > 0 else 0.0
        peak = PEAKS_TFLOPS[a.precision]
        x3 = a.precision == "bf16x3"
        fwd = algorithmic_flops_per_forward(T)
        line = {
            "metric": "motions/sec (B=128, T=196, 50-step DDPM)", "value": round(motions_s, 3), "unit": "motions/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16x3 (fp32 operands split into bf16 hi+lo, 3 bf16 MFMA products, fp32 accumulate; everything "
                     "else fp32)" if x3 else "f32", "data": "synthetic",
            "config": {"workload": f"HumanML3D text2motion, {DS}-step p_sample_loop with CFG 2.5 (2 denoiser forwards "
                                   f"per step), batch={B} per GPU, T={T}, 8-layer d=512 trans_enc MDM, random-init "
                                   f"weights, cached text embedding", "global_batch": GB, "diffusion_steps": DS,
                       "parallelism": f"dp{world}: batch shards, no data-path collective, all_gather of final samples"},
            "sample_steps_per_s": round(motions_s * DS, 1),
            "model_tflops": round(motions_s * DS * 2 * fwd / 1e12, 2),
            "roofline": {"bound": "mfma",
                         "kernel": ("gemm_bf16x3_kernel<Linear>" if x3 else "gemm_f32_kernel<RowMajor,RowMajor,Linear>")
                         + " (encoder GEMMs: in_proj, out_proj, linear1, linear2)",
                         "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                         "algorithmic_bytes": ALGORITHMIC_GEMM_BYTES_PER_LAUNCH if x3 else None,
                         "launches": lin["launches"],
                         "avg_launch_us": round(lin["ms"] * 1e3 / max(lin["launches"], 1), 2),
                         "executed_mfma_tflops": round(ach * (3 if x3 else 1), 2),
                         "peak_basis": ("dense bf16 MFMA (v_mfma_f32_32x32x16_bf16) 2.5 PFLOP/s; `achieved` counts the "
                                        "ALGORITHMIC fp32 flops 2MNK, the kernel executes 3x that on the matrix pipe, so "
                                        "frac <= 1/3 by construction" if x3 else
                                        "exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md")},
            "kernel_ms": {k: round(v["ms"], 3) for k, v in prof.items()},
        }
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(state, T, DS)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

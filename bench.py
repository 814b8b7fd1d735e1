"""Headline benchmark: motions/sec of the DDPM sampling hot path (BASELINE.json) on N MI355X of one node.

One "step" = one full `p_sample_loop` (x_T -> x_0: 50 diffusion steps, classifier-free guidance 2.5, i.e. 100
denoiser forwards) over one batch of 128 synthetic HumanML3D-shaped motions [128, 263, 1, 196] per GPU, with the
text embedding pre-cached and every input already resident in HBM (configs[1] of BASELINE.json).  N > 1: one
process per GPU, each rank samples its own 128-motion shard of a 128*N batch (weak scaling) and the final samples
are all-gathered with RCCL inside the timed region.

    python bench.py [--gpus N] [--steps K] [--warmup W]

works from a bare shell for every N: with N > 1 and no torchrun environment the script re-launches itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>`
(one rank per GPU over RCCL); launched by torchrun directly (RANK / WORLD_SIZE set) it just joins the group.

Prints ONE JSON line on rank 0.  Besides the contract fields it carries
  roofline      the dominant kernel class (encoder GEMMs) against the MFMA peak it runs on, from hipEvent pairs the
                library records around every launch of one extra, untimed-for-throughput loop (mdm_profile_*),
  kernel_ms     per-kernel-class totals of that loop,
  f32_mode      the same workload on the exact-fp32 MFMA kernels (one short extra pass; N = 1 only),
  dip           the DiP configuration (BASELINE.json configs[4] per GPU: trans_dec, 5 windows x 10 steps) -- N = 1 only,
  steps1000     BASELINE.json configs[2] (1000-step DDPM, batch 64) -- two timed passes, N = 1 only,
  small_batch   the same loop at batch 1 / 6 / 10 (what sample/generate.py runs by default; B1 = configs[0]'s shape), ms per call,
  trans_dec     the same loop on the reference's trans_dec + DistilBERT text-to-motion denoiser (README.md:254) at the headline batch,
  cpu_baseline  BASELINE.json configs[0]: the whole 50-step CFG loop at batch 1 on this box's host cores, twice -- the
                reference's own p_sample_loop where the upstream tree is mounted, else the oracle port (N = 1, rank 0),
  ranks         what the process group looked like (backend, world size) -- the evidence that RCCL carried the gather.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic flops of one MDM forward of one sample (S=197, d=512, ff=1024, L=8, J=263)
PEAKS_TFLOPS = {"f32": 157.3, "f16x3": 2500.0}      # MI355X_MICROARCH.md: fp32 MFMA / dense fp16 MFMA
PMC_PROFILE = os.path.join("profiles", "r06_pmc.json")


def algorithmic_flops_per_forward(T, d=512, ff=1024, L=8, J=263):
    S = T + 1
    per_layer = 2 * S * d * 3 * d + 2 * 2 * S * S * d + 2 * S * d * d + 4 * S * d * ff
    return 2 * T * J * d + L * per_layer + 2 * T * d * J + 3 * 2 * d * d


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed p_sample_loop passes")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=128, help="motions per GPU")
    ap.add_argument("--frames", type=int, default=196)
    ap.add_argument("--diffusion-steps", type=int, default=50)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "f32"],
                    help="arithmetic of the encoder GEMMs (include/mdm_hip.h mdm_set_precision)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the f32_mode, steps1000 and dip sub-records")
    ap.add_argument("--no-steps1000", action="store_true", help="skip the 1000-step (BASELINE.json configs[2]) sub-record")
    ap.add_argument("--quick", action="store_true", help="the headline line only: no sub-records, no CPU baseline")
    ap.add_argument("--no-small-batch", action="store_true", help="skip the B = 1 / 6 / 10 latency sub-record")
    # test infrastructure (tests/test_round2_cpu.py::test_bench_self_launches_two_ranks_from_a_bare_shell): the same launcher / sharding / gather / JSON code on CPU --
    # gloo ranks, kernels in the CPU emulator, a tiny model.  Never a measurement.
    ap.add_argument("--force-pg", action="store_true",
                    help="N = 1 only: still create the (one-rank) RCCL process group and run the gather through it")
    ap.add_argument("--engine-option", action="append", default=[], metavar="NAME=VALUE",
                    help="A/B runs: a handle option of include/mdm_hip.h mdm_set_option by its Python name, e.g. attn_direct_out=1")
    ap.add_argument("--emulate", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--layers", type=int, default=8, help=argparse.SUPPRESS)
    ap.add_argument("--latent-dim", type=int, default=512, help=argparse.SUPPRESS)
    a = ap.parse_args(argv)
    if a.quick:
        a.no_extras = a.no_cpu_baseline = True
    return a


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(a, argv):
    """`python bench.py --gpus N` from a bare shell: become the torchrun launcher of N ranks of this same command."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    # HSA_ENABLE_IPC_MODE_LEGACY=0 selects dmabuf IPC.  It is this image's documented requirement for multi-process GPU work
    # (the host driver supports dmabuf IPC only; without it RCCL / cross-process tensor sharing fails with
    # `hipIpcGetMemHandle: invalid argument`) and is already exported on the GPU boxes -- kept here (setdefault: never overrides)
    # so that a bare `python bench.py --gpus N` builds the same environment for its ranks.  Not measured by this repository:
    # no N > 1 box has been offered to it.
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def synthetic_y(B, T, device, seed, clip_dim=512):
    """model_kwargs['y'] as sample/generate.py:107-132 builds it for a text prompt batch (all frames valid)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    return {"mask": torch.ones(B, 1, 1, T, dtype=torch.bool, device=device),
            "lengths": torch.full((B,), T, dtype=torch.long, device=device),
            "text_embed": torch.randn(1, B, clip_dim, generator=g).to(device),
            "scale": torch.full((B,), 2.5, device=device)}


def cpu_baseline(state, T, dsteps, runs=2):
    """BASELINE.json configs[0] as written: the 50-step CFG p_sample_loop at batch = 1, T = 196 on this box's host cores,
    the WHOLE loop timed, `runs` times (both values reported; `value` is their mean).  When the upstream tree is mounted
    (MDM_REFERENCE_ROOT / /root/reference -- the build container, never the GPU box) the reference's own
    `SpacedDiffusion.p_sample_loop(ClassifierFreeSampleModel(MDM))` (diffusion/gaussian_diffusion.py:591-658) is what runs
    and `kind` is "reference"; otherwise the oracle port of the same loop (oracle/mdm_oracle.py, pinned to the reference by
    tests/golden).  Checker code used as a reported baseline only -- never the thing measured as `value` of the line."""
    import torch
    from oracle import mdm_oracle as orc
    from oracle import ref_harness
    B = 1
    sd = {k: v.detach().cpu().float() for k, v in state.items()}
    g = torch.Generator().manual_seed(0)
    y = {"mask": torch.ones(B, 1, 1, T, dtype=torch.bool), "lengths": torch.full((B,), T),
         "text_embed": torch.randn(1, B, 512, generator=g), "scale": torch.full((B,), 2.5)}
    shape = (B, 263, 1, T)
    thr = torch.get_num_threads()
    kind, runner = "port", None
    if ref_harness.reference_available():
        try:
            ref = ref_harness.build_reference_model(seed=0)
            ref_harness.load_reference_weights(ref, sd)
            cfg = ref_harness.reference_cfg(ref)
            rdiff = ref_harness.build_reference_diffusion(steps=dsteps)

            def runner():
                return rdiff.p_sample_loop(cfg, shape, clip_denoised=False, model_kwargs={"y": dict(y)}, skip_timesteps=0,
                                           init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False)
            kind = "reference"
        except Exception:   # an incomplete mount: fall back to the port, and say so through `kind`
            kind, runner = "port", None
    if runner is None:
        tab = orc.Tables(orc.named_betas("cosine", dsteps))

        def runner():
            x_T = torch.randn(shape, generator=g)
            noises = [torch.randn(shape, generator=g) for _ in range(dsteps)]
            return orc.sample_loop(sd, tab, shape, y, x_T, noises, cfg=True)
    def warm():
        # a short loop through the same code (thread pool, allocator, oneDNN primitive caches)
        if kind == "port":
            tabw = orc.Tables(orc.named_betas("cosine", 2))
            orc.sample_loop(sd, tabw, shape, y, torch.randn(shape, generator=g), [torch.randn(shape, generator=g)] * 2, cfg=True)
        else:
            ref_harness.build_reference_diffusion(steps=2).p_sample_loop(cfg, shape, clip_denoised=False,
                                                                         model_kwargs={"y": dict(y)})

    # torch's default intra-op thread count is every hardware thread of the box; at batch 1 (197-row GEMMs) that
    # over-subscription is SLOWER than a few cores (128 threads: ~17 s per motion on the GPU box, the 8-vCPU build container:
    # 2.3 s).  So the thread count is swept and the best setting is the value.
    all_thr = torch.get_num_threads()
    # sweep {8, 16, 32, 64} (those the box has) once each, then `runs` - 1 more passes at the best one; all threads once
    sweep = sorted({n for n in (8, 16, 32, 64) if n < all_thr} | ({all_thr} if all_thr <= 64 else set()))
    settings = [(n, 1) for n in sweep]      # (all of a 256-thread box's threads: 0.076 motions/s in round 3, 10x slower than 16)
    results = []
    with torch.no_grad():
        for nthr, n in settings:
            torch.set_num_threads(nthr)
            warm()
            ts = []
            for _ in range(n):
                t0 = time.perf_counter()
                out = runner()
                ts.append(time.perf_counter() - t0)
                assert bool(torch.isfinite(out).all())
            results.append((nthr, ts))
        best_thr = min(results, key=lambda r: sum(r[1]) / len(r[1]))[0]
        torch.set_num_threads(best_thr)
        for _ in range(max(runs - 1, 0)):        # the winner again: its value is a mean of `runs` whole loops
            t0 = time.perf_counter()
            out = runner()
            [r for r in results if r[0] == best_thr][0][1].append(time.perf_counter() - t0)
        torch.set_num_threads(all_thr)
    best_thr, best_ts = min(results, key=lambda r: sum(r[1]) / len(r[1]))
    mean = sum(best_ts) / len(best_ts)
    what = ("the reference's own SpacedDiffusion.p_sample_loop(ClassifierFreeSampleModel(MDM)) (diffusion/gaussian_diffusion.py:591-658)"
            if kind == "reference" else "oracle/mdm_oracle.sample_loop (torch-CPU restatement of the reference, pinned by tests/golden)")
    return {"value": round(B / mean, 4), "unit": "motions/s", "cores": best_thr, "kind": kind,
            "runs_motions_per_s": {str(nthr): [round(B / t, 4) for t in ts] for nthr, ts in results},
            "runs_s": {str(nthr): [round(t, 3) for t in ts] for nthr, ts in results},
            "sample_steps_per_s": round(B * dsteps / mean, 2),
            "sample": f"BASELINE.json configs[0]: {what}, CFG 2.5, batch={B}, T={T}, all {dsteps} diffusion steps, whole loop "
                      f"timed after a 2-step warm-up loop at {' / '.join(str(r[0]) for r in results)} intra-op threads "
                      f"({os.cpu_count()} logical host CPUs; keys of runs_*); value = the faster setting ({best_thr} threads, "
                      f"mean of {len(best_ts)}); a reported baseline, not a target"}


# Mean ALGORITHMIC HBM bytes of one encoder-GEMM launch at the headline shape (256 sequences x 197 tokens, D=512, FF=1024;
# DESIGN.md section 4): operand planes read once + weights once + output written once (+ residual planes), averaged over
# the four GEMMs of a layer.  in_proj 103+3+352 MB, out_proj 103+1+103+103, linear1 103+2+207, linear2 207+2+103+103.
ALGORITHMIC_GEMM_BYTES_PER_LAUNCH = int((458.9e6 + 310.9e6 + 311.9e6 + 415.2e6) / 4)


def measure_steps1000(mdm, model, dev, sync, T, layers, latent_dim, B=64, dsteps=1000, passes=2):
    """BASELINE.json configs[2]: the 1000-step DDPM p_sample_loop at batch = 64 on this GPU (`step-fusion stress`:
    diffusion/gaussian_diffusion.py:708 runs its loop body 1000x): one short warm-up loop (8-step schedule, same kernels
    and shapes) + `passes` timed passes of the whole 1000-step loop, same model / weights / arithmetic as the headline."""
    import torch
    from mdm_amd import model_util
    diff = model_util.create_gaussian_diffusion(model_util.default_args(diffusion_steps=dsteps, layers=layers,
                                                                        latent_dim=latent_dim))
    warm = model_util.create_gaussian_diffusion(model_util.default_args(diffusion_steps=8, layers=layers,
                                                                        latent_dim=latent_dim))
    y = synthetic_y(B, T, dev, seed=2000)
    shape = (B, 263, 1, T)
    warm.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, seed=1)
    sync()
    dts = []
    for k in range(passes):
        t0 = time.perf_counter()
        out = diff.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, seed=2 + k)
        sync()
        dts.append(time.perf_counter() - t0)
        assert out.shape[0] == B and bool(torch.isfinite(out).all())
    dt = sum(dts) / len(dts)
    fwd = algorithmic_flops_per_forward(T)
    return {"value": round(B / dt, 3), "unit": "motions/s", "sample_steps_per_s": round(B * dsteps / dt, 1),
            "ms_per_step": round(dt * 1e3, 2), "steps": passes, "passes_ms": [round(t * 1e3, 1) for t in dts],
            "model_tflops": round(B * dsteps / dt * 2 * fwd / 1e12, 2),
            "config": {"workload": f"BASELINE.json configs[2]: HumanML3D {dsteps}-step DDPM p_sample_loop with CFG 2.5, "
                                   f"batch={B}, T={T}, same model and f16x3 arithmetic as the headline; mean of {passes} timed passes",
                       "global_batch": B, "diffusion_steps": dsteps}}


def measure_trans_dec(dev, sync, T, dsteps, B=128, ntok=24, passes=2):
    """The reference's newest text-to-motion checkpoint shape (README.md:254 `humanml_trans_dec_512_bert-50steps`, trained by
    README.md:451: `--arch trans_dec --text_encoder_type bert --mask_frames`, NO prefix): sample/generate.py's plain 50-step CFG
    p_sample_loop over T frames at the headline batch -- the encoder's workload with a cross-attention block over the DistilBERT
    tokens in every layer (model/mdm.py:85-93, :255-270).  From 81 sequences on the decoder's GEMMs run on gemm_x3.h's
    sequence-sized tiles like the encoder's (csrc/decoder.h dec_sequence_tiles); one warm-up loop + `passes` timed loops."""
    import torch
    from mdm_amd import model_util
    from mdm_amd.cfg_sampler import ClassifierFreeSampleModel
    torch.manual_seed(0)
    args = model_util.default_args(diffusion_steps=dsteps, arch="trans_dec", text_encoder_type="bert", mask_frames=True)
    mdm, diff = model_util.create_model_and_diffusion(args)
    model = ClassifierFreeSampleModel(mdm).to(dev).eval()
    g = torch.Generator().manual_seed(5000)
    tl = torch.randint(6, ntok + 1, (B,), generator=g)
    tl[0] = ntok
    y = {"mask": torch.ones(B, 1, 1, T, dtype=torch.bool, device=dev), "lengths": torch.full((B,), T, dtype=torch.long, device=dev),
         "text_embed": (torch.randn(ntok, B, 768, generator=g).to(dev), (torch.arange(ntok)[None, :] >= tl[:, None]).to(dev)),
         "scale": torch.full((B,), 2.5, device=dev)}
    shape = (B, 263, 1, T)
    diff.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, seed=1)
    sync()
    dts = []
    for k in range(passes):
        t0 = time.perf_counter()
        out = diff.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, seed=2 + k)
        sync()
        dts.append(time.perf_counter() - t0)
        assert out.shape[0] == B and bool(torch.isfinite(out).all())
    dt = sum(dts) / len(dts)
    return {"value": round(B / dt, 3), "unit": "motions/s", "sample_steps_per_s": round(B * dsteps / dt, 1),
            "ms_per_step": round(dt * 1e3, 2), "steps": passes,
            "config": {"workload": f"HumanML3D text2motion on the trans_dec + DistilBERT denoiser (README.md:254, :451), {dsteps}-step "
                                   f"p_sample_loop with CFG 2.5, batch={B}, T={T}, {ntok} text tokens, mask_frames, f16x3; mean of "
                                   f"{passes} timed loops", "global_batch": B, "diffusion_steps": dsteps}}


def measure_small_batch(model, dev, sync, T, layers, latent_dim, dsteps, batches=(1, 6, 10), passes=3):
    """The latency regime the reference's own callers run (sample/generate.py:76,98: `--num_samples` = the batch, default 6;
    README.md:13 quotes per-call latency): the SAME 50-step CFG p_sample_loop at batch 1 (BASELINE.json configs[0]'s shape, the
    one `cpu_baseline` times on the host), 6 and 10 -- milliseconds per call, mean of `passes` calls after one warm-up call.
    Up to 80 sequences the encoder GEMMs run on csrc/gemm_x3s.h's 32 / 64-row tiles."""
    import torch
    from mdm_amd import model_util
    diff = model_util.create_gaussian_diffusion(model_util.default_args(diffusion_steps=dsteps, layers=layers, latent_dim=latent_dim))
    out = {}
    for B in batches:
        y = synthetic_y(B, T, dev, seed=3000 + B)
        shape = (B, 263, 1, T)
        diff.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, seed=1)
        sync()
        diff.check_finite = False          # (the seam's finite check syncs per call: asserted behind the clock instead)
        try:
            t0 = time.perf_counter()
            for k in range(passes):
                x = diff.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": y}, seed=2 + k)
            sync()
            dt = (time.perf_counter() - t0) / passes
        finally:
            diff.check_finite = True
        assert bool(torch.isfinite(x).all())
        out[f"B{B}"] = {"ms_per_call": round(dt * 1e3, 2), "motions_per_s": round(B / dt, 2)}
    out["config"] = {"workload": f"HumanML3D text2motion, {dsteps}-step p_sample_loop with CFG 2.5, T={T}, batch = 1 / 6 / 10: one "
                                 f"call = one loop (B1 is BASELINE.json configs[0]'s shape, what cpu_baseline times on the host)",
                     "passes": passes, "round3_ms_per_call": {"B1": 78.5, "B10": 82.7}}
    return out


def csrc_sha256():
    """Identity of the kernel sources the library was built from (what a committed PMC profile is tied to)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "motion-diffusion-model_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".h", ".hip")))
    for p in files + [os.path.join(ROOT, "include", "mdm_hip.h")]:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    # ... AND of how it was built: round 3 changed the binary with a flag (-fno-slp-vectorize) and no source edit
    import __graft_entry__ as ge
    h.update(" ".join(ge.HIP_FLAGS).encode())
    return h.hexdigest()


def lib_sha256():
    """sha256 of the product library itself (the binary the numbers of this line were taken on)."""
    from mdm_amd import _native
    h = hashlib.sha256()
    with open(_native.LIB_PATH, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def pmc_traffic_per_gemm_launch():
    """HBM-side bytes per encoder-GEMM launch from the committed rocprofv3 PMC passes of THIS command (tools/gpu_prof.sh
    + tools/pmc_to_json.py: FETCH_SIZE x2 per the gfx950 correction, WRITE_SIZE as is; separate --pmc passes).  PMC
    collection needs rocprofv3 around the process, so the live run can only quote the profile of the same build: the
    profile records the sha256 of the kernel sources it was taken on, and a profile of different sources yields null."""
    path = os.path.join(ROOT, PMC_PROFILE)
    if not os.path.isfile(path):
        return None, False, f"{PMC_PROFILE} absent"
    try:
        with open(path) as f:
            d = json.load(f)
        per_layer = {"in_proj": 1, "out_proj|linear2": 2, "linear1": 1}
        tot = sum(d["gemm"][k]["hbm_bytes"] * n for k, n in per_layer.items())
        val = int(tot / sum(per_layer.values()))
        if d.get("csrc_sha256") != csrc_sha256():
            # the kernels (or their build flags) changed after the PMC passes were taken: NOT quoted (re-run tools/gpu_prof.sh pmc)
            return None, True, (f"{PMC_PROFILE}: taken on OTHER kernel sources / build flags (csrc_sha256 differs): "
                                f"{val} bytes/launch there, not quoted here")
        return val, False, f"{PMC_PROFILE} (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE of this command)"
    except (KeyError, ValueError, OSError) as e:
        return None, False, f"{PMC_PROFILE} unreadable ({type(e).__name__})"


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    a = parse_args(argv)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a, argv))

    import torch
    import torch.distributed as dist

    import mdm_amd  # noqa: F401
    from mdm_amd import dist as mdist
    from mdm_amd import model_util
    from mdm_amd.cfg_sampler import ClassifierFreeSampleModel

    backend = "gloo" if a.emulate else "nccl"
    rank, world, local = mdist.init_from_env(backend, force=a.force_pg and a.gpus == 1)
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={world}")
    native_lib = None
    if a.emulate:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        from emu_lib import emu
        native_lib = emu()
        dev = torch.device("cpu")
        torch.set_num_threads(1)
    else:
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    B, T, DS = a.batch, a.frames, a.diffusion_steps
    torch.manual_seed(0)                                   # random-init weights of the named architecture
    args = model_util.default_args(diffusion_steps=DS, layers=a.layers, latent_dim=a.latent_dim,
                                   **({"pos_embed_max_len": 64} if a.emulate else {}))    # (emulator dry runs: a short timestep table)
    eopts = {kv.split("=", 1)[0]: int(kv.split("=", 1)[1]) for kv in a.engine_option}
    mdm, diffusion = model_util.create_model_and_diffusion(args, precision=a.precision, _native_lib=native_lib,
                                                           num_heads=a.latent_dim // 128, engine_options=eopts)
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
        sync()
        if dist.is_initialized():
            dist.barrier()
        sync()

    rank_ms = {}     # label -> per-rank [loop milliseconds] of the last timed() call (a straggler shows as max >> min)

    def timed(passes, warmup, seed0, label="headline"):
        for w in range(warmup):
            one_pass(w)
        fence()
        # the sampler seam's finite check (one reduction + host sync per loop, gaussian_diffusion.py _check_finite) is moved
        # OUT of the timed region: switched off here, and the same check is asserted on the last sample behind the clock
        # (since round 4; `finite_check_in_timed_region: false` in the line says so -- rounds 1-3 timed it inside: ~0.1 ms per loop)
        diffusion.check_finite = False
        try:
            t0 = time.perf_counter()
            out = None
            t_own = 0.0
            for k in range(passes):
                out = one_pass(seed0 + k)
            sync()
            t_own = time.perf_counter() - t0           # this rank's own loops + its part of the gathers, before the rendezvous
            fence()
            dt = time.perf_counter() - t0
        finally:
            diffusion.check_finite = True
        per_rank = [t_own * 1e3 / passes]
        if dist.is_initialized():
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            mine = torch.tensor([t_own * 1e3 / passes], dtype=torch.float64, device=dev)
            allr = torch.empty(world, dtype=torch.float64, device=dev)
            dist.all_gather_into_tensor(allr, mine)
            per_rank = [float(v) for v in allr.tolist()]
        rank_ms[label] = per_rank
        assert out.shape[0] == GB and bool(torch.isfinite(out).all())
        rank_ms["gathered_shape"] = list(out.shape)
        return dt

    dt = timed(a.steps, a.warmup, 100)
    rank_ms["headline"] = list(rank_ms["headline"])
    headline_shape = list(rank_ms["gathered_shape"])

    def flush_c_stdio():
        # RCCL printf()s a version banner into C stdio's buffer at communicator creation; flushed only at exit it would land
        # BEHIND the JSON line.  Push it out now so that the JSON line is the last line of stdout.
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass

    # ---- per-kernel-class timing of one more pass (rank 0's GPU), hipEvents on the launch stream
    eng = mdm.engine()
    eng.profile(True)
    one_pass(999)
    sync()
    prof = eng.profile_read()
    eng.profile(False)

    extras = world == 1 and not a.no_extras and not a.emulate
    f32_mode = dip = None
    if extras and a.precision != "f32":
        # the same workload on the exact-fp32 MFMA kernels (the on-device parity reference): one warm-up + one timed pass
        mdm.precision = "f32"
        f32_dt = timed(2, 1, 500, label="f32_mode") / 2
        mdm.precision = a.precision
        fwd = algorithmic_flops_per_forward(T)
        f32_tf = GB / f32_dt * DS * 2 * fwd / 1e12
        f32_mode = {"value": round(GB / f32_dt, 3), "unit": "motions/s", "ms_per_step": round(f32_dt * 1e3, 3), "steps": 2,
                    "model_tflops": round(f32_tf, 2), "frac_of_157.3TF": round(f32_tf / 157.3, 4),
                    "dtype": "f32 (v_mfma_f32_32x32x2_f32 everywhere)"}
    steps1000 = None
    if extras and a.precision != "f32" and not a.no_steps1000:
        steps1000 = measure_steps1000(mdm, model, dev, sync, T, a.layers, a.latent_dim)
    small_batch = None
    if extras and a.precision != "f32" and not a.no_small_batch:
        small_batch = measure_small_batch(model, dev, sync, T, a.layers, a.latent_dim, DS)
    trans_dec = None
    if extras and a.precision != "f32" and not a.no_small_batch and a.layers == 8 and a.latent_dim == 512:
        trans_dec = measure_trans_dec(dev, sync, T, DS)
    if extras or (world > 1 and not a.no_extras):
        # BASELINE.json configs[4] (DiP, 256 motions over 8 GPUs = 32 per GPU) at EVERY world size since round 5: each rank
        # generates its 32 motions (Philox streams by global sample index), the final all_gather is inside the timed region.
        # N > 1: this leg has only ever run as one rank on real hardware (no multi-GPU box has been offered); a
        # failure in it must not cost the headline line the driver's scaling record is made of, nor leave the other ranks blocked in
        # its collectives: bench_dip.measure runs its first generation without a collective, the ranks agree on an `ok` flag, and a
        # failure anywhere comes back as {"error": ...} IN the line.  `--emulate` (the CPU dry run of this launcher at
        # the real world size, tests/test_round2_cpu.py) runs the same leg on a tiny model in the emulator.
        import bench_dip
        if a.emulate:
            dip = bench_dip.measure(dev, rank=rank, world=world, B=1, steps=1, warmup=0, cpu=False, native_lib=native_lib, tiny=True)
        else:
            dip = bench_dip.measure(dev, rank=rank, world=world, B=32, steps=3, warmup=1, cpu=False, small_batch=(world == 1))

    if rank == 0:
        traffic, traffic_stale, traffic_src = pmc_traffic_per_gemm_launch()
        motions_s = GB * a.steps / dt
        lin = prof["linear"]
        ach = lin["flops"] / (lin["ms"] * 1e-3) / 1e12 if lin["ms"] > 0 else 0.0
        peak = PEAKS_TFLOPS[a.precision]
        x3 = a.precision == "f16x3"
        fwd = algorithmic_flops_per_forward(T)
        line = {
            "metric": "motions/sec (B=128, T=196, 50-step DDPM)", "value": round(motions_s, 3), "unit": "motions/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16x3 (fp32 operands split into fp16 hi+lo, 3 fp16 MFMA products, fp32 accumulate; everything "
                     "else fp32)" if x3 else "f32", "data": "synthetic",
            "config": {"workload": f"HumanML3D text2motion, {DS}-step p_sample_loop with CFG 2.5 (2 denoiser forwards "
                                   f"per step), batch={B} per GPU, T={T}, {a.layers}-layer d={a.latent_dim} trans_enc MDM, "
                                   f"random-init weights, cached text embedding", "global_batch": GB, "diffusion_steps": DS,
                       "parallelism": f"dp{world}: batch shards, no data-path collective, all_gather of final samples"},
            "ranks": {"world_size": dist.get_world_size() if dist.is_initialized() else 1,
                      "backend": (dist.get_backend() + (" (RCCL)" if dist.get_backend() == "nccl" else ""))
                      if dist.is_initialized() else "none (single process)",
                      "launcher": "torch.distributed.run" if "TORCHELASTIC_RUN_ID" in os.environ else "direct",
                      # each rank's own time per loop (its kernels + its side of the all-gather), before the closing barrier:
                      # the line's `value` uses the MAX over ranks of the barrier-to-barrier time; a straggler shows here
                      "loop_ms_per_rank": [round(v, 3) for v in rank_ms["headline"]],
                      "loop_ms_min": round(min(rank_ms["headline"]), 3), "loop_ms_max": round(max(rank_ms["headline"]), 3),
                      "gathered_shape": headline_shape},
            "build": {"csrc_sha256": csrc_sha256(), "lib_sha256": lib_sha256(),
                      "info": eng.lib.lib.mdm_build_info().decode() if hasattr(eng.lib.lib, "mdm_build_info") else None},
            "engine_options": eopts or None,         # non-default handle options of this run (A/B lines only)
            "finite_check_in_timed_region": False,   # asserted on the last sample behind the clock (rounds 1-3: inside, ~0.1 ms per loop)
            "sample_steps_per_s": round(motions_s * DS, 1),
            "model_tflops": round(motions_s * DS * 2 * fwd / 1e12, 2),
            "roofline": {"bound": "mfma",
                         "kernel": ("gemm_x3_kernel" if x3 else "gemm_f32_kernel<RowMajor,RowMajor,Linear>")
                         + " (encoder GEMMs: in_proj, out_proj, linear1, linear2)",
                         "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                         "traffic": traffic, "traffic_stale": traffic_stale, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "algorithmic_bytes": ALGORITHMIC_GEMM_BYTES_PER_LAUNCH if x3 else None,
                         "launches": lin["launches"],
                         "avg_launch_us": round(lin["ms"] * 1e3 / max(lin["launches"], 1), 2),
                         "executed_mfma_tflops": round(ach * (3 if x3 else 1), 2),
                         "peak_basis": ("dense fp16 MFMA (v_mfma_f32_32x32x16_f16) 2.5 PFLOP/s; `achieved` counts the "
                                        "ALGORITHMIC fp32 flops 2MNK, the kernel executes 3x that on the matrix pipe, so "
                                        "frac <= 1/3 by construction" if x3 else
                                        "exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), MI355X_MICROARCH.md")},
            "kernel_ms": {k: round(v["ms"], 3) for k, v in prof.items()},
        }
        if f32_mode is not None:
            line["f32_mode"] = f32_mode
        if steps1000 is not None:
            line["steps1000"] = steps1000
        if small_batch is not None:
            line["small_batch"] = small_batch
        if trans_dec is not None:
            line["trans_dec"] = trans_dec
        if dip is not None:
            line["dip"] = dip
        if world == 1 and not a.no_cpu_baseline and not a.emulate:
            line["cpu_baseline"] = cpu_baseline(state, T, DS)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()

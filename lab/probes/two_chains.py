"""Round 5 experiment: does the headline batch run faster as TWO independent half-batch chains that share the chip?

Every sample is an independent Markov chain, so B = 128 can run as two 50-step loops of 64 motions on two streams with no join before
the end.  Each chain's persistent GEMM launches are limited to half the CUs (probe library: MDM_X3_GRID_DIV=2) and the library's
one-chain-per-device ordering is lifted (MDM_CHAIN_FREE=1), so the two chains drift against each other: one chain's epilogue store
bursts and HBM-bound attention kernels fall into the other's k-loops (profiles/r05h_dephase.md read again: a phase spread between CUs
costs nothing up to 15 us -- the smoother output stream returns what the late start costs).

Run on the MI355X with MDM_HIP_LIB=<probe library>.  Prints: ms per B = 128 batch as one chain (full grid), as two sequential half
batches (full grid), as two concurrent half batches (half grid each), and whether the concurrent results equal the sequential ones bit
for bit (the co-residency check).
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import mdm_amd  # noqa: E402,F401
from mdm_amd import model_util  # noqa: E402
from mdm_amd.cfg_sampler import ClassifierFreeSampleModel  # noqa: E402


def build(dev, seed):
    torch.manual_seed(seed)
    args = model_util.default_args(diffusion_steps=50, layers=8, latent_dim=512)
    mdm, diffusion = model_util.create_model_and_diffusion(args, precision="f16x3", num_heads=4)
    model = ClassifierFreeSampleModel(mdm).to(dev).eval()
    diffusion.check_finite = False
    return model, diffusion


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    T = 196
    models = [build(dev, 0), build(dev, 0)]          # the same weights twice: two engines, two handles, two workspaces
    ys = [bench.synthetic_y(64, T, dev, seed=1000 + i) for i in range(2)]
    y_full = {k: (torch.cat([ys[0][k], ys[1][k]], dim=1 if k == "text_embed" else 0)) for k in ys[0]}
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    res = {}

    def loop(i, B, y, seed, base):
        model, diffusion = models[i]
        diffusion.sample_base = base
        return diffusion.p_sample_loop(model, (B, 263, 1, T), clip_denoised=False, model_kwargs={"y": y}, seed=seed)

    def timeit(fn, n):
        fn(0)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        out = None
        for k in range(n):
            out = fn(100 + k)
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) * 1e3 / n, out

    mode = os.environ.get("TWO_CHAINS_MODE", "all")
    if mode in ("all", "one"):
        res["one_chain_ms"], out_full = timeit(lambda s: loop(0, 128, y_full, s, 0), passes)
    if mode in ("all", "seq"):
        def seq(s):
            return [loop(0, 64, ys[0], s, 0), loop(0, 64, ys[1], s, 64)]
        res["two_sequential_ms"], out_seq = timeit(seq, passes)
    if mode in ("all", "par", "seq"):
        def par(s):
            outs = []
            for i in range(2):
                with torch.cuda.stream(streams[i]):
                    outs.append(loop(i, 64, ys[i], s, 64 * i))
            return outs
        for st in streams:
            st.wait_stream(torch.cuda.current_stream(dev))
        res["two_concurrent_ms"], out_par = timeit(par, passes)
        if mode != "par":
            res["concurrent_equals_sequential"] = bool(all(torch.equal(a, b) for a, b in zip(out_seq, out_par)))
            res["max_abs_diff"] = float(max((a - b).abs().max() for a, b in zip(out_seq, out_par)))
        res["finite"] = bool(all(torch.isfinite(o).all() for o in out_par))
    res["env"] = {k: os.environ.get(k) for k in ("MDM_HIP_LIB", "MDM_X3_GRID_DIV", "MDM_CHAIN_FREE")}
    res["motions_per_s"] = {k[:-3]: round(128e3 / v, 1) for k, v in res.items() if k.endswith("_ms")}
    print(json.dumps(res))


if __name__ == "__main__":
    main()

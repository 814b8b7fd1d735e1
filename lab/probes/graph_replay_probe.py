"""Round 6: does replaying a sampling loop from a hipGraph shorten the DEVICE-side gaps between its dependent kernels?
(`tools/rocpd_gaps.py` on the DiP chain: mean 2.66 us between a kernel's end and its successor's start, 12.7 % of the chain;
the guide's price list says eager == graph for a dependent boundary.)  Eager vs graph replay, same inputs, same box:
  * DiP window call (B = 32 / 1, 10 steps, CFG): 520 launches per call;
  * encoder 50-step loop at B = 1 / 6 (2,250 launches per loop).
Prints one JSON line per case: ms eager, ms graph replay, bit-equality of the two results."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import make_pair, synth_dip_state_dict, synth_dip_y, synth_state_dict, synth_y, to_dev  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, out


def case(name, model, diffusion, shape, y, n):
    diffusion.check_finite = False
    x = torch.randn(shape, device=DEV)
    eager_fn = lambda: diffusion.p_sample_loop(model, shape, noise=x, clip_denoised=False, model_kwargs={"y": dict(y)}, seed=7)   # noqa: E731
    ms_e, want = timeit(eager_fn, n)
    want = want.clone()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = eager_fn()
    ms_g, _ = timeit(lambda: g.replay() or out, n)
    torch.cuda.synchronize()
    print(json.dumps({"case": name, "ms_eager": round(ms_e, 3), "ms_graph_replay": round(ms_g, 3), "ratio": round(ms_g / ms_e, 4),
                      "bit_identical": bool(torch.equal(out, want))}), flush=True)


def main():
    sd = synth_dip_state_dict(seed=0)
    model, diffusion = make_pair(sd, 10, DEV, guided=True, context_len=20, pred_len=40, mask_frames=True)
    for B in (32, 1):
        y = to_dev(synth_dip_y(B, 40, 20, seed=3, text_lengths=[24] + [9 + (i % 12) for i in range(B - 1)]), DEV)
        case(f"DiP window call B={B}", model, diffusion, (B, 263, 1, 40), y, 20)
    sd = synth_state_dict(seed=0)
    model, diffusion = make_pair(sd, 50, DEV, guided=True)
    for B in (1, 6, 128):
        y = to_dev(synth_y(B, 196, seed=3), DEV)
        case(f"encoder 50-step loop B={B}", model, diffusion, (B, 263, 1, 196), y, 5 if B < 128 else 2)


if __name__ == "__main__":
    main()

import sys, torch
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from helpers import make_pair, synth_state_dict, synth_y, to_dev
DEV = "cuda:0"
model, diffusion = make_pair(synth_state_dict(seed=0), 2, DEV, guided=True)
diffusion.check_finite = False
ys = {B: to_dev(synth_y(B, 64, seed=9), DEV) for B in (2, 44)}
def loop(B, x):
    return diffusion.p_sample_loop(model, (B, 263, 1, 64), noise=x, clip_denoised=False, model_kwargs={"y": dict(ys[B])}, seed=7)
xs = {B: torch.randn(B, 263, 1, 64, device=DEV) for B in (2, 44)}
loop(2, xs[2]); torch.cuda.synchronize()
model.model.lengths_from_mask(ys[44], 64)
eng = model.model.engine()
print("options", {k: eng.get_option(k) for k in ("small_gemm_max_seqs", "small_gemm_row_tiles")}, flush=True)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        print("capturing?", torch.cuda.is_current_stream_capturing(), "stream", torch.cuda.current_stream().cuda_stream, "engine stream", eng.stream(), flush=True)
        out = loop(44, xs[44])
    print("no exception", flush=True)
except Exception as e:
    print("EXC", type(e).__name__, str(e)[:500], "| context:", repr(e.__context__)[:500], flush=True)

"""tools/repro_foreign_stream.py's SDPA case on a PROBE build (MDM_HIP_LIB = a library built from csrc + tools/dip_groups_probes.patch
with -DMDM_PROBES and the in-kernel checks): prints what the checks recorded (common.h g_ord_idx / g_ord_va / g_ord_vb).
    MDM_HIP_LIB=$PWD/build/libmdm_hip_probe_PC.so python tools/repro_foreign_sdpa_probe.py [reps]"""
import ctypes as C, os, struct, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from helpers import make_pair, synth_dip_state_dict, synth_dip_y, to_dev
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
sdd = synth_dip_state_dict(seed=0)
B = 8
model, diffusion = make_pair(sdd, 10, DEV, guided=True, context_len=20, pred_len=40, precision="f16x3")
y = to_dev(synth_dip_y(B, 40, 20, seed=2, text_lengths=[4, 11, 25, 8, 1, 17, 9, 30]), DEV)
run = lambda: diffusion.p_sample_loop(model, (B, 263, 1, 40), clip_denoised=False, model_kwargs={"y": y}, seed=7).cpu()
ref = run()
side = torch.cuda.Stream()
q = torch.randn(16, 8, 256, 64, device=DEV, dtype=torch.float16)
fails = 0
for rep in range(reps):
    with torch.cuda.stream(side):
        for i in range(400):
            q = torch.nn.functional.scaled_dot_product_attention(q, q, q)
    fails += int(not torch.equal(run(), ref))
    torch.cuda.synchronize()
print("one chain beside an SDPA stream: differing window loops", fails, "of", reps)
lib = C.CDLL(os.environ["MDM_HIP_LIB"])
lib.mdm_debug_get.argtypes = [C.c_int, C.POINTER(C.c_double)]
d = C.c_double(0.0)
lib.mdm_debug_get(200, C.byref(d)); n = int(d.value)
print("recorded by the in-kernel checks:", n)
f = lambda u: struct.unpack("<f", struct.pack("<I", u))[0]
for k in range(min(n, 256)):
    lib.mdm_debug_get(201 + k, C.byref(d)); idx = int(d.value)
    lib.mdm_debug_get(1000 + k, C.byref(d)); va = int(d.value)
    lib.mdm_debug_get(1256 + k, C.byref(d)); vb = int(d.value)
    kind = "patch read-back" if idx >> 24 else "register vs re-read"
    print(f"  {kind}: lane {idx & 255} field {(idx >> 8) & 255} wave {(idx >> 16) & 255}: {va:08x} ({f(va):.6g}) vs {vb:08x} ({f(vb):.6g})")

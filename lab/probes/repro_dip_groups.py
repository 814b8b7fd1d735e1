"""Reproducer (open issue, profiles/r02e_dip.md): the DiP window loop run as G concurrent sample groups (probe build only,
MDM_DIP_GROUPS=G) intermittently differs from the one-group result in the f16x3 mode.  Counts the differing window loops out of
80 and prints where each first difference appears (step, sample, frames, features).
    MDM_HIP_LIB=$PWD/motion-diffusion-model_amd/csrc/libmdm_hip_probe.so python tools/repro_dip_groups.py [G] [f16x3|f32]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from helpers import make_pair, synth_dip_state_dict, synth_dip_y, to_dev
DEV = "cuda:0"
sdd = synth_dip_state_dict(seed=0)
B = 8
model, diffusion = make_pair(sdd, 10, DEV, guided=True, context_len=20, pred_len=40, precision=(sys.argv[2] if len(sys.argv) > 2 else "f16x3"))
y = to_dev(synth_dip_y(B, 40, 20, seed=2, text_lengths=[4, 11, 25, 8, 1, 17, 9, 30]), DEV)
run = lambda: [d.cpu() for d in diffusion.p_sample_loop(model, (B, 263, 1, 40), clip_denoised=False, model_kwargs={"y": y}, seed=7,
                                                         dump_steps=list(range(10)))]
os.environ["MDM_DIP_GROUPS"] = "1"
ref = run()
os.environ["MDM_DIP_GROUPS"] = sys.argv[1] if len(sys.argv) > 1 else "2"
for rep in range(80):
    got = run()
    bad = [k for k in range(10) if not torch.equal(got[k], ref[k])]
    if not bad:
        continue
    k = bad[0]
    d = (got[k] - ref[k]).abs()[:, :, 0, :]          # [B, 263, 40]
    for b in range(B):
        if float(d[b].max()) > 0:
            fr = (d[b].max(dim=0).values > 0).nonzero().flatten().tolist()
            ft = (d[b].max(dim=1).values > 0).nonzero().flatten().tolist()
            print(f"  rep {rep} step {k} sample {b}: max {float(d[b].max()):.3e}, frames {fr[:6]}..{fr[-1]} ({len(fr)}), features {ft[:4]}..{ft[-1]} ({len(ft)})")
            # does the wrong block equal what ANOTHER sample (or this sample at the previous step) holds there?
            blk = got[k][b, :, 0, fr]
            for b2 in range(B):
                for kk, tag in ((k, "same step"), (k - 1, "previous step")):
                    if kk < 0 or (b2 == b and kk == k):
                        continue
                    e = float((blk - ref[kk][b2, :, 0, fr]).abs().max())
                    if e < 1e-3 * max(1.0, float(blk.abs().max())):
                        print(f"    == sample {b2} at the {tag} (max diff {e:.2e})")
    fails = globals().get("fails", 0) + 1
    globals()["fails"] = fails
print("FAILS", globals().get("fails", 0), "of 80")
if os.environ.get("MDM_DEC_DUP_OUT") == "1" or os.environ.get("MDM_DEC_DUP_IN") == "1" or os.environ.get("MDM_DEC_DUP_Q") == "1" or os.environ.get("MDM_PRINT_IDX") == "1":   # Output / InputProcess run twice per pass: elements in which the two results differ
    import ctypes as C
    from mdm_amd import _native
    lib = _native.load_probe()
    torch.cuda.synchronize()
    for slot in range(4, 8):
        d = C.c_double(0.0); lib.mdm_debug_get(100 + slot, C.byref(d)); bad = int(d.value)
        lib.mdm_debug_get(130 + slot, C.byref(d))
        print(f"DUP group {slot - 4}: {bad} differing elements in {int(d.value)} doubled launches")
    d = C.c_double(0.0); lib.mdm_debug_get(200, C.byref(d)); n = min(int(d.value), 256)
    idx = []
    for k in range(n):
        lib.mdm_debug_get(201 + k, C.byref(d)); idx.append(int(d.value))
    import struct
    vals, raw = [], []
    f = lambda u: struct.unpack("<f", struct.pack("<I", u))[0]
    for k in range(n):
        lib.mdm_debug_get(1000 + k, C.byref(d)); va = int(d.value)
        lib.mdm_debug_get(1256 + k, C.byref(d)); vb = int(d.value)
        f = lambda u: struct.unpack("<f", struct.pack("<I", u))[0]
        vals.append((idx[k] // 512, idx[k] % 512, f"{va:08x}", f"{vb:08x}", f(va), f(vb)))
        raw.append((idx[k], va, vb))
    vals.sort()
    inker = [(i, va, vb) for (i, va, vb) in raw if i >> 24]     # records of the in-kernel checks (gemm_f32.h MDM_F32_*_CHECK)
    print(f"records: {n} kept ({len(inker)} from the in-kernel checks)")
    for (i, va, vb) in inker[:64]:
        print(f"    in-kernel: lane {i & 255} field {(i >> 8) & 255} wave {(i >> 16) & 255}: {va:08x} ({f(va):.6g}) vs {vb:08x} ({f(vb):.6g})")
    print("DUP differing elements (row, col, first run bits, second run bits, values):")
    for v in [v for v in vals if v[0] < 32768][:64]:
        print("   ", v)
if os.environ.get("MDM_DEC_ORD") == "1":   # order probe (mdm_api.hip decoder_pass): LayerNorm rows finished vs OutputProcess's start
    import ctypes as C
    from mdm_amd import _native
    lib = _native.load_probe()
    torch.cuda.synchronize()
    for slot in range(4):
        v = []
        for what in range(4):
            d = C.c_double(0.0)
            lib.mdm_debug_get(100 + 10 * what + slot, C.byref(d))
            v.append(int(d.value))
        print(f"ORDER slot {slot}: violations {v[0]} of {v[3]} checks, largest shortfall {v[1]} rows, rows done {v[2]}")

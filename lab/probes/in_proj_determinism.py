"""in_proj alone (probe library, mdm_probe_in_proj), headline shape (256 sequences x 197 tokens, D = 512): repeated launches on
identical inputs, outputs compared bit for bit against the first launch's; for every differing launch: which plane (Q K V^T), which
sequences / heads, how many elements, the largest difference -- the signature of a stale operand stage (a whole tile) versus an
epilogue fault (rows / columns).  MDM_X3_PIPE=0/1 selects the k-loop.  Usage: python tools/in_proj_determinism.py [reps] [nseq]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mdm_amd  # noqa: F401
from mdm_amd import _native

lib = _native.load_probe()
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NSEQ = int(sys.argv[2]) if len(sys.argv) > 2 else 256
S, D, H = 197, 512, 4
SP = (S + 31) // 32 * 32
NKT = SP // 32
torch.manual_seed(0)
tok = torch.randn(NSEQ * S, D, device=dev)
w = torch.randn(3 * D, D, device=dev) / D ** 0.5
b = torch.randn(3 * D, device=dev) * (0.0 if os.environ.get("PROBE_ZERO_BIAS") else 1.0)
plane = NSEQ * SP * D
planes = torch.zeros(6 * plane, dtype=torch.int16, device=dev)
scratch = torch.empty(4 * NSEQ * S * D + 12 * D * D, dtype=torch.uint8, device=dev)
stream = torch.cuda.current_stream().cuda_stream


def run():
    lib.check(lib.mdm_probe_in_proj(tok.data_ptr(), w.data_ptr(), b.data_ptr(), planes.data_ptr(), NSEQ, S, D,
                                    scratch.data_ptr(), stream), "probe_in_proj")
    torch.cuda.synchronize()
    return planes.clone()


def f16(t):
    return t.view(torch.float16).float()


ref = run()
names = ["Qh", "Ql", "Kh", "Kl", "Vh", "Vl"]
bad = 0
for r in range(reps):
    out = run()
    if torch.equal(out, ref):
        continue
    bad += 1
    msg = []
    for pi, nm in enumerate(names):
        a, c = out[pi * plane:(pi + 1) * plane], ref[pi * plane:(pi + 1) * plane]
        ne = a != c
        if not bool(ne.any()):
            continue
        if pi < 4:   # Q / K: [seq][head][SP][128]
            idx = torch.nonzero(ne.view(NSEQ, H, SP, 128))
            seqs = sorted(set(idx[:, 0].tolist())); heads = sorted(set(idx[:, 1].tolist()))
            rows = sorted(set(idx[:, 2].tolist())); cols = sorted(set(idx[:, 3].tolist()))
            where = f"seq {seqs[:6]} head {heads} rows {rows[0]}..{rows[-1]} ({len(rows)}) cols {cols[0]}..{cols[-1]} ({len(cols)})"
        else:        # V^T: [seq][head][NKT][128 d][32 keys]
            idx = torch.nonzero(ne.view(NSEQ, H, NKT, 128, 32))
            seqs = sorted(set(idx[:, 0].tolist())); heads = sorted(set(idx[:, 1].tolist()))
            kts = sorted(set(idx[:, 2].tolist())); ds = sorted(set(idx[:, 3].tolist()))
            where = f"seq {seqs[:6]} head {heads} key tiles {kts} d {ds[0]}..{ds[-1]} ({len(ds)})"
        md = float((f16(a) - f16(c)).abs().max()) if pi % 2 == 0 else 0.0
        extra = ""
        if pi in (0, 2):   # Q / K: is the fp32 difference (hi + lo) the same in every row of a column, i.e. a per-column additive
            lo_a, lo_c = out[(pi + 1) * plane:(pi + 2) * plane], ref[(pi + 1) * plane:(pi + 2) * plane]      # vector (the bias)?
            sq, hd = seqs[0], heads[0]
            va = (f16(a) + f16(lo_a)).view(NSEQ, H, SP, 128)[sq, hd, :S]
            vc = (f16(c) + f16(lo_c)).view(NSEQ, H, SP, 128)[sq, hd, :S]
            d = (va - vc)
            colmean, colstd = d.mean(dim=0), d.std(dim=0)
            nzc = torch.nonzero(colmean.abs() > 1e-6).flatten().tolist()
            extra = (f"; seq {sq} head {hd}: per-column mean diff over rows (first 6 of {len(nzc)} columns) "
                     f"{[round(float(colmean[c_]), 5) for c_ in nzc[:6]]}, max row-std {float(colstd.max()):.2e}")
        msg.append(f"{nm}: {int(ne.sum())} elements, {where}" + (f", max-abs {md:.3e}" if pi % 2 == 0 else "") + extra)
    print(f"  launch {r}: " + " | ".join(msg), flush=True)
print(f"PIPE={os.environ.get('MDM_X3_PIPE', '1')} nseq={NSEQ}: {bad} of {reps} launches differ from the first")

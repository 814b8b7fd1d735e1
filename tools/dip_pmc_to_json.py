"""Condense the PMC passes of the DiP bench (tools/gpu_dip_pmc.sh: rocprofv3 --kernel-trace --pmc ... -- bench_dip.py, one pass per
counter group) into profiles/<name>.json: per kernel class the mean duration (kernel trace), FETCH_SIZE / WRITE_SIZE (KB as reported)
and the fabric-side bytes per launch -- FETCH_SIZE x 2 (128-byte requests tallied at 64 B: calibrated this round for 16-byte streaming,
64-byte-row LDS-DMA and 8-byte-row loads alike, profiles/r05d_pmc_calibration.md), WRITE_SIZE as is -- and the MFMA busy fraction.  Tied to
the kernel sources by bench.csrc_sha256() like profiles/r0N_pmc.json.
Usage: python tools/dip_pmc_to_json.py gpurun_out/<tag> profiles/r06_dip_pmc.json"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src, dst = sys.argv[1], sys.argv[2]


def classify(name):
    name = name.strip().strip("`")
    for pat, k in (("selfattn_block_kernel<2>", "cross_attention_seqhead"), ("selfattn_block_kernel<1>", "self_attention_seqhead"),
                   ("selfattn_block_kernel<0>", "self_attention_seqhead_layer0"), ("xattn_block_kernel", "cross_attention_block"),
                   ("attention_x3_kernel", "self_attention_x3"), ("attention_f32_kernel", "cross_attention_f32"),
                   ("outproj_finish_kernel", "outproj_finish"), ("gemm_f32_kernel", "gemm_f32 (memory K|V, InputProcess)")):
        if pat in name:
            return k
    if "gemm_x3s_kernel" in name or "true, " in name or "false, " in name:
        args = [a.strip() for a in name[name.rfind("<") + 1:name.rfind(">")].split(",")] if "<" in name else []
        if len(args) >= 12:      # <RT, NCB, NSUB, MULTI, ACT, RES, OUT_F32, OUT_PLANES, OUT_QKV, FOLD, OSTAT, EMBED>
            act, res, f32, planes, qkv, fold, ostat = args[4], args[5], args[6], args[7], args[8], args[9], args[10]
            if qkv == "true":
                return "gemm_x3s in_proj"
            if ostat == "true":
                return "gemm_x3s out_proj|cross out_proj|linear2" if res == "3" else "gemm_x3s out_proj layer0"
            if act == "1":
                return "gemm_x3s linear1"
            if f32 == "true":
                return "gemm_x3s q-projection|OutputProcess"
        return "gemm_x3s other"
    return None


ker = {}
for i in range(1, 9):
    f = os.path.join(src, f"pmc{i}.txt")
    if not os.path.isfile(f):
        continue
    cur = None
    for line in open(f):
        if line.startswith("=="):
            cur = classify(line[2:].split(" grid=")[0])
        elif cur and "mean" in line:
            ker.setdefault(cur, {})[line.split()[0]] = float(line.split()[-1])
ks = os.path.join(src, "kernel_stats.md")
if os.path.isfile(ks):
    for line in open(ks):
        if not line.startswith("| `"):
            continue
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        k = classify(cells[0])
        if k:
            e = ker.setdefault(k, {})
            calls, total = int(cells[-6]), float(cells[-5])
            e["calls"] = e.get("calls", 0) + calls
            e["total_ms"] = e.get("total_ms", 0.0) + total
out = {"kernels": {}}
for k, c in ker.items():
    e = dict(c)
    if "calls" in e and e["calls"]:
        e["avg_us"] = e["total_ms"] * 1e3 / e["calls"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["fabric_read_bytes"] = c["FETCH_SIZE"] * 1024 * 2
        e["fabric_write_bytes"] = c["WRITE_SIZE"] * 1024
        e["fabric_bytes"] = e["fabric_read_bytes"] + e["fabric_write_bytes"]
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)
    out["kernels"][k] = e
import bench  # noqa: E402
out["csrc_sha256"] = bench.csrc_sha256()
try:
    out["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    out["commit"] = None
out["source"] = f"tools/gpu_dip_pmc.sh {os.path.basename(src.rstrip('/'))} (rocprofv3 --kernel-trace --pmc around bench_dip.py, separate passes)"
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))

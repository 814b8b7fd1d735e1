"""Condense gpurun_out/<tag>/pmc*.txt (tools/gpu_prof.sh ... pmc) + kernel_stats.md into profiles/<name>.json:
per kernel the mean duration, FETCH_SIZE / WRITE_SIZE (KB as reported) and the derived HBM-side bytes per launch with the
gfx950 correction MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 64 B per 128-B request on wide streaming reads:
x2; WRITE_SIZE as is -- both calibrated in round 1 on layernorm_kernel, whose traffic is known exactly), MFMA busy fraction
and the shader clock.  The file records the sha256 of the kernel sources it was taken on (bench.csrc_sha256): bench.py
quotes `roofline.traffic` from it only while the sources are the same.
Usage: python tools/pmc_to_json.py gpurun_out/p3 profiles/r02_pmc.json"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src, dst = sys.argv[1], sys.argv[2]


def classify(name):
    """-> (section, key) for the kernels of the sampling loop; the GEMM is told apart by its template arguments
    <WAVES, ACT, RES, OUT_F32, OUT_PLANES, OUT_QKV, ABL, FOLD, OSTAT, EMBED, T16, F6> read from the END (names arrive
    truncated on the left)."""
    name = name.strip().strip("`")
    if "gemm_x3_kernel" in name or re.search(r"(true|false)(, (true|false)){4}>", name):
        args = [a.strip() for a in name[name.rfind("<") + 1:name.rfind(">")].split(",")] if "<" in name else \
            [a.strip() for a in name[:name.rfind(">")].split(",")]
        if len(args) >= 11:
            # round 3 appended a 13th template argument (PIPE): names arrive truncated on the LEFT, so count from the right --
            # 13 arguments (or 12 + a cut one) are told from 12 by whether the kernel name carries all of <8, ...>
            full = name[name.rfind("<") + 1:name.rfind(">")].split(",") if "<" in name else []
            has_pipe = len(full) == 13 or ("<" not in name and os.environ.get("PMC_PIPE_ARG", "1") == "1")
            tail = args[::-1][1:12] if has_pipe else args[::-1][:11]
            # round 4 appended a 14th argument (NCB, an integer): the last argument is then a number, PIPE the one before it
            if args[-1].strip().isdigit() and len(args) >= 13 and args[-2].strip() in ("true", "false"):
                tail = args[::-1][2:13]
            f6, t16, embed, ostat, fold, abl, qkv, planes, f32, res, act = tail
            b = lambda v: v == "true"   # noqa: E731
            if b(embed):
                return "gemm", "InputProcess"
            if b(qkv):
                return "gemm", "in_proj" if b(fold) else "in_proj_layer0"
            if b(ostat):
                return "gemm", "out_proj|linear2" if res == "3" else "out_proj_layer0"
            if act == "1":
                return "gemm", "linear1"
            if b(f32) and b(fold):
                return "gemm", "OutputProcess"
            return "gemm", "other<" + ",".join(args[-11:]) + ">"
    for pat, k in (("fold_layernorm_kernel", "fold_layernorm (mdm_prepare)"), ("attention_x3_kernel", "attention"), ("pose_to_planes_kernel", "pose_to_planes"), ("layernorm_kernel", "layernorm"),
                   ("outproj_finish_kernel", "outproj_finish"), ("cond_token_kernel", "cond_token")):
        if pat in name:
            return "other", k
    return None, None


ker = {}
for i in range(1, 9):
    f = os.path.join(src, f"pmc{i}.txt")
    if not os.path.isfile(f):
        continue
    cur = None
    for line in open(f):
        if line.startswith("=="):
            cur = classify(line[2:].split(" grid=")[0])
        elif cur and cur[0] and "mean" in line:
            ker.setdefault(cur, {})[line.split()[0]] = float(line.split()[-1])
ks = os.path.join(src, "kernel_stats.md")
if os.path.isfile(ks):
    for line in open(ks):
        if not line.startswith("| `"):
            continue
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        k = classify(cells[0])
        if k[0]:
            ker.setdefault(k, {})["avg_us"] = float(cells[3])
            ker[k]["calls"] = int(cells[1])
out = {"gemm": {}, "other": {}}
for (sec, k), c in ker.items():
    e = dict(c)
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_read_bytes"] = c["FETCH_SIZE"] * 1024 * 2
        e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
        e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
    if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
        e["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * cyc)   # 256 CUs x 4 SIMDs
        if "avg_us" in c:
            e["shader_clock_ghz_under_pmc"] = cyc / c["avg_us"] / 1e3
    out[sec][k] = e
import bench  # noqa: E402
out["csrc_sha256"] = bench.csrc_sha256()
try:
    out["commit"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    out["commit"] = None
out["source"] = f"tools/gpu_prof.sh {os.path.basename(src.rstrip('/'))} pmc (rocprofv3 --kernel-trace --pmc, separate passes)"
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))

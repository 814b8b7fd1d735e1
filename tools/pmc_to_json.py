"""Condense gpurun_out/<tag>/pmc*.txt (tools/gpu_prof.sh ... pmc) + kernel_stats.md into profiles/<name>.json:
per kernel the mean duration, FETCH_SIZE / WRITE_SIZE (KB as reported) and the derived HBM-side bytes per launch with the
gfx950 correction MI355X_MICROARCH.md prescribes (FETCH_SIZE counts 64 B per 128-B request on wide streaming reads:
x2; WRITE_SIZE as is -- both calibrated here on layernorm_kernel, whose traffic is known exactly), MFMA busy fraction
and the shader clock.  Usage: python tools/pmc_to_json.py gpurun_out/p3 profiles/r01_pmc.json"""
import json, re, sys, os

src, dst = sys.argv[1], sys.argv[2]
ker = {}

def key(name):
    name = name.strip()
    # template tails: <WAVES, ACT, RES, OUT_F32, OUT_PLANES, OUT_QKV, ABL, FOLD, OSTAT, EMBED, T16> (names arrive truncated on
    # the left); the patterns below match with or without the trailing EMBED / T16 arguments
    for pat, k in [(r"0, 0, false, false, true, 0, true, false(, false)?(, true|, false)?>", "gemm_f16x3<in_proj (LayerNorm folded) -> Q/K/V^T planes>"),
                   (r"0, 0, false, false, true, 0, false, false(, false)?(, true|, false)?>", "gemm_f16x3<in_proj layer 0 -> Q/K/V^T planes>"),
                   (r"0, 3, false, true, false, 0, false, true(, false)?(, true|, false)?>", "gemm_f16x3<out_proj | linear2, LayerNorm residual, planes + row stats>"),
                   (r"0, 2, false, true, false, 0, false, true(, false)?(, true|, false)?>", "gemm_f16x3<out_proj layer 0, planes + row stats>"),
                   (r"1, 0, false, true, false, 0, true, false(, false)?(, true|, false)?>", "gemm_f16x3<linear1 (LayerNorm folded) + GELU -> planes>"),
                   (r"0, 0, true, false, false, 0, true, false(, false)?(, true|, false)?>", "gemm_f16x3<OutputProcess (LayerNorm folded)>"),
                   (r"0, 1, false, true, false, 0, false, false, true(, true|, false)?>", "gemm_f16x3<InputProcess (EMBED)>"),
                   (r"pose_to_planes_kernel", "pose_to_planes"),
                   (r"attention_x3_kernel", "attention_f16x3"), (r"layernorm_kernel", "layernorm"),
                   (r"outproj_finish_kernel", "outproj_finish"), (r"EmbedEpilogue", "gemm_f32<InputProcess>")]:
        if re.search(pat, name):
            return k
    return None

for i in range(1, 9):
    f = os.path.join(src, f"pmc{i}.txt")
    if not os.path.isfile(f):
        continue
    cur = None
    for line in open(f):
        if line.startswith("=="):
            cur = key(line[2:])
        elif cur and "mean" in line:
            name, _, val = line.split()[0], None, float(line.split()[-1])
            ker.setdefault(cur, {})[name] = val
ks = os.path.join(src, "kernel_stats.md")
if os.path.isfile(ks):
    for line in open(ks):
        if not line.startswith("| `"):
            continue
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        k = key(cells[0])
        if k:
            ker.setdefault(k, {})["avg_us"] = float(cells[3])
            ker[k]["calls"] = int(cells[1])
out = {}
for k, c in ker.items():
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
    out[k] = e
json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))

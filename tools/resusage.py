"""Compile csrc/mdm_api.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "motion-diffusion-model_amd", "csrc", "mdm_api.hip")
out = os.path.join(root, "motion-diffusion-model_amd", "csrc", "libmdm_hip.so")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-pass-failed",
       "-Rpass-analysis=kernel-resource-usage", src, "-o", out] + sys.argv[1:]
p = subprocess.run(cmd, capture_output=True, text=True)
if p.returncode:
    print(p.stderr[-4000:]); sys.exit(1)
cur = {}
rows = []
for line in p.stderr.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line) or re.search(r"\d+:\d+:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        if cur: rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
if cur: rows.append(cur)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*", "", name.replace("void ", "").replace("mdm::", ""))[:70]
    print(f"{name:70s} vgpr={r.get('VGPRs','?'):>4s} agpr={r.get('AGPRs','?'):>3s} sgpr={r.get('TotalSGPRs','?'):>4s} "
          f"scratch={r.get('ScratchSize [bytes/lane]','?'):>4s} occ={r.get('Occupancy [waves/SIMD]','?')} "
          f"sspill={r.get('SGPRs Spill','?')} vspill={r.get('VGPRs Spill','?')}")

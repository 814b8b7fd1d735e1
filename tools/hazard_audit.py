"""Static wait-state audit of a gfx950 code object: point the disassembly auditor that found the
`v_fma_mix -> v_mfma` hazard (profiles/r05m_fma_mix_hazard.md) at ANY build of the library -- in particular the SLP build
(no -fno-slp-vectorize: packed fp32 VALU math), the one that returned rare wrong values beside a co-resident LDS-using kernel in
rounds 3-4 (profiles/r03g_dip_groups.md, r04c_packed_math.md).

Straight-line check per kernel, instruction by instruction: wait states are counted over the textual order, which is the fall-through
path (an unconditional branch / s_endpgm ends a path); a dependency across a TAKEN branch is not seen -- this is a search for sites,
not a proof.  Rules (LLVM's GCNHazardRecognizer for gfx940 / gfx950 as far as this path uses them; one wait state = one issued
instruction, `s_nop N` = N + 1):
  A  VALU (non-MFMA) writes a VGPR, v_mfma reads it (A, B or C operand)                      >= 2 wait states in between
  B  v_mfma (XDL) writes VGPRs, a VALU / memory instruction reads or overwrites one of them   >= passes + 2
     (what hipcc itself pads to: the tightest compiler-generated site of every opcode sits exactly there -- `per_opcode_min_wait`)
     passes: 32x32x16_f16 / 32x32x2_f32 ... by opcode table below (4 cycles per pass)
  C  v_pk_*_f32 / v_fma_mix* / v_permlane*_swap written register read by ANYTHING one instruction later: not a documented hazard
     (VALU -> VALU is interlocked); listed only as a count of how tightly packed results are consumed.

Usage: python tools/hazard_audit.py <lib.so | code-object> [--kernels substr,substr] [--verbose]
Prints one JSON line per rule with the number of sites, the tightest margins and (verbose) the first sites."""
import json
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# passes of the XDL ops this library issues (cycles / 4)
MFMA_PASSES = {"v_mfma_f32_32x32x16_f16": 8, "v_mfma_f32_32x32x16_bf16": 8, "v_mfma_f32_16x16x32_f16": 4, "v_mfma_f32_16x16x32_bf16": 4,
               "v_mfma_f32_32x32x2_f32": 16, "v_mfma_f32_16x16x4_f32": 8, "v_mfma_f32_32x32x8_f16": 8, "v_mfma_f32_16x16x16_f16": 4,
               "v_mfma_scale_f32_32x32x64_f8f6f4": 16, "v_mfma_f32_32x32x64_f8f6f4": 16, "v_mfma_scale_f32_16x16x128_f8f6f4": 8,
               "v_mfma_f32_16x16x128_f8f6f4": 8}


# the textual successor of these is NOT on the fall-through path
END_OF_PATH = ("s_branch", "s_endpgm", "s_setpc_b64", "s_swappc_b64")


def disassemble(path):
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        with open(path, "rb") as f, open(so, "wb") as g:
            g.write(f.read())
        subprocess.run([OBJDUMP, "--offloading", so], cwd=tmp, check=True, capture_output=True, text=True)
        cos = [f for f in os.listdir(tmp) if "amdgcn" in f]
        target = os.path.join(tmp, cos[0]) if cos else so
        return subprocess.run([OBJDUMP, "-d", target], check=True, capture_output=True, text=True).stdout


def vregs(tok):
    """v7 / v[4:7] -> register numbers; AGPRs are a separate file (a[..]) and are ignored (this library uses none)."""
    tok = tok.strip()
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def operands(ins):
    parts = ins.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    return parts[0], [t for t in re.split(r",\s*|\s+", parts[1]) if t]


def writes_reads(op, toks):
    """(written VGPRs, read VGPRs) of one instruction, conservatively: stores / atomics without return read everything; swap ops write
    both of their first two operands; everything else writes its first operand and reads the rest."""
    if not toks:
        return set(), set()
    if op.startswith(("global_store", "buffer_store", "flat_store", "ds_write", "ds_store", "scratch_store")) or op.startswith("s_"):
        return set(), set().union(*[vregs(t) for t in toks])
    if op.startswith(("global_load_lds", "buffer_load")) and "lds" in toks:
        return set(), set().union(*[vregs(t) for t in toks])
    if "swap" in op:
        w = vregs(toks[0]) | (vregs(toks[1]) if len(toks) > 1 else set())
        return w, set(w)
    rd = set().union(*[vregs(t) for t in toks[1:]]) if len(toks) > 1 else set()
    if op.startswith("v_mfma"):
        return vregs(toks[0]), rd
    return vregs(toks[0]), rd


def audit(asm, only=None, verbose=False):
    kernels, cur = [], None
    for ln in asm.splitlines():
        m = re.match(r"^[0-9a-f]+ <([^>]+)>:", ln)
        if m:
            cur = (m.group(1), [])
            kernels.append(cur)
            continue
        if cur is not None and ln.startswith("\t"):
            parts = ln.split("\t")
            if len(parts) > 1:
                ins = parts[1].split("//")[0].strip()
                if ins:
                    cur[1].append(ins)
    res = {"A": {"sites": 0, "checked": 0, "min_wait": None, "examples": []},
           "B": {"sites": 0, "checked": 0, "min_margin": None, "examples": []},
           "C": {"adjacent_consumers": 0, "packed_ops": 0}}
    nk = 0
    for name, ins in kernels:
        if only and not any(s in name for s in only):
            continue
        nk += 1
        dec = [(operands(i)) for i in ins]
        wr = [writes_reads(op, toks) for op, toks in dec]
        for i, (op, toks) in enumerate(dec):
            if op.startswith("v_mfma"):
                srcs = wr[i][1]
                # rule A: look back up to 2 wait states
                waits = 0
                for back in range(1, 4):
                    if i - back < 0 or waits >= 2:
                        break
                    pop, ptoks = dec[i - back]
                    if pop in END_OF_PATH:
                        break
                    if pop.startswith("v_") and not pop.startswith("v_mfma") and (wr[i - back][0] & srcs):
                        res["A"]["sites"] += 1
                        if len(res["A"]["examples"]) < 8:
                            res["A"]["examples"].append({"kernel": name[:80], "at": i, "valu": ins[i - back], "mfma": ins[i], "wait_states": waits})
                    waits += (int(ptoks[0], 0) + 1) if pop == "s_nop" and ptoks else 1
                # for the statistics: nearest VALU writer of any source within 8 instructions
                waits = 0
                for back in range(1, 9):
                    if i - back < 0:
                        break
                    pop, ptoks = dec[i - back]
                    if pop in END_OF_PATH:
                        break
                    if pop.startswith("v_") and not pop.startswith("v_mfma") and (wr[i - back][0] & srcs):
                        res["A"]["checked"] += 1
                        mw = res["A"]["min_wait"]
                        res["A"]["min_wait"] = waits if mw is None else min(mw, waits)
                        break
                    waits += (int(ptoks[0], 0) + 1) if pop == "s_nop" and ptoks else 1
                # rule B: look forward
                need = MFMA_PASSES.get(op, 16) + 2
                dst = wr[i][0]
                waits = 0
                for fwd in range(1, need + 8):
                    if i + fwd >= len(dec) or waits >= need + 4:
                        break
                    nop_, ntoks = dec[i + fwd]
                    if nop_ in END_OF_PATH:
                        break
                    if nop_.startswith("v_mfma"):
                        # a dependent MFMA (same accumulator) is its own, interlocked, case; an independent one occupies the pipe for
                        # its own passes: count them (conservatively 1 wait state as LLVM does)
                        waits += 1
                        continue
                    w2, r2 = wr[i + fwd]
                    if (w2 | r2) & dst and not nop_.startswith("s_"):
                        res["B"]["checked"] += 1
                        pm = res["B"].setdefault("per_opcode_min_wait", {})
                        pm[op] = min(pm.get(op, 10 ** 6), waits)
                        margin = waits - need
                        mm = res["B"]["min_margin"]
                        res["B"]["min_margin"] = margin if mm is None else min(mm, margin)
                        if margin < 0:
                            res["B"]["sites"] += 1
                            if len(res["B"]["examples"]) < 8:
                                res["B"]["examples"].append({"kernel": name[:80], "at": i, "mfma": ins[i], "user": ins[i + fwd], "wait_states": waits, "need": need})
                        break
                    waits += (int(ntoks[0], 0) + 1) if nop_ == "s_nop" and ntoks else 1
            if re.match(r"v_pk_[a-z0-9_]*f32|v_fma_mix|v_permlane\d+_swap", op):
                res["C"]["packed_ops"] += 1
                if i + 1 < len(dec) and (wr[i][0] & (wr[i + 1][1])):
                    res["C"]["adjacent_consumers"] += 1
    res["kernels"] = nk
    res["instructions"] = sum(len(i) for n, i in kernels if not only or any(s in n for s in only))
    if not verbose:
        for k in ("A", "B"):
            res[k]["examples"] = res[k]["examples"][:3]
    return res


if __name__ == "__main__":
    path = sys.argv[1]
    only = sys.argv[sys.argv.index("--kernels") + 1].split(",") if "--kernels" in sys.argv else None
    asm = disassemble(path) if not path.endswith(".s") else open(path).read()
    pk = sorted(set(re.findall(r"\bv_pk_[a-z0-9_]*f32\b", asm)))
    out = audit(asm, only, "--verbose" in sys.argv)
    out["file"] = path
    out["packed_fp32_opcodes"] = {m: asm.count(m) for m in pk}
    print(json.dumps(out, indent=1))

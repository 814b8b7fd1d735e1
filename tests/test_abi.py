"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/mdm_hip.h declares.
Only host-side entry points are called here (no kernel launches): argument validation and error reporting."""
import ctypes as C
import os
import re

import pytest

from helpers import ROOT


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()                                     # cross-compiles csrc/ for gfx950 if stale
    from mdm_amd import _native
    return _native.MdmLib(_native.LIB_PATH)


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "mdm_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mdm_[a-z_0-9]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(lib):
    from mdm_amd import _native
    names = _declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib.lib, n), f"{n} declared in include/mdm_hip.h but not exported"
    assert sorted(_native.EXPORTED_SYMBOLS) == names       # the ctypes view covers the whole header
    assert lib.mdm_abi_version() == 10


def test_probe_surface_is_not_in_the_production_library(lib):
    """include/mdm_hip_probe.h (experiment switches, the f16f6 kernel) is exported by libmdm_hip_probe.so only."""
    from mdm_amd import _native
    src = open(os.path.join(ROOT, "include", "mdm_hip_probe.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(mdm_[a-z_0-9]+)\s*\(", src)))
    assert names == sorted(_native.PROBE_SYMBOLS)
    probe = _native.MdmLib(_native.PROBE_LIB_PATH)
    for n in names:
        assert not hasattr(lib.lib, n), f"{n} leaked into the production library"
        assert hasattr(probe.lib, n)
    assert not lib.has_probes and probe.has_probes


def test_product_library_reads_no_environment_variable(lib):
    """rounds 3-4 steered the product library through ~10 environment variables, several read on the
    launch path.  Now: handle options (mdm_set_option); the product binary does not even import getenv (the probe build, the
    experiment surface of tools/, still does)."""
    import subprocess
    from mdm_amd import _native
    und = subprocess.run(["nm", "-D", "--undefined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und, [ln for ln in und.splitlines() if "getenv" in ln]
    probe = subprocess.run(["nm", "-D", "--undefined-only", _native.PROBE_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "getenv" in probe
    h = C.c_void_p()
    cfg = _cfg()
    lib.check(lib.mdm_create(C.byref(cfg), C.byref(h)), "mdm_create")
    v = C.c_int32(-1)
    lib.check(lib.mdm_get_option(h, _native.OPTIONS["small_gemm_max_seqs"], C.byref(v)), "mdm_get_option")
    assert v.value == 80
    lib.check(lib.mdm_set_option(h, _native.OPTIONS["small_gemm_max_seqs"], 0), "mdm_set_option")
    lib.check(lib.mdm_get_option(h, _native.OPTIONS["small_gemm_max_seqs"], C.byref(v)), "mdm_get_option")
    assert v.value == 0
    assert lib.mdm_set_option(h, 99, 1) == -1 and b"unknown key" in lib.lib.mdm_last_error()
    assert lib.mdm_set_option(h, _native.OPTIONS["small_gemm_row_tiles"], 7) == -1
    lib.mdm_destroy(h)


def test_product_library_contains_no_packed_fp32_valu_math(lib, tmp_path):
    """VERDICT r04, What's weak 2: the DiP path's rare wrong values beside a foreign LDS-using kernel were tied to PACKED fp32 VALU math
    (cured by -fno-slp-vectorize: 0 of 640 window loops) without a root cause, and one hand-written packed instruction stayed in the
    product: the 2-vector subtraction of the operand split (v_pk_add_f32, 4,327 instances).  Since round 5 the split takes
    v_fma_mix{lo,hi}_f16 instead (common.h split2_p16; bit-identical), and this test disassembles the gfx950 code object of the product
    library: not a single v_pk_*_f32 instruction may be left -- the whole instruction class is out of the binary, not just out of the
    vectorizer's reach."""
    import shutil
    import subprocess
    from mdm_amd import _native
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.isfile(objdump):
        pytest.skip("llvm-objdump of the ROCm toolchain is not installed")
    so = str(tmp_path / "lib.so")
    shutil.copy(_native.LIB_PATH, so)
    subprocess.run([objdump, "--offloading", so], cwd=str(tmp_path), check=True, capture_output=True, text=True)
    cos = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert len(cos) == 1, os.listdir(tmp_path)
    asm = subprocess.run([objdump, "-d", str(tmp_path / cos[0])], check=True, capture_output=True, text=True).stdout
    assert asm.count("v_mfma_f32_32x32x16_f16") > 1000                     # (it IS the kernels' disassembly)
    packed = sorted(set(re.findall(r"\bv_pk_[a-z0-9_]*f32\b", asm)))
    assert packed == [], {m: asm.count(m) for m in packed}
    assert asm.count("v_fma_mixlo_f16") > 1000
    # ... and every v_fma_mixhi_f16 of the split is followed by its two wait states: a v_mfma that reads the split's result one
    # instruction behind the bare pair returned wrong, run-to-run different values (xattn_block_kernel<4, 3>; common.h split2_p16,
    # profiles/r05m_fma_mix_hazard.md) -- hipcc's hazard recognizer does not see into inline asm
    ops = [ln.split("\t")[1].split()[0:2] for ln in asm.splitlines() if ln.startswith("\t") and len(ln.split("\t")) > 1 and ln.split("\t")[1].strip()]
    bare = sum(1 for a, b in zip(ops, ops[1:]) if a[0] == "v_fma_mixhi_f16" and b != ["s_nop", "1"])
    assert bare == 0, f"{bare} v_fma_mixhi_f16 without 's_nop 1' behind them"
    # the general form of the same audit, over every VALU instruction this library issues from inline asm (which hipcc's hazard
    # recognizer cannot see): a v_mfma never reads a register such an instruction wrote fewer than two wait states earlier
    asm_valu = ("v_fma_mixlo_f16", "v_fma_mixhi_f16", "v_fma_mix_f32", "v_permlane16_swap_b32", "v_permlane32_swap_b32")
    lines = [ln.split("\t")[1].split("//")[0].strip() for ln in asm.splitlines() if ln.startswith("\t") and len(ln.split("\t")) > 1]
    lines = [ln for ln in lines if ln]

    def regs(tok):
        m = re.match(r"v\[(\d+):(\d+)\]", tok)
        if m:
            return range(int(m.group(1)), int(m.group(2)) + 1)
        m = re.match(r"v(\d+)$", tok)
        return range(int(m.group(1)), int(m.group(1)) + 1) if m else ()

    close_calls = 0
    for i, ln in enumerate(lines):
        if not ln.startswith("v_mfma"):
            continue
        srcs = {r for tok in re.split(r"[ ,]+", ln)[2:] for r in regs(tok)}
        waits = 0
        for back in range(1, 3):                       # the two instructions in front of the MFMA
            prev = lines[i - back] if i - back >= 0 else ""
            op = prev.split()[0] if prev else ""
            if op.startswith(asm_valu) and waits < 2:
                written = {r for tok in re.split(r"[ ,]+", prev)[1:3] for r in regs(tok)} if "swap" in op else set(regs(re.split(r"[ ,]+", prev)[1]))
                if written & srcs:
                    close_calls += 1
            waits += (int(prev.split()[1]) + 1) if op == "s_nop" else 1
    assert close_calls == 0, f"{close_calls} v_mfma read an inline-asm VALU result fewer than two wait states behind its write"


def test_static_wait_state_audit_of_the_product_library(lib):
    """tools/hazard_audit.py -- the generalisation of the v_fma_mix audit above to EVERY VALU -> v_mfma read
    (>= 2 wait states) and every v_mfma write -> VALU / LDS / memory read or overwrite (>= passes + 2, the distance hipcc itself pads
    to) -- over the shipped gfx950 code object: no site.  (Run over the SLP build too, by hand: profiles/r06b_slp_hazard_audit.md.)"""
    import importlib.util
    from mdm_amd import _native
    if not os.path.isfile("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("llvm-objdump of the ROCm toolchain is not installed")
    spec = importlib.util.spec_from_file_location("hazard_audit", os.path.join(ROOT, "tools", "hazard_audit.py"))
    ha = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ha)
    res = ha.audit(ha.disassemble(_native.LIB_PATH))
    assert res["kernels"] > 90 and res["A"]["checked"] > 1000 and res["B"]["checked"] > 10      # (it found the pairs it is about)
    assert res["A"]["sites"] == 0 and res["A"]["min_wait"] >= 2, res["A"]
    assert res["B"]["sites"] == 0 and res["B"]["min_margin"] >= 0, res["B"]


def test_no_cuda_or_torch_in_the_abi():
    src = open(os.path.join(ROOT, "include", "mdm_hip.h")).read()
    assert "torch" not in src.lower().replace("pytorch", "").replace("a torch tensor", "")
    assert "#include <hip" not in src and "cuda" not in src.lower()


def _cfg(**over):
    from mdm_amd import _native
    base = dict(njoints=263, nfeats=1, latent_dim=512, ff_size=1024, num_layers=8, num_heads=4, clip_dim=512,
                max_len=5000, mask_frames=1)
    base.update(over)
    return _native.MdmConfig(**base)


def test_create_validates_shapes_and_reports_errors(lib):
    h = C.c_void_p()
    assert lib.mdm_create(C.byref(_cfg()), C.byref(h)) == 0 and h.value
    # state-dict contract of utils/model_util.py:8-15: unexpected keys and wrong sizes are rejected by name
    buf = (C.c_float * 16)()
    rc = lib.mdm_set_weight(h, b"seqTransEncoder.layers.8.linear1.bias", C.addressof(buf), 1024)
    assert rc == -1 and b"unexpected state-dict key" in lib.mdm_last_error()
    rc = lib.mdm_set_weight(h, b"embed_text.bias", C.addressof(buf), 511)
    assert rc == -1 and b"size mismatch" in lib.mdm_last_error()
    # nothing may run before every weight is registered and mdm_prepare has been called
    assert lib.mdm_prepare(h, C.addressof(buf), 64, None) == -2 and b"missing weight" in lib.mdm_last_error()
    assert lib.mdm_forward(h, 1, 1, 1, None, 1, 8, 0, 1, 1, 0, None) == -2
    assert lib.mdm_workspace_bytes(h, 256, 196) > 256 * 197 * (512 * 5 + 1024) * 4
    assert lib.mdm_const_bytes(h) >= 512 * 264 * 4 + 2 * 5000 * 512 * 4
    lib.mdm_destroy(h)
    for bad in (dict(latent_dim=384), dict(num_heads=8), dict(ff_size=1023), dict(num_layers=0)):
        h2 = C.c_void_p()
        assert lib.mdm_create(C.byref(_cfg(**bad)), C.byref(h2)) < 0
        assert lib.mdm_last_error()


def test_product_path_fails_loudly_without_the_extension(monkeypatch, tmp_path):
    from mdm_amd import _native
    monkeypatch.setenv("MDM_HIP_LIB", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_native, "_LIB", None)
    with pytest.raises(_native.MdmError, match="no CPU fallback"):
        _native.load_native()


def test_dip_window_loop_entry_points_validate_before_touching_the_device(lib):
    """mdm_sample_loop_dec / mdm_workspace_bytes_dec_loop (ABI 5): host-side contract only -- sizes, and the refusals that
    come before any launch (unprepared model, trans_enc handle, null pointers)."""
    from mdm_amd import _native
    h = C.c_void_p()
    assert lib.mdm_create(C.byref(_cfg(arch=_native.ARCH["trans_dec"], context_len=20, clip_dim=768)), C.byref(h)) == 0
    one = lib.mdm_workspace_bytes_dec(h, 64, 40, 24)
    loop = lib.mdm_workspace_bytes_dec_loop(h, 64, 40, 24, 10)
    # the loop adds the model-output buffer, the hoisted text K | V of 8 layers and the per-step time rows
    assert loop >= one + 64 * 263 * 40 * 4 + 8 * 64 * 24 * 1024 * 4 + 8 * 10 * 1024 * 4
    assert lib.mdm_workspace_bytes_dec_loop(h, 64, 40, 24, 0) == 0 and lib.mdm_workspace_bytes_dec_loop(None, 64, 40, 24, 10) == 0
    p = _native.MdmSampleDecParams()
    buf = (C.c_float * 16)()
    assert lib.mdm_sample_loop_dec(h, C.byref(p), C.addressof(buf), C.addressof(buf), 64, None) == -2     # not prepared
    assert lib.mdm_sample_loop_dec(None, C.byref(p), C.addressof(buf), C.addressof(buf), 64, None) < 0
    assert lib.mdm_last_error()
    lib.mdm_destroy(h)
    h2 = C.c_void_p()
    assert lib.mdm_create(C.byref(_cfg()), C.byref(h2)) == 0
    assert lib.mdm_sample_loop_dec(h2, C.byref(p), C.addressof(buf), C.addressof(buf), 64, None) < 0      # trans_enc handle
    lib.mdm_destroy(h2)

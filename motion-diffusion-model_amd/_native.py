"""ctypes binding of libmdm_hip.so (C ABI: include/mdm_hip.h).

The product path REQUIRES the HIP library: `load_native()` raises if it is missing -- there is no CPU
or eager-PyTorch fallback.  (tests/emu builds a CPU emulation of the same sources purely as test
infrastructure and points `MdmLib` at it explicitly.)
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libmdm_hip.so"
LIB_PATH = os.path.join(_HERE, "csrc", LIB_NAME)
PROBE_LIB_PATH = os.path.join(_HERE, "csrc", "libmdm_hip_probe.so")   # -DMDM_PROBES build: tools/ and probe-only tests

MDM_OK = 0
BRANCH_COND, BRANCH_UNCOND, BRANCH_BOTH = 0, 1, 2
ACT_NONE, ACT_GELU, ACT_SILU = 0, 1, 2
PRECISIONS = {"f32": 0, "f16x3": 1}
PROF_CLASSES = ("linear", "attention", "layernorm", "embed", "outproj", "elementwise")

EXPORTED_SYMBOLS = [
    "mdm_abi_version", "mdm_build_info", "mdm_last_error", "mdm_create", "mdm_destroy", "mdm_set_weight", "mdm_const_bytes",
    "mdm_prepare", "mdm_workspace_bytes", "mdm_forward", "mdm_sampler_step", "mdm_randn", "mdm_sample_loop",
    "mdm_linear", "mdm_layernorm", "mdm_attention", "mdm_profile_enable", "mdm_profile_read", "mdm_profile_reset",
    "mdm_set_precision", "mdm_linear_x3", "mdm_linear_x3_scratch_bytes", "mdm_attention_x3", "mdm_attention_x3_scratch_bytes",
    "mdm_recover_from_ric", "mdm_workspace_bytes_dec", "mdm_forward_dec", "mdm_workspace_bytes_dec_loop",
    "mdm_sample_loop_dec", "mdm_weights_in_range", "mdm_set_option", "mdm_get_option", "mdm_set_time_add",
]
# include/mdm_hip_probe.h: exported by the probe build only
PROBE_SYMBOLS = ["mdm_debug_set", "mdm_debug_get", "mdm_linear_f16f6", "mdm_linear_f16f6_scratch_bytes", "mdm_probe_in_proj"]
ABI_VERSION = 10
# include/mdm_hip.h MDM_OPT_*: per-handle run-time options (the library reads no environment variable)
OPTIONS = {"small_gemm_max_seqs": 1, "small_gemm_row_tiles": 2, "dec_fused_xattn": 3, "dec_fused_selfattn": 4, "attn_direct_out": 5,
           "dec_time_token": 6}
ARCH = {"trans_enc": 0, "trans_dec": 1}


class MdmConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("njoints", "nfeats", "latent_dim", "ff_size", "num_layers", "num_heads",
                                         "clip_dim", "max_len", "mask_frames", "arch", "context_len")]


class MdmStep(C.Structure):
    _fields_ = [("a_x0", C.c_float), ("a_xt", C.c_float), ("sigma", C.c_float), ("clip_denoised", C.c_int32),
                ("seed", C.c_uint64), ("sample_base", C.c_uint32), ("draw", C.c_uint32), ("const_noise", C.c_int32)]


class MdmSampleParams(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int32), ("num_timesteps", C.c_int32), ("start_index", C.c_int32),
        ("a_x0", C.c_void_p), ("a_xt", C.c_void_p), ("sigma", C.c_void_p), ("timestep_map", C.c_void_p),
        ("text_embed_dev", C.c_void_p), ("scale_dev", C.c_void_p), ("lengths_dev", C.c_void_p),
        ("inpaint_mask_dev", C.c_void_p), ("inpaint_motion_dev", C.c_void_p), ("noise_dev", C.c_void_p),
        ("seed", C.c_uint64), ("sample_base", C.c_uint32), ("clip_denoised", C.c_int32),
        ("force_uncond", C.c_int32), ("x0_dev", C.c_void_p), ("dump_steps", C.c_void_p), ("num_dump", C.c_int32),
        ("dump_dev", C.c_void_p), ("const_noise", C.c_int32),
    ]


class MdmSampleDecParams(C.Structure):
    """mdm_sample_dec_params_t: the loop block (T = pred_len, text_embed_dev = token-major text tokens) + the DiP inputs."""
    _fields_ = [("loop", MdmSampleParams), ("ntok", C.c_int32), ("prefix_dev", C.c_void_p),
                ("text_lengths_dev", C.c_void_p)]


class MdmError(RuntimeError):
    pass


class MdmLib:
    """Thin typed view of the shared library.  Pointers are passed as integers (tensor.data_ptr())."""

    def __init__(self, path):
        if not os.path.isfile(path):
            raise MdmError(
                f"{path} not found: the MI355X HIP extension is not built. Run `python __graft_entry__.py build` "
                f"(hipcc --offload-arch=gfx950). There is no CPU fallback for this path.")
        self.path = path
        lib = self.lib = C.CDLL(path)
        vp, i32, i64, u32, u64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_size_t
        P = C.POINTER
        sig = {
            "mdm_abi_version": (C.c_int, []),
            "mdm_last_error": (C.c_char_p, []),
            "mdm_build_info": (C.c_char_p, []),
            "mdm_create": (C.c_int, [P(MdmConfig), P(vp)]),
            "mdm_destroy": (None, [vp]),
            "mdm_set_weight": (C.c_int, [vp, C.c_char_p, vp, i64]),
            "mdm_const_bytes": (sz, [vp]),
            "mdm_prepare": (C.c_int, [vp, vp, sz, vp]),
            "mdm_workspace_bytes": (sz, [vp, i32, i32]),
            "mdm_forward": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]),
            "mdm_sampler_step": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, P(MdmStep), vp]),
            "mdm_randn": (C.c_int, [vp, vp, vp, f32, f32, i32, i32, u64, u32, u32, vp]),
            "mdm_sample_loop": (C.c_int, [vp, P(MdmSampleParams), vp, vp, sz, vp]),
            "mdm_linear": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
            "mdm_layernorm": (C.c_int, [vp, vp, vp, i32, i32, vp]),
            "mdm_attention": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, vp]),
            "mdm_set_precision": (C.c_int, [vp, i32]),
            "mdm_weights_in_range": (C.c_int, [vp, P(i32), vp]),
            "mdm_linear_x3_scratch_bytes": (sz, [i32, i32, i32]),
            "mdm_linear_x3": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
            "mdm_attention_x3_scratch_bytes": (sz, [i32, i32, i32]),
            "mdm_attention_x3": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, vp, sz, vp]),
            "mdm_recover_from_ric": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
            "mdm_workspace_bytes_dec": (sz, [vp, i32, i32, i32]),
            "mdm_forward_dec": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
            "mdm_workspace_bytes_dec_loop": (sz, [vp, i32, i32, i32, i32]),
            "mdm_sample_loop_dec": (C.c_int, [vp, P(MdmSampleDecParams), vp, vp, sz, vp]),
            "mdm_profile_enable": (C.c_int, [vp, C.c_int]),
            "mdm_profile_read": (C.c_int, [vp, i32, P(C.c_double), P(i64), P(C.c_double)]),
            "mdm_profile_reset": (C.c_int, [vp]),
            "mdm_set_option": (C.c_int, [vp, i32, i32]),
            "mdm_get_option": (C.c_int, [vp, i32, P(i32)]),
            "mdm_set_time_add": (C.c_int, [vp, vp, i32]),
        }
        probe_sig = {
            "mdm_debug_set": (C.c_int, [C.c_int, C.c_int]),
            "mdm_debug_get": (C.c_int, [C.c_int, P(C.c_double)]),
            "mdm_linear_f16f6_scratch_bytes": (sz, [i32, i32, i32]),
            "mdm_linear_f16f6": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, sz, vp]),
            "mdm_probe_in_proj": (C.c_int, [vp, vp, vp, vp, i32, i32, i32, vp, vp]),
        }
        self.has_probes = hasattr(lib, "mdm_debug_set")   # libmdm_hip_probe.so / the emulator build
        if self.has_probes:
            sig.update(probe_sig)
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if lib.mdm_abi_version() != ABI_VERSION:
            raise MdmError(f"{path}: ABI version {lib.mdm_abi_version()} != {ABI_VERSION}")
        self.build_info = dict(kv.split("=", 1) for kv in lib.mdm_build_info().decode().split(";"))
        # A GPU library built WITH packed fp32 VALU math (hipcc's SLP vectorizer) returns rare wrong DiP samples whenever another
        # LDS-using kernel shares a CU with its small GEMM (include/mdm_hip.h CONCURRENCY; profiles/r03g_dip_groups.md): refuse
        # it at load time instead of trusting whoever built it.  (The CPU emulator of tests/emu has no such hazard.)
        if self.build_info.get("emu") != "1" and self.build_info.get("slp") != "off" \
                and os.environ.get("MDM_ALLOW_SLP_BUILD") != "1":
            raise MdmError(f"{path} was built without -fno-slp-vectorize -DMDM_NO_SLP=1 (mdm_build_info: "
                           f"{lib.mdm_build_info().decode()}): rebuild with `python __graft_entry__.py`, or set "
                           f"MDM_ALLOW_SLP_BUILD=1 for an A/B experiment on an otherwise idle device")

    def check(self, rc, what):
        if rc != MDM_OK:
            msg = self.lib.mdm_last_error()
            raise MdmError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")

    def __getattr__(self, name):
        return getattr(self.lib, name)


_LIB = None


def load_native():
    """The one and only way the product path reaches its kernels; raises MdmError if the .so is absent."""
    global _LIB
    if _LIB is None:
        _LIB = MdmLib(os.environ.get("MDM_HIP_LIB", LIB_PATH))
    return _LIB


_PROBE_LIB = None


def load_probe():
    """libmdm_hip_probe.so (include/mdm_hip_probe.h): experiments and probe-only tests; never used by the seams."""
    global _PROBE_LIB
    if _PROBE_LIB is None:
        _PROBE_LIB = MdmLib(os.environ.get("MDM_HIP_PROBE_LIB", PROBE_LIB_PATH))
        if not _PROBE_LIB.has_probes:
            raise MdmError(f"{_PROBE_LIB.path} was not built with -DMDM_PROBES")
    return _PROBE_LIB

"""TEST INFRASTRUCTURE: build + load the CPU emulation of libmdm_hip (see hip_emu.h)."""
import fcntl
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SO = os.environ.get("MDM_EMU_SO", os.path.join(ROOT, "build", "libmdm_emu.so"))
SRC_DIR = os.path.join(ROOT, "motion-diffusion-model_amd", "csrc")
_lib = None


def _stale():
    if not os.path.isfile(SO):
        return True
    t = os.path.getmtime(SO)
    srcs = [os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR) if f.endswith((".h", ".hip"))]
    srcs += [os.path.join(HERE, "hip_emu.h"), os.path.join(ROOT, "include", "mdm_hip.h"),
             os.path.join(ROOT, "lab", "csrc_probe", "gemm_f16f6.h")]
    return any(os.path.getmtime(s) > t for s in srcs)


def emu():
    global _lib
    if _lib is None:
        os.makedirs(os.path.dirname(SO), exist_ok=True)
        with open(SO + ".lock", "w") as lk:      # several pytest-xdist workers may arrive here at once: one builds, the rest wait
            fcntl.flock(lk, fcntl.LOCK_EX)
            if _stale():
                subprocess.check_call([os.path.join(HERE, "build_emu.sh")], stdout=subprocess.DEVNULL)
        import mdm_amd._native as nat
        _lib = nat.MdmLib(SO)
    return _lib


def ptr(a):
    """Host pointer of a C-contiguous numpy array (plays the role of a device pointer in emulation)."""
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)

"""The emulator's LATE mode (tests/emu/hip_emu.h: MDM_EMU_LATE=1) withholds every asynchronous operation -- LDS-DMA pieces, untracked
register loads, untracked fragment reads -- until the counted wait that covers it, i.e. it is the adversarial half of the async model: a
too lenient `vmcnt` / `lgkmcnt` count shows up as a stale operand.  The mode is latched per process, so the round-5 kernels (the
(sequence, head) attention blocks in all three modes, the whole-block cross-attention kernel, the attention kernel's DIRECT form with
carried items, the masked plane route) are re-run here in a child interpreter with the variable set; the rest of the CPU suite runs
EARLY.  (Earlier rounds ran LATE by hand: profiles/r03a_pipe_emulator.md.)"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

SELECTION = ("(test_emulated_dip_fused_self_attention_block and 3-5-12) or (test_emulated_attention_direct_output_is_bit_identical and "
             "9-lengths1) or (test_emulated_dip_decoder_forward and False-f16x3) or (test_emulated_sequences_longer_than_224_tokens and f16x3-225)")
# one small case per kernel (~2 minutes of emulator time): the self-attention block (both in_proj forms), the cross-attention
# (sequence, head) kernel (the default route of the decoder forward), the DIRECT attention form with carried items; round 6: the
# streaming-softmax attention kernel's LDS-DMA ring (csrc/attention_long.h).  The whole-block
# cross-attention kernel and the larger shapes were run LATE by hand before their first GPU sessions (MDM_EMU_LATE=1 pytest -k fused).


@pytest.mark.slow
def test_round5_kernels_under_the_late_async_model():
    sys.path.insert(0, os.path.join(HERE, "emu"))
    from emu_lib import emu
    emu()                                              # build once, in this process
    env = dict(os.environ, MDM_EMU_LATE="1", MDM_TEST_SERIAL="1", OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    env.pop("PYTEST_XDIST_WORKER", None)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_emu_path.py"), "-x", "-q", "-p", "no:cacheprovider",
                        "-k", SELECTION], cwd=ROOT, env=env, capture_output=True, text=True, timeout=3000)
    tail = (r.stdout + r.stderr)[-1500:]
    assert r.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail

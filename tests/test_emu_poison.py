"""Uninitialised-memory check on the CPU emulator: every buffer the engine allocates (workspaces, outputs) is pre-filled with
NaN / 0xFF, so any value the kernels read before writing it -- pad rows, alignment tails, skipped key groups -- shows up as NaN in
the result instead of depending on what the allocator handed out.  (Found in round 2: the folded-LayerNorm GEMM read the
statistics of a tile's pad rows two floats behind the buffer when M is odd and the row has one partial; torch.empty memory is
usually finite on the GPU, so only the emulator ever saw the NaN, and only sometimes.)"""
import os
import sys
import types

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
from emu_lib import emu  # noqa: E402
from helpers import dip, dip_small_state_dict, make_pair, maxabs, orc, small_state_dict, synth_dip_y, synth_y  # noqa: E402


@pytest.fixture()
def poisoned(monkeypatch):
    import mdm_amd._engine as eng_mod
    real = torch

    class PoisonTorch(types.ModuleType):
        def __getattr__(self, k):
            return getattr(real, k)

        def empty(self, *a, **k):
            t = real.empty(*a, **k)
            return t.fill_(0xFF) if t.dtype == real.uint8 else (t.fill_(float("nan")) if t.is_floating_point() else t)

        def empty_like(self, x, **k):
            t = real.empty_like(x, **k)
            return t.fill_(float("nan")) if t.is_floating_point() else t

    monkeypatch.setattr(eng_mod, "torch", PoisonTorch("torch"))
    return emu()


@pytest.mark.parametrize("layers,B,T,lengths,guided", [(2, 1, 196, [150], False), (2, 2, 33, [33, 5], True)])
def test_encoder_reads_nothing_it_did_not_write(poisoned, gemm_path, layers, B, T, lengths, guided):
    if guided and gemm_path == "big":
        pytest.skip("the sequence-tile kernel is covered by the T = 196 case (emulator time)")
    sd = small_state_dict(num_layers=layers)
    model, _ = make_pair(sd, 2, "cpu", guided=guided, native_lib=poisoned, precision="f16x3")
    y = synth_y(B, T, seed=2, lengths=lengths)
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(0))
    t = torch.tensor([1, 0][:B])
    got = model(x, t, y=dict(y))
    want = orc.cfg_forward(sd, x, t, y, num_heads=2) if guided else orc.mdm_forward(sd, x, t, y, num_heads=2)
    assert not bool(torch.isnan(got).any())
    assert maxabs(got, want) < 2e-5


def test_dip_window_loop_reads_nothing_it_did_not_write(poisoned):
    B, C, P, steps = 3, 5, 12, 2
    sd = dip_small_state_dict(num_layers=1)
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=poisoned, context_len=C, pred_len=P, mask_frames=True)
    y = synth_dip_y(B, P, C, seed=3, text_lengths=[6, 3, 1], lengths=[12, 7, 9], scale=2.5)
    g = torch.Generator().manual_seed(4)
    seq = [torch.randn(B, 263, 1, P, generator=g) for _ in range(1 + steps)]
    got = diffusion.p_sample_loop(model, (B, 263, 1, P), clip_denoised=False, model_kwargs={"y": dict(y)}, noise_sequence=seq)
    tab = orc.Tables(orc.named_betas("cosine", steps))
    want = dip.dip_sample_loop(sd, tab, (B, 263, 1, P), y, seq[0], seq[1:], context_len=C, cfg=True, num_heads=2, mask_frames=True)
    assert not bool(torch.isnan(got).any())
    assert maxabs(got, want) < 5e-5

"""Regression test of the platform effect written up in profiles/r03g_dip_groups.md (runs on the MI355X box with `-m gpu`; the file
name sorts it LAST): the DiP window loop must stay bit-reproducible while a foreign stream of the same process keeps LDS-heavy
kernels of another library (fused scaled-dot-product attention) resident on the device.  With the library built the default way
(SLP-vectorised packed fp32 VALU math) 43 of 300 such window loops differed; built with -fno-slp-vectorize (what
__graft_entry__.build() does, tests/test_host_logic.py checks the flag) 0 of 360 did.  The encoder path never did (its kernels own a
CU's whole LDS) and is covered here as the control."""
import pytest
import torch

from helpers import make_pair, synth_dip_state_dict, synth_dip_y, to_dev

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _hammer(side, q, n):
    with torch.cuda.stream(side):
        for _ in range(n):
            q = torch.nn.functional.scaled_dot_product_attention(q, q, q)
    return q


def test_dip_window_loop_is_reproducible_beside_a_foreign_attention_stream():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    B = 8
    model, diffusion = make_pair(synth_dip_state_dict(seed=0), 10, DEV, guided=True, context_len=20, pred_len=40, precision="f16x3")
    y = to_dev(synth_dip_y(B, 40, 20, seed=2, text_lengths=[4, 11, 25, 8, 1, 17, 9, 30]), DEV)
    run = lambda: diffusion.p_sample_loop(model, (B, 263, 1, 40), clip_denoised=False, model_kwargs={"y": y}, seed=7).cpu()
    ref = run()                                   # idle device
    assert bool(torch.isfinite(ref).all())
    side = torch.cuda.Stream()
    q = torch.randn(16, 8, 256, 64, device=DEV, dtype=torch.float16)
    differing = 0
    for _ in range(24):
        q = _hammer(side, q, 400)
        differing += int(not torch.equal(run(), ref))
        torch.cuda.synchronize()
    print(f"[coresidency] DiP window loops differing beside the attention stream: {differing} of 24")
    assert differing == 0


def test_encoder_forward_is_reproducible_beside_a_foreign_attention_stream():
    """The control: the headline path (one CFG denoiser forward, ragged lengths) beside the same stream -- 0 of 120 forwards ever
    differed, with either build."""
    from helpers import synth_state_dict, synth_y
    B, T = 16, 196
    model, _ = make_pair(synth_state_dict(seed=0), 50, DEV, guided=True, precision="f16x3")
    y = to_dev(synth_y(B, T, seed=3, lengths=[T - (7 * i) % 150 for i in range(B)]), DEV)
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(1)).to(DEV)
    t = torch.full((B,), 25, dtype=torch.long, device=DEV)
    with torch.no_grad():
        ref = model(x, t, y=dict(y)).clone()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        q = torch.randn(16, 8, 256, 64, device=DEV, dtype=torch.float16)
        differing = 0
        for _ in range(10):
            q = _hammer(side, q, 300)
            outs = [model(x, t, y=dict(y)) for _ in range(3)]
            torch.cuda.synchronize()
            differing += sum(int(not torch.equal(o, ref)) for o in outs)
    print(f"[coresidency] encoder forwards differing beside the attention stream: {differing} of 30")
    assert differing == 0

"""Seam B1's other conditions (`y['action']`, `y['target_cond']`) on the CPU lock-step emulator of the kernel sources,
through the product's Python seams, against the oracle (which oracle/make_golden_r6b.py pins to the upstream reference's own run:
tests/golden/PIN_REPORT_r6b.json).  cond_mode='action' (model/mdm.py:224-226, :389-397) and --multi_target_cond
(model/mdm.py:197-199, :399-479; include/mdm_hip.h mdm_set_time_add)."""
import ctypes as C
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))

from emu_lib import emu  # noqa: E402
from helpers import dip, dip_small_state_dict, make_pair, maxabs, orc, small_state_dict, synth_dip_y, synth_y  # noqa: E402
from oracle.synth import HML_GOAL_JOINT_NAMES as NAMES  # noqa: E402
from oracle.synth import synth_a2m_state_dict, synth_target_params, synth_target_y  # noqa: E402

A2M = dict(dataset="humanact12", num_actions=12)


@pytest.fixture(scope="module")
def lib():
    return emu()


def a2m_y(B, T, seed, lengths, scale=2.5):
    y = synth_y(B, T, seed, lengths=lengths, scale=scale)
    del y["text_embed"]
    y["action"] = torch.randint(0, 12, (B, 1), generator=torch.Generator().manual_seed(seed + 5))
    return y


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_emulated_action_conditioned_forward_and_loop(lib, prec):
    """An action-to-motion checkpoint: 25 joints x 6 rot6d features (150 input features: the row-tile embedding does not apply, the
    fp32-operand one runs), emb = time_emb + action row, the unconditional branch = time_emb alone; bare model and under guidance."""
    B, T, steps = 2, 12, 2
    sd = synth_a2m_state_dict(seed=0, latent_dim=256, num_layers=1)
    y = a2m_y(B, T, seed=3, lengths=[12, 7])
    x = torch.randn(B, 25, 6, T, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([1, 0])
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, precision=prec, **A2M)
    assert model.model.njoints == 25 and model.model.nfeats == 6 and model.model.cond_mode == "action"
    kw = dict(num_heads=2)
    assert maxabs(model.model(x, t, y=dict(y)), orc.mdm_forward(sd, x, t, y, **kw)) < 2e-5
    assert maxabs(model.model(x, t, y={**y, "uncond": True}), orc.mdm_forward(sd, x, t, {**y, "uncond": True}, **kw)) < 2e-5
    assert maxabs(model(x, t, y=dict(y)), orc.cfg_forward(sd, x, t, y, **kw)) < 5e-5
    # the condition matters: another class row, another output
    y2 = {**y, "action": (y["action"] + 1) % 12}
    assert maxabs(model.model(x, t, y=dict(y2)), orc.mdm_forward(sd, x, t, y, **kw)) > 1e-3
    shape = (B, 25, 6, T)
    x_T, noises = orc.make_noise(shape, steps, 11)
    tab = orc.Tables(orc.named_betas("cosine", steps))
    seq = [x_T] + [n.contiguous() for n in noises]
    got = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)}, noise_sequence=seq)
    assert maxabs(got, orc.sample_loop(sd, tab, shape, y, x_T, noises, cfg=True, **kw)) < 5e-5
    got = diffusion.p_sample_loop(model.model, shape, clip_denoised=False, model_kwargs={"y": dict(y)}, noise_sequence=seq)
    assert maxabs(got, orc.sample_loop(sd, tab, shape, y, x_T, noises, cfg=False, **kw)) < 5e-5


@pytest.mark.parametrize("kind,prec", [("single", "f16x3"), ("multi", "f32")])
def test_emulated_target_condition_on_the_encoder(lib, gemm_path, kind, prec):
    """time_emb += embed_target_cond(...) reaches the encoder through the condition token of BOTH guidance branches: forward (cond,
    guided, force-masked target) and the fused loop, on both split-precision GEMM kernels (the token is written by the pose-transpose
    launch there) and in the f32 mode (cond_token_kernel)."""
    if prec == "f32" and gemm_path != "small":
        pytest.skip("the f32 mode has one GEMM kernel")
    B, T, steps = 3, 9, 2
    sd = {**small_state_dict(num_layers=1), **synth_target_params(kind, latent_dim=256)}
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, precision=prec, multi_target_cond=True,
                                 multi_encoder_type=kind)
    y = {**synth_y(B, T, seed=5, lengths=[T, 4, 6]), **synth_target_y(B, seed=5, first=1)}
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(2))
    t = torch.tensor([1, 0, 1])
    kw = dict(num_heads=2, goal_joint_names=NAMES)
    want = orc.mdm_forward(sd, x, t, y, **kw)
    without = orc.mdm_forward(sd, x, t, {**y, "target_uncond": True}, **kw)
    assert maxabs(want, without) > 1e-3                       # the fixture's target moves the output
    assert maxabs(model.model(x, t, y=dict(y)), want) < 2e-5
    assert maxabs(model.model(x, t, y={**y, "target_uncond": True}), without) < 2e-5
    assert maxabs(model(x, t, y=dict(y)), orc.cfg_forward(sd, x, t, y, **kw)) < 5e-5
    if gemm_path == "big":      # (the loop's condition-token launch is the same kernel: one route is enough -- emulator time)
        return
    shape = (B, 263, 1, T)
    x_T, noises = orc.make_noise(shape, steps, 11)
    got = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)},
                                  noise_sequence=[x_T] + [n.contiguous() for n in noises])
    tab = orc.Tables(orc.named_betas("cosine", steps))
    assert maxabs(got, orc.sample_loop(sd, tab, shape, y, x_T, noises, cfg=True, **kw)) < 5e-5


@pytest.mark.parametrize("kind,prec,masked", [("single", "f16x3", True), ("split", "f16x3", False), ("multi", "f32", False)])
def test_emulated_target_conditioned_dip(lib, kind, prec, masked):
    """The target-conditioned DiP (DiP.md:105): the target embedding is part of the timestep embedding, i.e. of EVERY row of the
    decoder's text memory, both branches.  Stand-alone forward (memory = text + time + target built per call), the window loop
    (target folded into the hoisted text part, the step's time row added in the attention kernel) and the step-by-step loop."""
    B, C, P, steps = 2, 5, 12, 2
    sd = {**dip_small_state_dict(num_layers=1), **synth_target_params(kind, latent_dim=256)}
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, context_len=C, pred_len=P, precision=prec,
                                 mask_frames=masked, multi_target_cond=True, multi_encoder_type=kind)
    y = {**synth_dip_y(B, P, C, seed=3, text_lengths=[6, 3], lengths=[12, 7] if masked else None, scale=2.5),
         **synth_target_y(B, seed=7, first=1)}
    x = torch.randn(B, 263, 1, P, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([1, 0])
    kw = dict(context_len=C, num_heads=2, mask_frames=masked, goal_joint_names=NAMES)
    want = dip.dip_forward(sd, x, t, y, **kw)
    assert maxabs(want, dip.dip_forward(sd, x, t, {**y, "target_uncond": True}, **kw)) > 1e-3
    assert maxabs(model.model(x, t, y=dict(y)), want) < 2e-5
    assert maxabs(model.model(x, t, y={**y, "uncond": True}), dip.dip_forward(sd, x, t, {**y, "uncond": True}, **kw)) < 2e-5
    assert maxabs(model(x, t, y=dict(y)), dip.dip_cfg_forward(sd, x, t, y, **kw)) < 5e-5
    g = torch.Generator().manual_seed(8)
    seq = [torch.randn(B, 263, 1, P, generator=g) for _ in range(1 + steps)]
    tab = orc.Tables(orc.named_betas("cosine", steps))
    want = dip.dip_sample_loop(sd, tab, (B, 263, 1, P), y, seq[0], seq[1:], context_len=C, cfg=True, num_heads=2, mask_frames=masked,
                               goal_joint_names=NAMES)
    run = lambda: diffusion.p_sample_loop(model, (B, 263, 1, P), clip_denoised=False, model_kwargs={"y": dict(y)},   # noqa: E731
                                          noise_sequence=seq)
    got = run()
    assert maxabs(got, want) < 5e-5
    diffusion.dip_stepwise = True
    assert maxabs(run(), got) < 2e-5


def test_time_add_binding_is_one_shot_and_checked(lib):
    """include/mdm_hip.h mdm_set_time_add: consumed by the next call (also by a failing one), refused when the batch differs, absent
    afterwards; and a target in `y` without a --multi_target_cond checkpoint is an error of the seam, not a silent no-op."""
    B, T = 2, 9
    sd = {**small_state_dict(num_layers=1), **synth_target_params("single", latent_dim=256)}
    model, _ = make_pair(sd, 2, "cpu", guided=False, native_lib=lib, multi_target_cond=True, multi_encoder_type="single")
    y = synth_y(B, T, seed=5)
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(2))
    t = torch.tensor([1, 0])
    plain = model(x, t, y=dict(y))
    eng = model.engine()
    g = torch.randn(B, 256)
    te, ts = model.text_embedding(y, "cpu"), t.to(torch.int64)
    with_g = eng.forward(x, ts, te, None, 0, time_add=g)
    assert maxabs(with_g, plain) > 1e-4
    assert torch.equal(eng.forward(x, ts, te, None, 0), plain)              # the binding did not survive its call
    g3 = torch.randn(B + 1, 256)
    lib.check(lib.mdm_set_time_add(eng.handle, g3.data_ptr(), B + 1), "mdm_set_time_add")
    with pytest.raises(Exception, match="mdm_set_time_add bound 3 samples"):
        eng.forward(x, ts, te, None, 0)
    assert torch.equal(eng.forward(x, ts, te, None, 0), plain)              # ... and a refused call consumed it
    lib.check(lib.mdm_set_time_add(eng.handle, g.data_ptr(), B), "mdm_set_time_add")
    lib.check(lib.mdm_set_time_add(eng.handle, None, 0), "mdm_set_time_add")   # NULL clears
    assert torch.equal(eng.forward(x, ts, te, None, 0), plain)
    assert lib.mdm_set_time_add(eng.handle, g.data_ptr(), 0) != 0
    with pytest.raises(ValueError, match="time_add must be"):
        eng.forward(x, ts, te, None, 0, time_add=g.double())
    plain_model, _ = make_pair(small_state_dict(num_layers=1), 2, "cpu", guided=False, native_lib=lib)
    with pytest.raises(ValueError, match="multi_target_cond"):
        plain_model(x, t, y={**y, **synth_target_y(B, seed=1)})


def test_target_modules_carry_the_reference_state_dict_keys():
    """load_state_dict of a --multi_target_cond checkpoint (utils/model_util.py:8-15 asserts unexpected_keys == []): every encoder
    flavour exposes exactly the reference's parameter names; the a2m model exposes embed_action and no embed_text."""
    from mdm_amd import model_util
    for kind in ("single", "split", "multi"):
        args = model_util.default_args(arch="trans_dec", text_encoder_type="bert", context_len=20, pred_len=40, layers=1,
                                       multi_target_cond=True, multi_encoder_type=kind)
        model, _ = model_util.create_model_and_diffusion(args)
        have = {k for k in model.state_dict() if k.startswith("embed_target_cond.")}
        assert have == set(synth_target_params(kind)), kind
        assert model.all_goal_joint_names == NAMES
    args = model_util.default_args(layers=1, **A2M)
    model, _ = model_util.create_model_and_diffusion(args)
    keys = set(model.state_dict())
    assert "embed_action.action_embedding" in keys and not any(k.startswith("embed_text.") for k in keys)
    assert model.input_process.poseEmbedding.weight.shape == (512, 150)


@pytest.mark.parametrize("prec,route", [("f16x3", "planes"), ("f32", "skeleton")])
def test_emulated_decoder_with_class_token(lib, engine_options, prec, route):
    """`--emb_trans_dec` (README `humanml-decoder-with-emb-512`; model/mdm.py:245-247, :256-257, :269-270): the timestep embedding leads
    the decoder's tgt sequence, over a one-token CLIP memory.  The library sees it as the context row of a context_len = 1 model whose
    embedded placeholder is overwritten (include/mdm_hip.h MDM_OPT_DEC_TIME_TOKEN): forward with per-sample timesteps (cond / uncond /
    guided, ragged frame masks), the window loop (uniform step timestep) and the step-by-step loop, on the operand-plane route (f16x3)
    and the fp32 skeleton (f32)."""
    from oracle.synth import synth_dip_state_dict
    B, T, steps = 2, 12, 2
    sd = synth_dip_state_dict(seed=0, latent_dim=256, num_layers=2 if route == "planes" else 1, bert_dim=512)
    model, diffusion = make_pair(sd, steps, "cpu", guided=True, native_lib=lib, precision=prec, text_encoder_type="clip",
                                 emb_trans_dec=True, mask_frames=True)
    assert model.model.lead_rows == 1 and model.model.engine().get_option("dec_time_token") == 1
    y = synth_y(B, T, seed=5, lengths=[T, 7])
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(2))
    t = torch.tensor([1, 0])
    kw = dict(context_len=0, num_heads=2, mask_frames=True, emb_trans_dec=True)
    want = dip.dip_forward(sd, x, t, y, **kw)
    assert maxabs(want, dip.dip_forward(sd, x, t, y, context_len=0, num_heads=2, mask_frames=True)) > 1e-2     # the class token matters
    assert maxabs(model.model(x, t, y=dict(y)), want) < 2e-5
    assert maxabs(model.model(x, t, y={**y, "uncond": True}), dip.dip_forward(sd, x, t, {**y, "uncond": True}, **kw)) < 2e-5
    assert maxabs(model(x, t, y=dict(y)), dip.dip_cfg_forward(sd, x, t, y, **kw)) < 5e-5
    shape = (B, 263, 1, T)
    x_T, noises = orc.make_noise(shape, steps, 11)
    seq = [x_T] + [n.contiguous() for n in noises]
    tab = orc.Tables(orc.named_betas("cosine", steps))
    want = dip.dip_sample_loop(sd, tab, shape, y, x_T, noises, cfg=True, **kw)
    run = lambda: diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)}, noise_sequence=seq)   # noqa: E731
    got = run()
    assert maxabs(got, want) < 5e-5
    diffusion.dip_stepwise = True
    assert maxabs(run(), got) < 2e-5


def test_emulated_unconstrained_action_dataset_model(lib):
    """README `humanact12_unconstrained` (`--unconstrained` on an action dataset: cond_mode='no_cond' over the 25 x 6 rot6d features,
    model/mdm.py:227-229): the condition token is the timestep embedding alone; bare model, 2-step loop."""
    B, T, steps = 2, 12, 2
    sd = synth_a2m_state_dict(seed=0, latent_dim=256, num_layers=1)
    del sd["embed_action.action_embedding"]
    model, diffusion = make_pair(sd, steps, "cpu", guided=False, native_lib=lib, unconstrained=True, **A2M)
    assert model.cond_mode == "no_cond" and model.njoints == 25
    y = synth_y(B, T, seed=3, lengths=[12, 7])
    del y["text_embed"]
    x = torch.randn(B, 25, 6, T, generator=torch.Generator().manual_seed(1))
    t = torch.tensor([1, 0])
    assert maxabs(model(x, t, y=dict(y)), orc.mdm_forward(sd, x, t, y, num_heads=2)) < 2e-5
    shape = (B, 25, 6, T)
    x_T, noises = orc.make_noise(shape, steps, 11)
    got = diffusion.p_sample_loop(model, shape, clip_denoised=False, model_kwargs={"y": dict(y)},
                                  noise_sequence=[x_T] + [n.contiguous() for n in noises])
    want = orc.sample_loop(sd, orc.Tables(orc.named_betas("cosine", steps)), shape, y, x_T, noises, cfg=False, num_heads=2)
    assert maxabs(got, want) < 5e-5

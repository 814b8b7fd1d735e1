"""The oracle restatement (oracle/mdm_oracle.py) against fixtures produced by the UPSTREAM REFERENCE
itself (oracle/make_golden.py, run in the build container).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import mdm_oracle as orc
from oracle.synth import synth_state_dict, synth_y

TOL = 2e-5   # fp32 reorder floor of two CPU implementations over a 50-step CFG trajectory is ~5e-6


@pytest.fixture(scope="module")
def sd():
    return synth_state_dict(seed=0)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"))


def test_pin_report_is_tight(golden_dir):
    rep = json.load(open(os.path.join(golden_dir, "PIN_REPORT.json")))
    assert rep["noise_stream_identical"] is True
    assert rep["schedule_maxabs_50"] < 1e-12 and rep["schedule_maxabs_1000"] < 1e-10
    for name, rec in rep["cases"].items():
        for k, v in rec.items():
            if k != "ref_absmax":
                assert v < TOL, (name, k, v)
    assert rep["recover_B3_T196"]["relative"] < 2e-6       # post-sampling transform (oracle/motion_oracle.py)


@pytest.mark.parametrize("steps", [50, 1000])
def test_schedule_tables(golden_dir, steps):
    g = _load(golden_dir, f"schedule_cosine_{steps}")
    tab = orc.Tables(orc.named_betas("cosine", steps))
    for nm in ("betas", "alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
               "posterior_mean_coef1", "posterior_mean_coef2", "sqrt_recip_alphas_cumprod",
               "sqrt_recipm1_alphas_cumprod"):
        np.testing.assert_allclose(getattr(tab, nm), g[nm], rtol=0, atol=1e-9)
    assert list(g["timestep_map"]) == list(range(steps))
    assert tab.posterior_mean_coef1[0] == 1.0 and tab.posterior_mean_coef2[0] == 0.0


def test_forward_golden(golden_dir, sd):
    g = _load(golden_dir, "fwd_B3_T196")
    B, T = 3, 196
    y = synth_y(B, T, seed=int(g["y_seed"]), lengths=list(g["lengths"]))
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    t = torch.from_numpy(g["t"])
    oc = orc.mdm_forward(sd, x, t, y)
    ou = orc.mdm_forward(sd, x, t, {**y, "uncond": True})
    og = orc.cfg_forward(sd, x, t, y)
    assert np.abs(oc.numpy() - g["out_cond"]).max() < TOL
    assert np.abs(ou.numpy() - g["out_uncond"]).max() < TOL
    assert np.abs(og.numpy() - g["out_cfg"]).max() < TOL
    gm = _load(golden_dir, "fwd_nomask_B3_T196")
    onm = orc.mdm_forward(sd, x, t, y, mask_frames=False)
    assert np.abs(onm.numpy() - gm["out_cond"]).max() < TOL
    # the key-padding mask matters for the padded samples, and only for them
    d = np.abs(onm.numpy() - g["out_cond"]).reshape(B, -1).max(1)
    assert d[0] < TOL and d[1] > 1e-3 and d[2] > 1e-3


LOOPS = ["loop50_nocfg_B2_T64", "ddim50_B2_T64", "ddim50_eta1_B2_T64", "inpaint50_B2_T64", "skip20_init_B2_T64"]


def _run_loop_case(g, sd, dtype=torch.float32):
    steps, B, T, seed = int(g["steps"]), int(g["B"]), int(g["T"]), int(g["seed"])
    skip = int(g["skip"])
    shape = (B, 263, 1, T)
    y = synth_y(B, T, seed=seed + 1000, lengths=list(g["lengths"]), scale=float(g["scale"]))
    gi = torch.Generator().manual_seed(seed + 2000)
    init_image = torch.randn(*shape, generator=gi) if bool(g["init"]) else None
    if bool(g["inpaint"]):
        m = torch.zeros(shape, dtype=torch.bool)
        m[:, :4, :, :] = True
        m[..., : T // 4] = True
        y["inpainting_mask"] = m
        y["inpainted_motion"] = torch.randn(*shape, generator=gi)
    tab = orc.Tables(orc.named_betas("cosine", steps))
    x_T, noises = orc.make_noise(shape, steps - skip, seed)
    return orc.sample_loop(sd, tab, shape, y, x_T, noises, cfg=bool(g["cfg"]), ddim=bool(g["ddim"]),
                           eta=float(g["eta"]), skip_timesteps=skip, init_image=init_image, return_all=True,
                           dtype=dtype)


@pytest.mark.parametrize("name", LOOPS)
def test_loop_golden_small(golden_dir, sd, name):
    g = _load(golden_dir, name)
    final, _ = _run_loop_case(g, sd)
    assert np.abs(final.numpy() - g["final"]).max() < TOL


def test_inpainting_reproduces_fixed_region(golden_dir, sd):
    g = _load(golden_dir, "inpaint50_B2_T64")
    # at t=0 coef1=1, coef2=0, no noise => the returned sample equals the blended x0
    B, T, seed = 2, 64, int(g["seed"])
    gi = torch.Generator().manual_seed(seed + 2000)
    motion = torch.randn(B, 263, 1, T, generator=gi)
    assert np.array_equal(g["final"][:, :4], motion.numpy()[:, :4])
    assert np.array_equal(g["final"][..., : T // 4], motion.numpy()[..., : T // 4])


@pytest.mark.slow
def test_loop_golden_full_T196(golden_dir, sd):
    g = _load(golden_dir, "loop50_B2_T196")
    final, traj = _run_loop_case(g, sd)
    assert np.abs(final.numpy() - g["final"]).max() < TOL
    for k in g["dump_steps"]:
        assert np.abs(traj[int(k)].numpy() - g[f"dump{int(k)}"]).max() < TOL


@pytest.mark.slow
def test_loop_golden_1000_steps(golden_dir, sd):
    g = _load(golden_dir, "loop1000_B1_T32")
    final, _ = _run_loop_case(g, sd)
    assert np.abs(final.numpy() - g["final"]).max() < TOL


def test_motion_oracle_matches_reference_golden(golden_dir):
    """oracle/motion_oracle.py (post-sampling transform) against the upstream functions' output."""
    from oracle import motion_oracle as mo
    from oracle.make_golden_motion import motion_inputs
    g = np.load(os.path.join(golden_dir, "recover_B3_T196.npz"))
    sample, mean, std = motion_inputs(int(g["B"]), int(g["T"]), int(g["seed"]))
    got = mo.recover_from_ric(sample.numpy(), mean.numpy(), std.numpy(), int(g["joints"]))
    assert got.shape == g["out"].shape == (3, 22, 3, 196)
    # positions integrate 196 frames of root velocity (|out| up to ~1e2 here): fp32-rounding-level RELATIVE agreement
    assert float(np.abs(got - g["out"]).max()) < 2e-6 * float(np.abs(g["out"]).max())


# ---- DiP: oracle/dip_oracle.py against the upstream reference's own outputs (oracle/make_golden_dip.py)
DIP_TOL_FWD, DIP_TOL_AR = 2e-5, 1e-4    # CFG scale 7.5 over 3 windows x 10 steps amplifies the fp32 reorder floor


def test_dip_pin_report_is_tight(golden_dir):
    rep = json.load(open(os.path.join(golden_dir, "PIN_REPORT.json")))["dip"]
    assert max(rep["fwd_B3"][k] for k in ("cond", "uncond", "cfg")) < DIP_TOL_FWD
    assert rep["fwd_masked_B3"]["cond"] < DIP_TOL_FWD
    assert rep["ar10_B2_F100"]["final"] < DIP_TOL_AR
    for name in ("dip_dynamic_text_B2_P3", "dip_dynamic_text_B4_P2"):      # round 6: --dynamic_text_path
        assert rep[name]["final"] < DIP_TOL_AR and rep[name]["vs_first_prompt_everywhere"] > 1.0


def test_dip_forward_golden(golden_dir):
    from oracle import dip_oracle as dip
    from oracle.synth import synth_dip_state_dict, synth_dip_y
    sdd = synth_dip_state_dict(seed=0)
    for name, masked in (("dip_fwd_B3", False), ("dip_fwd_masked_B3", True)):
        g = _load(golden_dir, name)
        B = 3
        y = synth_dip_y(B, 40, 20, seed=int(g["y_seed"]), text_lengths=list(g["text_lengths"]),
                        lengths=list(g["lengths"]) if masked else None)
        x = torch.randn(B, 263, 1, 40, generator=torch.Generator().manual_seed(int(g["x_seed"])))
        t = torch.from_numpy(g["t"])
        kw = dict(context_len=20, mask_frames=masked)
        assert np.abs(dip.dip_forward(sdd, x, t, y, **kw).numpy() - g["out_cond"]).max() < DIP_TOL_FWD
        if not masked:
            assert np.abs(dip.dip_forward(sdd, x, t, {**y, "uncond": True}, **kw).numpy() - g["out_uncond"]).max() < DIP_TOL_FWD
            assert np.abs(dip.dip_cfg_forward(sdd, x, t, y, **kw).numpy() - g["out_cfg"]).max() < DIP_TOL_FWD
    # the frames mask matters for the padded samples only, the text mask for every sample with pad tokens
    gm, g0 = _load(golden_dir, "dip_fwd_masked_B3"), _load(golden_dir, "dip_fwd_B3")
    d = np.abs(gm["out_cond"] - g0["out_cond"]).reshape(3, -1).max(1)
    assert d[0] == 0.0 and d[1] > 1e-3 and d[2] > 1e-3


def test_dip_autoregressive_golden(golden_dir):
    from oracle import dip_oracle as dip
    from oracle.synth import synth_dip_state_dict, synth_dip_y
    g = _load(golden_dir, "dip_ar10_B2_F100")
    steps, B, frames, seed = int(g["steps"]), int(g["B"]), int(g["frames"]), int(g["seed"])
    sdd = synth_dip_state_dict(seed=0)
    y = synth_dip_y(B, 40, 20, seed=int(g["y_seed"]), text_lengths=list(g["text_lengths"]), scale=float(g["scale"]))
    tab = orc.Tables(orc.named_betas("cosine", steps))
    chunks = dip.make_noise_chunks((B, 263, 1, 40), steps, seed, 3)
    out = dip.autoregressive_sample(sdd, tab, (B, 263, 1, frames), y, chunks, context_len=20, pred_len=40,
                                    required_frames=frames, cfg=True)
    assert out.shape == (B, 263, 1, frames)
    assert np.abs(out.numpy() - g["final"]).max() < DIP_TOL_AR


@pytest.mark.parametrize("name", ["dip_dynamic_text_B2_P3", "dip_dynamic_text_B4_P2"])
def test_dip_dynamic_text_golden(golden_dir, name):
    """`--dynamic_text_path` (sample/generate.py:63-65, :134-142): the reference's own AutoRegressiveSampler + p_sample_loop run
    with a functional BERT stand-in (oracle/synth.py synth_bert) -- window i runs on prompt i of every sample, re-encoded per window
    (diffusion/gaussian_diffusion.py:633-635) -- against the oracle's restatement of that.  B != Ntok and B == Ntok."""
    from oracle import dip_oracle as dip
    from oracle.synth import synth_bert_encode_text, synth_dip_dynamic_y, synth_dip_state_dict
    g = _load(golden_dir, name)
    steps, B, frames, seed = int(g["steps"]), int(g["B"]), int(g["frames"]), int(g["seed"])
    prompts = [str(p) for p in g["prompts"]]
    assert frames == 40 * len(prompts)
    sdd = synth_dip_state_dict(seed=0)
    y = synth_dip_dynamic_y(B, 40, 20, seed=int(g["y_seed"]), prompts=prompts, scale=float(g["scale"]))
    assert y["text_embed"][0].shape[:3] == (B, max(len(p.split()) for p in prompts) + 2, len(prompts))
    tab = orc.Tables(orc.named_betas("cosine", steps))
    chunks = dip.make_noise_chunks((B, 263, 1, 40), steps, seed, len(prompts))
    out = dip.autoregressive_sample(sdd, tab, (B, 263, 1, frames), y, chunks, context_len=20, pred_len=40,
                                    required_frames=frames, cfg=True, encode_text=synth_bert_encode_text)
    assert np.abs(out.numpy() - g["final"]).max() < DIP_TOL_AR


def test_long_sequence_goldens(golden_dir):
    """Round 6: the reference's own forward and 10-step guided loop at T = 400 (401 tokens; oracle/make_golden_r6.py) -- the oracle is
    pinned beyond the 196 frames of every earlier fixture, where csrc/attention_long.h takes over on the device."""
    rep = json.load(open(os.path.join(golden_dir, "PIN_REPORT_r6.json")))["cases"]
    assert max(rep["fwd_B2_T400"][k] for k in ("cond", "uncond", "cfg")) < 1e-5 and rep["loop10_B2_T400"]["final"] < 2e-5
    sd = synth_state_dict(0)
    g = _load(golden_dir, "fwd_B2_T400")
    B, T = 2, 400
    y = synth_y(B, T, seed=int(g["y_seed"]), lengths=list(g["lengths"]))
    x = torch.randn(B, 263, 1, T, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    t = torch.from_numpy(g["t"])
    assert np.abs(orc.mdm_forward(sd, x, t, y).numpy() - g["out_cond"]).max() < 1e-5
    assert np.abs(orc.cfg_forward(sd, x, t, y).numpy() - g["out_cfg"]).max() < 2e-5


def test_condition_goldens(golden_dir):
    """Seam B1's other conditions: the restatement against the reference's own outputs for cond_mode='action' and --multi_target_cond
    (oracle/make_golden_r6b.py; PIN_REPORT_r6b.json) -- forward of every flavour here, the loops through the pin report."""
    from oracle import dip_oracle as dip
    from oracle.synth import (HML_GOAL_JOINT_NAMES, synth_a2m_state_dict, synth_dip_state_dict, synth_dip_y, synth_target_params,
                              synth_target_y)
    rep = json.load(open(os.path.join(golden_dir, "PIN_REPORT_r6b.json")))
    assert len(rep["cases"]) == 12
    for name, rec in rep["cases"].items():
        for k, v in rec.items():
            if k in ("effect_of_target", "vs_no_class_token"):      # (how much the condition under test moves the output)
                assert v > 1e-2, (name, v)
            elif k != "ref_absmax":
                assert v < (5e-5 if "ar10" in name else TOL), (name, k, v)
    # action-to-motion forward
    g = _load(golden_dir, "a2m_fwd_B3_T60")
    sd = synth_a2m_state_dict(seed=0)
    y = synth_y(3, 60, seed=int(g["y_seed"]), lengths=list(g["lengths"]))
    del y["text_embed"]
    y["action"] = torch.from_numpy(g["action"])
    x = torch.randn(3, 25, 6, 60, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    t = torch.from_numpy(g["t"])
    assert np.abs(orc.mdm_forward(sd, x, t, y).numpy() - g["out_cond"]).max() < TOL
    assert np.abs(orc.mdm_forward(sd, x, t, {**y, "uncond": True}).numpy() - g["out_uncond"]).max() < TOL
    assert np.abs(orc.cfg_forward(sd, x, t, y).numpy() - g["out_cfg"]).max() < TOL
    # target-conditioned DiP forward, every encoder flavour
    g = _load(golden_dir, "dip_target_fwd_B4")
    ys = int(g["y_seed"])
    y = {**synth_dip_y(4, 40, 20, seed=ys, text_lengths=list(g["text_lengths"])), **synth_target_y(4, seed=ys)}
    x = torch.randn(4, 263, 1, 40, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    t = torch.from_numpy(g["t"])
    for kind in ("single", "split", "multi"):
        sd = {**synth_dip_state_dict(seed=0), **synth_target_params(kind, seed=0)}
        kw = dict(context_len=20, goal_joint_names=HML_GOAL_JOINT_NAMES)
        assert np.abs(dip.dip_forward(sd, x, t, y, **kw).numpy() - g["out_cond_" + kind]).max() < TOL, kind
        assert np.abs(dip.dip_forward(sd, x, t, {**y, "uncond": True}, **kw).numpy() - g["out_uncond_" + kind]).max() < TOL, kind


def test_full_length_trans_dec_golden(golden_dir):
    """The trans_dec restatement without a prefix (README.md:254's checkpoint shape) against the reference's forward at T = 196."""
    from oracle import dip_oracle as dip
    from oracle.synth import synth_dip_state_dict, synth_dip_y
    g = _load(golden_dir, "transdec_B2_T196")
    sd = synth_dip_state_dict(seed=0)
    y = synth_dip_y(2, 196, 1, seed=int(g["y_seed"]), text_lengths=list(g["text_lengths"]), lengths=list(g["lengths"]), scale=float(g["scale"]))
    y.pop("prefix")
    x = torch.randn(2, 263, 1, 196, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    t = torch.from_numpy(g["t"])
    assert np.abs(dip.dip_forward(sd, x, t, y, context_len=0, mask_frames=True).numpy() - g["out_cond"]).max() < TOL
    assert np.abs(dip.dip_cfg_forward(sd, x, t, y, context_len=0, mask_frames=True).numpy() - g["out_cfg"]).max() < TOL


def test_decoder_with_class_token_golden(golden_dir):
    """`--emb_trans_dec` (README humanml-decoder-with-emb-512): the restatement against the reference's forward."""
    from oracle import dip_oracle as dip
    from oracle.synth import synth_dip_state_dict
    g = _load(golden_dir, "decemb_B3_T60")
    sd = synth_dip_state_dict(seed=0, bert_dim=512)
    y = synth_y(3, 60, seed=int(g["y_seed"]), lengths=list(g["lengths"]))
    x = torch.randn(3, 263, 1, 60, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    t = torch.from_numpy(g["t"])
    kw = dict(context_len=0, mask_frames=True, emb_trans_dec=True)
    assert np.abs(dip.dip_forward(sd, x, t, y, **kw).numpy() - g["out_cond"]).max() < TOL
    assert np.abs(dip.dip_forward(sd, x, t, {**y, "uncond": True}, **kw).numpy() - g["out_uncond"]).max() < TOL
    assert np.abs(dip.dip_cfg_forward(sd, x, t, y, **kw).numpy() - g["out_cfg"]).max() < TOL

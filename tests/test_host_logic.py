"""Host-side logic of the sampler seam (no kernels): schedule folding, respacing, factory / state-dict contract."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import make_pair, model_util, orc, synth_state_dict
from mdm_amd import dist as mdist
from mdm_amd import gaussian_diffusion as gd
from mdm_amd.respace import SpacedDiffusion, space_timesteps


def _diff(steps, small=True, respacing=None):
    return SpacedDiffusion(use_timesteps=space_timesteps(steps, respacing or [steps]),
                           betas=gd.get_named_beta_schedule("cosine", steps), model_mean_type=gd.ModelMeanType.START_X,
                           model_var_type=gd.ModelVarType.FIXED_SMALL if small else gd.ModelVarType.FIXED_LARGE,
                           loss_type=gd.LossType.MSE)


@pytest.mark.parametrize("steps", [50, 1000])
def test_tables_match_reference_golden(golden_dir, steps):
    g = np.load(os.path.join(golden_dir, f"schedule_cosine_{steps}.npz"))
    d = _diff(steps)
    for nm in ("betas", "alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
               "posterior_mean_coef1", "posterior_mean_coef2", "sqrt_recip_alphas_cumprod",
               "sqrt_recipm1_alphas_cumprod"):
        np.testing.assert_allclose(getattr(d, nm), g[nm], rtol=0, atol=1e-9)
    assert d.timestep_map == list(range(steps)) == list(g["timestep_map"])
    a0, at, sg = d.ddpm_coefficients()
    assert a0.dtype == at.dtype == sg.dtype == np.float32
    assert a0[0] == 1.0 and at[0] == 0.0 and sg[0] == 0.0            # final sample == last x0


@pytest.mark.parametrize("ddim,eta", [(False, 0.0), (True, 0.0), (True, 1.0)])
def test_folded_step_equals_oracle_step(ddim, eta):
    """x_prev = a_x0*x0 + a_xt*x_t + sigma*eps with host-folded scalars == the reference's tensor formulas."""
    steps = 50
    d, tab = _diff(steps), orc.Tables(orc.named_betas("cosine", steps))
    a0, at, sg = d.ddim_coefficients(eta) if ddim else d.ddpm_coefficients()
    g = torch.Generator().manual_seed(1)
    x, x0, nz = (torch.randn(3, 263, 1, 16, generator=g) for _ in range(3))
    for i in (49, 30, 1, 0):
        t = torch.full((3,), i, dtype=torch.long)
        # the oracle evaluated in fp64 tensors: the reference's fp32 DDIM form eps = (c x - x0)/d cancels badly where
        # d = sqrt(1/abar - 1) is tiny or huge; the fold is the same algebra done once in fp64 on the host
        xd, x0d, nzd = x.double(), x0.double(), nz.double()
        want = orc.ddim_step(tab, xd, x0d, t, nzd, eta) if ddim else orc.ddpm_step(tab, xd, x0d, t, nzd)
        got = float(a0[i]) * x0d + float(at[i]) * xd + float(sg[i]) * nzd
        err = float((got - want).abs().max())
        # (the reference casts each table entry to fp32 before use -- oracle._coef -- and the DDIM form divides by
        #  d, which amplifies that rounding to ~3e-5 at small t; the DDPM form has no such division)
        assert err < (2e-4 if ddim else 5e-6), (i, err)


def test_fixed_large_variance():
    d = _diff(50, small=False)
    _, _, sg = d.ddpm_coefficients()
    tab = orc.Tables(orc.named_betas("cosine", 50))
    np.testing.assert_allclose(sg[1:], np.exp(0.5 * tab.fixed_large_log_variance.astype(np.float32))[1:], rtol=1e-6)


def test_space_timesteps_and_respaced_betas():
    assert space_timesteps(50, [50]) == set(range(50))
    assert space_timesteps(1000, "ddim50") == set(range(0, 1000, 20))
    assert sorted(space_timesteps(10, [3])) == [0, 4, 9] or len(space_timesteps(10, [3])) == 3
    assert len(space_timesteps(1000, "10,10,30")) == 50
    with pytest.raises(ValueError):
        space_timesteps(10, [20])
    use = space_timesteps(1000, "ddim50")
    d = _diff(1000, respacing="ddim50")
    nb, tmap = orc.respace_betas(orc.named_betas("cosine", 1000), use)
    np.testing.assert_allclose(d.betas, nb, rtol=0, atol=1e-12)
    assert d.timestep_map == tmap and d.num_timesteps == 50


def test_state_dict_contract(golden_dir):
    """Our module must present exactly the reference's state-dict keys/shapes (tests/golden/state_dict_keys.json was
    dumped from the reference constructor) and accept a checkpoint through load_model_wo_clip."""
    ref = json.load(open(os.path.join(golden_dir, "state_dict_keys.json")))
    model, _ = model_util.create_model_and_diffusion(model_util.default_args())
    ours = {k: list(v.shape) for k, v in model.state_dict().items() if not k.startswith("clip_model.")}
    assert ours == {k: list(v) for k, v in ref.items()}
    sd = synth_state_dict(0)
    ck = dict(sd)
    ck["sequence_pos_encoder.pe"] = torch.zeros(5000, 1, 512)                   # present in real checkpoints, dropped
    ck["embed_timestep.sequence_pos_encoder.pe"] = torch.zeros(5000, 1, 512)
    model_util.load_model_wo_clip(model, ck)
    assert torch.equal(model.embed_text.weight, sd["embed_text.weight"])
    assert model.sequence_pos_encoder.pe.abs().sum() > 0                         # table recomputed, not loaded
    with pytest.raises(AssertionError):
        model_util.load_model_wo_clip(model, {**ck, "bogus.weight": torch.zeros(1)})


def test_wrapper_and_scope_errors():
    sd = synth_state_dict(0, num_layers=1)
    model, diffusion = make_pair(sd, 50, "cpu", guided=True)
    # attribute pass-through of the guidance wrapper (utils/sampler_util.py:18-25, :36-38)
    assert model.njoints == 263 and model.nfeats == 1 and model.cond_mode == "text" and model.data_rep == "hml_vec"
    assert model.text_encoder_type == "clip" and model.model.cond_mask_prob == 0.1
    assert model.rot2xyz(torch.ones(1), pose_rep="xyz").item() == 1.0
    with pytest.raises(AssertionError):
        from mdm_amd.cfg_sampler import ClassifierFreeSampleModel
        m0, _ = model_util.create_model_and_diffusion(model_util.default_args(cond_mask_prob=0.0, layers=1))
        ClassifierFreeSampleModel(m0)
    for kw in (dict(arch="trans_dec", emb_trans_dec=True, context_len=20, pred_len=40), dict(arch="gru"),
               dict(context_len=20, pred_len=40), dict(text_encoder_type="bert"), dict(arch="trans_dec", dataset="humanact12")):
        with pytest.raises(NotImplementedError):
            model_util.create_model_and_diffusion(model_util.default_args(**kw))
    # the class-token decoder (README humanml-decoder-with-emb-512) constructs: one lead row, no prefix
    me, _ = model_util.create_model_and_diffusion(model_util.default_args(arch="trans_dec", emb_trans_dec=True, layers=1))
    assert me.lead_rows == 1 and not me.is_prefix_comp and me.MAX_FRAMES == me.MAX_TOKENS - 1
    # DiP configuration (DiP.md): constructs, keeps the reference's state-dict keys
    md, _ = model_util.create_model_and_diffusion(model_util.default_args(
        arch="trans_dec", text_encoder_type="bert", context_len=20, pred_len=40, layers=1))
    assert md.clip_dim == 768 and md.precision == "f16x3" and md.is_prefix_comp
    keys = set(md.state_dict().keys())
    assert "seqTransDecoder.layers.0.multihead_attn.in_proj_weight" in keys and "seqTransDecoder.layers.0.norm3.bias" in keys
    assert md.embed_text.weight.shape == (512, 768)
    with pytest.raises(RuntimeError):
        md.encode_text(["a person walks"])         # DistilBERT is not on the path: the embedding must be cached
    with pytest.raises(NotImplementedError):
        diffusion.p_sample_loop(model, (1, 263, 1, 8), cond_fn=lambda *a: None, model_kwargs={"y": {}})
    with pytest.raises(NotImplementedError):
        diffusion.p_sample_loop(torch.nn.Linear(2, 2), (1, 263, 1, 8))           # foreign model
    with pytest.raises(NotImplementedError):
        diffusion.ddim_sample_loop(model, (1, 263, 1, 8), dump_steps=[1])        # as the reference (:900-901)
    with pytest.raises(NotImplementedError):
        diffusion.training_losses()


def test_shard_bounds_and_y_slicing():
    for B, W in ((1024, 8), (10, 4), (5, 8)):
        cuts = [mdist.shard_bounds(B, r, W) for r in range(W)]
        assert cuts[0][0] == 0 and cuts[-1][1] == B
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(W - 1))
        sizes = [b - a for a, b in cuts]
        assert max(sizes) - min(sizes) <= 1
    y = {"mask": torch.ones(6, 1, 1, 4, dtype=torch.bool), "lengths": torch.arange(6), "scale": torch.arange(6.0),
         "text_embed": torch.arange(6.0).view(1, 6, 1).expand(1, 6, 3), "text": list("abcdef"), "uncond": False}
    s = mdist.shard_y(y, 2, 5)
    assert s["mask"].shape[0] == 3 and s["lengths"].tolist() == [2, 3, 4] and s["text"] == ["c", "d", "e"]
    assert s["text_embed"].shape == (1, 3, 3) and s["text_embed"][0, 0, 0] == 2 and s["uncond"] is False


def test_library_is_built_without_slp_vectorisation():
    """profiles/r03g_dip_groups.md: packed fp32 VALU operations out of hipcc's SLP vectorizer lost their last 16 lanes on the MI355X
    whenever an LDS-heavy workgroup of another kernel shared the CU (DiP path: 100 of 520 window loops); the flag is the cure
    (0 of 520) and must not silently disappear from the build recipe."""
    import __graft_entry__ as ge
    assert "-fno-slp-vectorize" in ge.HIP_FLAGS
    assert "--offload-arch=gfx950" in ge.HIP_FLAGS
    assert "-DMDM_NO_SLP=1" in ge.HIP_FLAGS        # the marker mdm_build_info() reports; the loaders check it


def test_loader_refuses_a_gpu_library_built_with_slp(monkeypatch):
    """nothing at LOAD time checked how the .so was built.  mdm_build_info() (ABI 8) carries the marker and
    MdmLib refuses a non-emulator library without it (MDM_ALLOW_SLP_BUILD=1 opens it for A/B experiments)."""
    import ctypes as C
    from mdm_amd import _native
    lib = _native.MdmLib(_native.LIB_PATH)
    assert lib.build_info["slp"] == "off" and lib.build_info["probes"] == "0" and lib.build_info["emu"] == "0"
    assert _native.MdmLib(_native.PROBE_LIB_PATH).build_info["probes"] == "1"

    real_cdll = C.CDLL

    class Fake:
        """the real library with a different answer from mdm_build_info"""
        def __init__(self, path):
            self._l = real_cdll(path)
            self.mdm_build_info = lambda: b"slp=on;probes=0;emu=0;planes=f16"

        def __getattr__(self, n):
            return getattr(self._l, n)

    monkeypatch.setattr(C, "CDLL", Fake)
    with pytest.raises(_native.MdmError, match="fno-slp-vectorize"):
        _native.MdmLib(_native.LIB_PATH)
    monkeypatch.setenv("MDM_ALLOW_SLP_BUILD", "1")
    assert _native.MdmLib(_native.LIB_PATH).build_info["slp"] == "on"


def test_engine_key_sees_every_kind_of_weight_change():
    """the engine cache key must move after load_state_dict(assign=True), after re-assigning ANY
    nn.Parameter and after `p.data = other` on a non-first parameter -- round 3's quick key (first address + version sum over a
    cached parameter list) saw none of the three and kept serving the old weights."""
    from mdm_amd.mdm import MDM

    class FakeEngine:
        made = 0

        def __init__(self, cfg, lib=None, precision="f16x3", options=None):
            FakeEngine.made += 1

        def bind(self, state, device):
            self.state = {k: v.clone() for k, v in state.items()}

    import mdm_amd.mdm as mm
    orig = mm.Engine
    mm.Engine = FakeEngine
    try:
        model, _ = model_util.create_model_and_diffusion(model_util.default_args(layers=2))
        model.eval()
        e0 = model.engine()
        assert model.engine() is e0 and FakeEngine.made == 1                         # stable while nothing changes
        lin2 = model.seqTransEncoder.layers[1].linear2
        with torch.no_grad():
            lin2.bias.add_(1.0)                                                      # in-place: version bump
        e1 = model.engine()
        assert e1 is not e0 and float(e1.state["seqTransEncoder.layers.1.linear2.bias"][0]) == float(lin2.bias.detach()[0])
        lin2.bias.data = torch.full_like(lin2.bias, 3.0)                            # p.data = other (non-first parameter)
        e2 = model.engine()
        assert e2 is not e1 and float(e2.state["seqTransEncoder.layers.1.linear2.bias"][0]) == 3.0
        lin2.weight = torch.nn.Parameter(torch.zeros_like(lin2.weight))            # re-assigned nn.Parameter
        e3 = model.engine()
        assert e3 is not e2 and float(e3.state["seqTransEncoder.layers.1.linear2.weight"].abs().max()) == 0.0
        sd = {k: torch.full_like(v, 0.5) for k, v in model.state_dict().items()}
        model.load_state_dict(sd, assign=True)                                       # fresh tensors, versions reset
        e4 = model.engine()
        assert e4 is not e3 and float(e4.state["embed_text.bias"][0]) == 0.5
        assert model.engine() is e4
        model.engine_options["small_gemm_max_seqs"] = 0                              # round 5: the handle's options are part of the key
        e5 = model.engine()
        assert e5 is not e4 and model.engine() is e5
    finally:
        mm.Engine = orig

"""Generate tests/golden/*.npz by running the UPSTREAM REFERENCE itself (build container only).

    python oracle/make_golden.py            # writes tests/golden/ + tests/golden/PIN_REPORT.json

Every fixture stores only seeds/config + the reference's outputs; weights and inputs are rebuilt
deterministically from `oracle/synth.py` + `oracle/mdm_oracle.make_noise`.  The report records the
max-abs difference between the reference and the restatement `oracle/mdm_oracle.py` per case -- that
is what "pins" the oracle (the upstream repo has no tests of its own).
"""
import io
import json
import contextlib
import os
import sys
from copy import deepcopy

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_harness as rh          # noqa: E402
from oracle import mdm_oracle as orc          # noqa: E402
from oracle.synth import synth_state_dict, synth_y  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def ref_model(sd, **over):
    m = rh.build_reference_model(seed=123, **over)
    missing, unexpected = m.load_state_dict(sd, strict=False)    # utils/model_util.py:13-15 contract
    assert len(unexpected) == 0, unexpected
    assert all(k.startswith("clip_model.") or "sequence_pos_encoder" in k for k in missing), missing
    m.eval()   # NB: the reference's MDM.train() override returns None (mdm.py:291-293)
    return m


def maxabs(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    report = {"torch": torch.__version__, "cases": {}}
    sd = synth_state_dict(seed=0)
    model = ref_model(sd)
    cfgm = rh.reference_cfg(model)

    # ---- case 1: single forward, mixed lengths / timesteps, cond + uncond  (mdm.py:189-283)
    B, T = 3, 196
    y = synth_y(B, T, seed=11, lengths=[196, 120, 57])
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, 263, 1, T, generator=g)
    t = torch.tensor([49, 25, 0])
    with torch.no_grad():
        oc = model(x, t, deepcopy(y))
        yu = deepcopy(y); yu["uncond"] = True
        ou = model(x, t, yu)
        og = cfgm(x, t, deepcopy(y))
    o_c = orc.mdm_forward(sd, x, t, y)
    o_u = orc.mdm_forward(sd, x, t, {**y, "uncond": True})
    o_g = orc.cfg_forward(sd, x, t, y)
    report["cases"]["fwd_B3_T196"] = {"cond": maxabs(oc, o_c), "uncond": maxabs(ou, o_u), "cfg": maxabs(og, o_g),
                                      "ref_absmax": float(oc.abs().max())}
    np.savez_compressed(os.path.join(OUT, "fwd_B3_T196.npz"), x_seed=5, y_seed=11, lengths=[196, 120, 57],
                        t=t.numpy(), out_cond=oc.numpy(), out_uncond=ou.numpy(), out_cfg=og.numpy())

    # ---- case 1b: mask_frames=False checkpoint flavour (mdm.py:243)
    model_nomask = ref_model(sd, mask_frames=False)
    with torch.no_grad():
        onm = model_nomask(x, t, deepcopy(y))
    report["cases"]["fwd_nomask"] = {"cond": maxabs(onm, orc.mdm_forward(sd, x, t, y, mask_frames=False))}
    np.savez_compressed(os.path.join(OUT, "fwd_nomask_B3_T196.npz"), x_seed=5, y_seed=11, lengths=[196, 120, 57],
                        t=t.numpy(), out_cond=onm.numpy())

    def run_loop(name, steps, B, T, lengths, seed, *, cfg=True, ddim=False, eta=0.0, skip=0, init=False,
                 inpaint=False, dump=(), scale=2.5):
        diff = rh.build_reference_diffusion(steps=steps)
        tab = orc.Tables(orc.named_betas("cosine", steps))
        y = synth_y(B, T, seed=seed + 1000, lengths=lengths, scale=scale)
        shape = (B, 263, 1, T)
        extra = {}
        gi = torch.Generator().manual_seed(seed + 2000)
        init_image = torch.randn(*shape, generator=gi) if init else None
        if inpaint:
            m = torch.zeros(shape, dtype=torch.bool)
            m[:, :4, :, :] = True                  # root features fixed
            m[..., : T // 4] = True                # first quarter of the frames fixed
            y["inpainting_mask"] = m
            y["inpainted_motion"] = torch.randn(*shape, generator=gi)
        mdl = cfgm if cfg else model
        dumps = []
        torch.manual_seed(seed)                    # utils/fixseed.py:6-10 -> global generator
        yy = deepcopy(y)
        with torch.no_grad():
            if ddim:
                ref = diff.ddim_sample_loop(mdl, shape, clip_denoised=False, model_kwargs={"y": yy}, eta=eta,
                                            skip_timesteps=skip, init_image=init_image)
                dumps = []
            else:
                kw = dict(clip_denoised=False, model_kwargs={"y": yy}, skip_timesteps=skip, init_image=init_image,
                          progress=False, noise=None, const_noise=False)
                if dump:
                    dumps = diff.p_sample_loop(mdl, shape, dump_steps=list(dump), **kw)
                    torch.manual_seed(seed)
                    yy = deepcopy(y)
                ref = diff.p_sample_loop(mdl, shape, dump_steps=None, **kw)
        x_T, noises = orc.make_noise(shape, steps - skip, seed)
        mine, traj = orc.sample_loop(sd, tab, shape, y, x_T, noises, cfg=cfg, ddim=ddim, eta=eta,
                                     skip_timesteps=skip, init_image=init_image, return_all=True)
        rec = {"final": maxabs(ref, mine), "ref_absmax": float(ref.abs().max())}
        save = dict(steps=steps, B=B, T=T, lengths=np.asarray(lengths), seed=seed, cfg=cfg, ddim=ddim, eta=eta,
                    skip=skip, init=init, inpaint=inpaint, scale=scale, final=ref.numpy())
        for k, dmp in zip(dump, dumps):
            rec[f"dump{k}"] = maxabs(dmp, traj[k])
            save[f"dump{k}"] = dmp.numpy()
        save["dump_steps"] = np.asarray(list(dump), dtype=np.int64)
        report["cases"][name] = rec
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **save)
        print(name, rec, flush=True)

    run_loop("loop50_B2_T196", 50, 2, 196, [196, 150], seed=3, dump=(0, 24))
    run_loop("loop50_nocfg_B2_T64", 50, 2, 64, [64, 40], seed=4, cfg=False)
    run_loop("ddim50_B2_T64", 50, 2, 64, [64, 33], seed=5, ddim=True, eta=0.0)
    run_loop("ddim50_eta1_B2_T64", 50, 2, 64, [64, 64], seed=6, ddim=True, eta=1.0)
    run_loop("inpaint50_B2_T64", 50, 2, 64, [64, 50], seed=7, inpaint=True)
    run_loop("skip20_init_B2_T64", 50, 2, 64, [64, 64], seed=8, skip=20, init=True)
    run_loop("loop1000_B1_T32", 1000, 1, 32, [32], seed=9)

    # ---- noise-stream identity: global-generator draws == oracle.make_noise
    torch.manual_seed(77)
    a = torch.randn(2, 263, 1, 8); b = torch.randn_like(a)
    c = torch.randn_like(torch.empty(8, 2, 263, 1).permute(1, 2, 3, 0))   # strides of a reference `sample`
    x_T, ns = orc.make_noise((2, 263, 1, 8), 2, 77)
    report["noise_stream_identical"] = bool(torch.equal(a, x_T) and torch.equal(b, ns[0]) and torch.equal(c, ns[1]))

    # ---- schedule tables vs the reference's own (gaussian_diffusion.py:166-202)
    for steps in (50, 1000):
        diff = rh.build_reference_diffusion(steps=steps)
        tab = orc.Tables(orc.named_betas("cosine", steps))
        worst = 0.0
        for nm in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
                   "sqrt_recipm1_alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                   "posterior_mean_coef1", "posterior_mean_coef2"):
            worst = max(worst, float(np.abs(getattr(diff, nm) - getattr(tab, nm)).max()))
        report[f"schedule_maxabs_{steps}"] = worst
        np.savez_compressed(os.path.join(OUT, f"schedule_cosine_{steps}.npz"),
                            **{nm: getattr(diff, nm) for nm in
                               ("betas", "alphas_cumprod", "posterior_variance", "posterior_log_variance_clipped",
                                "posterior_mean_coef1", "posterior_mean_coef2", "sqrt_recip_alphas_cumprod",
                                "sqrt_recipm1_alphas_cumprod")},
                            timestep_map=np.asarray(diff.timestep_map))

    # ---- state-dict key contract
    keys = {k: list(v.shape) for k, v in rh.reference_state_dict(model).items()}
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)

    with open(os.path.join(OUT, "PIN_REPORT.json"), "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print(json.dumps(report, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

"""Generate tests/golden/dip_*.npz by running the UPSTREAM REFERENCE's DiP path itself (build container only):
`MDM(arch='trans_dec', text_encoder_type='bert', context_len=20, pred_len=40)` under `ClassifierFreeSampleModel`,
`SpacedDiffusion.p_sample_loop` driven by `AutoRegressiveSampler` (utils/sampler_util.py:41-81).  DistilBERT itself is
not on the path (its output is cached in y['text_embed'], sample/generate.py:147-160) and is replaced by a stub that
returns the synthetic embedding.  Records the restatement's (oracle/dip_oracle.py) deviation in PIN_REPORT.json["dip"].

    python oracle/make_golden_dip.py
"""
import contextlib
import io
import json
import os
import sys
import types
from copy import deepcopy

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_harness as rh          # noqa: E402
from oracle import mdm_oracle as orc          # noqa: E402
from oracle import dip_oracle as dip          # noqa: E402
from oracle.synth import synth_bert, synth_bert_encode_text, synth_dip_dynamic_y, synth_dip_state_dict, synth_dip_y  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CONTEXT, PRED = 20, 40


class _StubBert(nn.Module):
    """model/BERT/BERT_encoder.py:12-32 returns (last_hidden_state [B, Ntok, 768], attention_mask bool)."""
    preset = None

    def forward(self, texts):
        # `preset`: the cached synthetic embedding of the static-text cases; else the functional stand-in (a prompt -> its own seeded
        # token rows, oracle/synth.py synth_bert) that the dynamic-text cases need, where upstream re-encodes per window
        return self.preset if self.preset is not None else synth_bert(texts)


def ref_dip_model(sd, **over):
    rh._install_stubs()
    import model.mdm as ref_mdm
    stub = _StubBert()
    ref_mdm.load_bert = lambda path: stub
    torch.manual_seed(123)
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref_mdm.MDM(**rh.mdm_kwargs(arch="trans_dec", text_encoder_type="bert", context_len=CONTEXT, pred_len=PRED,
                                        mask_frames=False, **over))
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert len(unexpected) == 0, unexpected
    assert all(k.startswith("clip_model.") or "sequence_pos_encoder" in k for k in missing), missing
    m.eval()
    return m, stub


def maxabs(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


DYNAMIC_CASES = {
    # name: (B, prompts, seed).  Token counts (words + 2) 5 / 7 / 4 != B; second case B == Ntok == 4 (a sample-major slice passes
    # every shape check there)
    "dip_dynamic_text_B2_P3": (2, ["a person walks forward", "the person turns around and waves", "sits down"], 41),
    "dip_dynamic_text_B4_P2": (4, ["jumps high", "runs"], 43),
}


def dynamic_text(sd, model, stub, cfgm, rep):
    """`--dynamic_text_path` (sample/generate.py:63-65, :134-142; utils/sampler_util.py:52, :66-71; the re-encode of
    diffusion/gaussian_diffusion.py:633-635): one prompt per 40-frame prediction window, CFG 7.5, run through the reference's own
    AutoRegressiveSampler + p_sample_loop with a functional BERT stand-in."""
    from utils.sampler_util import AutoRegressiveSampler
    steps = 10
    diff = rh.build_reference_diffusion(steps=steps)
    tab = orc.Tables(orc.named_betas("cosine", steps))
    stub.preset = None
    args = types.SimpleNamespace(pred_len=PRED, context_len=CONTEXT, autoregressive_include_prefix=False)
    for name, (B, prompts, seed) in DYNAMIC_CASES.items():
        frames = len(prompts) * PRED                               # generate.py:65
        y = synth_dip_dynamic_y(B, PRED, CONTEXT, seed=seed + 1000, prompts=prompts)
        sampler = AutoRegressiveSampler(args, diff.p_sample_loop, frames)
        shape = (B, 263, 1, frames)
        torch.manual_seed(seed)
        with torch.no_grad():
            ref = sampler.sample(cfgm, shape, clip_denoised=False, model_kwargs={"y": deepcopy(y)}, skip_timesteps=0,
                                 init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False)
        chunks = dip.make_noise_chunks((B, 263, 1, PRED), steps, seed, len(prompts))
        mine = dip.autoregressive_sample(sd, tab, shape, y, chunks, context_len=CONTEXT, pred_len=PRED,
                                         required_frames=frames, cfg=True, encode_text=synth_bert_encode_text)
        # what the semantics is NOT: every window on the first prompt
        y_static = {**y, "text": [prompts[0]] * B, "text_embed": synth_bert_encode_text([prompts[0]] * B)}
        other = dip.autoregressive_sample(sd, tab, shape, y_static, chunks, context_len=CONTEXT, pred_len=PRED,
                                          required_frames=frames, cfg=True)
        rep[name] = {"final": maxabs(ref, mine), "ref_absmax": float(ref.abs().max()),
                     "vs_first_prompt_everywhere": maxabs(ref, other)}
        np.savez_compressed(os.path.join(OUT, name + ".npz"), steps=steps, B=B, frames=frames, seed=seed, y_seed=seed + 1000,
                            prompts=np.array(prompts), scale=7.5, final=ref.numpy())


def main():
    torch.set_num_threads(8)
    rep = {}
    sd = synth_dip_state_dict(seed=0)
    model, stub = ref_dip_model(sd)
    cfgm = rh.reference_cfg(model)
    if "--only-dynamic" in sys.argv:      # the fixtures of round 6 alone (the others regenerate bit-identically, but take minutes)
        dynamic_text(sd, model, stub, cfgm, rep)
        path = os.path.join(OUT, "PIN_REPORT.json")
        report = json.load(open(path))
        report.setdefault("dip", {}).update(rep)
        with open(path, "w") as f:
            json.dump(report, f, indent=1, sort_keys=True)
        print(json.dumps(rep, indent=1, sort_keys=True))
        return

    # ---- single forward: cond / uncond / CFG, ragged text lengths
    B = 3
    y = synth_dip_y(B, PRED, CONTEXT, seed=21, text_lengths=[12, 7, 20])
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, 263, 1, PRED, generator=g)
    t = torch.tensor([9, 4, 0])
    with torch.no_grad():
        oc = model(x, t, deepcopy(y))
        yu = deepcopy(y); yu["uncond"] = True
        ou = model(x, t, yu)
        og = cfgm(x, t, deepcopy(y))
    kw = dict(context_len=CONTEXT)
    rep["fwd_B3"] = {"cond": maxabs(oc, dip.dip_forward(sd, x, t, y, **kw)),
                     "uncond": maxabs(ou, dip.dip_forward(sd, x, t, {**y, "uncond": True}, **kw)),
                     "cfg": maxabs(og, dip.dip_cfg_forward(sd, x, t, y, **kw)), "ref_absmax": float(oc.abs().max())}
    np.savez_compressed(os.path.join(OUT, "dip_fwd_B3.npz"), x_seed=6, y_seed=21, text_lengths=[12, 7, 20], t=t.numpy(),
                        out_cond=oc.numpy(), out_uncond=ou.numpy(), out_cfg=og.numpy())

    # ---- the same with a frames mask (mask_frames=True flavour, mdm.py:242-244)
    model_m, _ = ref_dip_model(sd, )
    model_m.mask_frames = True
    ym = synth_dip_y(B, PRED, CONTEXT, seed=21, text_lengths=[12, 7, 20], lengths=[40, 25, 33])
    with torch.no_grad():
        om = model_m(x, t, deepcopy(ym))
    rep["fwd_masked_B3"] = {"cond": maxabs(om, dip.dip_forward(sd, x, t, ym, mask_frames=True, **kw))}
    np.savez_compressed(os.path.join(OUT, "dip_fwd_masked_B3.npz"), x_seed=6, y_seed=21, text_lengths=[12, 7, 20],
                        lengths=[40, 25, 33], t=t.numpy(), out_cond=om.numpy())

    # ---- autoregressive sampling: 10 diffusion steps per 40-frame window, 100 frames = 3 windows, CFG 7.5
    steps, B, frames, seed = 10, 2, 100, 31
    diff = rh.build_reference_diffusion(steps=steps)
    tab = orc.Tables(orc.named_betas("cosine", steps))
    y = synth_dip_y(B, PRED, CONTEXT, seed=seed + 1000, text_lengths=[9, 15])
    enc, pad = y["text_embed"]
    stub.preset = (enc.permute(1, 0, 2).contiguous(), ~pad)            # what DistilBERT would have returned
    from utils.sampler_util import AutoRegressiveSampler
    args = types.SimpleNamespace(pred_len=PRED, context_len=CONTEXT, autoregressive_include_prefix=False)
    sampler = AutoRegressiveSampler(args, diff.p_sample_loop, frames)
    shape = (B, 263, 1, frames)
    torch.manual_seed(seed)
    with torch.no_grad():
        ref = sampler.sample(cfgm, shape, clip_denoised=False, model_kwargs={"y": deepcopy(y)}, skip_timesteps=0,
                             init_image=None, progress=False, dump_steps=None, noise=None, const_noise=False)
    chunks = dip.make_noise_chunks((B, 263, 1, PRED), steps, seed, 3)
    mine = dip.autoregressive_sample(sd, tab, shape, y, chunks, context_len=CONTEXT, pred_len=PRED,
                                     required_frames=frames, cfg=True)
    rep["ar10_B2_F100"] = {"final": maxabs(ref, mine), "ref_absmax": float(ref.abs().max())}
    np.savez_compressed(os.path.join(OUT, "dip_ar10_B2_F100.npz"), steps=steps, B=B, frames=frames, seed=seed,
                        y_seed=seed + 1000, text_lengths=[9, 15], scale=7.5, final=ref.numpy())

    dynamic_text(sd, model, stub, cfgm, rep)

    keys = {k: list(v.shape) for k, v in rh.reference_state_dict(model).items()}
    with open(os.path.join(OUT, "dip_state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)
    path = os.path.join(OUT, "PIN_REPORT.json")
    report = json.load(open(path))
    report["dip"] = rep
    with open(path, "w") as f:
        json.dump(report, f, indent=1, sort_keys=True)
    print(json.dumps(rep, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()

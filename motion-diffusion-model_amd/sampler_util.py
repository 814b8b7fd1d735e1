"""`utils/sampler_util.py` seam: the live copy of `ClassifierFreeSampleModel` (:10-38) and DiP's
`AutoRegressiveSampler` (:41-81) over the MI355X denoiser.

The autoregressive loop is host control flow around `sample_fn` (one `p_sample_loop` per prediction window); the
windows' arithmetic runs in libmdm_hip.so.  Unlike the reference it does not deep-copy the whole kwargs per window
(utils/sampler_util.py:62): only the `y` dict is shallow-copied, with the new prefix swapped in.

Dynamic text (`--dynamic_text_path`, sample/generate.py:63-65, :134-142: a prompt per prediction window).  Upstream slices the
cached embedding `(text_embed[0][:, :, i], text_embed[1][:, i])` (:69) -- a SAMPLE-major `[B, Ntok, 768]` block, because
generate.py:139 stacked the token-major encoding `[Ntok, P, 768]` of the P prompts behind a new batch axis -- and never reads
that slice: `p_sample_loop` re-encodes `y['text']` whenever the key is present (diffusion/gaussian_diffusion.py:633-635), so what
upstream computes is "window i runs on prompt i of every sample".  This seam keeps a cached embedding when one is present
(gaussian_diffusion.py `_loop`), so the slice it hands over must BE that: the window's prompts in the token-major layout
`bert_encode_text` returns (`[Ntok, B, 768]`, model/mdm.py:185) with their pad mask `[B, Ntok]`.  Pinned against the reference
itself by tests/golden/dip_dynamic_text_*.npz (oracle/make_golden_dip.py; round 5 copied upstream's slice and was wrong).
"""
import torch

from .cfg_sampler import ClassifierFreeSampleModel  # noqa: F401  (same class, both import paths of the reference)


class AutoRegressiveSampler():
    def __init__(self, args, sample_fn, required_frames=196):
        self.sample_fn = sample_fn
        self.args = args
        self.required_frames = required_frames

    def sample(self, model, shape, **kargs):
        pred_len, context_len = self.args.pred_len, self.args.context_len
        n_iterations = (self.required_frames // pred_len) + int(self.required_frames % pred_len > 0)
        samples_buf = []
        y0 = kargs['model_kwargs']['y']
        cur_prefix = y0['prefix'].clone()                     # init with data
        dynamic_text_mode = 'text' in y0 and type(y0['text'][0]) == list
        if getattr(self.args, 'autoregressive_include_prefix', False):
            samples_buf.append(cur_prefix)
        autoregressive_shape = list(shape)
        autoregressive_shape[-1] = pred_len
        for i in range(n_iterations):
            y = dict(y0)
            y['prefix'] = cur_prefix
            if dynamic_text_mode:                             # a prompt per prediction window (:66-71)
                y['text'] = [s[i] for s in y0['text']]
                if getattr(model, 'text_encoder_type', 'clip') != 'bert':
                    raise NotImplementedError('DiP model only supports BERT text encoder at the moment.')
                enc, pad = y0['text_embed']                   # [B, Ntok, P, 768], [B, P, Ntok] (generate.py:139-140)
                if enc.dim() != 4 or pad.dim() != 3 or enc.shape[0] != pad.shape[0] or enc.shape[2] != pad.shape[1]:
                    raise ValueError("dynamic text: y['text_embed'] must be (enc [B, Ntok, P, dim], pad [B, P, Ntok]) as "
                                     f"sample/generate.py:139-140 builds it; got {tuple(enc.shape)}, {tuple(pad.shape)}")
                y['text_embed'] = (enc[:, :, i].permute(1, 0, 2).contiguous(), pad[:, i].contiguous())   # token-major
            cur_kargs = dict(kargs)
            cur_kargs['model_kwargs'] = {**kargs['model_kwargs'], 'y': y}
            sample = self.sample_fn(model, autoregressive_shape, **cur_kargs)
            samples_buf.append(sample[..., -pred_len:].clone())
            cur_prefix = sample[..., -context_len:].clone()
        return torch.cat(samples_buf, dim=-1)[..., :self.required_frames]   # 200 -> 196

"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the DiP path: the `trans_dec` denoiser with a
DistilBERT text memory, prefix completion and the autoregressive sampler.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; the product path never does.

Pinned against the upstream reference by oracle/make_golden_dip.py (tests/golden/dip_*.npz, PIN_REPORT.json "dip").
Scope: emb_policy='add', emb_trans_dec=False, text_encoder_type='bert' -- the configuration DiP ships with -- and, since
round 4, text_encoder_type='clip' (one memory token per sample, mdm.py:261-262; pinned by oracle/make_golden_r4.py).

Reference lines restated (paths relative to the upstream tree):
  model/mdm.py:85-93      nn.TransformerDecoder of post-norm nn.TransformerDecoderLayer (gelu)
  model/mdm.py:203-206    prefix completion: x = cat(prefix, x); mask gets `context_len` leading ones
  model/mdm.py:208-220    text memory: embed_text(mask_cond(enc_text)) + time_emb  (emb_policy 'add')
  model/mdm.py:241-247    frames mask (no leading step column for trans_dec without emb_trans_dec)
  model/mdm.py:255-270    tgt = pos_enc(InputProcess(x)); decoder(tgt, memory, memory_key_padding_mask, tgt_key_padding_mask);
                          with emb_trans_dec (the `humanml-decoder-with-emb-512` checkpoint) the timestep embedding leads tgt
  model/mdm.py:277-283    keep the completed suffix, OutputProcess
  utils/sampler_util.py:41-81   AutoRegressiveSampler.sample
torch: nn.TransformerDecoderLayer.forward (norm_first=False): x = norm1(x + sa(x)); x = norm2(x + mha(x, mem)); x = norm3(x + ff(x))
"""
import math

import torch
import torch.nn.functional as F

from oracle import mdm_oracle as orc


def _mha(sd, p, xq, xkv, key_pad, num_heads, dtype):
    """F.multi_head_attention_forward with packed in_proj weights: xq [N, Sq, d], xkv [N, Sk, d], key_pad [N, Sk] bool."""
    N, Sq, d = xq.shape
    Sk = xkv.shape[1]
    hd = d // num_heads
    w, b = sd[p + "in_proj_weight"].to(dtype), sd[p + "in_proj_bias"].to(dtype)
    q = F.linear(xq, w[:d], b[:d]).view(N, Sq, num_heads, hd).transpose(1, 2)
    k = F.linear(xkv, w[d:2 * d], b[d:2 * d]).view(N, Sk, num_heads, hd).transpose(1, 2)
    v = F.linear(xkv, w[2 * d:], b[2 * d:]).view(N, Sk, num_heads, hd).transpose(1, 2)
    sc = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if key_pad is not None:
        sc = sc.masked_fill(key_pad[:, None, None, :], float("-inf"))
    a = (torch.softmax(sc, dim=-1) @ v).transpose(1, 2).reshape(N, Sq, d)
    return F.linear(a, sd[p + "out_proj.weight"].to(dtype), sd[p + "out_proj.bias"].to(dtype))


def decoder_layer(sd, i, x, mem, tgt_pad, mem_pad, num_heads, dtype):
    """One post-norm nn.TransformerDecoderLayer (mdm.py:87-91).  x [N, S, d], mem [N, Sm, d] (batch first here)."""
    p = f"seqTransDecoder.layers.{i}."
    d = x.shape[-1]

    def ln(v, n):
        return F.layer_norm(v, (d,), sd[p + n + ".weight"].to(dtype), sd[p + n + ".bias"].to(dtype), 1e-5)

    x = ln(x + _mha(sd, p + "self_attn.", x, x, tgt_pad, num_heads, dtype), "norm1")
    x = ln(x + _mha(sd, p + "multihead_attn.", x, mem, mem_pad, num_heads, dtype), "norm2")
    h = orc._lin(x, sd, p + "linear1", dtype)
    h = 0.5 * h * (1.0 + torch.erf(h * (1.0 / math.sqrt(2.0))))
    return ln(x + orc._lin(h, sd, p + "linear2", dtype), "norm3")


def dip_forward(sd, x, timesteps, y, *, context_len, num_heads=4, mask_frames=False, pe=None, dtype=torch.float32,
                goal_joint_names=(), emb_trans_dec=False):
    """MDM.forward for arch='trans_dec', text_encoder_type='bert' (mdm.py:189-283).

    x [B, J, 1, pred_len]; y: 'prefix' [B, J, 1, context_len], 'text_embed' = (enc [Ntok, B, 768], pad [B, Ntok] bool,
    True = no token), 'mask' [B, 1, 1, pred_len] bool, 'uncond'?  ->  [B, J, 1, pred_len]."""
    B, J, Fe, _ = x.shape
    d = sd["input_process.poseEmbedding.weight"].shape[0]
    L = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("seqTransDecoder.layers."))
    if pe is None:
        pe = orc.positional_table(5000, d, dtype)
    x = x.to(dtype)
    time_emb = orc.timestep_embedding(sd, timesteps, pe, dtype)                   # [B, d]           mdm.py:195
    tgt = orc.target_embedding(sd, y, goal_joint_names, dtype)                    # mdm.py:197-199 (the target-conditioned DiP, DiP.md:105)
    if tgt is not None:
        time_emb = time_emb + tgt
    mask = y["mask"]
    if context_len > 0:                                                           # mdm.py:203-206
        x = torch.cat([y["prefix"].to(dtype), x], dim=-1)
        mask = torch.cat([torch.ones(B, 1, 1, context_len, dtype=mask.dtype), mask], dim=-1)
    S = x.shape[-1]
    if isinstance(y["text_embed"], tuple):
        enc, text_pad = y["text_embed"]
    else:      # text_encoder_type='clip' (mdm.py:261-262): ONE memory token per sample [1, B, clip_dim], no memory pad mask
        enc, text_pad = y["text_embed"], torch.zeros(B, y["text_embed"].shape[0], dtype=torch.bool)
    enc = enc.to(dtype)
    if text_pad.shape[0] == 1 and B > 1:                                          # mdm.py:215-216
        text_pad = text_pad.repeat_interleave(B, dim=0)
    if y.get("uncond", False):                                                    # mdm.py:155-156
        enc = torch.zeros_like(enc)
    mem = orc._lin(enc, sd, "embed_text", dtype) + time_emb[None]                 # [Ntok, B, d]     mdm.py:217-219
    mem = mem.transpose(0, 1)                                                     # batch first
    h = orc._lin(x.permute(0, 3, 1, 2).reshape(B, S, J * Fe), sd, "input_process.poseEmbedding", dtype)
    tgt_pad = None
    if mask_frames and mask.shape[-1] > 1:                                        # mdm.py:242-244
        tgt_pad = ~mask[..., :S].reshape(B, S)
        if emb_trans_dec:                                                         # mdm.py:245-247: the step token is never masked
            tgt_pad = torch.cat([torch.zeros(B, 1, dtype=torch.bool), tgt_pad], dim=1)
    if emb_trans_dec:                                                             # mdm.py:256-257: the TIMESTEP embedding (not emb) leads tgt
        h = torch.cat([time_emb[:, None, :], h], dim=1)
    seq = h + pe[:h.shape[1]][None]                                               # mdm.py:259-260
    for i in range(L):
        seq = decoder_layer(sd, i, seq, mem, tgt_pad, text_pad, num_heads, dtype)  # mdm.py:265
    if emb_trans_dec:
        seq = seq[:, 1:]                                                          # mdm.py:269-270
    seq = seq[:, context_len:]                                                    # mdm.py:278-279
    out = orc._lin(seq, sd, "output_process.poseFinal", dtype)
    return out.reshape(B, S - context_len, J, Fe).permute(0, 2, 3, 1).contiguous()


def dip_cfg_forward(sd, x, timesteps, y, **kw):
    """ClassifierFreeSampleModel.forward (utils/sampler_util.py:27-34) over the decoder."""
    oc = dip_forward(sd, x, timesteps, y, **kw)
    ou = dip_forward(sd, x, timesteps, {**y, "uncond": True}, **kw)
    return ou + y["scale"].to(oc.dtype).view(-1, 1, 1, 1) * (oc - ou)


def dip_sample_loop(sd, tab, shape, y, x_T, step_noise, *, context_len, cfg=True, num_heads=4, mask_frames=False,
                    dtype=torch.float32, goal_joint_names=(), emb_trans_dec=False):
    """p_sample_loop (gaussian_diffusion.py:591-727) of one prediction window with an injected noise sequence."""
    B = shape[0]
    pe = orc.positional_table(5000, sd["input_process.poseEmbedding.weight"].shape[0], dtype)
    fwd = dip_cfg_forward if cfg else dip_forward
    img = x_T.to(dtype)
    for k, i in enumerate(range(tab.num_timesteps)[::-1]):
        t = torch.full((B,), i, dtype=torch.long)
        x0 = fwd(sd, img, t, y, context_len=context_len, num_heads=num_heads, mask_frames=mask_frames, pe=pe, dtype=dtype,
                 goal_joint_names=goal_joint_names, emb_trans_dec=emb_trans_dec)
        img = orc.ddpm_step(tab, img, x0, t, step_noise[k].to(dtype))
    return img


def autoregressive_sample(sd, tab, shape, y, noise_chunks, *, context_len, pred_len, required_frames,
                          include_prefix=False, encode_text=None, **kw):
    """AutoRegressiveSampler.sample (utils/sampler_util.py:47-81).  noise_chunks[i] = (x_T, [eps_k]).

    Static text: every window runs on y['text_embed'].  Dynamic text (`--dynamic_text_path`; y['text'][b] is the LIST of prompts
    of sample b, utils/sampler_util.py:52): window i keeps prompt i of every sample (:66-68) and -- the line that decides what
    upstream computes -- p_sample_loop RE-ENCODES y['text'] whenever that key is present (diffusion/gaussian_diffusion.py:633-635),
    so the sampler's slice of the cached embedding (:69) is overwritten before any forward reads it: window i runs on
    `encode_text([prompt i of sample b for b])`."""
    n_iter = required_frames // pred_len + int(required_frames % pred_len > 0)
    cur_prefix = y["prefix"].clone()
    buf = [cur_prefix] if include_prefix else []
    ar_shape = list(shape)
    ar_shape[-1] = pred_len
    dynamic = "text" in y and type(y["text"][0]) == list
    for i in range(n_iter):
        x_T, eps = noise_chunks[i]
        yi = {**y, "prefix": cur_prefix}
        if dynamic:
            yi["text"] = [s[i] for s in y["text"]]
            yi["text_embed"] = encode_text(yi["text"])
        sample = dip_sample_loop(sd, tab, ar_shape, yi, x_T, eps, context_len=context_len, **kw)
        buf.append(sample[..., -pred_len:].clone())
        cur_prefix = sample[..., -context_len:].clone()
    return torch.cat(buf, dim=-1)[..., :required_frames]


def make_noise_chunks(shape, steps, seed, n_chunks):
    """Consecutive draws of the global generator under torch.manual_seed(seed): per chunk one randn(*shape), then
    `steps` randn_like (the first contiguous, the rest in the permuted strides of the previous sample) -- see
    mdm_oracle.make_noise for the layout subtlety."""
    B, J, Fe, T = shape
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n_chunks):
        x_T = torch.randn(*shape, generator=g)
        eps = []
        for k in range(steps):
            if k == 0:
                eps.append(torch.randn(*shape, generator=g))
            else:
                eps.append(torch.empty_strided(shape, (J * Fe, Fe, 1, B * J * Fe)).normal_(generator=g))
        out.append((x_T, eps))
    return out

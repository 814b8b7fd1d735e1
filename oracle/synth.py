"""TEST/BENCH INFRASTRUCTURE: deterministic synthetic inputs for the MDM hot path.

No checkpoints or datasets are available offline, so weights are synthetic.
They are produced by *this* generator (CPU torch.Generator, bit-reproducible across
machines with the same torch build) instead of the reference constructor, so that the
GPU box -- where /root/reference does not exist -- can rebuild exactly the weights the
golden fixtures were made with.  Key names / shapes are the reference's.
"""
import math

import torch


def synth_state_dict(seed=0, latent_dim=512, ff_size=1024, num_layers=8, input_feats=263, clip_dim=512):
    g = torch.Generator().manual_seed(seed)

    def U(shape, bound):
        return (torch.rand(*shape, generator=g) * 2.0 - 1.0) * bound

    def linear(prefix, out_f, in_f, sd):
        b = 1.0 / math.sqrt(in_f)          # nn.Linear default: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(in))
        sd[prefix + ".weight"] = U((out_f, in_f), b)
        sd[prefix + ".bias"] = U((out_f,), b)

    d = latent_dim
    sd = {}
    linear("input_process.poseEmbedding", d, input_feats, sd)
    for i in range(num_layers):
        p = f"seqTransEncoder.layers.{i}."
        sd[p + "self_attn.in_proj_weight"] = U((3 * d, d), math.sqrt(6.0 / (4 * d)))   # xavier_uniform
        sd[p + "self_attn.in_proj_bias"] = U((3 * d,), 0.02)
        linear(p + "self_attn.out_proj", d, d, sd)
        linear(p + "linear1", ff_size, d, sd)
        linear(p + "linear2", d, ff_size, sd)
        for n in ("norm1", "norm2"):     # non-trivial affine so gamma/beta handling is exercised
            sd[p + n + ".weight"] = 1.0 + 0.1 * torch.randn(d, generator=g)
            sd[p + n + ".bias"] = 0.05 * torch.randn(d, generator=g)
    linear("embed_timestep.time_embed.0", d, d, sd)
    linear("embed_timestep.time_embed.2", d, d, sd)
    linear("embed_text", d, clip_dim, sd)
    linear("output_process.poseFinal", input_feats, d, sd)
    return sd


def synth_y(B, T, seed, lengths=None, scale=2.5, clip_dim=512):
    """`model_kwargs['y']` as sample/generate.py:107-132 builds it, with a random cached text embedding."""
    g = torch.Generator().manual_seed(seed)
    if lengths is None:
        lengths = [T] * B
    lengths = torch.as_tensor(lengths, dtype=torch.long)
    mask = (torch.arange(T)[None, :] < lengths[:, None]).view(B, 1, 1, T)
    return {"mask": mask, "lengths": lengths,
            "text_embed": torch.randn(1, B, clip_dim, generator=g),
            "scale": torch.ones(B) * scale}


def synth_state_dict_hostile(seed=0, latent_dim=512, ff_size=1024, num_layers=8, input_feats=263, clip_dim=512):
    """"Trained-like" hostile statistics on top of `synth_state_dict` (same keys / shapes): what random-init weights never
    show and checkpoints do.  Deterministic in `seed`.
      * LayerNorm gamma log-normal over [0.05, 8] with a handful of channels pinned to both ends;
      * a few OUTLIER CHANNELS: LayerNorm beta and the biases of the linears feeding the residual stream 50-300x the rest;
      * every weight matrix has a few rows 10x the rest (outlier output features);
      * the text embedding is scaled 20x by `synth_y_hostile`.
    The residual stream then carries channels hundreds of times larger than the typical one, LayerNorm row means far from
    zero relative to the bulk, and GEMM operands with a wide dynamic range inside every 32-wide block of k."""
    sd = synth_state_dict(seed, latent_dim, ff_size, num_layers, input_feats, clip_dim)
    g = torch.Generator().manual_seed(seed + 7919)
    d = latent_dim
    out_ch = torch.randperm(d, generator=g)[:4]            # the model-wide outlier channels of the residual stream

    def rows10(w, n=3):
        idx = torch.randperm(w.shape[0], generator=g)[:n]
        w[idx] *= 10.0

    def outlier_vec(v, idx, lo=50.0, hi=300.0):
        typ = v.abs().mean().clamp_min(1e-3)
        mag = lo + (hi - lo) * torch.rand(len(idx), generator=g)
        sign = torch.where(torch.rand(len(idx), generator=g) < 0.5, -1.0, 1.0)
        v[idx] = sign * mag * typ

    for i in range(num_layers):
        p = f"seqTransEncoder.layers.{i}."
        for n in ("norm1", "norm2"):
            gam = torch.exp(0.7 * torch.randn(d, generator=g)).clamp(0.05, 8.0)
            ends = torch.randperm(d, generator=g)[:12]
            gam[ends[:6]] = 8.0
            gam[ends[6:]] = 0.05
            sd[p + n + ".weight"] = gam
            outlier_vec(sd[p + n + ".bias"], out_ch)
        for n in ("self_attn.in_proj_weight", "self_attn.out_proj.weight", "linear1.weight", "linear2.weight"):
            rows10(sd[p + n])
        outlier_vec(sd[p + "self_attn.out_proj.bias"], out_ch)
        outlier_vec(sd[p + "linear2.bias"], out_ch)
        outlier_vec(sd[p + "linear1.bias"], torch.randperm(ff_size, generator=g)[:4])
    rows10(sd["input_process.poseEmbedding.weight"])
    outlier_vec(sd["input_process.poseEmbedding.bias"], out_ch)
    rows10(sd["embed_text.weight"])
    rows10(sd["output_process.poseFinal.weight"])
    return sd


def synth_y_hostile(B, T, seed, lengths=None, scale=2.5, clip_dim=512):
    """`synth_y` with the cached text embedding scaled 20x (CLIP features are not unit-variance)."""
    y = synth_y(B, T, seed, lengths, scale, clip_dim)
    y["text_embed"] = y["text_embed"] * 20.0
    return y


def synth_dip_state_dict(seed=0, latent_dim=512, ff_size=1024, num_layers=8, input_feats=263, bert_dim=768):
    """DiP (`--arch trans_dec --text_encoder_type bert`): reference key names / shapes of model/mdm.py:85-93, :127."""
    g = torch.Generator().manual_seed(seed)

    def U(shape, bound):
        return (torch.rand(*shape, generator=g) * 2.0 - 1.0) * bound

    def linear(prefix, out_f, in_f, sd):
        b = 1.0 / math.sqrt(in_f)
        sd[prefix + ".weight"] = U((out_f, in_f), b)
        sd[prefix + ".bias"] = U((out_f,), b)

    d = latent_dim
    sd = {}
    linear("input_process.poseEmbedding", d, input_feats, sd)
    for i in range(num_layers):
        p = f"seqTransDecoder.layers.{i}."
        for a in ("self_attn", "multihead_attn"):
            sd[p + a + ".in_proj_weight"] = U((3 * d, d), math.sqrt(6.0 / (4 * d)))
            sd[p + a + ".in_proj_bias"] = U((3 * d,), 0.02)
            linear(p + a + ".out_proj", d, d, sd)
        linear(p + "linear1", ff_size, d, sd)
        linear(p + "linear2", d, ff_size, sd)
        for n in ("norm1", "norm2", "norm3"):
            sd[p + n + ".weight"] = 1.0 + 0.1 * torch.randn(d, generator=g)
            sd[p + n + ".bias"] = 0.05 * torch.randn(d, generator=g)
    linear("embed_timestep.time_embed.0", d, d, sd)
    linear("embed_timestep.time_embed.2", d, d, sd)
    linear("embed_text", d, bert_dim, sd)
    linear("output_process.poseFinal", input_feats, d, sd)
    return sd


def synth_dip_y(B, pred_len, context_len, seed, text_lengths, scale=7.5, lengths=None, bert_dim=768, njoints=263):
    """`model_kwargs['y']` of the autoregressive DiP call (sample/generate.py:107-160 with --autoregressive): a cached
    DistilBERT embedding (last_hidden_state [Ntok, B, 768], pad mask [B, Ntok] True = no token) and a data prefix."""
    g = torch.Generator().manual_seed(seed)
    ntok = int(max(text_lengths))
    tl = torch.as_tensor(text_lengths, dtype=torch.long)
    if lengths is None:
        lengths = [pred_len] * B
    lengths = torch.as_tensor(lengths, dtype=torch.long)
    return {"mask": (torch.arange(pred_len)[None, :] < lengths[:, None]).view(B, 1, 1, pred_len), "lengths": lengths,
            "text": ["synthetic prompt"] * B,
            "text_embed": (torch.randn(ntok, B, bert_dim, generator=g), torch.arange(ntok)[None, :] >= tl[:, None]),
            "prefix": torch.randn(B, njoints, 1, context_len, generator=g),
            "scale": torch.ones(B) * scale}


def synth_bert(texts, bert_dim=768, junk=5.0):
    """Functional stand-in for model/BERT/BERT_encoder.py:26-32 `BERT.forward(texts)`: -> (last_hidden_state [len(texts), Ntok, 768],
    attention_mask bool [len(texts), Ntok]).  A prompt maps to a block of (words + 2) token rows -- [CLS] w1 .. wn [SEP], as the
    tokenizer counts them -- drawn from a generator seeded by the CRC of the string, so the SAME prompt gives the SAME rows in
    whatever batch it is encoded (what a deterministic encoder with an attention mask does); the batch is right-padded to its
    longest prompt (`padding=True`) and the pad rows are filled with junk (a real encoder leaves arbitrary values there: they must be
    masked, never read).  Used to pin `--dynamic_text_path` (sample/generate.py:63-65, :134-142), where upstream re-encodes one
    prompt per prediction window."""
    import zlib
    blocks = []
    for s in texts:
        g = torch.Generator().manual_seed(zlib.crc32(s.encode("utf-8")))
        blocks.append(torch.randn(len(s.split()) + 2, bert_dim, generator=g))
    ntok = max(b.shape[0] for b in blocks)
    out = torch.empty(len(texts), ntok, bert_dim)
    mask = torch.zeros(len(texts), ntok, dtype=torch.bool)
    gj = torch.Generator().manual_seed(ntok * 7919 + len(texts))
    for i, b in enumerate(blocks):
        out[i, :b.shape[0]] = b
        out[i, b.shape[0]:] = junk * torch.randn(ntok - b.shape[0], bert_dim, generator=gj)
        mask[i, :b.shape[0]] = True
    return out, mask


def synth_bert_encode_text(texts):
    """model/mdm.py:180-187 `bert_encode_text` over `synth_bert`: -> (enc [Ntok, B, 768], pad [B, Ntok], True = no token)."""
    out, mask = synth_bert(texts)
    return out.permute(1, 0, 2), ~mask


def synth_dip_dynamic_y(B, pred_len, context_len, seed, prompts, scale=7.5, njoints=263):
    """`model_kwargs['y']` as sample/generate.py:130-142 leaves it under `--dynamic_text_path`: the P prompts of the file encoded
    ONCE as a batch, then `y['text']` = the prompt list per sample and `y['text_embed']` = (enc [B, Ntok, P, 768] -- the token-major
    [Ntok, P, 768] encoding repeated per sample --, pad [B, P, Ntok])."""
    g = torch.Generator().manual_seed(seed)
    enc, pad = synth_bert_encode_text(list(prompts))                  # [Ntok, P, 768], [P, Ntok]
    return {"mask": torch.ones(B, 1, 1, pred_len, dtype=torch.bool), "lengths": torch.full((B,), pred_len, dtype=torch.long),
            "text": [list(prompts)] * B,
            "text_embed": (enc.unsqueeze(0).repeat(B, 1, 1, 1), pad.unsqueeze(0).repeat(B, 1, 1)),
            "prefix": torch.randn(B, njoints, 1, context_len, generator=g),
            "scale": torch.ones(B) * scale}


# ---- conditions beside the text (round 6, second part): target locations (model/mdm.py:197-199, :399-479) and action classes (:224-226, :389-397)
HML_GOAL_JOINT_NAMES = ["pelvis", "left_foot", "right_foot", "left_wrist", "right_wrist", "head"]     # utils/model_util.py:45


def synth_target_params(kind, seed=0, latent_dim=512, names=HML_GOAL_JOINT_NAMES, num_layers=1):
    """The `embed_target_cond.*` entries of a `--multi_target_cond` checkpoint (`--multi_encoder_type` single | split | multi:
    EmbedTargetLocSingle / Split / Multi, model/mdm.py:399-479) with the reference's key names and shapes."""
    g = torch.Generator().manual_seed(seed + 77)
    ext = list(names) + ["traj", "heading"]

    def linear(prefix, out_f, in_f, sd):
        b = 1.0 / math.sqrt(in_f)
        sd[prefix + ".weight"] = (torch.rand(out_f, in_f, generator=g) * 2.0 - 1.0) * b
        sd[prefix + ".bias"] = (torch.rand(out_f, generator=g) * 2.0 - 1.0) * b

    sd, p = {}, "embed_target_cond."
    if kind == "single":
        linear(p + "mlp.0", latent_dim, 4 * len(ext), sd)
        for i in range(num_layers):
            linear(p + f"mlp.{2 * i + 2}", latent_dim, latent_dim, sd)
    elif kind == "split":
        w = latent_dim // len(ext)
        for j in range(len(ext)):
            linear(p + f"mini_mlps.{j}.0", w, 4, sd)
            for i in range(num_layers):
                linear(p + f"mini_mlps.{j}.{2 * i + 2}", w, w, sd)
    elif kind == "multi":
        for n in ext:
            linear(p + f"target_loc_emb.{n}.0", latent_dim, 3, sd)
            linear(p + f"target_loc_emb.{n}.2", latent_dim, latent_dim, sd)
        sd[p + "target_all_loc_emb.weights"] = 0.5 + torch.rand(len(ext), generator=g)     # (the reference inits randn: sum may be ~0)
    else:
        raise ValueError(kind)
    return sd


def synth_target_y(B, seed, names=HML_GOAL_JOINT_NAMES, first=0):
    """y['target_cond'] [B, n_ext, 3], y['target_joint_names'] (per sample: an array of goal joints, from none to all, 'traj' among
    them), y['is_heading'] [B] -- what data_loaders/humanml/data/dataset.py hands the model under --multi_target_cond."""
    import numpy as np
    g = torch.Generator().manual_seed(seed + 99)
    ext = list(names) + ["traj"]
    sel = []
    for b in range(B):
        k = [0, 1, len(ext), 3][(b + first) % 4]
        perm = torch.randperm(len(ext), generator=g)[:k].tolist()
        sel.append(np.array([ext[i] for i in sorted(perm)], dtype=str))
    return {"target_cond": torch.randn(B, len(names) + 2, 3, generator=g),
            "target_joint_names": sel, "is_heading": [bool((b // 2) % 2 == 0) for b in range(B)]}


def synth_a2m_state_dict(seed=0, num_actions=12, latent_dim=512, num_layers=8, input_feats=150):
    """An action-to-motion checkpoint (`cond_mode='action'`: humanact12 / uestc, utils/model_util.py:33-37, :47-52): the encoder of
    synth_state_dict over njoints * nfeats = 25 * 6 rot6d features, `embed_action.action_embedding` [num_actions, d]
    (model/mdm.py:389-397) in place of embed_text."""
    sd = synth_state_dict(seed=seed, latent_dim=latent_dim, num_layers=num_layers, input_feats=input_feats)
    del sd["embed_text.weight"], sd["embed_text.bias"]
    g = torch.Generator().manual_seed(seed + 55)
    sd["embed_action.action_embedding"] = torch.randn(num_actions, latent_dim, generator=g)
    return sd

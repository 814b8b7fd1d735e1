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

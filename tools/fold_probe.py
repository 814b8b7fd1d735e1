"""CPU probe (no GPU): the PRODUCTION arithmetic of the split-precision encoder, restated in torch, on the oracle's guided loop.

tools/precision_probe.py only swaps the GEMM operand decomposition.  This probe restates what csrc/gemm_x3.h + attention_x3.h
actually compute in the default mode:
  * the residual stream lives as hi + lo 16-bit planes of the PRE-norm sums (so every stored activation is rounded to hi + lo);
  * LayerNorm is folded into its consumers: W.LN(x) + b = rstd * (W'.x - mean * colsum(W')) + (b + W.beta), W' = W * gamma, with
    mean / rstd rebuilt from per-256-column partial sums -- `naive`: (sum x, sum x^2) and var = E[x^2] - mean^2 in fp32 (round 1);
    `chan`: (sum x, sum (x - tile mean)^2) merged with Chan's formula (this round);
  * QK^T and PV on split operands (three products), softmax in fp32;
  * dt = bf16 (round 1, "bf16x3") or fp16 ("f16x3").
and reports the max-abs trajectory error against the fp32 and fp64 oracles on the standard and on the hostile synthetic weights.
Usage: python tools/fold_probe.py [--hostile] [B] [steps]"""
import math
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mdm_oracle as orc  # noqa: E402
from oracle.synth import synth_state_dict, synth_state_dict_hostile, synth_y, synth_y_hostile  # noqa: E402


def split(x, dt):
    hi = x.to(dt).float()
    lo = (x - hi).to(dt).float()
    return hi, lo


def planes(x, dt):
    if dt is None:
        return x
    hi, lo = split(x, dt)
    return hi + lo


WSCALE = 1.0    # --wscale=S: weights are split as hi/lo of w * S (a power of two) and the product is scaled back


def mm3(a, w, dt):
    """a [.., K] x w [N, K] -> three-product split GEMM, fp32 accumulate"""
    ah, al = split(a, dt)
    wh, wl = split(w * WSCALE, dt)
    return (ah @ wh.t() + (ah @ wl.t() + al @ wh.t())) * (1.0 / WSCALE)


def row_stats(x, mode, tile=256):
    """(mean, rstd) of the rows of x [.., D] from per-tile partials, in fp32"""
    D = x.shape[-1]
    xt = x.reshape(*x.shape[:-1], D // tile, tile)
    s1 = xt.sum(-1)
    if mode == "naive":
        s2 = (xt * xt).sum(-1)
        mean = s1.sum(-1) / D
        var = (s2.sum(-1) / D - mean * mean).clamp_min(0.0)
    else:   # chan: per-wave (32 columns) centred partials merged, then per-tile, then across tiles
        mt = s1 / tile
        m2 = ((xt - mt[..., None]) ** 2).sum(-1)
        mean = s1.sum(-1) / D
        var = (m2.sum(-1) + (tile * (mt - mean[..., None]) ** 2).sum(-1)) / D
    return mean[..., None], torch.rsqrt(var + 1e-5)[..., None]


class X3Model:
    def __init__(self, sd, dt, stats, heads=4, ablate=""):
        self.sd, self.dt, self.stats, self.H = sd, dt, stats, heads
        # what-if switches (component attribution): "p" stored activations keep fp32 (no plane rounding), "a" exact fp32
        # attention, "f" LayerNorm applied to the operand instead of folded into the weights
        self.ab = ablate
        self.L = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("seqTransEncoder.layers."))
        self.pe = orc.positional_table(5000, sd["input_process.poseEmbedding.weight"].shape[0])

    def pl(self, x):
        return x if "p" in self.ab else planes(x, self.dt)

    def lin_ln(self, x, w, b, g, be):
        """W.LN(x) + b"""
        mean, rstd = row_stats(x, self.stats)
        if "f" in self.ab:
            return mm3((x - mean) * rstd * g + be, w, self.dt) + b, mean, rstd
        wf, cs, bf = self.fold(w, b, g, be)
        return rstd * (mm3(x, wf, self.dt) - mean * cs) + bf, mean, rstd

    def fold(self, w, b, g, be):
        wf = w * g[None, :]
        return wf, wf.double().sum(1).float(), b + (w.double() @ be.double()).float()

    def attention(self, qkv, key_pad):
        N, S, d3 = qkv.shape
        d = d3 // 3
        hd = d // self.H
        q, k, v = qkv.split(d, dim=-1)
        if "a" in self.ab:
            q, k, v = (t.view(N, S, self.H, hd).transpose(1, 2) for t in (q, k, v))
            sc = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
            if key_pad is not None:
                sc = sc.masked_fill(key_pad[:, None, None, :], float("-inf"))
            return self.pl((torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(N, S, d))
        q = planes(q * (1.0 / math.sqrt(hd)), self.dt).view(N, S, self.H, hd).transpose(1, 2)
        k = planes(k, self.dt).view(N, S, self.H, hd).transpose(1, 2)
        v = planes(v, self.dt).view(N, S, self.H, hd).transpose(1, 2)
        qh, ql = split(q, self.dt)
        kh, kl = split(k, self.dt)
        sc = qh @ kh.transpose(-1, -2) + (qh @ kl.transpose(-1, -2) + ql @ kh.transpose(-1, -2))
        if key_pad is not None:
            sc = sc.masked_fill(key_pad[:, None, None, :], float("-inf"))
        p = torch.exp(sc - sc.amax(-1, keepdim=True))
        inv = 1.0 / p.sum(-1, keepdim=True)
        ph, pl = split(p, self.dt)
        vh, vl = split(v, self.dt)
        o = (ph @ vh + (ph @ vl + pl @ vh)) * inv
        return self.pl(o.transpose(1, 2).reshape(N, S, d))

    def encoder(self, seq, key_pad):
        sd, dt = self.sd, self.dt
        xb = self.pl(seq)                          # layer 0 input: the embedding, as planes
        for l in range(self.L):
            p = f"seqTransEncoder.layers.{l}."
            wq, bq = sd[p + "self_attn.in_proj_weight"], sd[p + "self_attn.in_proj_bias"]
            if l == 0:
                qkv = mm3(xb, wq, dt) + bq
                res = xb
            else:
                g, be = sd[f"seqTransEncoder.layers.{l - 1}.norm2.weight"], sd[f"seqTransEncoder.layers.{l - 1}.norm2.bias"]
                qkv, mean, rstd = self.lin_ln(xb, wq, bq, g, be)
                res = (xb - mean) * rstd * g + be
            att = self.attention(qkv, key_pad)
            xa = self.pl(mm3(att, sd[p + "self_attn.out_proj.weight"], dt) + sd[p + "self_attn.out_proj.bias"] + res)
            g1, be1 = sd[p + "norm1.weight"], sd[p + "norm1.bias"]
            h, mean, rstd = self.lin_ln(xa, sd[p + "linear1.weight"], sd[p + "linear1.bias"], g1, be1)
            h = self.pl(0.5 * h * (1.0 + torch.erf(h * (1.0 / math.sqrt(2.0)))))
            res = (xa - mean) * rstd * g1 + be1
            xb = self.pl(mm3(h, sd[p + "linear2.weight"], dt) + sd[p + "linear2.bias"] + res)
        return xb

    def forward_both(self, x, t, y):
        """-> projected rows of both branches (the CFG combine happens after the projection, as in the product)"""
        sd, dt = self.sd, self.dt
        B, J, Fe, T = x.shape
        temb = orc.timestep_embedding(sd, t, self.pe, torch.float32)
        enc = y["text_embed"][0]
        c = F.linear(enc, sd["embed_text.weight"], sd["embed_text.bias"]) + temb
        u = sd["embed_text.bias"][None] + temb
        h = x.permute(0, 3, 1, 2).reshape(B, T, J * Fe)
        h = mm3(self.pl(h), sd["input_process.poseEmbedding.weight"], dt) + sd["input_process.poseEmbedding.bias"]
        fm = ~y["mask"][..., :T].reshape(B, T)
        key_pad = torch.cat([torch.zeros(B, 1, dtype=torch.bool), fm], dim=1)
        seq = torch.cat([torch.cat([c[:, None], h], 1), torch.cat([u[:, None], h], 1)], 0) + self.pe[: T + 1][None]
        xb = self.encoder(seq, torch.cat([key_pad, key_pad], 0))
        g, be = sd[f"seqTransEncoder.layers.{self.L - 1}.norm2.weight"], sd[f"seqTransEncoder.layers.{self.L - 1}.norm2.bias"]
        out, _, _ = self.lin_ln(xb, sd["output_process.poseFinal.weight"], sd["output_process.poseFinal.bias"], g, be)
        out = out[:, 1:].reshape(2 * B, T, J, Fe).permute(0, 2, 3, 1)
        oc, ou = out[:B], out[B:]
        return ou + y["scale"].view(-1, 1, 1, 1) * (oc - ou)


def loop(model, tab, shape, y, x_T, noises):
    img = x_T.clone()
    B = shape[0]
    for k, i in enumerate(range(tab.num_timesteps - 1, -1, -1)):
        t = torch.full((B,), i, dtype=torch.long)
        x0 = model.forward_both(img, t, y)
        img = orc.ddpm_step(tab, img, x0, t, noises[k])
    return img


def main():
    pos = [a for a in sys.argv[1:] if not a.startswith("--")]
    B = int(pos[0]) if len(pos) > 0 else 2
    steps = int(pos[1]) if len(pos) > 1 else 50
    hostile = "--hostile" in sys.argv
    global WSCALE
    for a in sys.argv:
        if a.startswith("--wscale="):
            WSCALE = float(a[len("--wscale="):])
    T = 196
    sd = synth_state_dict_hostile(0) if hostile else synth_state_dict(0)
    tab = orc.Tables(orc.named_betas("cosine", steps))
    shape = (B, 263, 1, T)
    if "--fixture" in sys.argv:      # the inputs of tests/golden/hostile_loop50_B2_T196.npz (oracle/make_golden_r2.py)
        y = synth_y_hostile(B, T, seed=1031, lengths=[196, 150])
        x_T, noises = orc.make_noise(shape, steps, seed=31)
    else:
        y = (synth_y_hostile if hostile else synth_y)(B, T, seed=7, lengths=[T, T - 50][:B] + [T] * max(0, B - 2))
        x_T, noises = orc.make_noise(shape, steps, seed=3)
    with torch.no_grad():
        ref = orc.sample_loop(sd, tab, shape, y, x_T, noises, cfg=True)
        r64 = orc.sample_loop(sd, tab, shape, y, x_T, noises, cfg=True, dtype=torch.float64)
        print(f"{'hostile' if hostile else 'standard'} weights: |x0| max {ref.abs().max().item():.3f}; fp32 oracle vs fp64 oracle "
              f"{(ref.double() - r64).abs().max().item():.3e}", flush=True)
        abl = [a[len("--ablate="):] for a in sys.argv if a.startswith("--ablate=")]
        if abl:     # component attribution on the fp16 split with merged (chan) statistics
            for ab in abl[0].split(","):
                got = loop(X3Model(sd, torch.float16, "chan", ablate=ab), tab, shape, y, x_T, noises)
                print(f"f16x3 chan ablate={ab!r:6s} vs fp32 oracle {(got - ref).abs().max().item():.3e}   vs fp64 "
                      f"{(got.double() - r64).abs().max().item():.3e}", flush=True)
            return
        for dt, name in ((torch.bfloat16, "bf16x3"), (torch.float16, "f16x3")):
            for stats in ("naive", "chan"):
                got = loop(X3Model(sd, dt, stats), tab, shape, y, x_T, noises)
                print(f"{name:7s} fold/{stats:5s}  vs fp32 oracle {(got - ref).abs().max().item():.3e}   vs fp64 "
                      f"{(got.double() - r64).abs().max().item():.3e}", flush=True)


if __name__ == "__main__":
    main()

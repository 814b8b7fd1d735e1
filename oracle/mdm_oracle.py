"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A CPU restatement (torch CPU tensors, fp32 by default, fp64 on request) of the reference's
DDPM sampling hot path.  It exists only as the checker for the HIP path: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.

Parity pinning: the upstream repo ships NO tests / golden vectors for this path,
so this restatement is pinned against the reference *itself*, imported in the build
container by `oracle/make_golden.py` (fixtures + max-abs report in tests/golden/).  See
tests/test_oracle_golden.py.

Every function cites the reference lines it follows (paths relative to the upstream tree).
State-dict key names are the reference's.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------
# schedules  (diffusion/gaussian_diffusion.py:22-66, :166-202; diffusion/respace.py:9-88)
# ----------------------------------------------------------------------------------------


def named_betas(name, steps, scale_betas=1.0):
    """gaussian_diffusion.py:22-46 get_named_beta_schedule (+ :49-66 betas_for_alpha_bar)."""
    if name == "linear":
        s = scale_betas * 1000.0 / steps
        return np.linspace(s * 1e-4, s * 2e-2, steps, dtype=np.float64)
    if name == "cosine":
        def abar(u):
            return math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2
        return np.array([min(1.0 - abar((i + 1) / steps) / abar(i / steps), 0.999) for i in range(steps)],
                        dtype=np.float64)
    raise NotImplementedError(name)


def respace_betas(betas, use_timesteps):
    """respace.py:74-88: betas of the sub-sequence + the map back to original indices."""
    ac = np.cumprod(1.0 - np.asarray(betas, dtype=np.float64))
    keep = sorted(set(use_timesteps))
    new, last, tmap = [], 1.0, []
    for i, a in enumerate(ac):
        if i in keep:
            new.append(1.0 - a / last)
            last = a
            tmap.append(i)
    return np.array(new, dtype=np.float64), tmap


class Tables:
    """The fp64 per-timestep arrays of gaussian_diffusion.py:166-202."""

    def __init__(self, betas):
        b = np.asarray(betas, dtype=np.float64)
        self.betas = b
        self.num_timesteps = len(b)
        a = 1.0 - b
        self.alphas_cumprod = np.cumprod(a)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1.0)
        self.posterior_variance = b * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = b * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(a) / (1.0 - self.alphas_cumprod)
        # FIXED_LARGE tables (gaussian_diffusion.py:329-333)
        self.fixed_large_variance = np.append(self.posterior_variance[1], b[1:])
        self.fixed_large_log_variance = np.log(self.fixed_large_variance)


def _coef(arr, t, dtype):
    """gaussian_diffusion.py:1602-1615: fp64 table -> gather -> .float() (a per-sample scalar)."""
    return torch.from_numpy(arr)[t].to(torch.float32).to(dtype).view(-1, 1, 1, 1)


# ----------------------------------------------------------------------------------------
# denoiser  (model/mdm.py:189-283, :296-386; torch nn/modules/transformer.py post-norm path)
# ----------------------------------------------------------------------------------------


def positional_table(max_len, d, dtype=torch.float32):
    """mdm.py:301-305 (computed in fp32 there)."""
    pe = torch.zeros(max_len, d)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2).float() * (-np.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(dtype)


def _lin(x, sd, prefix, dtype):
    return F.linear(x, sd[prefix + ".weight"].to(dtype), sd[prefix + ".bias"].to(dtype))


def timestep_embedding(sd, timesteps, pe, dtype):
    """mdm.py:329-330: time_embed(pe[t]) -> [B, d]."""
    h = _lin(pe[timesteps], sd, "embed_timestep.time_embed.0", dtype)
    h = h * torch.sigmoid(h)  # SiLU
    return _lin(h, sd, "embed_timestep.time_embed.2", dtype)


def _silu_mlp(sd, prefix, h, dtype):
    """nn.Sequential(Linear, (SiLU, Linear) x n) whose Linear layers sit at the even indices of `prefix` (model/mdm.py:405-408)."""
    n = 1 + max(int(k[len(prefix):].split(".")[0]) for k in sd if k.startswith(prefix) and k.endswith(".weight"))
    for i in range(0, n, 2):
        if i > 0:
            h = h * torch.sigmoid(h)
        h = _lin(h, sd, prefix + str(i), dtype)
    return h


def target_embedding(sd, y, names, dtype=torch.float32):
    """`self.mask_cond(self.embed_target_cond(y['target_cond'], y['target_joint_names'], y['is_heading'])[None],
    force_mask=y.get('target_uncond', False))` (model/mdm.py:197-199) -> [B, d], or None when y carries no target.  The encoder
    flavour (`--multi_encoder_type`) is read off the checkpoint's keys: EmbedTargetLocSingle (model/mdm.py:399-419), Split (:422-449),
    Multi (:451-479, with utils/misc.py:5-16 WeightedSum).  `names` = the model's all_goal_joint_names (utils/model_util.py:45)."""
    if "target_cond" not in y:
        return None
    ext = list(names) + ["traj", "heading"]
    inp = y["target_cond"].to(dtype)                                              # [B, n_ext, 3]
    B = inp.shape[0]
    chosen = []
    for b in range(B):                                                            # :413-416 / :440-443 / :471-472
        js = [str(j) for j in y["target_joint_names"][b]]
        chosen.append(js + ["heading"] if y["is_heading"][b] else js)
    p = "embed_target_cond."
    if p + "mlp.0.weight" in sd or p + "mini_mlps.0.0.weight" in sd:
        validity = torch.zeros(B, len(ext), 1, dtype=dtype)
        for b in range(B):
            for j in chosen[b]:
                validity[b, ext.index(j)] = 1.0
        mi = torch.cat([inp, validity], dim=-1)                                   # [B, n_ext, 4]
        if p + "mlp.0.weight" in sd:                                              # single: one MLP over the flattened joints
            out = _silu_mlp(sd, p + "mlp.", mi.reshape(B, -1), dtype)
        else:                                                                     # split: one narrow MLP per joint, concatenated
            out = torch.cat([_silu_mlp(sd, p + f"mini_mlps.{i}.", mi[:, i], dtype) for i in range(len(ext))], dim=-1)
    else:                                                                         # multi: per-joint MLPs of the CHOSEN joints, weighted sum
        d = sd[p + f"target_loc_emb.{ext[0]}.2.weight"].shape[0]
        w = sd[p + "target_all_loc_emb.weights"].to(dtype)
        out = torch.zeros(B, d, dtype=dtype)
        for b in range(B):
            rows = torch.zeros(len(ext), d, dtype=dtype)
            for j in chosen[b]:
                rows[ext.index(j)] = _silu_mlp(sd, p + f"target_loc_emb.{j}.", inp[b, ext.index(j)], dtype)
            out[b] = torch.matmul(w / w.sum(), rows)
    if y.get("target_uncond", False):                                             # mask_cond(force_mask=True): mdm.py:155-156
        out = torch.zeros_like(out)
    return out


def encoder_layer(sd, i, x, key_pad, num_heads, dtype):
    """One post-norm nn.TransformerEncoderLayer (mdm.py:77-81; torch transformer.py:951-983):
    x = LN1(x + out_proj(MHA(x)));  x = LN2(x + W2 gelu_erf(W1 x)).   x: [N, S, d] (batch first here)."""
    p = f"seqTransEncoder.layers.{i}."
    N, S, d = x.shape
    hd = d // num_heads
    qkv = F.linear(x, sd[p + "self_attn.in_proj_weight"].to(dtype), sd[p + "self_attn.in_proj_bias"].to(dtype))
    q, k, v = qkv.split(d, dim=-1)
    q = q.view(N, S, num_heads, hd).transpose(1, 2)
    k = k.view(N, S, num_heads, hd).transpose(1, 2)
    v = v.view(N, S, num_heads, hd).transpose(1, 2)
    sc = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
    if key_pad is not None:  # [N, S] bool, True = ignore key  (mdm.py:241-247)
        sc = sc.masked_fill(key_pad[:, None, None, :], float("-inf"))
    a = torch.softmax(sc, dim=-1) @ v
    a = a.transpose(1, 2).reshape(N, S, d)
    a = _lin(a, sd, p + "self_attn.out_proj", dtype)
    x = F.layer_norm(x + a, (d,), sd[p + "norm1.weight"].to(dtype), sd[p + "norm1.bias"].to(dtype), 1e-5)
    h = _lin(x, sd, p + "linear1", dtype)
    h = 0.5 * h * (1.0 + torch.erf(h * (1.0 / math.sqrt(2.0))))  # exact GELU (activation="gelu")
    h = _lin(h, sd, p + "linear2", dtype)
    x = F.layer_norm(x + h, (d,), sd[p + "norm2.weight"].to(dtype), sd[p + "norm2.bias"].to(dtype), 1e-5)
    return x


def mdm_forward(sd, x, timesteps, y, num_heads=4, mask_frames=True, pe=None, dtype=torch.float32, goal_joint_names=()):
    """MDM.forward for arch='trans_enc', data_rep='hml_vec' | 'rot6d'  (mdm.py:189-283); cond_mode 'text', or 'action' when the
    state dict holds `embed_action.action_embedding` (mdm.py:224-226, :389-397), or 'no_cond' when it holds neither (:227-229).

    x [B, J, F, T]; timesteps [B] int64; y: {'text_embed' [1,B,clip] | 'action' [B,1] int, 'mask' [B,1,1,T] bool, 'uncond'?,
    'target_cond'? (+ 'target_joint_names', 'is_heading', 'target_uncond'?: mdm.py:197-199)}.  Returns [B, J, F, T].
    """
    sd = {k: v for k, v in sd.items()}
    B, J, Fe, T = x.shape
    d = sd["input_process.poseEmbedding.weight"].shape[0]
    L = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("seqTransEncoder.layers."))
    if pe is None:
        pe = positional_table(5000, d, dtype)
    x = x.to(dtype)
    time_emb = timestep_embedding(sd, timesteps, pe, dtype)                      # mdm.py:195
    tgt = target_embedding(sd, y, goal_joint_names, dtype)                       # mdm.py:197-199
    if tgt is not None:
        time_emb = time_emb + tgt
    if "embed_action.action_embedding" in sd:                                    # mdm.py:224-226, :393-396
        act = sd["embed_action.action_embedding"].to(dtype)[y["action"][:, 0].to(torch.long)]
        if y.get("uncond", False):
            act = torch.zeros_like(act)
        emb = time_emb + act
    elif "embed_text.weight" not in sd:                                          # mdm.py:227-229 (no_cond)
        emb = time_emb
    else:
        enc_text = y["text_embed"].to(dtype)[0]                                  # mdm.py:210-211  [B, clip]
        if y.get("uncond", False):                                               # mdm.py:155-156, :208
            enc_text = torch.zeros_like(enc_text)
        emb = _lin(enc_text, sd, "embed_text", dtype) + time_emb                 # mdm.py:218-220  [B, d]
    h = x.permute(0, 3, 1, 2).reshape(B, T, J * Fe)                              # mdm.py:345 (batch-first here)
    h = _lin(h, sd, "input_process.poseEmbedding", dtype)                        # mdm.py:348
    key_pad = None
    if mask_frames and y["mask"].shape[-1] > 1:                                  # mdm.py:242-247
        fm = ~y["mask"][..., :T].reshape(B, T)
        key_pad = torch.cat([torch.zeros(B, 1, dtype=torch.bool), fm], dim=1)
    seq = torch.cat([emb[:, None, :], h], dim=1) + pe[: T + 1][None]             # mdm.py:251-252
    for i in range(L):                                                           # mdm.py:253
        seq = encoder_layer(sd, i, seq, key_pad, num_heads, dtype)
    out = _lin(seq[:, 1:], sd, "output_process.poseFinal", dtype)                # mdm.py:253 [1:], :375
    return out.reshape(B, T, J, Fe).permute(0, 2, 3, 1).contiguous()             # mdm.py:384-385


def cfg_forward(sd, x, timesteps, y, **kw):
    """ClassifierFreeSampleModel.forward (model/cfg_sampler.py:25-32 == utils/sampler_util.py:27-34)."""
    yu = dict(y)
    yu["uncond"] = True
    oc = mdm_forward(sd, x, timesteps, y, **kw)
    ou = mdm_forward(sd, x, timesteps, yu, **kw)
    return ou + y["scale"].to(oc.dtype).view(-1, 1, 1, 1) * (oc - ou)


# ----------------------------------------------------------------------------------------
# sampler  (gaussian_diffusion.py:226-244, :270-381, :489-541, :591-727, :729-779, :876-990)
# ----------------------------------------------------------------------------------------


def q_sample(tab, x0, t, noise):
    """gaussian_diffusion.py:226-244."""
    return (_coef(tab.sqrt_alphas_cumprod, t, x0.dtype) * x0
            + _coef(tab.sqrt_one_minus_alphas_cumprod, t, x0.dtype) * noise)


def predict_x0(model_fn, x, t, y, clip_denoised=False):
    """p_mean_variance up to pred_xstart for START_X (gaussian_diffusion.py:298-304, :347-362)."""
    out = model_fn(x, t, y)
    if "inpainting_mask" in y and "inpainted_motion" in y:                        # :300-304
        m = y["inpainting_mask"]
        out = out * (~m) + y["inpainted_motion"].to(out.dtype) * m
    if clip_denoised:
        out = out.clamp(-1, 1)
    return out


def ddpm_step(tab, x, x0, t, noise, fixed_large=False):
    """p_sample given pred_xstart (gaussian_diffusion.py:246-268, :344-345, :367-369, :525-540)."""
    dt = x.dtype
    mean = _coef(tab.posterior_mean_coef1, t, dt) * x0 + _coef(tab.posterior_mean_coef2, t, dt) * x
    logv = _coef(tab.fixed_large_log_variance if fixed_large else tab.posterior_log_variance_clipped, t, dt)
    nz = (t != 0).to(dt).view(-1, 1, 1, 1)
    return mean + nz * torch.exp(0.5 * logv) * noise


def ddim_step(tab, x, x0, t, noise, eta=0.0):
    """ddim_sample (gaussian_diffusion.py:729-779, eps from :400-404)."""
    dt = x.dtype
    eps = (_coef(tab.sqrt_recip_alphas_cumprod, t, dt) * x - x0) / _coef(tab.sqrt_recipm1_alphas_cumprod, t, dt)
    ab = _coef(tab.alphas_cumprod, t, dt)
    abp = _coef(tab.alphas_cumprod_prev, t, dt)
    sigma = eta * torch.sqrt((1 - abp) / (1 - ab)) * torch.sqrt(1 - ab / abp)
    mean = x0 * torch.sqrt(abp) + torch.sqrt(1 - abp - sigma ** 2) * eps
    nz = (t != 0).to(dt).view(-1, 1, 1, 1)
    return mean + nz * sigma * noise


def sample_loop(sd, tab, shape, y, x_T, step_noise, *, cfg=True, ddim=False, eta=0.0, clip_denoised=False,
                skip_timesteps=0, init_image=None, timestep_map=None, num_heads=4, mask_frames=True,
                dtype=torch.float32, return_all=False, const_noise=False, fixed_large=False, goal_joint_names=()):
    """p_sample_loop / ddim_sample_loop with an injected noise sequence.

    x_T: the torch.randn(*shape) of gaussian_diffusion.py:691; step_noise[k]: the k-th randn_like of
    :525 / :770 (k counts loop iterations, i.e. t = T-1-skip-k).  Follows :660-727 (+ :693-700 for
    init_image/skip_timesteps, respace.py:125-130 for the timestep map).
    """
    B = shape[0]
    pe = positional_table(5000, sd["input_process.poseEmbedding.weight"].shape[0], dtype)
    fwd = cfg_forward if cfg else mdm_forward

    def model_fn(x, t, yy):
        tt = t if timestep_map is None else torch.as_tensor(timestep_map, dtype=torch.long)[t]
        return fwd(sd, x, tt, yy, num_heads=num_heads, mask_frames=mask_frames, pe=pe, dtype=dtype,
                   goal_joint_names=goal_joint_names)

    img = x_T.to(dtype)
    indices = list(range(tab.num_timesteps - skip_timesteps))[::-1]
    if skip_timesteps and init_image is None:
        init_image = torch.zeros_like(img)
    if init_image is not None:
        t0 = torch.full((B,), indices[0], dtype=torch.long)
        img = q_sample(tab, init_image.to(dtype), t0, img)
    traj = []
    for k, i in enumerate(indices):
        t = torch.full((B,), i, dtype=torch.long)
        x0 = predict_x0(model_fn, img, t, y, clip_denoised)
        nz = step_noise[k].to(dtype)
        if const_noise:                                                           # gaussian_diffusion.py:527-528
            nz = nz[[0]].repeat(B, 1, 1, 1)
        img = ddim_step(tab, img, x0, t, nz, eta) if ddim else ddpm_step(tab, img, x0, t, nz, fixed_large)
        if return_all:
            traj.append(img.clone())
    return (img, traj) if return_all else img


def make_noise(shape, steps, seed):
    """The CPU noise stream the reference draws under torch.manual_seed(seed) (utils/fixseed.py:6-10):
    one randn(*shape) (gaussian_diffusion.py:691), then one randn_like(x) per step (:525 / :770).

    Layout subtlety restated from the reference's behaviour: from the 2nd step on, `x` is the previous
    `sample`, which inherits the *permuted* strides of OutputProcess' `.permute(1, 2, 3, 0)` (mdm.py:385):
    memory order [T, B, J, F].  `randn_like` preserves those strides and ATen's CPU normal_() takes its
    non-contiguous (serial, double-precision Box-Muller) path for them, so the values are NOT the
    contiguous-fill stream.  We reproduce that by drawing into an identically-strided tensor.
    Returned tensors are logical [B, J, F, T]; call .contiguous() before handing them to a device.
    """
    B, J, Fe, T = shape
    g = torch.Generator().manual_seed(seed)
    x_T = torch.randn(*shape, generator=g)
    noises = []
    for k in range(steps):
        if k == 0:
            noises.append(torch.randn(*shape, generator=g))
        else:
            n = torch.empty_strided(shape, (J * Fe, Fe, 1, B * J * Fe))
            noises.append(n.normal_(generator=g))
    return x_T, noises

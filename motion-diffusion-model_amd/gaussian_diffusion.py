"""`GaussianDiffusion` -- the reference's sampler seam (diffusion/gaussian_diffusion.py:112-205, :226-268,
:270-381, :489-541, :591-727, :729-779, :876-990) on the MI355X HIP path.

What is different from the reference, by design:
  * the per-timestep fp64 tables are folded ON THE HOST into three fp32 scalars per step
    (x_prev = a_x0 * x0 + a_xt * x_t + sigma * eps), so the six `_extract_into_tensor` H2D copies per step
    (gaussian_diffusion.py:1602-1615) and the ~20 elementwise kernels disappear;
  * `p_sample_loop` / `ddim_sample_loop` over our `MDM` (optionally inside `ClassifierFreeSampleModel`) run
    as ONE native call that enqueues the whole loop on the current stream (`mdm_sample_loop`);
  * fresh noise comes from a counter-based Philox stream keyed by (seed, global sample index, draw), so a
    batch sharded across GPUs produces the same samples as the unsharded batch.  The seed is drawn from
    torch's global CPU generator, so `utils/fixseed.py` still makes runs reproducible.  Parity tests inject
    the reference's CPU noise stream instead (`noise_sequence=`).
Scope: ModelMeanType.START_X with FIXED_SMALL / FIXED_LARGE variance (the only configuration
utils/model_util.py:75-116 creates).  What has no native path -- cond_fn / denoised_fn guidance, randomize_class, PLMS,
training losses, foreign (non-MI355X) models -- is handed to the REFERENCE's own `diffusion` package when that is
importable (the drop-in situation: sample/generate.py runs from the reference tree), driving whatever
model it is given step by step; without the reference on the path those calls raise NotImplementedError.
"""
import enum
import importlib
import math

import numpy as np
import torch

from .cfg_sampler import ClassifierFreeSampleModel
from .mdm import MDM


def get_named_beta_schedule(schedule_name, num_diffusion_timesteps, scale_betas=1.):
    """gaussian_diffusion.py:22-46."""
    if schedule_name == "linear":
        scale = scale_betas * 1000 / num_diffusion_timesteps
        return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)
    if schedule_name == "cosine":
        return betas_for_alpha_bar(num_diffusion_timesteps,
                                   lambda t: math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2)
    raise NotImplementedError(f"unknown beta schedule: {schedule_name}")


def betas_for_alpha_bar(num_diffusion_timesteps, alpha_bar, max_beta=0.999):
    """gaussian_diffusion.py:49-66."""
    n = num_diffusion_timesteps
    return np.array([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)])


class ModelMeanType(enum.Enum):
    PREVIOUS_X = enum.auto()
    START_X = enum.auto()
    EPSILON = enum.auto()


class ModelVarType(enum.Enum):
    LEARNED = enum.auto()
    FIXED_SMALL = enum.auto()
    FIXED_LARGE = enum.auto()
    LEARNED_RANGE = enum.auto()


class LossType(enum.Enum):
    MSE = enum.auto()
    RESCALED_MSE = enum.auto()
    KL = enum.auto()
    RESCALED_KL = enum.auto()

    def is_vb(self):
        return self == LossType.KL or self == LossType.RESCALED_KL


def _unwrap(model):
    """-> (mdm, guided) if `model` is our MDM or ClassifierFreeSampleModel(our MDM) -- possibly inside a
    respace._WrappedModel -- else (None, False)."""
    if not isinstance(model, torch.nn.Module) and hasattr(model, "timestep_map") and hasattr(model, "model"):
        model = model.model
    if isinstance(model, ClassifierFreeSampleModel) and isinstance(model.model, MDM):
        return model.model, True
    if isinstance(model, MDM):
        return model, False
    return None, False


class _MappedModel:
    """respace.py:113-134 `_WrappedModel`: the model sees the ORIGINAL timestep of a respaced process."""

    def __init__(self, model, timestep_map):
        self.model, self.timestep_map = model, timestep_map

    def __call__(self, x, ts, **kwargs):
        tmap = torch.tensor(self.timestep_map, device=ts.device, dtype=ts.dtype)
        return self.model(x, tmap[ts], **kwargs)

    def __getattr__(self, name):          # .parameters(), .encode_text, .model ...: whatever the loops poke at
        return getattr(self.__dict__["model"], name)


class _ReferenceTwin:
    """Method proxy onto the reference's GaussianDiffusion: every sampler / loss entry point takes the model first."""

    def __init__(self, base, timestep_map):
        self.base, self.timestep_map = base, timestep_map
        self.identity = timestep_map == list(range(len(timestep_map)))

    def __getattr__(self, name):
        fn = getattr(self.__dict__["base"], name)

        def call(model, *a, **k):
            return fn(model if self.identity else _MappedModel(model, self.timestep_map), *a, **k)
        return call


class GaussianDiffusion:
    def __init__(self, *, betas, model_mean_type, model_var_type, loss_type, rescale_timesteps=False, **kargs):
        self.model_mean_type = model_mean_type
        self.model_var_type = model_var_type
        self.loss_type = loss_type
        self.rescale_timesteps = rescale_timesteps
        for k, v in kargs.items():            # lambda_* / data_rep: training-loss knobs, kept as attributes
            setattr(self, k, v)
        self._ctor_kwargs = dict(betas=np.array(betas, dtype=np.float64), rescale_timesteps=rescale_timesteps, **kargs)
        self._ctor_enums = (model_mean_type.name, model_var_type.name, loss_type.name)
        self._ref_twin = None
        # compared by name so that the reference's own enum objects are accepted (drop-in via utils/model_util.py)
        if model_mean_type.name != "START_X":
            raise NotImplementedError("only ModelMeanType.START_X (utils/model_util.py:77: predict_xstart=True)")
        if model_var_type.name not in ("FIXED_SMALL", "FIXED_LARGE"):
            raise NotImplementedError("only FIXED_SMALL / FIXED_LARGE variance (utils/model_util.py:98-109)")
        if rescale_timesteps:
            raise NotImplementedError("rescale_timesteps=True is never used by the reference (utils/model_util.py:82)")

        # gaussian_diffusion.py:166-202, float64
        betas = np.array(betas, dtype=np.float64)
        self.betas = betas
        assert len(betas.shape) == 1, "betas must be 1-D"
        assert (betas > 0).all() and (betas <= 1).all()
        self.num_timesteps = int(betas.shape[0])
        alphas = 1.0 - betas
        self.alphas_cumprod = np.cumprod(alphas, axis=0)
        self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
        self.alphas_cumprod_next = np.append(self.alphas_cumprod[1:], 0.0)
        self.sqrt_alphas_cumprod = np.sqrt(self.alphas_cumprod)
        self.sqrt_one_minus_alphas_cumprod = np.sqrt(1.0 - self.alphas_cumprod)
        self.log_one_minus_alphas_cumprod = np.log(1.0 - self.alphas_cumprod)
        self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
        self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
        self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
        self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
        self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)
        self.timestep_map = list(range(self.num_timesteps))   # identity unless SpacedDiffusion overrides it

    # ---- what has no native path goes to the reference's own sampler (when importable) -----------------
    def _reference(self, why):
        """The reference's own `GaussianDiffusion` over this object's (respaced) betas, for what has no native path (SURVEY.md
        8a: "fall back to the reference implementation" for cond_fn, denoised_fn, randomize_class, PLMS, training losses
        and foreign models).  It drives any callable `model(x, t, **model_kwargs)` -- the MI355X MDM included -- one step at a
        time in torch; a non-identity timestep map is applied by wrapping the model, as respace.py:113-134 does.
        Raises NotImplementedError when the reference's `diffusion` package is not importable."""
        if self._ref_twin is None:
            try:
                rgd = importlib.import_module("diffusion.gaussian_diffusion")
            except ImportError as e:
                raise NotImplementedError(
                    f"{why}: outside the MI355X hot path (SURVEY.md 8a), and the reference's `diffusion` package is not "
                    f"importable to take over ({e}); put the reference tree on PYTHONPATH") from None
            if issubclass(rgd.GaussianDiffusion, GaussianDiffusion):
                raise NotImplementedError(f"{why}: diffusion.gaussian_diffusion.GaussianDiffusion has been rebound to the "
                                          f"MI355X class, so there is no reference implementation to hand over to")
            mean, var, loss = self._ctor_enums
            kw = dict(self._ctor_kwargs)          # betas: the respaced ones (SpacedDiffusion passes them up, respace.py:74-88)
            kw.update(model_mean_type=rgd.ModelMeanType[mean], model_var_type=rgd.ModelVarType[var], loss_type=rgd.LossType[loss])
            self._ref_twin = _ReferenceTwin(rgd.GaussianDiffusion(**kw), list(self.timestep_map))
        return self._ref_twin

    # ---- host-folded per-step scalars ----------------------------------------------------------------
    def ddpm_coefficients(self):
        """(a_x0, a_xt, sigma)[i] with the reference's fp32 rounding points: each table entry is cast to fp32
        first (`.float()` in _extract_into_tensor, :1612), sigma = exp(0.5 * logvar) in fp32 (:540), and the
        `t != 0` mask (:530-532) is folded into sigma."""
        if self.model_var_type.name == "FIXED_LARGE":                             # :329-333
            logvar = np.log(np.append(self.posterior_variance[1], self.betas[1:]))
        else:                                                                     # :334-337
            logvar = self.posterior_log_variance_clipped
        a_x0 = self.posterior_mean_coef1.astype(np.float32)
        a_xt = self.posterior_mean_coef2.astype(np.float32)
        sigma = np.exp(np.float32(0.5) * logvar.astype(np.float32)).astype(np.float32)
        sigma[0] = 0.0
        return a_x0, a_xt, sigma

    def ddim_coefficients(self, eta=0.0):
        """ddim_sample (:729-779) folded:  eps = (c*x - x0)/d ;  x_prev = x0*sqrt(abp) + sqrt(1-abp-s^2)*eps + s*z
        => a_x0 = sqrt(abp) - r/d,  a_xt = r*c/d,  r = sqrt(1 - abp - s^2)."""
        ab, abp = self.alphas_cumprod, self.alphas_cumprod_prev
        s = eta * np.sqrt((1 - abp) / (1 - ab)) * np.sqrt(1 - ab / abp)
        r = np.sqrt(np.maximum(1 - abp - s ** 2, 0.0))
        c, d = self.sqrt_recip_alphas_cumprod, self.sqrt_recipm1_alphas_cumprod
        a_x0 = (np.sqrt(abp) - r / d).astype(np.float32)
        a_xt = (r * c / d).astype(np.float32)
        sigma = s.astype(np.float32)
        sigma[0] = 0.0
        return a_x0, a_xt, sigma

    # ---- q(x_t | x_0) -----------------------------------------------------------------------------
    def q_sample(self, x_start, t, noise=None):
        """gaussian_diffusion.py:226-244 (t: [B] tensor, may differ per sample)."""
        if noise is None:
            noise = torch.randn_like(x_start)
        assert noise.shape == x_start.shape
        a = torch.from_numpy(self.sqrt_alphas_cumprod).to(t.device)[t].float().view(-1, 1, 1, 1)
        s = torch.from_numpy(self.sqrt_one_minus_alphas_cumprod).to(t.device)[t].float().view(-1, 1, 1, 1)
        return a * x_start + s * noise

    # ---- one step (the p_sample / ddim_sample seam, usable with any callable model) -------------------
    def _model_x0_parts(self, model, x, t, model_kwargs):
        """-> (out_cond, out_uncond|None, scale|None, engine): model evaluation split so that the CFG combine
        can fuse into the step kernel."""
        mdm, guided = _unwrap(model)
        y = (model_kwargs or {}).get('y', {})
        ts = self._map_timesteps(t)
        if mdm is not None and guided:
            oc, ou = mdm.forward_both(x, ts, y)
            scale = y['scale'].to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
            return oc, ou, scale, mdm.engine()
        if mdm is not None:
            return mdm(x, ts, **(model_kwargs or {})), None, None, mdm.engine()
        raise NotImplementedError("this sampler drives the MI355X MDM (optionally wrapped in "
                                  "ClassifierFreeSampleModel); foreign models are out of scope")

    def _map_timesteps(self, t):
        if self.timestep_map == list(range(len(self.timestep_map))):
            return t
        return torch.tensor(self.timestep_map, device=t.device, dtype=t.dtype)[t]      # respace.py:125-130

    def _uniform_index(self, t):
        """The common timestep index of the batch, or None when the samples sit at different timesteps (the reference's
        p_sample takes any `t`: gaussian_diffusion.py:489-541 gathers its coefficients per sample)."""
        i = int(t[0])
        return i if bool((t == i).all()) else None

    def _step(self, model, x, t, coefs, clip_denoised, denoised_fn, cond_fn, model_kwargs, noise, draw, index=None,
              const_noise=False):
        assert denoised_fn is None and cond_fn is None      # callers route those to the reference's sampler
        y = (model_kwargs or {}).get('y', {})
        i = self._uniform_index(t) if index is None else index   # (a loop knows its index: no device round trip)
        oc, ou, scale, eng = self._model_x0_parts(model, x, t, model_kwargs)
        im = y.get('inpainting_mask', None)
        imo = y.get('inpainted_motion', None)
        if im is not None and imo is not None:
            # the reference asserts both have the model output's shape (:301-303); broadcastable inputs are expanded, the
            # kernel indexes a full [B, J, F, T] array
            im = im.to(device=x.device).expand(x.shape).to(torch.uint8).contiguous()
            imo = imo.to(device=x.device, dtype=torch.float32).expand(x.shape).contiguous()
        else:
            im = imo = None
        if noise is not None:
            noise = noise.to(x.device, torch.float32)
            if const_noise:
                noise = noise[[0]].repeat(x.shape[0], 1, 1, 1)                    # :527-528
            noise = noise.contiguous()
        a_x0, a_xt, sigma = coefs
        seed, base = self._rng_state()
        if i is not None:
            x_prev, x0 = eng.sampler_step(x.contiguous(), oc, ou, scale, im, imo, noise,
                                          float(a_x0[i]), float(a_xt[i]), float(sigma[i]), clip_denoised,
                                          seed=seed, sample_base=base, draw=draw, want_x0=True, const_noise=const_noise)
            return {"sample": x_prev, "pred_xstart": x0}
        # per-sample timesteps (never on a loop's path): ONE denoiser evaluation of the whole batch above (the forward takes
        # mixed t), then the fused step kernel sample by sample with that sample's coefficients; sample b keeps its own
        # Philox stream (global index base + b), const_noise keeps sample 0's
        xs, x0s = [], []
        xc = x.contiguous()
        for b, ib in enumerate(t.tolist()):
            sl = slice(b, b + 1)
            xp, x0 = eng.sampler_step(xc[sl], oc[sl], None if ou is None else ou[sl], None if scale is None else scale[sl],
                                      None if im is None else im[sl], None if imo is None else imo[sl],
                                      None if noise is None else noise[sl].contiguous(),
                                      float(a_x0[ib]), float(a_xt[ib]), float(sigma[ib]), clip_denoised, seed=seed,
                                      sample_base=base if const_noise else base + b, draw=draw, want_x0=True,
                                      const_noise=const_noise)
            xs.append(xp)
            x0s.append(x0)
        return {"sample": torch.cat(xs), "pred_xstart": torch.cat(x0s)}

    def p_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None,
                 const_noise=False, noise=None):
        """gaussian_diffusion.py:489-541.  `noise` (extra kwarg) injects eps; otherwise the Philox stream is used."""
        if denoised_fn is not None or cond_fn is not None or _unwrap(model)[0] is None:
            self._no_stream_kwargs("p_sample", noise=noise)
            return self._reference("p_sample with cond_fn / denoised_fn / a foreign model").p_sample(
                model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                model_kwargs=model_kwargs, const_noise=const_noise)
        return self._step(model, x, t, self.ddpm_coefficients(), clip_denoised, None, None, model_kwargs,
                          noise, draw=self._next_draw(), const_noise=const_noise)

    def ddim_sample(self, model, x, t, clip_denoised=True, denoised_fn=None, cond_fn=None, model_kwargs=None, eta=0.0,
                    noise=None):
        """gaussian_diffusion.py:729-779."""
        if denoised_fn is not None or cond_fn is not None or _unwrap(model)[0] is None:
            self._no_stream_kwargs("ddim_sample", noise=noise)
            return self._reference("ddim_sample with cond_fn / denoised_fn / a foreign model").ddim_sample(
                model, x, t, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                model_kwargs=model_kwargs, eta=eta)
        return self._step(model, x, t, self.ddim_coefficients(eta), clip_denoised, None, None, model_kwargs,
                          noise, draw=self._next_draw())

    # RNG bookkeeping for the step-at-a-time API
    _seed = None
    _draw = 0
    sample_base = 0          # global index of this process' first sample (set by dist.sample_sharded)
    # Two switches of the seam itself, plain attributes of the diffusion object (rounds 2-4 read environment variables here):
    check_finite = True      # one isfinite reduction + host sync per LOOP (_check_finite); False for fully asynchronous pipelines
    dip_stepwise = False     # trans_dec: compose the window loop from mdm_forward_dec + mdm_sampler_step (what p_sample does)
                             # instead of the one-call mdm_sample_loop_dec -- the tests hold the two against each other

    def _rng_state(self):
        if self._seed is None:
            self._seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        return self._seed, self.sample_base

    def _next_draw(self):
        self._draw += 1
        return self._draw

    def reseed(self, seed=None):
        """Start a new Philox stream (seed drawn from torch's global CPU generator unless given)."""
        self._seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if seed is None else int(seed)
        self._draw = 0
        return self._seed

    # ---- the loops ---------------------------------------------------------------------------------
    def p_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                      model_kwargs=None, device=None, progress=False, skip_timesteps=0, init_image=None,
                      randomize_class=False, cond_fn_with_grad=False, dump_steps=None, const_noise=False,
                      noise_sequence=None, seed=None):
        """gaussian_diffusion.py:591-658.  Returns the final sample (or the list of dumped steps)."""
        if self._needs_reference(model, denoised_fn, cond_fn, cond_fn_with_grad, randomize_class):
            self._no_stream_kwargs("p_sample_loop", noise_sequence=noise_sequence, seed=seed)
            return self._reference("p_sample_loop with cond_fn / denoised_fn / randomize_class / a foreign model").p_sample_loop(
                model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                model_kwargs=model_kwargs, device=device, progress=progress, skip_timesteps=skip_timesteps,
                init_image=init_image, randomize_class=randomize_class, cond_fn_with_grad=cond_fn_with_grad,
                dump_steps=dump_steps, const_noise=const_noise)
        return self._loop(model, shape, self.ddpm_coefficients(), noise, clip_denoised, model_kwargs, device,
                          skip_timesteps, init_image, dump_steps, const_noise, noise_sequence, seed)

    @staticmethod
    def _no_stream_kwargs(where, **given):
        """`noise_sequence=` / `seed=` / p_sample's `noise=` are extensions of THIS implementation (injected or Philox noise
        streams).  A call that is handed to the reference's sampler draws from torch's global generator instead, so these
        arguments cannot be honoured there: refuse them loudly instead of returning an unrelated trajectory."""
        bad = [k for k, v in given.items() if v is not None]
        if bad:
            raise ValueError(f"{where}: {', '.join(bad)} cannot be combined with cond_fn / denoised_fn / randomize_class / a "
                             f"foreign model -- that call is executed by the reference's sampler, whose noise comes from "
                             f"torch's global generator (seed it with torch.manual_seed)")

    @staticmethod
    def _needs_reference(model, denoised_fn, cond_fn, cond_fn_with_grad, randomize_class):
        return (denoised_fn is not None or cond_fn is not None or bool(cond_fn_with_grad) or bool(randomize_class)
                or _unwrap(model)[0] is None)

    def ddim_sample_loop(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                         model_kwargs=None, device=None, progress=False, eta=0.0, skip_timesteps=0, init_image=None,
                         randomize_class=False, cond_fn_with_grad=False, dump_steps=None, const_noise=False,
                         noise_sequence=None, seed=None):
        """gaussian_diffusion.py:876-923."""
        if dump_steps is not None:
            raise NotImplementedError()          # as the reference (:900-901)
        if const_noise == True:                  # noqa: E712  (:902-903)
            raise NotImplementedError()
        if self._needs_reference(model, denoised_fn, cond_fn, cond_fn_with_grad, randomize_class):
            self._no_stream_kwargs("ddim_sample_loop", noise_sequence=noise_sequence, seed=seed)
            return self._reference("ddim_sample_loop with cond_fn / denoised_fn / randomize_class / a foreign model").ddim_sample_loop(
                model, shape, noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                model_kwargs=model_kwargs, device=device, progress=progress, eta=eta, skip_timesteps=skip_timesteps,
                init_image=init_image, randomize_class=randomize_class, cond_fn_with_grad=cond_fn_with_grad)
        return self._loop(model, shape, self.ddim_coefficients(eta), noise, clip_denoised, model_kwargs, device,
                          skip_timesteps, init_image, None, False, noise_sequence, seed)

    def _loop(self, model, shape, coefs, noise, clip_denoised, model_kwargs, device, skip_timesteps, init_image,
              dump_steps, const_noise, noise_sequence, seed):
        mdm, guided = _unwrap(model)
        model_kwargs = {} if model_kwargs is None else model_kwargs
        y = model_kwargs.get('y', {})
        if device is None:
            device = next(mdm.parameters()).device
        assert isinstance(shape, (tuple, list))
        shape = tuple(int(s) for s in shape)
        B, J, Fe, T = shape
        if 'text' in y.keys() and 'text_embed' not in y.keys():
            # encoding once instead of each iteration (gaussian_diffusion.py:633-635); caches into the caller's dict.
            # THE ONE DELIBERATE DIFFERENCE OF THIS SEAM: upstream re-encodes y['text'] on every p_sample_loop call even when
            # y['text_embed'] is already there (it overwrites the cache with the same tensor: generate.py:130-132 made it with the
            # same call on the same prompts, and the encoders are frozen, eval-mode, deterministic).  Here a cached embedding is
            # USED -- the text encoders are outside the hot path (north_star: "the cached CLIP text embedding is consumed
            # as-is") and absent offline.  The only caller for which upstream's overwrite changes the VALUE is DiP's dynamic-text
            # mode, whose sampler hands over a sample-major slice upstream never reads: sampler_util.AutoRegressiveSampler here
            # hands over the token-major embedding of the window's prompts instead, which is what the re-encode yields
            # (tests/golden/dip_dynamic_text_*.npz: the reference run itself).
            y['text_embed'] = model.encode_text(y['text'])
        if mdm.arch == 'trans_dec' and self.dip_stepwise:
            # the window loop one native forward + one step kernel at a time (what p_sample composes); kept as the
            # cross-check of the native window loop below
            return self._loop_stepwise(model, mdm, shape, coefs, noise, clip_denoised, model_kwargs, device,
                                       skip_timesteps, init_image, dump_steps, noise_sequence, seed, const_noise)

        eng = mdm.engine()
        with torch.no_grad():
            seed = self.reseed(seed)
            base = self.sample_base
            start = self.num_timesteps - 1 - int(skip_timesteps)
            # x at index `start` (:688-700)
            if noise is not None:
                img = noise.to(device=device, dtype=torch.float32).contiguous().clone()
            elif noise_sequence is not None:
                img = noise_sequence[0].to(device=device, dtype=torch.float32).contiguous().clone()
            else:
                img = None
            if skip_timesteps and init_image is None:
                init_image = torch.zeros(shape, dtype=torch.float32, device=device)
            if init_image is not None:
                init_image = init_image.to(device=device, dtype=torch.float32).contiguous()
                img = eng.randn(shape, device, seed, base, 0, init=init_image, eps=img,
                                a=float(np.float32(self.sqrt_alphas_cumprod[start])),
                                s=float(np.float32(self.sqrt_one_minus_alphas_cumprod[start])))
            elif img is None:
                img = eng.randn(shape, device, seed, base, 0)

            dec = mdm.arch == 'trans_dec'
            if dec:
                prefix, tokens, tok_lengths, lengths = mdm._dec_inputs(img, y)
            else:
                te = mdm.text_embedding(y, device) if mdm.cond_mode != 'no_cond' and not y.get('uncond', False) else None
            time_add = mdm.target_embedding(y, device)            # y['target_cond'], model/mdm.py:197-199: once per loop
            scale = None
            if guided:
                scale = y['scale'].to(device=device, dtype=torch.float32).reshape(-1).contiguous()
                assert scale.numel() == B
            if not dec:
                lengths = mdm.lengths_from_mask(y, T)
                if lengths is not None:
                    lengths = lengths.to(device)
            im = imo = None
            if 'inpainting_mask' in y.keys() and 'inpainted_motion' in y.keys():      # :300-304
                im = y['inpainting_mask'].to(device=device).expand(shape).to(torch.uint8).contiguous()
                imo = y['inpainted_motion'].to(device=device, dtype=torch.float32).expand(shape).contiguous()
            nz = None
            if noise_sequence is not None:
                nsteps = start + 1
                assert len(noise_sequence) >= 1 + nsteps, "noise_sequence = [x_T, eps_0, ..., eps_{nsteps-1}]"
                nz = torch.stack([n.to(device=device, dtype=torch.float32).contiguous()
                                  for n in noise_sequence[1:1 + nsteps]])
                if const_noise:                                                     # :527-528 on the injected stream
                    nz = nz[:, :1].expand(nz.shape)
                nz = nz.contiguous()
            a_x0, a_xt, sigma = coefs
            kept = sorted(set(int(k) for k in dump_steps if 0 <= int(k) <= start)) if dump_steps is not None else None
            common = dict(a_x0=a_x0, a_xt=a_xt, sigma=sigma, timestep_map=self.timestep_map, start_index=start,
                          scale=scale, lengths=lengths, inpaint_mask=im, inpaint_motion=imo, noise=nz, seed=seed,
                          sample_base=base, clip_denoised=clip_denoised, dump_steps=kept, const_noise=const_noise,
                          time_add=time_add)
            if dec:      # one DiP prediction window (sample/generate.py's autoregressive loop calls this per window)
                out, _, dumps = eng.sample_loop_dec(img, prefix=prefix, text_tokens=tokens, text_lengths=tok_lengths,
                                                    force_uncond=bool(y.get('uncond', False)), **common)
            else:
                out, _, dumps = eng.sample_loop(
                    img, text_embed=te, force_uncond=bool(y.get('uncond', False)) or mdm.cond_mode == 'no_cond', **common)
        if self.check_finite:
            self._check_finite(out, mdm)
        if dump_steps is not None:      # the reference appends in loop order (:654-657)
            return [dumps[j] for j in range(len(kept))] if kept else []
        return out

    @staticmethod
    def _check_finite(sample, mdm):
        """The fp16 operand planes of the default mode do not saturate: an activation beyond +-65504 turns into inf and the
        sample into NaN.  One reduction per LOOP (the caller reads the sample back right after, sample/generate.py:163)
        makes that loud and actionable.  `diffusion.check_finite = False` skips it (fully asynchronous pipelines)."""
        if bool(torch.isfinite(sample).all()):
            return
        raise FloatingPointError(
            "the sampling loop produced non-finite values" + (
                ": in the default precision='f16x3' the GEMM operands live as fp16 hi+lo planes (|x| <= 65504); this "
                "checkpoint's activations probably leave that range -- construct the model with precision='f32'"
                if mdm.precision != "f32" else " in the exact-fp32 mode: check the checkpoint / inputs"))

    def _loop_stepwise(self, model, mdm, shape, coefs, noise, clip_denoised, model_kwargs, device, skip_timesteps,
                       init_image, dump_steps, noise_sequence, seed, const_noise=False):
        """The trans_dec window loop as p_sample composes it: one native forward + one fused step kernel per iteration
        (`diffusion.dip_stepwise = True`).  The default is mdm_sample_loop_dec, which hoists the step-invariant text work."""
        eng = mdm.engine()
        with torch.no_grad():
            seed = self.reseed(seed)
            base = self.sample_base
            start = self.num_timesteps - 1 - int(skip_timesteps)
            if noise is not None:
                img = noise.to(device=device, dtype=torch.float32).contiguous().clone()
            elif noise_sequence is not None:
                img = noise_sequence[0].to(device=device, dtype=torch.float32).contiguous().clone()
            else:
                img = None
            if skip_timesteps and init_image is None:
                init_image = torch.zeros(shape, dtype=torch.float32, device=device)
            if init_image is not None:
                img = eng.randn(shape, device, seed, base, 0,
                                init=init_image.to(device=device, dtype=torch.float32).contiguous(), eps=img,
                                a=float(np.float32(self.sqrt_alphas_cumprod[start])),
                                s=float(np.float32(self.sqrt_one_minus_alphas_cumprod[start])))
            elif img is None:
                img = eng.randn(shape, device, seed, base, 0)
            dumps = []
            for k, i in enumerate(range(start, -1, -1)):
                t = torch.full((shape[0],), i, device=device, dtype=torch.long)
                nz = None if noise_sequence is None else noise_sequence[1 + k]
                out = self._step(model, img, t, coefs, clip_denoised, None, None, model_kwargs, nz, draw=1 + k, index=i,
                                 const_noise=const_noise)
                img = out["sample"]
                if dump_steps is not None and k in dump_steps:       # the loop's enumerate index (:637-655), not t
                    dumps.append(img.clone())
        if self.check_finite:
            self._check_finite(img, mdm)
        return dumps if dump_steps is not None else img

    # ---- progressive generators (the reference yields per step; kept for callers that iterate) ---------
    def p_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
                                  model_kwargs=None, device=None, progress=False, skip_timesteps=0, init_image=None,
                                  randomize_class=False, cond_fn_with_grad=False, const_noise=False, _ddim_eta=None,
                                  noise_sequence=None):
        """gaussian_diffusion.py:660-727: yields {'sample', 'pred_xstart'} per step (one native forward + one fused
        step kernel per iteration; use p_sample_loop for the fully-fused loop).  `noise_sequence` = [x_T, eps_0, ...]
        (extra kwarg) injects the noise stream, as in p_sample_loop."""
        if self._needs_reference(model, denoised_fn, cond_fn, cond_fn_with_grad, randomize_class):
            self._no_stream_kwargs("p_sample_loop_progressive", noise_sequence=noise_sequence)
            ref = self._reference("progressive sampling with cond_fn / denoised_fn / randomize_class / a foreign model")
            kw = dict(noise=noise, clip_denoised=clip_denoised, denoised_fn=denoised_fn, cond_fn=cond_fn,
                      model_kwargs=model_kwargs, device=device, progress=progress, skip_timesteps=skip_timesteps,
                      init_image=init_image, randomize_class=randomize_class, cond_fn_with_grad=cond_fn_with_grad)
            if _ddim_eta is None:
                yield from ref.p_sample_loop_progressive(model, shape, const_noise=const_noise, **kw)
            else:
                yield from ref.ddim_sample_loop_progressive(model, shape, eta=_ddim_eta, **kw)
            return
        mdm, _ = _unwrap(model)
        if noise is None and noise_sequence is not None:
            noise = noise_sequence[0]
        if device is None:
            device = next(mdm.parameters()).device
        shape = tuple(int(s) for s in shape)
        eng = mdm.engine()
        seed = self.reseed()
        start = self.num_timesteps - 1 - int(skip_timesteps)
        with torch.no_grad():
            img = noise.to(device=device, dtype=torch.float32).contiguous() if noise is not None else None
            if skip_timesteps and init_image is None:
                init_image = torch.zeros(shape, dtype=torch.float32, device=device)
            if init_image is not None:
                img = eng.randn(shape, device, seed, self.sample_base, 0,
                                init=init_image.to(device=device, dtype=torch.float32).contiguous(), eps=img,
                                a=float(np.float32(self.sqrt_alphas_cumprod[start])),
                                s=float(np.float32(self.sqrt_one_minus_alphas_cumprod[start])))
            elif img is None:
                img = eng.randn(shape, device, seed, self.sample_base, 0)
            coefs = self.ddpm_coefficients() if _ddim_eta is None else self.ddim_coefficients(_ddim_eta)
            for k, i in enumerate(range(start, -1, -1)):
                t = torch.full((shape[0],), i, device=device, dtype=torch.long)
                nz = None if noise_sequence is None else noise_sequence[1 + k]
                out = self._step(model, img, t, coefs, clip_denoised, None, None, model_kwargs, nz,
                                 draw=self._next_draw(), index=i, const_noise=const_noise)
                yield out
                img = out["sample"]

    def ddim_sample_loop_progressive(self, model, shape, noise=None, clip_denoised=True, denoised_fn=None,
                                     cond_fn=None, model_kwargs=None, device=None, progress=False, eta=0.0,
                                     skip_timesteps=0, init_image=None, randomize_class=False,
                                     cond_fn_with_grad=False, noise_sequence=None):
        """gaussian_diffusion.py:925-990."""
        return self.p_sample_loop_progressive(model, shape, noise, clip_denoised, denoised_fn, cond_fn, model_kwargs,
                                              device, progress, skip_timesteps, init_image, randomize_class,
                                              cond_fn_with_grad, False, _ddim_eta=eta, noise_sequence=noise_sequence)

    # ---- no native path: the reference's own implementation takes over when it is importable -----------------
    def training_losses(self, *a, **k):
        return self._reference("training_losses (training is outside the MI355X sampling hot path, SURVEY.md 2)").training_losses(*a, **k)

    def plms_sample_loop(self, *a, **k):
        return self._reference("plms_sample_loop (no live caller in the reference, gaussian_diffusion.py:1099-1187)").plms_sample_loop(*a, **k)

    def plms_sample_loop_progressive(self, *a, **k):
        return self._reference("plms_sample_loop_progressive").plms_sample_loop_progressive(*a, **k)

    def p_sample_with_grad(self, *a, **k):
        return self._reference("p_sample_with_grad (gaussian_diffusion.py:543-589)").p_sample_with_grad(*a, **k)

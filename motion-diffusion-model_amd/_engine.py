"""Native engine handle: owns one `mdm_model_t` (include/mdm_hip.h) per (module, device) and the torch
tensors that back its pointers.  PyTorch is used for device memory and streams only.
"""
import contextlib
import ctypes as C

import numpy as np
import torch

from . import _native as nat


def _ptr(t):
    return None if t is None else t.data_ptr()


def _on_own_device(method):
    """Run an Engine method with the engine's device current (see Engine._on_device)."""
    import functools

    @functools.wraps(method)
    def guarded(self, *a, **k):
        with self._on_device():
            return method(self, *a, **k)
    return guarded


# Options every NEW engine starts with (include/mdm_hip.h mdm_set_option; names: _native.OPTIONS).  Empty in production: the library's
# defaults.  The parity suites pin a kernel route here (tests/conftest.py gemm_path) -- an explicit, in-process setter; rounds 3-4
# steered the library through environment variables read on its launch path.
DEFAULT_OPTIONS = {}


class Engine:
    """Binds a state-dict (reference key names) to a native model and runs forward / sample loops."""

    def __init__(self, cfg, lib=None, precision="f16x3", options=None):
        self.lib = lib if lib is not None else nat.load_native()
        self.is_emulation = not self.lib.path.endswith(nat.LIB_NAME)
        self.cfg = nat.MdmConfig(**cfg)
        h = C.c_void_p()
        self.lib.check(self.lib.mdm_create(C.byref(self.cfg), C.byref(h)), "mdm_create")
        self.handle = h
        self.set_precision(precision)
        for k, v in {**DEFAULT_OPTIONS, **(options or {})}.items():
            self.set_option(k, v)
        self._weights = {}      # name -> tensor kept alive
        self._const_ws = None
        self._ws = None
        self.device = None
        self.ready = False

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.mdm_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def set_option(self, name, value):
        """include/mdm_hip.h mdm_set_option (the header has the full text); takes effect with the next call.
          'small_gemm_max_seqs'   default 80: up to how many sequences a forward runs on the row-tile GEMM kernel (gemm_x3s.h);
                                  0 = sequence-sized tiles only, which also sends trans_dec (DiP) to its fp32-skeleton route
          'small_gemm_row_tiles'  0 = by size (default), 1 = 32-row tiles, 2 = 64-row tiles
          'dec_fused_xattn'       trans_dec cross-attention block: 3 = by size (default: 2 below 144 row tiles, 1 from there on),
                                  2 = q projection + memory attention per (sequence, head) in one kernel + the out_proj GEMM,
                                  1 = the whole block as one kernel, 0 = three launches.  An explicit 1 / 2 whose shapes are not
                                  covered (1: latent_dim 256 / 512 and <= 96 memory tokens; 2: windows and memories of <= 64
                                  tokens) takes the other fused form if that applies, else 0
          'dec_fused_selfattn'    1 (default) = in_proj + self-attention of a trans_dec layer as one kernel per (sequence, head)
                                  for sequences of <= 64 tokens, 0 = two launches
          'attn_direct_out'       0 (default); 1 = the encoder attention kernel stores its output planes straight from the
                                  accumulators (measured slower, profiles/r05e_attention_direct.md; A/B only)
          'dec_time_token'        a model-structure switch, set by MDM for `emb_trans_dec` checkpoints: 1 = the one context row of a
                                  context_len = 1 trans_dec model is the timestep embedding (model/mdm.py:256-257), not a prefix frame"""
        if name not in nat.OPTIONS:
            raise ValueError(f"unknown engine option {name!r}: one of {sorted(nat.OPTIONS)}")
        self.lib.check(self.lib.mdm_set_option(self.handle, nat.OPTIONS[name], int(value)), f"mdm_set_option({name})")

    def get_option(self, name):
        v = C.c_int32()
        self.lib.check(self.lib.mdm_get_option(self.handle, nat.OPTIONS[name], C.byref(v)), f"mdm_get_option({name})")
        return v.value

    def set_precision(self, precision):
        """'f16x3' (default: split-precision fp16 hi/lo MFMA, three products per fp32 product) or 'f32' (exact-fp32 MFMA)."""
        if precision not in nat.PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(nat.PRECISIONS)}, got {precision!r}")
        # refuse BEFORE switching: a caller that catches the error (to fall back) must not be left with a live engine in
        # f16x3 on weight planes that do not fit it
        self._check_weight_range(precision)
        self.lib.check(self.lib.mdm_set_precision(self.handle, nat.PRECISIONS[precision]), "mdm_set_precision")
        self.precision = precision

    def _check_weight_range(self, precision=None):
        """The f16x3 operand planes hold w * 2^8 as fp16 hi + lo: a weight (or LayerNorm-gamma-folded weight) with
        |w| >= 255.9 does not fit (include/mdm_hip.h mdm_weights_in_range).  Loud at bind / mode switch, not as NaN samples."""
        precision = self.precision if precision is None else precision
        if getattr(self, "ready", False) and precision != "f32" and not getattr(self, "weights_in_range", True):
            raise nat.MdmError(
                "this checkpoint does not fit the default precision='f16x3': a weight matrix (possibly scaled by the "
                "LayerNorm gamma folded into it) has an entry of magnitude >= 255.9, beyond the fp16 operand planes "
                "(w * 2^8 <= 65504).  Construct the model with precision='f32': the exact-fp32 "
                "MFMA mode has fp32's range.")

    # ---- plumbing -------------------------------------------------------------------------
    def _check_device(self, t):
        if not self.is_emulation and not t.is_cuda:
            raise nat.MdmError("the MI355X HIP path needs tensors on a cuda (ROCm) device; got " + str(t.device))

    def stream(self):
        if self.device is not None and self.device.type == "cuda":
            return torch.cuda.current_stream(self.device).cuda_stream
        return None

    def _on_device(self):
        """Every native call runs with the engine's device current: the library launches on the calling thread's HIP
        device, so a model on cuda:1 must not be driven while cuda:0 is current (foreign stream, foreign pointers)."""
        if self.device is not None and self.device.type == "cuda":
            return torch.cuda.device(self.device)
        return contextlib.nullcontext()

    def bind(self, state, device):
        """Register every tensor of `state` (reference state-dict keys incl. 'sequence_pos_encoder.pe')."""
        self.device = torch.device(device)
        with self._on_device():
            self._bind(state)

    def _bind(self, state):
        self._weights = {}
        for name, t in state.items():
            t = t.detach().to(device=self.device, dtype=torch.float32).contiguous()
            self._check_device(t)
            self._weights[name] = t
            self.lib.check(self.lib.mdm_set_weight(self.handle, name.encode(), t.data_ptr(), t.numel()),
                           f"mdm_set_weight({name})")
        nbytes = self.lib.mdm_const_bytes(self.handle)
        self._const_ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self.lib.check(self.lib.mdm_prepare(self.handle, self._const_ws.data_ptr(), nbytes, self.stream()), "mdm_prepare")
        ok = C.c_int32(1)
        self.lib.check(self.lib.mdm_weights_in_range(self.handle, C.byref(ok), self.stream()), "mdm_weights_in_range")
        self.weights_in_range = bool(ok.value)
        self.ready = True
        self._check_weight_range()

    @_on_own_device
    def workspace(self, nseq, T):
        need = self.lib.mdm_workspace_bytes(self.handle, nseq, T)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    # ---- per-kernel-class timing (bench.py roofline) ------------------------------------------------
    @_on_own_device
    def profile(self, on):
        self.lib.check(self.lib.mdm_profile_reset(self.handle), "mdm_profile_reset")
        self.lib.check(self.lib.mdm_profile_enable(self.handle, int(bool(on))), "mdm_profile_enable")

    @_on_own_device
    def profile_read(self):
        """{class: {"ms": total, "launches": n, "flops": algorithmic flops}} of everything recorded since profile(True)."""
        out = {}
        for i, name in enumerate(nat.PROF_CLASSES):
            ms, n, fl = C.c_double(), C.c_int64(), C.c_double()
            self.lib.check(self.lib.mdm_profile_read(self.handle, i, C.byref(ms), C.byref(n), C.byref(fl)),
                           "mdm_profile_read")
            out[name] = {"ms": ms.value, "launches": n.value, "flops": fl.value}
        return out

    def _bind_time_add(self, time_add, B):
        """include/mdm_hip.h mdm_set_time_add: the target-location embedding [B, latent_dim] (model/mdm.py:197-199) the NEXT native
        call adds to the timestep embedding of its samples, both guidance branches; one-shot, so every call binds its own."""
        if time_add is None:
            return
        if time_add.dim() != 2 or time_add.shape[0] != B or time_add.shape[1] != self.cfg.latent_dim or \
                time_add.dtype != torch.float32 or not time_add.is_contiguous():
            raise ValueError(f"time_add must be a contiguous float32 [B={B}, {self.cfg.latent_dim}] block; got {tuple(time_add.shape)} "
                             f"{time_add.dtype}")
        self._check_device(time_add)
        self.lib.check(self.lib.mdm_set_time_add(self.handle, time_add.data_ptr(), B), "mdm_set_time_add")

    @_on_own_device
    def linear(self, h, weight, bias, silu=False):
        """act(h [M, K] . weight [N, K]^T + bias) by the library's exact-fp32 GEMM (include/mdm_hip.h mdm_linear; act = SiLU or none).
        Its k order is independent of M (csrc/gemm_f32.h launch_gemm_f32_t), so a row's result does not depend on which other rows
        share the launch -- what the per-loop condition encoders need to stay shard-invariant.  K is padded to a multiple of 4."""
        h = h.to(dtype=torch.float32)
        w = weight.detach().to(dtype=torch.float32)
        K = h.shape[1]
        if K % 4:
            h = torch.nn.functional.pad(h, (0, 4 - K % 4))
            w = torch.nn.functional.pad(w, (0, 4 - K % 4))
        h, w, b = h.contiguous(), w.contiguous(), bias.detach().to(dtype=torch.float32).contiguous()
        self._check_device(h)
        out = torch.empty((h.shape[0], w.shape[0]), dtype=torch.float32, device=h.device)
        self.lib.check(self.lib.mdm_linear(h.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), h.shape[0], w.shape[0],
                                           h.shape[1], 2 if silu else 0, self.stream()), "mdm_linear")
        return out

    # ---- MDM.forward ------------------------------------------------------------------------
    @_on_own_device
    def forward(self, x, timesteps, text_embed, lengths, branches, time_add=None):
        B, J, Fe, T = x.shape
        self._check_device(x)
        self._bind_time_add(time_add, B)
        nb = 2 if branches == nat.BRANCH_BOTH else 1
        out = torch.empty((nb * B, J, Fe, T), dtype=torch.float32, device=x.device)
        ws = self.workspace(nb * B, T)
        self.lib.check(self.lib.mdm_forward(self.handle, x.data_ptr(), timesteps.data_ptr(), _ptr(text_embed),
                                            _ptr(lengths), B, T, branches, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                            self.stream()), "mdm_forward")
        return out

    # ---- MDM.forward, trans_dec (DiP) ----------------------------------------------------------
    @_on_own_device
    def forward_dec(self, x, prefix, timesteps, text_tokens, text_lengths, lengths, branches, time_add=None):
        """x [B,J,F,pred_len], prefix [B,J,F,context_len] | None, text_tokens [ntok,B,dim],
        text_lengths [B] int32, lengths [B] int32 | None  ->  [B or 2B, J, F, pred_len]."""
        B, J, Fe, P = x.shape
        self._check_device(x)
        ntok = int(text_tokens.shape[0])        # also sizes the (bias + time) memory of the unconditional branch
        nb = 2 if branches == nat.BRANCH_BOTH else 1
        out = torch.empty((nb * B, J, Fe, P), dtype=torch.float32, device=x.device)
        need = self.lib.mdm_workspace_bytes_dec(self.handle, nb * B, P, ntok)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._bind_time_add(time_add, B)
        self.lib.check(self.lib.mdm_forward_dec(self.handle, x.data_ptr(), _ptr(prefix), timesteps.data_ptr(),
                                                _ptr(text_tokens), text_lengths.data_ptr(), _ptr(lengths), B, P, ntok,
                                                branches, out.data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                                self.stream()), "mdm_forward_dec")
        return out

    # ---- fused sampler pieces -----------------------------------------------------------------
    @_on_own_device
    def sampler_step(self, x_t, out_cond, out_uncond, scale, inpaint_mask, inpaint_motion, noise, a_x0, a_xt, sigma,
                     clip_denoised=False, seed=0, sample_base=0, draw=0, want_x0=False, const_noise=False):
        B = x_t.shape[0]
        per = x_t[0].numel()
        x_prev = torch.empty_like(x_t)
        x0 = torch.empty_like(x_t) if want_x0 else None
        st = nat.MdmStep(a_x0, a_xt, sigma, int(bool(clip_denoised)), seed, sample_base, draw, int(bool(const_noise)))
        self.lib.check(self.lib.mdm_sampler_step(x_t.data_ptr(), out_cond.data_ptr(), _ptr(out_uncond), _ptr(scale),
                                                 _ptr(inpaint_mask), _ptr(inpaint_motion), _ptr(noise),
                                                 x_prev.data_ptr(), _ptr(x0), B, per, C.byref(st), self.stream()),
                       "mdm_sampler_step")
        return x_prev, x0

    @_on_own_device
    def randn(self, shape, device, seed, sample_base, draw, init=None, eps=None, a=0.0, s=1.0):
        out = torch.empty(shape, dtype=torch.float32, device=device)
        self._check_device(out)
        B = shape[0]
        self.lib.check(self.lib.mdm_randn(out.data_ptr(), _ptr(init), _ptr(eps), a, s, B, out[0].numel(), seed,
                                          sample_base, draw, self.stream()), "mdm_randn")
        return out

    @staticmethod
    def _loop_params(x, T, a_x0, a_xt, sigma, timestep_map, start_index, text_embed, scale, lengths, inpaint_mask,
                     inpaint_motion, noise, seed, sample_base, clip_denoised, force_uncond, want_x0, dump_steps,
                     const_noise):
        """-> (MdmSampleParams, x0, dumps, keepalive): the block mdm_sample_loop and mdm_sample_loop_dec share."""
        n = len(a_x0)
        a0 = np.ascontiguousarray(a_x0, dtype=np.float32)
        at = np.ascontiguousarray(a_xt, dtype=np.float32)
        sg = np.ascontiguousarray(sigma, dtype=np.float32)
        tm = np.ascontiguousarray(timestep_map, dtype=np.int32)
        assert len(at) == n and len(sg) == n and len(tm) == n
        x0 = torch.empty_like(x) if want_x0 else None
        dumps = dsteps = None
        if dump_steps:
            dsteps = np.ascontiguousarray(sorted(dump_steps), dtype=np.int32)
            dumps = torch.empty((len(dsteps),) + tuple(x.shape), dtype=torch.float32, device=x.device)
        p = nat.MdmSampleParams(
            B=x.shape[0], T=T, num_timesteps=n, start_index=int(start_index),
            a_x0=a0.ctypes.data, a_xt=at.ctypes.data, sigma=sg.ctypes.data, timestep_map=tm.ctypes.data,
            text_embed_dev=_ptr(text_embed), scale_dev=_ptr(scale), lengths_dev=_ptr(lengths),
            inpaint_mask_dev=_ptr(inpaint_mask), inpaint_motion_dev=_ptr(inpaint_motion), noise_dev=_ptr(noise),
            seed=int(seed), sample_base=int(sample_base), clip_denoised=int(bool(clip_denoised)),
            force_uncond=int(bool(force_uncond)), x0_dev=_ptr(x0),
            dump_steps=(dsteps.ctypes.data if dsteps is not None else None),
            num_dump=(len(dsteps) if dsteps is not None else 0), dump_dev=_ptr(dumps),
            const_noise=int(bool(const_noise)))
        return p, x0, dumps, (a0, at, sg, tm, dsteps)

    @_on_own_device
    def sample_loop_dec(self, x, *, prefix, text_tokens, text_lengths, a_x0, a_xt, sigma, timestep_map, start_index, scale,
                        lengths, inpaint_mask=None, inpaint_motion=None, noise=None, seed=0, sample_base=0,
                        clip_denoised=False, force_uncond=False, want_x0=False, dump_steps=None, const_noise=False,
                        time_add=None):
        """In-place window loop of the trans_dec (DiP) denoiser on x [B,J,F,pred_len]; inputs as forward_dec."""
        B, J, Fe, P = x.shape
        self._check_device(x)
        ntok = int(text_tokens.shape[0])
        p, x0, dumps, keep = self._loop_params(x, P, a_x0, a_xt, sigma, timestep_map, start_index, text_tokens, scale,
                                               lengths, inpaint_mask, inpaint_motion, noise, seed, sample_base,
                                               clip_denoised, force_uncond, want_x0, dump_steps, const_noise)
        pd = nat.MdmSampleDecParams(loop=p, ntok=ntok, prefix_dev=_ptr(prefix), text_lengths_dev=_ptr(text_lengths))
        nb = 2 if scale is not None else 1
        need = self.lib.mdm_workspace_bytes_dec_loop(self.handle, nb * B, P, ntok, int(start_index) + 1)
        if self._ws is None or self._ws.numel() < need or self._ws.device != self.device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._bind_time_add(time_add, B)
        self.lib.check(self.lib.mdm_sample_loop_dec(self.handle, C.byref(pd), x.data_ptr(), self._ws.data_ptr(),
                                                    self._ws.numel(), self.stream()), "mdm_sample_loop_dec")
        del keep
        return x, x0, dumps

    @_on_own_device
    def sample_loop(self, x, *, a_x0, a_xt, sigma, timestep_map, start_index, text_embed, scale, lengths,
                    inpaint_mask=None, inpaint_motion=None, noise=None, seed=0, sample_base=0, clip_denoised=False,
                    force_uncond=False, want_x0=False, dump_steps=None, const_noise=False, time_add=None):
        """In-place loop on x [B,J,F,T] (x at index start_index).  Returns (x, x0 or None, dumps or None)."""
        B, J, Fe, T = x.shape
        self._check_device(x)
        p, x0, dumps, keep = self._loop_params(x, T, a_x0, a_xt, sigma, timestep_map, start_index, text_embed, scale,
                                               lengths, inpaint_mask, inpaint_motion, noise, seed, sample_base,
                                               clip_denoised, force_uncond, want_x0, dump_steps, const_noise)
        nb = 2 if scale is not None else 1
        ws = self.workspace(nb * B, T)
        self._bind_time_add(time_add, B)
        self.lib.check(self.lib.mdm_sample_loop(self.handle, C.byref(p), x.data_ptr(), ws.data_ptr(), ws.numel(),
                                                self.stream()), "mdm_sample_loop")
        del keep
        return x, x0, dumps

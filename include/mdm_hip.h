/* mdm_hip.h -- C ABI of libmdm_hip.so: the MI355X (gfx950) implementation of MDM's DDPM sampling hot path.
 *
 * The upstream project (GuyTevet/motion-diffusion-model) has no FFI: its boundary is three Python
 * duck-typed seams.  This header is the native layer *beneath* those seams; each entry
 * point names the reference interface it replaces (paths relative to the upstream tree).  The Python
 * mirror of the seams lives in motion-diffusion-model_amd/ and INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - every pointer named *_dev / documented "device" is a caller-owned HIP device pointer (e.g. memory of
 *     a torch tensor); the library never allocates, frees or retains tensor memory except the weight
 *     pointers registered with mdm_set_weight (which must stay valid while the model is used);
 *   - all work is enqueued asynchronously on the hipStream_t passed as `stream` (void* to keep this header
 *     free of HIP types); nothing synchronises the device;
 *   - every function returns MDM_OK (0) or a negative MDM_E* code; mdm_last_error() gives the message of the
 *     calling thread's last failure.  No C++ exception crosses the ABI;
 *   - a model handle may be used from one thread / one stream at a time.
 *   - CONCURRENCY: ONE chain of this library's kernels per device.  The library itself enforces it: every call that
 *     enqueues kernels takes a per-device lock for the duration of the (asynchronous) enqueue and, when the previous call
 *     on this device used a DIFFERENT stream, makes `stream` wait (hipStreamWaitEvent) for the event recorded behind
 *     that call's kernels.  Two model handles, two streams or two host threads on one GPU therefore run their kernels
 *     back to back, never side by side.  Why: in the f16x3 mode the DiP path's small eight-wave GEMM returns rare wrong
 *     values (one 16-lane row of one register reads as zero) when a workgroup of a DIFFERENT kernel that uses LDS is
 *     resident on the same CU at the same time -- measured with this library's own chains on side streams AND with another
 *     library's fused attention kernels (scaled-dot-product attention) on a foreign stream (17 of 60 DiP window loops differed); never on disjoint CUs,
 *     never in the f32 mode, and not on the encoder path (trans_enc: its kernels own a CU's whole LDS; 0 of 120 forwards
 *     beside the same foreign stream).  Not cache coherence, not kernel ordering (profiles/r03g_dip_groups.md); and NOT a
 *     wait-state hazard that a disassembly can show: round 6 pointed the static auditor that found round 5's
 *     `v_fma_mix -> v_mfma` hazard (tools/hazard_audit.py: VALU write -> MFMA read, MFMA write -> VALU / memory read or
 *     overwrite, every kernel, fall-through paths) at the SLP build that corrupts -- 12,806 v_pk_{add,mul,fma}_f32, 0 sites, hipcc
 *     pads its packed instructions exactly like its plain ones (profiles/r06b_slp_hazard_audit.md).  The cause stays
 *     unidentified -- looked for with the tool, not found.  What cures it is building without packed fp32 VALU math (-fno-slp-vectorize, the
 *     way __graft_entry__.build() builds this library: 0 of 520 window loops in the regimes that corrupted 100 of 520).
 *     PRECAUTION FOR CALLERS, because the effect is not understood: while mdm_forward_dec / mdm_sample_loop_dec work is in
 *     flight on a device and the results matter, keep other LDS-using kernels off it; never build the library with SLP
 *     vectorisation if they cannot be kept off (mdm_build_info() says how a binary was built; the loaders check it).  The
 *     encoder calls need no such care.
 *   - hipGraph CAPTURE: supported for a loop that stays on ONE stream (since ABI 8 a same-stream call makes no HIP runtime
 *     call besides its kernel launches and device-to-device copies: the per-device guard is a host mutex).  The warm-up may have
 *     run on ANOTHER stream (graph-capture helpers of ML frameworks capture on a side stream of their own): since ABI 9 the guard asks hipStreamIsCapturing and
 *     skips its cross-stream event while `stream` is being captured (tests/test_gpu_round5.py replays such a capture).  Run one
 *     warm-up call of the SAME shapes (batch, frames, text tokens) first: a kernel instantiation opts in to its dynamic LDS on first
 *     use, and a first use on a capturing stream is refused with MDM_EUNSUPPORTED (round 6; it used to issue the function-attribute
 *     call inside the capture).  Keep mdm_profile_enable off, and order the
 *     graph's REPLAYS against other users of the device yourself.  Not supported: a capture that contains calls from two
 *     different streams.
 *   - all tensors are fp32, dense, in the reference's layouts: poses [B, njoints, nfeats, T] (T contiguous).
 */
#ifndef MDM_HIP_H
#define MDM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDM_OK 0
#define MDM_EINVAL (-1)   /* bad argument / shape                              */
#define MDM_ESTATE (-2)   /* missing weight, mdm_prepare not called, ...       */
#define MDM_ENOSPC (-3)   /* workspace too small                               */
#define MDM_EHIP (-4)     /* a HIP runtime call failed                         */
#define MDM_EUNSUPPORTED (-5)

#define MDM_ABI_VERSION 10

typedef struct mdm_model mdm_model_t;

#define MDM_ARCH_TRANS_ENC 0  /* model/mdm.py:75-84: nn.TransformerEncoder, condition token in front of the frames   */
#define MDM_ARCH_TRANS_DEC 1  /* model/mdm.py:85-93: nn.TransformerDecoder over a text-token memory (DiP, DiP.md)    */

/* Hyper-parameters of model/mdm.py:11-135 (data_rep='hml_vec', cond_mode='text'). */
typedef struct mdm_config {
  int32_t njoints;      /* 263  (utils/model_util.py:43)                        */
  int32_t nfeats;       /* 1                                                    */
  int32_t latent_dim;   /* 512  (utils/parser_util.py:106); multiple of 256     */
  int32_t ff_size;      /* 1024 (utils/model_util.py:63); multiple of 32        */
  int32_t num_layers;   /* 8                                                    */
  int32_t num_heads;    /* 4  -> head dim must be 128                           */
  int32_t clip_dim;     /* 512: width of y['text_embed']                        */
  int32_t max_len;      /* rows of the positional table `sequence_pos_encoder.pe` (5000) */
  int32_t mask_frames;  /* args.mask_frames (model/mdm.py:48, :243)             */
  int32_t arch;         /* MDM_ARCH_TRANS_ENC (0, default) | MDM_ARCH_TRANS_DEC  */
  int32_t context_len;  /* trans_dec: prefix frames of the completion task (model/mdm.py:59, :203-206); 0 otherwise.
                         * With MDM_ARCH_TRANS_DEC, clip_dim is the width of the text-token embeddings (768, DistilBERT:
                         * model/mdm.py:121-127), emb_policy is 'add' and emb_trans_dec is False (the DiP configuration) */
} mdm_config_t;

int mdm_abi_version(void);
/* How the binary was built, as "key=value;..." -- "slp=off|on;probes=0|1;emu=0|1;planes=f16|bf16".  "slp=off" means the
 * library was compiled with -fno-slp-vectorize -DMDM_NO_SLP=1 (no packed fp32 VALU math), the only build in which the DiP path
 * is immune to the co-residency corruption described under CONCURRENCY above; the Python loader (_native.py) and the ctypes
 * stub of INTEGRATION.md refuse any other product library unless MDM_ALLOW_SLP_BUILD=1 is set (A/B experiments).  ABI 8. */
const char* mdm_build_info(void);
const char* mdm_last_error(void);

/* MDM(...) constructor (model/mdm.py:11-135) -- shapes only, no weights yet. */
int mdm_create(const mdm_config_t* cfg, mdm_model_t** out);
void mdm_destroy(mdm_model_t* m);

/* Run-time options of a model handle (ABI 9).  The library reads NO environment variable: everything that steers which kernels
 * a call runs is either derived from the shapes or set here, per handle, and takes effect with the next call (a call resolves
 * its route once; set options between calls, not from another thread during one).  No counterpart in the reference.
 *   MDM_OPT_SMALL_GEMM_MAX_SEQS   up to how many token sequences (B, or 2B under guidance) a forward runs its GEMMs on the
 *                                 32 / 64-row tiles of the latency regime (csrc/gemm_x3s.h) instead of the sequence-sized tiles
 *                                 (csrc/gemm_x3.h).  Default 80 (the measured cross-over, profiles/r05k_crossovers.md).  0: never -- which also sends the
 *                                 trans_dec (DiP) decoder to its fp32-skeleton route (csrc/gemm_f32.h).  The parity tests use it
 *                                 to hold every route against the reference's fixtures.
 *   MDM_OPT_SMALL_GEMM_ROW_TILES  0 (default): 32-row tiles up to 12 sequences, 64-row tiles above; 1 / 2: pin 32 / 64 rows.
 *   MDM_OPT_DEC_FUSED_XATTN       the cross-attention block of a trans_dec layer on the operand-plane route -- query projection with
 *                                 norm1 folded, attention over the text memory, out_proj + residual + row statistics (model/mdm.py:
 *                                 85-93, :263-265).  3 (default): by size -- 2 below 144 32-row tiles (DiP's 32 motions per GPU: 120), 1
 *                                 from there on.  2: projection + attention of a (sequence, head) in one kernel
 *                                 (csrc/selfattn_block.h, CROSS mode: windows and memories of at most 64 tokens), then the out_proj
 *                                 GEMM; 1: the whole block as ONE kernel (csrc/xattn_block.h: latent_dim 256 / 512, <= 96 memory
 *                                 tokens); 0: as three launches around the exact-fp32 attention kernel (round 4's form).  An explicit
 *                                 1 or 2 whose shapes are not covered takes the other fused form if that one applies, else 0.
 *   MDM_OPT_DEC_FUSED_SELFATTN    1 (default): in_proj + self-attention of a trans_dec layer on the operand-plane route run as ONE
 *                                 kernel per (sequence, head) for sequences of at most 64 tokens (csrc/selfattn_block.h: DiP's 20 + 40);
 *                                 0: in_proj into Q / K / V^T planes + the attention kernel (two launches; A/B and tests). */
#define MDM_OPT_SMALL_GEMM_MAX_SEQS 1
#define MDM_OPT_SMALL_GEMM_ROW_TILES 2
#define MDM_OPT_DEC_FUSED_XATTN 3
#define MDM_OPT_DEC_FUSED_SELFATTN 4
/*   MDM_OPT_ATTN_DIRECT_OUT       the split-precision self-attention kernel (csrc/attention_x3.h) writes its output planes straight from
 *                                 the accumulators and requests the next (sequence, head)'s first key tiles in front of those stores,
 *                                 instead of staging the output through its LDS ring (A/B of round 5: profiles/r05e_attention_direct.md). */
#define MDM_OPT_ATTN_DIRECT_OUT 5
/*   MDM_OPT_DEC_TIME_TOKEN        a MODEL-STRUCTURE switch (ABI 10; set once, before the first forward): `--emb_trans_dec` (model/mdm.py:101,
 *                                 :245-247, :256-257, :269-270 -- the `humanml-decoder-with-emb-512` checkpoint): the timestep embedding
 *                                 leads the trans_dec tgt sequence as a class token.  Create the model with context_len = 1 -- that row
 *                                 is the class token: always a valid key, dropped from the output like a prefix frame -- pass a
 *                                 `prefix_dev` of B zero frames [B, njoints, nfeats, 1] (a placeholder; the library overwrites the row it
 *                                 embeds to with time_embed(t) (+ the mdm_set_time_add row) + pe[0] in every branch), and count the row
 *                                 in `lengths_dev` as context rows are.  0 (default): context rows are embedded prefix frames (DiP). */
#define MDM_OPT_DEC_TIME_TOKEN 6
int mdm_set_option(mdm_model_t* m, int32_t key, int32_t value);
int mdm_get_option(const mdm_model_t* m, int32_t key, int32_t* value);

/* y['target_cond'] -- the target-location condition of the `--multi_target_cond` checkpoints (model/mdm.py:197-199; the target-conditioned
 * DiP of DiP.md:105) -- reaches the denoiser as a per-sample vector ADDED TO THE TIMESTEP EMBEDDING, in every guidance branch
 * (`time_emb += mask_cond(embed_target_cond(...), force_mask=y.get('target_uncond'))`): through it into the condition token of the
 * trans_enc sequence (mdm.py:219-220, :251) or into every row of the trans_dec text memory (:217-219, :263-265).  The embedding
 * itself (a SiLU MLP over <= 8 joints x 4 numbers per sample, model/mdm.py:399-479) does not depend on x or t: the host evaluates it
 * once per loop, like the text encoder, and binds the result here (ABI 10):
 *   add_dev  [B, latent_dim] fp32, caller-owned, alive until the consuming call's work on its stream is done; NULL clears a binding.
 * ONE-SHOT: the next mdm_forward / mdm_forward_dec / mdm_sample_loop / mdm_sample_loop_dec call on this handle consumes the binding
 * (and fails with MDM_EINVAL, consuming it, if its batch is not B); calls after it run without one.  Under MDM_BRANCH_BOTH / guidance,
 * row b is added to sample b of both branches.  A force-masked target (y['target_uncond']) is simply no binding. */
int mdm_set_time_add(mdm_model_t* m, const float* add_dev, int32_t B);

/* load_state_dict (utils/model_util.py:8-15): register the device pointer of one state-dict tensor under
 * its REFERENCE key, e.g. "seqTransEncoder.layers.3.self_attn.in_proj_weight".  `numel` is checked against
 * the shape the config implies.  The positional table is registered as "sequence_pos_encoder.pe"
 * ([max_len, latent_dim], computed by the host exactly as model/mdm.py:301-305 does). */
int mdm_set_weight(mdm_model_t* m, const char* name, const float* dev_ptr, int64_t numel);

/* Bytes of caller-owned, model-lifetime device scratch for derived tables (16-byte-aligned padded
 * poseEmbedding weight, the time-MLP table TimestepEmbedder(pe[t]) for every t: model/mdm.py:316-330). */
size_t mdm_const_bytes(const mdm_model_t* m);
/* Validates that every weight is present and (re)builds the derived tables.  Call again after weights change. */
int mdm_prepare(mdm_model_t* m, void* const_ws_dev, size_t const_ws_bytes, void* stream);

/* Range check of the split-precision weight planes.  The f16x3 arithmetic carries every weight matrix (and every
 * LayerNorm-gamma-folded weight matrix: gamma[k] * W[n][k]) as fp16 hi + lo planes of w * 2^8, i.e. it needs
 * |w| < 255.9 (activations: |x| <= 65504, checked on the samples by the Python seam).  mdm_prepare's pack kernels record a
 * violation in a device flag; this call reads it back (ONE stream synchronisation -- call it once after mdm_prepare).
 * *in_range = 1: every plane is finite; 0: use MDM_PREC_F32 for this checkpoint (the Python seam raises and says so).
 * No counterpart in the reference (its fp32 `addmm`s have fp32's range). */
int mdm_weights_in_range(mdm_model_t* m, int32_t* in_range, void* stream);

/* Arithmetic of the model's dense contractions:
 *   MDM_PREC_F32     exact fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere: bit-for-bit an fp32 fma chain; the on-device
 *                    parity reference (157 TFLOP/s peak);
 *   MDM_PREC_F16X3   default: operands split into fp16 hi + lo planes (11 + 11 significant bits), three fp16 MFMA products
 *                    per fp32 product (hi*hi + hi*lo + lo*hi), fp32 accumulation: ~2^-22 relative per product while the
 *                    operands stay inside fp16's range (|x| <= 2 * 65504; 2^-25 absolute below 2^-14) -- on the reference's
 *                    golden trajectories this is the fp32 re-association floor (2.5 PFLOP/s fp16 peak / 3 passes).
 * In MDM_PREC_F16X3 every GEMM of MDM.forward runs on the split kernel -- in_proj / out_proj / linear1 / linear2, the
 * 263-wide InputProcess / OutputProcess projections (K resp. N padded) -- and so do the attention contractions (QK^T, PV),
 * on planes written by the in_proj epilogue.  LayerNorm statistics, GELU, softmax, the time / text embeddings and the
 * sampler update are fp32 in both modes.  May be called any time after mdm_create. */
#define MDM_PREC_F32 0
#define MDM_PREC_F16X3 1
int mdm_set_precision(mdm_model_t* m, int32_t mode);

/* Per-call activation workspace for `nseq` token sequences (B, or 2B under classifier-free guidance)
 * of `nframes` frames each. */
size_t mdm_workspace_bytes(const mdm_model_t* m, int32_t nseq, int32_t nframes);

#define MDM_BRANCH_COND 0    /* y['uncond'] absent/False                     */
#define MDM_BRANCH_UNCOND 1  /* y['uncond'] == True  (model/mdm.py:155-156, :208) */
#define MDM_BRANCH_BOTH 2    /* both, batched: out = [cond(B) ; uncond(B)]   */

/* MDM.forward(x, timesteps, y)  (model/mdm.py:189-283).
 *   x_dev          [B, njoints, nfeats, T]
 *   timesteps_dev  [B] int64
 *   text_embed_dev [B, clip_dim] = y['text_embed'][0]; may be NULL for MDM_BRANCH_UNCOND
 *   lengths_dev    the key-padding mask of model/mdm.py:241-247 (`~y['mask']` -> src_key_padding_mask), in one of two forms:
 *                  [B] int32 valid-FRAME COUNTS when every y['mask'] row is a prefix mask (what data_loaders/tensors.py:3-8
 *                  builds; the kernels' fast path), or -- since ABI 7 -- [9 B] int32 for arbitrary masks: lengths[b] >= 0 is
 *                  sample b's count as before, lengths[b] == -1 says that its mask is the BITMAP in the eight words
 *                  lengths[B + 8 b .. B + 8 b + 7] (bit j of word i set: frame 32 i + j is a valid key; frames >= T ignored);
 *                  the condition token is always a valid key.  NULL = no key-padding mask (all frames valid, or
 *                  mask_frames == 0).  The same two forms hold wherever this header says `lengths_dev`
 *   T              1 <= T < cfg.max_len (the positional table, model/mdm.py:55; the condition token takes one row).  Up to 223
 *                  frames the attention keeps a query's scores in registers (exact softmax); longer sequences run the GEMMs on row
 *                  tiles at every batch size and the attention with a streaming softmax (csrc/attention_long.h, round 6).  A
 *                  BITMAP covers 256 frames: frames beyond it are masked -- use counts for longer prefix masks
 *   out_dev        [B or 2B, njoints, nfeats, T]                                                   */
int mdm_forward(mdm_model_t* m, const float* x_dev, const int64_t* timesteps_dev, const float* text_embed_dev,
                const int32_t* lengths_dev, int32_t B, int32_t T, int32_t branches, float* out_dev, void* ws_dev,
                size_t ws_bytes, void* stream);

/* MDM.forward for MDM_ARCH_TRANS_DEC (model/mdm.py:189-283 with is_prefix_comp, text_encoder_type='bert' or 'clip').  Arithmetic as
 * mdm_set_precision says: in MDM_PREC_F16X3 (default) the whole decoder stack runs on fp16 hi/lo operand planes -- the small-tile
 * split-precision GEMMs, the split-precision self-attention, all three LayerNorms of a layer folded -- with or without a frame
 * mask; exact fp32 MFMA in MDM_PREC_F32.
 *   x_dev            [B, njoints, nfeats, pred_len]      the window being denoised
 *   prefix_dev       [B, njoints, nfeats, context_len]   y['prefix'] (NULL iff context_len == 0)
 *   timesteps_dev    [B] int64
 *   text_tokens_dev  [ntok, B, clip_dim]  = y['text_embed'][0] (DistilBERT last_hidden_state, token-major as
 *                    bert_encode_text returns it, model/mdm.py:180-187); may be NULL for MDM_BRANCH_UNCOND
 *   text_lengths_dev [B] int32: tokens of each prompt (y['text_embed'][1] is a suffix pad mask: the tokenizer pads on the
 *                    right, model/BERT/BERT_encoder.py:28-30) -- the memory_key_padding_mask of mdm.py:265
 *   lengths_dev      the tgt_key_padding_mask of model/mdm.py:241-247 over the context_len + pred_len window, in the two forms
 *                    of mdm_forward's lengths_dev ([B] counts, or [9 B] counts + bitmaps) -- here the counts / bits INCLUDE the
 *                    context_len prefix frames (always valid, mdm.py:203-206) and there is no condition token -- or NULL
 *   out_dev          [B or 2B, njoints, nfeats, pred_len]  the completed suffix (mdm.py:278-279)              */
size_t mdm_workspace_bytes_dec(const mdm_model_t* m, int32_t nseq, int32_t pred_len, int32_t ntok);
int mdm_forward_dec(mdm_model_t* m, const float* x_dev, const float* prefix_dev, const int64_t* timesteps_dev,
                    const float* text_tokens_dev, const int32_t* text_lengths_dev, const int32_t* lengths_dev, int32_t B,
                    int32_t pred_len, int32_t ntok, int32_t branches, float* out_dev, void* ws_dev, size_t ws_bytes,
                    void* stream);

/* One fused sampler update given model outputs (the tail of GaussianDiffusion.p_sample / ddim_sample,
 * diffusion/gaussian_diffusion.py:489-541, :729-779, incl. ClassifierFreeSampleModel's combine
 * utils/sampler_util.py:34 and the inpainting blend :300-304):
 *   x0     = out_uncond ? out_uncond + scale[b]*(out_cond - out_uncond) : out_cond
 *   x0     = inpaint_mask ? (mask ? motion : x0) : x0 ;  clamp to [-1,1] if clip_denoised
 *   x_prev = a_x0*x0 + a_xt*x_t + sigma*eps,   eps = noise_dev ? noise_dev : Philox(seed, sample, draw)
 * a_x0/a_xt/sigma are the host-folded per-step scalars (posterior_mean_coef1/2, exp(.5*log_var)*[t!=0]). */
typedef struct mdm_step {
  float a_x0, a_xt, sigma;
  int32_t clip_denoised;
  uint64_t seed;         /* Philox key                                           */
  uint32_t sample_base;  /* global index of local sample 0 (shard-invariant RNG) */
  uint32_t draw;         /* draw index: 0 = x_T, 1+k = k-th loop iteration       */
  int32_t const_noise;   /* 1: eps of sample 0 for every sample (gaussian_diffusion.py:527-528) */
} mdm_step_t;

int mdm_sampler_step(const float* x_t_dev, const float* out_cond_dev, const float* out_uncond_dev,
                     const float* scale_dev, const uint8_t* inpaint_mask_dev, const float* inpaint_motion_dev,
                     const float* noise_dev, float* x_prev_dev, float* x0_dev, int32_t B, int32_t per_sample,
                     const mdm_step_t* step, void* stream);

/* th.randn(*shape) / q_sample (gaussian_diffusion.py:691, :226-244, :693-700) from the counter-based stream:
 *   out = init ? a*init + s*eps : eps,  eps = eps_dev ? eps_dev : Philox(seed, sample_base+b, draw).     */
int mdm_randn(float* out_dev, const float* init_dev, const float* eps_dev, float a, float s, int32_t B,
              int32_t per_sample, uint64_t seed, uint32_t sample_base, uint32_t draw, void* stream);

/* GaussianDiffusion.p_sample_loop / ddim_sample_loop (diffusion/gaussian_diffusion.py:591-727, :876-990)
 * over ClassifierFreeSampleModel(MDM) (or bare MDM when scale_dev == NULL), entirely on `stream`:
 * per step  InputProcess GEMM -> condition token -> 8 encoder layers -> OutputProcess GEMM whose epilogue
 * performs the CFG combine + posterior/DDIM update in place on x.  Host arrays are indexed by the
 * (respaced) diffusion index i = start_index .. 0. */
typedef struct mdm_sample_params {
  int32_t B, T;
  int32_t num_timesteps;        /* length of the host tables below                                   */
  int32_t start_index;          /* first i (= num_timesteps-1-skip_timesteps)                        */
  const float* a_x0;            /* host [num_timesteps]                                              */
  const float* a_xt;            /* host [num_timesteps]                                              */
  const float* sigma;           /* host [num_timesteps]  (0 at i == 0)                               */
  const int32_t* timestep_map;  /* host [num_timesteps]: model timestep for index i (respace.py:125-130) */
  const float* text_embed_dev;  /* [B, clip_dim] or NULL (unconditional)                             */
  const float* scale_dev;       /* [B] guidance scale y['scale'], NULL = no CFG (single branch)      */
  const int32_t* lengths_dev;   /* [B] or NULL                                                       */
  const uint8_t* inpaint_mask_dev;   /* [B,J,F,T] or NULL  (y['inpainting_mask'])                    */
  const float* inpaint_motion_dev;   /* [B,J,F,T] or NULL  (y['inpainted_motion'])                   */
  const float* noise_dev;       /* [nsteps, B,J,F,T] injected per-step noise, NULL = Philox          */
  uint64_t seed;
  uint32_t sample_base;
  int32_t clip_denoised;
  int32_t force_uncond;         /* 1: y['uncond']=True for the single-branch case                    */
  float* x0_dev;                /* optional: pred_xstart of the last executed step                   */
  const int32_t* dump_steps;    /* host, ascending loop indices k to snapshot (p_sample_loop dump_steps) */
  int32_t num_dump;
  float* dump_dev;              /* [num_dump, B,J,F,T]                                               */
  int32_t const_noise;          /* p_sample const_noise=True (gaussian_diffusion.py:527-528): every sample receives the
                                 * step noise of (global) sample 0 -- noise[[0]].repeat(B, 1, 1, 1)     */
} mdm_sample_params_t;

/* x_dev [B,J,F,T]: in = x at index start_index (x_T, or q_sample(init) -- see mdm_randn), out = sample. */
int mdm_sample_loop(mdm_model_t* m, const mdm_sample_params_t* p, float* x_dev, void* ws_dev, size_t ws_bytes,
                    void* stream);

/* The same loop over the MDM_ARCH_TRANS_DEC (DiP) denoiser: one call = one p_sample_loop over a prediction window
 * (sample/generate.py's autoregressive loop calls it once per window with the previous window's tail as y['prefix']).
 * `loop.T` is pred_len and `loop.text_embed_dev` the token-major text tokens [ntok, B, clip_dim] of mdm_forward_dec.
 * What does not change over the steps of a window is computed once per call: embed_text over the tokens and, per layer,
 * the key | value projections of the text part of the cross-attention memory; a step adds its projected timestep
 * embedding (one [2D] row per layer) while the attention kernel stages K / V.  The sampler update runs in place on x. */
typedef struct mdm_sample_dec_params {
  mdm_sample_params_t loop;
  int32_t ntok;
  const float* prefix_dev;           /* [B, njoints, nfeats, context_len]; NULL iff context_len == 0 */
  const int32_t* text_lengths_dev;   /* [B] tokens per prompt                                        */
} mdm_sample_dec_params_t;
size_t mdm_workspace_bytes_dec_loop(const mdm_model_t* m, int32_t nseq, int32_t pred_len, int32_t ntok, int32_t nsteps);
int mdm_sample_loop_dec(mdm_model_t* m, const mdm_sample_dec_params_t* p, float* x_dev, void* ws_dev, size_t ws_bytes,
                        void* stream);

/* Opt-in per-launch timing, for bench.py's roofline line.  While enabled, every kernel the model launches is
 * bracketed by a hipEvent pair on the launch stream and bucketed by kernel class; mdm_profile_read waits for
 * the events of one class and returns their summed duration, the launch count and the ALGORITHMIC flops of
 * those launches (2MNK per GEMM, 4 S^2 hd per (sequence, head) of attention; 0 for the HBM-bound classes).
 * The event records perturb the stream slightly, so throughput is always timed with profiling off. */
#define MDM_PROF_LINEAR 0       /* encoder GEMMs: in_proj, out_proj, linear1, linear2 */
#define MDM_PROF_ATTENTION 1
#define MDM_PROF_LAYERNORM 2
#define MDM_PROF_EMBED 3        /* InputProcess GEMM                                  */
#define MDM_PROF_OUTPROJ 4      /* OutputProcess GEMM (+ fused sampler update)        */
#define MDM_PROF_ELEMENTWISE 5  /* condition token, Philox noise                      */
#define MDM_PROF_NUM 6
int mdm_profile_enable(mdm_model_t* m, int on);
int mdm_profile_read(mdm_model_t* m, int32_t category, double* total_ms, int64_t* launches, double* flops);
int mdm_profile_reset(mdm_model_t* m);
/* Building blocks, exported for the parity tests and for callers that compose their own layers.
 *   mdm_linear:    out[M,N] = act(in[M,K] . w[N,K]^T + bias) (+ res)     act: 0 none, 1 gelu(erf), 2 silu
 *   mdm_layernorm: in-place LayerNorm over rows of D (eps 1e-5)
 *   mdm_attention: softmax(QK^T + keypad) V per (sequence, head) on a packed [nseq*S, 3D] qkv buffer whose
 *                  Q columns are pre-scaled by 1/sqrt(head_dim); lengths as in mdm_forward (indexed seq % B). */
int mdm_linear(const float* in_dev, const float* w_dev, const float* bias_dev, const float* res_dev, float* out_dev,
               int32_t M, int32_t N, int32_t K, int32_t act, void* stream);
/*   mdm_linear_x3: the same contract as mdm_linear computed by the split-precision kernel (K % 32 == 0);
 *                      `scratch_dev` (mdm_linear_x3_scratch_bytes) receives the 16-bit hi / lo planes of both operands. */
size_t mdm_linear_x3_scratch_bytes(int32_t M, int32_t N, int32_t K);
int mdm_linear_x3(const float* in_dev, const float* w_dev, const float* bias_dev, const float* res_dev,
                      float* out_dev, int32_t M, int32_t N, int32_t K, int32_t act, void* scratch_dev,
                      size_t scratch_bytes, void* stream);
int mdm_layernorm(float* x_dev, const float* gamma_dev, const float* beta_dev, int32_t rows, int32_t D, void* stream);
int mdm_attention(const float* qkv_dev, float* out_dev, const int32_t* lengths_dev, int32_t nseq, int32_t B,
                  int32_t S, int32_t D, int32_t H, void* stream);
/*   mdm_attention_x3: the same contract computed by the split-precision kernel the f16x3 mode uses (QK^T and PV
 *                  as three 16-bit MFMA products each, fp32 softmax); `scratch_dev` receives the Q/K/V^T hi / lo planes. */
size_t mdm_attention_x3_scratch_bytes(int32_t nseq, int32_t S, int32_t D);
int mdm_attention_x3(const float* qkv_dev, float* out_dev, const int32_t* lengths_dev, int32_t nseq, int32_t B,
                         int32_t S, int32_t D, int32_t H, void* scratch_dev, size_t scratch_bytes, void* stream);

/* Post-sampling transform of sample/generate.py:160-166, on the device:
 *   data = x * std + mean                         (data_loaders/humanml/data/dataset.py:132-133 inv_transform)
 *   recover_from_ric(data, joints)                (data_loaders/humanml/scripts/motion_process.py:437-452, :366-385)
 * x_dev [B, njoints_feat, 1, T] (the sampler's output, normalised HumanML3D features: 263 wide, 22 joints; KIT 251 / 21),
 * mean_dev / std_dev [njoints_feat], out_dev [B, joints, 3, T] -- the layout generate.py:166 produces by permute. */
int mdm_recover_from_ric(const float* x_dev, const float* mean_dev, const float* std_dev, float* out_dev, int32_t B,
                         int32_t T, int32_t njoints_feat, int32_t joints, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDM_HIP_H */

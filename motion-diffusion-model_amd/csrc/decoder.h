// decoder.h -- part of the ONE translation unit csrc/mdm_api.hip (the C ABI of libmdm_hip.so); split out of it in round 6
// (source health, no behaviour change).  DiP: MDM.forward for arch='trans_dec' (model/mdm.py:85-93, :203-206, :255-283).
#pragma once
// (included inside mdm_api.hip's extern "C" block; everything here sits in anonymous namespaces)

// ---- DiP: trans_dec denoiser: the fp32 skeleton (f32 mode; MDM_OPT_SMALL_GEMM_MAX_SEQS = 0) and, in the
// default f16x3 mode, the operand-plane route with its (sequence, head) attention blocks (decoder_layers_planes) ---------------
namespace {
struct DecWorkspace {
  float *tok, *qkv, *att, *ffn, *mem, *kv, *proj;
  float *stat[2];    // [M][D/32][2] partial LayerNorm statistics of the residual stream (gemm_f32.h LnFold), ping-pong
  // the plane path (decoder_pass_planes): the residual stream as two ping-pong pairs of hi | lo operand planes; the attention
  // outputs and the GELU output as planes over att / ffn; Q / K / V^T planes over qkv
  p16_t *xh[2], *xl[2];
  p16_t *atth, *attl, *ffnh, *ffnl;
  QkvPlanes qp;
  // window loop only (nsteps > 0): what is constant over the steps of one p_sample_loop
  float *out;        // [nseq][J*F*pred_len] model output of the current step
  float *kv_text;    // [nseq*ntok][L][2D]   Wkv_l . (text part of the memory)            (no bias)
  float *kv_time;    // [nsteps][L][2D]      Wkv_l . time_table[timestep of step k] + b_kv_l
  float *time_rows;  // [nsteps][D]          the gathered time-embedding rows
  float *o_text;     // [L][B][D]            Wo_l . (value row of the unconditional sequence's text part) + bo_l   (sequence-tile route
  float *o_time;     // [L][nsteps][D]       Wo_l . (value row of the step's time part)                              under guidance)
  size_t bytes;
};
DecWorkspace carve_dec(const mdm_model* m, int nseq, int S, int ntok, int B, void* base, int nsteps = 0, int pred_len = 0) {
  const size_t D = m->cfg.latent_dim, FF = m->cfg.ff_size, M = (size_t)nseq * S, Mm = (size_t)nseq * ntok;
  size_t off = 0;
  auto take = [&](size_t floats) {
    size_t o = off;
    off += align_up(floats * sizeof(float), 256);
    return base ? reinterpret_cast<float*>(static_cast<char*>(base) + o) : nullptr;
  };
  DecWorkspace w;
  w.tok = take(M * D);
  const size_t NKT = ((size_t)S + 31) / 32, SP = 32 * NKT;
  w.qkv = take((size_t)nseq * SP * 3 * D);   // self-attention: packed q|k|v rows [M][3D] (or six 16-bit planes of nseq*SP*D: the
                                             // plane path); cross-attention: the projected queries [M][D]
  w.att = take(M * D);
  w.ffn = take(M * FF);
  for (int i = 0; i < 2; ++i) {   // two 16-bit planes = one fp32 array's worth of bytes
    float* tp = take(M * D);
    w.xh[i] = reinterpret_cast<p16_t*>(tp);
    w.xl[i] = tp ? w.xh[i] + M * D : nullptr;
  }
  w.atth = reinterpret_cast<p16_t*>(w.att);
  w.attl = w.att ? w.atth + M * D : nullptr;
  w.ffnh = reinterpret_cast<p16_t*>(w.ffn);
  w.ffnl = w.ffn ? w.ffnh + M * FF : nullptr;
  {
    const size_t plane = (size_t)nseq * SP * D;
    p16_t* q = reinterpret_cast<p16_t*>(w.qkv);
    w.qp = QkvPlanes{q, q ? q + plane : nullptr, q ? q + 2 * plane : nullptr, q ? q + 3 * plane : nullptr,
                     q ? q + 4 * plane : nullptr, q ? q + 5 * plane : nullptr, (int)SP, (int)NKT, m->cfg.num_heads};
  }
  w.mem = take(Mm * D);           // text memory [nseq][ntok][D]
  w.kv = take(Mm * 2 * D);        // its key | value projections of the current layer
  w.proj = take((size_t)ntok * B * D);   // embed_text(enc_text), token-major
  w.stat[0] = take(M * (D / LN_PART_COLS) * 2);
  w.stat[1] = take(M * (D / LN_PART_COLS) * 2);
  w.out = w.kv_text = w.kv_time = w.time_rows = w.o_text = w.o_time = nullptr;
  if (nsteps > 0) {
    const size_t L = m->cfg.num_layers;
    w.out = take((size_t)nseq * m->jf * pred_len);
    w.kv_text = take(L * Mm * 2 * D);
    w.kv_time = take(L * nsteps * 2 * D);
    w.time_rows = take((size_t)nsteps * D);
    w.o_text = take(L * (size_t)B * D);
    w.o_time = take(L * (size_t)nsteps * D);
  }
  w.bytes = off;
  return w;
}
}  // namespace

namespace {
int check_dec_shapes(const mdm_model_t* m, const char* who, const float* prefix, int B, int pred_len, int ntok) {
  const int C = m->cfg.context_len, S = C + pred_len;
  const std::string w(who);
  if (m->cfg.arch != MDM_ARCH_TRANS_DEC) return fail(MDM_ESTATE, w + ": the model was created as trans_enc");
  if ((C > 0) != (prefix != nullptr)) return fail(MDM_EINVAL, w + ": prefix must be given iff context_len > 0");
  if (B <= 0 || pred_len <= 0 || S > m->cfg.max_len) return fail(MDM_EINVAL, w + ": need B >= 1 and context_len + pred_len <= the positional table's length");
  if (ntok <= 0 || ntok > 512) return fail(MDM_EINVAL, w + ": 1 <= text tokens <= 512");
  if (S > m->cfg.max_len) return fail(MDM_EINVAL, w + ": window longer than the positional table");
  return MDM_OK;
}

// One evaluation of the trans_dec denoiser.  hoist_step < 0: the stand-alone forward (memory = text + time built here from
// `timesteps`, projected per layer).  hoist_step = k >= 0: step k of a window loop -- ws.kv_text / ws.kv_time are filled,
// the memory is never materialised and the per-layer memory projection is skipped.
struct DecHoist {         // step k of a window loop: where the hoisted projections of the (whole) batch live
  int step = -1;          // < 0: not hoisted
  int nsteps = 0;
  const float* kv_text = nullptr;   // [nbranch * kv_B * ntok][L][2D]: row stride L * 2D, layer l at + l * 2D
  const float* kv_time = nullptr;   // [nsteps][L][2D]
  int kv_B = 0, kv_b0 = 0;          // this pass covers samples kv_b0 .. kv_b0 + B - 1 of kv_B
  int t_model = 0;                  // the step's model timestep (the class token of MDM_OPT_DEC_TIME_TOKEN needs its table row)
  // the unconditional half's cross-attention block as row constants, hoisted with the projections (null: made per step):
  const float* o_text = nullptr;    // [L][kv_B][D]  Wo_l . v_text(unconditional sequence b) + bo_l
  const float* o_time = nullptr;    // [L][nsteps][D]  Wo_l . v_time(step)
};
// `--emb_trans_dec` (model/mdm.py:256-257, the `humanml-decoder-with-emb-512` checkpoint): the TIMESTEP embedding leads the decoder's
// tgt sequence as a class token.  The library sees it as the one context row of a context_len = 1 model (always a valid key, dropped
// from the output like a prefix frame); the embedding GEMM has written an embedded placeholder frame there -- overwritten here, in
// every branch, with time_embed(t) (+ the target embedding) + pe[0], as fp32 and / or planes.
int dec_time_token_rows(mdm_model_t* m, float* tok, p16_t* th, p16_t* tl, const int64_t* timesteps, const DecHoist& hz, int B, int nseq,
                        int S, hipStream_t s) {
  if (!m->dec_time_token) return MDM_OK;
  const int D = m->cfg.latent_dim;
  ProfScope ps(&m->prof, MDM_PROF_ELEMENTWISE, 0.0, s);
  MDM_LAUNCH(cond_token_kernel, dim3(nseq), dim3(128), 0, s, tok, (const float*)nullptr, (const float*)nullptr,
             (const float*)m->time_table, reinterpret_cast<const long long*>(hz.step >= 0 ? nullptr : timesteps), hz.t_model,
             m->W("sequence_pos_encoder.pe"), B, S, D, 0, (int)m->cfg.max_len, th, tl,
             m->time_add != nullptr ? m->time_add + (size_t)hz.kv_b0 * D : (const float*)nullptr);
  return rt_launch_status();
}
// The sampler update of a window-loop step, handed DOWN to the plane route: its transposing tail kernel (outproj_finish_kernel
// mode 1) then performs guidance combine + inpainting blend + clamp + posterior / DDIM update + inline Philox in place on x, exactly
// as the encoder loop's tail does -- one launch and one [nseq, J, P] round trip through memory fewer per step than
// OutputProcess -> sampler_step_kernel (same arithmetic, element for element).  `done` says whether the route applied it.
constexpr int kXattnOneKernelWgs = 144;    // MDM_OPT_DEC_FUSED_XATTN = 3: from this many 32-row tiles on, xattn_block.h's one-kernel block
struct DecTail {
  const float* scale = nullptr;      // [B] or null (single branch)
  float* x = nullptr;                // [B, J, F, P]: x_t in, x_{t-1} out
  float* x0_out = nullptr;
  NoiseSource ns{};
  const unsigned char* inpaint_mask = nullptr;
  const float* inpaint_motion = nullptr;
  StepCoefs co{};
  bool done = false;
};
// The decoder stack on 16-bit operand planes: the f16x3 mode at the sizes the reference's DiP callers run (model/mdm.py:255-270
// under sample/generate.py's autoregressive windows: 2 x 32 sequences of 20 + 40 tokens = 3,840 token rows).  That is the row
// count of the encoder's latency regime, so the six GEMMs of a layer run on gemm_x3s.h's 32 / 64-row tiles straight from planes
// (the fp32 skeleton of gemm_f32.h splits its operands inside the k-loop: 28 us per GEMM at this size), the self-attention on
// attention_x3.h's Q / K / V^T planes, and only the cross-attention -- 24 memory tokens whose keys / values are hoisted fp32 --
// stays on attention_f32.h (fp32 queries in, planes out).  All three LayerNorms of a layer are folded exactly as in the encoder
// (row statistics per 128 columns from the producer, merged by the consumer); the residual stream ping-pongs between two plane
// pairs because a GEMM cannot write the array its residual's statistics are read from.
// Frame masks (tgt_key_padding_mask, model/mdm.py:241-247 -- what DiP.md:181's `--mask_frames` recipe hands over on every call)
// travel as counts / bitmaps into attention_x3.h with lead = 0 since round 5.
// Not taken (the fp32 skeleton below stays): f32 mode, sample groups of the probe build; mdm_set_option(MDM_OPT_SMALL_GEMM_MAX_SEQS,
// 0) forces the skeleton (tests, A/B runs).  There is no upper row count: the alternative
// is not gemm_x3.h's sequence tiles (a 60-token sequence fills a quarter of one) but the skeleton, and the planes win at every
// size measured (B = 32: 544 vs 391 motions/s, B = 64: 660 vs 448; profiles/r04h_dip_planes.md).
// (sequences of 129 .. 224 tokens, more of them than MDM_OPT_SMALL_GEMM_MAX_SEQS: see decoder_layers_planes)
inline bool dec_sequence_tiles(const mdm_model* m, int nseq, int S) {
  return nseq > m->x3s.max_seqs && S > 128 && S <= X3_TM && x3_waves_setting() == 8;
}
inline bool dec_on_planes(const mdm_model* m, int M, int S, const DecHoist& hz, int B) {
  (void)M;
  return m->precision == MDM_PREC_F16X3 && m->x3s.max_seqs > 0 &&
         m->cfg.latent_dim % 256 == 0 && m->cfg.ff_size % 256 == 0 && m->out_planes_f.hi != nullptr &&
         (hz.step < 0 || (hz.kv_b0 == 0 && hz.kv_B == B));
}

int decoder_layers_planes(mdm_model_t* m, const DecWorkspace& ws, const float* x, const float* prefix, const int64_t* timesteps,
                          const int32_t* text_lengths, const int32_t* len, int B, int pred_len, int ntok, int nbranch, float* out,
                          hipStream_t s, const DecHoist& hz, DecTail* tail) {
  const int C = m->cfg.context_len, S = C + pred_len, D = m->cfg.latent_dim, H = m->cfg.num_heads, FF = m->cfg.ff_size;
  const int nseq = nbranch * B, M = nseq * S, Mm = nseq * ntok;
  Profiler* pf = &m->prof;
  const float qscale = 1.0f / sqrtf((float)ATT_HD);
  const bool hoisted = hz.step >= 0;
  int cur = 0;   // which plane pair holds the layer input
  {  // tgt tokens: InputProcess over cat(prefix, x) + positional rows, written as planes (the fp32 copy in ws.tok is not read)
    PoseGatherLoader al{x, S, m->jf, B * S, prefix, C};
    RowMajorLoader bl{m->w_in_pad, m->jf_pad, D, m->jf_pad};
    EmbedEpilogue ep{ws.tok, m->W("input_process.poseEmbedding.bias"), m->W("sequence_pos_encoder.pe"), B, S, S, D, nbranch,
                     ws.xh[cur], ws.xl[cur], 0};
    ProfScope ps(pf, MDM_PROF_EMBED, 2.0 * B * S * (double)D * m->jf, s);
    launch_gemm_f32(al, bl, ep, B * S, D, m->jf_pad, s, true);
    if (int rc = rt_launch_status()) return rc;
  }
  if (int rc = dec_time_token_rows(m, ws.tok, ws.xh[cur], ws.xl[cur], timesteps, hz, B, nseq, S, s)) return rc;
  const X3sShape shape = x3s_shape(m->x3s, (M + 196) / 197);   // the encoder's 32- / 64-row threshold, in its token rows
  // Which GEMM kernel: the row tiles of gemm_x3s.h (DiP's windows at every batch size: a 60-token sequence fills a quarter of a
  // sequence tile) -- or, for LONG sequences at LARGE batch, gemm_x3.h's sequence-sized tiles exactly as the encoder chooses them
  // (use_small_gemm): the reference's full-length trans_dec checkpoint (README.md:254 humanml_trans_dec_512_bert-50steps: 196 frames,
  // no prefix, plain p_sample_loop) at the headline batch is the encoder's shape with a cross-attention block per layer.
  const bool small = !dec_sequence_tiles(m, nseq, S);
  const int scols = small ? x3s_tn(shape.ncb) : 256, parts = (D + scols - 1) / scols;
  const float inv_dim = 1.0f / (float)D;
  auto LN = [&]() { LnArgs a; a.small = small; a.shape = shape; a.stat_cols = scols; a.parts = parts; a.inv_dim = inv_dim; return a; };
  const X3Operand attp{ws.atth, ws.attl}, ffnp{ws.ffnh, ws.ffnl};
  float* q32 = ws.tok;   // the projected cross-attention queries [M][D]
  // A trans_dec tgt sequence carries no condition token (emb_trans_dec: the timestep embedding, the same in both branches), so under
  // guidance the two branches enter layer 0 with IDENTICAL rows and stay identical through its self-attention block (in_proj,
  // attention, out_proj + residual): they part at the first cross-attention.  The sequence-tile route (throughput-bound) runs that
  // block on the conditional half and copies the result; bit-identical to computing it twice.
  const bool share0 = !small && nbranch == 2 && !(m->fused_selfattn && selfattn_block_supported(D, S));
  const int M0 = share0 ? B * S : M;
  for (int l = 0; l < m->cfg.num_layers; ++l) {
    const mdm_model::DecFold& F = m->dec_fold[l];
    const mdm_model::DecPlanes& P = m->dec_planes[l];
    p16_t *Xh = ws.xh[cur], *Xl = ws.xl[cur], *Yh = ws.xh[cur ^ 1], *Yl = ws.xl[cur ^ 1];
    const X3Operand X{Xh, Xl}, Y{Yh, Yl};
    float *sX = ws.stat[cur], *sY = ws.stat[cur ^ 1];
    // ---- Y = X' + self_attn(X'), X' = norm3(l-1)(X) (the embedded tokens for l = 0).  Sequences of at most 64 tokens (DiP: 20 + 40):
    // in_proj + attention of a (sequence, head) in one kernel (selfattn_block.h: Q / K / V^T never leave the CU); else in_proj into
    // operand planes + attention_x3.h
    if (m->fused_selfattn && selfattn_block_supported(D, S)) {
      SelfAttnArgs sa{};
      sa.x = X; sa.xstat = l == 0 ? nullptr : sX; sa.w = P.in_proj;
      sa.bias = l == 0 ? m->L(l, "self_attn.in_proj_bias") : F.b_in; sa.colsum = l == 0 ? nullptr : F.c_in;
      sa.qscale = qscale; sa.lengths = len; sa.lead = 0; sa.B = B; sa.oh = ws.atth; sa.ol = ws.attl;
      sa.M = M; sa.S = S; sa.D = D; sa.H = H; sa.stat_parts = parts; sa.stat_cols = scols; sa.inv_dim = inv_dim; sa.acc_scale = kX3AccScale;
      ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * M * 3.0 * D * (double)D + 4.0 * nseq * H * (double)S * S * ATT_HD, s);
      const int rc = launch_seqhead_block(sa, l != 0 ? 1 : 0, s);
      if (rc == -1 || rc == -3) return lds_fail(rc, "self-attention block");
      if (rc != 0) return fail(MDM_EUNSUPPORTED, "self-attention block: unsupported shape");
      if (int rc2 = rt_launch_status()) return rc2;
    } else {
    if (l == 0) {
      LnArgs a = LN();
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 6, X, P.in_proj, m->L(l, "self_attn.in_proj_bias"), a, nullptr, nullptr,
                                nullptr, &ws.qp, M0, 3 * D, D, S, D, D, qscale, s)) return rc;
    } else {
      LnArgs a = LN(); a.astat = sX; a.colsum = F.c_in;
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 0, X, P.in_proj, F.b_in, a, nullptr, nullptr, nullptr, &ws.qp, M, 3 * D, D,
                                S, D, D, qscale, s)) return rc;
    }
    if (int rc = launch_attention_x3(pf, ws.qp, len, share0 && l == 0 ? B : nseq, B, S, D, nullptr, ws.atth, ws.attl, s, /*lead=*/0,
                                     m->attn_direct)) return rc;
    }   // !fused self-attention
    {
      LnArgs a = LN(); a.res = X; a.ostat = sY;
      if (l >= 1) { a.rstat = sX; a.rgamma = m->L(l - 1, "norm3.weight"); a.rbeta = m->L(l - 1, "norm3.bias"); }
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, l == 0 ? 1 : 2, attp, P.out_proj, m->L(l, "self_attn.out_proj.bias"), a,
                                nullptr, Yh, Yl, nullptr, l == 0 ? M0 : M, D, D, S, D, 0, 1.f, s)) return rc;
    }
    if (share0 && l == 0) {   // the unconditional half of layer 0's self-attention block IS the conditional half: copy its rows
      ProfScope ps(pf, MDM_PROF_ELEMENTWISE, 0.0, s);
      if (int rc = rt_copy(Yh + (size_t)M0 * D, Yh, (size_t)M0 * D * sizeof(p16_t), s)) return rc;
      if (int rc = rt_copy(Yl + (size_t)M0 * D, Yl, (size_t)M0 * D * sizeof(p16_t), s)) return rc;
      if (int rc = rt_copy(sY + (size_t)M0 * parts * 2, sY, (size_t)M0 * parts * 2 * sizeof(float), s)) return rc;
    }
    // ---- X = norm1(Y) + multihead_attn(norm1(Y), memory, memory).  One kernel (xattn_block.h: q projection with norm1 folded ->
    // attention over the memory -> out_proj + norm1 residual + row statistics) where its shapes are covered; else three launches:
    // fp32 queries (pre-scaled) from the small GEMM, the exact-fp32 attention kernel over k | v of the memory, the small GEMM again
    // by size (3): the one-kernel block re-reads all of Wq | Wo per 32-row tile -- it pays once its nseq * ceil(S / 32) workgroups
    // fill the chip (same-box, motions/s, one kernel vs (sequence, head) form: B = 32 per GPU / 128 tiles 598 vs 630, B = 48 / 192 tiles
    // 641 vs 605, B = 64 / 256 tiles 736 vs 707: profiles/r05c section 5)
    const int xb_wgs = nseq * ((S + XB_TR - 1) / XB_TR);
    const int xmode = m->fused_xattn == 3 ? ((xb_wgs >= kXattnOneKernelWgs && xattn_block_supported(D, ntok) && scols == 128) ? 1 : 2)
                                          : m->fused_xattn;
    // a form whose shapes are not covered falls to the OTHER fused form before the three-launch one (an explicit 1 at
    // latent_dim 768 / 1024 used to drop straight to 0 although 2 applies)
    const bool can_sh = crossattn_block_supported(D, S, ntok), can_one = xattn_block_supported(D, ntok) && scols == 128;
    const bool seqhead = (xmode == 2 && can_sh) || (xmode == 1 && !can_one && can_sh);
    const bool fused = !seqhead && xmode != 0 && can_one;
    if (!hoisted) {
      const float* wc = m->L(l, "multihead_attn.in_proj_weight");
      const float* bc = m->L(l, "multihead_attn.in_proj_bias");
      if (int rc = launch_linear(pf, ws.mem, D, wc + (size_t)D * D, bc + D, nullptr, ws.kv, Mm, 2 * D, D, ACT_NONE, 0, 1.f, s, true)) return rc;
    }
    if (seqhead) {
      SelfAttnArgs ca{};
      ca.x = Y; ca.xstat = sY; ca.w = P.q; ca.bias = F.b_q; ca.colsum = F.c_q; ca.qscale = qscale;
      ca.lengths = nullptr; ca.lead = 0; ca.B = B; ca.oh = ws.atth; ca.ol = ws.attl;
      ca.M = M; ca.S = S; ca.D = D; ca.H = H; ca.stat_parts = parts; ca.stat_cols = scols; ca.inv_dim = inv_dim; ca.acc_scale = kX3AccScale;
      if (!hoisted) {
        ca.mk = ws.kv; ca.mv = ws.kv + D; ca.kadd = ca.vadd = nullptr; ca.kv_B = 0; ca.kv_b0 = 0;
      } else {
        const float* kvt = hz.kv_text + (size_t)l * 2 * D;
        const float* row = hz.kv_time + ((size_t)hz.step * m->cfg.num_layers + l) * 2 * D;
        ca.mk = kvt; ca.mv = kvt + D; ca.kadd = row; ca.vadd = row + D; ca.kv_B = hz.kv_B; ca.kv_b0 = hz.kv_b0;
      }
      ca.ldkv = hoisted ? m->cfg.num_layers * 2 * D : 2 * D; ca.text_lengths = text_lengths; ca.ntok = ntok;
      {
        ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * M * (double)D * D + 4.0 * M * (double)ntok * D, s);
        const int rc = launch_seqhead_block(ca, 2, s);
        if (rc == -1 || rc == -3) return lds_fail(rc, "cross-attention (sequence, head) kernel");
        if (rc != 0) return fail(MDM_EUNSUPPORTED, "cross-attention (sequence, head) kernel: unsupported shape");
        if (int rc2 = rt_launch_status()) return rc2;
      }
      LnArgs a = LN(); a.res = Y; a.rstat = sY; a.rgamma = m->L(l, "norm1.weight"); a.rbeta = m->L(l, "norm1.bias"); a.ostat = sX;
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 2, attp, P.out_proj2, m->L(l, "multihead_attn.out_proj.bias"), a, nullptr,
                                Xh, Xl, nullptr, M, D, D, S, D, 0, 1.f, s)) return rc;
    } else if (fused) {
      XattnArgs xa{};
      xa.y = Y; xa.ystat = sY; xa.wq = P.q; xa.cq = F.c_q; xa.bq = F.b_q; xa.qscale = qscale;
      if (!hoisted) {
        xa.k = ws.kv; xa.v = ws.kv + D; xa.kadd = xa.vadd = nullptr; xa.kv_B = 0; xa.kv_b0 = 0;
      } else {
        const float* kvt = hz.kv_text + (size_t)l * 2 * D;
        const float* row = hz.kv_time + ((size_t)hz.step * m->cfg.num_layers + l) * 2 * D;
        xa.k = kvt; xa.v = kvt + D; xa.kadd = row; xa.vadd = row + D; xa.kv_B = hz.kv_B; xa.kv_b0 = hz.kv_b0;
      }
      xa.ldkv = hoisted ? m->cfg.num_layers * 2 * D : 2 * D; xa.text_lengths = text_lengths; xa.ntok = ntok; xa.B = B;
      xa.wo = P.out_proj2; xa.bo = m->L(l, "multihead_attn.out_proj.bias");
      xa.gamma = m->L(l, "norm1.weight"); xa.beta = m->L(l, "norm1.bias");
      xa.oh = Xh; xa.ol = Xl; xa.ostat = sX; xa.M = M; xa.S = S; xa.inv_dim = inv_dim; xa.acc_scale = kX3AccScale;
      // (profiled as ONE launch of the GEMM class: 2 D^2 per row twice + the attention contractions)
      ProfScope ps(pf, MDM_PROF_LINEAR, 4.0 * M * (double)D * D + 4.0 * M * (double)ntok * D, s);
      const int rc = launch_xattn_block(xa, D, s);
      if (rc == -1 || rc == -3) return lds_fail(rc, "cross-attention block");
      if (rc != 0) return fail(MDM_EUNSUPPORTED, "cross-attention block: unsupported shape");
      if (int rc2 = rt_launch_status()) return rc2;
    } else {
    // sequence-tile route under guidance: the unconditional half (branch 1 = sequences B .. 2B-1) needs neither the q projection nor
    // the attention -- every query's output is the sequence's one projected value row (elementwise.h uncond_xattn_rows_kernel); the
    // cross out_proj below runs over all rows as before.  -75 us of a 1,138 us layer at B = 128 (profiles/r06h).
    const bool skip_uncond = !small && nbranch == 2;
    const int nseq_q = skip_uncond ? B : nseq, Mq = nseq_q * S;
    {
      LnArgs a = LN(); a.astat = sY; a.colsum = F.c_q;
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 4, Y, P.q, F.b_q, a, q32, nullptr, nullptr, nullptr, Mq, D, D, S, D, D, qscale,
                                s)) return rc;
    }
    const float* vsrc; const float* vadd = nullptr; size_t vstride; int vseq0;
    if (!hoisted) {
      const AttnF32Args a{q32, D, ws.kv, ws.kv + D, 2 * D, S, ntok, text_lengths, 0, B};
      if (int rc = launch_attention_args(pf, a, nullptr, nseq_q, D, H, ws.atth, ws.attl, s)) return rc;
      vsrc = ws.kv + D; vstride = (size_t)ntok * 2 * D; vseq0 = B;
    } else {
      const float* kvt = hz.kv_text + (size_t)l * 2 * D;
      const float* row = hz.kv_time + ((size_t)hz.step * m->cfg.num_layers + l) * 2 * D;
      AttnF32Args a{q32, D, kvt, kvt + D, m->cfg.num_layers * 2 * D, S, ntok, text_lengths, 0, B};
      a.kadd = row;
      a.vadd = row + D;
      a.kv_B = hz.kv_B;
      a.kv_b0 = hz.kv_b0;
      if (int rc = launch_attention_args(pf, a, nullptr, nseq_q, D, H, ws.atth, ws.attl, s)) return rc;
      vsrc = kvt + D; vadd = row + D; vstride = (size_t)ntok * m->cfg.num_layers * 2 * D; vseq0 = hz.kv_B + hz.kv_b0;
    }
    // ... and with it the whole block of those sequences is a row-constant: x' = norm1(y) + (Wo . v + bo).  Where the statistics
    // partials are 256 columns wide and D / 4 threads fit a workgroup (every latent_dim the route runs), one small GEMM makes the B
    // vectors o and uncond_xblock_rows_kernel writes the rows; the cross out_proj then covers the conditional half only.
    const bool xblock_uncond = skip_uncond && scols == 256 && D % 256 == 0 && D <= 1024;
    if (skip_uncond && !xblock_uncond) {
      ProfScope ps(pf, MDM_PROF_ELEMENTWISE, 0.0, s);
      MDM_LAUNCH(uncond_xattn_rows_kernel, dim3(B * ((S + 15) / 16)), dim3(D / 4 > 256 ? 256 : D / 4), 0, s, ws.atth, ws.attl, vsrc,
                 vstride, vadd, S, D, B, vseq0);
      if (int rc = rt_launch_status()) return rc;
    }
    if (xblock_uncond) {
      const float* o1; const float* o2 = nullptr;
      if (hoisted && hz.o_text != nullptr && hz.kv_b0 == 0 && hz.kv_B == B) {   // a window loop made the row constants once (loops.h)
        o1 = hz.o_text + (size_t)l * B * D;
        o2 = hz.o_time + ((size_t)l * hz.nsteps + hz.step) * D;
      } else {
        float* vs = q32 + (size_t)Mq * D;          // (the projected queries fill the conditional half of q32 only)
        float* ov = vs + (size_t)B * D;
        {
          ProfScope ps(pf, MDM_PROF_ELEMENTWISE, 0.0, s);
          MDM_LAUNCH(gather_value_rows_kernel, dim3(B), dim3(128), 0, s, vs, vsrc, vstride, vadd, D, vseq0);
          if (int rc = rt_launch_status()) return rc;
        }
        if (int rc = launch_linear(pf, vs, D, m->L(l, "multihead_attn.out_proj.weight"), m->L(l, "multihead_attn.out_proj.bias"), nullptr,
                                   ov, B, D, D, ACT_NONE, 0, 1.f, s)) return rc;
        o1 = ov;
      }
      ProfScope ps(pf, MDM_PROF_ELEMENTWISE, 0.0, s);
      MDM_LAUNCH(uncond_xblock_rows_kernel, dim3((Mq + 7) / 8), dim3(D / 4), 0, s, (const p16_t*)Yh, (const p16_t*)Yl, (const float*)sY,
                 m->L(l, "norm1.weight"), m->L(l, "norm1.bias"), o1, o2, Xh, Xl, sX, Mq, Mq, S, D, inv_dim);
      if (int rc = rt_launch_status()) return rc;
    }
    {
      LnArgs a = LN(); a.res = Y; a.rstat = sY; a.rgamma = m->L(l, "norm1.weight"); a.rbeta = m->L(l, "norm1.bias"); a.ostat = sX;
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 2, attp, P.out_proj2, m->L(l, "multihead_attn.out_proj.bias"), a, nullptr,
                                Xh, Xl, nullptr, xblock_uncond ? Mq : M, D, D, S, D, 0, 1.f, s)) return rc;
    }
    }   // !fused
    // ---- Y = norm2(X) + linear2(gelu(linear1(norm2(X))))
    {
      LnArgs a = LN(); a.astat = sX; a.colsum = F.c_1;
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 3, X, P.linear1, F.b_1, a, nullptr, ws.ffnh, ws.ffnl, nullptr, M, FF, D, S, D,
                                0, 1.f, s)) return rc;
    }
    {
      LnArgs a = LN(); a.res = X; a.rstat = sX; a.rgamma = m->L(l, "norm2.weight"); a.rbeta = m->L(l, "norm2.bias"); a.ostat = sY;
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 2, ffnp, P.linear2, m->L(l, "linear2.bias"), a, nullptr, Yh, Yl, nullptr, M, D, FF, S, D, 0, 1.f, s)) return rc;
    }
    cur ^= 1;   // the layer's output (pre-norm3) is the next layer's X
  }
  // ---- OutputProcess over the completed suffix (mdm.py:278-282) with the last norm3 folded in: every token's features as an
  // fp32 row (in the dead qkv region), then the transposing tail over token rows context_len .. S-1
  float* out_tok = ws.qkv;
  const int ldo = m->jf_out;
  ProfScope ps(pf, MDM_PROF_OUTPROJ, 2.0 * nseq * pred_len * (double)D * m->jf, s);
  {
    LnArgs a = LN(); a.astat = ws.stat[cur]; a.colsum = m->c_out;
    if (int rc = launch_x3_ln(nullptr, MDM_PROF_OUTPROJ, 4, X3Operand{ws.xh[cur], ws.xl[cur]}, m->out_planes_f, m->b_out, a, out_tok,
                              nullptr, nullptr, nullptr, M, ldo, D, S, D, 0, 1.f, s)) return rc;
  }
  if (tail != nullptr) {   // window loop: the step's sampler update in the tail kernel (the guidance branches are rows b and B + b)
    MDM_LAUNCH(outproj_finish_kernel, dim3((pred_len + 31) / 32, (m->jf + 31) / 32, B), dim3(256), 0, s, (const float*)out_tok,
               ldo, S, pred_len, m->jf, B, tail->scale, 1, tail->x, tail->x0_out, (const float*)tail->x, tail->ns,
               tail->inpaint_mask, tail->inpaint_motion, tail->co);
    tail->done = true;
    return rt_launch_status();
  }
  MDM_LAUNCH(outproj_finish_kernel, dim3((pred_len + 31) / 32, (m->jf + 31) / 32, nseq), dim3(256), 0, s, (const float*)out_tok,
             ldo, S, pred_len, m->jf, B, (const float*)nullptr, 0, out, (float*)nullptr, (const float*)nullptr, NoiseSource{},
             (const unsigned char*)nullptr, (const float*)nullptr, StepCoefs{});
  return rt_launch_status();
}

int decoder_pass(mdm_model_t* m, const DecWorkspace& ws, const float* x, const float* prefix, const int64_t* timesteps,
                 const float* text_tokens, const int32_t* text_lengths, const int32_t* lengths, int B, int pred_len,
                 int ntok, int branches, float* out, hipStream_t s, const DecHoist& hz, DecTail* tail = nullptr) {
  const int C = m->cfg.context_len, S = C + pred_len, D = m->cfg.latent_dim, H = m->cfg.num_heads, FF = m->cfg.ff_size;
  const int nbranch = (branches == MDM_BRANCH_BOTH) ? 2 : 1;
  const int nseq = nbranch * B, M = nseq * S, Mm = nseq * ntok;
  Profiler* pf = &m->prof;
  const int* len = m->cfg.mask_frames ? lengths : nullptr;
  const float qscale = 1.0f / sqrtf((float)ATT_HD);
  const bool x3 = m->precision == MDM_PREC_F16X3;   // GEMM arithmetic (gemm_f32.h X3); attention and LayerNorm statistics stay fp32
  const bool hoisted = hz.step >= 0;

  // ---- text memory: embed_text over every token (cond branch), + time embedding (mdm.py:217-219)
  if (!hoisted && branches != MDM_BRANCH_UNCOND)
    if (int rc = launch_linear(nullptr, text_tokens, m->cfg.clip_dim, m->W("embed_text.weight"), m->W("embed_text.bias"),
                               nullptr, ws.proj, ntok * B, D, m->cfg.clip_dim, ACT_NONE, 0, 1.f, s)) return rc;
  if (!hoisted) {
    ProfScope ps(pf, MDM_PROF_ELEMENTWISE, 0.0, s);
    MDM_LAUNCH(text_memory_kernel, dim3(Mm), dim3(128), 0, s, ws.mem, (const float*)ws.proj, m->W("embed_text.bias"),
               (const float*)m->time_table, reinterpret_cast<const long long*>(timesteps), B, ntok, D,
               (branches == MDM_BRANCH_UNCOND) ? 0 : 1, (int)m->cfg.max_len, m->time_add);
    if (int rc = rt_launch_status()) return rc;
  }
  if (dec_on_planes(m, M, S, hz, B))
    return decoder_layers_planes(m, ws, x, prefix, timesteps, text_lengths, len, B, pred_len, ntok, nbranch, out, s, hz,
                                 (tail != nullptr && (nbranch == 1) == (tail->scale == nullptr)) ? tail : nullptr);
  // ---- tgt tokens: InputProcess over cat(prefix, x) + positional rows (mdm.py:203-206, :239, :259-260); both branches
  {
    PoseGatherLoader al{x, S, m->jf, B * S, prefix, C};
    RowMajorLoader bl{m->w_in_pad, m->jf_pad, D, m->jf_pad};
    EmbedEpilogue ep{ws.tok, m->W("input_process.poseEmbedding.bias"), m->W("sequence_pos_encoder.pe"), B, S, S, D, nbranch,
                     nullptr, nullptr, 0};
    ProfScope ps(pf, MDM_PROF_EMBED, 2.0 * B * S * (double)D * m->jf, s);
    launch_gemm_f32(al, bl, ep, B * S, D, m->jf_pad, s, x3);
    if (int rc = rt_launch_status()) return rc;
  }
  if (int rc = dec_time_token_rows(m, ws.tok, nullptr, nullptr, timesteps, hz, B, nseq, S, s)) return rc;
  // ---- nn.TransformerDecoder (mdm.py:265; post-norm layers, no final norm).  The three LayerNorms of a layer are folded
  // into the GEMMs around them (gemm_f32.h LnFold): ws.tok holds the PRE-norm sums y, `pend` says which LayerNorm its readers
  // have to apply (none for the embedded tokens entering layer 0); only the last norm3 runs as a kernel, for OutputProcess.
  LnFold pend{};
  int sp = 0;
  auto fold_of = [&](int l, const char* norm) {
    LnFold f;
    f.stat = ws.stat[sp];
    f.gamma = m->L(l, (std::string(norm) + ".weight").c_str());
    f.beta = m->L(l, (std::string(norm) + ".bias").c_str());
    f.parts = D / LN_PART_COLS;
    f.inv_dim = 1.0f / (float)D;
    return f;
  };
  const LnFold none{};
  for (int l = 0; l < m->cfg.num_layers; ++l) {
    const mdm_model::DecFold& F = m->dec_fold[l];
    const mdm_model::DecPlanes& P = m->dec_planes[l];
    const bool folded = pend.stat != nullptr;   // false only for the embedded tokens entering layer 0
    // x = norm1(x + self_attn(x))
    if (int rc = launch_linear_lnfold(pf, ws.tok, D, pend, folded ? F.w_in : m->L(l, "self_attn.in_proj_weight"), P.in_proj,
                                      folded ? F.b_in : m->L(l, "self_attn.in_proj_bias"), folded ? F.c_in : nullptr, nullptr,
                                      none, ws.qkv, nullptr, M, 3 * D, D, ACT_NONE, D, qscale, s, x3)) return rc;
    // (hoisted: `lengths` is the WHOLE batch's array -- counts, then the ABI-7 bitmaps -- and this pass covers samples kv_b0 ..)
    if (int rc = launch_attention(pf, ws.qkv, ws.att, len, nseq, B, S, D, H, nullptr, nullptr, s, /*lead=*/0,
                                  hoisted ? hz.kv_B : 0, hoisted ? hz.kv_b0 : 0)) return rc;
    if (int rc = launch_linear_lnfold(pf, ws.att, D, none, m->L(l, "self_attn.out_proj.weight"), P.out_proj,
                                      m->L(l, "self_attn.out_proj.bias"), nullptr, ws.tok, pend, ws.tok, ws.stat[sp ^ 1], M, D, D,
                                      ACT_NONE, 0, 1.f, s, x3)) return rc;
    sp ^= 1;
    pend = fold_of(l, "norm1");
    // x = norm2(x + multihead_attn(x, memory, memory)): q from the tokens, k | v from the memory (packed in_proj rows)
    const float* wc = m->L(l, "multihead_attn.in_proj_weight");
    const float* bc = m->L(l, "multihead_attn.in_proj_bias");
    if (int rc = launch_linear_lnfold(pf, ws.tok, D, pend, F.w_q, P.q, F.b_q, F.c_q, nullptr, none, ws.qkv, nullptr, M, D, D,
                                      ACT_NONE, D, qscale, s, x3)) return rc;
    if (!hoisted) {
      if (int rc = launch_linear(pf, ws.mem, D, wc + (size_t)D * D, bc + D, nullptr, ws.kv, Mm, 2 * D, D, ACT_NONE, 0, 1.f, s, x3)) return rc;
      const AttnF32Args a{ws.qkv, D, ws.kv, ws.kv + D, 2 * D, S, ntok, text_lengths, 0, B};
      if (int rc = launch_attention_args(pf, a, ws.att, nseq, D, H, nullptr, nullptr, s)) return rc;
    } else {
      const float* kvt = hz.kv_text + (size_t)l * 2 * D;
      const float* row = hz.kv_time + ((size_t)hz.step * m->cfg.num_layers + l) * 2 * D;
      AttnF32Args a{ws.qkv, D, kvt, kvt + D, m->cfg.num_layers * 2 * D, S, ntok, text_lengths, 0, B};
      a.kadd = row;
      a.vadd = row + D;
      a.kv_B = hz.kv_B;
      a.kv_b0 = hz.kv_b0;
      if (int rc = launch_attention_args(pf, a, ws.att, nseq, D, H, nullptr, nullptr, s)) return rc;
    }
    if (int rc = launch_linear_lnfold(pf, ws.att, D, none, m->L(l, "multihead_attn.out_proj.weight"), P.out_proj2,
                                      m->L(l, "multihead_attn.out_proj.bias"), nullptr, ws.tok, pend, ws.tok, ws.stat[sp ^ 1], M,
                                      D, D, ACT_NONE, 0, 1.f, s, x3)) return rc;
    sp ^= 1;
    pend = fold_of(l, "norm2");
    // x = norm3(x + linear2(gelu(linear1(x))))
    if (int rc = launch_linear_lnfold(pf, ws.tok, D, pend, F.w_1, P.linear1, F.b_1, F.c_1, nullptr, none, ws.ffn, nullptr, M, FF,
                                      D, ACT_GELU, 0, 1.f, s, x3)) return rc;
    if (int rc = launch_linear_lnfold(pf, ws.ffn, FF, none, m->L(l, "linear2.weight"), P.linear2, m->L(l, "linear2.bias"), nullptr,
                                      ws.tok, pend, ws.tok, ws.stat[sp ^ 1], M, D, FF, ACT_NONE, 0, 1.f, s, x3)) return rc;
    sp ^= 1;
    pend = fold_of(l, "norm3");
  }
  if (pend.stat != nullptr)
    if (int rc = launch_layernorm(pf, ws.tok, pend.gamma, pend.beta, M, D, nullptr, nullptr, s)) return rc;
  // ---- OutputProcess over the completed suffix (mdm.py:278-282): token rows context_len .. S-1 of every sequence
  RowMajorLoader al{m->W("output_process.poseFinal.weight"), D, m->jf, D};
  CfgTokenLoader bl{ws.tok, nullptr, nseq, pred_len, S, D, nseq * pred_len, C};
  OutProjEpilogue ep{};
  ep.bias = m->W("output_process.poseFinal.bias");
  ep.out = out;
  ep.T = pred_len; ep.JF = m->jf; ep.mode = 0;
  ProfScope ps(pf, MDM_PROF_OUTPROJ, 2.0 * nseq * pred_len * (double)D * m->jf, s);
  launch_gemm_f32(al, bl, ep, m->jf, nseq * pred_len, D, s, x3, /*weight_is_a=*/true);
  return rt_launch_status();
}
}  // namespace

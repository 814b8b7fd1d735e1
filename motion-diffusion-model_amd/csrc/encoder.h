// encoder.h -- part of the ONE translation unit csrc/mdm_api.hip (the C ABI of libmdm_hip.so); split out of it in round 6
// (source health, no behaviour change).  MDM.forward for arch='trans_enc' (model/mdm.py:189-283): InputProcess, the 8 post-norm encoder layers, OutputProcess.
#pragma once

namespace {

// InputProcess in split precision (8-wave kernel): poses -> planes [B*T][jf_k] (in the dead ffn region) -> GEMM whose epilogue
// adds the positional rows and writes the frame tokens of every branch as planes.
// `ct`: the condition tokens of the nbranch * B sequences are written by the same launch as the pose transpose.
int embed_frames_x3(mdm_model* m, const Workspace& ws, const float* x, int B, int T, int nbranch, hipStream_t s, const CondTokArgs& ct) {
  const int D = m->cfg.latent_dim, KP = m->jf_k;
  ProfScope ps(&m->prof, MDM_PROF_EMBED, 2.0 * B * T * (double)D * m->jf, s);
  p16_t* ph = reinterpret_cast<p16_t*>(ws.ffn);
  p16_t* pl = ph + (size_t)B * T * KP;
  MDM_LAUNCH(pose_planes_cond_kernel, dim3(((T + 31) / 32) * (KP / 32) * B + nbranch * B), dim3(256), 0, s, x, ph, pl, T, m->jf, KP, B, ct);
  if (int rc = rt_launch_status()) return rc;
  LnArgs a;
  a.res_f32 = m->W("sequence_pos_encoder.pe");
  a.emb_T = T; a.emb_B = B; a.emb_nbranch = nbranch;
  a.small = use_small_gemm(m, nbranch * B, T + 1) && KP == 288;
  a.shape = x3s_shape(m->x3s, nbranch * B);
  return launch_x3_ln(nullptr, MDM_PROF_EMBED, 5, X3Operand{ph, pl}, m->in_planes, m->W("input_process.poseEmbedding.bias"), a,
                      nullptr, ws.tokh, ws.tokl, nullptr, B * T, D, KP, T + 1, D, 0, 1.f, s);
}
inline bool use_embed_x3(const mdm_model* m, int T) {
  // (longer sequences: the row-tile form of the same GEMM where it exists -- 263 features -- else the fp32-operand embedding below)
  return m->precision == MDM_PREC_F16X3 && x3_waves_setting() == 8 &&
         (T + 1 <= X3_TM || (use_small_gemm(m, 1, T + 1) && m->jf_k == 288));
}

// Tokens for every sequence: frame tokens via the InputProcess GEMM, token 0 via cond_token_kernel.
int embed_tokens(mdm_model* m, const Workspace& ws, const float* x, const long long* timesteps,
                 long long t_uniform_unused, const float* cond_emb, int B, int T, int nbranch,
                 int uncond_from_branch, hipStream_t s) {
  (void)t_uniform_unused;
  const int D = m->cfg.latent_dim, S = T + 1;
  PoseGatherLoader al{x, T, m->jf, B * T};
  RowMajorLoader bl{m->w_in_pad, m->jf_pad, D, m->jf_pad};
  const bool x3 = m->precision == MDM_PREC_F16X3;
  EmbedEpilogue ep{ws.tok, m->W("input_process.poseEmbedding.bias"), m->W("sequence_pos_encoder.pe"), B, T, S, D,
                   nbranch, x3 ? ws.tokh : nullptr, x3 ? ws.tokl : nullptr};
  if (use_embed_x3(m, T)) {   // token 0 of every sequence rides in the transpose kernel of the frame embedding: one launch fewer per step
    const CondTokArgs ct{ws.tok, cond_emb, m->W("embed_text.bias"), m->time_table, timesteps, 0, m->W("sequence_pos_encoder.pe"), B, S, D,
                         uncond_from_branch, (int)m->cfg.max_len, ws.tokh, ws.tokl, m->time_add};
    return embed_frames_x3(m, ws, x, B, T, nbranch, s, ct);
  } else {
    ProfScope ps(&m->prof, MDM_PROF_EMBED, 2.0 * B * T * (double)D * m->jf, s);
    launch_gemm_f32(al, bl, ep, B * T, D, m->jf_pad, s);
  }
  if (int rc = rt_launch_status()) return rc;
  ProfScope ps(&m->prof, MDM_PROF_ELEMENTWISE, 0.0, s);
  MDM_LAUNCH(cond_token_kernel, dim3(nbranch * B), dim3(128), 0, s, ws.tok, cond_emb, m->W("embed_text.bias"),
             (const float*)m->time_table, timesteps, 0, m->W("sequence_pos_encoder.pe"), B, S, D, uncond_from_branch,
             (int)m->cfg.max_len, x3 ? ws.tokh : (p16_t*)nullptr, x3 ? ws.tokl : (p16_t*)nullptr, m->time_add);
  return rt_launch_status();
}

// seqTransEncoder: num_layers post-norm layers over ws.tok [nseq*S, D] (in place).
int encoder(mdm_model* m, const Workspace& ws, int nseq, int B, int S, const int* lengths, hipStream_t s) {
  Profiler* pf = &m->prof;
  const int D = m->cfg.latent_dim, FF = m->cfg.ff_size, H = m->cfg.num_heads, M = nseq * S;
  const float qscale = 1.0f / sqrtf((float)(D / H));
  if (m->precision == MDM_PREC_F16X3 && m->lnfold && x3_waves_setting() == 8 && (S <= X3_TM || use_small_gemm(m, nseq, S))) {
    // No LayerNorm kernels: xb = tokh|tokl holds the layer input / the post-FFN PRE-norm sum, xa the post-attention
    // pre-norm sum, each with per-row partial (sum, sum^2) written by its producer; consumers fold the normalisation
    // (gemm_x3.h X3Epilogue).  Layer 0's input (the embedding) is not normalised: plain in_proj, plain residual.
    const X3Operand xb{ws.tokh, ws.tokl}, xa{ws.xah, ws.xal}, attp{ws.atth, ws.attl}, ffnp{ws.ffnh, ws.ffnl};
    // few sequences: the latency regime -- every GEMM of the stack on gemm_x3s.h's 32 / 64-row tiles (row statistics per 128
    // columns); else gemm_x3.h's sequence-sized tiles (per 256)
    const bool small = use_small_gemm(m, nseq, S);
    const X3sShape shape = x3s_shape(m->x3s, nseq);
    const int scols = small ? x3s_tn(shape.ncb) : 256;
    const int parts = (D + scols - 1) / scols;
    const float inv_dim = 1.0f / (float)D;
    auto LN = [&]() { LnArgs a; a.small = small; a.shape = shape; a.stat_cols = scols; a.parts = parts; a.inv_dim = inv_dim; return a; };
    // (Running the stack over two half-batches, so that every producer -> consumer hand-over stays inside the 256 MB Infinity
    // Cache, was built and measured: 1.5 % SLOWER on the same box -- profiles/r02_ab.md -- and removed.)
    // (Running in_proj -> attention and / or linear1 -> linear2 one guidance branch at a time, so that the 352 MB of Q / K / V^T
    // planes or the 207 MB of GELU planes stay inside the 256 MB Infinity Cache between producer and consumer, was built and
    // measured in round 4: attention 2 x 59.4 us against 111.5, in_proj 2 x 132.1 against 257.3, whole loop 1.0-1.5 % SLOWER on the
    // same box -- profiles/r04i_halves.md -- and removed.)
    // (Running the batch as TWO concurrent half-batch chains on two streams, each GEMM launch on half the CUs, so that one chain's
    // epilogue store bursts fall into the other's k-loops: round 5, probe-library hooks MDM_CHAIN_FREE / MDM_X3_GRID_DIV,
    // lab/probes/two_chains.py -- 3.5 % SLOWER, bit-identical results: profiles/r05l_two_chains.md.)
    for (int l = 0; l < m->cfg.num_layers; ++l) {
      const mdm_model::LayerPlanes& P = m->planes[l];
      const mdm_model::LayerFold& F = m->fold[l];
      if (l == 0 && small) {
        LnArgs a = LN();
        if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 6, xb, P.in_proj, m->L(l, "self_attn.in_proj_bias"), a, nullptr, nullptr,
                                  nullptr, &ws.qp, M, 3 * D, D, S, D, D, qscale, s)) return rc;
      } else if (l == 0) {
        if (int rc = launch_in_proj_x3(pf, xb, P.in_proj, m->L(l, "self_attn.in_proj_bias"), ws.qp, nseq, S, D, qscale, s)) return rc;
      } else {
        LnArgs a = LN(); a.astat = ws.stat2; a.colsum = F.c_qkv;
        if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 0, xb, F.in_proj, F.b_qkv, a, nullptr, nullptr, nullptr, &ws.qp, M,
                                  3 * D, D, S, D, D, qscale, s)) return rc;
      }
      if (int rc = launch_attention_x3(pf, ws.qp, lengths, nseq, B, S, D, nullptr, ws.atth, ws.attl, s, 1, m->attn_direct)) return rc;
      {  // xa = att.Wo + bo + layer input (normalised on the fly for l >= 1), + row statistics
        LnArgs a = LN(); a.res = xb; a.ostat = ws.stat1;
        if (l >= 1) { a.rstat = ws.stat2; a.rgamma = m->L(l - 1, "norm2.weight"); a.rbeta = m->L(l - 1, "norm2.bias"); }
        if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, l == 0 ? 1 : 2, attp, P.out_proj, m->L(l, "self_attn.out_proj.bias"),
                                  a, nullptr, ws.xah, ws.xal, nullptr, M, D, D, S, D, 0, 1.f, s)) return rc;
      }
      {  // ffn = gelu(LN1(xa).W1 + b1), LN1 folded
        LnArgs a = LN(); a.astat = ws.stat1; a.colsum = F.c_1;
        if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 3, xa, F.linear1, F.b_1, a, nullptr, ws.ffnh, ws.ffnl, nullptr, M, FF,
                                  D, S, D, 0, 1.f, s)) return rc;
      }
      {  // xb = ffn.W2 + b2 + LN1(xa), + row statistics
        LnArgs a = LN(); a.res = xa; a.rstat = ws.stat1; a.rgamma = m->L(l, "norm1.weight"); a.rbeta = m->L(l, "norm1.bias");
        a.ostat = ws.stat2;
        if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 2, ffnp, P.linear2, m->L(l, "linear2.bias"), a, nullptr, ws.tokh,
                                  ws.tokl, nullptr, M, D, FF, S, D, 0, 1.f, s)) return rc;
      }
    }
    return 0;   // the encoder's output is LN2(L-1)(xb): folded into OutputProcess (outproj_x3)
  }
  if (m->precision == MDM_PREC_F16X3) {
    // tok (fp32, residual stream) travels with its split planes tokh/tokl; attention and GELU outputs exist only as planes
    const X3Operand tokp{ws.tokh, ws.tokl}, attp{ws.atth, ws.attl}, ffnp{ws.ffnh, ws.ffnl};
    for (int l = 0; l < m->cfg.num_layers; ++l) {
      const mdm_model::LayerPlanes& P = m->planes[l];
      if (int rc = launch_in_proj_x3(pf, tokp, P.in_proj, m->L(l, "self_attn.in_proj_bias"), ws.qp, nseq, S, D, qscale, s)) return rc;
      if (int rc = launch_attention_x3(pf, ws.qp, lengths, nseq, B, S, D, nullptr, ws.atth, ws.attl, s, 1, m->attn_direct)) return rc;
      // the residual stream lives as planes only (value = hi + lo): the GEMM writes the pre-norm sum as fp32, LayerNorm
      // turns it back into planes and does NOT write fp32 (one 103 MB stream less per LayerNorm)
      if (int rc = launch_linear_x3(pf, attp, P.out_proj, m->L(l, "self_attn.out_proj.bias"), nullptr, ws.tok, nullptr,
                                    nullptr, M, D, D, ACT_NONE, 0, 1.f, S, s, tokp)) return rc;
      if (int rc = launch_layernorm(pf, ws.tok, m->L(l, "norm1.weight"), m->L(l, "norm1.bias"), M, D, ws.tokh, ws.tokl, s, false)) return rc;
      if (int rc = launch_linear_x3(pf, tokp, P.linear1, m->L(l, "linear1.bias"), nullptr, nullptr, ws.ffnh, ws.ffnl, M,
                                    FF, D, ACT_GELU, 0, 1.f, S, s)) return rc;
      if (int rc = launch_linear_x3(pf, ffnp, P.linear2, m->L(l, "linear2.bias"), nullptr, ws.tok, nullptr, nullptr, M, D,
                                    FF, ACT_NONE, 0, 1.f, S, s, tokp)) return rc;
      if (int rc = launch_layernorm(pf, ws.tok, m->L(l, "norm2.weight"), m->L(l, "norm2.bias"), M, D, ws.tokh, ws.tokl, s, false)) return rc;
    }
    return 0;
  }
  for (int l = 0; l < m->cfg.num_layers; ++l) {
    if (int rc = launch_linear(pf, ws.tok, D, m->L(l, "self_attn.in_proj_weight"), m->L(l, "self_attn.in_proj_bias"),
                               nullptr, ws.qkv, M, 3 * D, D, ACT_NONE, D, qscale, s)) return rc;
    if (int rc = launch_attention(pf, ws.qkv, ws.att, lengths, nseq, B, S, D, H, nullptr, nullptr, s)) return rc;
    if (int rc = launch_linear(pf, ws.att, D, m->L(l, "self_attn.out_proj.weight"), m->L(l, "self_attn.out_proj.bias"),
                               ws.tok, ws.tok, M, D, D, ACT_NONE, 0, 1.f, s)) return rc;
    if (int rc = launch_layernorm(pf, ws.tok, m->L(l, "norm1.weight"), m->L(l, "norm1.bias"), M, D, nullptr, nullptr, s)) return rc;
    if (int rc = launch_linear(pf, ws.tok, D, m->L(l, "linear1.weight"), m->L(l, "linear1.bias"), nullptr, ws.ffn, M,
                               FF, D, ACT_GELU, 0, 1.f, s)) return rc;
    if (int rc = launch_linear(pf, ws.ffn, FF, m->L(l, "linear2.weight"), m->L(l, "linear2.bias"), ws.tok, ws.tok, M,
                               D, FF, ACT_NONE, 0, 1.f, s)) return rc;
    if (int rc = launch_layernorm(pf, ws.tok, m->L(l, "norm2.weight"), m->L(l, "norm2.bias"), M, D, nullptr, nullptr, s)) return rc;
  }
  return 0;
}

// OutputProcess, split precision: every sequence's tokens x poseFinal -> fp32 rows in the (dead) qkv region, then the
// transposing / fusing tail kernel (elementwise.h outproj_finish_kernel).
int outproj_x3(mdm_model* m, const Workspace& ws, int nseq, int B, int T, const float* scale, int mode, float* out,
               float* x0_out, const float* x_t, NoiseSource noise, const unsigned char* inpaint_mask,
               const float* inpaint_motion, StepCoefs co, hipStream_t s) {
  const int D = m->cfg.latent_dim, S = T + 1, ldo = m->jf_out;
  float* out_tok = ws.qkv;
  ProfScope ps(&m->prof, MDM_PROF_OUTPROJ, 2.0 * nseq * T * (double)D * m->jf, s);
  if (m->lnfold && x3_waves_setting() == 8 && (S <= X3_TM || use_small_gemm(m, nseq, S))) {   // the final LayerNorm is folded into this GEMM
    LnArgs a; a.astat = ws.stat2; a.colsum = m->c_out; a.inv_dim = 1.0f / (float)D;
    a.small = use_small_gemm(m, nseq, S);            // (the same decision the encoder took: who wrote stat2)
    a.shape = x3s_shape(m->x3s, nseq);
    a.stat_cols = a.small ? x3s_tn(a.shape.ncb) : 256;
    a.parts = (D + a.stat_cols - 1) / a.stat_cols;
    if (int rc = launch_x3_ln(nullptr, MDM_PROF_OUTPROJ, 4, X3Operand{ws.tokh, ws.tokl}, m->out_planes_f, m->b_out, a,
                              out_tok, nullptr, nullptr, nullptr, nseq * S, ldo, D, S, D, 0, 1.f, s)) return rc;
  } else if (int rc = launch_linear_x3(nullptr, X3Operand{ws.tokh, ws.tokl}, m->out_planes, m->out_bias_pad, nullptr, out_tok,
                                       nullptr, nullptr, nseq * S, ldo, D, ACT_NONE, 0, 1.f, S, s)) return rc;
  const int nb = (mode == 1) ? B : nseq;
  MDM_LAUNCH(outproj_finish_kernel, dim3((T + 31) / 32, (m->jf + 31) / 32, nb), dim3(256), 0, s, (const float*)out_tok,
             ldo, S, T, m->jf, B, scale, mode, out, x0_out, x_t, noise, inpaint_mask, inpaint_motion, co);
  return rt_launch_status();
}

}  // namespace

// api_launch.h -- part of the ONE translation unit csrc/mdm_api.hip (the C ABI of libmdm_hip.so); split out of it in round 6
// (source health, no behaviour change).  The encoder workspace and the launchers of the building-block kernels (LayerNorm, attention, the GEMM families).
#pragma once

namespace {

struct Workspace {
  float *tok, *qkv, *att, *ffn, *cond;
  QkvPlanes qp;         // f16x3 mode: the in_proj epilogue writes Q/K/V^T planes over the qkv region
  p16_t *xah, *xal;    // folded-LayerNorm mode: planes of the post-attention pre-norm sum (alias tok)
  float *stat1, *stat2; // folded-LayerNorm mode: per-row partial (sum, sum^2) of xa / of tokh|tokl
  p16_t *tokh, *tokl;  // split planes of tok (f16x3 mode)
  p16_t *atth, *attl;  // alias att: the attention output is only consumed by the out_proj GEMM
  p16_t *ffnh, *ffnl;  // alias ffn: the GELU output is only consumed by the linear2 GEMM
  size_t bytes;
};

Workspace carve(const mdm_model* m, int nseq, int T, void* base) {
  const size_t D = m->cfg.latent_dim, FF = m->cfg.ff_size, S = (size_t)T + 1, M = (size_t)nseq * S;
  size_t off = 0;
  auto take = [&](size_t floats) {
    size_t o = off;
    off += align_up(floats * sizeof(float), 256);
    return base ? reinterpret_cast<float*>(static_cast<char*>(base) + o) : nullptr;
  };
  Workspace w;
  w.tok = take(M * D);
  const size_t NKT = (S + 31) / 32, SP = 32 * NKT;
  w.qkv = take((size_t)nseq * SP * 3 * D);  // fp32 [M][3D] (f32 mode) or six 16-bit planes of nseq*SP*D (f16x3 mode)
  w.att = take(M * D);
  w.ffn = take(M * FF);
  w.cond = take((size_t)nseq * D);
  float* tp = take(M * D);  // two 16-bit planes = one fp32 array's worth of bytes
  w.tokh = reinterpret_cast<p16_t*>(tp);
  w.tokl = tp ? w.tokh + M * D : nullptr;
  w.atth = reinterpret_cast<p16_t*>(w.att);
  w.attl = w.att ? w.atth + M * D : nullptr;
  w.ffnh = reinterpret_cast<p16_t*>(w.ffn);
  w.ffnl = w.ffn ? w.ffnh + M * FF : nullptr;
  w.xah = reinterpret_cast<p16_t*>(w.tok);
  w.xal = w.tok ? w.xah + M * D : nullptr;
  const size_t parts = (D + 127) / 128;   // per-row partial statistics: per 256 columns (gemm_x3.h) or per 128 (gemm_x3s.h)
  w.stat1 = take(M * parts * 2 + 4);   // (+ 16 bytes: gemm_x3.h stats_dma's last unit of a tile that starts 8-byte aligned)
  w.stat2 = take(M * parts * 2 + 4);
  {
    const size_t plane = (size_t)nseq * SP * D;
    p16_t* q = reinterpret_cast<p16_t*>(w.qkv);
    w.qp = QkvPlanes{q, q ? q + plane : nullptr, q ? q + 2 * plane : nullptr, q ? q + 3 * plane : nullptr,
                     q ? q + 4 * plane : nullptr, q ? q + 5 * plane : nullptr, (int)SP, (int)NKT, m->cfg.num_heads};
  }
  w.bytes = off;
  return w;
}

int launch_layernorm(Profiler* pf, float* x, const float* g, const float* b, int rows, int D, p16_t* xh, p16_t* xl,
                     hipStream_t s, bool write_f32 = true) {
  ProfScope ps(pf, MDM_PROF_LAYERNORM, 0.0, s);
  const dim3 grid((rows + 3) / 4), block(256);
  switch (D / 256) {
    case 1: { auto k = &layernorm_kernel<1>; MDM_LAUNCH(k, grid, block, 0, s, x, g, b, rows, 1e-5f, xh, xl, (int)write_f32); break; }
    case 2: { auto k = &layernorm_kernel<2>; MDM_LAUNCH(k, grid, block, 0, s, x, g, b, rows, 1e-5f, xh, xl, (int)write_f32); break; }
    case 3: { auto k = &layernorm_kernel<3>; MDM_LAUNCH(k, grid, block, 0, s, x, g, b, rows, 1e-5f, xh, xl, (int)write_f32); break; }
    case 4: { auto k = &layernorm_kernel<4>; MDM_LAUNCH(k, grid, block, 0, s, x, g, b, rows, 1e-5f, xh, xl, (int)write_f32); break; }
    default: return fail(MDM_EUNSUPPORTED, "layernorm: D must be 256, 512, 768 or 1024");
  }
  return rt_launch_status();
}

template <int NKT>
int launch_attention_t(const AttnF32Args& a, float* out, int nseq, int D, int H, p16_t* oh, p16_t* ol, hipStream_t s) {
  auto k = &attention_f32_kernel<NKT>;
  const int nqt = (a.Sq + 31) / 32;
  const size_t lds = attention_lds_bytes(NKT, nqt);
  if (int rc = rt_allow_lds(k, lds)) return rc;
  MDM_LAUNCH(k, dim3(nseq * H), dim3(64 * nqt), lds, s, a, out, D, H, oh, ol);
  return rt_launch_status();
}

// exact-fp32 attention with separate query / key-value sources (attention_f32.h AttnF32Args)
int launch_attention_args(Profiler* pf, const AttnF32Args& a, float* out, int nseq, int D, int H, p16_t* oh, p16_t* ol,
                          hipStream_t s) {
  ProfScope ps(pf, MDM_PROF_ATTENTION, 4.0 * nseq * H * (double)a.Sq * a.Sk * ATT_HD, s);
  if (D != H * ATT_HD) return fail(MDM_EUNSUPPORTED, "attention: head_dim must be 128");
  if (a.Sq < 1 || a.Sk < 1) return fail(MDM_EUNSUPPORTED, "attention: no tokens");
  if (a.Sq > 224 || a.Sk > 224) {   // streaming softmax over 32-key tiles (attention_long.h): any length
    const int nqb = al_query_blocks(a.Sq);
    MDM_LAUNCH(attention_f32_long_kernel, dim3(nseq * H * nqb), dim3(256), al_f32_lds_bytes(), s, a, out, D, H, oh, ol, nqb);
    return rt_launch_status();
  }
  switch ((a.Sk + 31) / 32) {
    case 1: return launch_attention_t<1>(a, out, nseq, D, H, oh, ol, s);
    case 2: return launch_attention_t<2>(a, out, nseq, D, H, oh, ol, s);
    case 3: return launch_attention_t<3>(a, out, nseq, D, H, oh, ol, s);
    case 4: return launch_attention_t<4>(a, out, nseq, D, H, oh, ol, s);
    case 5: return launch_attention_t<5>(a, out, nseq, D, H, oh, ol, s);
    case 6: return launch_attention_t<6>(a, out, nseq, D, H, oh, ol, s);
    default: return launch_attention_t<7>(a, out, nseq, D, H, oh, ol, s);
  }
}

// self-attention over packed qkv rows [nseq*S][3D]; `lead` tokens in front of the frames are never masked
int launch_attention(Profiler* pf, const float* qkv, float* out, const int* lengths, int nseq, int B, int S, int D,
                     int H, p16_t* oh, p16_t* ol, hipStream_t s, int lead = 1, int len_B = 0, int len_b0 = 0) {
  AttnF32Args a{qkv, 3 * D, qkv + D, qkv + 2 * D, 3 * D, S, S, lengths, lead, B};
  a.len_B = len_B;     // `lengths` covers len_B samples, this launch samples len_b0 .. len_b0 + B - 1 of them (0: exactly B)
  a.len_b0 = len_b0;
  return launch_attention_args(pf, a, out, nseq, D, H, oh, ol, s);
}

#ifdef MDM_PROBES
int g_ax_ablate = 0;   // mdm_debug_set(3, code): timing experiments on the NKT = 7 attention kernel (attention_x3.h ABL)
#endif
template <int NKT, int ABL = 0, bool DIRECT = false>
int launch_attention_x3_t(const QkvPlanes& qp, const int* lengths, int nseq, int B, int S, int D, float* out, p16_t* oh,
                          p16_t* ol, hipStream_t s, int lead) {
  auto k = &attention_x3_kernel<NKT, ABL, DIRECT>;
  const size_t lds = attention_x3_lds_bytes(NKT);
  if (int rc = rt_allow_lds(k, lds)) return rc;
  // two workgroups (query halves) per (sequence, head); the item <-> block mapping pairs blocks b and b + 8 (same XCD),
  // so the number of items is rounded up to a multiple of 8 and surplus workgroups exit
  // persistent workgroups, two per CU (the grid stays a multiple of 16 so that a workgroup keeps its query half)
  const int items = nseq * qp.H, groups = (items + 7) / 8;
  const int grid = std::min(groups * 16, std::max(16, x3_grid_limit(2) / 16 * 16));
  MDM_LAUNCH(k, dim3(grid), dim3(256), lds, s, qp, lengths, S, D, B, lead, out, oh, ol, items);
  return rt_launch_status();
}

// split-precision attention on the operand planes written by the in_proj epilogue (or qkv_pack_kernel)
// `lead` tokens in front of the frames are never masked (trans_enc: the condition token; trans_dec: none -- its `lengths` count the
// context_len prefix frames as frames)
int launch_attention_x3(Profiler* pf, const QkvPlanes& qp, const int* lengths, int nseq, int B, int S, int D, float* out,
                        p16_t* oh, p16_t* ol, hipStream_t s, int lead = 1, bool direct = false) {
  ProfScope ps(pf, MDM_PROF_ATTENTION, 4.0 * nseq * qp.H * (double)S * S * AX_HD, s);
  if (D != qp.H * AX_HD) return fail(MDM_EUNSUPPORTED, "attention: head_dim must be 128");
  if (S < 1) return fail(MDM_EUNSUPPORTED, "attention: no tokens");
  if (S > 224) {   // streaming softmax over the same operand planes (attention_long.h): any length
    if (qp.NKT != (S + 31) / 32 || qp.SP != 32 * qp.NKT) return fail(MDM_EINVAL, "attention: the operand planes do not match the sequence length");
    const int nqb = al_query_blocks(S);
    MDM_LAUNCH(attention_x3_long_kernel, dim3(nseq * qp.H * nqb), dim3(256), al_x3_lds_bytes(), s, qp, lengths, S, D, B, lead, out, oh, ol, nqb);
    return rt_launch_status();
  }
  if (direct && out == nullptr && oh != nullptr) {   // planes straight from the accumulators (attention_x3.h DIRECT)
    switch (qp.NKT) {
      case 1: return launch_attention_x3_t<1, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
      case 2: return launch_attention_x3_t<2, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
      case 3: return launch_attention_x3_t<3, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
      case 4: return launch_attention_x3_t<4, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
      case 5: return launch_attention_x3_t<5, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
      case 6: return launch_attention_x3_t<6, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
      default: return launch_attention_x3_t<7, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    }
  }
  switch (qp.NKT) {
    case 1: return launch_attention_x3_t<1>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    case 2: return launch_attention_x3_t<2>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    case 3: return launch_attention_x3_t<3>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    case 4: return launch_attention_x3_t<4>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    case 5: return launch_attention_x3_t<5>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    case 6: return launch_attention_x3_t<6>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    default:
#ifdef MDM_PROBES
      static const int env_abl = [] { const char* e = getenv("MDM_AX_ABL"); return e != nullptr ? atoi(e) : 0; }();   // whole-bench A/B runs
      switch (g_ax_ablate != 0 ? g_ax_ablate : env_abl) {
        case 1: return launch_attention_x3_t<7, 1>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 2: return launch_attention_x3_t<7, 2>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 3: return launch_attention_x3_t<7, 3>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 4: return launch_attention_x3_t<7, 4>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 8: return launch_attention_x3_t<7, 8>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 16: return launch_attention_x3_t<7, 16>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 32: return launch_attention_x3_t<7, 32>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 48: return launch_attention_x3_t<7, 48>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 63: return launch_attention_x3_t<7, 63>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 64: return launch_attention_x3_t<7, 64>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 128: return launch_attention_x3_t<7, 128>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 192: return launch_attention_x3_t<7, 192>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        default: break;
      }
#endif
      return launch_attention_x3_t<7>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
  }
}

#ifdef MDM_PROBES
// TEST-ONLY mode (mdm_debug_set(5, 1)): the `f32` mode's encoder GEMMs on the f16f6 kernel, UNFUSED -- operands packed per
// call into a library-owned scratch (the one exception to "the caller owns every buffer": a debug path) -- so that the
// f16f6 arithmetic can be held against the reference's golden trajectories through the product's own seams before the fused
// path exists.  Not a performance path.
int g_f6_linear = 0;
void* g_f6_dbg_scratch = nullptr;
size_t g_f6_dbg_bytes = 0;
int launch_linear_f6_debug(Profiler* pf, const float* in, int ld_in, const float* w, const float* bias, const float* res,
                           float* out, int M, int N, int K, int act, int scale_cols, float col_scale, hipStream_t s) {
  const size_t wfrag = align_up(x3_packed_weight_elems(N, K) * 2, 256);
  const size_t need = f6_plane_bytes(M, K) + 2 * wfrag;
  if (need > g_f6_dbg_bytes) {
#ifdef MDM_EMU
    free(g_f6_dbg_scratch);
    g_f6_dbg_scratch = malloc(need);
#else
    if (hipDeviceSynchronize() != hipSuccess) return fail(MDM_EHIP, "f16f6 debug mode: synchronize failed");
    if (g_f6_dbg_scratch != nullptr) (void)hipFree(g_f6_dbg_scratch);
    if (hipMalloc(&g_f6_dbg_scratch, need) != hipSuccess) { g_f6_dbg_scratch = nullptr; g_f6_dbg_bytes = 0; return fail(MDM_EHIP, "f16f6 debug mode: hipMalloc failed"); }
#endif
    g_f6_dbg_bytes = need;
  }
  ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * M * (double)N * K, s);
  char* base = static_cast<char*>(g_f6_dbg_scratch);
  const F6Planes pa = f6_carve(base, M, K);
  p16_t* wfh = reinterpret_cast<p16_t*>(base + f6_plane_bytes(M, K));
  p16_t* wfl = reinterpret_cast<p16_t*>(base + f6_plane_bytes(M, K) + wfrag);
  MDM_LAUNCH(pack_f16f6_kernel, dim3((M * (K / 32) + 255) / 256), dim3(256), 0, s, in, pa, M, K, ld_in);
  if (int rc = rt_launch_status()) return rc;
  const int npad = (N + 31) / 32 * 32;
  MDM_LAUNCH(pack_weight_f16f6_kernel, dim3((npad * (K / 32) + 255) / 256), dim3(256), 0, s, w, wfh, wfl, N, K);
  if (int rc = rt_launch_status()) return rc;
  X3Epilogue ep{out, bias, res, nullptr, nullptr, nullptr, nullptr, N, scale_cols, col_scale, QkvPlanes{}, 0, 0,
                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1.f, 1, 1, 1, 1.f};
  const X3Operand a{reinterpret_cast<const p16_t*>(pa.h16), reinterpret_cast<const p16_t*>(pa.rec)};
  const int rc = launch_gemm_f16f6(a, X3Weights{wfh, wfl}, ep, M, N, K, act, s);
  if (rc != 0) return fail(MDM_EUNSUPPORTED, "f16f6 debug mode: launch failed");
  return rt_launch_status();
}
#endif

// x3: the split-precision arithmetic on this (fp32-in-memory) skeleton -- the DiP decoder's GEMMs in the f16x3 mode
int launch_linear(Profiler* pf, const float* in, int ld_in, const float* w, const float* bias, const float* res,
                  float* out, int M, int N, int K, int act, int scale_cols, float col_scale, hipStream_t s, bool x3 = false) {
#ifdef MDM_PROBES
  if (g_f6_linear && K % 32 == 0 && N % 4 == 0 && ld_in % 4 == 0 && (scale_cols % 256 == 0) &&
      (act == ACT_NONE || (act == ACT_GELU && res == nullptr)))
    return launch_linear_f6_debug(pf, in, ld_in, w, bias, res, out, M, N, K, act, scale_cols, col_scale, s);
#endif
  ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * M * (double)N * K, s);
  if (K % 4 != 0 || ld_in % 4 != 0) return fail(MDM_EINVAL, "linear: K and the row stride must be multiples of 4");
  RowMajorLoader al{in, ld_in, M, K};
  RowMajorLoader bl{w, K, N, K};
  LinearEpilogue ep{out, bias, res, N, act, scale_cols, col_scale, nullptr, nullptr};
  launch_gemm_f32(al, bl, ep, M, N, K, s, x3);
  return rt_launch_status();
}

// linear with LayerNorms folded in (gemm_f32.h LnLinearEpilogue): `a_ln` set = the A operand is a pre-norm sum and w / bias are
// the gamma-folded ones with column sums `colsum`; res_ln set = the residual is LN(res); ostat = where the partial statistics
// of the written rows go (or null)
// (x3: the weights come as the fragment-ordered planes `wp` of the same matrix, gemm_f32.h X3FragB)
int launch_linear_lnfold(Profiler* pf, const float* in, int ld_in, const LnFold& a_ln, const float* w, X3Weights wp,
                         const float* bias, const float* colsum, const float* res, const LnFold& res_ln, float* out,
                         float* ostat, int M, int N, int K, int act, int scale_cols, float col_scale, hipStream_t s, bool x3) {
  ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * M * (double)N * K, s);
  if (K % 16 != 0 || ld_in % 4 != 0 || N % LN_PART_COLS != 0) return fail(MDM_EINVAL, "linear (LayerNorm fold): bad K / N");
  RowMajorLoader al{in, ld_in, M, K};
  LnLinearEpilogue ep{out, bias, N, act, scale_cols, col_scale, a_ln, colsum, res, res_ln, ostat};
#ifdef MDM_PROBES   // A/B switch of the probe library: MDM_DEC_FRAGB=0 sends the layer weights through the fp32 loader again
  static const bool fragb = [] { const char* e = getenv("MDM_DEC_FRAGB"); return e == nullptr || e[0] != '0'; }();
#else
  constexpr bool fragb = true;
#endif
  if (x3 && wp.hi != nullptr && fragb) {
    X3FragB bl{wp.hi, wp.lo, (N + 31) / 32, K};
    launch_gemm_f32(al, bl, ep, M, N, K, s, true);
  } else {
    RowMajorLoader bl{w, K, N, K};
    launch_gemm_f32(al, bl, ep, M, N, K, s, x3);
  }
  return rt_launch_status();
}

// f16x3 GEMM on pre-split operands; writes fp32 `out` and/or split planes oh/ol.  seq_len > 0 tells the tiler that
// the M rows are token sequences of that length (tile = whole sequences).
int launch_linear_x3(Profiler* pf, X3Operand a, X3Weights w, const float* bias, const float* res, float* out,
                     p16_t* oh, p16_t* ol, int M, int N, int K, int act, int scale_cols, float col_scale, int seq_len,
                     hipStream_t s, X3Operand res_planes = X3Operand{nullptr, nullptr}) {
  if (K % X3_BK != 0) return fail(MDM_EINVAL, "f16x3 linear: K must be a multiple of 32");
  if (N % 4 != 0) return fail(MDM_EINVAL, "f16x3 linear: N must be a multiple of 4");
  ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * M * (double)N * K, s);
  X3Epilogue ep{out, bias, res, res_planes.hi, res_planes.lo, oh, ol, N, scale_cols, col_scale, QkvPlanes{}, 0, 0,
                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1.f, 1, 1, 1};
#ifdef MDM_PROBES
  if (g_x3_delay > 1) ep.emb_B = g_x3_delay;
#endif
  const int rc = launch_gemm_x3(a, w, ep, M, N, K, act, seq_len, s, g_x3_ablate);
  if (rc == -2) return fail(MDM_EUNSUPPORTED, "f16x3 linear: unsupported (activation, residual, output) combination");
  return rt_launch_status();
}

// in_proj in split precision: tokens -> Q (pre-scaled) / K / V^T operand planes of attention_x3.h
int launch_in_proj_x3(Profiler* pf, X3Operand a, X3Weights w, const float* bias, const QkvPlanes& qp, int nseq, int S,
                      int D, float qscale, hipStream_t s) {
  if (D % X3_BK != 0) return fail(MDM_EINVAL, "f16x3 in_proj: latent_dim must be a multiple of 32");
  ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * nseq * S * 3.0 * D * (double)D, s);
  X3Epilogue ep{nullptr, bias, nullptr, nullptr, nullptr, nullptr, nullptr, 3 * D, D, qscale, qp, S, D,
                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1.f, 1, 1, 1};
  const int rc = launch_gemm_x3_qkv(a, w, ep, nseq, S, D, s);
  if (rc == -2) return fail(MDM_EUNSUPPORTED, "f16x3 in_proj: sequences longer than 224 tokens");
  return rt_launch_status();
}

// One GEMM of the folded-LayerNorm encoder (gemm_x3.h launch_gemm_x3_ln kinds)
struct LnArgs {
  const float* astat = nullptr; const float* colsum = nullptr;                                   // FOLD
  X3Operand res{nullptr, nullptr}; const float* rstat = nullptr; const float* rgamma = nullptr; const float* rbeta = nullptr;  // residual
  float* ostat = nullptr;                                                                       // OSTAT
  int parts = 1; float inv_dim = 1.f;
  const float* res_f32 = nullptr; int emb_T = 1, emb_B = 1, emb_nbranch = 1;                     // EMBED (kind 5)
  bool small = false;      // the small-row-count kernel (gemm_x3s.h): the whole forward runs on one of the two kernels
  X3sShape shape{1, 1};    // ... and on ONE tile shape of it (x3s_shape(m->x3s, nseq))
  int stat_cols = 256;     // columns per partial of astat / rstat (what the PRODUCER's kernel wrote)
};
// The latency regime (gemm_x3s.h): a forward of at most MDM_OPT_SMALL_GEMM_MAX_SEQS sequences runs its GEMMs on 32 / 64-row tiles --
// and so does EVERY forward whose sequences are longer than gemm_x3.h's 224-row sequence tile (round 6: the row tiles do not care how
// long a sequence is; attention_long.h takes the attention)
inline bool use_small_gemm(const mdm_model* m, int nseq, int S) {
  return m->precision == MDM_PREC_F16X3 && m->lnfold && (nseq <= m->x3s.max_seqs || S > X3_TM) &&
         m->cfg.latent_dim % 128 == 0 && m->cfg.latent_dim % 256 == 0 && m->cfg.ff_size % 256 == 0;
}
int launch_x3_ln(Profiler* pf, int prof_cat, int kind, X3Operand a, X3Weights w, const float* bias, const LnArgs& ln,
                 float* out, p16_t* oh, p16_t* ol, const QkvPlanes* qp, int M, int N, int K, int S, int D,
                 int scale_cols, float col_scale, hipStream_t s) {
  if (K % X3_BK != 0 || N % 4 != 0) return fail(MDM_EINVAL, "f16x3 linear: K % 32 and N % 4 must be 0");
  // partial statistics per row: D / 256 on gemm_x3.h's tiles (<= 4), D / 128 on gemm_x3s.h's (<= 8); D <= 1024 (mdm_create)
  if (ln.parts < 1 || ln.parts > (ln.small ? 8 : 4)) return fail(MDM_EUNSUPPORTED, "folded LayerNorm: too many partial sums per row (D <= 1024)");
  ProfScope ps(pf, prof_cat, 2.0 * M * (double)N * K, s);
  X3Epilogue ep{out, bias, ln.res_f32, ln.res.hi, ln.res.lo, oh, ol, N, scale_cols, col_scale, qp ? *qp : QkvPlanes{}, S, D,
                ln.astat, ln.colsum, ln.rstat, ln.rgamma, ln.rbeta, ln.ostat, ln.parts, ln.inv_dim, ln.emb_T, ln.emb_B,
                ln.emb_nbranch};
  ep.stat_cols = ln.stat_cols;
  bool small = ln.small;
#ifdef MDM_PROBES   // (bisection of a misbehaving instantiation: bit k = GEMM kind k may run on the small kernel; results are wrong
                    // when producer and consumer of a row-statistics array disagree about their geometry)
  if (const char* e = getenv("MDM_X3S_KINDS")) small = small && ((atoi(e) >> kind) & 1);
#endif
  if (small) {
    // rows are grouped by sequence only where the epilogue needs (sequence, token) -- in_proj's Q / K / V^T planes, InputProcess's
    // (sample, frame); every other GEMM tiles its M rows CONTIGUOUSLY: 197 tokens are three 64-row tiles plus one of 5 rows, i.e.
    // a quarter of the workgroups of a sequence-aligned launch would do 8 % of a tile's work (B = 6: 37 row tiles instead of 48)
    const int group_rows = (kind == 0 || kind == 6) ? S : (kind == 5 ? ln.emb_T : M);
    const int rc = launch_gemm_x3s(kind, ln.shape, a, w, ep, M, N, K, group_rows, s);
    if (rc == -1 || rc == -3) return lds_fail(rc, "f16x3 linear (small tiles)");
    if (rc == -2) return fail(MDM_EUNSUPPORTED, "f16x3 linear (small tiles): unsupported shape (K must be 288 or a multiple of 256)");
#if defined(MDM_PROBES) && !defined(MDM_EMU)
    if (getenv("MDM_X3S_TRACE")) {      // bring-up: which launch faults
      fprintf(stderr, "[x3s] kind %d M %d N %d K %d launched\n", kind, M, N, K); fflush(stderr);
      const hipError_t e = hipStreamSynchronize(s);
      fprintf(stderr, "[x3s] kind %d done: %s\n", kind, hipGetErrorString(e)); fflush(stderr);
    }
#endif
    return rt_launch_status();
  }
  if (kind == 6) {   // layer 0's in_proj without a folded LayerNorm on the sequence-tile kernel (only reached by the bisection switch)
    const int rc6 = launch_gemm_x3_qkv(a, w, ep, M / S, S, D, s);
    if (rc6 != 0) return fail(MDM_EUNSUPPORTED, "f16x3 in_proj: launch failed");
    return rt_launch_status();
  }
  const int rpt = (kind == 0) ? S : x3_rows_per_tile(M, kind == 5 ? ln.emb_T : S);
  const int rc = launch_gemm_x3_ln(kind, a, w, ep, M, N, K, rpt, s);
  if (rc == -1 || rc == -3) return lds_fail(rc, "f16x3 linear");
  if (rc == -2) return fail(MDM_EUNSUPPORTED, "f16x3 linear: unsupported folded-LayerNorm GEMM kind");
  return rt_launch_status();
}

// fp32 [N][K] weights -> fragment-ordered hi/lo planes (gemm_x3.h header); K % 16 == 0
int launch_pack_weights(const float* src, p16_t* hi, p16_t* lo, int N, int K, hipStream_t s, int* overflow = nullptr) {
  if (K % 16 != 0) return fail(MDM_EINVAL, "pack_weights: K must be a multiple of 16");
  const size_t n = x3_packed_weight_elems(N, K) / 8;
  const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  MDM_LAUNCH(pack_weight_planes_kernel, dim3(grid), dim3(256), 0, s, src, hi, lo, N, K, overflow);
  return rt_launch_status();
}

int launch_split(const float* src, p16_t* hi, p16_t* lo, size_t n, hipStream_t s) {
  if (n % 4 != 0) return fail(MDM_EINVAL, "split: element count must be a multiple of 4");
  const size_t n4 = n / 4;
  const int grid = (int)std::min<size_t>((n4 + 255) / 256, 4096);
  MDM_LAUNCH(split_planes_kernel, dim3(grid), dim3(256), 0, s, src, hi, lo, n4);
  return rt_launch_status();
}

}  // namespace

// gemm_x3s8.h -- EXPERIMENT (not product code; round 4, unmeasured): gemm_x3s_kernel's 64-row x 128-column tile on an EIGHT-wave
// workgroup whose two wave groups split K.
//
// Why (profiles/r04j_x3s_timeline.md): with one wave per SIMD the k-loop of a 64 x 128 x 512 tile takes 10.4 k cycles for 6.1 k
// cycles of MFMA -- every chunk boundary (counted wait, barrier, LDS-DMA issue, first fragment round trip) is exposed -- and a
// second resident workgroup of the same launch runs the same phase at the same time (16.4 k each).  Here wave group G (waves
// 4 G .. 4 G + 3) contracts the chunks G, G + 2, ... into its own accumulators from its own pair of LDS buffers and its own W
// stream: every SIMD holds two independent MFMA streams whose chunk boundaries need not coincide.  Behind the loops the groups
// exchange one 32-row half of their partial tile through LDS (4 KB per wave) and group G finishes row sub-tile G: half the
// epilogue rounds, half the residual loads per wave.  128 + 10 KB of LDS: one workgroup per CU -- for launches of at most one
// tile per CU (out_proj, cross q / out_proj, linear2 at 3,840 rows), the others stay on gemm_x3s_kernel.
//
// Same operands, weight planes, epilogue algebra and statistics geometry as gemm_x3s.h (kinds 1, 2, 3, 4: no Q / K / V^T planes, no
// InputProcess); the contraction is summed as (chunks 0, 2, ..) + (chunks 1, 3, ..), so results differ from gemm_x3s_kernel's in
// the last bits.  K must be a multiple of 2 x 128.
#pragma once
#include "gemm_x3s.h"

namespace mdm {

constexpr int X3S8_WAVES = 8;
constexpr int x3s8_patch_base(int nsub) { return 4 * x3s_buf_bytes(2, nsub); }
constexpr int x3s8_tab_base(int nsub) { return x3s8_patch_base(nsub) + X3S8_WAVES * X3_PATCH_BYTES; }
constexpr int x3s8_part_base(int nsub) { return x3s8_tab_base(nsub) + 64 * 8; }
constexpr int x3s8_lds_bytes(int nsub) { return x3s8_part_base(nsub) + 4 * 64 * 8; }

template <int NSUB, int ACT, int RES, bool OUT_F32, bool OUT_PLANES, bool FOLD, bool OSTAT>
__global__ __launch_bounds__(64 * X3S8_WAVES, 1) void gemm_x3s8_kernel(X3Operand A, X3Weights W, X3Epilogue ep, int M, int N, int K,
                                                                      int group_rows, int tiles_per_group, int tiles_n,
                                                                      int total) {
  MDM_DYN_SMEM(unsigned char, lds);
  constexpr int RT = 2, TR = 64, TN = 128, D = 8, NBLK = 4;
  constexpr int BUF = x3s_buf_bytes(RT, NSUB);
  constexpr int PW = NSUB * RT / 2;                      // LDS-DMA pieces (1 KB) per wave and chunk (4 waves of a group share a chunk)
  constexpr int LW = 2;                                  // W loads per wave and sub-step
  static_assert(NSUB == D, "this form: 128-k chunks, ring of one chunk's sub-steps");
  static_assert(LW * (D - 1) + PW <= 63 && LW * NSUB <= 63, "vmcnt range");
  static_assert(RES == 0 || RES == 2 || RES == 3, "plane residual or none");
  constexpr bool LN_TABS = FOLD || RES == 3;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef MDM_EMU
  const int wid8 = tid >> 6;
#else
  const int wid8 = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int G = wid8 >> 2, wid = wid8 & 3;              // wave group (K half, then row sub-tile), wave of the group (column block)
  const int r = lane & 31, h = lane >> 5;

  const int lid = xcd_remap((int)blockIdx.x, total);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int grp = tile_m / tiles_per_group, tig = tile_m - grp * tiles_per_group;
  const int m0 = grp * group_rows + tig * TR;
  const int rows_valid = min(TR, group_rows - tig * TR);
  const int n0 = tile_n * TN;
  const int nk2 = K / (NSUB * 16) / 2;                  // chunks per group
  unsigned char* const gbuf = lds + G * (2 * BUF);      // this group's two chunk buffers

  const int schunk = (lane & 3) ^ ((lane >> 4) & 3);
  auto issue_chunk = [&](int k, int buf) {              // the group's k-th chunk = chunk G + 2 k of the contraction
    const int c = G + 2 * k;
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int q = wid + 4 * i;
      const int g = q % (2 * RT), p = (q / (2 * RT)) % 2, ms = q / (4 * RT);
      const int arow = min(m0 + g * 16 + (lane >> 2), M - 1);
      const p16_t* src = (p ? A.lo : A.hi) + (size_t)arow * K + (size_t)c * (NSUB * 16) + ms * 32 + schunk * 8;
      glds16(src, gbuf + buf * BUF + ((ms * 2 + p) * 2 * RT + g) * 1024);
    }
  };
  const uint32_t wbase = (uint32_t)min((n0 >> 5) + wid, (N + 31) / 32 - 1) * (uint32_t)(K / 16) * 512u + (uint32_t)lane * 8u;
  const int nsub_group = nk2 * NSUB;                    // sub-steps of this group
  p16x8 wsh[D] = {}, wsl[D] = {};
  auto issue_w = [&](auto slot_tag, int s) __attribute__((always_inline)) {   // s: the group's s-th sub-step
    constexpr int sl = decltype(slot_tag)::value;
    const int ss = s < nsub_group ? s : s - nsub_group;               // past the end: a harmless re-fetch keeps the wait counts uniform
    const int gg = (G + 2 * (ss / NSUB)) * NSUB + (ss % NSUB);          // its index over the whole K
    gload16_refill(wsh[sl], W.hi + wbase + (uint32_t)gg * 512u);
    gload16_refill(wsl[sl], W.lo + wbase + (uint32_t)gg * 512u);
  };

  issue_chunk(0, 0);
  static_for<D>([&](auto s_tag) __attribute__((always_inline)) { issue_w(s_tag, decltype(s_tag)::value); });

  // ---- the epilogue's operands, requested under the k-loop: group G finishes row sub-tile G (tile rows 32 G .. 32 G + 31)
  const int prow = lane >> 3, pc4 = (lane & 7) * 4;
  const int nb = n0 + wid * 32;
  const int n4 = nb + pc4;
  const bool col_ok = n4 < N;
  const float4 b4 = col_ok ? ld4(ep.bias + n4) : zero4();
  float4 c4 = zero4(), g4 = zero4(), be4 = zero4();
  if constexpr (FOLD) { if (col_ok) c4 = ld4(ep.colsum + n4); }
  if constexpr (RES == 3) { if (col_ok) { g4 = ld4(ep.rgamma + n4); be4 = ld4(ep.rbeta + n4); } }
  uint2 rrh[RES != 0 ? 4 : 1], rrl[RES != 0 ? 4 : 1];
  if constexpr (RES != 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int rit = 32 * G + 8 * q + prow, m = m0 + rit;
      const bool ok = rit < rows_valid && m < M && col_ok;
      const size_t o = (size_t)m * ep.ld + n4;
      rrh[q] = ok ? *reinterpret_cast<const uint2*>(ep.resh + o) : make_uint2(0u, 0u);
      rrl[q] = ok ? *reinterpret_cast<const uint2*>(ep.resl + o) : make_uint2(0u, 0u);
    }
  }
  float2* const stab = reinterpret_cast<float2*>(lds + x3s8_tab_base(NSUB));
  if constexpr (LN_TABS) {
    if (tid < TR) {
      const float* st = FOLD ? ep.astat : ep.rstat;
      const int m = m0 + tid;
      float2 v = make_float2(0.f, 0.f);
      if (tid < rows_valid && m < M) {
        const float* q = st + (size_t)m * ep.stat_parts * 2;
        float4 p01 = zero4(), p23 = zero4();
        if (ep.stat_parts == 4) { p01 = ld4(q); p23 = ld4(q + 4); }
        else if (ep.stat_parts == 2) p01 = ld4(q);
        else if (ep.stat_parts == 1) { const float2 t = *reinterpret_cast<const float2*>(q); p01.x = t.x; p01.y = t.y; }
        else { p01 = ld4(q); const float2 t = *reinterpret_cast<const float2*>(q + 4); p23.x = t.x; p23.y = t.y; }
        const float cols = (float)ep.stat_cols, icols = 1.0f / cols;
        const float mean = ((p01.x + p01.z) + (p23.x + p23.z)) * ep.inv_dim;
        const int np = ep.stat_parts;
        const float d0 = p01.x * icols - mean, d1 = np > 1 ? p01.z * icols - mean : 0.f;
        const float d2 = np > 2 ? p23.x * icols - mean : 0.f, d3 = np > 3 ? p23.z * icols - mean : 0.f;
        const float m2 = (p01.y + p01.w) + (p23.y + p23.w) + cols * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
        v = make_float2(mean, 1.0f / sqrtf(m2 * ep.inv_dim + 1e-5f));
      }
      stab[tid] = v;
    }
  }

  f32x16 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  const int sw = (r >> 2) & 3;
  const uint32_t fr0 = (uint32_t)(r * 64 + ((h ^ sw) * 16)), fr1 = (uint32_t)(r * 64 + (((2 + h) ^ sw) * 16));
#ifndef MDM_EMU
  const uint32_t gbuf_addr = lds_addr_of(gbuf);
#endif
  p16x8 fah[2][RT], fal[2][RT];
  auto read_frags = [&](auto j_tag, int buf) __attribute__((always_inline)) {
    constexpr int j = decltype(j_tag)::value, ms = j / 2, ks = j % 2;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
#ifdef MDM_EMU
      lds_read16(fah[j & 1][t], gbuf + buf * BUF, (uint32_t)(((ms * 2 + 0) * 2 * RT) * 1024 + t * 2048) + (ks ? fr1 : fr0));
      lds_read16(fal[j & 1][t], gbuf + buf * BUF, (uint32_t)(((ms * 2 + 1) * 2 * RT) * 1024 + t * 2048) + (ks ? fr1 : fr0));
#else
      constexpr uint32_t OH = (uint32_t)(((ms * 2 + 0) * 2 * RT) * 1024), OL = (uint32_t)(((ms * 2 + 1) * 2 * RT) * 1024);
      const uint32_t ad = gbuf_addr + (uint32_t)buf * BUF + (ks ? fr1 : fr0) + (uint32_t)t * 2048u;
      lds_read16<(int)(OH & 32767u)>(fah[j & 1][t], ad + (OH & ~32767u));
      lds_read16<(int)(OL & 32767u)>(fal[j & 1][t], ad + (OL & ~32767u));
#endif
    }
  };
  auto wait_frags = [&](auto j_tag, auto younger_tag) __attribute__((always_inline)) {
    constexpr int j = decltype(j_tag)::value, n = decltype(younger_tag)::value;
    lds_wait<n>(fah[j & 1][0], fal[j & 1][0], fah[j & 1][1], fal[j & 1][1]);
  };

  for (int k = 0; k < nk2; ++k) {
    const int buf = k & 1;
    // NSUB == D: no W wait of chunk k - 1 lies behind the pieces of chunk k -- wait for them here (younger = the LW * NSUB W loads
    // issued since).  The barrier is the WORKGROUP's: both groups run nk2 iterations, so they meet here once per chunk pair
    vmem_wait<LW * NSUB>(wsh[0], wsl[0]);
    wg_barrier_nodrain();
    issue_chunk(min(k + 1, nk2 - 1), buf ^ 1);
    read_frags(std::integral_constant<int, 0>{}, buf);
    static_for<NSUB>([&](auto j_tag) __attribute__((always_inline)) {
      constexpr int j = decltype(j_tag)::value, sl = j % D;
      if constexpr (j + 1 < NSUB) read_frags(std::integral_constant<int, j + 1>{}, buf);
      constexpr int NW = LW * (D - 1) + PW;      // (j < D always: the next chunk's pieces were issued in front of every refill)
      vmem_wait<NW>(wsh[sl], wsl[sl]);
      wait_frags(j_tag, std::integral_constant<int, (j + 1 < NSUB) ? 2 * RT : 0>{});
#ifndef MDM_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = mfma_p16(fal[j & 1][t], wsh[sl], acc[t]);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = mfma_p16(fah[j & 1][t], wsl[sl], acc[t]);
#pragma unroll
      for (int t = 0; t < RT; ++t) acc[t] = mfma_p16(fah[j & 1][t], wsh[sl], acc[t]);
#ifndef MDM_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      issue_w(std::integral_constant<int, sl>{}, k * NSUB + j + D);
    });
  }
  // the closing wait NAMES every slot register (gemm_x3s.h: the register-recycling hazard of in-place refills)
  static_for<D / 4>([&](auto q_tag) __attribute__((always_inline)) {
    constexpr int q = 4 * decltype(q_tag)::value;
    vmem_wait<0>(wsh[q], wsl[q], wsh[q + 1], wsl[q + 1], wsh[q + 2], wsl[q + 2], wsh[q + 3], wsl[q + 3]);
  });

  // ---- exchange: group G keeps row sub-tile G of its partial and hands sub-tile 1 - G to the other group.  4 KB per wave
  // (lane-major float4s: conflict-free) in the group's own buffer 0 -- every wave of the workgroup is past its k-loop first
  wg_barrier();
  {
    float4* xo = reinterpret_cast<float4*>(gbuf + wid * 4096);
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const f32x16& give = G ? acc[0] : acc[1];
      xo[e4 * 64 + lane] = make_float4(give[4 * e4 + 0], give[4 * e4 + 1], give[4 * e4 + 2], give[4 * e4 + 3]);
    }
  }
  wg_barrier();
  f32x16 mine;
  {
    const float4* xi = reinterpret_cast<const float4*>(lds + (G ^ 1) * (2 * BUF) + wid * 4096);
#pragma unroll
    for (int e4 = 0; e4 < 4; ++e4) {
      const float4 v = xi[e4 * 64 + lane];
      const f32x16& keep = G ? acc[1] : acc[0];
      mine[4 * e4 + 0] = keep[4 * e4 + 0] + v.x; mine[4 * e4 + 1] = keep[4 * e4 + 1] + v.y;
      mine[4 * e4 + 2] = keep[4 * e4 + 2] + v.z; mine[4 * e4 + 3] = keep[4 * e4 + 3] + v.w;
    }
  }

  // ---- epilogue of row sub-tile G (gemm_x3s.h's rounds for t = G)
  float* patch = reinterpret_cast<float*>(lds + x3s8_patch_base(NSUB)) + wid8 * (X3_PATCH_BYTES / 4);
  const float accs = ep.acc_scale;
  constexpr bool COL_SCALE = x3_has_col_scale(ACT, RES);
  auto finish4 = [&](float4 v4, float2 st) __attribute__((always_inline)) {
    const float mult4 = (COL_SCALE && n4 < ep.scale_cols) ? ep.col_scale : 1.f;
    v4.x *= accs; v4.y *= accs; v4.z *= accs; v4.w *= accs;
    if constexpr (FOLD) {
      v4.x = st.y * (v4.x - st.x * c4.x) + b4.x; v4.y = st.y * (v4.y - st.x * c4.y) + b4.y;
      v4.z = st.y * (v4.z - st.x * c4.z) + b4.z; v4.w = st.y * (v4.w - st.x * c4.w) + b4.w;
    } else {
      v4.x += b4.x; v4.y += b4.y; v4.z += b4.z; v4.w += b4.w;
    }
    if (ACT == ACT_GELU) { v4.x = gelu_erf_fast(v4.x); v4.y = gelu_erf_fast(v4.y); v4.z = gelu_erf_fast(v4.z); v4.w = gelu_erf_fast(v4.w); }
    else if (ACT == ACT_SILU) { v4.x = silu(v4.x); v4.y = silu(v4.y); v4.z = silu(v4.z); v4.w = silu(v4.w); }
    if constexpr (COL_SCALE) { v4.x *= mult4; v4.y *= mult4; v4.z *= mult4; v4.w *= mult4; }
    return v4;
  };
  float2* const part_all = reinterpret_cast<float2*>(lds + x3s8_part_base(NSUB));   // OSTAT: [block][row] partials
  float2* part = part_all + wid * TR;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
#pragma unroll
    for (int e = 0; e < 4; ++e) patch[((e + 4 * h) << 5) + r] = mine[4 * g + e];
    wave_lds_fence();
    float4 v4 = ld4(&patch[prow * 32 + pc4]);
    wave_lds_fence();
    const int rit = 32 * G + 8 * g + prow, m = m0 + rit;
    const bool row_ok = rit < rows_valid && m < M;
    float2 st = make_float2(0.f, 1.f);
    if constexpr (LN_TABS) st = stab[rit];
    v4 = finish4(v4, FOLD ? st : make_float2(0.f, 1.f));
    if constexpr (RES == 2 || RES == 3) {
      const uint2 a = rrh[g], b = rrl[g];
      float4 x4 = make_float4(p16_to_f32((p16_t)(a.x & 0xffffu)) + p16_to_f32((p16_t)(b.x & 0xffffu)),
                              p16_to_f32((p16_t)(a.x >> 16)) + p16_to_f32((p16_t)(b.x >> 16)),
                              p16_to_f32((p16_t)(a.y & 0xffffu)) + p16_to_f32((p16_t)(b.y & 0xffffu)),
                              p16_to_f32((p16_t)(a.y >> 16)) + p16_to_f32((p16_t)(b.y >> 16)));
      if constexpr (RES == 3) {
        x4.x = (x4.x - st.x) * st.y * g4.x + be4.x; x4.y = (x4.y - st.x) * st.y * g4.y + be4.y;
        x4.z = (x4.z - st.x) * st.y * g4.z + be4.z; x4.w = (x4.w - st.x) * st.y * g4.w + be4.w;
      }
      v4 = add4(v4, x4);
    }
    if constexpr (OSTAT) {
      const float s1 = sum_lanes8((v4.x + v4.y) + (v4.z + v4.w));
      const float mw = s1 * (1.0f / 32.0f);
      const float dx = v4.x - mw, dy = v4.y - mw, dz = v4.z - mw, dw = v4.w - mw;
      const float m2 = sum_lanes8((dx * dx + dy * dy) + (dz * dz + dw * dw));
      if ((lane & 7) == 0) part[rit] = make_float2(s1, m2);
    }
    if (row_ok && col_ok) {
      const size_t o = (size_t)m * ep.ld + n4;
      if constexpr (OUT_PLANES) split4_store(ep.oh + o, ep.ol + o, v4);
      if constexpr (OUT_F32) st4(ep.out + o, v4);
    }
  }
  if constexpr (OSTAT) {
    wg_barrier();
    if (tid < rows_valid && m0 + tid < M) {
      float s1 = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < NBLK; ++w4) s1 += part_all[w4 * TR + tid].x;
      const float mt = s1 * (1.0f / TN);
      float m2 = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < NBLK; ++w4) {
        const float2 v = part_all[w4 * TR + tid];
        const float dm = v.x * (1.0f / 32.0f) - mt;
        m2 += v.y + 32.0f * dm * dm;
      }
      *reinterpret_cast<float2*>(ep.ostat + ((size_t)(m0 + tid) * tiles_n + tile_n) * 2) = make_float2(s1, m2);
    }
  }
}

template <int ACT, int RES, bool OUT_F32, bool OUT_PLANES, bool FOLD, bool OSTAT>
inline int launch_gemm_x3s8_t(const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N, int K, int group_rows,
                              hipStream_t stream) {
  constexpr int NS = 8;
  auto kfn = &gemm_x3s8_kernel<NS, ACT, RES, OUT_F32, OUT_PLANES, FOLD, OSTAT>;
  if (K % (2 * NS * 16) != 0 || M % group_rows != 0) return -2;
  if (!x3_has_col_scale(ACT, RES) && ep.scale_cols > 0) return -2;
  if (OSTAT && N % 128 != 0) return -2;
  constexpr int LDS = x3s8_lds_bytes(NS);
#ifndef MDM_EMU
  static bool configured[kMaxDevices] = {};
  bool& done = configured[rt_device_ordinal()];
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
    done = true;
  }
#endif
  const int tpg = (group_rows + 63) / 64, tiles_m = (M / group_rows) * tpg, tiles_n = (N + 127) / 128;
  const int total = tiles_m * tiles_n;
  MDM_LAUNCH(kfn, dim3(total), dim3(64 * X3S8_WAVES), LDS, stream, A, W, ep, M, N, K, group_rows, tpg, tiles_n, total);
  return 0;
}

// Which GEMM kinds take the 8-wave form (MDM_X3S8_KINDS, bit k = kind k; default: the 512-column kinds 1, 2, 4 -- one tile per CU at
// DiP's 3,840 rows; 0 = none): only 64-row tiles, only K % 256 == 0
inline int x3s8_kinds() {
  const char* e = getenv("MDM_X3S8_KINDS");
  return e != nullptr ? atoi(e) : ((1 << 1) | (1 << 2) | (1 << 4));
}
inline int launch_gemm_x3s_or_x3s8(int kind, X3sShape sh, const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N,
                                   int K, int group_rows, hipStream_t s) {
  if (sh.rt == 2 && sh.ncb == 1 && K % 256 == 0 && ((x3s8_kinds() >> kind) & 1)) {
    switch (kind) {
      case 1: return launch_gemm_x3s8_t<ACT_NONE, 2, false, true, false, true>(A, W, ep, M, N, K, group_rows, s);
      case 2: return launch_gemm_x3s8_t<ACT_NONE, 3, false, true, false, true>(A, W, ep, M, N, K, group_rows, s);
      case 3: return launch_gemm_x3s8_t<ACT_GELU, 0, false, true, true, false>(A, W, ep, M, N, K, group_rows, s);
      case 4: return launch_gemm_x3s8_t<ACT_NONE, 0, true, false, true, false>(A, W, ep, M, N, K, group_rows, s);
      default: break;
    }
  }
  return launch_gemm_x3s(kind, sh, A, W, ep, M, N, K, group_rows, s);
}

}  // namespace mdm

// Split-precision ("bf16x3") MFMA GEMM for the encoder's dense contractions:
//     C[m][n] = sum_k A[m][k] * W[n][k],   A = Ah + Al,  W = Wh + Wl  (bf16 planes, see common.h split_bf16)
//             ~ sum_k  Ah*Wh + Ah*Wl + Al*Wh            three v_mfma_f32_32x32x16_bf16 passes, fp32 accumulate.
//
// Why: the reference computes these `addmm`s in fp32 (model/mdm.py:77-84 -> torch TransformerEncoderLayer) and
// BASELINE's parity bar is 1e-3 max-abs over a 50-step guided trajectory.  gfx950 has no TF32; exact-fp32 MFMA
// peaks at 157 TFLOP/s, bf16 MFMA at 2.5 PFLOP/s, so three bf16 passes carry a ~2^-16-relative fp32 product at up to
// ~5x the fp32 rate (measured trajectory error ~3e-5).  Replaces in_proj / out_proj / linear1 /
// linear2; the 263-wide input/output projections stay on the exact-fp32 kernel (gemm_f32.h).
//
// Data layout: both operands are stored K-contiguous as two bf16 planes [rows][K] (hi, lo).  Weights are split
// once in mdm_prepare; activations are split by the PRODUCING kernel's epilogue (LayerNorm, attention, GELU), so the
// main loop is pure LDS-DMA + MFMA.
//
// Machine mapping (gfx950): 512 threads = 8 waves, ONE PERSISTENT workgroup per CU that walks its share of the
// output tiles.  Block tile = up to 224 rows x 256 columns; the row extent is chosen by the host as a whole number
// of token sequences (S = 197 -> one sequence per tile), so the headline shape (256 sequences, N in {512, 1024,
// 1536}) gives every CU exactly N/256 equal tiles.  Wave w owns columns [32w, 32w+32) x all 7 row sub-tiles
// (7 accumulators = 112 VGPRs); its W fragment feeds 21 MFMAs per 16-deep k sub-step.  BK = 32; one LDS stage =
// Ah|Al [224][32] + Wh|Wl [256][32] bf16 = 60 KB; A ring of 3, W ring of 2, plus a 1 KB per-wave epilogue patch.
//   * global -> LDS by global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass).  The loads run AHEAD of the
//     MFMAs as two streams with their own (tile, k) cursors -- W one step ahead, A two -- and simply roll over into
//     the NEXT tile of this workgroup, so the pipeline never drains at a tile boundary: the epilogue of tile j runs
//     with the first stages of tile j+1 already in LDS (no per-tile prologue latency, K is only 16-32 steps deep).
//   * the 7-8 LDS-DMA instructions of a step are issued one at a time BETWEEN the MFMA units of that step, not as a
//     burst behind the barrier (a burst leaves the matrix pipe idle for the ~100 cycles each one takes to issue);
//     they are retired by ONE counted s_waitcnt vmcnt + raw s_barrier per K step;
//   * the LDS image of a plane tile is row-major with 64-byte rows; the 16-byte chunk index is XOR-swizzled with
//     (row>>2)&3 so the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots of the 256-byte bank row.
//     LDS-DMA writes lane-linearly, so the swizzle is applied to the per-lane SOURCE address and to the reads
//     (cdna_hip_programming.md rule 21);
//   * XCD-aware tile order (the column tiles of one row panel are adjacent inside an XCD's contiguous chunk) keeps a
//     sequence's activation planes in one L2; the weight planes of a layer (<= 3 MB) stay L2-resident everywhere.
#pragma once
#include "attention_bf16x3.h"  // QkvPlanes: the in_proj epilogue writes the attention kernel's operand planes
#include "common.h"
#include "gemm_f32.h"  // ACT_* enums
#include <type_traits>

namespace mdm {

constexpr int X3_TM = 224, X3_TN = 256, X3_BK = 32, X3_THREADS = 512;
constexpr int X3_MSUB = X3_TM / 32;                              // 7 row sub-tiles
constexpr int X3_A_BYTES = X3_TM * X3_BK * 2;                    // one A plane tile: 14336
constexpr int X3_W_BYTES = X3_TN * X3_BK * 2;                    // one W plane tile: 16384
constexpr int X3_A_STAGE = 2 * X3_A_BYTES;                       // Ah|Al: 28672
constexpr int X3_W_STAGE = 2 * X3_W_BYTES;                       // Wh|Wl: 32768
constexpr int X3_A_RING = 3, X3_W_RING = 2;                      // activations come from MALL/HBM: prefetch 2 ahead;
                                                                 // the weight tile is L2-hot (every CU reads it): 1 ahead
constexpr int X3_W_BASE = X3_A_RING * X3_A_STAGE;
constexpr int X3_PATCH_BASE = X3_W_BASE + X3_W_RING * X3_W_STAGE;  // 151552
constexpr int X3_PATCH_BYTES = 8 * 32 * 4;                         // per wave: 8 rows x 32 columns fp32
constexpr int X3_LDS_BYTES = X3_PATCH_BASE + 8 * X3_PATCH_BYTES;   // 159744 (<= 163840)
constexpr int X3_A_GROUPS = X3_A_STAGE / 1024, X3_W_GROUPS = X3_W_STAGE / 1024;  // 28 + 32 LDS-DMA wave-instructions

struct X3Operand {
  const bf16_t* hi;
  const bf16_t* lo;
};

// v = act(acc + bias[n]) * (n < scale_cols ? col_scale : 1) + (res ? res[m][n] : 0) -> fp32 out and/or split planes,
// or (OUT_QKV) the attention operand planes of attention_bf16x3.h.
struct X3Epilogue {
  float* out;        // [M][ld] or null
  const float* bias;
  const float* res;  // [M][ld] or null; may alias out
  bf16_t* oh;        // [M][ld] split planes or null
  bf16_t* ol;
  int ld;
  int scale_cols;
  float col_scale;
  QkvPlanes qkv;     // OUT_QKV only
  int S, D;          // OUT_QKV only: tokens per sequence (= rows per tile), model width (N = 3 D)
};

// exact-GELU with erf from Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7, i.e. fp32-rounding class): one v_exp, one v_rcp
// and a 5-term Horner chain instead of the ~40-instruction libm erff; used only in this split-precision path.
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = 1.0f / (1.0f + 0.3275911f * z);
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
#ifdef MDM_EMU
  const float erfc_z = p * t * expf(-z * z);
#else
  const float erfc_z = p * t * __expf(-z * z);
#endif
  const float erf_abs = 1.0f - erfc_z;
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// One of the two load streams (A or W): which tile of this workgroup and which k step it will fetch next, and the
// per-lane source element offsets of its (up to) four LDS-DMA pieces for that tile.
struct X3Cursor {
  int v;        // virtual tile id (blockIdx.x + j * gridDim.x)
  int k;        // next k step
  uint32_t off[4];
};

// ABL (profiling experiments only, 0 in production): 1 = no epilogue stores, 2 = no loads after the prologue,
// 4 = no MFMAs, 8 = LDS-DMA issued as a burst at the top of the step instead of between the MFMA units,
// 32 = the PAIRED MFMA schedule (see the main loop).
template <int ACT, bool HAS_RES, bool OUT_F32, bool OUT_PLANES, bool OUT_QKV, int ABL>
__global__ __launch_bounds__(X3_THREADS, 2) void gemm_bf16x3_kernel(X3Operand A, X3Operand W, X3Epilogue ep, int M, int N,
                                                                     int K, int rows_per_tile, int tiles_n, int total) {
  MDM_DYN_SMEM(unsigned char, lds);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef MDM_EMU
  const int wid = tid >> 6;
#else
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int r = lane & 31, h = lane >> 5;
  const int nk = K / X3_BK;
  const int gstride = (int)gridDim.x;

  auto tile_origin = [&](int v, int& m0, int& n0) {
    const int lid = xcd_remap(v, total);
    const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
    m0 = tile_m * rows_per_tile;
    n0 = tile_n * X3_TN;
  };

  // ---- LDS-DMA sources.  A stage image = 28 groups of 1 KB (16 rows x 64 B): groups 0-13 Ah, 14-27 Al; W stage image =
  // 32 groups: 0-15 Wh, 16-31 Wl.  Wave w issues groups w, w+8, ...  Lane -> (row = lane>>2, stored chunk = lane&3);
  // the logical k-chunk it fetches is stored ^ ((row>>2)&3) = (lane&3) ^ ((lane>>4)&3).  Rows past the tile / matrix
  // are clamped (their products are never stored).
  const int schunk = (lane & 3) ^ ((lane >> 4) & 3);
  auto aim_a = [&](X3Cursor& c) {
    int m0, n0;
    tile_origin(c.v, m0, n0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qa = wid + 8 * i;  // < 28 for i < 3, and for i == 3 when wid < 4
      const int ga = (qa < 14) ? qa : qa - 14;
      const int arow = min(m0 + ga * 16 + (lane >> 2), M - 1);
      c.off[i] = (uint32_t)arow * (uint32_t)K + schunk * 8;
    }
  };
  auto aim_w = [&](X3Cursor& c) {
    int m0, n0;
    tile_origin(c.v, m0, n0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int qw = wid + 8 * i;  // < 32
      const int gw = (qw < 16) ? qw : qw - 16;
      const int wrow = min(n0 + gw * 16 + (lane >> 2), N - 1);
      c.off[i] = (uint32_t)wrow * (uint32_t)K + schunk * 8;
    }
  };
  auto piece_a = [&](const X3Cursor& c, int i, int buf) {
    if (i < 3 || wid < 4)  // groups wid + 8 i < 28
      glds16(((wid + 8 * i < 14) ? A.hi : A.lo) + c.off[i] + c.k * X3_BK, lds + buf * X3_A_STAGE + wid * 1024 + i * 8192);
  };
  auto piece_w = [&](const X3Cursor& c, int i, int buf) {
    glds16(((wid + 8 * i < 16) ? W.hi : W.lo) + c.off[i] + c.k * X3_BK,
           lds + X3_W_BASE + buf * X3_W_STAGE + wid * 1024 + i * 8192);
  };
  // past its last tile a stream simply re-fetches that tile (at most two wasted stages per workgroup): the step body
  // then needs no "is there anything left to load" branches at all
  auto advance_a = [&](X3Cursor& c) {
    if (++c.k == nk) {
      c.k = 0;
      if (c.v + gstride < total) { c.v += gstride; aim_a(c); }
    }
  };
  auto advance_w = [&](X3Cursor& c) {
    if (++c.k == nk) {
      c.k = 0;
      if (c.v + gstride < total) { c.v += gstride; aim_w(c); }
    }
  };

  // ---- fragment read offsets (bytes inside a plane tile): row*64 + ((ksub*2 + h) ^ sw)*16, sw = (row>>2)&3
  const int sw = (r >> 2) & 3;
  const int fa = r * 64;                 // + t*2048 per row sub-tile
  const int fw = (wid * 32 + r) * 64;
#ifndef MDM_EMU
  const uint32_t lds_base = lds_addr_of(lds);
#endif

  int v = (int)blockIdx.x;
  if (v >= total) return;
  X3Cursor ca{v, 0, {0, 0, 0, 0}}, cw{v, 0, {0, 0, 0, 0}};
  aim_a(ca);
  aim_w(cw);

  // Pipeline invariant: at the top of global step g the LDS holds A(g), W(g) [landed, visible] and A(g+1) [in flight].
  // During the step W(g+1) THEN A(g+2) are issued; at its end wait until at most the A(g+2) pieces (3 per wave; waves
  // 0-3 issue a 4th that is then also waited for) are outstanding -- loads retire in order, so W(g+1) and A(g+1) have
  // landed -- and barrier once.  g runs across tile boundaries; epilogue stores only make the count more conservative.
#pragma unroll
  for (int i = 0; i < 4; ++i) piece_w(cw, i, 0);
  advance_w(cw);
#pragma unroll
  for (int i = 0; i < 4; ++i) piece_a(ca, i, 0);
  advance_a(ca);
#pragma unroll
  for (int i = 0; i < 4; ++i) piece_a(ca, i, 1);
  advance_a(ca);
  wait_vmem_upto3();
  wg_barrier();

  int abuf = 0, wbuf = 0;
  for (; v < total; v += gstride) {
    int m0, n0;
    tile_origin(v, m0, n0);
    f32x16 acc[X3_MSUB];
#pragma unroll
    for (int t = 0; t < X3_MSUB; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    // this lane's bias, fetched (and its wait retired: hipcc waits vmcnt(0) for a tracked load) at the START of the tile,
    // so that the epilogue's untracked residual loads are not drained by it
    const int ncol0 = n0 + wid * 32;                     // this wave's first column (wave-uniform)
    const int nc = ncol0 + r;                            // this lane's column in the accumulator layout
    float bias = (nc < N) ? ep.bias[nc] : 0.f;
#ifndef MDM_EMU
    asm volatile("" : "+v"(bias));
#endif

    for (int kt = 0; kt < nk; ++kt) {
      const int abuf_ld = abuf >= 1 ? abuf - 1 : 2;  // (abuf + 2) % 3
      // 14 units per stage (2 k sub-steps x 7 row sub-tiles), each = 2 A-fragment reads + 3 MFMAs, software-pipelined
      // DEPTH units deep: the reads of unit u+DEPTH are issued, then a COUNTED wait (2*DEPTH younger reads may stay in
      // flight) retires unit u's, then its 3 MFMAs go.  One LDS-DMA piece rides behind each of the first eight units: W(g+1) pieces
      // first, then A(g+2).
      constexpr int DEPTH = 2, RING = DEPTH + 1;  // fragment-read lookahead in units
      bf16x8 wh[2], wl[2], ah[RING], al[RING];
#ifdef MDM_EMU
      const unsigned char* sa = lds + abuf * X3_A_STAGE;
      const unsigned char* sw_ = lds + X3_W_BASE + wbuf * X3_W_STAGE;
#define X3_RD_A(dst, plane, t, ks) lds_read16(dst, sa, (plane) * X3_A_BYTES + fa + (t) * 2048 + ((((ks) * 2 + h) ^ sw) * 16))
#define X3_RD_W(dst, plane, ks) lds_read16(dst, sw_, (plane) * X3_W_BYTES + fw + ((((ks) * 2 + h) ^ sw) * 16))
#else
      // per-lane LDS byte addresses of this stage's fragments for k sub-step 0 / 1 (the XOR swizzle moves with ks)
      const uint32_t sa0 = lds_base + abuf * X3_A_STAGE + fa, sw0 = lds_base + X3_W_BASE + wbuf * X3_W_STAGE + fw;
      const uint32_t aaddr[2] = {sa0 + ((h ^ sw) * 16), sa0 + (((2 + h) ^ sw) * 16)};
      const uint32_t waddr[2] = {sw0 + ((h ^ sw) * 16), sw0 + (((2 + h) ^ sw) * 16)};
#define X3_RD_A(dst, plane, t, ks) do { if constexpr (!(ABL & 512)) lds_read16<(plane) * X3_A_BYTES + (t) * 2048>(dst, aaddr[ks]); } while (0)
#define X3_RD_W(dst, plane, ks) do { if constexpr (!(ABL & 512)) lds_read16<(plane) * X3_W_BYTES>(dst, waddr[ks]); } while (0)
#endif
      if constexpr ((ABL & 32) != 0) {
        // PAIRED schedule: units 2p and 2p+1 (always different row sub-tiles) are issued as M_a M_b M_a M_b M_a M_b, so
        // consecutive MFMAs never share an accumulator and the fragment reads of pair p+2 / the LDS-DMA pieces can sit
        // BETWEEN them in the matrix pipe's shadow instead of behind a same-accumulator triple (a filler inside such a
        // triple costs ~43 cycles, behind it ~6 per instruction: MI355X_MICROARCH.md cycle constants).
        constexpr int NP = X3_MSUB;  // 7 pairs per stage
        bf16x8 fr[3][4];            // ring of pair fragment sets: {ah_a, al_a, ah_b, al_b}
        auto rd_pair = [&](auto p_tag) __attribute__((always_inline)) {
          constexpr int p = decltype(p_tag)::value;
          if constexpr (p < NP) {
            constexpr int ua = 2 * p, ub = 2 * p + 1;
            X3_RD_A(fr[p % 3][0], 0, ua % X3_MSUB, ua / X3_MSUB);
            X3_RD_A(fr[p % 3][1], 1, ua % X3_MSUB, ua / X3_MSUB);
            X3_RD_A(fr[p % 3][2], 0, ub % X3_MSUB, ub / X3_MSUB);
            X3_RD_A(fr[p % 3][3], 1, ub % X3_MSUB, ub / X3_MSUB);
          }
        };
        X3_RD_W(wh[0], 0, 0);
        X3_RD_W(wl[0], 1, 0);
        X3_RD_W(wh[1], 0, 1);
        X3_RD_W(wl[1], 1, 1);
        rd_pair(std::integral_constant<int, 0>{});
        rd_pair(std::integral_constant<int, 1>{});
#ifndef MDM_EMU
#define X3_PIN() __builtin_amdgcn_sched_barrier(0)
#else
#define X3_PIN()
#endif
        static_for<NP>([&](auto p_tag) __attribute__((always_inline)) {
          constexpr int p = decltype(p_tag)::value;
          constexpr int ua = 2 * p, ub = 2 * p + 1;
          constexpr int ta = ua % X3_MSUB, ka = ua / X3_MSUB, tb = ub % X3_MSUB, kb = ub / X3_MSUB;
          constexpr int s = p % 3, sn = (p + 2) % 3;
          constexpr int younger = (p + 1 < NP) ? 4 : 0;  // pair p+1's reads may stay in flight
          if constexpr (p == 0)
            lds_wait<younger>(fr[s][0], fr[s][1], fr[s][2], fr[s][3], wh[0], wl[0], wh[1], wl[1]);
          else
            lds_wait<younger>(fr[s][0], fr[s][1], fr[s][2], fr[s][3]);
          X3_PIN();
          constexpr bool rd = (p + 2 < NP);
          constexpr int un = 2 * (p + 2);  // first unit of pair p+2
          acc[ta] = mfma_bf16(fr[s][1], wh[ka], acc[ta]);
          X3_PIN();
          if constexpr (rd) X3_RD_A(fr[sn][0], 0, un % X3_MSUB, un / X3_MSUB);
          X3_PIN();
          acc[tb] = mfma_bf16(fr[s][3], wh[kb], acc[tb]);
          X3_PIN();
          if constexpr (rd) X3_RD_A(fr[sn][1], 1, un % X3_MSUB, un / X3_MSUB);
          X3_PIN();
          acc[ta] = mfma_bf16(fr[s][0], wl[ka], acc[ta]);
          X3_PIN();
          if constexpr (rd) X3_RD_A(fr[sn][2], 0, (un + 1) % X3_MSUB, (un + 1) / X3_MSUB);
          X3_PIN();
          acc[tb] = mfma_bf16(fr[s][2], wl[kb], acc[tb]);
          X3_PIN();
          if constexpr (rd) X3_RD_A(fr[sn][3], 1, (un + 1) % X3_MSUB, (un + 1) / X3_MSUB);
          X3_PIN();
          acc[ta] = mfma_bf16(fr[s][0], wh[ka], acc[ta]);
          X3_PIN();
          if constexpr (!(ABL & 2)) {
            if constexpr (p < 2) piece_w(cw, 2 * p, wbuf ^ 1);
            else if constexpr (p < 4) piece_a(ca, 2 * (p - 2), abuf_ld);
          }
          X3_PIN();
          acc[tb] = mfma_bf16(fr[s][2], wh[kb], acc[tb]);
          X3_PIN();
          if constexpr (!(ABL & 2)) {
            if constexpr (p < 2) piece_w(cw, 2 * p + 1, wbuf ^ 1);
            else if constexpr (p < 4) piece_a(ca, 2 * (p - 2) + 1, abuf_ld);
          }
          X3_PIN();
        });
#undef X3_PIN
      } else {
      X3_RD_W(wh[0], 0, 0);
      X3_RD_W(wl[0], 1, 0);
      X3_RD_W(wh[1], 0, 1);
      X3_RD_W(wl[1], 1, 1);
      constexpr int NU = 2 * X3_MSUB;  // units per stage
      static_for<NU + DEPTH>([&](auto u_tag) __attribute__((always_inline)) {
        constexpr int u = decltype(u_tag)::value;
        if constexpr (u < NU) {
          constexpr int ks = u / X3_MSUB, t = u - ks * X3_MSUB;
          X3_RD_A(ah[u % RING], 0, t, ks);
          X3_RD_A(al[u % RING], 1, t, ks);
        }
        if constexpr (u >= DEPTH) {
          constexpr int uv = u - DEPTH, ks = uv / X3_MSUB, t = uv - ks * X3_MSUB;
#ifndef MDM_EMU
          // experiment: a wave that is BEHIND in its step outranks its SIMD partner, so the two waves of a SIMD finish
          // the step together instead of one parking at the barrier while the other runs alone
          if constexpr ((ABL & 64) != 0) {
            if constexpr (uv == 0) __builtin_amdgcn_s_setprio(1);
            if constexpr (uv == X3_MSUB) __builtin_amdgcn_s_setprio(0);
          }
          if constexpr ((ABL & 128) != 0) {
            if constexpr (uv == 0) __builtin_amdgcn_s_setprio(3);
            if constexpr (uv == 4) __builtin_amdgcn_s_setprio(2);
            if constexpr (uv == 7) __builtin_amdgcn_s_setprio(1);
            if constexpr (uv == 11) __builtin_amdgcn_s_setprio(0);
          }
#endif
          // reads allowed to stay in flight: those of the (up to) DEPTH younger units
          constexpr int younger = 2 * ((NU - 1 - uv) < DEPTH ? (NU - 1 - uv) : DEPTH);
          if constexpr (uv == 0) lds_wait<younger>(ah[0], al[0], wh[0], wl[0], wh[1], wl[1]);
          else lds_wait<younger>(ah[uv % RING], al[uv % RING]);
#ifndef MDM_EMU
          __builtin_amdgcn_sched_barrier(0);  // the MFMAs below must not be hoisted above the wait (rule 18)
#endif
          if constexpr ((ABL & 4) != 0) {
#ifndef MDM_EMU
            asm volatile("" ::"v"(al[uv % RING]), "v"(ah[uv % RING]), "v"(wh[ks]), "v"(wl[ks]));
#endif
          } else {
            acc[t] = mfma_bf16(al[uv % RING], wh[ks], acc[t]);
            acc[t] = mfma_bf16(ah[uv % RING], wl[ks], acc[t]);
            acc[t] = mfma_bf16(ah[uv % RING], wh[ks], acc[t]);
          }
#ifndef MDM_EMU
          __builtin_amdgcn_sched_barrier(0);  // keep the same-accumulator triple back to back (no filler inside)
#endif
          if constexpr (!(ABL & 2)) {
            if constexpr (uv < 4) piece_w(cw, uv, wbuf ^ 1);
            else if constexpr (uv < 8) piece_a(ca, uv - 4, abuf_ld);
          }
#ifndef MDM_EMU
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      });
      }
#undef X3_RD_A
#undef X3_RD_W
      advance_w(cw);
      advance_a(ca);
      if (ABL & 2) wait_vmem_all();
      else wait_vmem_upto3();
      if (!(ABL & 256)) wg_barrier();   // 256: experiment, no per-step barrier (only meaningful with 2 = no loads)
      abuf = (abuf == 2) ? 0 : abuf + 1;
      wbuf ^= 1;
    }

    // ---- epilogue.  In the accumulator layout a lane owns ONE column and 16 rows of each 32x32 sub-tile, which would
    // mean 4-byte (fp32) / 2-byte (planes) stores: the store tail is issue-bound (cdna_hip_programming.md T21).  So
    // each wave transposes 8 rows x 32 columns at a time (accumulator registers 4g..4g+3 of both lane halves) through
    // its private 1 KB LDS patch -- disjoint from the operand rings, which already hold the next tile's first stages
    // -- and writes 16 bytes per lane: lane -> (row = lane>>3, 4 consecutive columns).
    const float mult = (nc < ep.scale_cols) ? ep.col_scale : 1.f;
    const int m_end = min(M, m0 + rows_per_tile);
    float* patch = reinterpret_cast<float*>(lds + X3_PATCH_BASE) + wid * (X3_PATCH_BYTES / 4);  // [8][32] fp32
    const int prow = lane >> 3, pc4 = (lane & 7) * 4;
    const int n4 = ncol0 + pc4;                          // first of this lane's 4 columns in the row layout

    if (OUT_QKV) {
      // in_proj -> attention operand planes (attention_bf16x3.h).  rows_per_tile == S: tile row == token, tile_m == sequence.
      const int Dm = ep.D, Sq = ep.S, SPq = ep.qkv.SP, Hq = ep.qkv.H;
      const int which = ncol0 / Dm, hcol = ncol0 - which * Dm, head = hcol >> 7, d0 = hcol & 127;
      const size_t shq = (size_t)(m0 / rows_per_tile) * Hq + head;
      if (ncol0 < N && !(ABL & 1)) {
        if (which == 2) {
          // V^T: accumulator registers 8 s2 .. 8 s2 + 7 of a lane ARE positions 8h .. 8h+7 of 16-key group s2
#pragma unroll
          for (int t = 0; t < X3_MSUB; ++t) {
            if (t < ep.qkv.NKT) {
#pragma unroll
              for (int s2 = 0; s2 < 2; ++s2) {
                float vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  const int tok = 32 * t + mfma_row(8 * s2 + j, h);
                  vv[j] = (tok < Sq) ? acc[t][8 * s2 + j] + bias : 0.f;
                }
                bf16x8 vh8, vl8;
                split8(vv, vh8, vl8);
                const size_t o = ((shq * ep.qkv.NKT + t) * AX_HD + d0 + r) * 32 + 16 * s2 + 8 * h;
                *reinterpret_cast<bf16x8*>(ep.qkv.vh + o) = vh8;
                *reinterpret_cast<bf16x8*>(ep.qkv.vl + o) = vl8;
              }
            }
          }
        } else {
          bf16_t* dh = which == 0 ? ep.qkv.qh : ep.qkv.kh;
          bf16_t* dl = which == 0 ? ep.qkv.ql : ep.qkv.kl;
#pragma unroll
          for (int t = 0; t < X3_MSUB; ++t) {
            if (t < ep.qkv.NKT) {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int e = 0; e < 4; ++e) patch[((e + 4 * h) << 5) + r] = (acc[t][4 * g + e] + bias) * mult;
                wave_lds_fence();
                const int tok = 32 * t + 8 * g + prow;
                float4 v4 = ld4(&patch[prow * 32 + pc4]);
                if (tok >= Sq) v4 = zero4();
                const size_t o = (shq * SPq + tok) * AX_HD + d0 + pc4;
                split4_store(dh + o, dl + o, v4);
                wave_lds_fence();
              }
            }
          }
        }
      }
    } else {
      // residual tile: streamed two row sub-tiles ahead of its use through untracked loads (common.h gload16_async);
      // rows past the matrix are clamped (loaded, never stored)
      f32x4 rr[3][4];
      auto res_issue = [&](auto t_tag) __attribute__((always_inline)) {
        constexpr int t = decltype(t_tag)::value;
        if constexpr (HAS_RES && t < X3_MSUB) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int m = min(m0 + t * 32 + 8 * g + prow, M - 1);
            gload16_async(rr[t % 3][g], ep.res + (size_t)m * ep.ld + (n4 < N ? n4 : 0));
          }
        }
      };
      if (!(ABL & 1)) {
        res_issue(std::integral_constant<int, 0>{});
        res_issue(std::integral_constant<int, 1>{});
      }
      // 28 rounds (row sub-tile t, register group g): patch write -> 16-byte patch read -> store.  Round j+1's patch
      // writes (and the activation math feeding them) are issued between round j's read and its store, so the LDS round
      // trip of one round hides under the VALU work of the next (a wave's LDS operations execute in order).
      auto patch_write = [&](auto j_tag) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value, t = j / 4, g = j % 4;
        if constexpr (j < 4 * X3_MSUB) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[t][4 * g + e] + bias;
            if (ACT == ACT_GELU) x = gelu_erf_fast(x);
            else if (ACT == ACT_SILU) x = silu(x);
            patch[((e + 4 * h) << 5) + r] = x * mult;
          }
        }
      };
      patch_write(std::integral_constant<int, 0>{});
      static_for<4 * X3_MSUB>([&](auto j_tag) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value, t = j / 4, g = j % 4;
        if constexpr (HAS_RES && g == 0) {
          if (!(ABL & 1)) {
            res_issue(std::integral_constant<int, t + 2>{});
            constexpr int younger = 4 * ((X3_MSUB - 1 - t) < 2 ? (X3_MSUB - 1 - t) : 2);
            vmem_wait<younger>(rr[t % 3][0], rr[t % 3][1], rr[t % 3][2], rr[t % 3][3]);
          }
        }
        wave_lds_fence();
        float4 v4 = ld4(&patch[prow * 32 + pc4]);
        wave_lds_fence();
        patch_write(std::integral_constant<int, j + 1>{});
        if (!(ABL & 1)) {
          const int m = m0 + t * 32 + 8 * g + prow;
          if (m < m_end && n4 < N) {  // N % 4 == 0
            const size_t o = (size_t)m * ep.ld + n4;
            if (HAS_RES) {
              const f32x4 q4 = rr[t % 3][g];
              v4.x += q4[0]; v4.y += q4[1]; v4.z += q4[2]; v4.w += q4[3];
            }
            if (OUT_F32) st4(ep.out + o, v4);
            if (OUT_PLANES) split4_store(ep.oh + o, ep.ol + o, v4);
          }
        }
      });
    }
  }
  wait_vmem_all();  // the streams' last (unused) LDS-DMA stages must land before this workgroup's LDS is released
}

// rows per block tile: a whole number of sequences when the row space is sequence-structured (keeps the tile count a
// multiple of the sequence count -> no ragged last wave of workgroups), else the full 224.
inline int x3_rows_per_tile(int M, int seq_len) {
  if (seq_len > 0 && seq_len <= X3_TM && M % seq_len == 0) return (X3_TM / seq_len) * seq_len;
  return X3_TM;
}

// persistent grid: one workgroup per CU (the kernel needs ~156 KB of the CU's 160 KB LDS)
inline int x3_grid_limit() {
#ifdef MDM_EMU
  return 3;  // small, so that the emulator exercises the tile roll-over paths
#else
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
    cus = cus / 8 * 8;  // xcd_remap keeps a workgroup on one XCD only if the stride is a multiple of 8
    if (cus <= 0) cus = 8;
  }
  return cus;
#endif
}

template <int ACT, bool HAS_RES, bool OUT_F32, bool OUT_PLANES, bool OUT_QKV, int ABL = 0>
inline int launch_gemm_bf16x3_t(const X3Operand& A, const X3Operand& W, const X3Epilogue& ep, int M, int N, int K,
                                int rpt, hipStream_t stream) {
  const int tiles_m = (M + rpt - 1) / rpt, tiles_n = (N + X3_TN - 1) / X3_TN;
  const int total = tiles_m * tiles_n;
  auto kfn = &gemm_bf16x3_kernel<ACT, HAS_RES, OUT_F32, OUT_PLANES, OUT_QKV, ABL>;
#ifndef MDM_EMU
  static bool configured = false;  // per instantiation
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                            X3_LDS_BYTES) != hipSuccess)
      return -1;
    configured = true;
  }
#endif
  const int grid = std::min(total, x3_grid_limit());
  MDM_LAUNCH(kfn, dim3(grid), dim3(X3_THREADS), X3_LDS_BYTES, stream, A, W, ep, M, N, K, rpt, tiles_n, total);
  return 0;
}

// runtime (act, res, outputs) -> one of the instantiations the encoder needs
inline int launch_gemm_bf16x3(const X3Operand& A, const X3Operand& W, const X3Epilogue& ep, int M, int N, int K, int act,
                              int seq_len, hipStream_t s, int ablate = 0) {
  const bool res = ep.res != nullptr, f32 = ep.out != nullptr, pl = ep.oh != nullptr;
  const int rpt = x3_rows_per_tile(M, seq_len);
  if (ablate != 0) {  // profiling experiments (mdm_debug_set): only the plain fp32-out variant is instantiated
    if (!(act == ACT_NONE && !res && f32 && !pl)) return -2;
    switch (ablate) {
      case 1: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 1>(A, W, ep, M, N, K, rpt, s);
      case 2: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 2>(A, W, ep, M, N, K, rpt, s);
      case 4: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 4>(A, W, ep, M, N, K, rpt, s);
      case 32: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 32>(A, W, ep, M, N, K, rpt, s);
      case 258: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 258>(A, W, ep, M, N, K, rpt, s);
      case 514: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 514>(A, W, ep, M, N, K, rpt, s);
      case 770: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 770>(A, W, ep, M, N, K, rpt, s);
      case 3: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 3>(A, W, ep, M, N, K, rpt, s);
      case 771: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 771>(A, W, ep, M, N, K, rpt, s);
      case 34: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 34>(A, W, ep, M, N, K, rpt, s);
      default: return -2;
    }
  }
  if (act == ACT_NONE && !res && f32 && !pl) return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false>(A, W, ep, M, N, K, rpt, s);
  if (act == ACT_NONE && res && f32 && !pl) return launch_gemm_bf16x3_t<ACT_NONE, true, true, false, false>(A, W, ep, M, N, K, rpt, s);
  if (act == ACT_GELU && !res && !f32 && pl) return launch_gemm_bf16x3_t<ACT_GELU, false, false, true, false>(A, W, ep, M, N, K, rpt, s);
  if (act == ACT_GELU && !res && f32 && !pl) return launch_gemm_bf16x3_t<ACT_GELU, false, true, false, false>(A, W, ep, M, N, K, rpt, s);
  if (act == ACT_GELU && res && f32 && !pl) return launch_gemm_bf16x3_t<ACT_GELU, true, true, false, false>(A, W, ep, M, N, K, rpt, s);
  if (act == ACT_SILU && !res && f32 && !pl) return launch_gemm_bf16x3_t<ACT_SILU, false, true, false, false>(A, W, ep, M, N, K, rpt, s);
  return -2;
}

// in_proj: tokens [nseq*S][D] x W [3D][D] -> the attention operand planes; one sequence per tile (tile row == token)
inline int launch_gemm_bf16x3_qkv(const X3Operand& A, const X3Operand& W, const X3Epilogue& ep, int nseq, int S, int D,
                                  hipStream_t s) {
  if (S > X3_TM) return -2;
  return launch_gemm_bf16x3_t<ACT_NONE, false, false, false, true>(A, W, ep, nseq * S, 3 * D, D, S, s);
}

}  // namespace mdm

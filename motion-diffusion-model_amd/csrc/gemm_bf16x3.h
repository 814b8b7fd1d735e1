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
// 4 = no MFMAs, 8 = LDS-DMA issued as a burst at the top of the step instead of between the MFMA units.
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

    for (int kt = 0; kt < nk; ++kt) {
      const int abuf_ld = abuf >= 1 ? abuf - 1 : 2;  // (abuf + 2) % 3
      if ((ABL & 8) && !(ABL & 2)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) piece_w(cw, i, wbuf ^ 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) piece_a(ca, i, abuf_ld);
      }
      const unsigned char* sa = lds + abuf * X3_A_STAGE;
      const unsigned char* sw_ = lds + X3_W_BASE + wbuf * X3_W_STAGE;
      // 14 units per stage (2 k sub-steps x 7 row sub-tiles), each = 2 A-fragment reads + 3 MFMAs, software-pipelined
      // two units deep so that a ds_read's latency hides under the 6 MFMAs of the two units before it.  One LDS-DMA
      // piece rides behind each of the first eight units: W(g+1) pieces first, then A(g+2).
      bf16x8 wh[2], wl[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int co = ((ks * 2 + h) ^ sw) * 16;
        wh[ks] = *reinterpret_cast<const bf16x8*>(sw_ + fw + co);
        wl[ks] = *reinterpret_cast<const bf16x8*>(sw_ + X3_W_BYTES + fw + co);
      }
      bf16x8 ah[3], al[3];
#pragma unroll
      for (int u = 0; u < 2 * X3_MSUB + 2; ++u) {
        if (u < 2 * X3_MSUB) {
          const int ks = u / X3_MSUB, t = u - ks * X3_MSUB;
          const int co = ((ks * 2 + h) ^ sw) * 16;
          ah[u % 3] = *reinterpret_cast<const bf16x8*>(sa + fa + t * 2048 + co);
          al[u % 3] = *reinterpret_cast<const bf16x8*>(sa + X3_A_BYTES + fa + t * 2048 + co);
#ifndef MDM_EMU
          __builtin_amdgcn_sched_barrier(0);  // pin: these reads are issued two units ahead of their MFMAs
#endif
        }
        if (u >= 2) {
          const int uv = u - 2, ks = uv / X3_MSUB, t = uv - ks * X3_MSUB;
          if (ABL & 4) {
#ifndef MDM_EMU
            asm volatile("" ::"v"(al[uv % 3]), "v"(ah[uv % 3]), "v"(wh[ks]), "v"(wl[ks]));
#endif
          } else {
            acc[t] = mfma_bf16(al[uv % 3], wh[ks], acc[t]);
            acc[t] = mfma_bf16(ah[uv % 3], wl[ks], acc[t]);
            acc[t] = mfma_bf16(ah[uv % 3], wh[ks], acc[t]);
          }
          if (!(ABL & 8) && !(ABL & 2)) {
            if (uv < 4) piece_w(cw, uv, wbuf ^ 1);
            else if (uv < 8) piece_a(ca, uv - 4, abuf_ld);
          }
#ifndef MDM_EMU
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      }
      advance_w(cw);
      advance_a(ca);
      if (ABL & 2) wait_vmem_all();
      else wait_vmem_upto3();
      wg_barrier();
      abuf = (abuf == 2) ? 0 : abuf + 1;
      wbuf ^= 1;
    }

    // ---- epilogue.  In the accumulator layout a lane owns ONE column and 16 rows of each 32x32 sub-tile, which would
    // mean 4-byte (fp32) / 2-byte (planes) stores: the store tail is issue-bound (cdna_hip_programming.md T21).  So
    // each wave transposes 8 rows x 32 columns at a time (accumulator registers 4g..4g+3 of both lane halves) through
    // its private 1 KB LDS patch -- disjoint from the operand rings, which already hold the next tile's first stages
    // -- and writes 16 bytes per lane: lane -> (row = lane>>3, 4 consecutive columns).
    const int ncol0 = n0 + wid * 32;                     // this wave's first column (wave-uniform)
    const int nc = ncol0 + r;                            // this lane's column in the accumulator layout
    const float bias = (nc < N) ? ep.bias[nc] : 0.f;
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
#pragma unroll
      for (int t = 0; t < X3_MSUB; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[t][4 * g + e] + bias;
            if (ACT == ACT_GELU) x = gelu_erf_fast(x);
            else if (ACT == ACT_SILU) x = silu(x);
            patch[((e + 4 * h) << 5) + r] = x * mult;
          }
          wave_lds_fence();
          if (!(ABL & 1)) {
            const int m = m0 + t * 32 + 8 * g + prow;
            float4 v4 = ld4(&patch[prow * 32 + pc4]);
            if (m < m_end && n4 < N) {  // N % 4 == 0
              const size_t o = (size_t)m * ep.ld + n4;
              if (HAS_RES) {
                const float4 rr = ld4(ep.res + o);
                v4.x += rr.x; v4.y += rr.y; v4.z += rr.z; v4.w += rr.w;
              }
              if (OUT_F32) st4(ep.out + o, v4);
              if (OUT_PLANES) split4_store(ep.oh + o, ep.ol + o, v4);
            }
          }
          wave_lds_fence();
        }
      }
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
      case 5: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 5>(A, W, ep, M, N, K, rpt, s);
      case 6: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 6>(A, W, ep, M, N, K, rpt, s);
      case 8: return launch_gemm_bf16x3_t<ACT_NONE, false, true, false, false, 8>(A, W, ep, M, N, K, rpt, s);
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

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
// Machine mapping (gfx950): 512 threads = 8 waves, one workgroup per CU.  Block tile = up to 224 rows x 256 columns;
// the row extent is chosen by the host as a whole number of token sequences (S = 197 -> one sequence per tile), so
// the headline shape (256 sequences, N in {512, 1024, 1536}) gives exactly N/256 equal tiles per CU: no tail wave.
// Wave w owns columns [32w, 32w+32) x all 7 row sub-tiles (7 accumulators = 112 VGPRs); its W fragment feeds 21
// MFMAs per 16-deep k sub-step.  BK = 32; one LDS stage = Ah|Al [224][32] + Wh|Wl [256][32] bf16 = 60 KB, two stages.
//   * global -> LDS by global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass); the next stage is issued
//     before the MFMAs of the current one (2 waves/SIMD x 42 MFMAs = 2.7k matrix-pipe cycles of cover) and retired
//     by ONE s_waitcnt vmcnt(0) + s_barrier per K step;
//   * the LDS image of a plane tile is row-major with 64-byte rows; the 16-byte chunk index is XOR-swizzled with
//     (row>>2)&3 so the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots of the 256-byte bank row.
//     LDS-DMA writes lane-linearly, so the swizzle is applied to the per-lane SOURCE address and to the reads
//     (cdna_hip_programming.md rule 21);
//   * XCD-aware tile order (the column tiles of one row panel are adjacent inside an XCD's contiguous chunk) keeps a
//     sequence's activation planes in one L2; the weight planes of a layer (<= 3 MB) stay L2-resident everywhere.
#pragma once
#include "common.h"
#include "gemm_f32.h"  // ACT_* enums

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
constexpr int X3_LDS_BYTES = X3_W_BASE + X3_W_RING * X3_W_STAGE;  // 151552
constexpr int X3_A_GROUPS = X3_A_STAGE / 1024, X3_W_GROUPS = X3_W_STAGE / 1024;  // 28 + 32 LDS-DMA wave-instructions

struct X3Operand {
  const bf16_t* hi;
  const bf16_t* lo;
};

// v = act(acc + bias[n]) * (n < scale_cols ? col_scale : 1) + (res ? res[m][n] : 0) -> fp32 out and/or split planes.
struct X3Epilogue {
  float* out;        // [M][ld] or null
  const float* bias;
  const float* res;  // [M][ld] or null; may alias out
  bf16_t* oh;        // [M][ld] split planes or null
  bf16_t* ol;
  int ld;
  int scale_cols;
  float col_scale;
  int ablate;        // profiling experiments only (0 in production): 1 = no epilogue stores, 2 = no loads after stage 0,
                     // 4 = no MFMAs
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

template <int ACT, bool HAS_RES, bool OUT_F32, bool OUT_PLANES>
__global__ __launch_bounds__(X3_THREADS, 2) void gemm_bf16x3_kernel(X3Operand A, X3Operand W, X3Epilogue ep, int M, int N,
                                                                     int K, int rows_per_tile, int tiles_n) {
  MDM_DYN_SMEM(unsigned char, lds);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef MDM_EMU
  const int wid = tid >> 6;
#else
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int r = lane & 31, h = lane >> 5;

  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int m0 = tile_m * rows_per_tile, n0 = tile_n * X3_TN;

  // ---- LDS-DMA sources.  A stage image = 28 groups of 1 KB (16 rows x 64 B): groups 0-13 Ah, 14-27 Al; W stage image =
  // 32 groups: 0-15 Wh, 16-31 Wl.  Wave w issues groups w, w+8, ...  Lane -> (row = lane>>2, stored chunk = lane&3);
  // the logical k-chunk it fetches is stored ^ ((row>>2)&3) = (lane&3) ^ ((lane>>4)&3).  Rows past the tile / matrix
  // are clamped (their products are never stored).
  const int schunk = (lane & 3) ^ ((lane >> 4) & 3);
  const bf16_t* asrc[4];
  const bf16_t* wsrc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int qa = wid + 8 * i;  // < 28 for i < 3, and for i == 3 when wid < 4
    const int ga = (qa < 14) ? qa : qa - 14;
    const int arow = min(m0 + ga * 16 + (lane >> 2), M - 1);
    asrc[i] = ((qa < 14) ? A.hi : A.lo) + (size_t)arow * K + schunk * 8;
    const int qw = wid + 8 * i;  // < 32
    const int gw = (qw < 16) ? qw : qw - 16;
    const int wrow = min(n0 + gw * 16 + (lane >> 2), N - 1);
    wsrc[i] = ((qw < 16) ? W.hi : W.lo) + (size_t)wrow * K + schunk * 8;
  }
  auto stage_a = [&](int buf, int k0) {
    unsigned char* base = lds + buf * X3_A_STAGE + wid * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (wid + 8 * i < X3_A_GROUPS) glds16(asrc[i] + k0, base + i * 8192);
  };
  auto stage_w = [&](int buf, int k0) {
    unsigned char* base = lds + X3_W_BASE + buf * X3_W_STAGE + wid * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(wsrc[i] + k0, base + i * 8192);
  };

  // ---- fragment read offsets (bytes inside a plane tile): row*64 + ((ksub*2 + h) ^ sw)*16, sw = (row>>2)&3
  const int sw = (r >> 2) & 3;
  const int fa = r * 64;                 // + t*2048 per row sub-tile
  const int fw = (wid * 32 + r) * 64;

  f32x16 acc[X3_MSUB];
#pragma unroll
  for (int t = 0; t < X3_MSUB; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // Pipeline: at the top of step kt the LDS holds A(kt), W(kt) [landed, visible] and A(kt+1) [in flight].  Issue
  // W(kt+1) THEN A(kt+2); after the MFMAs wait until at most the A(kt+2) pieces (3 per wave; waves 0-3 issue a 4th that
  // is then also waited for) are outstanding -- loads retire in order, so W(kt+1) and A(kt+1) have landed -- and
  // barrier once.
  const int nk = K / X3_BK;
  stage_w(0, 0);
  stage_a(0, 0);
  if (nk > 1) {
    stage_a(1, X3_BK);
    wait_vmem_upto3();
  } else {
    wait_vmem_all();
  }
  wg_barrier();
  int abuf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (!(ep.ablate & 2)) {
      if (kt + 1 < nk) stage_w((kt + 1) & 1, (kt + 1) * X3_BK);
      if (kt + 2 < nk) stage_a(abuf >= 1 ? abuf - 1 : 2, (kt + 2) * X3_BK);  // (abuf + 2) % 3
    }
    const unsigned char* sa = lds + abuf * X3_A_STAGE;
    const unsigned char* sw_ = lds + X3_W_BASE + (kt & 1) * X3_W_STAGE;
    // 14 units per stage (2 k sub-steps x 7 row sub-tiles), each = 2 A-fragment reads + 3 MFMAs, software-pipelined
    // two units deep so that a ds_read's latency hides under the 6 MFMAs of the two units before it.
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
        const int v = u - 2, ks = v / X3_MSUB, t = v - ks * X3_MSUB;
        if (ep.ablate & 4) {
#ifndef MDM_EMU
          asm volatile("" ::"v"(al[v % 3]), "v"(ah[v % 3]), "v"(wh[ks]), "v"(wl[ks]));
#endif
          continue;
        }
        acc[t] = mfma_bf16(al[v % 3], wh[ks], acc[t]);
        acc[t] = mfma_bf16(ah[v % 3], wl[ks], acc[t]);
        acc[t] = mfma_bf16(ah[v % 3], wh[ks], acc[t]);
#ifndef MDM_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    }
    if (kt + 2 < nk) wait_vmem_upto3();
    else wait_vmem_all();
    wg_barrier();
    abuf = (abuf == 2) ? 0 : abuf + 1;
  }

  // ---- epilogue.  In the accumulator layout a lane owns ONE column and 16 rows of each 32x32 sub-tile, which would
  // mean 4-byte (fp32) / 2-byte (planes) stores: the store tail is issue-bound (cdna_hip_programming.md T21).  So each
  // wave transposes its sub-tile through a private 4 KB LDS patch (the operand rings are dead after the last barrier)
  // and writes 16 bytes per lane: lane -> (row = lane>>3 (+8 per pass), 4 consecutive columns).
  const int nc = n0 + wid * 32 + r;                    // this lane's column in the accumulator layout
  const float bias = (nc < N) ? ep.bias[nc] : 0.f;
  const float mult = (nc < ep.scale_cols) ? ep.col_scale : 1.f;
  const int m_end = min(M, m0 + rows_per_tile);
  float* patch = reinterpret_cast<float*>(lds) + wid * 1024;  // [32][32] fp32
  const int prow = lane >> 3, pc4 = (lane & 7) * 4;
  const int n4 = n0 + wid * 32 + pc4;                  // first of this lane's 4 columns in the row layout
#pragma unroll
  for (int t = 0; t < X3_MSUB; ++t) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      float v = acc[t][e] + bias;
      if (ACT == ACT_GELU) v = gelu_erf_fast(v);
      else if (ACT == ACT_SILU) v = silu(v);
      patch[mfma_row(e, h) * 32 + r] = v * mult;
    }
    wave_lds_fence();
    if (!(ep.ablate & 1)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = prow + 8 * i;
        const int m = m0 + t * 32 + row;
        float4 v = ld4(&patch[row * 32 + pc4]);
        if (m < m_end && n4 < N) {  // N % 4 == 0
          const size_t o = (size_t)m * ep.ld + n4;
          if (HAS_RES) {
            const float4 rr = ld4(ep.res + o);
            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
          }
          if (OUT_F32) st4(ep.out + o, v);
          if (OUT_PLANES) split4_store(ep.oh + o, ep.ol + o, v);
        }
      }
    }
    wave_lds_fence();
  }
}

// rows per block tile: a whole number of sequences when the row space is sequence-structured (keeps the tile count a
// multiple of the sequence count -> no ragged last wave of workgroups), else the full 224.
inline int x3_rows_per_tile(int M, int seq_len) {
  if (seq_len > 0 && seq_len <= X3_TM && M % seq_len == 0) return (X3_TM / seq_len) * seq_len;
  return X3_TM;
}

template <int ACT, bool HAS_RES, bool OUT_F32, bool OUT_PLANES>
inline int launch_gemm_bf16x3_t(const X3Operand& A, const X3Operand& W, const X3Epilogue& ep, int M, int N, int K,
                                int seq_len, hipStream_t stream) {
  const int rpt = x3_rows_per_tile(M, seq_len);
  const int tiles_m = (M + rpt - 1) / rpt, tiles_n = (N + X3_TN - 1) / X3_TN;
  auto kfn = &gemm_bf16x3_kernel<ACT, HAS_RES, OUT_F32, OUT_PLANES>;
#ifndef MDM_EMU
  static bool configured = false;  // per instantiation
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize,
                            X3_LDS_BYTES) != hipSuccess)
      return -1;
    configured = true;
  }
#endif
  MDM_LAUNCH(kfn, dim3(tiles_m * tiles_n), dim3(X3_THREADS), X3_LDS_BYTES, stream, A, W, ep, M, N, K, rpt, tiles_n);
  return 0;
}

// runtime (act, res, outputs) -> one of the instantiations the encoder needs
inline int launch_gemm_bf16x3(const X3Operand& A, const X3Operand& W, const X3Epilogue& ep, int M, int N, int K, int act,
                              int seq_len, hipStream_t s) {
  const bool res = ep.res != nullptr, f32 = ep.out != nullptr, pl = ep.oh != nullptr;
  if (act == ACT_NONE && !res && f32 && !pl) return launch_gemm_bf16x3_t<ACT_NONE, false, true, false>(A, W, ep, M, N, K, seq_len, s);
  if (act == ACT_NONE && res && f32 && !pl) return launch_gemm_bf16x3_t<ACT_NONE, true, true, false>(A, W, ep, M, N, K, seq_len, s);
  if (act == ACT_GELU && !res && !f32 && pl) return launch_gemm_bf16x3_t<ACT_GELU, false, false, true>(A, W, ep, M, N, K, seq_len, s);
  if (act == ACT_GELU && !res && f32 && !pl) return launch_gemm_bf16x3_t<ACT_GELU, false, true, false>(A, W, ep, M, N, K, seq_len, s);
  if (act == ACT_GELU && res && f32 && !pl) return launch_gemm_bf16x3_t<ACT_GELU, true, true, false>(A, W, ep, M, N, K, seq_len, s);
  if (act == ACT_SILU && !res && f32 && !pl) return launch_gemm_bf16x3_t<ACT_SILU, false, true, false>(A, W, ep, M, N, K, seq_len, s);
  return -2;
}

}  // namespace mdm

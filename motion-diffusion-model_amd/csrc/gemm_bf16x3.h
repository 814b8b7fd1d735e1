// Split-precision ("bf16x3") MFMA GEMM for the encoder's dense contractions:
//     C[m][n] = sum_k A[m][k] * W[n][k],   A = Ah + Al,  W = Wh + Wl  (bf16 planes, see common.h split_bf16)
//             ~ sum_k  Ah*Wh + Ah*Wl + Al*Wh            three v_mfma_f32_32x32x16_bf16 passes, fp32 accumulate.
//
// Why: the reference computes these `addmm`s in fp32 (model/mdm.py:77-84 -> torch TransformerEncoderLayer) and
// BASELINE's parity bar is 1e-3 max-abs over a 50-step guided trajectory.  gfx950 has no TF32; exact-fp32 MFMA
// peaks at 157 TFLOP/s, bf16 MFMA at 2.5 PFLOP/s, so three bf16 passes carry a ~2^-16-relative fp32 product at up to
// ~5x the fp32 rate (measured trajectory error ~4e-5).  Replaces in_proj / out_proj / linear1 /
// linear2; the 263-wide input/output projections stay on the exact-fp32 kernel (gemm_f32.h).
//
// Data layout: both operands are stored K-contiguous as two bf16 planes [rows][K] (hi, lo).  Weights are split
// once in mdm_prepare; activations are split by the PRODUCING kernel's epilogue (LayerNorm, attention, GELU), so the
// main loop is pure LDS-DMA + MFMA.
//
// Machine mapping (gfx950): 256 threads = 4 waves (2x2), block tile 128x128, wave tile 64x64 = 2x2 MFMA tiles
// (64 accumulator VGPRs), BK = 32.  One LDS stage = 4 planes x [128 rows][32 k] bf16 = 32 KB; two stages (64 KB,
// two workgroups per CU).
//   * global -> LDS by global_load_lds_dwordx4 (no staging VGPRs, no ds_write pass); the next stage is issued
//     before the MFMAs of the current one and retired by ONE s_waitcnt vmcnt(0) + s_barrier per K step;
//   * the LDS image of a plane tile is row-major with 64-byte rows; the 16-byte chunk index is XOR-swizzled with
//     (row>>2)&3 so the 16 lanes of a ds_read_b128 group hit 16 distinct 16-byte slots of the 256-byte bank row.
//     LDS-DMA writes lane-linearly, so the swizzle is applied to the per-lane SOURCE address and to the reads
//     (cdna_hip_programming.md rule 21);
//   * per 16-deep k sub-step a wave issues 8 ds_read_b128 and 12 MFMAs (each fragment feeds 2-3 products);
//   * XCD-aware tile order (n fastest inside an XCD's contiguous chunk) keeps an A row-panel in one L2.
#pragma once
#include "common.h"
#include "gemm_f32.h"  // LinearEpilogue

namespace mdm {

constexpr int X3_BM = 128, X3_BN = 128, X3_BK = 32, X3_THREADS = 256;
constexpr int X3_PLANE_BYTES = X3_BM * X3_BK * 2;  // 8 KB
constexpr int X3_STAGE_BYTES = 4 * X3_PLANE_BYTES;  // Ah, Al, Wh, Wl

struct X3Operand {
  const bf16_t* hi;
  const bf16_t* lo;
};

template <class EP>
__global__ __launch_bounds__(X3_THREADS, 2) void gemm_bf16x3_kernel(X3Operand A, X3Operand W, EP ep, int M, int N,
                                                                     int K, int tiles_n) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * X3_STAGE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = wid >> 1, wn = wid & 1;

  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int m0 = tile_m * X3_BM, n0 = tile_n * X3_BN;

  // ---- LDS-DMA source offsets.  One wave instruction fills 16 rows x 64 B of a plane tile: lane -> (row = lane>>2,
  // stored chunk = lane&3); the logical k-chunk it must fetch is stored ^ ((row>>2)&3) = (lane&3) ^ ((lane>>4)&3).
  const int srow = wid * 16 + (lane >> 2);
  const int schunk = (lane & 3) ^ ((lane >> 4) & 3);
  size_t a_off[2], w_off[2];
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    a_off[half] = (size_t)min(m0 + half * 64 + srow, M - 1) * K + schunk * 8;
    w_off[half] = (size_t)min(n0 + half * 64 + srow, N - 1) * K + schunk * 8;
  }
  unsigned char* const lds_wave = lds + wid * 16 * 64;  // + stage + plane + half*4096

  auto stage = [&](int buf, int k0) {
    unsigned char* base = lds_wave + buf * X3_STAGE_BYTES;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      glds16(A.hi + a_off[half] + k0, base + 0 * X3_PLANE_BYTES + half * 4096);
      glds16(A.lo + a_off[half] + k0, base + 1 * X3_PLANE_BYTES + half * 4096);
      glds16(W.hi + w_off[half] + k0, base + 2 * X3_PLANE_BYTES + half * 4096);
      glds16(W.lo + w_off[half] + k0, base + 3 * X3_PLANE_BYTES + half * 4096);
    }
  };

  // ---- fragment read offsets (bytes inside a plane tile): row*64 + ((ksub*2 + h) ^ sw)*16, sw = (row>>2)&3
  const int sw = (r >> 2) & 3;
  int fa[2], fw[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    fa[t] = (wm * 64 + t * 32 + r) * 64;
    fw[t] = (wn * 64 + t * 32 + r) * 64;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int nk = K / X3_BK;
  stage(0, 0);
  wait_vmem_all();
  wg_barrier();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * X3_BK);
    const unsigned char* sb = lds + cur * X3_STAGE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int co = ((ks * 2 + h) ^ sw) * 16;
      bf16x8 ah[2], al[2], wh[2], wl[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        ah[t] = *reinterpret_cast<const bf16x8*>(sb + 0 * X3_PLANE_BYTES + fa[t] + co);
        al[t] = *reinterpret_cast<const bf16x8*>(sb + 1 * X3_PLANE_BYTES + fa[t] + co);
        wh[t] = *reinterpret_cast<const bf16x8*>(sb + 2 * X3_PLANE_BYTES + fw[t] + co);
        wl[t] = *reinterpret_cast<const bf16x8*>(sb + 3 * X3_PLANE_BYTES + fw[t] + co);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = mfma_bf16(al[i], wh[j], acc[i][j]);
          acc[i][j] = mfma_bf16(ah[i], wl[j], acc[i][j]);
          acc[i][j] = mfma_bf16(ah[i], wh[j], acc[i][j]);
        }
    }
    wait_vmem_all();
    wg_barrier();
  }

  typename EP::Col cc[2];
  bool nv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + r;
    nv[j] = n < N;
    cc[j] = ep.col(nv[j] ? n : 0);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + wm * 64 + i * 32 + mfma_row(e, h);
      if (m < M) {
        const typename EP::Row rc = ep.row(m);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (nv[j]) ep.store(rc, cc[j], acc[i][j][e]);
      }
    }
}

template <class EP>
inline void launch_gemm_bf16x3(const X3Operand& A, const X3Operand& W, const EP& ep, int M, int N, int K,
                               hipStream_t stream) {
  const int tiles_m = (M + X3_BM - 1) / X3_BM, tiles_n = (N + X3_BN - 1) / X3_BN;
  auto kfn = &gemm_bf16x3_kernel<EP>;
  MDM_LAUNCH(kfn, dim3(tiles_m * tiles_n), dim3(X3_THREADS), 0, stream, A, W, ep, M, N, K, tiles_n);
}

}  // namespace mdm

// SEED of the next split-precision GEMM ("f16f6"), NOT used by the model yet:
//     C[m][n] = sum_k A[m][k] * W[n][k],   a*w ~ ah*wh + q6(ah)*q6(wl) + q6(al)*q6(wh)
// ah = fp16(a), al = a - ah (same for w); q6 = MX-FP6 (OCP Microscaling E2M3) with one power-of-two scale (E8M0) per 32
// consecutive k.  The main term is ONE v_mfma_f32_32x32x16_f16 pass; the two cross terms -- 2^-11 of it, so ~5 bits do -- run on
// v_mfma_scale_f32_32x32x64_f8f6f4, which gfx950 executes at four times the fp16 rate: 1.5 pass-equivalents instead of the
// three 16-bit passes of gemm_x3.h (measured at the matrix pipe: 1.75x; 50-step trajectory error 9.8e-5 vs 4.4e-5 for bf16x3, bar 1e-3:
// tools/precision_probe.py, tools/mx/, profiles/r01h_*).  Replaces nothing in the reference that gemm_x3.h does not
// already replace (the `addmm`s of model/mdm.py:77-84).
//
// This file pins down what the production kernel will be built on and is exercised by the emulator and GPU parity tests
// through mdm_linear_f16f6: the quantiser, the operand plane layout, the fragment <-> plane mapping of both instructions,
// and a REFERENCE kernel (one wave per 32x32 output tile, fragments straight from global memory -- correct, not fast).
//
// Operand planes of a matrix X [R][K] (K % 32 == 0):
//     h16 [R][K]             fp16 hi
//     rec [R][K/32][16 dw]   one 64-byte record per row and 32-k block, in four 16-byte chunks (element j of a part's code
//                            string at bit 6j of its dwords c0..c5):  [hi c0-c3 | lo c0-c3 | hi c4 c5, hi scale (127 + e), 0 |
//                            lo c4 c5, lo scale, 0] -- lane half h of the production k-loop reads chunk h in k sub-step 0 and
//                            chunk 2 + h in sub-step 1, exactly the two 16-byte reads it issues on today's lo plane
// i.e. 2 + 2 bytes per element, the byte geometry of gemm_x3.h's hi / lo planes (its LDS-DMA staging carries over).
// Fragments, per 32-k block: fp16 MFMA, k sub-step s in {0, 1}: lane (r = lane & 31, h = lane >> 5) holds k = 16 s + 8 h .. + 7 of
// row (column) r; scaled MFMA (K = 64 = both cross terms of the block): A lane (r, h) holds the block's 32 codes of hi (h = 0) /
// lo (h = 1) and that part's scale in byte 0; B lane (c, h) holds the codes of lo (h = 0) / hi (h = 1) of column c.
#pragma once
#include "../../motion-diffusion-model_amd/csrc/common.h"
#include "../../motion-diffusion-model_amd/csrc/gemm_f32.h"  // ACT_* enums

namespace mdm {

struct F6Planes {
  f16_t* h16;
  uint32_t* rec;
};
inline size_t f6_align256(size_t v) { return (v + 255) / 256 * 256; }
inline size_t f6_plane_bytes(int R, int K) { return f6_align256((size_t)R * K * 2) + f6_align256((size_t)R * (K / 32) * 64); }
inline F6Planes f6_carve(void* base, int R, int K) {
  return F6Planes{static_cast<f16_t*>(base),
                  reinterpret_cast<uint32_t*>(static_cast<char*>(base) + f6_align256((size_t)R * K * 2))};
}

// shared exponent e of an MX block with maximum magnitude amax: elements are x * 2^-e with |.| < 8 (E2M3 tops out at 7.5)
__host__ __device__ inline int mx_block_exp(float amax) {
  if (!(amax > 0.f)) return 0;
  int ex;
  frexpf(amax, &ex);          // amax = f * 2^ex, f in [0.5, 1): floor(log2 amax) = ex - 1
  return (ex - 1) - 2;        // E2M3's largest binade is [4, 8)
}
// FP6 E2M3 (1 sign, 2 exponent bits with bias 1, 3 mantissa bits; no inf / nan): round to nearest even, saturate at 7.5
__host__ __device__ inline uint32_t fp6_e2m3_encode(float x) {
  const uint32_t s = x < 0.f ? 1u : 0u;
  float v = fminf(fabsf(x), 7.5f);
  const int ex = v >= 4.f ? 2 : (v >= 2.f ? 1 : 0);          // [0, 2) shares the step 1/8 (subnormals + first binade)
  const float step = ex == 2 ? 0.5f : (ex == 1 ? 0.25f : 0.125f);
  const float q = fminf(rintf(v / step) * step, 7.5f);       // may land on the first value of the next binade
  uint32_t code;
  if (q < 1.f) code = (uint32_t)(q * 8.f);
  else {
    const int e2 = q >= 4.f ? 2 : (q >= 2.f ? 1 : 0);
    const float m = q * (e2 == 2 ? 0.25f : (e2 == 1 ? 0.5f : 1.f)) - 1.f;
    code = ((uint32_t)(e2 + 1) << 3) | (uint32_t)(m * 8.f);
  }
  return (s << 5) | code;
}
__host__ __device__ inline float fp6_e2m3_decode(uint32_t code) {
  const uint32_t s = (code >> 5) & 1u, e = (code >> 3) & 3u, m = code & 7u;
  const float v = (e == 0) ? m * 0.125f : (1.f + m * 0.125f) * (e == 1 ? 1.f : (e == 2 ? 2.f : 4.f));
  return s ? -v : v;
}

// fp32 [R][K] -> planes; one thread per (row, 32-k block).  (The production path will never run this: its planes are written
// by the producing GEMM's epilogue, where a wave's 32 output columns are exactly one block.)
__global__ __launch_bounds__(256) void pack_f16f6_kernel(const float* __restrict__ x, F6Planes p, int R, int K, int ld) {
  const int nb = K / 32;
  const int blk = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (blk >= R * nb) return;
  const int row = blk / nb, b = blk - row * nb;
  const float* src = x + (size_t)row * ld + b * 32;      // ld: row stride of the fp32 source (>= K)
  f16_t* dst = p.h16 + (size_t)row * K + b * 32;
  float mh = 0.f, ml = 0.f;
  for (int j = 0; j < 32; ++j) {
    const f16_t hv = (f16_t)src[j];
    dst[j] = hv;
    mh = fmaxf(mh, fabsf((float)hv));
    ml = fmaxf(ml, fabsf(src[j] - (float)hv));
  }
  const int eh = mx_block_exp(mh), el = mx_block_exp(ml);
  uint32_t wh[7] = {0, 0, 0, 0, 0, 0, 0}, wl[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 32; ++j) {
    const float hv = (float)dst[j];
    const uint32_t ch = fp6_e2m3_encode(ldexpf(hv, -eh)), cl = fp6_e2m3_encode(ldexpf(src[j] - hv, -el));
    const int bit = 6 * j, w = bit >> 5, o = bit & 31;
    wh[w] |= ch << o;
    wl[w] |= cl << o;
    if (o > 26) {
      wh[w + 1] |= ch >> (32 - o);
      wl[w + 1] |= cl >> (32 - o);
    }
  }
  uint32_t* rec = p.rec + (size_t)blk * 16;
  for (int w = 0; w < 4; ++w) {
    rec[w] = wh[w];
    rec[4 + w] = wl[w];
  }
  rec[8] = wh[4]; rec[9] = wh[5]; rec[10] = (uint32_t)(127 + eh); rec[11] = 0;
  rec[12] = wl[4]; rec[13] = wl[5]; rec[14] = (uint32_t)(127 + el); rec[15] = 0;
}

// Weights fp32 [N][K] -> the fragment-ordered planes the production k-loop streams straight into registers (gemm_x3.h
// header: [n/32][k/16][lane][16 B], rows >= N zero): plane `hi` holds fp16 values (unscaled: gemm_x3.h's own planes hold w * 2^8); plane `lo` holds,
// per 32-k block, for B-operand lane half 0 the codes of LO and for half 1 the codes of HI (the scaled MFMA pairs them with
// the A operand's hi / lo codes): k16 slot 2 kb: code dwords c0-c3, slot 2 kb + 1: [c4, c5, scale, 0].  One thread per (n, block).
__global__ __launch_bounds__(256) void pack_weight_f16f6_kernel(const float* __restrict__ w, p16_t* __restrict__ hi,
                                                                p16_t* __restrict__ lo, int N, int K) {
  const int npad = (N + 31) / 32 * 32, nb = K / 32;
  const int idx = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (idx >= npad * nb) return;
  const int n = idx / nb, kb = idx - n * nb;
  float hv[32], lv[32], mh = 0.f, ml = 0.f;
  for (int j = 0; j < 32; ++j) {
    const float x = (n < N) ? w[(size_t)n * K + kb * 32 + j] : 0.f;
    hv[j] = (float)(f16_t)x;
    lv[j] = x - hv[j];
    mh = fmaxf(mh, fabsf(hv[j]));
    ml = fmaxf(ml, fabsf(lv[j]));
  }
  const int eh = mx_block_exp(mh), el = mx_block_exp(ml);
  uint32_t ch[7] = {0, 0, 0, 0, 0, 0, 0}, cl[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 32; ++j) {
    const uint32_t a = fp6_e2m3_encode(ldexpf(hv[j], -eh)), b = fp6_e2m3_encode(ldexpf(lv[j], -el));
    const int bit = 6 * j, q = bit >> 5, o = bit & 31;
    ch[q] |= a << o;
    cl[q] |= b << o;
    if (o > 26) {
      ch[q + 1] |= a >> (32 - o);
      cl[q + 1] |= b >> (32 - o);
    }
  }
  const size_t wk16 = (size_t)(K / 16);
  for (int k8 = 0; k8 < 4; ++k8) {           // fp16 hi values: 8 consecutive k -> one 16-byte fragment piece
    const int lane = (n & 31) + 32 * (k8 & 1);
    const size_t o = (((size_t)(n >> 5) * wk16 + (size_t)(2 * kb + (k8 >> 1))) * 64 + lane) * 8;
    f16_t* dst = reinterpret_cast<f16_t*>(hi + o);
    for (int j = 0; j < 8; ++j) dst[j] = (f16_t)hv[8 * k8 + j];
  }
  for (int half = 0; half < 2; ++half) {
    const uint32_t* c = half == 0 ? cl : ch;
    const uint32_t sc = (uint32_t)(127 + (half == 0 ? el : eh));
    const int lane = (n & 31) + 32 * half;
    uint32_t* d0 = reinterpret_cast<uint32_t*>(lo + (((size_t)(n >> 5) * wk16 + (size_t)(2 * kb)) * 64 + lane) * 8);
    uint32_t* d1 = reinterpret_cast<uint32_t*>(lo + (((size_t)(n >> 5) * wk16 + (size_t)(2 * kb + 1)) * 64 + lane) * 8);
    d0[0] = c[0]; d0[1] = c[1]; d0[2] = c[2]; d0[3] = c[3];
    d1[0] = c[4]; d1[1] = c[5]; d1[2] = sc; d1[3] = 0;
  }
}

// REFERENCE kernel: one wave per 32x32 tile of out = act(A.W^T + bias) (+ res); ragged M / N: clamped loads, masked stores
template <int ACT>
__global__ __launch_bounds__(64) void gemm_f16f6_ref_kernel(F6Planes A, F6Planes W, const float* __restrict__ bias,
                                                            const float* res, float* out, int M, int N, int K) {
  const int lane = (int)threadIdx.x, r = lane & 31, h = lane >> 5;
  const int m0 = (int)blockIdx.y * 32, n0 = (int)blockIdx.x * 32;
  const int nb = K / 32;
  const size_t ar = (size_t)min(m0 + r, M - 1), wr = (size_t)min(n0 + r, N - 1);
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  // One 32-k block per iteration -- the production kernel's k step (gemm_x3.h stages are 32 deep): two fp16 MFMAs for the
  // main term and ONE scaled MFMA whose K = 64 holds BOTH cross terms of the block: lane half 0 supplies q6(ah) (A) against
  // q6(wl) (B), lane half 1 supplies q6(al) against q6(wh); the K reduction adds the two.
  for (int kb = 0; kb < nb; ++kb) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const f16x8 a = *reinterpret_cast<const f16x8*>(A.h16 + ar * K + kb * 32 + 16 * s + 8 * h);
      const f16x8 w = *reinterpret_cast<const f16x8*>(W.h16 + wr * K + kb * 32 + 16 * s + 8 * h);
      acc = mfma_f16(a, w, acc);
    }
    const uint32_t* ra = A.rec + (ar * nb + kb) * 16;
    const uint32_t* rw = W.rec + (wr * nb + kb) * 16;
    i32x8 a6, w6;
    const int pa = h, pw = 1 - h;            // A: h = 0 codes of hi, h = 1 of lo;  W: h = 0 codes of lo, h = 1 of hi
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      a6[w] = (int)ra[4 * pa + w];
      w6[w] = (int)rw[4 * pw + w];
    }
    a6[4] = (int)ra[8 + 4 * pa]; a6[5] = (int)ra[9 + 4 * pa];
    w6[4] = (int)rw[8 + 4 * pw]; w6[5] = (int)rw[9 + 4 * pw];
    a6[6] = a6[7] = w6[6] = w6[7] = 0;
    acc = mfma_mx_fp6(a6, w6, acc, (int)ra[10 + 4 * pa], (int)rw[10 + 4 * pw]);
  }
  const int n = n0 + r;
  if (n >= N) return;
  const float bv = bias[n];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int m = m0 + (i & 3) + 8 * (i >> 2) + 4 * h;
    if (m < M) {
      float v = acc[i] + bv;
      if constexpr (ACT == ACT_GELU) v = gelu_erf(v);
      if constexpr (ACT == ACT_SILU) v = silu(v);
      if (res != nullptr) v += res[(size_t)m * N + n];
      out[(size_t)m * N + n] = v;
    }
  }
}

}  // namespace mdm

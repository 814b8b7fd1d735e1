// GEMM-level check of the split "fp16 + 2 x MX-FP6" product on the hardware (see tools/mx/mx_probe.hip and DESIGN.md section 8):
// device pack kernel (fp32 -> fp16 hi plane, FP6/E2M3 codes + E8M0 block scales of hi and lo) and a deliberately simple GEMM
// (one wave per 32x32 output tile, fragments loaded straight from global memory) using the real instructions, compared with
// fp64 and with host emulations of this scheme and of the production f16x3 scheme.  Test infrastructure, not product code;
// what it pins down for the production kernel: the quantiser, the plane layouts that feed the fragments, the error per GEMM.
//   hipcc --offload-arch=gfx950 -O3 tools/mx/mx_gemm_probe.hip -o build/mx_gemm_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---- the quantiser (host + device): one MX block = 32 consecutive k of one row; element format FP6 E2M3 (max 7.5)
__host__ __device__ inline int mx_block_exp(float amax) {   // shared exponent e: elements are x * 2^-e, |.| < 8
  if (!(amax > 0.f)) return 0;
  int ex;
  frexpf(amax, &ex);          // amax = f * 2^ex, f in [0.5, 1)  ->  floor(log2 amax) = ex - 1
  return (ex - 1) - 2;
}
__host__ __device__ inline int fp6_e2m3_encode(float x) {   // x already scaled; round to nearest even, saturate at 7.5
  const int s = x < 0.f;
  float v = fabsf(x);
  if (v > 7.5f) v = 7.5f;
  int ex = 0;
  if (v >= 4.f) ex = 2; else if (v >= 2.f) ex = 1;
  const float step = ldexpf(1.f, ex - 3);
  float q = rintf(v / step) * step;   // may land on the next binade's first value: encoded below from q itself
  if (q > 7.5f) q = 7.5f;
  int code;
  if (q < 1.f) code = (int)(q * 8.f);
  else {
    const int e2 = q >= 4.f ? 2 : (q >= 2.f ? 1 : 0);
    code = ((e2 + 1) << 3) | (int)((q / ldexpf(1.f, e2) - 1.f) * 8.f);
  }
  return (s << 5) | code;
}
__host__ __device__ inline float fp6_e2m3_decode(int code) {
  const int s = code >> 5, e = (code >> 3) & 3, m = code & 7;
  const float v = (e == 0) ? m * 0.125f : ldexpf(1.f + m * 0.125f, e - 1);
  return s ? -v : v;
}

// planes of one operand matrix X [R][K] (K % 64 == 0)
struct Planes {
  _Float16* h16;        // [R][K]
  uint32_t* c6h;        // [R][K/32][6]   FP6 codes of the hi part, element j of the block at bit 6j
  uint32_t* c6l;        // [R][K/32][6]   ... of the lo part
  uint8_t* sh;          // [R][K/32]      E8M0 scale bytes (127 + e)
  uint8_t* sl;
};

__global__ void pack_kernel(const float* __restrict__ x, Planes p, int R, int K) {
  const int blk = blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (row, 32-block): a probe, not a tuned kernel
  const int nb = K / 32;
  if (blk >= R * nb) return;
  const int row = blk / nb, b = blk - row * nb;
  const float* src = x + (size_t)row * K + b * 32;
  float hi[32], lo[32], mh = 0.f, ml = 0.f;
  for (int j = 0; j < 32; ++j) {
    const _Float16 h = (_Float16)src[j];
    p.h16[(size_t)row * K + b * 32 + j] = h;
    hi[j] = (float)h;
    lo[j] = src[j] - hi[j];
    mh = fmaxf(mh, fabsf(hi[j]));
    ml = fmaxf(ml, fabsf(lo[j]));
  }
  const int eh = mx_block_exp(mh), el = mx_block_exp(ml);
  uint32_t wh[6] = {0, 0, 0, 0, 0, 0}, wl[6] = {0, 0, 0, 0, 0, 0};
  for (int j = 0; j < 32; ++j) {
    const uint32_t ch = (uint32_t)fp6_e2m3_encode(ldexpf(hi[j], -eh)), cl = (uint32_t)fp6_e2m3_encode(ldexpf(lo[j], -el));
    const int bit = 6 * j, w = bit >> 5, o = bit & 31;
    wh[w] |= ch << o;
    wl[w] |= cl << o;
    if (o > 26) { wh[w + 1] |= ch >> (32 - o); wl[w + 1] |= cl >> (32 - o); }
  }
  for (int w = 0; w < 6; ++w) { p.c6h[(size_t)blk * 6 + w] = wh[w]; p.c6l[(size_t)blk * 6 + w] = wl[w]; }
  p.sh[blk] = (uint8_t)(127 + eh);
  p.sl[blk] = (uint8_t)(127 + el);
}

// C[m][n] = sum_k A[m][k] W[n][k]; one wave per 32x32 tile; lane = (r = lane & 31, h = lane >> 5)
__global__ __launch_bounds__(64) void gemm_f16f6_naive(Planes A, Planes W, float* __restrict__ C, int M, int N, int K) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  const int nb = K / 32;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  const size_t ar = (size_t)(m0 + r), wr = (size_t)(n0 + r);
  for (int k0 = 0; k0 < K; k0 += 64) {
    // main term: 4 x v_mfma_f32_32x32x16_f16; A-operand lane holds k = 16 s + 8 h .. + 7 of row r (B: of column r)
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const f16x8 a = *reinterpret_cast<const f16x8*>(A.h16 + ar * K + k0 + 16 * s + 8 * h);
      const f16x8 w = *reinterpret_cast<const f16x8*>(W.h16 + wr * K + k0 + 16 * s + 8 * h);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, w, acc, 0, 0, 0);
    }
    // cross terms: lane holds the 32 consecutive k of block (k0 / 32 + h) of its row / column, plus that block's scale
    const size_t ab = ar * nb + (k0 >> 5) + h, wb = wr * nb + (k0 >> 5) + h;
    i32x8 ah6, al6, wh6, wl6;
#pragma unroll
    for (int w = 0; w < 6; ++w) {
      ah6[w] = (int)A.c6h[ab * 6 + w]; al6[w] = (int)A.c6l[ab * 6 + w];
      wh6[w] = (int)W.c6h[wb * 6 + w]; wl6[w] = (int)W.c6l[wb * 6 + w];
    }
    ah6[6] = ah6[7] = al6[6] = al6[7] = wh6[6] = wh6[7] = wl6[6] = wl6[7] = 0;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ah6, wl6, acc, 2, 2, 0, (int)A.sh[ab], 0, (int)W.sl[wb]);
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(al6, wh6, acc, 2, 2, 0, (int)A.sl[ab], 0, (int)W.sh[wb]);
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    C[(size_t)(m0 + row) * N + n0 + r] = acc[i];
  }
}

static Planes alloc_planes(int R, int K) {
  Planes p;
  const size_t nb = (size_t)R * (K / 32);
  CK(hipMalloc(&p.h16, (size_t)R * K * 2)); CK(hipMalloc(&p.c6h, nb * 24)); CK(hipMalloc(&p.c6l, nb * 24));
  CK(hipMalloc(&p.sh, nb)); CK(hipMalloc(&p.sl, nb));
  return p;
}

static float bf16_round(float x) {   // round to nearest even bf16
  uint32_t u; std::memcpy(&u, &x, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  u &= 0xFFFF0000u;
  float y; std::memcpy(&y, &u, 4);
  return y;
}

struct HostQ {   // host emulation of the quantiser for one row: hi (fp16), q6(hi), q6(lo) as plain floats
  std::vector<float> hi, qh, ql;
};
static void host_quant(const float* x, int K, HostQ& q) {
  q.hi.resize(K); q.qh.resize(K); q.ql.resize(K);
  for (int b = 0; b < K / 32; ++b) {
    float lo[32], mh = 0.f, ml = 0.f;
    for (int j = 0; j < 32; ++j) {
      const float h = (float)(_Float16)x[b * 32 + j];
      q.hi[b * 32 + j] = h;
      lo[j] = x[b * 32 + j] - h;
      mh = std::fmax(mh, std::fabs(h)); ml = std::fmax(ml, std::fabs(lo[j]));
    }
    const int eh = mx_block_exp(mh), el = mx_block_exp(ml);
    for (int j = 0; j < 32; ++j) {
      q.qh[b * 32 + j] = std::ldexp(fp6_e2m3_decode(fp6_e2m3_encode(std::ldexp(q.hi[b * 32 + j], -eh))), eh);
      q.ql[b * 32 + j] = std::ldexp(fp6_e2m3_decode(fp6_e2m3_encode(std::ldexp(lo[j], -el))), el);
    }
  }
}

static void run_case(const char* name, int M, int N, int K, float outlier, unsigned seed) {
  std::mt19937 rng(seed);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> a((size_t)M * K), w((size_t)N * K);
  for (auto& v : a) v = nd(rng);
  if (outlier > 0.f)                                      // a few large activations per row (pre-norm residual sums have them)
    for (int m = 0; m < M; ++m)
      for (int t = 0; t < 4; ++t) a[(size_t)m * K + (rng() % K)] *= outlier;
  const float ws = 1.0f / std::sqrt((float)K);
  for (auto& v : w) v = nd(rng) * ws;
  float *da, *dw, *dc;
  CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&dw, w.size() * 4)); CK(hipMalloc(&dc, (size_t)M * N * 4));
  CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice));
  Planes pa = alloc_planes(M, K), pw = alloc_planes(N, K);
  hipLaunchKernelGGL(pack_kernel, dim3((M * (K / 32) + 255) / 256), dim3(256), 0, 0, da, pa, M, K);
  hipLaunchKernelGGL(pack_kernel, dim3((N * (K / 32) + 255) / 256), dim3(256), 0, 0, dw, pw, N, K);
  hipLaunchKernelGGL(gemm_f16f6_naive, dim3(N / 32, M / 32), dim3(64), 0, 0, pa, pw, dc, M, N, K);
  CK(hipDeviceSynchronize());
  std::vector<float> c((size_t)M * N);
  CK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
  // host: exact fp64, emulated f16+f6x2, emulated f16x3 -- on a sample of rows (all columns)
  std::vector<HostQ> wq(N);
  for (int n = 0; n < N; ++n) host_quant(&w[(size_t)n * K], K, wq[n]);
  double se_dev = 0, se_emu = 0, se_b3 = 0, s2 = 0, mx_dev = 0, mx_emu = 0, mx_b3 = 0, mx_de = 0;
  size_t cnt = 0;
  for (int m = 0; m < M; m += std::max(1, M / 64)) {
    HostQ aq;
    host_quant(&a[(size_t)m * K], K, aq);
    for (int n = 0; n < N; ++n) {
      double ex = 0, emu = 0, b3 = 0;
      for (int k = 0; k < K; ++k) {
        const float av = a[(size_t)m * K + k], wv = w[(size_t)n * K + k];
        ex += (double)av * wv;
        emu += (double)aq.hi[k] * wq[n].hi[k] + (double)aq.qh[k] * wq[n].ql[k] + (double)aq.ql[k] * wq[n].qh[k];
        const float abh = bf16_round(av), abl = bf16_round(av - abh), wbh = bf16_round(wv), wbl = bf16_round(wv - wbh);
        b3 += (double)abh * wbh + (double)abh * wbl + (double)abl * wbh;
      }
      const double d = c[(size_t)m * N + n];
      se_dev += (d - ex) * (d - ex); se_emu += (emu - ex) * (emu - ex); se_b3 += (b3 - ex) * (b3 - ex); s2 += ex * ex;
      mx_dev = std::fmax(mx_dev, std::fabs(d - ex)); mx_emu = std::fmax(mx_emu, std::fabs(emu - ex));
      mx_b3 = std::fmax(mx_b3, std::fabs(b3 - ex)); mx_de = std::fmax(mx_de, std::fabs(d - emu));
      ++cnt;
    }
  }
  const double rms = std::sqrt(s2 / cnt);
  printf("%-28s M=%d N=%d K=%d  |C| rms %.3f\n", name, M, N, K, rms);
  printf("    f16+f6x2 on the GPU   : rms err %.3e (%.2e of rms)  max-abs %.3e   | device vs host emulation max %.3e\n",
         std::sqrt(se_dev / cnt), std::sqrt(se_dev / cnt) / rms, mx_dev, mx_de);
  printf("    f16+f6x2 host emulation: rms err %.3e (%.2e of rms)  max-abs %.3e\n", std::sqrt(se_emu / cnt), std::sqrt(se_emu / cnt) / rms, mx_emu);
  printf("    f16x3   host emulation: rms err %.3e (%.2e of rms)  max-abs %.3e\n", std::sqrt(se_b3 / cnt), std::sqrt(se_b3 / cnt) / rms, mx_b3);
  CK(hipFree(da)); CK(hipFree(dw)); CK(hipFree(dc));
}

int main() {
  run_case("out_proj-like", 2048, 512, 512, 0.f, 1);
  run_case("linear2-like (K=1024)", 2048, 512, 1024, 0.f, 2);
  run_case("pre-norm rows with outliers", 2048, 512, 512, 12.f, 3);
  return 0;
}

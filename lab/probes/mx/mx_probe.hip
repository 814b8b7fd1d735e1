// Probe for the NEXT GEMM scheme: an fp32 product carried by ONE fp16 MFMA pass plus two cross terms on
// block-scaled MX-FP6 (E2M3) operands, v_mfma_scale_f32_32x32x64_f8f6f4, which gfx950 runs at four times the fp16 rate:
//     a*w ~ ah*wh + q6(ah)*q6(wl) + q6(al)*q6(wh)       ah = fp16(a), al = a - ah (same for w); q6 = MX-FP6 with one 2^e per 32 k
// i.e. 1.5 pass-equivalents instead of the three bf16 passes of gemm_x3.h (tools/precision_probe.py: 9.8e-5 max-abs on
// the 50-step guided trajectory; f16x3 4.4e-5; bar 1e-3).  Stand-alone (hipcc tools/mx/mx_probe.hip -o build/mx_probe):
//   part A  operand / scale semantics of the scaled MFMA against a host reference (per-lane random FP6 codes + E8M0 scales)
//   part B  matrix-pipe throughput of the instruction mixes, K = 64 per group and accumulator:
//             f16x3      12 x v_mfma_f32_32x32x16_bf16            (today)
//             f16+f6x2     4 x v_mfma_f32_32x32x16_f16 + 2 x scaled 32x32x64 FP6
//             f16+f8x2     4 x f16 + 2 x scaled 32x32x64 FP8
//             f6 only      2 x scaled FP6 (raw rate of the new instruction)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- part A
__global__ void mx_one(const i32x8* a, const i32x8* b, const int* sa, const int* sb, float* d) {
  const int lane = threadIdx.x;
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[lane], b[lane], c, 2 /*A: FP6 E2M3*/, 2 /*B*/, 0, sa[lane], 0, sb[lane]);
  for (int i = 0; i < 16; ++i) d[lane * 16 + i] = c[i];
}

static float fp6_e2m3(int code) {   // OCP MX v1.0: 1 sign, 2 exponent (bias 1), 3 mantissa; no inf / nan
  const int s = code >> 5, e = (code >> 3) & 3, m = code & 7;
  const float v = (e == 0) ? m / 8.0f : std::ldexp(1.0f + m / 8.0f, e - 1);
  return s ? -v : v;
}

static int part_a() {
  std::vector<i32x8> a(64), b(64);
  std::vector<int> sa(64), sb(64);
  std::vector<float> av(64 * 32), bv(64 * 32);
  uint32_t rng = 12345u;
  auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  for (int l = 0; l < 64; ++l) {
    uint64_t bits_a[3] = {0, 0, 0}, bits_b[3] = {0, 0, 0};   // 192 bits each
    for (int j = 0; j < 32; ++j) {
      const int ca = next() & 63, cb = next() & 63;
      av[l * 32 + j] = fp6_e2m3(ca);
      bv[l * 32 + j] = fp6_e2m3(cb);
      const int bit = 6 * j;
      for (int t = 0; t < 6; ++t) {
        if ((ca >> t) & 1) bits_a[(bit + t) >> 6] |= 1ull << ((bit + t) & 63);
        if ((cb >> t) & 1) bits_b[(bit + t) >> 6] |= 1ull << ((bit + t) & 63);
      }
    }
    for (int w = 0; w < 6; ++w) {
      a[l][w] = (int)(uint32_t)(bits_a[w >> 1] >> (32 * (w & 1)));
      b[l][w] = (int)(uint32_t)(bits_b[w >> 1] >> (32 * (w & 1)));
    }
    a[l][6] = a[l][7] = b[l][6] = b[l][7] = 0;
    sa[l] = 120 + (int)(next() % 12);   // E8M0: 2^(byte - 127): 2^-7 .. 2^4
    sb[l] = 122 + (int)(next() % 9);
  }
  i32x8 *da, *db; int *dsa, *dsb; float* dd;
  CK(hipMalloc(&da, 64 * sizeof(i32x8))); CK(hipMalloc(&db, 64 * sizeof(i32x8)));
  CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 64 * 16 * 4));
  CK(hipMemcpy(da, a.data(), 64 * sizeof(i32x8), hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), 64 * sizeof(i32x8), hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mx_one, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  CK(hipDeviceSynchronize());
  std::vector<float> d(64 * 16);
  CK(hipMemcpy(d.data(), dd, 64 * 16 * 4, hipMemcpyDeviceToHost));
  // reference: D[i][j] = sum over k-half h and element jj of  2^(sa[i,h]-127) a[i,h,jj] * 2^(sb[j,h]-127) b[j,h,jj]
  double worst = 0, worst_t = 0, big = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int reg = 0; reg < 16; ++reg) {
      const int col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      double ref = 0, ref_t = 0;
      for (int h = 0; h < 2; ++h)
        for (int jj = 0; jj < 32; ++jj) {
          ref += std::ldexp((double)av[(row + 32 * h) * 32 + jj], sa[row + 32 * h] - 127) *
                 std::ldexp((double)bv[(col + 32 * h) * 32 + jj], sb[col + 32 * h] - 127);
          ref_t += std::ldexp((double)av[(col + 32 * h) * 32 + jj], sa[col + 32 * h] - 127) *
                   std::ldexp((double)bv[(row + 32 * h) * 32 + jj], sb[row + 32 * h] - 127);
        }
      worst = std::fmax(worst, std::fabs(d[lane * 16 + reg] - ref));
      worst_t = std::fmax(worst_t, std::fabs(d[lane * 16 + reg] - ref_t));
      big = std::fmax(big, std::fabs(ref));
    }
  printf("part A: scaled FP6 32x32x64, per-lane E8M0 scales: max |D - ref| = %.3e (A rows / B cols), %.3e (transposed), |ref| max %.3e -> %s\n",
         worst, worst_t, big, worst < 1e-5 * big ? "operand + scale semantics as assumed" : "MISMATCH");
  return worst < 1e-5 * big ? 0 : 1;
}

// ---------------------------------------------------------------- part A2: the 16x16x128 form (the GEMM's 16-row last sub-tile)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void mx_one16(const i32x8* a, const i32x8* b, const int* sa, const int* sb, float* d) {
  const int lane = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[lane], b[lane], c, 2, 2, 0, sa[lane], 0, sb[lane]);
  for (int i = 0; i < 4; ++i) d[lane * 4 + i] = c[i];
}

static int part_a2() {
  std::vector<i32x8> a(64), b(64);
  std::vector<int> sa(64), sb(64);
  std::vector<float> av(64 * 32), bv(64 * 32);
  uint32_t rng = 4242u;
  auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
  for (int l = 0; l < 64; ++l) {
    uint64_t bits_a[3] = {0, 0, 0}, bits_b[3] = {0, 0, 0};
    for (int j = 0; j < 32; ++j) {
      const int ca = next() & 63, cb = next() & 63;
      av[l * 32 + j] = fp6_e2m3(ca);
      bv[l * 32 + j] = fp6_e2m3(cb);
      const int bit = 6 * j;
      for (int t = 0; t < 6; ++t) {
        if ((ca >> t) & 1) bits_a[(bit + t) >> 6] |= 1ull << ((bit + t) & 63);
        if ((cb >> t) & 1) bits_b[(bit + t) >> 6] |= 1ull << ((bit + t) & 63);
      }
    }
    for (int w = 0; w < 6; ++w) {
      a[l][w] = (int)(uint32_t)(bits_a[w >> 1] >> (32 * (w & 1)));
      b[l][w] = (int)(uint32_t)(bits_b[w >> 1] >> (32 * (w & 1)));
    }
    a[l][6] = a[l][7] = b[l][6] = b[l][7] = 0;
    sa[l] = 121 + (int)(next() % 10);
    sb[l] = 123 + (int)(next() % 8);
  }
  i32x8 *da, *db; int *dsa, *dsb; float* dd;
  CK(hipMalloc(&da, 64 * sizeof(i32x8))); CK(hipMalloc(&db, 64 * sizeof(i32x8)));
  CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dd, 64 * 4 * 4));
  CK(hipMemcpy(da, a.data(), 64 * sizeof(i32x8), hipMemcpyHostToDevice));
  CK(hipMemcpy(db, b.data(), 64 * sizeof(i32x8), hipMemcpyHostToDevice));
  CK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mx_one16, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  CK(hipDeviceSynchronize());
  std::vector<float> d(64 * 4);
  CK(hipMemcpy(d.data(), dd, 64 * 4 * 4, hipMemcpyDeviceToHost));
  // assumed: lane = (row|col = lane & 15, k block g = lane >> 4), 32 consecutive k per lane; D[reg]: row 4 (lane >> 4) + reg, col lane & 15
  double worst = 0, big = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int reg = 0; reg < 4; ++reg) {
      const int col = lane & 15, row = 4 * (lane >> 4) + reg;
      double ref = 0;
      for (int g = 0; g < 4; ++g)
        for (int jj = 0; jj < 32; ++jj)
          ref += std::ldexp((double)av[(row + 16 * g) * 32 + jj], sa[row + 16 * g] - 127) *
                 std::ldexp((double)bv[(col + 16 * g) * 32 + jj], sb[col + 16 * g] - 127);
      worst = std::fmax(worst, std::fabs(d[lane * 4 + reg] - ref));
      big = std::fmax(big, std::fabs(ref));
    }
  printf("part A2: scaled FP6 16x16x128, lane = (row|col = lane & 15, k block = lane >> 4): max |D - ref| = %.3e, |ref| max %.3e -> %s\n",
         worst, big, worst < 1e-5 * big ? "as assumed" : "MISMATCH");
  return worst < 1e-5 * big ? 0 : 1;
}

// ---------------------------------------------------------------- part B
template <int MODE>
__global__ __launch_bounds__(256, 2) void mix_kernel(float* out, int iters, const int* seeds) {
  constexpr int NACC = 4;
  f32x16 acc[NACC];
  for (int t = 0; t < NACC; ++t)
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  // operands: data-dependent, finite, of moderate size (the chip's clock under MFMA load depends on the operand bits)
  f16x8 ah[4], wh;
  s16x8 bh[4], bl[4], wbh, wbl;
  i32x8 q6a, q6b, q6c;
  for (int k = 0; k < 4; ++k)
    for (int i = 0; i < 8; ++i) {
      const float v = ((seeds[(tid + 17 * i + 5 * k) & 1023] & 0xFFFF) - 32768) * (1.0f / 32768.0f);
      ah[k][i] = (_Float16)v;
      bh[k][i] = (short)(__float_as_uint(v) >> 16);
      bl[k][i] = (short)(__float_as_uint(v * 0.0039f) >> 16);
    }
  for (int i = 0; i < 8; ++i) {
    const float v = ((seeds[(tid * 3 + i) & 1023] & 0xFFFF) - 32768) * (1.0f / 32768.0f);
    wh[i] = (_Float16)v;
    wbh[i] = (short)(__float_as_uint(v) >> 16);
    wbl[i] = (short)(__float_as_uint(v * 0.0039f) >> 16);
    q6a[i] = seeds[(tid + i) & 1023];
    q6b[i] = seeds[(tid * 7 + i) & 1023];
    q6c[i] = seeds[(tid * 11 + i) & 1023];
  }
  const int sc = 120 + (tid & 7);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      if constexpr (MODE == 0) {          // f16x3: K = 64 -> 4 k sub-steps x 3 products
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl[k], wbh, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[k], wbl, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh[k], wbh, acc[t], 0, 0, 0);
        }
      } else {
        if constexpr (MODE == 1 || MODE == 2) {
#pragma unroll
          for (int k = 0; k < 4; ++k) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[k], wh, acc[t], 0, 0, 0);
        }
        if constexpr (MODE == 1 || MODE == 3) {
          acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(q6a, q6b, acc[t], 2, 2, 0, sc, 0, sc);
          acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(q6c, q6a, acc[t], 2, 2, 0, sc, 0, sc);
        } else {
          acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(q6a, q6b, acc[t], 0, 0, 0, sc, 0, sc);
          acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(q6c, q6a, acc[t], 0, 0, 0, sc, 0, sc);
        }
      }
    }
  }
  float s = 0.f;
  for (int t = 0; t < NACC; ++t)
    for (int i = 0; i < 16; ++i) s += acc[t][i];
  out[tid] = s;
}

template <int MODE>
static double run_mix(const char* name, float* out, const int* seeds, int blocks, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(mix_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters / 10, seeds);   // warm-up
  CK(hipDeviceSynchronize());
  double best = 1e30, sum = 0;
  const int reps = 5;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(mix_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, seeds);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::fmin(best, (double)ms);
    sum += ms;
  }
  const double flops = 2.0 * 32 * 32 * 64 * 4 /*accumulators*/ * (double)iters * blocks * 4 /*waves*/;
  printf("part B: %-10s %8.3f ms mean (best %.3f)  -> %7.1f TFLOP/s of fp32-equivalent product work (K=64 groups)\n", name,
         sum / reps, best, flops / (sum / reps * 1e-3) / 1e12);
  return sum / reps;
}

int main() {
  int rc = part_a();
  rc |= part_a2();
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int blocks = prop.multiProcessorCount * 2;   // two 4-wave blocks per CU = two waves per SIMD, like gemm_f16x3
  float* out; int* seeds;
  CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
  std::vector<int> hs(1024);
  uint32_t rng = 777u;
  for (auto& v : hs) { rng = rng * 1664525u + 1013904223u; v = (int)rng; }
  CK(hipMalloc(&seeds, 4096));
  CK(hipMemcpy(seeds, hs.data(), 4096, hipMemcpyHostToDevice));
  const int iters = 20000;
  printf("%s, %d CUs, %d blocks x 4 waves, %d iterations x 4 accumulators\n", prop.name, prop.multiProcessorCount, blocks, iters);
  for (int round = 0; round < 2; ++round) {
    const double t0 = run_mix<0>("f16x3", out, seeds, blocks, iters);
    const double t1 = run_mix<1>("f16+f6x2", out, seeds, blocks, iters);
    const double t2 = run_mix<2>("f16+f8x2", out, seeds, blocks, iters);
    run_mix<3>("f6 only", out, seeds, blocks, iters);
    printf("        speed-up over f16x3 at the matrix pipe: f16+f6x2 %.2fx, f16+f8x2 %.2fx\n", t0 / t1, t0 / t2);
  }
  return rc;
}

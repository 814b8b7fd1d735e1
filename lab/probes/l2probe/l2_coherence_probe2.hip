// Second platform probe (see l2_coherence_probe.hip): closer to the failing pattern -- TWO identical kernel chains on two
// streams, each over its own [3840][512] fp32 buffer:  tile_reader (480 workgroups; the 8 workgroups of a 64-row tile read the same
// rows, like a GEMM's A operand) -> row_writer (960 workgroups of 4 waves, one row per wave, read-modify-write in place like
// layernorm_kernel) -> tile_checker (15 .. 50 workgroups of 512 threads re-reading rows, like OutputProcess) counting rows
// that do not hold the value the writer just stored.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)
constexpr int ROWS = 3840, COLS = 512;

__global__ void tile_reader(const float* __restrict__ x, float* __restrict__ sink) {
  const int tile = blockIdx.x / 8;
  float s = 0.f;
  for (int i = threadIdx.x; i < 64 * COLS / 4; i += blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x + (size_t)tile * 64 * COLS)[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == -1234.5f) sink[blockIdx.x] = s;
}
__global__ void row_writer(float* x, float add) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  float4* p = reinterpret_cast<float4*>(x + (size_t)row * COLS);
  for (int i = lane; i < COLS / 4; i += 64) {
    float4 v = p[i];
    v.x += add; v.y += add; v.z += add; v.w += add;
    p[i] = v;
  }
}
__global__ void tile_checker(const float* x, float want, int rows_per_block, int* __restrict__ stale) {
  const int r0 = blockIdx.x * rows_per_block;
  int bad = 0;
  for (int i = threadIdx.x; i < rows_per_block * COLS / 4; i += blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x + (size_t)r0 * COLS)[i];
    bad += (v.x != want) + (v.y != want) + (v.z != want) + (v.w != want);
  }
  if (bad) atomicAdd(&stale[blockIdx.x], bad);
}

int main() {
  const int chains = 4, iters = 2000;
  float* x[chains]; float* sink[chains]; int* stale[chains]; hipStream_t st[chains];
  for (int c = 0; c < chains; ++c) {
    CK(hipMalloc(&x[c], (size_t)ROWS * COLS * 4)); CK(hipMalloc(&sink[c], 480 * 4)); CK(hipMalloc(&stale[c], 64 * 4));
    CK(hipMemset(x[c], 0, (size_t)ROWS * COLS * 4)); CK(hipMemset(stale[c], 0, 64 * 4));
    CK(hipStreamCreateWithFlags(&st[c], hipStreamNonBlocking));
  }
  for (int nch : {1, 2, 4}) {
    for (int c = 0; c < chains; ++c) { CK(hipMemset(x[c], 0, (size_t)ROWS * COLS * 4)); CK(hipMemset(stale[c], 0, 64 * 4)); }
    CK(hipDeviceSynchronize());
    for (int it = 1; it <= iters; ++it)
      for (int c = 0; c < nch; ++c) {
        hipLaunchKernelGGL(tile_reader, dim3(480), dim3(512), 0, st[c], x[c], sink[c]);
        hipLaunchKernelGGL(row_writer, dim3(ROWS / 4), dim3(256), 0, st[c], x[c], 1.0f);
        hipLaunchKernelGGL(tile_checker, dim3(60), dim3(512), 0, st[c], x[c], (float)it, 64, stale[c]);
      }
    CK(hipDeviceSynchronize());
    long total = 0;
    for (int c = 0; c < nch; ++c) {
      std::vector<int> h(64);
      CK(hipMemcpy(h.data(), stale[c], 64 * 4, hipMemcpyDeviceToHost));
      for (int v : h) total += v;
    }
    printf("%d concurrent chain(s), %d iterations each: %ld stale elements seen\n", nch, iters, total);
  }
  return 0;
}

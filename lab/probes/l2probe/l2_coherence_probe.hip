// Platform probe for the open issue of profiles/r02e_dip.md: do same-stream kernel boundaries on this MI355X make a buffer that
// one XCD rewrote visible to readers on the other XCDs -- with one stream active, and with a second stream dispatching kernels?
//   hipcc --offload-arch=gfx950 -O2 tools/l2probe/l2_coherence_probe.hip -o build/l2_coherence_probe && build/l2_coherence_probe
// Sequence per iteration on stream A:  reader (every workgroup caches x) -> writer (ONE workgroup rewrites x with a new value)
// -> [optional tiny kernel] -> checker (every workgroup counts elements that still hold the OLD value).  Stream B (optional):
// an endless supply of small kernels over its own buffer.  Prints the number of iterations in which any workgroup saw old data.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

__global__ void reader(const int* __restrict__ x, int n, int* __restrict__ sink) {
  int s = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
  if (s == 0x7fffffff) sink[blockIdx.x] = s;   // never true: keeps the loads
}
__global__ void writer(int* x, int n, int v) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) x[i] = v;
}
__global__ void checker(const int* x, int n, int v, int* __restrict__ stale) {
  int bad = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) bad += (x[i] != v);
  if (bad) atomicAdd(&stale[blockIdx.x], bad);
}
__global__ void nop(int) {}
__global__ void busy(float* y, int n) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = y[i] * 1.0001f + 0.1f;
}

int run(int iters, bool second_stream, bool extra_nop, int nblocks, int n) {
  int *x, *sink, *stale;
  float* y;
  CK(hipMalloc(&x, n * sizeof(int)));
  CK(hipMalloc(&sink, nblocks * sizeof(int)));
  CK(hipMalloc(&stale, nblocks * sizeof(int)));
  CK(hipMalloc(&y, (1 << 20) * sizeof(float)));
  CK(hipMemset(stale, 0, nblocks * sizeof(int)));
  CK(hipMemset(x, 0, n * sizeof(int)));
  CK(hipMemset(y, 0, (1 << 20) * sizeof(float)));
  hipStream_t a, b;
  CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
  std::vector<int> h(nblocks);
  int bad_iters = 0;
  for (int it = 1; it <= iters; ++it) {
    if (second_stream)
      for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(busy, dim3(240), dim3(512), 0, b, y, 1 << 20);
    hipLaunchKernelGGL(reader, dim3(nblocks), dim3(256), 0, a, x, n, sink);
    hipLaunchKernelGGL(writer, dim3(1), dim3(256), 0, a, x, n, it);
    if (extra_nop) hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, a, 0);
    hipLaunchKernelGGL(checker, dim3(nblocks), dim3(256), 0, a, x, n, it, stale);
    if (it % 64 == 0 || it == iters) {
      CK(hipStreamSynchronize(a));
      CK(hipMemcpy(h.data(), stale, nblocks * sizeof(int), hipMemcpyDeviceToHost));
      int any = 0;
      for (int v : h) any += v;
      if (any) { ++bad_iters; CK(hipMemset(stale, 0, nblocks * sizeof(int))); }
    }
  }
  CK(hipDeviceSynchronize());
  CK(hipFree(x)); CK(hipFree(sink)); CK(hipFree(stale)); CK(hipFree(y));
  CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b));
  return bad_iters;
}

int main() {
  const int iters = 4096, nblocks = 256, n = 16384;   // 64 KB buffer: stays in every XCD's L2
  printf("one stream:                      %d of %d check windows saw stale data\n", run(iters, false, false, nblocks, n), iters / 64);
  printf("two streams:                     %d of %d check windows saw stale data\n", run(iters, true, false, nblocks, n), iters / 64);
  printf("two streams + extra kernel:      %d of %d check windows saw stale data\n", run(iters, true, true, nblocks, n), iters / 64);
  return 0;
}

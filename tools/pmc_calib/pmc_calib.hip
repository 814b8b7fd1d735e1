// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 by REQUEST WIDTH: kernels that move an exactly
// known number of bytes through the access patterns of this repository's hot kernels, over buffers larger than the 256 MB Infinity
// Cache.  Run once per counter pass (tools/gpu_r5_pmc_calib.sh); tools/rocpd_pmc.py prints the counters per kernel name, this program
// prints the true bytes.  Patterns (one 1 KB row = 512 fp16 of a [rows][512] operand plane):
//   read16_stream   16 B per lane, a wave reads 1 KB contiguous          (W fragments, K / V^T tiles: 128-byte requests)
//   dma_rows64      global_load_lds_dwordx4, 16 rows x 64 B per wave      (A chunks of the GEMMs: 64 bytes per row)
//   read8_rows64    8 B per lane, 8 lanes = 64 B per row, 8 rows per wave (plane residuals)
//   write8_rows64   8 B per lane, 8 lanes = 64 B per row                  (plane stores of the epilogues)
//   write16_stream  16 B per lane contiguous                              (V^T stores, fp32 outputs)
// Build: hipcc --offload-arch=gfx950 -O3 pmc_calib.hip -o ../../build/pmc_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr size_t ROW = 1024;   // bytes per row

__global__ __launch_bounds__(256) void read16_stream(const uint4* __restrict__ src, uint4* __restrict__ sink, size_t n16) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
    const uint4 v = src[i];
    acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w;
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = acc;   // never true: keeps the loads
}

// one wave instruction = 16 rows x 64 B: lane -> (row = lane >> 2, 16-byte chunk = lane & 3); a workgroup walks the 16 64-byte
// k-blocks of its 16-row groups, i.e. every byte of the matrix is requested exactly once
__global__ __launch_bounds__(256) void dma_rows64(const unsigned char* __restrict__ src, uint4* __restrict__ sink, size_t rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t groups = rows / 16;
  for (size_t g = (size_t)blockIdx.x * 4 + w; g < groups; g += (size_t)gridDim.x * 4) {
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      const unsigned char* p = src + (g * 16 + (lane >> 2)) * ROW + kb * 64 + (lane & 3) * 16;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                       (__attribute__((address_space(3))) void*)(lds + w * 16384 + kb * 1024), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
  }
  __syncthreads();
  if (lds[threadIdx.x * 16] == 0x5a && lds[threadIdx.x * 16 + 1] == 0xa5 && lds[3] == 0x77) sink[0] = make_uint4(1, 2, 3, 4);
}

__global__ __launch_bounds__(256) void read8_rows64(const unsigned char* __restrict__ src, uint4* __restrict__ sink, size_t rows) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint2 acc = make_uint2(0, 0);
  const size_t groups = rows / 8;
  for (size_t g = (size_t)blockIdx.x * 4 + w; g < groups; g += (size_t)gridDim.x * 4) {
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
      const uint2 v = *reinterpret_cast<const uint2*>(src + (g * 8 + (lane >> 3)) * ROW + kb * 64 + (lane & 7) * 8);
      acc.x ^= v.x; acc.y ^= v.y;
    }
  }
  if (acc.x == 0x12345678u && acc.y == 0x9abcdef0u) sink[0] = make_uint4(acc.x, acc.y, 0, 0);
}

__global__ __launch_bounds__(256) void write8_rows64(unsigned char* __restrict__ dst, size_t rows) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const size_t groups = rows / 8;
  for (size_t g = (size_t)blockIdx.x * 4 + w; g < groups; g += (size_t)gridDim.x * 4) {
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
      *reinterpret_cast<uint2*>(dst + (g * 8 + (lane >> 3)) * ROW + kb * 64 + (lane & 7) * 8) = make_uint2((unsigned)g, (unsigned)kb);
  }
}

__global__ __launch_bounds__(256) void write16_stream(uint4* __restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = make_uint4((unsigned)i, 1, 2, 3);
}

int main(int argc, char** argv) {
  const size_t bytes = (argc > 1 ? strtoull(argv[1], nullptr, 10) : 1024ull) << 20;   // MiB, default 1 GiB
  const size_t rows = bytes / ROW, n16 = bytes / 16;
  unsigned char *a = nullptr, *b = nullptr;
  uint4* sink = nullptr;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&b, bytes));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(a, 1, bytes));
  CK(hipMemset(b, 2, bytes));
  CK(hipDeviceSynchronize());
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_rows64), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const int grid = 2048;
  for (int rep = 0; rep < 2; ++rep) {        // two dispatches of each: the reader takes the mean
    hipLaunchKernelGGL(read16_stream, dim3(grid), dim3(256), 0, 0, reinterpret_cast<const uint4*>(a), sink, n16);
    hipLaunchKernelGGL(dma_rows64, dim3(grid), dim3(256), 65536, 0, b, sink, rows);
    hipLaunchKernelGGL(read8_rows64, dim3(grid), dim3(256), 0, 0, a, sink, rows);
    hipLaunchKernelGGL(write8_rows64, dim3(grid), dim3(256), 0, 0, b, rows);
    hipLaunchKernelGGL(write16_stream, dim3(grid), dim3(256), 0, 0, reinterpret_cast<uint4*>(a), n16);
    CK(hipDeviceSynchronize());
  }
  CK(hipGetLastError());
  printf("{\"bytes_per_dispatch\": %zu, \"kernels\": [\"read16_stream\", \"dma_rows64\", \"read8_rows64\", \"write8_rows64\", \"write16_stream\"]}\n", bytes);
  return 0;
}

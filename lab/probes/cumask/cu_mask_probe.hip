// Which physical CUs does a hipExtStreamCreateWithCUMask mask select on this device?  For the three mask shapes of
// mdm_api.hip AuxStreams::ensure (cu_mask_mode 1 / 2 / 3, stream index 0) every workgroup of a 4096-workgroup launch records
// (XCC_ID, SE_ID, CU_ID); the host prints, per mode, the set of XCCs and the (SE, CU) pairs seen.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <map>
#include <vector>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

__global__ void where(unsigned* out) {
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // stay a little so that the launch spreads over every CU the mask allows
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(8);
}

int main() {
  const int N = 4096;
  unsigned* d; CK(hipMalloc(&d, N * 8));
  std::vector<unsigned> h(2 * N);
  for (int mode = 0; mode <= 3; ++mode) {
    for (int owner_sel = 0; owner_sel < (mode == 0 ? 1 : 2); ++owner_sel) {
      hipStream_t st;
      if (mode == 0) CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      else {
        uint32_t mask[8] = {};
        for (int b = 0; b < 256; ++b) {
          const int owner = mode == 1 ? b % 4 : mode == 2 ? (b / 8) % 4 : b / 64;
          if (owner == owner_sel) mask[b >> 5] |= 1u << (b & 31);
        }
        CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
      }
      CK(hipMemsetAsync(d, 0xff, N * 8, st));
      hipLaunchKernelGGL(where, dim3(N), dim3(64), 0, st, d);
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(h.data(), d, N * 8, hipMemcpyDeviceToHost));
      std::map<unsigned, std::set<unsigned>> per_xcc;
      for (int i = 0; i < N; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
      }
      printf("mode %d stream %d: %zu XCCs:", mode, owner_sel, per_xcc.size());
      for (auto& kv : per_xcc) printf(" xcc%u(%zu CUs)", kv.first, kv.second.size());
      printf("\n   xcc%u (se.sh.cu):", per_xcc.begin()->first);
      for (unsigned v : per_xcc.begin()->second) printf(" %u.%u.%u", v >> 8, (v >> 4) & 1, v & 0xf);
      printf("\n");
      CK(hipStreamDestroy(st));
    }
  }
  return 0;
}

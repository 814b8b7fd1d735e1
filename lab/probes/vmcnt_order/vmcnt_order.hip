// Does gfx950 retire vector-memory operations of DIFFERENT kinds -- LDS-DMA (global_load_lds) and ordinary loads that return
// to VGPRs -- in issue order on the ONE vmcnt counter?  A counted wait `s_waitcnt vmcnt(N)` in a queue that mixes both kinds
// (the pipelined k-loop of gemm_x3.h did exactly that) is only meaningful if they do.
//
// Each wave: an OLDER operation to a cold (never touched, HBM) address, a YOUNGER one to a hot (just touched) address, then
// `s_waitcnt vmcnt(1)` and an immediate check of the OLDER operation's destination (pre-set to a sentinel).
//   test 0  older = LDS-DMA (cold)   younger = VGPR load (hot)     stale LDS  => the VGPR load retired first
//   test 1  older = VGPR load (cold) younger = LDS-DMA (hot)       stale VGPR => the LDS-DMA retired first
//   test 2  older = VGPR load (cold) younger = VGPR load (hot)     control (same kind)
//   test 3  older = LDS-DMA (cold)   younger = LDS-DMA (hot)       control (same kind)
// Build: hipcc --offload-arch=gfx950 -O2 tools/vmcnt_order/vmcnt_order.hip -o build/vmcnt_order ; run: build/vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

constexpr unsigned SENT = 0xDEADBEEFu;

template <int TEST>
__global__ __launch_bounds__(64) void probe(const unsigned* __restrict__ cold, const unsigned* __restrict__ hot,
                                            unsigned long long* stale, unsigned long long* checked, int iters,
                                            size_t cold_words_per_wg) {
  __shared__ __attribute__((aligned(16))) unsigned lds[2 * 256];   // two 1 KB pieces
  const int lane = threadIdx.x;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
  unsigned long long bad = 0, n = 0;
  const unsigned* hp = hot + ((blockIdx.x & 63) * 256) + lane * 4;     // this lane's 16 hot bytes
  for (int it = 0; it < iters; ++it) {
    const size_t cw = (size_t)blockIdx.x * cold_words_per_wg + (size_t)it * 256 + lane * 4;   // a fresh 1 KB per iteration
    const unsigned* cp = cold + cw;
    // touch the hot line (and wait), preset the LDS pieces to the sentinel
    unsigned warm;
    asm volatile("global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(warm) : "v"(hp) : "memory");
    for (int j = 0; j < 8; ++j) lds[lane + 64 * j] = SENT;
    __syncthreads();
    unsigned got = SENT, other = 0;
    const unsigned rd = lds0 + lane * 16;   // first dword of this lane's 16 bytes in piece 0
    if constexpr (TEST == 0) {
      asm volatile(
          "s_mov_b32 m0, %4\n"
          "global_load_lds_dwordx4 %2, off\n"          // older: cold -> LDS piece 0
          "global_load_dword %1, %3, off\n"            // younger: hot -> VGPR
          "s_waitcnt vmcnt(1)\n"
          "ds_read_b32 %0, %5\n"
          "s_waitcnt lgkmcnt(0)\n"
          "s_waitcnt vmcnt(0)\n"
          : "=&v"(got), "=&v"(other)
          : "v"(cp), "v"(hp), "s"(lds0), "v"(rd)
          : "memory", "m0");
      if (got != cp[0]) ++bad;   // cp[0] re-read after everything landed
    } else if constexpr (TEST == 1) {
      got = SENT;
      unsigned snap;
      asm volatile(
          "s_mov_b32 m0, %4\n"
          "global_load_dword %0, %2, off\n"            // older: cold -> VGPR
          "global_load_lds_dwordx4 %3, off\n"          // younger: hot -> LDS piece 0
          "s_waitcnt vmcnt(1)\n"
          "v_mov_b32 %1, %0\n"                          // what the register holds right behind the counted wait
          "s_waitcnt vmcnt(0)\n"
          : "+&v"(got), "=&v"(snap)
          : "v"(cp), "v"(hp), "s"(lds0)
          : "memory", "m0");
      if (snap != cp[0]) ++bad;
    } else if constexpr (TEST == 2) {
      got = SENT;
      unsigned snap;
      asm volatile(
          "global_load_dword %0, %3, off\n"            // older: cold -> VGPR
          "global_load_dword %1, %4, off\n"            // younger: hot -> VGPR
          "s_waitcnt vmcnt(1)\n"
          "v_mov_b32 %2, %0\n"
          "s_waitcnt vmcnt(0)\n"
          : "+&v"(got), "=&v"(other), "=&v"(snap)
          : "v"(cp), "v"(hp)
          : "memory");
      if (snap != cp[0]) ++bad;
    } else {
      asm volatile(
          "s_mov_b32 m0, %3\n"
          "global_load_lds_dwordx4 %1, off\n"          // older: cold -> LDS piece 0
          "s_add_u32 m0, m0, 1024\n"
          "global_load_lds_dwordx4 %2, off\n"          // younger: hot -> LDS piece 1
          "s_waitcnt vmcnt(1)\n"
          "ds_read_b32 %0, %4\n"
          "s_waitcnt lgkmcnt(0)\n"
          "s_waitcnt vmcnt(0)\n"
          : "=&v"(got)
          : "v"(cp), "v"(hp), "s"(lds0), "v"(rd)
          : "memory", "m0");
      if (got != cp[0]) ++bad;
    }
    ++n;
    __syncthreads();
    (void)warm; (void)other;
  }
  atomicAdd(stale, bad);
  atomicAdd(checked, n);
}

int main() {
  const int wgs = 1024, iters = 64;
  const size_t cold_words_per_wg = (size_t)iters * 256;
  const size_t cold_words = (size_t)wgs * cold_words_per_wg;      // 64 MB per test, a fresh region each
  unsigned *cold[4], *hot;
  CHECK(hipMalloc(&hot, 64 * 1024));
  std::vector<unsigned> h(cold_words);
  for (size_t i = 0; i < cold_words; ++i) h[i] = (unsigned)(i * 2654435761u) | 1u;   // never equals the sentinel pattern by luck
  std::vector<unsigned> hh(16 * 1024, 0x11111111u);
  CHECK(hipMemcpy(hot, hh.data(), 64 * 1024, hipMemcpyHostToDevice));
  unsigned long long *cnt;
  CHECK(hipMalloc(&cnt, 64));
  for (int rep = 0; rep < 3; ++rep) {
    for (int t = 0; t < 4; ++t) {
      CHECK(hipMalloc(&cold[t], cold_words * 4));
      CHECK(hipMemcpy(cold[t], h.data(), cold_words * 4, hipMemcpyHostToDevice));
    }
    // evict caches: stream a big buffer
    unsigned* junk; CHECK(hipMalloc(&junk, 1ull << 30)); CHECK(hipMemset(junk, 1, 1ull << 30)); CHECK(hipDeviceSynchronize());
    for (int t = 0; t < 4; ++t) {
      CHECK(hipMemset(cnt, 0, 64));
      switch (t) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(64), 0, 0, cold[t], hot, cnt, cnt + 1, iters, cold_words_per_wg); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(64), 0, 0, cold[t], hot, cnt, cnt + 1, iters, cold_words_per_wg); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(64), 0, 0, cold[t], hot, cnt, cnt + 1, iters, cold_words_per_wg); break;
        default: hipLaunchKernelGGL(probe<3>, dim3(wgs), dim3(64), 0, 0, cold[t], hot, cnt, cnt + 1, iters, cold_words_per_wg); break;
      }
      CHECK(hipDeviceSynchronize());
      unsigned long long r[2];
      CHECK(hipMemcpy(r, cnt, 16, hipMemcpyDeviceToHost));
      const char* names[4] = {"older LDS-DMA(cold), younger VGPR load(hot)", "older VGPR load(cold), younger LDS-DMA(hot)",
                              "older VGPR load(cold), younger VGPR load(hot)  [control]", "older LDS-DMA(cold), younger LDS-DMA(hot)  [control]"};
      printf("rep %d test %d  %-62s: %llu stale of %llu lane-checks behind vmcnt(1)\n", rep, t, names[t], r[0], r[1]);
    }
    CHECK(hipFree(junk));
    for (int t = 0; t < 4; ++t) CHECK(hipFree(cold[t]));
  }
  return 0;
}

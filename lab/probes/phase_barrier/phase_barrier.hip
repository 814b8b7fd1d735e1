// Round 6, VERDICT r05 item 3 step 0: what does a phase boundary cost on the MI355X when it is (a) a dependent kernel launch, (b) a
// barrier among the FOUR workgroups that own a DiP sequence (one per head / column tile) with their 32 KB results handed over through
// L2 / the fabric, (c) a device-wide barrier -- measured with the hand-over checked word by word.
//
//   hipcc --offload-arch=gfx950 -O3 -o phase_barrier phase_barrier.hip && ./phase_barrier
//
// Work of a phase (the shape of one DiP decoder phase at B = 32 per GPU): 256 workgroups x 256 threads; workgroup (g, c) -- group g of
// 4 = one sequence, member c -- READS the 4 x 32 KB the group's members wrote in the previous phase (64 rows x 128 columns x two fp16
// planes each), checks every word, and WRITES its own 32 KB for this phase.  `payload = 0` variants skip the data (barrier alone).
// Variants of the group barrier: members on ONE XCD (blocks b, b+8, b+16, b+24: block b runs on XCD b % 8 as observed) or spread over
// four XCDs (consecutive blocks); plain stores + agent release / acquire fences, or write-through (sc0 sc1) stores and loads without
// fences (the two valid forms of /opt/skills/guides/MI355X_MICROARCH.md "inter-workgroup visibility").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int NWG = 256, NT = 256, GROUP = 4;
constexpr int WORDS = 32 * 1024 / 16;          // 16-byte words per workgroup per phase (2048)
constexpr int PER_T = WORDS / NT;              // 8 per thread

struct Params {
  uint4* buf[2];          // [NWG][WORDS] each
  unsigned* ctr;          // group counters [NWG / GROUP] (stride 32 words: one cache line each) | grid counter | xcc counters
  unsigned* err;
  int nphase, payload, same_xcd, wt;
};

__device__ __forceinline__ uint4 expect(int wg, int phase, int i) {
  return make_uint4((unsigned)wg * 2654435761u + (unsigned)i, (unsigned)phase, (unsigned)i ^ 0x5a5a5a5au, (unsigned)(wg + phase * 131 + i * 7));
}
__device__ __forceinline__ void group_of(int b, int same_xcd, int& g, int& c) {
  if (same_xcd) { g = (b & 7) + 8 * (b >> 5); c = (b >> 3) & 3; }      // members b, b+8, b+16, b+24 (one XCD)
  else { g = b >> 2; c = b & 3; }                                          // consecutive blocks: four XCDs
}
__device__ __forceinline__ int member(int g, int c, int same_xcd) {
  return same_xcd ? ((g & 7) + 32 * (g >> 3) + 8 * c) : (g * 4 + c);
}
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st16(uint4* p, uint4 v, int wt) {
  if (wt) { const v4u w = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(w) : "memory"); }
  else *p = v;
}
__device__ __forceinline__ uint4 ld16(const uint4* p, int wt) {
  if (wt) {
    v4u w;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(w) : "v"(p) : "memory");
    return make_uint4(w.x, w.y, w.z, w.w);
  }
  return *p;
}

__device__ __forceinline__ void phase_body(const Params& P, int b, int g, int c, int phase) {
  const int t = threadIdx.x;
  if (!P.payload) return;
  if (phase > 0) {
    unsigned bad = 0;
    const uint4* src = P.buf[(phase - 1) & 1];
    for (int m = 0; m < GROUP; ++m) {
      const int wg = member(g, m, P.same_xcd);
      uint4 v[PER_T];
      if (P.wt) {      // eight write-through-coherent loads in flight, one wait (the asm wait names every destination)
        v4u w[PER_T];
#pragma unroll
        for (int k = 0; k < PER_T; ++k)
          asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(w[k]) : "v"(src + (size_t)wg * WORDS + k * NT + t) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]) :: "memory");
#pragma unroll
        for (int k = 0; k < PER_T; ++k) v[k] = make_uint4(w[k].x, w[k].y, w[k].z, w[k].w);
      } else {
#pragma unroll
        for (int k = 0; k < PER_T; ++k) v[k] = src[(size_t)wg * WORDS + k * NT + t];
      }
#pragma unroll
      for (int k = 0; k < PER_T; ++k) {
        const uint4 e = expect(wg, phase - 1, k * NT + t);
        bad += (v[k].x != e.x) | (v[k].y != e.y) | (v[k].z != e.z) | (v[k].w != e.w);
      }
    }
    if (bad) atomicAdd(P.err, bad);
  }
  uint4* dst = P.buf[phase & 1] + (size_t)b * WORDS;
#pragma unroll
  for (int k = 0; k < PER_T; ++k) st16(dst + k * NT + t, expect(b, phase, k * NT + t), P.wt);
}

__device__ __forceinline__ void arrive(unsigned* ctr, int wt) {
  // every thread's stores are issued; make them visible at agent scope, then publish
  if (wt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // write-through stores: acknowledged = at the fabric
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!wt) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
__device__ __forceinline__ void wait_for(unsigned* ctr, unsigned target, int wt) {
  if (threadIdx.x == 0) {
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    if (!wt) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  __syncthreads();
}

// (a) one launch per phase
__global__ __launch_bounds__(NT) void k_launch(Params P, int phase) {
  int g, c; group_of(blockIdx.x, P.same_xcd, g, c);
  phase_body(P, blockIdx.x, g, c, phase);
}
// (b) persistent, barrier among the 4 members of a group
__global__ __launch_bounds__(NT) void k_group(Params P) {
  int g, c; group_of(blockIdx.x, P.same_xcd, g, c);
  unsigned* ctr = P.ctr + g * 32;
  for (int phase = 0; phase < P.nphase; ++phase) {
    phase_body(P, blockIdx.x, g, c, phase);
    arrive(ctr, P.wt);
    wait_for(ctr, (unsigned)GROUP * (phase + 1), P.wt);
  }
}
// (c) persistent, device-wide barrier: one flat counter
__global__ __launch_bounds__(NT) void k_grid(Params P) {
  int g, c; group_of(blockIdx.x, P.same_xcd, g, c);
  unsigned* ctr = P.ctr + 64 * 32;
  for (int phase = 0; phase < P.nphase; ++phase) {
    phase_body(P, blockIdx.x, g, c, phase);
    arrive(ctr, P.wt);
    wait_for(ctr, (unsigned)NWG * (phase + 1), P.wt);
  }
}
// (c') device-wide, hierarchical: per-XCD counter (blocks b % 8), its last arriver goes to the top counter and releases its XCD
__global__ __launch_bounds__(NT) void k_grid_xcd(Params P) {
  int g, c; group_of(blockIdx.x, P.same_xcd, g, c);
  const int x = blockIdx.x & 7;
  unsigned* xc = P.ctr + (65 + x) * 32, *top = P.ctr + 73 * 32, *gen = P.ctr + (74 + x) * 32;
  for (int phase = 0; phase < P.nphase; ++phase) {
    phase_body(P, blockIdx.x, g, c, phase);
    if (P.wt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (!P.wt) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
      const unsigned old = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == 32u * (phase + 1)) {                    // this XCD's last arriver
        __hip_atomic_fetch_add(top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 8u * (phase + 1)) __builtin_amdgcn_s_sleep(1);
        __hip_atomic_store(gen, (unsigned)(phase + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(phase + 1)) __builtin_amdgcn_s_sleep(1);
      }
      if (!P.wt) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}
// where do blocks run?  (the same-XCD grouping is a speed assumption, never a correctness one)
__global__ void k_census(int* xcc) {
  if (threadIdx.x == 0) { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); xcc[blockIdx.x] = (int)(v & 0xf); }
}

static double run(const char* name, Params P, int kind, hipStream_t s, int reps) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30, sum = 0;
  unsigned herr = 0;
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipMemsetAsync(P.ctr, 0, 128 * 32 * sizeof(unsigned), s));
    CK(hipMemsetAsync(P.err, 0, sizeof(unsigned), s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    if (kind == 0) for (int p = 0; p < P.nphase; ++p) hipLaunchKernelGGL(k_launch, dim3(NWG), dim3(NT), 0, s, P, p);
    else if (kind == 1) hipLaunchKernelGGL(k_group, dim3(NWG), dim3(NT), 0, s, P);
    else if (kind == 2) hipLaunchKernelGGL(k_grid, dim3(NWG), dim3(NT), 0, s, P);
    else hipLaunchKernelGGL(k_grid_xcd, dim3(NWG), dim3(NT), 0, s, P);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(&herr, P.err, sizeof(unsigned), hipMemcpyDeviceToHost));
    if (r == 0) continue;                       // warm-up
    const double us = ms * 1e3 / P.nphase;
    best = us < best ? us : best; sum += us;
  }
  printf("{\"variant\": \"%s\", \"payload\": %d, \"same_xcd\": %d, \"write_through\": %d, \"phases\": %d, \"us_per_phase_min\": %.3f, \"us_per_phase_mean\": %.3f, \"bad_words\": %u}\n",
         name, P.payload, P.same_xcd, P.wt, P.nphase, best, sum / reps, herr);
  fflush(stdout);
  return best;
}

int main(int argc, char** argv) {
  const int nphase = argc > 1 ? atoi(argv[1]) : 2000, reps = 5;
  hipStream_t s; CK(hipStreamCreate(&s));
  Params P{};
  CK(hipMalloc(&P.buf[0], (size_t)NWG * WORDS * 16)); CK(hipMalloc(&P.buf[1], (size_t)NWG * WORDS * 16));
  CK(hipMalloc(&P.ctr, 128 * 32 * sizeof(unsigned))); CK(hipMalloc(&P.err, sizeof(unsigned)));
  CK(hipMemset(P.buf[0], 0, (size_t)NWG * WORDS * 16)); CK(hipMemset(P.buf[1], 0, (size_t)NWG * WORDS * 16));
  P.nphase = nphase;
  int* xcc; CK(hipMalloc(&xcc, NWG * sizeof(int)));
  hipLaunchKernelGGL(k_census, dim3(NWG), dim3(64), 0, s, xcc);
  std::vector<int> hx(NWG); CK(hipMemcpy(hx.data(), xcc, NWG * sizeof(int), hipMemcpyDeviceToHost));
  int agree = 0; for (int b = 0; b < NWG; ++b) agree += hx[b] == (b & 7);
  printf("{\"census\": \"blocks whose XCC_ID == blockIdx %% 8\", \"agree\": %d, \"of\": %d}\n", agree, NWG);
  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_group, NT, 0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("{\"device\": \"%s\", \"cus\": %d, \"blocks_per_cu_k_group\": %d}\n", prop.gcnArchName, prop.multiProcessorCount, occ);
  if (prop.multiProcessorCount * occ < NWG) { fprintf(stderr, "grid would not be co-resident\n"); return 2; }
  for (int payload = 0; payload < 2; ++payload) {
    P.payload = payload;
    for (int wt = 0; wt < (payload ? 2 : 1); ++wt) {
      P.wt = wt;
      P.same_xcd = 1; run("launch per phase", P, 0, s, reps);
      P.same_xcd = 1; run("group-of-4 barrier, one XCD", P, 1, s, reps);
      P.same_xcd = 0; run("group-of-4 barrier, four XCDs", P, 1, s, reps);
      P.same_xcd = 1; run("device barrier, flat counter", P, 2, s, reps);
      P.same_xcd = 1; run("device barrier, per-XCD hierarchy", P, 3, s, reps);
    }
  }
  return 0;
}

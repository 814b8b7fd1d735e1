// TEST INFRASTRUCTURE ONLY: a lock-step host emulator for the HIP kernels in
// motion-diffusion-model_amd/csrc.  There is no GPU in the build container, so the CPU test-suite
// compiles the *same* kernel sources with -DMDM_EMU against this header and executes every thread
// of a workgroup as a ucontext fiber on one OS thread:
//   * __syncthreads()             -> workgroup barrier between fibers
//   * wave collectives (shuffles, MFMA) -> 64-fiber exchange through a per-wave buffer
//   * MFMA lane<->element maps    -> exactly the gfx950 layouts (cdna_hip_programming.md section 3):
//       v_mfma_f32_32x32x2_f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31],
//                                D[reg]: col=l&31, row=(reg&3)+8*(reg>>2)+4*(l>>5); k-ordered fmaf chain
//       v_mfma_f32_32x32x16_bf16: A/B 8 bf16 per lane, k = 8*(l>>5)+e ; same D layout, fp32 accumulate
//       v_mfma_f32_16x16x32_bf16, v_permlane32_swap / v_permlane16_swap (the GEMM's 16-row last sub-tile)
// It validates index math / masking / epilogues, NOT timing, and is never part of the product path.
#pragma once
#include <ucontext.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
typedef void* hipStream_t;
using std::min;
using std::max;

namespace emu {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short p16x8 __attribute__((ext_vector_type(8)));

// Context switches: ucontext's swapcontext() makes two sigprocmask system calls per switch, and a lock-step emulation of a
// 512-thread workgroup switches at every barrier / wave collective -- a third of the suite's time was kernel time.  On x86-64
// the fibers switch through a six-register stack swap instead; other hosts keep ucontext.
#if defined(__x86_64__)
#define MDM_EMU_FASTCTX 1
struct LightCtx { void* sp = nullptr; };
extern "C" __attribute__((naked, noinline)) inline void mdm_emu_switch(LightCtx* /*from: rdi*/, LightCtx* /*to: rsi*/) {
  asm volatile(
      "pushq %rbp\n\tpushq %rbx\n\tpushq %r12\n\tpushq %r13\n\tpushq %r14\n\tpushq %r15\n\t"
      "movq %rsp, (%rdi)\n\t"
      "movq (%rsi), %rsp\n\t"
      "popq %r15\n\tpopq %r14\n\tpopq %r13\n\tpopq %r12\n\tpopq %rbx\n\tpopq %rbp\n\t"
      "ret");
}
#endif

struct Fiber {
#ifdef MDM_EMU_FASTCTX
  LightCtx ctx;
#else
  ucontext_t ctx;
#endif
  char* stack = nullptr;
  bool done = false;
  dim3 tid;
  // LATE mode (MDM_EMU_LATE=1): asynchronous operations of this lane that have been ISSUED but whose effect is withheld
  // until a counted wait covers them -- vector-memory queue (LDS-DMA pieces, untracked global loads) and LDS-read queue
  struct Pending { void* dst; const void* src; int bytes; };
  std::vector<Pending> vmq, ldsq;
};

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct WaveBuf {
  float fa[64], fb[64];
  p16x8 ha[64], hb[64];
  uint32_t ua[64], ub[64];
  i32x8 qa[64], qb[64];      // scaled-MFMA operands
  int sa[64], sb[64];        // and their scale registers
  int arrive = 0;
  unsigned gen = 0;
};

struct Block {
  std::vector<Fiber> fibers;
  std::vector<WaveBuf> waves;
  int nthreads = 0;
  int bar_arrive = 0;
  unsigned bar_gen = 0;
  dim3 bid, bdim, gdim;
  char* dyn = nullptr;
#ifdef MDM_EMU_FASTCTX
  LightCtx sched;
#else
  ucontext_t sched;
#endif
  int cur = 0;
  const std::function<void()>* body = nullptr;
};

inline Block*& g_block() { static Block* b = nullptr; return b; }
inline Block& blk() { return *g_block(); }
inline Fiber& cur() { return blk().fibers[blk().cur]; }
inline int lane_id() { return blk().cur & 63; }
inline int wave_id() { return blk().cur >> 6; }

#ifdef MDM_EMU_FASTCTX
inline void yield() { Block& b = blk(); mdm_emu_switch(&b.fibers[b.cur].ctx, &b.sched); }
#else
inline void yield() { Block& b = blk(); swapcontext(&b.fibers[b.cur].ctx, &b.sched); }
#endif

inline void block_barrier() {
  Block& b = blk();
  unsigned g = b.bar_gen;
  if (++b.bar_arrive == b.nthreads) { b.bar_arrive = 0; ++b.bar_gen; return; }
  while (b.bar_gen == g) yield();
}
inline void wave_barrier() {
  Block& b = blk();
  WaveBuf& w = b.waves[wave_id()];
  int wsize = std::min(64, b.nthreads - wave_id() * 64);
  unsigned g = w.gen;
  if (++w.arrive == wsize) { w.arrive = 0; ++w.gen; return; }
  while (w.gen == g) yield();
}

inline float shfl_f32(float v, int src_lane) {
  WaveBuf& w = blk().waves[wave_id()];
  w.fa[lane_id()] = v;
  wave_barrier();
  float r = w.fa[src_lane & 63];
  wave_barrier();
  return r;
}

inline f32x16 mfma_f32_32x32x2(float a, float b, f32x16 c) {
  WaveBuf& w = blk().waves[wave_id()];
  int l = lane_id();
  w.fa[l] = a;
  w.fb[l] = b;
  wave_barrier();
  int j = l & 31, h = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * h;
    float acc = c[r];
    acc = fmaf(w.fa[i], w.fb[j], acc);            // k = 0
    acc = fmaf(w.fa[i + 32], w.fb[j + 32], acc);  // k = 1
    c[r] = acc;
  }
  wave_barrier();
  return c;
}

inline float bf16_to_f32(short s) {
  uint32_t u = ((uint32_t)(uint16_t)s) << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

inline f32x16 mfma_f32_32x32x16_bf16(p16x8 a, p16x8 b, f32x16 c) {
  WaveBuf& w = blk().waves[wave_id()];
  int l = lane_id();
  w.ha[l] = a;
  w.hb[l] = b;
  wave_barrier();
  int j = l & 31, h = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * h;
    float acc = c[r];
    for (int g = 0; g < 2; ++g)
      for (int e = 0; e < 8; ++e)
        acc = fmaf(bf16_to_f32(w.ha[i + 32 * g][e]), bf16_to_f32(w.hb[j + 32 * g][e]), acc);
    c[r] = acc;
  }
  wave_barrier();
  return c;
}

// v_mfma_f32_32x32x16_f16: the bf16 form's layout with fp16 elements
inline f32x16 mfma_f32_32x32x16_f16(f16x8 a, f16x8 b, f32x16 c) {
  WaveBuf& w = blk().waves[wave_id()];
  int l = lane_id();
  w.ha[l] = __builtin_bit_cast(p16x8, a);
  w.hb[l] = __builtin_bit_cast(p16x8, b);
  wave_barrier();
  int j = l & 31, h = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * h;
    float acc = c[r];
    for (int g = 0; g < 2; ++g) {
      const f16x8 fa = __builtin_bit_cast(f16x8, w.ha[i + 32 * g]), fb = __builtin_bit_cast(f16x8, w.hb[j + 32 * g]);
      for (int e = 0; e < 8; ++e) acc = fmaf((float)fa[e], (float)fb[e], acc);
    }
    c[r] = acc;
  }
  wave_barrier();
  return c;
}

// v_mfma_scale_f32_32x32x64_f8f6f4, both operands FP6 E2M3: lane l holds 32 six-bit codes (element j at bit 6j) of k-block
// l>>5 of row / column l&31 and the block's E8M0 scale in byte 0 of its scale register (semantics checked on the MI355X:
// tools/mx/mx_probe.hip part A)
inline float fp6_e2m3_value(const i32x8& v, int j) {
  const int bit = 6 * j, w = bit >> 5, o = bit & 31;
  uint32_t code = ((uint32_t)v[w]) >> o;
  if (o > 26) code |= ((uint32_t)v[w + 1]) << (32 - o);
  code &= 63u;
  const uint32_t s = code >> 5, e = (code >> 3) & 3u, m = code & 7u;
  const float x = (e == 0) ? m * 0.125f : ldexpf(1.0f + m * 0.125f, (int)e - 1);
  return s ? -x : x;
}
inline f32x16 mfma_scale_f32_32x32x64_fp6(i32x8 a, i32x8 b, f32x16 c, int scale_a, int scale_b) {
  WaveBuf& w = blk().waves[wave_id()];
  int l = lane_id();
  w.qa[l] = a;
  w.qb[l] = b;
  w.sa[l] = scale_a;
  w.sb[l] = scale_b;
  wave_barrier();
  int j = l & 31, h = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int i = (r & 3) + 8 * (r >> 2) + 4 * h;
    float acc = c[r];
    for (int g = 0; g < 2; ++g) {
      const float s = ldexpf(1.0f, ((w.sa[i + 32 * g] & 255) - 127) + ((w.sb[j + 32 * g] & 255) - 127));
      float part = 0.f;
      for (int e = 0; e < 32; ++e) part = fmaf(fp6_e2m3_value(w.qa[i + 32 * g], e), fp6_e2m3_value(w.qb[j + 32 * g], e), part);
      acc = fmaf(part, s, acc);
    }
    c[r] = acc;
  }
  wave_barrier();
  return c;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_16x16x32_bf16: A row / B column = l&15, k = 8*(l>>4)+e; D[reg]: row 4*(l>>4)+reg, column l&15
inline f32x4 mfma_f32_16x16x32_bf16(p16x8 a, p16x8 b, f32x4 c) {
  WaveBuf& w = blk().waves[wave_id()];
  int l = lane_id();
  w.ha[l] = a;
  w.hb[l] = b;
  wave_barrier();
  int j = l & 15, g = l >> 4;
  for (int r = 0; r < 4; ++r) {
    int i = 4 * g + r;
    float acc = c[r];
    for (int kg = 0; kg < 4; ++kg)
      for (int e = 0; e < 8; ++e)
        acc = fmaf(bf16_to_f32(w.ha[i + 16 * kg][e]), bf16_to_f32(w.hb[j + 16 * kg][e]), acc);
    c[r] = acc;
  }
  wave_barrier();
  return c;
}

// v_mfma_f32_16x16x32_f16: the bf16 form's layout with fp16 elements
inline f32x4 mfma_f32_16x16x32_f16(f16x8 a, f16x8 b, f32x4 c) {
  WaveBuf& w = blk().waves[wave_id()];
  int l = lane_id();
  w.ha[l] = __builtin_bit_cast(p16x8, a);
  w.hb[l] = __builtin_bit_cast(p16x8, b);
  wave_barrier();
  int j = l & 15, g = l >> 4;
  for (int r = 0; r < 4; ++r) {
    int i = 4 * g + r;
    float acc = c[r];
    for (int kg = 0; kg < 4; ++kg) {
      const f16x8 fa = __builtin_bit_cast(f16x8, w.ha[i + 16 * kg]), fb = __builtin_bit_cast(f16x8, w.hb[j + 16 * kg]);
      for (int e = 0; e < 8; ++e) acc = fmaf((float)fa[e], (float)fb[e], acc);
    }
    c[r] = acc;
  }
  wave_barrier();
  return c;
}

// v_permlane32_swap_b32: lanes 32-63 of vdst <-> lanes 0-31 of src
inline void permlane32_swap(uint32_t& vdst, uint32_t& src) {
  WaveBuf& w = blk().waves[wave_id()];
  int l = lane_id();
  w.ua[l] = vdst;
  w.ub[l] = src;
  wave_barrier();
  uint32_t nv = (l < 32) ? w.ua[l] : w.ub[l - 32];
  uint32_t ns = (l < 32) ? w.ua[l + 32] : w.ub[l];
  wave_barrier();
  vdst = nv;
  src = ns;
}
// v_permlane16_swap_b32: odd 16-lane rows of vdst <-> even rows of src
inline void permlane16_swap(uint32_t& vdst, uint32_t& src) {
  WaveBuf& w = blk().waves[wave_id()];
  int l = lane_id();
  w.ua[l] = vdst;
  w.ub[l] = src;
  wave_barrier();
  bool odd = (l >> 4) & 1;
  uint32_t nv = odd ? w.ub[l - 16] : w.ua[l];
  uint32_t ns = odd ? w.ub[l] : w.ua[l + 16];
  wave_barrier();
  vdst = nv;
  src = ns;
}

// ---- LATE mode: the adversarial half of the asynchronous-operation model.  By default (EARLY) an LDS-DMA / untracked load /
// untracked ds_read takes effect the moment it is issued -- the earliest the hardware could deliver it, which exposes
// write-after-read hazards (a stage refilled while somebody still reads it).  With MDM_EMU_LATE=1 the effect is withheld
// until the issuing lane executes a counted wait that covers the operation (`s_waitcnt vmcnt(N)` / `lgkmcnt(N)`: everything
// but the N youngest operations of that queue) -- the latest the hardware may deliver it, which exposes read-after-write
// hazards (data consumed before the wait + barrier that makes it valid) and wrong wait counts.  Registers awaiting a
// withheld load hold a NaN pattern.  Operations the hardware also counts but the emulator executes synchronously (plain
// loads, stores, compiler-tracked LDS accesses) only make the hardware's waits STRICTER than the model's.
inline bool late_mode() {
  static const bool on = [] { const char* e = getenv("MDM_EMU_LATE"); return e != nullptr && e[0] == '1'; }();
  return on;
}
inline void apply_pending(std::vector<Fiber::Pending>& q, size_t keep) {
  if (q.size() <= keep) return;
  const size_t n = q.size() - keep;
  for (size_t i = 0; i < n; ++i) memcpy(q[i].dst, q[i].src, (size_t)q[i].bytes);
  q.erase(q.begin(), q.begin() + (long)n);
}
inline void vm_issue(void* dst, const void* src, int bytes, bool poison_dst) {
  if (!late_mode()) { memcpy(dst, src, (size_t)bytes); return; }
  if (poison_dst) memset(dst, 0xFF, (size_t)bytes);
  cur().vmq.push_back(Fiber::Pending{dst, src, bytes});
}
inline void vm_wait(int n) { if (late_mode()) apply_pending(cur().vmq, (size_t)n); }
inline void lds_issue(void* dst, const void* src, int bytes) {
  if (!late_mode()) { memcpy(dst, src, (size_t)bytes); return; }
  memset(dst, 0xFF, (size_t)bytes);
  cur().ldsq.push_back(Fiber::Pending{dst, src, bytes});
}
inline void lgkm_wait(int n) { if (late_mode()) apply_pending(cur().ldsq, (size_t)n); }

inline void fiber_entry() {
  Block& b = blk();
  (*b.body)();
  apply_pending(b.fibers[b.cur].vmq, 0);   // s_endpgm: whatever is still in flight lands (LATE mode)
  b.fibers[b.cur].ldsq.clear();            // a register load nobody waited for has no observer
  b.fibers[b.cur].done = true;
#ifdef MDM_EMU_FASTCTX
  mdm_emu_switch(&b.fibers[b.cur].ctx, &b.sched);
  __builtin_trap();      // a finished fiber is never resumed
#else
  swapcontext(&b.fibers[b.cur].ctx, &b.sched);
#endif
}

inline void run_block(Block& b) {
  g_block() = &b;
  const size_t STK = 256 * 1024;
  for (int t = 0; t < b.nthreads; ++t) {
    Fiber& f = b.fibers[t];
    f.done = false;
    if (!f.stack) f.stack = (char*)malloc(STK);
#ifdef MDM_EMU_FASTCTX
    // initial frame: six zeroed callee-saved registers, then fiber_entry as the return address, placed so that the stack is
    // 16-byte aligned + 8 at fiber_entry's first instruction (as after a call)
    uintptr_t top = ((uintptr_t)f.stack + STK) & ~(uintptr_t)15;
    void** sp = (void**)(top - 16);
    *sp = (void*)(void (*)())fiber_entry;
    for (int i = 0; i < 6; ++i) *--sp = nullptr;
    f.ctx.sp = sp;
#else
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack;
    f.ctx.uc_stack.ss_size = STK;
    f.ctx.uc_link = &b.sched;
    makecontext(&f.ctx, (void (*)())fiber_entry, 0);
#endif
  }
  int remaining = b.nthreads;
  while (remaining > 0) {
    for (int t = 0; t < b.nthreads; ++t) {
      if (b.fibers[t].done) continue;
      b.cur = t;
#ifdef MDM_EMU_FASTCTX
      mdm_emu_switch(&b.sched, &b.fibers[t].ctx);
#else
      swapcontext(&b.sched, &b.fibers[t].ctx);
#endif
      if (b.fibers[t].done) --remaining;
    }
  }
}

// Launch: every block of the grid runs sequentially; `max_blocks` (env MDM_EMU_MAX_BLOCKS) lets a
// test execute only a sample of the blocks of a big grid.
inline void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
  Block b;
  b.nthreads = block.x * block.y * block.z;
  b.fibers.resize(b.nthreads);
  b.waves.resize((b.nthreads + 63) / 64);
  b.bdim = block;
  b.gdim = grid;
  b.body = &body;
  std::vector<char> dyn(shmem + 64);
  b.dyn = (char*)(((uintptr_t)dyn.data() + 15) & ~(uintptr_t)15);
  for (int t = 0; t < b.nthreads; ++t) {
    b.fibers[t].tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
  }
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        b.bid = dim3(x, y, z);
        b.bar_arrive = 0;
        for (auto& w : b.waves) w.arrive = 0;
        run_block(b);
      }
  for (auto& f : b.fibers) free(f.stack);
  g_block() = nullptr;
}

}  // namespace emu

#define threadIdx (emu::cur().tid)
#define blockIdx (emu::blk().bid)
#define blockDim (emu::blk().bdim)
#define gridDim (emu::blk().gdim)
#define __syncthreads() emu::block_barrier()
#define MDM_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(emu::blk().dyn)
#define MDM_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  emu::launch((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })

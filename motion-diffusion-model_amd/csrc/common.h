// Shared device helpers for the MDM hot-path kernels (gfx950 / CDNA4 only).
//
// The only non-HIP build of these sources is the CPU test emulator (tests/emu, -DMDM_EMU), which
// exists because the build container has no GPU; it is test infrastructure, never a product path.
#pragma once
#ifdef MDM_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define MDM_DYN_SMEM(type, name) \
  extern __shared__ __attribute__((aligned(16))) unsigned char mdm_dyn_smem_raw[]; \
  type* name = reinterpret_cast<type*>(mdm_dyn_smem_raw)
#define MDM_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__)
#endif
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace mdm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short p16x8 __attribute__((ext_vector_type(8)));

constexpr int kWave = 64;

// Ordinal of the calling thread's current HIP device: host-side caches of per-device facts (CU count, function attributes)
// are indexed by it, so that one process may drive several GPUs (one model handle per device).
constexpr int kMaxDevices = 64;
inline int rt_device_ordinal() {
#ifdef MDM_EMU
  return 0;
#else
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
#endif
}

#ifndef MDM_EMU
// The one-time opt-in of a kernel instantiation to more than 64 KB of dynamic LDS, per device (`configured` = that instantiation's
// static bool[kMaxDevices]).  0: set (now or earlier); -1: the HIP call failed; -3: this would be the instantiation's FIRST use and
// `stream` is being captured into a hipGraph -- a function-attribute call does not belong inside a capture (a warm-up
// with another token count / window shape leaves an instantiation unconfigured): the caller reports MDM_EUNSUPPORTED and asks for a
// warm-up call of the same shapes outside the capture (include/mdm_hip.h "hipGraph CAPTURE").  The capture query runs on a first
// use only: a configured instantiation costs one array read per launch.
template <class KF> inline int rt_dyn_lds_once(KF kfn, int bytes, bool* configured, hipStream_t stream) {
  bool& done = configured[rt_device_ordinal()];
  if (done) return 0;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return -3;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return -1;
  done = true;
  return 0;
}
#endif

// MDM_EARLY_KERNARGS (default 1; -DMDM_EARLY_KERNARGS=0 is the A/B build): pin the scalar loads of the kernel arguments a kernel's FIRST
// vector loads depend on into its entry block.  hipcc sinks part of them behind the first s_waitcnt + the (branchy) tile arithmetic --
// gemm_x3s_kernel: sizes first, wait, tile index, THEN the operand pointers; selfattn_block_kernel: the plane pointer re-read from the
// kernarg segment with a run-time offset at its point of use -- i.e. two dependent scalar-memory round trips in front of the first
// LDS-DMA request of kernels that last 10-30 us.  An empty asm statement that names the values as inputs makes them live at the top, so
// all s_loads go out in one batch.  Used in the two kernels of the latency regime only (round 6, profiles/r06e_early_kernargs.md: DiP
// +1.0 % same box, per 40-frame call -1.4 %); the persistent kernels of the headline pay their prologue once per 150-250 us and are left alone.
#ifndef MDM_EARLY_KERNARGS
#define MDM_EARLY_KERNARGS 1
#endif
#if MDM_EARLY_KERNARGS && !defined(MDM_EMU)
#define MDM_KERNARGS_NOW(...) asm volatile("" ::__VA_ARGS__)
// a kernel-argument pointer as an opaque SGPR value: `(p ? a.lo : a.hi)` on the raw arguments makes hipcc index the kernarg SEGMENT
// with a run-time offset (s_load ... s46) -- another dependent scalar-memory round trip at the point of use
template <class T> __device__ __forceinline__ T* rt_sgpr_ptr(T* p) { asm volatile("" : "+s"(p)); return p; }
#else
#define MDM_KERNARGS_NOW(...) do { } while (0)
template <class T> __device__ __forceinline__ T* rt_sgpr_ptr(T* p) { return p; }
#endif

// v_mfma_f32_32x32x2_f32: exact-fp32 matrix FMA (64 cyc/SIMD, 157 TF chip peak).
//   lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
//   D[reg] is row i = (reg&3) + 8*(reg>>2) + 4*(l>>5), column j = l&31.
__device__ __forceinline__ f32x16 mfma_f32(float a, float b, f32x16 c) {
#ifdef MDM_EMU
  return emu::mfma_f32_32x32x2(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
#endif
}

// ---------------------------------------------------------------------------------------------
// The 16-bit element of the split-precision ("x3") operand planes.  DEFAULT: IEEE fp16 ("f16x3": 11 + 11 significant bits per
// fp32 value).  -DMDM_SPLIT_BF16 builds the round-1 bfloat16 form ("f16x3": 8 + 8 bits, fp32's exponent range) for A/B
// runs.  Same bytes, same MFMA rate (v_mfma_f32_32x32x16_f16 / _bf16), same kernels: only these helpers differ.
// Why fp16: on "trained-like" weights (oracle/synth.py synth_state_dict_hostile) the bf16 split is 10x the fp32 reference's own
// rounding noise, the fp16 split sits AT that noise (tools/precision_probe.py, tools/fold_probe.py).
// Range: |x| <= 65504; beyond it the split yields inf and the sample NaN (a model whose activations leave fp16's range
// needs the f32 mode -- the sampler seam checks the result and says so); below 2^-14 the planes keep an absolute
// precision of 2^-25 (fp16 subnormals -- gfx950's MFMA does not flush them).
// ---------------------------------------------------------------------------------------------
#ifdef MDM_SPLIT_BF16
constexpr bool kSplitF16 = false;
#else
constexpr bool kSplitF16 = true;
#endif
typedef _Float16 f16_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// WEIGHTS are split as hi / lo of (w * 2^8) and the accumulators scaled back by 2^-8 (exact): nn.Linear weights are O(1/sqrt(K))
// ~ 0.03, whose lo part (~2^-16) would be an fp16 SUBNORMAL with only ~8 significant bits -- the weight would carry 2^-20
// instead of 2^-23.  Scaled, lo is a normal number for every |w| >= 2^-11; |w| < 255 keeps hi finite.  (bf16 build: 1.)
constexpr float kX3WeightScale = kSplitF16 ? 256.f : 1.f;
constexpr float kX3AccScale = kSplitF16 ? 1.f / 256.f : 1.f;

// v_mfma_f32_32x32x16_{f16,bf16}: lane l supplies 8 consecutive-k elements of row/col (l&31), k-block (l>>5).
__device__ __forceinline__ f32x16 mfma_p16(p16x8 a, p16x8 b, f32x16 c) {
#ifdef MDM_EMU
  if constexpr (kSplitF16) return emu::mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c);
  else return emu::mfma_f32_32x32x16_bf16(a, b, c);
#else
  if constexpr (kSplitF16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#endif
}

// ---- the instructions of the "f16f6" seed GEMM (gemm_f16f6.h)
typedef int i32x8 __attribute__((ext_vector_type(8)));
// v_mfma_f32_32x32x16_f16: operand / result layout of mfma_p16 above, fp16 elements.
__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
#ifdef MDM_EMU
  return emu::mfma_f32_32x32x16_f16(a, b, c);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
// v_mfma_scale_f32_32x32x64_f8f6f4 with both operands FP6 E2M3 (cbsz = blgp = 2): lane l supplies the 32 consecutive k of
// k-block (l>>5) of row / column (l&31) as 32 six-bit codes (element j at bit 6j of dwords 0-5; dwords 6-7 unused) and that
// block's E8M0 scale (2^(byte - 127)) in byte 0 of scale_a / scale_b; D layout as above.  Semantics verified on the MI355X
// against a host reference (tools/mx/mx_probe.hip part A, profiles/r01h_mx_probe.txt).
__device__ __forceinline__ f32x16 mfma_mx_fp6(i32x8 a, i32x8 b, f32x16 c, int scale_a, int scale_b) {
#ifdef MDM_EMU
  return emu::mfma_scale_f32_32x32x64_fp6(a, b, c, scale_a, scale_b);
#else
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, scale_a, 0, scale_b);
#endif
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x32_{f16,bf16}: lane l supplies 8 consecutive-k elements of row/col (l&15), k-block (l>>4) (k = 8*(l>>4)+e);
// D[reg] is row 4*(l>>4) + reg, column l&15.
__device__ __forceinline__ f32x4 mfma16_p16(p16x8 a, p16x8 b, f32x4 c) {
#ifdef MDM_EMU
  if constexpr (kSplitF16) return emu::mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c);
  else return emu::mfma_f32_16x16x32_bf16(a, b, c);
#else
  if constexpr (kSplitF16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#endif
}

// Two v_mfma_f32_32x32x16_bf16 operand fragments of the same 32 rows -- w0: k 0-15, w1: k 16-31 -- become the two
// v_mfma_f32_16x16x32_bf16 fragments over the same 32 k -- w0: rows 0-15, w1: rows 16-31.  As rows of 16 lanes,
//   in   w0 = [r0-15 k0-7 | r16-31 k0-7 | r0-15 k8-15 | r16-31 k8-15]    w1 = the same with k + 16
//   v_permlane32_swap (lanes 32-63 of w0 <-> lanes 0-31 of w1), then v_permlane16_swap (odd rows of w0 <-> even rows of w1)
//   out  w0 = [r0-15 k0-7 | r0-15 k8-15 | r0-15 k16-23 | r0-15 k24-31]   w1 = the same for r16-31
__device__ __forceinline__ void frag32_to_frag16(p16x8& w0, p16x8& w1) {
  u32x4 a = __builtin_bit_cast(u32x4, w0), b = __builtin_bit_cast(u32x4, w1);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t x = a[i], y = b[i];
#ifdef MDM_EMU
    emu::permlane32_swap(x, y);
    emu::permlane16_swap(x, y);
#else
    auto s32 = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    auto s16 = __builtin_amdgcn_permlane16_swap(s32[0], s32[1], false, false);
    x = s16[0];
    y = s16[1];
#endif
    a[i] = x;
    b[i] = y;
  }
  w0 = __builtin_bit_cast(p16x8, a);
  w1 = __builtin_bit_cast(p16x8, b);
}

// value of `v` in lane `src_lane` (wave-uniform control flow required)
__device__ __forceinline__ float lane_bcast(float v, int src_lane) {
#ifdef MDM_EMU
  return emu::shfl_f32(v, src_lane);
#else
  return __shfl(v, src_lane, 64);
#endif
}

// row of accumulator register `reg` inside a 32x32 MFMA tile, for lane-half h = lane>>5
__device__ __forceinline__ int mfma_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

__device__ __forceinline__ float shfl_xor_f32(float v, int mask) {
#ifdef MDM_EMU
  return emu::shfl_f32(v, emu::lane_id() ^ mask);
#else
  return __shfl_xor(v, mask, 64);
#endif
}

// Sum over the 8 consecutive lanes that share lane>>3 (every lane gets the total): three DPP moves (quad_perm xor 1,
// quad_perm xor 2, row_half_mirror) at VALU speed; __shfl_xor lowers to ds_bpermute_b32, an LDS round trip per step.
__device__ __forceinline__ float sum_lanes8(float v) {
#ifdef MDM_EMU
  v += shfl_xor_f32(v, 1);
  v += shfl_xor_f32(v, 2);
  v += shfl_xor_f32(v, 4);
  return v;
#else
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
  return v;
#endif
}

// ---- key-padding masks (model/mdm.py:241-247).  A sample's frame mask reaches the attention kernels through the `lengths` array:
// lengths[b] >= 0 is a valid-FRAME COUNT (a prefix mask: every mask data_loaders/tensors.py builds); lengths[b] == -1 says
// that the sample's mask is an arbitrary BITMAP in the eight words lengths[B + 8 b .. B + 8 b + 7] (bit j of word i: frame
// 32 i + j is valid), the array then holding 9 B ints (include/mdm_hip.h).  Keys in front of the frames (`lead`: the
// condition token of trans_enc) are always valid.  key = 32 * KT + row; KT is a compile-time key-tile index.
template <int KT>
__device__ __forceinline__ bool key_valid_bits(const uint32_t* __restrict__ w, int row, int lead) {
  const int f = row - lead;                        // frame index inside key tile KT (may be -1: the last frame of tile KT - 1)
  uint32_t word;
  if (f >= 0) word = w[KT];
  else word = (KT > 0) ? w[KT > 0 ? KT - 1 : 0] : 0xffffffffu;   // KT == 0: key < lead, always valid
  return (word >> (f & 31)) & 1u;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// ---------------------------------------------------------------------------------------------
// Split-precision helpers: x = hi + lo, hi = rne16(x), lo = rne16(x - hi)  (fp16: +O(2^-22 |x|); bf16 build: +O(2^-17 |x|)).
// A fp32 product a*w is then carried by three 16-bit MFMA products  ah*wh + ah*wl + al*wh  (the al*wl term is dropped),
// accumulated in fp32: SURVEY.md section 7 "Precision vs. peak".
// ---------------------------------------------------------------------------------------------
typedef unsigned short p16_t;

__host__ __device__ __forceinline__ float p16_to_f32(p16_t b) {
  if constexpr (kSplitF16) {
    return (float)__builtin_bit_cast(f16_t, b);
  } else {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
  }
}

__host__ __device__ __forceinline__ p16_t f32_to_p16(float x) {
  if constexpr (kSplitF16) {
    // round-to-nearest-even, NOT saturating: a value beyond +-65504 becomes inf and surfaces as NaN in the sample -- loud,
    // and caught by the sampler seam (gaussian_diffusion.py _check_finite) -- instead of being clamped into a plausible
    // wrong number.  (A v_med3 clamp per value also cost 1.3 % of the whole loop: profiles/r02_ab.md.)
    return __builtin_bit_cast(p16_t, (f16_t)x);
  } else {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(unsigned short, (__bf16)x);   // v_cvt_pk_bf16_f32 (round-to-nearest-even)
#else
    uint32_t u;
    __builtin_memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (p16_t)(u >> 16);
#endif
  }
}

__host__ __device__ __forceinline__ void split_p16(float x, p16_t& hi, p16_t& lo) {
  hi = f32_to_p16(x);
  lo = f32_to_p16(x - p16_to_f32(hi));
}

// Two values -> packed (hi, hi) and (lo, lo) dwords.  On the device (fp16 planes): hi = v_cvt_pk_f16_f32 (a packed CONVERSION),
// lo = fp16(x - float(hi)) in ONE instruction per value: v_fma_mix{lo,hi}_f16 takes the fp16 half of `hi2` and the fp32 value as
// mixed-precision sources (hi * -1.0 + x: exact in fp32, then one RNE rounding to fp16) -- bit for bit the value of the two-step form
// (tests/test_gpu_round4.py::test_operand_split_is_bit_exact...), 3 instructions per pair instead of 6, same speed (366.7 / 366.5 vs
// 366.7 / 367.1 motions/s, profiles/r04b_ab.md).  Default since round 5 BECAUSE IT CONTAINS NO PACKED fp32 VALU MATH: the two-step form
// subtracts on a 2-vector (v_pk_add_f32) -- the one packed fp32 instruction that was left in the product library (4,327 instances)
// after -fno-slp-vectorize, i.e. the instruction class of the unexplained co-residency corruption (include/mdm_hip.h CONCURRENCY,
// profiles/r04c_packed_math.md).  tests/test_abi.py disassembles the product library and fails on any v_pk_*_f32.
// -DMDM_SPLIT_PKSUB builds the two-step form for A/B runs.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_p16(float a, float b, uint32_t& hi2, uint32_t& lo2) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (kSplitF16) {
    const f32x2 v = {a, b};
    const f16x2 h = __builtin_convertvector(v, f16x2);       // v_cvt_pk_f16_f32
    hi2 = __builtin_bit_cast(uint32_t, h);
#ifndef MDM_SPLIT_PKSUB
    uint32_t l;
    // HARDWARE FINDING (round 5, profiles/r05m_fma_mix_hazard.md): a v_mfma that reads a register written by v_fma_mix{lo,hi}_f16 needs
    // wait states in between, and hipcc's hazard recognizer does not look into inline asm -- it put the MFMA one instruction behind the
    // pair, and xattn_block_kernel<4, 3> (70 memory tokens: its K / V / P fragments go from this split straight into MFMAs) returned
    // 7e-2 errors that changed from run to run; every other kernel got its distance by luck of the schedule.  Two wait states behind
    // the pair cure it (bisected on the MI355X: the two-step form and this form pass, the bare pair and the pair as one statement fail).
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(hi2), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\ts_nop 1" : "+v"(l) : "v"(hi2), "v"(b));
    lo2 = l;
#else
    const f32x2 r = v - __builtin_convertvector(h, f32x2);   // v_pk_add_f32
    const f16x2 l = __builtin_convertvector(r, f16x2);
    lo2 = __builtin_bit_cast(uint32_t, l);
#endif
    return;
  }
#endif
  p16_t h0, l0, h1, l1;
  split_p16(a, h0, l0);
  split_p16(b, h1, l1);
  hi2 = (uint32_t)h0 | ((uint32_t)h1 << 16);
  lo2 = (uint32_t)l0 | ((uint32_t)l1 << 16);
}

// 4 consecutive values -> 8-byte packed hi and lo groups
// 8 consecutive fp32 values -> one MFMA operand fragment each of hi and lo
__device__ __forceinline__ void split8(const float* v, p16x8& hi, p16x8& lo) {
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) split2_p16(v[2 * j], v[2 * j + 1], h[j], l[j]);
  hi = __builtin_bit_cast(p16x8, u32x4{h[0], h[1], h[2], h[3]});
  lo = __builtin_bit_cast(p16x8, u32x4{l[0], l[1], l[2], l[3]});
}

__device__ __forceinline__ void split4_store(p16_t* hi_p, p16_t* lo_p, float4 v) {
  uint32_t h01, l01, h23, l23;
  split2_p16(v.x, v.y, h01, l01);
  split2_p16(v.z, v.w, h23, l23);
  *reinterpret_cast<uint2*>(hi_p) = make_uint2(h01, h23);
  *reinterpret_cast<uint2*>(lo_p) = make_uint2(l01, l23);
}

// Direct global -> LDS copy of 16 bytes per lane (global_load_lds_dwordx4): the LDS destination is
// `lds_wave_base + lane*16` (wave-uniform base, lane-linear image); the global source is per lane.
__device__ __forceinline__ void glds16(const void* gsrc_lane, void* lds_wave_base) {
#ifdef MDM_EMU
  emu::vm_issue(static_cast<char*>(lds_wave_base) + 16 * emu::lane_id(), gsrc_lane, 16, false);
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}

// the same with the non-temporal cache policy (aux = 2): for streams ONE workgroup reads once
__device__ __forceinline__ void glds16_nt(const void* gsrc_lane, void* lds_wave_base) {
#ifdef MDM_EMU
  emu::vm_issue(static_cast<char*>(lds_wave_base) + 16 * emu::lane_id(), gsrc_lane, 16, false);
#else
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc_lane,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 2);
#endif
}

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N-1>{}), so that the index can
// feed template arguments / inline-asm immediates (a `#pragma unroll` loop variable cannot)
template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// LDS fragment read that hipcc does NOT track (cdna_hip_programming.md 5.7 form (ii)): hipcc's waitcnt pass retires
// tracked ds_reads with lgkmcnt(0) only (measured: a full drain every third MFMA unit halves the matrix-pipe duty of
// the f16x3 main loop), so the hot loops issue their reads through lds_read16 and retire them IN ORDER with a counted
// lds_wait<N>(regs...) that names the registers becoming valid -- no consumer of those registers can be scheduled
// above the wait, and any copy the compiler made earlier is dead.
#ifdef MDM_EMU
__device__ __forceinline__ void lds_read16(p16x8& dst, const unsigned char* base, uint32_t byte_off) {
  emu::lds_issue(&dst, base + byte_off, 16);   // EARLY mode: immediate; LATE mode: withheld until a covering lds_wait
}
template <int N> __device__ __forceinline__ void lds_wait(p16x8&, p16x8&) { emu::lgkm_wait(N); }
template <int N> __device__ __forceinline__ void lds_wait(p16x8&, p16x8&, p16x8&, p16x8&) { emu::lgkm_wait(N); }
template <int N> __device__ __forceinline__ void lds_wait(p16x8&, p16x8&, p16x8&, p16x8&, p16x8&, p16x8&) { emu::lgkm_wait(N); }
template <int N>
__device__ __forceinline__ void lds_wait(p16x8&, p16x8&, p16x8&, p16x8&, p16x8&, p16x8&, p16x8&, p16x8&) { emu::lgkm_wait(N); }
#else
// `lds_addr` = 32-bit LDS byte address (lds_addr_of), IMM = compile-time byte offset < 65536
template <int IMM> __device__ __forceinline__ void lds_read16(p16x8& dst, uint32_t lds_addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(lds_addr), "i"(IMM));
}
template <int N> __device__ __forceinline__ void lds_wait(p16x8& a, p16x8& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N));
}
template <int N> __device__ __forceinline__ void lds_wait(p16x8& a, p16x8& b, p16x8& c, p16x8& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait(p16x8& a, p16x8& b, p16x8& c, p16x8& d, p16x8& e, p16x8& f, p16x8& g,
                                         p16x8& h) {
  asm volatile("s_waitcnt lgkmcnt(%8)"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
               : "i"(N));
}
template <int N>
__device__ __forceinline__ void lds_wait(p16x8& a, p16x8& b, p16x8& c, p16x8& d, p16x8& e, p16x8& f) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "i"(N));
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
  return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)p;
}
#endif

// Global 16-byte load that hipcc does NOT track (same form (ii) as lds_read16): beside an LDS-DMA in flight hipcc retires
// every tracked global load with vmcnt(0), i.e. one full memory latency per load.  The GEMM epilogue streams its
// residual tile through these instead, two row sub-tiles ahead, retired by a counted vmem_wait<N> where N = number of
// LOADS issued after the ones awaited (loads return in order; stores in between can only make the wait stricter).
#ifdef MDM_EMU
__device__ __forceinline__ void gload16_async(f32x4& dst, const float* p) { emu::vm_issue(&dst, p, 16, true); }
template <int N> __device__ __forceinline__ void vmem_wait(f32x4&, f32x4&, f32x4&, f32x4&) { emu::vm_wait(N); }
// 16-bit-plane flavour: one MFMA operand fragment (the pipelined k-loop's W stream, gemm_x3.h)
__device__ __forceinline__ void gload16_async(p16x8& dst, const p16_t* p) { emu::vm_issue(&dst, p, 16, true); }
__device__ __forceinline__ void gload16_refill(p16x8& slot, const p16_t* p) { emu::vm_issue(&slot, p, 16, true); }
template <int N> __device__ __forceinline__ void vmem_wait(p16x8&, p16x8&) { emu::vm_wait(N); }
template <int N> __device__ __forceinline__ void vmem_wait(p16x8&, p16x8&, p16x8&, p16x8&) { emu::vm_wait(N); }
template <int N> __device__ __forceinline__ void vmem_wait(p16x8&, p16x8&, p16x8&, p16x8&, p16x8&, p16x8&) { emu::vm_wait(N); }
template <int N>
__device__ __forceinline__ void vmem_wait(p16x8&, p16x8&, p16x8&, p16x8&, p16x8&, p16x8&, p16x8&, p16x8&) { emu::vm_wait(N); }
#else
// ("=&v": the destination never doubles as the address pair -- see gload16_refill below)
__device__ __forceinline__ void gload16_async(f32x4& dst, const float* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
template <int N> __device__ __forceinline__ void vmem_wait(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N) : "memory");
}
// 16-bit-plane flavour: one MFMA operand fragment (the pipelined k-loop's W stream, gemm_x3.h)
__device__ __forceinline__ void gload16_async(p16x8& dst, const p16_t* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
// IN-PLACE refill of a register slot whose previous value was an MFMA operand until just now.  The slot is an input AND the
// output ("+v"): its old value stays live -- in these very registers -- up to this instruction, so hipcc can neither recycle
// the registers as scratch behind the last MFMA that read them (address arithmetic, lane swaps done "in place"), nor let the
// destination double as the address pair.  Found on the MI355X (profiles/r03b_pipe_determinism.md): with a plain "=v"
// destination hipcc wrote the refill's own address into the slot one instruction behind the last of three dependent MFMAs
// that read it, and one launch in ~6 returned 4-lane groups of one wave's 32 columns computed on a half-rewritten fragment.
__device__ __forceinline__ void gload16_refill(p16x8& slot, const p16_t* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(slot) : "v"(p) : "memory");
}
template <int N> __device__ __forceinline__ void vmem_wait(p16x8& a, p16x8& b) {
  asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "i"(N) : "memory");
}
template <int N> __device__ __forceinline__ void vmem_wait(p16x8& a, p16x8& b, p16x8& c, p16x8& d) {
  asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(N) : "memory");
}
template <int N> __device__ __forceinline__ void vmem_wait(p16x8& a, p16x8& b, p16x8& c, p16x8& d, p16x8& e, p16x8& f) {
  asm volatile("s_waitcnt vmcnt(%6)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) : "i"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void vmem_wait(p16x8& a, p16x8& b, p16x8& c, p16x8& d, p16x8& e, p16x8& f, p16x8& g, p16x8& h) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
               : "i"(N)
               : "memory");
}
#endif

// 8-byte flavour (four 16-bit elements of one plane): the residual stream of the f16x3 mode lives only as hi/lo planes
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#ifdef MDM_EMU
__device__ __forceinline__ void gload8_async(u32x2& dst, const void* p) { emu::vm_issue(&dst, p, 8, true); }
template <int N>
__device__ __forceinline__ void vmem_wait(u32x2&, u32x2&, u32x2&, u32x2&, u32x2&, u32x2&, u32x2&, u32x2&) { emu::vm_wait(N); }
#else
__device__ __forceinline__ void gload8_async(u32x2& dst, const void* p) {
  asm volatile("global_load_dwordx2 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
template <int N>
__device__ __forceinline__ void vmem_wait(u32x2& a, u32x2& b, u32x2& c, u32x2& d, u32x2& e, u32x2& f, u32x2& g, u32x2& h) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)
               : "i"(N)
               : "memory");
}
#endif

// (float)(fp16 half HALF of `packed`) + c in ONE instruction: v_fma_mix_f32 converts its 16-bit source on the fly (h * 1.0 + c),
// where the plain form is a v_cvt_f32_f16 and an add per element.  fp16 planes only (kSplitF16).
template <int HALF> __device__ __forceinline__ float f16_half_plus(uint32_t packed, float c) {
#ifdef MDM_EMU
  return (float)__builtin_bit_cast(f16x2, packed)[HALF] + c;
#else
  float d;
  if constexpr (HALF == 0) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(c));
  else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(packed), "v"(c));
  return d;
#endif
}

// A window of `bytes` bytes of global memory addressed through a raw buffer descriptor (V#, stride 0) by 32-bit byte offsets:
// the buffer unit drops stores and returns 0 for loads whose offset lies past the window, so a GEMM tile's pad rows and the
// columns past the matrix cost neither a predicate (exec-mask branch) nor a clamp, and a round's address is ONE 32-bit add
// instead of a 64-bit multiply-add per plane.  (The range check covers VGPR offset + immediate, not the SGPR offset: the row
// term has to be in the VGPR.)  `CLIP_OFF` is an offset that is out of every window.
constexpr uint32_t CLIP_OFF = 0x80000000u;
typedef int i32x4 __attribute__((ext_vector_type(4)));
struct ClipWin {
#ifdef MDM_EMU
  char* base;
  uint32_t bytes;
#else
  i32x4 d;
#endif
};
__device__ __forceinline__ ClipWin clip_win(const void* base, uint32_t bytes) {
  ClipWin w;
#ifdef MDM_EMU
  w.base = const_cast<char*>(static_cast<const char*>(base));
  w.bytes = bytes;
#else
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  w.d[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
  w.d[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)((b >> 32) & 0xffffu));   // stride 0, no swizzle
  w.d[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  w.d[3] = 0x00020000;                                                              // raw buffer, 32-bit data format
#endif
  return w;
}
__device__ __forceinline__ void clip_store8(const ClipWin& w, uint32_t off, u32x2 v) {
#ifdef MDM_EMU
  if (off < w.bytes && off + 8 <= w.bytes) *reinterpret_cast<u32x2*>(w.base + off) = v;
#else
  // 8 bytes of store data: no VMEM-store-data hazard window on gfx9 (that one starts above 8 bytes)
  asm volatile("buffer_store_dwordx2 %0, %1, %2, 0 offen" : : "v"(v), "v"(off), "s"(w.d) : "memory");
#endif
}
__device__ __forceinline__ void split4_store_clip(const ClipWin& hi_w, const ClipWin& lo_w, uint32_t off, float4 v) {
  uint32_t h01, l01, h23, l23;
  split2_p16(v.x, v.y, h01, l01);
  split2_p16(v.z, v.w, h23, l23);
  clip_store8(hi_w, off, u32x2{h01, h23});
  clip_store8(lo_w, off, u32x2{l01, l23});
}
// untracked load (cf. gload8_async): retire it with vmem_wait
__device__ __forceinline__ void clip_load8_async(u32x2& dst, const ClipWin& w, uint32_t off) {
#ifdef MDM_EMU
  static const uint32_t zeros[2] = {0u, 0u};
  const bool in = off < w.bytes && off + 8 <= w.bytes;
  emu::vm_issue(&dst, in ? static_cast<const void*>(w.base + off) : static_cast<const void*>(zeros), 8, true);
#else
  asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=&v"(dst) : "v"(off), "s"(w.d) : "memory");
#endif
}

// Workgroup barrier that does NOT drain the vector-memory queue (cdna_hip_programming.md section 5: __syncthreads()
// would wait vmcnt(0) while an LDS-DMA is in flight); pair it with an explicit wait where the data is consumed.
__device__ __forceinline__ void wg_barrier() {
#ifdef MDM_EMU
  emu::lgkm_wait(0);
  emu::block_barrier();
#else
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0); the builtin (unlike inline asm) is visible to hipcc's waitcnt pass
  __builtin_amdgcn_s_barrier();
#endif
}
// The bare rendezvous: neither queue is drained -- untracked fragment reads (lds_read16) and LDS-DMA pieces stay in flight
// across it.  What it orders is only what each wave retired with its own counted waits BEFORE arriving.
__device__ __forceinline__ void wg_barrier_nodrain() {
#ifdef MDM_EMU
  emu::block_barrier();
#else
  __builtin_amdgcn_s_barrier();
#endif
}
// Lanes of one wave exchange data through a wave-private LDS region: the hardware executes a wave's LDS
// instructions in order, so no barrier instruction is needed -- only the emulator (lanes are fibers) must rendezvous.
__device__ __forceinline__ void wave_lds_fence() {
#ifdef MDM_EMU
  emu::wave_barrier();
#else
  __builtin_amdgcn_wave_barrier();
#endif
}
__device__ __forceinline__ void wait_vmem_upto3() {  // at most 3 vector-memory operations of this wave still pending
#ifdef MDM_EMU
  emu::vm_wait(3);
#else
  __builtin_amdgcn_s_waitcnt(0x0F73);  // vmcnt(3)
#endif
}
__device__ __forceinline__ void wait_vmem_all() {
#ifdef MDM_EMU
  emu::vm_wait(0);
#else
  __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#endif
}
template <int N> __device__ __forceinline__ void wait_vmem_upto() {   // vmcnt(N), N <= 15 in this encoding
  static_assert(N >= 0 && N <= 15, "vmcnt low bits only");
#ifdef MDM_EMU
  emu::vm_wait(N);
#else
  __builtin_amdgcn_s_waitcnt(0x0F70 | N);
#endif
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }

// XCD-aware bijective remap of a linear workgroup id (cdna_hip_programming.md T1): hardware places
// block b on XCD b%8; give each XCD a contiguous chunk of logical tiles so neighbours share its L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  if (nwg < 2 * nx) return bid;
  int xcd = bid % nx, q = nwg / nx, r = nwg % nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + bid / nx;
}

// ---------------------------------------------------------------------------------------------
// Counter-based RNG: Philox4x32-10 keyed by (seed), counter = (element, global sample, step, stream).
// One call per output element keeps the stream independent of sharding and of the launch geometry.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                                      uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                       uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Standard normal for (element e of global sample n, draw index `step`): Box-Muller on two 24-bit uniforms.
__host__ __device__ __forceinline__ float philox_normal(uint64_t seed, uint32_t elem, uint32_t sample, uint32_t step) {
  uint32_t r[4];
  philox4x32_10(elem, sample, step, 0x4d444d31u /* "MDM1" */, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const float u1 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

}  // namespace mdm

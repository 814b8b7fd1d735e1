// Fused multi-head self-attention for the MDM encoder in split precision (three 16-bit MFMA products per fp32
// product, fp32 accumulate -- the same arithmetic class as gemm_x3.h), fed by the fp16 hi/lo planes that the
// in_proj GEMM's epilogue writes in exactly the layouts this kernel wants.
//
// Replaces, per layer, torch's head split / scaled_dot_product_attention / head merge
// (F.multi_head_attention_forward under nn.TransformerEncoderLayer, model/mdm.py:77-84, :253) and
// the key-padding mask of mdm.py:241-247.  The exact-fp32 kernel (attention_f32.h) remains the `f32` mode.
//
// Operand planes (all 16-bit -- fp16 by default, common.h kSplitF16 --, hi and lo; SP = tokens padded to a multiple of 32, NKT = SP/32):
//   Q, K   [nseq][H][SP][128]            row = token, Q pre-scaled by 1/sqrt(128); pad rows (>= S): never read
//   V^T    [nseq][H][NKT][128][32]       per 32-key tile: row = d, 32 keys of that tile in MFMA ORDER: inside each
//                                        group of 16 keys, position p holds key (p&3) + 8*((p>>2)&1) + 4*(p>>3).
//                                        That is the order in which a lane of a 32x32 MFMA accumulator holds its 16
//                                        rows, so (a) the in_proj epilogue stores V^T straight from its accumulators
//                                        with 16-byte stores and (b) the probabilities, which come out of phase 1 in
//                                        accumulator registers, are already the matching B operand.  Pad keys: finite.
//
// Two workgroups per (sequence, head); a wave owns 32 queries.  Everything is computed TRANSPOSED so the
// softmax axis is lane-local (cdna_hip_programming.md T12's "swapped QK^T"):
//   phase 1  St[key][query] = K . Q^T    A = K fragments (LDS, XOR-swizzled 256-byte rows, conflict-free b128 reads)
//                                        B = this wave's Q fragments (registers, loaded once)
//   softmax  per lane over its 16 keys per tile x NKT tiles + ONE cross-half shuffle; normalisation deferred to O
//   phase 2  Ot[d][query]  = V^T . P^T   A = V^T fragments (LDS, 64-byte rows swizzled like gemm_x3.h)
//                                        B = split(P) straight from the phase-1 registers
// K and V^T arrive by global_load_lds_dwordx4 (no staging registers) through a ring of LDS slots (see the kernel).
#pragma once
#include "common.h"

namespace mdm {

constexpr int AX_HD = 128;
constexpr int AX_OLD = AX_HD + 4;  // fp32 output staging row stride (floats)

struct QkvPlanes {
  p16_t *qh, *ql, *kh, *kl, *vh, *vl;
  int SP, NKT, H;
};

// position p (0..15) inside a 16-key group  <->  key offset, the MFMA accumulator row order (see header)
__host__ __device__ __forceinline__ int ax_key_of_pos(int p) { return (p & 3) + 8 * ((p >> 2) & 1) + 4 * (p >> 3); }

// Streaming form.  256 threads = 4 waves per workgroup, TWO workgroups per (sequence, head) -- query tiles 0-3 and 4-6 --
// and two workgroups resident per CU (<= 68 KB of LDS, <= 256 VGPRs each), so that one workgroup's load / softmax /
// store phases hide behind the other's MFMAs (the one-workgroup-per-CU predecessor held all of K in LDS and exposed a
// 112 KB load at the start of each of its four back-to-back items: 152 us per launch for 456 MB).
//   * K then V^T stream through a ring of four 16 KB LDS slots as 2*NKT uniform tiles (K tile: hi|lo [32 keys][256 B],
//     V^T tile: hi|lo [128 d][64 B]); every wave issues 4 LDS-DMA pieces per tile, three tiles ahead; one counted
//     vmcnt + one barrier per tile;
//   * all NKT score tiles of a wave's 32 queries stay in registers, so the softmax is exact (no online rescaling).
//   * the workgroups are PERSISTENT (grid = two per CU): a workgroup walks over its items, and while it finishes item i
//     (last V^T tile, output) the first key tile and the Q fragments of item i+1 are already on their way -- the load
//     latency at the start of an item and the output phase at its end were 26 + 19 of the kernel's 105 us otherwise.
constexpr int AX_SLOT = 16384, AX_RING = 4;
constexpr int AX_OST = 64 + 4;   // output staging row stride (floats): 32 queries x 64 d per wave and pass, in ring slots 1-3
constexpr int AX_KMASK_BYTES = 1024;   // per-key additive mask (0 / -inf) of the current item: only written for bitmap masks
constexpr int ax_lds_bytes(int) { return AX_RING * AX_SLOT + AX_KMASK_BYTES; }
static_assert(AX_SLOT + 4 * 32 * AX_OST * 4 <= AX_RING * AX_SLOT, "output staging must fit behind ring slot 0");

// ABL (timing experiments only, 0 in production; results are garbage): 1 = no MFMAs, 2 = no LDS fragment reads,
// 4 = no softmax, 8 = no output staging / stores, 16 = no K / V^T streaming after the prologue, 32 = no per-tile barrier.
// Results stay CORRECT with: 64 = non-temporal K / V^T LDS-DMA, 128 = non-temporal plane stores (A/B: profiles/r04b_ab.md).
// DIRECT (round 5; VERDICT r04 item 4, the item-boundary drain): the output leaves the accumulators as 8-byte plane stores
// (lane = query row, 4 consecutive d per register quad) instead of being staged through ring slots 1-3 -- so those slots are free the
// moment the item's last tile has been consumed, and the NEXT item's key tiles 1 and 2 are requested right behind the closing
// rendezvous, in front of the stores; the item then starts without a queue drain: its first three tile waits count the stores that are
// still in flight (32 per active wave; vmcnt counts stores on gfx9).  Planes only (the model's path); the fp32-output form of the
// building-block API keeps the staged epilogue.
template <int N> __device__ __forceinline__ void ax_vm_wait() {      // s_waitcnt vmcnt(N), N <= 63 (common.h's builtin form stops at 15)
#ifdef MDM_EMU
  emu::vm_wait(N);
#else
  asm volatile("s_waitcnt vmcnt(%0)" : : "i"(N) : "memory");
#endif
}
// DIRECT's plane stores are COUNTED by the next item's first tile waits (AX_NST = 32 per active wave): they are issued from inline
// asm, one instruction per plane and register quad, so that the count cannot follow a compiler decision to merge, split or reorder
// them (the C++ form relied on hipcc emitting exactly two 8-byte stores per call).
__device__ __forceinline__ void ax_split4_store_counted(p16_t* hi_p, p16_t* lo_p, float4 v) {
#ifdef MDM_EMU
  split4_store(hi_p, lo_p, v);
#else
  uint32_t h01, l01, h23, l23;
  split2_p16(v.x, v.y, h01, l01);
  split2_p16(v.z, v.w, h23, l23);
  asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(hi_p), "v"(u32x2{h01, h23}) : "memory");
  asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(lo_p), "v"(u32x2{l01, l23}) : "memory");
#endif
}
template <int NKT, int ABL = 0, bool DIRECT = false>
__global__ __launch_bounds__(256, 2) void attention_x3_kernel(QkvPlanes P, const int* __restrict__ lengths,
                                                                    int S, int D, int B, int lead, float* __restrict__ out,
                                                                    p16_t* __restrict__ oh, p16_t* __restrict__ ol,
                                                                    int items) {
  MDM_DYN_SMEM(unsigned char, lds);
  constexpr int SP = 32 * NKT;
  constexpr int NTILES = 2 * NKT;   // K tiles then V^T tiles

  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef MDM_EMU
  const int w = tid >> 6;
#else
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int r = lane & 31, h = lane >> 5;
  const int H = P.H;
  // virtual blocks vb and vb+8 run on the same XCD (block -> XCD b % 8; the grid is a multiple of 16): they are the two query
  // halves of one (sequence, head), so the second reader of K / V^T hits L2
  int vb = (int)blockIdx.x;
  const int vb_total = (items + 7) / 8 * 16, vb_step = (int)gridDim.x;
  const int half = (vb >> 3) & 1;          // the same for every item of this workgroup (vb_step % 16 == 0)
  auto item_of = [](int b) { return (b >> 4) * 8 + (b & 7); };
  if (item_of(vb) >= items) return;   // whole workgroup (uniform); only the last group of 8 items can be ragged
  const int qt = 4 * half + w;            // this wave's query tile
  const bool active = qt < NKT;           // (the second half has NKT - 4 tiles; idle waves still move data and sync)
  const bool last_group = S > 32 * (NKT - 1) + 16;   // does the sequence reach into the last 16-key group?

  // ---- tile t -> ring slot t & 3.  16 pieces of 1 KB per tile, wave w issues pieces 4w .. 4w+3 (piece = plane, idx):
  // K tile kt: idx covers 4 keys x 256 B; lane -> (row = lane>>4, stored chunk = lane&15) fetches chunk ^ (key & 15);
  // V^T tile kt: idx covers 16 d-rows x 64 B; lane -> (row = lane>>2, stored chunk = lane&3) fetches chunk ^ ((d>>2)&3).
  // `lv` = the lane id, re-laundered per item (see the item loop): the per-lane source offsets of the 4 * NTILES pieces are
  // the same for every item, and hipcc otherwise hoists all of them out of the item loop and spills them (120 VGPRs)
  auto issue_tile = [&](size_t sh, int t, int lv) {   // tile t of (sequence, head) sh
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 4 * w + i, plane = j >> 3, idx = j & 7;
      unsigned char* dst = lds + (t & (AX_RING - 1)) * AX_SLOT + plane * 8192 + idx * 1024;
      if (t < NKT) {
        // pad keys (>= S) are fetched from the last real row instead: finite, cache-resident, and no HBM traffic for rows
        // nobody wrote; the swizzle still follows the LDS row
        const int key = 32 * t + 4 * idx + (lv >> 4);
        const p16_t* src = (plane ? P.kl : P.kh) + (sh * SP + (size_t)min(key, S - 1)) * AX_HD + (((lv & 15) ^ (key & 15)) * 8);
        if constexpr ((ABL & 64) != 0) glds16_nt(src, dst); else glds16(src, dst);
      } else {
        const int d = 16 * idx + (lv >> 2);
        const p16_t* src = (plane ? P.vl : P.vh) + ((sh * NKT + (size_t)(t - NKT)) * AX_HD + d) * 32 + (((lv & 3) ^ ((d >> 2) & 3)) * 8);
        if constexpr ((ABL & 64) != 0) glds16_nt(src, dst); else glds16(src, dst);
      }
    }
  };
  // ---- this wave's Q fragments: query q = 32 qt + r, k-step st covers d = 16 st + 8h .. +7
  p16x8 qh[8], ql[8];
  auto load_q = [&](size_t sh) {
    const size_t qo = (sh * SP + min(32 * (active ? qt : 0) + r, S - 1)) * AX_HD + 8 * h;   // pad queries: last real row
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      qh[st] = *reinterpret_cast<const p16x8*>(P.qh + qo + 16 * st);
      ql[st] = *reinterpret_cast<const p16x8*>(P.ql + qo + 16 * st);
    }
  };
  issue_tile((size_t)item_of(vb), 0, lane);
  load_q((size_t)item_of(vb));
  bool first_item = true;     // DIRECT: items after the first start with tiles 1, 2 requested and the previous item's stores in flight
  constexpr int AX_NST = 32;  // DIRECT: plane stores of an active wave per item (4 d blocks x 4 register quads x hi, lo)

  for (;;) {   // ---- one item = one (sequence, head); vb advances by the grid size
  const int item = item_of(vb);
  const int seq = item / H, head = item - seq * H;
  const size_t sh = (size_t)item;   // == seq * H + head
  const int vb_next = vb + vb_step;
  const bool has_next = vb_next < vb_total && item_of(vb_next) < items;
  // the `lead` tokens in front of the frames are never masked: trans_enc's condition token (lead = 1; frame j - 1 must be < length,
  // mdm.py:241-247), none for trans_dec (lead = 0: its counts / bitmaps cover the context_len prefix frames, mdm.py:203-206)
  int nvalid = S;
  const uint32_t* kbits = nullptr;   // arbitrary frame mask of this sequence (common.h key_valid_bits), else a count
  if (lengths != nullptr) {
    const int cnt = lengths[seq % B];
    if (cnt >= 0) nvalid = min(S, lead + cnt);
    else kbits = reinterpret_cast<const uint32_t*>(lengths + B + 8 * (seq % B));
  }
  if (tid < 32 * NKT) {
    // additive key mask of this item, key `tid`: the lead tokens are always valid; frame f = key - lead by the count
    // or by its bitmap bit; keys >= S never.  Read at the softmax, NKT workgroup barriers from here; the previous item's
    // softmax is at least NKT barriers in the past.
    const int f = max(tid - lead, 0);
    const bool ok = kbits == nullptr ? tid < nvalid : (tid < S && (tid < lead || ((kbits[f >> 5] >> (f & 31)) & 1u)));
    reinterpret_cast<float*>(lds + AX_RING * AX_SLOT)[tid] = ok ? 0.f : -INFINITY;
  }
  int lv = lane;
#ifndef MDM_EMU
  asm volatile("" : "+v"(lv));   // opaque per item: nothing derived from it is hoisted out of the item loop
#endif

  const bool carried = DIRECT && !first_item;     // tiles 1, 2 are on their way and the previous item's stores may be in flight
  if (!carried) {
    wait_vmem_all();          // Q and tile 0 of this item (requested under the previous item), the previous item's stores
    wg_barrier();             // every wave is done with the previous item's output staging (ring slots 1-3)
    if (NTILES > 1) issue_tile(sh, 1, lv);
    if (NTILES > 2) issue_tile(sh, 2, lv);
  }

  f32x16 p[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int e = 0; e < 16; ++e) p[kt][e] = 0.f;
  f32x16 o[4];      // zeroed at the first V^T tile: must not be live (64 VGPRs) while the Q fragments are
  float inv = 1.f;

  // lane parts of the fragment addresses inside a slot: K row r, chunk h ^ (r & 15) (k step st flips chunk bits 1..3);
  // V^T row r of a 32-d block, chunk h ^ ((r>>2)&3) (s2 flips chunk bit 1)
  const uint32_t klane = (uint32_t)(r * 256 + ((h ^ (r & 15)) * 16));
  const uint32_t vlane = (uint32_t)(r * 64 + ((h ^ ((r >> 2) & 3)) * 16));
#ifndef MDM_EMU
  const uint32_t lbase = lds_addr_of(lds);
#endif

  static_for<NTILES>([&](auto t_tag) __attribute__((always_inline)) {
    constexpr int t = decltype(t_tag)::value;
    constexpr int slot = t & (AX_RING - 1);
    // tile t landed?  tiles t+1, t+2 (4 pieces each of this wave) may stay in flight; LDS-DMA retires in order
    constexpr int ahead = (NTILES - 1 - t) < 2 ? (NTILES - 1 - t) : 2;
    if constexpr (DIRECT && t < 3 && NTILES > 3) {
      // a carried item: behind tile t sit (t == 0: the 16 Q fragment loads,) the tiles requested since and the previous item's
      // stores -- issued behind tile 2, in front of tile 3
      // (the emulator's queue holds the untracked operations only -- LDS-DMA pieces; compiler-tracked loads and stores take effect
      // at once there: kTrk = 0)
#ifdef MDM_EMU
      constexpr int kTrk = 0;
#else
      constexpr int kTrk = 1;
#endif
      if (carried && active) ax_vm_wait<4 * ahead + kTrk * (AX_NST + (t == 0 ? 16 : 0))>();
      else if (carried) ax_vm_wait<4 * ahead + kTrk * (t == 0 ? 16 : 0)>();
      else wait_vmem_upto<4 * ahead>();
    } else if constexpr (!(ABL & 16)) wait_vmem_upto<4 * ahead>();
    if constexpr (!(ABL & 32)) wg_barrier();  // tile t visible to every wave; every wave is done with tile t-1 (whose slot is refilled now)
    if constexpr (t + 3 < NTILES && !(ABL & 16)) issue_tile(sh, t + 3, lv);
    if constexpr (t == NTILES - 1) {
      // the next item's first key tile (slot 0: its last tenant, tile NTILES-2 or earlier, is consumed) and Q fragments
      // (their registers have been dead since phase 1) travel under this item's last V^T tile and its output phase
      // (the Q load is unconditional -- the last item re-fetches its own -- so that the old fragments are dead after phase 1
      // on every path; conditionally kept, they would stay live through phase 2: 64 VGPRs, 120 spilled)
      if (has_next) issue_tile((size_t)item_of(vb_next), 0, lv);
      load_q(has_next ? (size_t)item_of(vb_next) : sh);
    }

    if constexpr (t < NKT) {
      // ---- phase 1, key tile t: St[key][query] += K . Q^T, three products per 16-deep k step
      if (active) {
        p16x8 kh[3], kl[3];
#ifdef MDM_EMU
#define AX_RD_K(dst, plane, st) lds_read16(dst, lds, slot * AX_SLOT + (plane) * 8192 + (klane ^ ((st) << 5)))
#else
        const uint32_t kb = lbase + klane;
#define AX_RD_K(dst, plane, st) lds_read16<slot * AX_SLOT + (plane) * 8192>(dst, kb ^ (uint32_t)((st) << 5))
#endif
        static_for<8 + 2>([&](auto u_tag) __attribute__((always_inline)) {
          constexpr int u = decltype(u_tag)::value;
          if constexpr (u < 8 && !(ABL & 2)) {
            AX_RD_K(kh[u % 3], 0, u);
            AX_RD_K(kl[u % 3], 1, u);
          }
          if constexpr (u >= 2) {
            constexpr int st = u - 2;
            constexpr int younger = 2 * ((7 - st) < 2 ? (7 - st) : 2);
            lds_wait<younger>(kh[st % 3], kl[st % 3]);
#ifndef MDM_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
            if constexpr (ABL & 1) {
#ifndef MDM_EMU
              asm volatile("" ::"v"(kl[st % 3]), "v"(kh[st % 3]), "v"(qh[st]), "v"(ql[st]));
#endif
            } else {
              p[t] = mfma_p16(kl[st % 3], qh[st], p[t]);
              p[t] = mfma_p16(kh[st % 3], ql[st], p[t]);
              p[t] = mfma_p16(kh[st % 3], qh[st], p[t]);
            }
#ifndef MDM_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
        });
#undef AX_RD_K
      }
      if constexpr (t == NKT - 1 && !(ABL & 4)) {
        // ---- softmax over keys: lane-local + one cross-half exchange; 1/sum is applied to the output
        float mx = -INFINITY;
        {
          // key-padding mask: the item's per-key additive mask (0 or -inf; count or bitmap form of `lengths`, built at the
          // item start) is read as four 16-byte groups per key tile -- a lane's sixteen scores of a tile are keys
          // 8 g + 4 h .. + 3, g = 0..3 (common.h mfma_row).  One code path for both mask forms: bitmap words in registers at
          // this kernel's register peak cost 147 spilled VGPRs (round 3, measured: the kernel twice as slow).
          const float* km = reinterpret_cast<const float*>(lds + AX_RING * AX_SLOT) + 4 * h;
#pragma unroll
          for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float4 m4 = ld4(km + kt * 32 + 8 * g);
              const float s0 = p[kt][4 * g] + m4.x, s1 = p[kt][4 * g + 1] + m4.y, s2 = p[kt][4 * g + 2] + m4.z,
                          s3 = p[kt][4 * g + 3] + m4.w;
              p[kt][4 * g] = s0; p[kt][4 * g + 1] = s1; p[kt][4 * g + 2] = s2; p[kt][4 * g + 3] = s3;
              mx = fmaxf(mx, fmaxf(fmaxf(s0, s1), fmaxf(s2, s3)));
            }
        }
        mx = fmaxf(mx, shfl_xor_f32(mx, 32));
        // fp16 planes: the probabilities are split as hi / lo of p * 2^10 -- free, by lowering the subtracted maximum by
        // 10 ln 2; the factor cancels in 1 / sum.  Unscaled, a typical p ~ 1/S = 0.005 has an fp16-SUBNORMAL lo part
        // (2^-25 absolute = 2^-17 relative): P.V would be the one bf16-class product left in the path.
        if constexpr (kSplitF16) mx -= 6.931471805599453f;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
#ifdef MDM_EMU
            const float pe = expf(p[kt][e] - mx);  // exp(-inf) = 0 for masked keys; key 0 is always valid
#else
            const float pe = __expf(p[kt][e] - mx);  // v_exp_f32 of a non-positive argument (2 instructions, ~2 ulp)
#endif
            p[kt][e] = pe;
            sum += pe;
          }
        sum += shfl_xor_f32(sum, 32);
        inv = 1.0f / sum;
      }
    } else {
      // ---- phase 2, key tile kt: Ot[d][query] += V^T . P^T; the B operand of k-step (kt, s2) is split(p[kt][8 s2 .. +7])
      constexpr int kt = t - NKT;
      if constexpr (kt == 0) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
      }
      if (active) {
        p16x8 vh[3], vl[3], ph, pl;
#ifdef MDM_EMU
#define AX_RD_V(dst, plane, s2, dt) lds_read16(dst, lds, slot * AX_SLOT + (plane) * 8192 + (dt) * 2048 + (vlane ^ ((s2) << 5)))
#else
        const uint32_t vb = lbase + vlane;
#define AX_RD_V(dst, plane, s2, dt) lds_read16<slot * AX_SLOT + (plane) * 8192 + (dt) * 2048>(dst, vb ^ (uint32_t)((s2) << 5))
#endif
        static_for<8 + 2>([&](auto u_tag) __attribute__((always_inline)) {
          constexpr int u = decltype(u_tag)::value;
          if constexpr (u < 8 && !(ABL & 2)) {
            AX_RD_V(vh[u % 3], 0, u / 4, u % 4);
            AX_RD_V(vl[u % 3], 1, u / 4, u % 4);
          }
          if constexpr (u >= 2) {
            constexpr int uv = u - 2, s2 = uv / 4, dt = uv % 4;
            if constexpr (dt == 0) {
              float pv[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) pv[j] = p[kt][8 * s2 + j];
              split8(pv, ph, pl);
            }
            constexpr int younger = 2 * ((7 - uv) < 2 ? (7 - uv) : 2);
            lds_wait<younger>(vh[uv % 3], vl[uv % 3]);
#ifndef MDM_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
            // a 16-key group wholly past the sequence (only the last one can be) has p == 0 and V^T rows that the in_proj
            // GEMM's 208-row tiles never wrote: skipped, not multiplied
            if constexpr (ABL & 1) {
#ifndef MDM_EMU
              asm volatile("" ::"v"(vl[uv % 3]), "v"(vh[uv % 3]), "v"(ph), "v"(pl));
#endif
            } else if (kt < NKT - 1 || s2 == 0 || last_group) {
              o[dt] = mfma_p16(vl[uv % 3], ph, o[dt]);
              o[dt] = mfma_p16(vh[uv % 3], pl, o[dt]);
              o[dt] = mfma_p16(vh[uv % 3], ph, o[dt]);
            }
#ifndef MDM_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
        });
#undef AX_RD_V
      }
    }
  });
  wg_barrier();  // every wave is done reading the ring: slots 1-3 become the output staging area (slot 0 is being refilled)
  if constexpr (DIRECT) {
    // the next item's key tiles 1 and 2 into the free slots, then this item's output straight from the accumulators
    if (has_next) {
      if (NTILES > 1) issue_tile((size_t)item_of(vb_next), 1, lv);
      if (NTILES > 2) issue_tile((size_t)item_of(vb_next), 2, lv);
    }
    if (active) {
      const int qq = 32 * qt + r;
      if (qq < S) {
        const size_t obase = ((size_t)seq * S + qq) * D + head * AX_HD + 4 * h;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int g = 0; g < 4; ++g)
            ax_split4_store_counted(oh + obase + 32 * dt + 8 * g, ol + obase + 32 * dt + 8 * g,
                                    make_float4(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv));
      }
    }
    first_item = false;
    if (!has_next) break;
    vb = vb_next;
    continue;
  }

  // ---- stage this wave's O[32 queries][128 d] in two passes of 64 d (fp32, row stride AX_OST, wave-private) and store
  // coalesced: accumulator rows mfma_row(4g..4g+3, h) are 4 consecutive d
  float* so = reinterpret_cast<float*>(lds + AX_SLOT) + w * (32 * AX_OST);
  if (active && !(ABL & 8)) {
    const size_t obase = (size_t)seq * S * D + head * AX_HD;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int dt = 2 * pass + dd;
          st4(&so[r * AX_OST + dd * 32 + 8 * g + 4 * h],
              make_float4(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv));
        }
      wave_lds_fence();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + (lv >> 4), qq = 32 * qt + row, c4 = lv & 15;   // lane -> (row % 4, 4 consecutive d)
        if (qq < S) {
          const float4 v = ld4(&so[row * AX_OST + 4 * c4]);
          const size_t oo = obase + (size_t)qq * D + 64 * pass + 4 * c4;
          if (out != nullptr) st4(out + oo, v);
          if (oh != nullptr) {   // planes for the out_proj f16x3 GEMM
            if constexpr ((ABL & 128) != 0) {
#ifndef MDM_EMU
              uint32_t h01, l01, h23, l23;
              split2_p16(v.x, v.y, h01, l01);
              split2_p16(v.z, v.w, h23, l23);
              __builtin_nontemporal_store(u32x2{h01, h23}, reinterpret_cast<u32x2*>(oh + oo));
              __builtin_nontemporal_store(u32x2{l01, l23}, reinterpret_cast<u32x2*>(ol + oo));
#endif
            } else {
              split4_store(oh + oo, ol + oo, v);
            }
          }
        }
      }
      wave_lds_fence();
    }
  }
#ifndef MDM_EMU
  if constexpr ((ABL & 8) != 0) {   // keep the computation alive without the staging / stores
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) asm volatile("" ::"v"(o[dt]));
    asm volatile("" ::"v"(inv));
  }
#endif
  if (!has_next) break;
  vb = vb_next;
  }   // item loop
  wait_vmem_all();   // nothing of this workgroup may still be in flight towards its LDS when it is released
}

inline size_t attention_x3_lds_bytes(int nkt) { return (size_t)ax_lds_bytes(nkt); }

// Test / building-block helper: fp32 packed qkv [nseq*S][3D] (Q pre-scaled) -> the plane layouts above, pads zeroed.
// One thread per (sequence, head, padded token, d); 2-byte scattered stores -- not a hot-path kernel (in the model the
// in_proj GEMM epilogue writes these planes directly).
__global__ __launch_bounds__(256) void qkv_pack_kernel(const float* __restrict__ qkv, QkvPlanes P, int nseq, int S,
                                                       int D) {
  const int SP = P.SP, H = P.H;
  const size_t total = (size_t)nseq * H * SP * AX_HD;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int d = (int)(i % AX_HD);
    const int tok = (int)((i / AX_HD) % SP);
    const size_t shd = i / ((size_t)AX_HD * SP);
    const int head = (int)(shd % H), seq = (int)(shd / H);
    float q = 0.f, k = 0.f, v = 0.f;
    if (tok < S) {
      const float* row = qkv + ((size_t)seq * S + tok) * 3 * D + head * AX_HD + d;
      q = row[0]; k = row[D]; v = row[2 * D];
    }
    p16_t a, b;
    const size_t rk = (shd * SP + tok) * AX_HD + d;
    split_p16(q, a, b); P.qh[rk] = a; P.ql[rk] = b;
    split_p16(k, a, b); P.kh[rk] = a; P.kl[rk] = b;
    const int kt = tok >> 5, k32 = tok & 31, g16 = k32 >> 4, k16 = k32 & 15;
    int pos = 0;  // inverse of ax_key_of_pos
    for (int pp = 0; pp < 16; ++pp)
      if (ax_key_of_pos(pp) == k16) pos = pp;
    const size_t vk = ((shd * P.NKT + kt) * AX_HD + d) * 32 + 16 * g16 + pos;
    split_p16(v, a, b); P.vh[vk] = a; P.vl[vk] = b;
  }
}

}  // namespace mdm

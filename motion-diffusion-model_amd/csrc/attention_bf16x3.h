// Fused multi-head self-attention for the MDM encoder in split precision (three bf16 MFMA products per fp32
// product, fp32 accumulate -- the same arithmetic class as gemm_bf16x3.h), fed by the bf16 hi/lo planes that the
// in_proj GEMM's epilogue writes in exactly the layouts this kernel wants.
//
// Replaces, per layer, torch's head split / scaled_dot_product_attention / head merge
// (F.multi_head_attention_forward under nn.TransformerEncoderLayer, model/mdm.py:77-84, :253) and
// the key-padding mask of mdm.py:241-247.  The exact-fp32 kernel (attention_f32.h) remains the `f32` mode.
//
// Operand planes (all bf16, hi and lo; SP = tokens padded to a multiple of 32, NKT = SP/32):
//   Q, K   [nseq][H][SP][128]            row = token, Q pre-scaled by 1/sqrt(128); pad rows: any FINITE values
//   V^T    [nseq][H][NKT][128][32]       per 32-key tile: row = d, 32 keys of that tile in MFMA ORDER: inside each
//                                        group of 16 keys, position p holds key (p&3) + 8*((p>>2)&1) + 4*(p>>3).
//                                        That is the order in which a lane of a 32x32 MFMA accumulator holds its 16
//                                        rows, so (a) the in_proj epilogue stores V^T straight from its accumulators
//                                        with 16-byte stores and (b) the probabilities, which come out of phase 1 in
//                                        accumulator registers, are already the matching B operand.  Pad keys: finite.
//
// One workgroup per (sequence, head); wave w owns queries [32w, 32w+32).  Everything is computed TRANSPOSED so the
// softmax axis is lane-local (cdna_hip_programming.md T12's "swapped QK^T"):
//   phase 1  St[key][query] = K . Q^T    A = K fragments (LDS, XOR-swizzled 256-byte rows, conflict-free b128 reads)
//                                        B = this wave's Q fragments (registers, loaded once)
//   softmax  per lane over its 16 keys per tile x NKT tiles + ONE cross-half shuffle; normalisation deferred to O
//   phase 2  Ot[d][query]  = V^T . P^T   A = V^T fragments (LDS, 64-byte rows swizzled like gemm_bf16x3.h)
//                                        B = split(P) straight from the phase-1 registers
// K and V^T arrive by global_load_lds_dwordx4 (no staging registers).  The first V^T key tiles are fetched into spare LDS
// while phase 1 runs, the rest over the dead K planes while the softmax runs (LDS plan at the kernel).
#pragma once
#include "common.h"

namespace mdm {

constexpr int AX_HD = 128;
constexpr int AX_OLD = AX_HD + 4;  // fp32 output staging row stride (floats)

struct QkvPlanes {
  bf16_t *qh, *ql, *kh, *kl, *vh, *vl;
  int SP, NKT, H;
};

// position p (0..15) inside a 16-key group  <->  key offset, the MFMA accumulator row order (see header)
__host__ __device__ __forceinline__ int ax_key_of_pos(int p) { return (p & 3) + 8 * ((p >> 2) & 1) + 4 * (p >> 3); }

__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bf16_t a, b;
    split_bf16(v[j], a, b);
    hi[j] = (short)a;
    lo[j] = (short)b;
  }
}

// LDS plan (bytes).  Region A = the two K planes (2 * SP * 256); region B = as many 16 KB V^T key tiles (hi 8 KB | lo 8 KB)
// as still fit under 160 KB: they are fetched WHILE phase 1 runs.  The remaining V^T tiles are fetched into region A once
// every wave is done with K (they land during the softmax and the first part of phase 2); finally the fp32 output tile
// is staged over everything.
constexpr int ax_plane_bytes(int nkt) { return nkt * 32 * 256; }
constexpr int ax_early_tiles(int nkt) { return nkt < 3 ? nkt : 3; }  // 3 x 16 KB beside K(NKT = 7) = 160 KB; also keeps
                                                                     // every ds_read immediate offset below 64 KB
constexpr int ax_lds_bytes(int nkt) {
  const int kv = 2 * ax_plane_bytes(nkt) + ax_early_tiles(nkt) * 16384, o = nkt * 32 * AX_OLD * 4;
  return kv > o ? kv : o;
}

template <int NKT>
__global__ __launch_bounds__(64 * NKT) void attention_bf16x3_kernel(QkvPlanes P, const int* __restrict__ lengths,
                                                                      int S, int D, int B, float* __restrict__ out,
                                                                      bf16_t* __restrict__ oh, bf16_t* __restrict__ ol) {
  MDM_DYN_SMEM(unsigned char, lds);
  constexpr int NT = 64 * NKT;
  constexpr int SP = 32 * NKT;
  constexpr int PLANE = ax_plane_bytes(NKT);
  constexpr int NB = ax_early_tiles(NKT);   // V^T key tiles prefetched into region B during phase 1
  constexpr int REGION_B = 2 * PLANE;
  constexpr int LATE_PIECES = (16 * (NKT - NB)) / NKT;  // LDS-DMA pieces of the late tiles every wave issues at least

  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef MDM_EMU
  const int w = tid >> 6;
#else
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int r = lane & 31, h = lane >> 5;
  const int H = P.H;
  const int seq = blockIdx.x / H, head = blockIdx.x - seq * H;
  const size_t sh = (size_t)seq * H + head;

  int nvalid = S;  // token 0 (the condition token) is never masked; frame j-1 must be < length (mdm.py:241-247)
  if (lengths != nullptr) nvalid = min(S, 1 + lengths[seq % B]);

  // ---- K planes -> region A.  One LDS-DMA instruction = 1 KB = 4 rows of 256 B; lane -> (row = lane>>4, stored chunk =
  // lane&15) fetches logical chunk (lane&15) ^ (row&15): the involution the fragment reads below repeat.
  {
    const bf16_t* kbase[2] = {P.kh + sh * SP * AX_HD, P.kl + sh * SP * AX_HD};
    for (int i = w; i < 16 * NKT; i += NKT) {
      const int plane = i / (8 * NKT), idx = i - plane * (8 * NKT);
      const int row = 4 * idx + (lane >> 4);
      const int chunk = (lane & 15) ^ (row & 15);
      glds16(kbase[plane] + (size_t)row * AX_HD + chunk * 8, lds + plane * PLANE + idx * 1024);
    }
  }
  // ---- this wave's Q fragments: query q = 32w + r, k-step st covers d = 16 st + 8h .. +7
  bf16x8 qh[8], ql[8];
  {
    const size_t qo = (sh * SP + 32 * w + r) * AX_HD + 8 * h;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      qh[st] = *reinterpret_cast<const bf16x8*>(P.qh + qo + 16 * st);
      ql[st] = *reinterpret_cast<const bf16x8*>(P.ql + qo + 16 * st);
    }
  }
  wait_vmem_all();
  wg_barrier();

  // ---- V^T tiles.  Per key tile 16 KB = hi [128 d][64 B] | lo [128 d][64 B]; one LDS-DMA instruction = 16 rows; lane ->
  // (row = lane>>2, stored chunk = lane&3) fetches logical chunk (lane&3) ^ ((row>>2)&3).  Tile kt lives in region B
  // (kt < NB) or, once K is dead, in region A.
  const bf16_t* vbase[2] = {P.vh + sh * SP * AX_HD, P.vl + sh * SP * AX_HD};
  auto v_tile_off = [&](int kt) { return kt < NB ? REGION_B + kt * 16384 : (kt - NB) * 16384; };
  auto issue_v = [&](int kt_lo, int kt_hi) {
    for (int i = 16 * kt_lo + w; i < 16 * kt_hi; i += NKT) {
      const int kt = i >> 4, j = i & 15, plane = j >> 3, idx = j & 7;   // 8 pieces of 16 d-rows per plane
      const int row = 16 * idx + (lane >> 2);                            // d
      const int chunk = (lane & 3) ^ ((row >> 2) & 3);
      glds16(vbase[plane] + ((size_t)kt * AX_HD + row) * 32 + chunk * 8, lds + v_tile_off(kt) + plane * 8192 + idx * 1024);
    }
  };
  issue_v(0, NB);

  // ---- phase 1: score tiles St[key][query], three products per 16-deep k step.  The 8*NKT (key tile, k step) units run as
  // ONE software pipeline: fragment reads two units ahead through untracked ds_reads, counted waits (common.h).
  f32x16 p[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int e = 0; e < 16; ++e) p[kt][e] = 0.f;
  {
    // lane part of a K fragment address: row r of the tile, chunk h ^ (r & 15); k step st flips chunk bits 1..3
    const uint32_t klane = (uint32_t)(r * 256 + ((h ^ (r & 15)) * 16));
    bf16x8 kh[3], kl[3];
    constexpr int NU1 = 8 * NKT;
#ifdef MDM_EMU
#define AX_RD_K(dst, plane, kt, st) lds_read16(dst, lds, (plane) * PLANE + (kt) * 8192 + (klane ^ ((st) << 5)))
#else
    const uint32_t kb0 = lds_addr_of(lds) + klane, kb1 = kb0 + PLANE;
#define AX_RD_K(dst, plane, kt, st) lds_read16<(kt) * 8192>(dst, ((plane) ? kb1 : kb0) ^ (uint32_t)((st) << 5))
#endif
    static_for<NU1 + 2>([&](auto u_tag) __attribute__((always_inline)) {
      constexpr int u = decltype(u_tag)::value;
      if constexpr (u < NU1) {
        AX_RD_K(kh[u % 3], 0, u / 8, u % 8);
        AX_RD_K(kl[u % 3], 1, u / 8, u % 8);
      }
      if constexpr (u >= 2) {
        constexpr int uv = u - 2, kt = uv / 8, st = uv % 8;
        constexpr int younger = 2 * ((NU1 - 1 - uv) < 2 ? (NU1 - 1 - uv) : 2);
        lds_wait<younger>(kh[uv % 3], kl[uv % 3]);
#ifndef MDM_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
        p[kt] = mfma_bf16(kl[uv % 3], qh[st], p[kt]);
        p[kt] = mfma_bf16(kh[uv % 3], ql[st], p[kt]);
        p[kt] = mfma_bf16(kh[uv % 3], qh[st], p[kt]);
#ifndef MDM_EMU
        __builtin_amdgcn_sched_barrier(0);
#endif
      }
    });
#undef AX_RD_K
  }

  wg_barrier();          // every wave is done reading K
  issue_v(NB, NKT);      // the late V^T tiles go over the K planes

  // ---- softmax over keys: lane-local + one cross-half exchange; 1/sum is applied to the output
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = kt * 32 + mfma_row(e, h);
      const float sc = (key < nvalid) ? p[kt][e] : -INFINITY;
      p[kt][e] = sc;
      mx = fmaxf(mx, sc);
    }
  mx = fmaxf(mx, shfl_xor_f32(mx, 32));
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float pe = expf(p[kt][e] - mx);  // exp(-inf) = 0 for masked keys; key 0 is always valid
      p[kt][e] = pe;
      sum += pe;
    }
  sum += shfl_xor_f32(sum, 32);
  const float inv = 1.0f / sum;

  // the early tiles were issued before the late ones and LDS-DMA retires in order: allow the late pieces to stay in flight
#ifndef MDM_EMU
  if constexpr (NB < NKT) __builtin_amdgcn_s_waitcnt(0x0F70 | (LATE_PIECES & 15) | ((LATE_PIECES >> 4) << 14));
  else __builtin_amdgcn_s_waitcnt(0x0F70);
#endif
  wg_barrier();

  // ---- phase 2: Ot[d][query]; the B operand of k-step (kt, s2) is split(p[kt][8 s2 .. 8 s2 + 7]); units = (kt, s2, d tile)
  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  {
    // lane part of a V^T fragment address: row r of a 32-d block, chunk h ^ ((r>>2)&3); s2 flips chunk bit 1
    const uint32_t vlane = (uint32_t)(r * 64 + ((h ^ ((r >> 2) & 3)) * 16));
    bf16x8 vh[3], vl[3];
#ifdef MDM_EMU
#define AX_RD_V(dst, plane, kt, s2, dt) \
  lds_read16(dst, lds, ((kt) < NB ? REGION_B + (kt) * 16384 : ((kt) - NB) * 16384) + (plane) * 8192 + (dt) * 2048 + (vlane ^ ((s2) << 5)))
#else
    const uint32_t va = lds_addr_of(lds) + vlane, vb = va + REGION_B;
#define AX_RD_V(dst, plane, kt, s2, dt) \
  lds_read16<((kt) < NB ? (kt) * 16384 : ((kt) - NB) * 16384) + (plane) * 8192 + (dt) * 2048>(dst, ((kt) < NB ? vb : va) ^ (uint32_t)((s2) << 5))
#endif
    auto phase2 = [&](auto lo_tag, auto hi_tag) __attribute__((always_inline)) {
      constexpr int KT_LO = decltype(lo_tag)::value, KT_HI = decltype(hi_tag)::value;
      constexpr int NU2 = 8 * (KT_HI - KT_LO);   // units of this part: (kt, s2, dt)
      if constexpr (NU2 > 0) {
        bf16x8 ph, pl;
        static_for<NU2 + 2>([&](auto u_tag) __attribute__((always_inline)) {
          constexpr int u = decltype(u_tag)::value;
          if constexpr (u < NU2) {
            constexpr int kt = KT_LO + u / 8, s2 = (u / 4) % 2, dt = u % 4;
            AX_RD_V(vh[u % 3], 0, kt, s2, dt);
            AX_RD_V(vl[u % 3], 1, kt, s2, dt);
          }
          if constexpr (u >= 2) {
            constexpr int uv = u - 2, kt = KT_LO + uv / 8, s2 = (uv / 4) % 2, dt = uv % 4;
            if constexpr (dt == 0) {
              float pv[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) pv[j] = p[kt][8 * s2 + j];
              split8(pv, ph, pl);
            }
            constexpr int younger = 2 * ((NU2 - 1 - uv) < 2 ? (NU2 - 1 - uv) : 2);
            lds_wait<younger>(vh[uv % 3], vl[uv % 3]);
#ifndef MDM_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
            o[dt] = mfma_bf16(vl[uv % 3], ph, o[dt]);
            o[dt] = mfma_bf16(vh[uv % 3], pl, o[dt]);
            o[dt] = mfma_bf16(vh[uv % 3], ph, o[dt]);
#ifndef MDM_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
        });
      }
    };
    phase2(std::integral_constant<int, 0>{}, std::integral_constant<int, NB>{});
    if constexpr (NB < NKT) {
      wait_vmem_all();
      wg_barrier();
      phase2(std::integral_constant<int, NB>{}, std::integral_constant<int, NKT>{});
    }
#undef AX_RD_V
  }
  wg_barrier();  // every wave is done reading V^T

  // ---- stage O[q][d] (fp32, row stride AX_OLD) and store coalesced: rows mfma_row(4g..4g+3, h) are 4 consecutive d
  float* so = reinterpret_cast<float*>(lds);
  const int q = 32 * w + r;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = dt * 32 + 8 * g + 4 * h;
      st4(&so[q * AX_OLD + d0], make_float4(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv,
                                            o[dt][4 * g + 3] * inv));
    }
  wg_barrier();
  const size_t obase = (size_t)seq * S * D + head * AX_HD;
  for (int idx = tid; idx < SP * 32; idx += NT) {
    const int qq = idx >> 5, c4 = idx & 31;
    if (qq < S) {
      const float4 v = ld4(&so[qq * AX_OLD + 4 * c4]);
      const size_t oo = obase + (size_t)qq * D + 4 * c4;
      if (out != nullptr) st4(out + oo, v);
      if (oh != nullptr) split4_store(oh + oo, ol + oo, v);  // planes for the out_proj bf16x3 GEMM
    }
  }
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
    bf16_t a, b;
    const size_t rk = (shd * SP + tok) * AX_HD + d;
    split_bf16(q, a, b); P.qh[rk] = a; P.ql[rk] = b;
    split_bf16(k, a, b); P.kh[rk] = a; P.kl[rk] = b;
    const int kt = tok >> 5, k32 = tok & 31, g16 = k32 >> 4, k16 = k32 & 15;
    int pos = 0;  // inverse of ax_key_of_pos
    for (int pp = 0; pp < 16; ++pp)
      if (ax_key_of_pos(pp) == k16) pos = pp;
    const size_t vk = ((shd * P.NKT + kt) * AX_HD + d) * 32 + 16 * g16 + pos;
    split_bf16(v, a, b); P.vh[vk] = a; P.vl[vk] = b;
  }
}

}  // namespace mdm

// Self- / cross-attention for sequences of MORE than 224 tokens (round 6; VERDICT r05 "What's missing" 5): streaming softmax.
//
// attention_x3.h / attention_f32.h keep all score tiles of a query row in registers (seven 32-key tiles: the softmax is exact, no
// rescaling) -- the one size limit of the seam (mdm_amd/mdm.py MAX_TOKENS = 224; the reference is bounded only by its positional
// table, model/mdm.py:55, :251-253).  The two kernels below take over above that: the SAME operands in the SAME layouts (the in_proj
// epilogue's Q / K / V^T planes with runtime SP / NKT; packed fp32 q | k | v rows), the same transposed formulation (a lane owns a
// query, its softmax axis is lane-local + one cross-half exchange), but the key tiles are walked ONCE with a running maximum m and
// sum l per query:
//     m' = max(m, max_j s_j),  c = exp(m - m'),  p_j = exp(s_j - m'),  l = c l + sum_j p_j,  O = c O + V^T p,   out = O / l
// (the same sum as the exact form, associated tile by tile; the parity tests hold it to the forward / loop tolerances of the suite).
// Key-padding masks as in the exact kernels: valid-key counts (keys are a prefix: the tiles past the last valid key are not even
// loaded) or per-sample bitmaps (at most 256 frames, mdm_amd/mdm.py frame_mask_lengths).  HumanML3D / KIT stop at 196 frames
// (sample/generate.py:32), so no BASELINE configuration runs these kernels: they are sized for correctness and a sane speed
// (LDS-DMA ring as in attention_x3.h, plain tracked fragment reads), not tuned.
#pragma once
#include "attention_f32.h"
#include "attention_x3.h"

namespace mdm {

constexpr int AL_QT = 4;                 // query tiles (= waves) of a workgroup: 128 queries
constexpr int al_x3_lds_bytes() { return AX_RING * AX_SLOT; }
inline int al_query_blocks(int Sq) { return ((Sq + 31) / 32 + AL_QT - 1) / AL_QT; }

// frame validity of key `key` under the bitmap form of `lengths` (common.h key_valid_bits, with a run-time tile index)
__device__ __forceinline__ bool al_key_valid(const uint32_t* kbits, int key, int S, int lead) {
  if (key >= S) return false;
  if (key < lead) return true;
  const int f = key - lead;
  return f < 256 && ((kbits[f >> 5] >> (f & 31)) & 1u);
}

// ---- split precision: Q / K / V^T operand planes (attention_x3.h layouts), 3 fp16 MFMA products per fp32 product.
// grid = nseq * H * nqb workgroups of 4 waves; workgroup (item, qb) owns queries [128 qb, 128 qb + 128) of (sequence, head) `item`.
// K and V^T tiles alternate through the ring of four 16 KB slots: tile u = (key tile u >> 1, V^T if u & 1), three tiles ahead,
// one counted vmcnt + one barrier per tile (attention_x3.h's scheme with a run-time tile count).
// (launch bounds without a residency promise: with two workgroups per CU promised the kernel spilled 88 bytes per lane)
__global__ __launch_bounds__(256) void attention_x3_long_kernel(QkvPlanes P, const int* __restrict__ lengths, int S, int D, int B,
                                                                   int lead, float* __restrict__ out, p16_t* __restrict__ oh,
                                                                   p16_t* __restrict__ ol, int nqb) {
  MDM_DYN_SMEM(unsigned char, lds);
  const int SP = P.SP, NKT = P.NKT, H = P.H;
  const int tid = threadIdx.x, lane = tid & 63;
#ifdef MDM_EMU
  const int w = tid >> 6;
#else
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int r = lane & 31, h = lane >> 5;
  const int item = (int)blockIdx.x / nqb, qb = (int)blockIdx.x - item * nqb;
  const int seq = item / H, head = item - seq * H;
  const size_t sh = (size_t)item;
  const int qt = AL_QT * qb + w;
  const bool active = 32 * qt < S;

  int nvalid = S;
  const uint32_t* kbits = nullptr;
  if (lengths != nullptr) {
    const int cnt = lengths[seq % B];
    if (cnt >= 0) nvalid = min(S, lead + cnt);
    else kbits = reinterpret_cast<const uint32_t*>(lengths + B + 8 * (seq % B));
  }
  // key tiles that hold a valid key (count form: a prefix).  A sequence without any valid key gives NaN like the exact kernels
  // (and the reference): one tile is still walked.
  const int nkt = kbits != nullptr ? (S + 31) / 32 : max(1, (nvalid + 31) / 32);
  const int ntiles = 2 * nkt;

  auto issue_tile = [&](int u) {   // tile u of this (sequence, head): 16 pieces of 1 KB, wave w issues pieces 4 w .. 4 w + 3
    const int kt = u >> 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int j = 4 * w + i, plane = j >> 3, idx = j & 7;
      unsigned char* dst = lds + (u & (AX_RING - 1)) * AX_SLOT + plane * 8192 + idx * 1024;
      if ((u & 1) == 0) {   // K tile: 4 keys x 256 B per piece; lane -> (row = lane >> 4, stored chunk = lane & 15) fetches chunk ^ (key & 15)
        const int key = 32 * kt + 4 * idx + (lane >> 4);
        const p16_t* src = (plane ? P.kl : P.kh) + (sh * SP + (size_t)min(key, S - 1)) * AX_HD + (((lane & 15) ^ (key & 15)) * 8);
        glds16(src, dst);
      } else {              // V^T tile: 16 d-rows x 64 B per piece; lane -> (row = lane >> 2, stored chunk = lane & 3) fetches chunk ^ ((d >> 2) & 3)
        const int d = 16 * idx + (lane >> 2);
        const p16_t* src = (plane ? P.vl : P.vh) + ((sh * NKT + (size_t)kt) * AX_HD + d) * 32 + (((lane & 3) ^ ((d >> 2) & 3)) * 8);
        glds16(src, dst);
      }
    }
  };

  issue_tile(0);
  // this wave's Q fragments: query q = 32 qt + r (pad queries: the last real row), k-step st covers d = 16 st + 8 h .. + 7
  p16x8 qh[8], ql[8];
  {
    const size_t qo = (sh * SP + min(32 * (active ? qt : 0) + r, S - 1)) * AX_HD + 8 * h;
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      qh[st] = *reinterpret_cast<const p16x8*>(P.qh + qo + 16 * st);
      ql[st] = *reinterpret_cast<const p16x8*>(P.ql + qo + 16 * st);
    }
  }
  if (ntiles > 1) issue_tile(1);
  if (ntiles > 2) issue_tile(2);

  const uint32_t klane = (uint32_t)(r * 256 + ((h ^ (r & 15)) * 16));
  const uint32_t vlane = (uint32_t)(r * 64 + ((h ^ ((r >> 2) & 3)) * 16));
  f32x16 o[4], sc;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
#pragma unroll
  for (int e = 0; e < 16; ++e) sc[e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int u = 0; u < ntiles; ++u) {
    // tile u landed?  tiles u + 1, u + 2 (4 pieces each of this wave) may stay in flight: LDS-DMA retires in order
    const int ahead = min(2, ntiles - 1 - u);
    if (ahead == 2) wait_vmem_upto<8>();
    else if (ahead == 1) wait_vmem_upto<4>();
    else wait_vmem_all();
    wg_barrier();   // tile u visible to every wave; every wave is done with tile u - 1 (whose slot is refilled now)
    if (u + 3 < ntiles) issue_tile(u + 3);
    const unsigned char* slot = lds + (u & (AX_RING - 1)) * AX_SLOT;
    const int kt = u >> 1;
    if ((u & 1) == 0) {
      // ---- scores of key tile kt: St[key][query] = K . Q^T, three products per 16-deep k step
      if (active) {
#pragma unroll
        for (int e = 0; e < 16; ++e) sc[e] = 0.f;
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          const p16x8 kh = *reinterpret_cast<const p16x8*>(slot + (klane ^ (uint32_t)(st << 5)));
          const p16x8 kl = *reinterpret_cast<const p16x8*>(slot + 8192 + (klane ^ (uint32_t)(st << 5)));
          sc = mfma_p16(kl, qh[st], sc);
          sc = mfma_p16(kh, ql[st], sc);
          sc = mfma_p16(kh, qh[st], sc);
        }
      }
    } else if (active) {
      // ---- running softmax over the tile's 32 keys (a lane holds 16 of them, its other half the rest), then Ot += V^T . P^T
      float mx = -INFINITY;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = 32 * kt + mfma_row(e, h);
        const bool ok = kbits == nullptr ? key < nvalid : al_key_valid(kbits, key, S, lead);
        const float s = ok ? sc[e] : -INFINITY;
        sc[e] = s;
        mx = fmaxf(mx, s);
      }
      mx = fmaxf(mx, shfl_xor_f32(mx, 32));
      const float m_new = fmaxf(m_run, mx);
      if (m_new != -INFINITY) {        // (a tile without a valid key before any valid key -- bitmap masks only -- contributes nothing)
        // fp16 planes: the probabilities are split as hi / lo of p * 2^10 (attention_x3.h); the factor cancels in 1 / l
        const float mref = kSplitF16 ? m_new - 6.931471805599453f : m_new;
        const float c = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float pe = expf(sc[e] - mref);    // exp(-inf) = 0 for masked keys
          sc[e] = pe;
          psum += pe;
        }
        psum += shfl_xor_f32(psum, 32);
        l_run = l_run * c + psum;
        m_run = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int e = 0; e < 16; ++e) o[dt][e] *= c;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          // a 16-key group wholly past the sequence has p == 0 and V^T entries nobody wrote: skipped, not multiplied
          if (32 * kt + 16 * s2 < S) {
            float pv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) pv[j] = sc[8 * s2 + j];
            p16x8 ph, pl;
            split8(pv, ph, pl);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
              const p16x8 vh = *reinterpret_cast<const p16x8*>(slot + dt * 2048 + (vlane ^ (uint32_t)(s2 << 5)));
              const p16x8 vl = *reinterpret_cast<const p16x8*>(slot + 8192 + dt * 2048 + (vlane ^ (uint32_t)(s2 << 5)));
              o[dt] = mfma_p16(vl, ph, o[dt]);
              o[dt] = mfma_p16(vh, pl, o[dt]);
              o[dt] = mfma_p16(vh, ph, o[dt]);
            }
          }
        }
      }
    }
  }
  wait_vmem_all();   // nothing of this workgroup may still be in flight towards its LDS when it is released
  // ---- output straight from the accumulators: lane = query row, registers 4 g .. 4 g + 3 = 4 consecutive d
  const int qq = 32 * qt + r;
  if (active && qq < S) {
    const float inv = 1.0f / l_run;
    const size_t obase = ((size_t)seq * S + qq) * D + head * AX_HD + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v = make_float4(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        const size_t oo = obase + 32 * dt + 8 * g;
        if (out != nullptr) st4(out + oo, v);
        if (oh != nullptr) split4_store(oh + oo, ol + oo, v);
      }
  }
}

// ---- exact fp32 (the `f32` mode, DiP's fp32 skeleton and its memory attention): AttnF32Args as attention_f32_kernel, any Sq / Sk.
// grid = nseq * H * nqb workgroups of 4 waves; per key tile the 32 K rows and 32 V rows are staged in LDS (33 KB).
constexpr int al_f32_lds_bytes() { return (32 * ATT_KLD + 32 * ATT_HD) * (int)sizeof(float); }
__global__ __launch_bounds__(256) void attention_f32_long_kernel(AttnF32Args a, float* __restrict__ out, int D, int H,
                                                                 p16_t* __restrict__ oh, p16_t* __restrict__ ol, int nqb) {
  MDM_DYN_SMEM(float, smem);
  float* const ks = smem;
  float* const vs = smem + 32 * ATT_KLD;
  const int S = a.Sk, Sq = a.Sq;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int item = (int)blockIdx.x / nqb, qb = (int)blockIdx.x - item * nqb;
  const int seq = item / H, head = item - seq * H;
  const int ld = a.ldkv;
  const float* qbase = a.q + (size_t)seq * Sq * a.ldq + head * ATT_HD;
  int kseq = seq;
  if (a.kv_B > 0) {
    const int br = seq / a.B;
    kseq = br * a.kv_B + a.kv_b0 + (seq - br * a.B);
  }
  const float* kbase = a.k + (size_t)kseq * S * ld + head * ATT_HD;
  const float* vbase = a.v + (size_t)kseq * S * ld + head * ATT_HD;

  int nvalid = S;
  const uint32_t* kbits = nullptr;
  if (a.lengths != nullptr) {
    const int lb = a.len_b0 + seq % a.B, LB = a.len_B > 0 ? a.len_B : a.B;
    const int cnt = a.lengths[lb];
    if (cnt >= 0) nvalid = min(S, a.lead + cnt);
    else kbits = reinterpret_cast<const uint32_t*>(a.lengths + LB + 8 * lb);
  }
  const int nkt = kbits != nullptr ? (S + 31) / 32 : max(1, (nvalid + 31) / 32);

  // Q fragment: query row q, this lane-half's 64 d's
  const int q = 32 * (AL_QT * qb + w) + r;
  float qf[64];
  {
    const float* qp = qbase + (size_t)q * a.ldq + 64 * h;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float4 v = (q < Sq) ? ld4(qp + 4 * j) : zero4();
      qf[4 * j + 0] = v.x; qf[4 * j + 1] = v.y; qf[4 * j + 2] = v.z; qf[4 * j + 3] = v.w;
    }
  }
  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int kt = 0; kt < nkt; ++kt) {
    __syncthreads();   // every wave is done with the previous tile
    for (int idx = tid; idx < 32 * 32; idx += 256) {   // rows >= S zero-filled
      const int row = idx >> 5, key = 32 * kt + row, c4 = idx & 31;
      float4 kv = zero4(), vv = zero4();
      if (key < S) {
        kv = ld4(kbase + (size_t)key * ld + 4 * c4);
        vv = ld4(vbase + (size_t)key * ld + 4 * c4);
        if (a.kadd != nullptr) kv = add4(kv, ld4(a.kadd + head * ATT_HD + 4 * c4));
        if (a.vadd != nullptr) vv = add4(vv, ld4(a.vadd + head * ATT_HD + 4 * c4));
      }
      st4(&ks[row * ATT_KLD + 4 * c4], kv);
      st4(&vs[row * ATT_HD + 4 * c4], vv);
    }
    __syncthreads();
    f32x16 sc;
#pragma unroll
    for (int e = 0; e < 16; ++e) sc[e] = 0.f;
    const float* kp = &ks[r * ATT_KLD + 64 * h];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float4 kf = ld4(kp + 4 * c);
      sc = mfma_f32(kf.x, qf[4 * c + 0], sc);
      sc = mfma_f32(kf.y, qf[4 * c + 1], sc);
      sc = mfma_f32(kf.z, qf[4 * c + 2], sc);
      sc = mfma_f32(kf.w, qf[4 * c + 3], sc);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = 32 * kt + mfma_row(e, h);
      const bool ok = kbits == nullptr ? key < nvalid : al_key_valid(kbits, key, S, a.lead);
      const float s = ok ? sc[e] : -INFINITY;
      sc[e] = s;
      mx = fmaxf(mx, s);
    }
    mx = fmaxf(mx, shfl_xor_f32(mx, 32));
    const float m_new = fmaxf(m_run, mx);
    if (m_new != -INFINITY) {
      const float c = (m_run == -INFINITY) ? 0.f : expf(m_run - m_new);
      float psum = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pe = expf(sc[e] - m_new);
        sc[e] = pe;
        psum += pe;
      }
      psum += shfl_xor_f32(psum, 32);
      l_run = l_run * c + psum;
      m_run = m_new;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[dt][e] *= c;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float* vp = &vs[mfma_row(e, h) * ATT_HD + r];
        const float pv = sc[e];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = mfma_f32(vp[32 * dt], pv, o[dt]);
      }
    }
  }
  if (q < Sq) {
    const float inv = 1.0f / l_run;
    const size_t obase = ((size_t)seq * Sq + q) * D + head * ATT_HD + 4 * h;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {   // rows mfma_row(4 g .. 4 g + 3, h) are 4 consecutive d's
        const float4 v = make_float4(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        const size_t oo = obase + 32 * dt + 8 * g;
        if (out != nullptr) st4(out + oo, v);
        if (oh != nullptr) split4_store(oh + oo, ol + oo, v);
      }
  }
}

}  // namespace mdm

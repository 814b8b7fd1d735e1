// selfattn_block.h -- in_proj + self-attention of one short sequence and head as ONE kernel (DiP decoder, split precision).
//
// Replaces, per nn.TransformerDecoderLayer (model/mdm.py:85-93; torch transformer.py _sa_block), the packed in_proj of
// F.multi_head_attention_forward and its scaled_dot_product_attention with the tgt_key_padding_mask of model/mdm.py:241-247, :263-265
//     qkv = norm3_prev(x) . Win^T + bin          (norm3 of the previous layer folded for l >= 1; plain for layer 0)
//     att = softmax(q k^T / sqrt(128) + mask) v  per head
// which rounds 4-5 ran as TWO dependent launches at DiP's 3,840 rows: gemm_x3s kind 0 (768 workgroups writing 35 MB of Q / K / V^T
// planes) and attention_x3_kernel<2> (512 workgroups reading them back): 25.8 + 14.2 us per layer (profiles/r05b_xattn_block.md).
// A DiP sequence is 20 + 40 = 60 tokens: ONE 64-row tile.  So a workgroup owns (sequence, head): it contracts the sequence's rows
// with the head's 3 x 128 in_proj columns (N = 384, K = D) and attends on chip -- Q, K and V^T never leave the CU:
//   * 256 workgroups at DiP's per-GPU batch (64 sequences x 4 heads): every CU busy, W bytes per MFMA as in a 64 x 128 tile
//     (768 KB of W per workgroup for 64 x 384 outputs); 4 waves, wave w owns one 32-column block of EACH of Q, K, V (d 32 w .. + 31);
//   * A (the sequence's operand planes) streams through two 16 KB LDS buffers in 64-k chunks (gemm_x3s.h's image and loop: LDS-DMA
//     pieces, counted vmcnt waits across a bare barrier, fragment reads one sub-step ahead); W straight to registers through a ring
//     of four sub-step slots refilled in place;
//   * the Q and K blocks are computed TRANSPOSED (acc = W . x^T: a lane holds a token's 4 consecutive d per register quad) and
//     written, folded / biased / scaled / split, into k-blocked fragment images with 8-byte LDS stores; the V block in standard
//     orientation (a lane holds a d's tokens in accumulator order) goes into the V^T image with 16-byte stores whose key order IS the
//     accumulator order (attention_x3.h: the probabilities are then the matching B operand as they are);
//   * attention as in attention_x3.h (St = K Q^T, exact softmax in registers with the per-key additive mask, Ot = V^T P^T): wave w
//     takes query tile w & 1 and the d half w >> 1 of the output (both waves of a query tile compute its 64 x 32 scores: 48 MFMAs);
//     the output goes to the attention planes [M][D] that out_proj reads, 8 bytes per lane.
// Sequences of more than 64 tokens keep the two-launch form (decoder.h).
//
// CROSS mode (MODE 2) -- the same decomposition for the cross-attention of the layer (model/mdm.py:263-265: multihead_attn(x, memory,
// memory) with x = norm1(y)): the workgroup of (sequence, head) projects ONLY the head's 128 query columns (norm1 folded; one W block
// per wave), fills the K and V^T images from the hoisted fp32 memory projections (+ the step's projected time row) of its sequence --
// at most 64 text tokens, masked by the prompt's token count -- and attends as above; the attention planes then feed the out_proj GEMM
// (gemm_x3s.h kind 2).  Measured against xattn_block.h's whole-block kernel (every workgroup re-reads all of Wq | Wo for 32 rows:
// W-delivery-bound on half the CUs, 35 us): profiles/r05c_seqhead_blocks.md.
#pragma once
#include "gemm_x3s.h"

namespace mdm {

struct SelfAttnArgs {
  X3Operand x;             // [M][D] operand planes of the layer input (pre-norm sum for FOLD)
  const float* xstat;      // FOLD: [M][stat_parts][2] partial row statistics (gemm_x3s.h OSTAT format, stat_cols columns each)
  X3Weights w;             // in_proj [3D][D], fragment-ordered planes (gamma-folded for FOLD)
  const float* bias;       // [3D] (folded for FOLD)
  const float* colsum;     // FOLD: [3D]
  float qscale;            // 1 / sqrt(128)
  const int* lengths;      // frame mask: counts / bitmaps (include/mdm_hip.h lengths_dev) or null
  int lead, B;             // keys in front of the frames that are never masked; samples (lengths index = sequence % B)
  p16_t* oh;               // [M][D] attention output planes
  p16_t* ol;
  int M, S, D, H;          // rows, tokens per sequence (<= 64), width, heads
  int stat_parts, stat_cols;
  float inv_dim, acc_scale;
  // CROSS mode: the projected text memory of the layer (xattn_block.h XattnArgs: rows (kseq * ntok + tok) * ldkv, head h at + 128 h)
  const float* mk;
  const float* mv;
  int ldkv;
  const float* kadd;       // [D] added to every key / value row (the step's projected time embedding) or null
  const float* vadd;
  const int* text_lengths; // [B] valid memory tokens
  int ntok, kv_B, kv_b0;
};

constexpr int SB_WAVES = 4, SB_TR = 64;
constexpr int SB_NSUB = 8;                                   // 16-deep sub-steps per A chunk (128 k: four rendezvous per 512 k; the 64-k
                                                             // chunks of the first version paid ~1.2 k cycles per boundary, r05c)
constexpr int SB_BUF = SB_NSUB * 32 * 2 * 64;                // one chunk of A: 4 k-blocks x (hi, lo) x 4 row groups x 1 KB = 32 KB
constexpr int SB_IMG = 32768;                                // Q, K: 64 tokens x 128 d; V^T: 128 d x 64 keys (hi | lo planes)
// The Q image ALIASES A buffer 0: it is written by the epilogue, when the contraction has retired (a barrier in between).  K and V^T
// have regions of their own: in CROSS mode they are filled from the memory BEFORE the contraction.
constexpr int SB_Q = 0, SB_K = 2 * SB_BUF, SB_V = SB_K + SB_IMG;
constexpr int SB_VEC = SB_V + SB_IMG;                        // bias[384] | colsum[384] of this head
constexpr int SB_TAB = SB_VEC + 2 * 384 * 4;                 // (mean, rstd) of the 64 rows
constexpr int SB_MASK = SB_TAB + SB_TR * 8;                  // additive key mask (0 / -inf) of the 64 keys
constexpr int SB_LDS = SB_MASK + SB_TR * 4;
#if defined(MDM_PROBES) && !defined(MDM_EMU)
// PROBE BUILD ONLY: wave 0's shader-clock stamps of ONE selected launch (mdm_debug_set(11, n): the n-th launch of this kernel after the
// call; mdm_debug_get(300000 + 8 * workgroup + i)): i = 0 kernel entry, 1 chunk 0 / vectors / mask visible (first barrier passed),
// 2 contraction retired, 3 fragment images complete, 4 attention arithmetic done, 5 last plane store issued.
constexpr int SB_TL_WGS = 1024;
__device__ unsigned long long g_sb_tl[8 * SB_TL_WGS];
__device__ int g_sb_tl_on;
#define SB_STAMP(i)                                                                                    \
  do {                                                                                                 \
    if (tl_on && tid == 0 && blockIdx.x < SB_TL_WGS) g_sb_tl[8 * blockIdx.x + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define SB_STAMP(i) do { } while (0)
#endif

// MODE 0: self-attention, plain in_proj (layer 0); 1: self-attention, norm3 of the previous layer folded; 2: CROSS (norm1 folded)
template <int MODE>
__global__ __launch_bounds__(64 * SB_WAVES, 1) void selfattn_block_kernel(SelfAttnArgs a, int total) {
  MDM_DYN_SMEM(unsigned char, lds);
  constexpr bool FOLD = MODE != 0, CROSS = MODE == 2;
  constexpr int NWB = CROSS ? 1 : 3;           // W blocks per wave: Q (and K, V of the same 32 d)
  constexpr int WD = 4;                        // W ring: four sub-step slots per wave (hi + lo fragment of each of its blocks)
  constexpr int LW = 2 * NWB;                  // W loads per wave and sub-step
  constexpr int PW = SB_NSUB * 2 / 2;          // LDS-DMA pieces (1 KB) per wave and chunk: 32 pieces over 4 waves
  static_assert(SB_NSUB > WD && SB_NSUB % WD == 0 && LW * (WD - 1) + PW <= 63 && LW * WD <= 63,
                "slot <-> sub-step map across chunks (gemm_x3s.h: NSUB > D); vmcnt range");

  const int tid = threadIdx.x, lane = tid & 63;
#if defined(MDM_PROBES) && !defined(MDM_EMU)
  const bool tl_on = g_sb_tl_on != 0;
#endif
  SB_STAMP(0);
  MDM_KERNARGS_NOW("s"(a.x.hi), "s"(a.x.lo), "s"(a.w.hi), "s"(a.w.lo), "s"(a.xstat), "s"(a.bias), "s"(a.colsum), "s"(a.H), "s"(a.D), "s"(a.M), "s"(a.S), "s"(total));
  const p16_t* const xh_p = rt_sgpr_ptr(a.x.hi);
  const p16_t* const xl_p = rt_sgpr_ptr(a.x.lo);
  const float* const bias_p = rt_sgpr_ptr(a.bias);
  const float* const colsum_p = rt_sgpr_ptr(a.colsum);
#ifdef MDM_EMU
  const int wid = tid >> 6;
#else
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int r = lane & 31, h = lane >> 5;
  const int lid = xcd_remap((int)blockIdx.x, total);
  const int seq = lid / a.H, head = lid - seq * a.H;     // the heads of a sequence are neighbours on one XCD: they share its rows
  const int D = a.D, M = a.M, S = a.S;
  const int m0 = seq * S;
  const int nchunks = D / (SB_NSUB * 16);
  const int nsub_total = D / 16;

  float* const vec = reinterpret_cast<float*>(lds + SB_VEC);
  float2* const stab = reinterpret_cast<float2*>(lds + SB_TAB);
  float* const kmask = reinterpret_cast<float*>(lds + SB_MASK);

  // ---- A stream (gemm_x3s.h issue_chunk, RT = 2): piece q = wid + 4 i: g = q % 4 (16-row group), p = (q / 4) % 2, ms = q / 8
  const int schunk = (lane & 3) ^ ((lane >> 4) & 3);
  auto issue_chunk = [&](int c, int buf) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int q = wid + SB_WAVES * i;
      const int g = q & 3, p = (q >> 2) & 1, ms = q >> 3;       // (32 pieces per chunk: ms = 0 .. 3)
      const int arow = min(m0 + g * 16 + (lane >> 2), M - 1);
      const p16_t* src = (p ? xl_p : xh_p) + (size_t)arow * D + (size_t)c * (SB_NSUB * 16) + ms * 32 + schunk * 8;
      glds16(src, lds + buf * SB_BUF + ((ms * 2 + p) * 4 + g) * 1024);
    }
  };
  // ---- W stream: this wave's block of Q (which = 0), K (1), V (2): packed rows which * D + head * 128 + 32 wid .. + 31
  uint32_t wbase[NWB];
#pragma unroll
  for (int wh = 0; wh < NWB; ++wh)
    wbase[wh] = (uint32_t)((wh * D + head * 128) / 32 + wid) * (uint32_t)nsub_total * 512u + (uint32_t)lane * 8u;
  p16x8 wsh[WD * NWB] = {}, wsl[WD * NWB] = {};       // slot d, block wh: [d * NWB + wh]
  auto issue_w = [&](auto slot_tag, int gj) __attribute__((always_inline)) {
    constexpr int sl = decltype(slot_tag)::value;
    const int gg = gj < nsub_total ? gj : gj - nsub_total;          // past the end: a harmless re-fetch keeps the wait counts uniform
#pragma unroll
    for (int wh = 0; wh < NWB; ++wh) {
      gload16_refill(wsh[sl * NWB + wh], a.w.hi + wbase[wh] + (uint32_t)gg * 512u);
      gload16_refill(wsl[sl * NWB + wh], a.w.lo + wbase[wh] + (uint32_t)gg * 512u);
    }
  };
  issue_chunk(0, 0);
  static_for<WD>([&](auto s_tag) __attribute__((always_inline)) { issue_w(s_tag, decltype(s_tag)::value); });
  // (the W prologue travels UNDER the compiler-tracked loads below -- hipcc retires those with vmcnt(0), i.e. behind the W slots;
  // issuing the W prologue behind them instead was measured and is worse: two dependent round trips in front of the first MFMA,
  // 8.5 k / 15.7 k cycles to the first rendezvous against 7.7 k / 10.8 k, profiles/r05c_seqhead_blocks.md section 4)

  // ---- this head's per-column vectors, the rows' (mean, rstd), the additive key mask
  for (int i = tid; i < (FOLD ? 2 : 1) * 32 * NWB; i += 64 * SB_WAVES) {
    const int which_vec = i / (32 * NWB), j = i - which_vec * (32 * NWB), wh = j / 32, c = (j - wh * 32) * 4;
    const float* src = (which_vec == 0 ? bias_p : colsum_p) + wh * D + head * 128 + c;
    st4(vec + which_vec * 384 + wh * 128 + c, ld4(src));
  }
  // 32-key tiles in use.  CROSS: by the memory's token count (DiP's 24 tokens: one -- the second is neither filled nor multiplied).
  // Self-attention: always both (a compile-time 2: as a run-time bound the branches cost 1.5 k cycles per launch, r05c section 4)
  const int nkt = CROSS ? (a.ntok + 31) / 32 : 2;
  int kseq = seq;                                   // CROSS: local (branch, sample) -> sequence of the memory projections
  if constexpr (CROSS) {
    const int br = seq / a.B, bl = seq - br * a.B;
    if (a.kv_B > 0) kseq = br * a.kv_B + a.kv_b0 + bl;
  }
  if (tid < SB_TR) {
    if constexpr (FOLD) {
      const int m = m0 + tid;
      float2 v = make_float2(0.f, 0.f);         // pad rows: (0, 0) -> every folded value is the finite constant b'
      if (tid < S && m < M) {
        const int np = a.stat_parts;            // gemm_x3s.h's merge of the row's partial statistics (<= 8)
        const float* q = a.xstat + (size_t)m * np * 2;
        float4 pp[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          pp[i] = zero4();
          if (2 * i + 1 < np && (np & 1) == 0) pp[i] = ld4(q + 4 * i);
          else {
            if (2 * i < np) { const float2 t = *reinterpret_cast<const float2*>(q + 4 * i); pp[i].x = t.x; pp[i].y = t.y; }
            if (2 * i + 1 < np) { const float2 t = *reinterpret_cast<const float2*>(q + 4 * i + 2); pp[i].z = t.x; pp[i].w = t.y; }
          }
        }
        const float cols = (float)a.stat_cols, icols = 1.0f / cols;
        const float mean = ((pp[0].x + pp[0].z) + (pp[1].x + pp[1].z) + ((pp[2].x + pp[2].z) + (pp[3].x + pp[3].z))) * a.inv_dim;
        float m2 = 0.f, dd = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d0 = 2 * i < np ? pp[i].x * icols - mean : 0.f, d1 = 2 * i + 1 < np ? pp[i].z * icols - mean : 0.f;
          m2 += pp[i].y + pp[i].w;
          dd += d0 * d0 + d1 * d1;
        }
        m2 += cols * dd;
        v = make_float2(mean, 1.0f / sqrtf(m2 * a.inv_dim + 1e-5f));
      }
      stab[tid] = v;
    }
    // key `tid`: the lead tokens are always valid; frame f = key - lead by the count or by its bitmap bit; keys >= S never
    // (CROSS: the keys are the memory tokens, valid up to the prompt's token count)
    bool ok = CROSS ? tid < min(a.ntok, a.text_lengths[seq % a.B]) : tid < S;
    if (!CROSS && a.lengths != nullptr && ok) {
      const int bl = seq % a.B, cnt = a.lengths[bl];
      if (cnt >= 0) ok = tid < a.lead + cnt;
      else {
        const uint32_t* kbits = reinterpret_cast<const uint32_t*>(a.lengths + a.B + 8 * bl);
        const int f = max(tid - a.lead, 0);
        ok = tid < a.lead || ((kbits[f >> 5] >> (f & 31)) & 1u);
      }
    }
    kmask[tid] = ok ? 0.f : -INFINITY;
  }

  if constexpr (CROSS) {
    // ---- K and V^T images from the fp32 memory projections of this sequence and head: key-major rows are read as float4
    // (coalesced: 32 lanes = one key's 128 d).  K goes in as (row = key, k = d) WITHOUT the step's time row: it would add the same
    // q . k_time to every key's score of a query, which the softmax removes.  V^T goes in as (row = d, k = key in the accumulator's
    // key order: position p of a 16-key group holds key (p & 3) + 8 ((p >> 2) & 1) + 4 (p >> 3), attention_x3.h) with the time row
    // added.  Keys >= ntok: zeros.
    const float* kb = a.mk + (size_t)kseq * a.ntok * a.ldkv + head * 128;
    const float* vb = a.mv + (size_t)kseq * a.ntok * a.ldkv + head * 128;
    // (Two other mappings of the V^T fill were measured and are slower, profiles/r05c section 4: one 16-byte chunk of four d rows per
    // thread from float4 loads -- lanes 256 bytes apart, 8-way bank conflicts: 15.5 k cycles to the first rendezvous; one d row per
    // lane from dword loads -- conflict-free stores, but four dependent load -> split -> store rounds: 25.4 k.  This one issues all
    // 16 float4 loads of a thread up front and pays 2-byte scattered stores: 10.8 k.)
    const float4 va = a.vadd != nullptr ? ld4(a.vadd + head * 128 + 4 * (tid & 31)) : zero4();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if ((i >> 2) >= nkt) continue;                   // keys 32 (i >> 2) .. + 31 of this round: a key tile nobody reads
      const int idx = tid + 64 * SB_WAVES * i, key = idx >> 5, d = 4 * (idx & 31);      // 64 keys x 32 float4
      float4 kv = zero4(), vv = zero4();
      if (key < a.ntok) {
        kv = ld4(kb + (size_t)key * a.ldkv + d);
        vv = add4(ld4(vb + (size_t)key * a.ldkv + d), va);
      }
      // K: token `key`, d .. d + 3: k-block d / 32, chunk (d % 32) / 8 swizzled by the row, half (d % 8) / 4
      unsigned char* kd = lds + SB_K + (((d >> 5) * 2) * 4) * 1024 + key * 64 + ((((d >> 3) & 3) ^ ((key >> 2) & 3)) * 16) + (d & 4) * 2;
      split4_store(reinterpret_cast<p16_t*>(kd), reinterpret_cast<p16_t*>(kd + 4 * 1024), kv);
      // V^T: rows d .. d + 3, key position `pos` inside its 16-key group
      const int kt = key >> 5, s2 = (key >> 4) & 1, k16 = key & 15;
      const int pos = (k16 & 3) + 4 * ((k16 >> 3) & 1) + 8 * ((k16 >> 2) & 1);
      const float v4[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int dr = d + e;
        unsigned char* vd = lds + SB_V + ((kt * 2) * 8) * 1024 + dr * 64 + (((2 * s2 + (pos >> 3)) ^ ((dr >> 2) & 3)) * 16) + (pos & 7) * 2;
        p16_t hi, lo;
        split_p16(v4[e], hi, lo);
        *reinterpret_cast<p16_t*>(vd) = hi;
        *reinterpret_cast<p16_t*>(vd + 8 * 1024) = lo;
      }
    }
  }

  f32x16 acc[NWB][2];                                 // block wh (Q, K transposed; V standard), row sub-tile t
#pragma unroll
  for (int wh = 0; wh < NWB; ++wh)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[wh][t][e] = 0.f;

  // fragment read addresses (gemm_x3s.h): row r of sub-tile t, 16-byte chunk (ks * 2 + h) ^ sw of its 64-byte row
  const int sw = (r >> 2) & 3;
  const uint32_t fr0 = (uint32_t)(r * 64 + ((h ^ sw) * 16)), fr1 = (uint32_t)(r * 64 + (((2 + h) ^ sw) * 16));
#ifndef MDM_EMU
  const uint32_t lds_base = lds_addr_of(lds);
#endif
  p16x8 fah[2][2], fal[2][2];      // fragments of sub-step j (set j & 1), row sub-tile t
  auto read_frags = [&](auto j_tag, int buf) __attribute__((always_inline)) {
    constexpr int j = decltype(j_tag)::value, ms = j / 2, ks = j % 2;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#ifdef MDM_EMU
      lds_read16(fah[j & 1][t], lds + buf * SB_BUF, (uint32_t)(((ms * 2 + 0) * 4) * 1024 + t * 2048) + (ks ? fr1 : fr0));
      lds_read16(fal[j & 1][t], lds + buf * SB_BUF, (uint32_t)(((ms * 2 + 1) * 4) * 1024 + t * 2048) + (ks ? fr1 : fr0));
#else
      constexpr uint32_t OH = (uint32_t)(((ms * 2 + 0) * 4) * 1024), OL = (uint32_t)(((ms * 2 + 1) * 4) * 1024);
      const uint32_t ad = lds_base + (uint32_t)buf * SB_BUF + (ks ? fr1 : fr0) + (uint32_t)t * 2048u;
      lds_read16<(int)OH>(fah[j & 1][t], ad);
      lds_read16<(int)OL>(fal[j & 1][t], ad);
#endif
    }
  };

  for (int c = 0; c < nchunks; ++c) {
    const int buf = c & 1;
    // chunk c landed (this wave's pieces): c == 0 -- the WD W sub-steps of the prologue are younger; c > 0 -- the counted W waits of
    // chunk c - 1 (sub-steps >= WD, all issued behind the pieces) have already retired them (gemm_x3s.h, NSUB > D)
    if (c == 0) vmem_wait<LW * WD>(wsh[0], wsl[0]);
    wg_barrier_nodrain();                 // every wave's pieces visible; every wave is past chunk c - 1, whose buffer refills now
    if (c == 0) SB_STAMP(1);
    issue_chunk(min(c + 1, nchunks - 1), buf ^ 1);      // (last chunk: a harmless re-fetch keeps the counts uniform)
    read_frags(std::integral_constant<int, 0>{}, buf);
    static_for<SB_NSUB>([&](auto j_tag) __attribute__((always_inline)) {
      constexpr int j = decltype(j_tag)::value, sl = j % WD;
      if constexpr (j + 1 < SB_NSUB) read_frags(std::integral_constant<int, j + 1>{}, buf);
      // W(c, j): younger = the WD - 1 sub-steps behind it (+ the next chunk's pieces when it was issued in front of them)
      constexpr int NW = LW * (WD - 1) + (j < WD ? PW : 0);
      if constexpr (CROSS) vmem_wait<NW>(wsh[sl], wsl[sl]);
      else vmem_wait<NW>(wsh[sl * 3], wsl[sl * 3], wsh[sl * 3 + 1], wsl[sl * 3 + 1], wsh[sl * 3 + 2], wsl[sl * 3 + 2]);
      lds_wait<(j + 1 < SB_NSUB) ? 4 : 0>(fah[j & 1][0], fal[j & 1][0], fah[j & 1][1], fal[j & 1][1]);
#ifndef MDM_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      // Q, K: acc = W . x^T (lane = token);  V: acc = x . W^T (lane = d).  Consecutive MFMAs on different accumulators.
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[0][t] = mfma_p16(wsh[sl * NWB + 0], fal[j & 1][t], acc[0][t]);
        if constexpr (!CROSS) {
          acc[1][t] = mfma_p16(wsh[sl * NWB + 1], fal[j & 1][t], acc[1][t]);
          acc[NWB - 1][t] = mfma_p16(fal[j & 1][t], wsh[sl * NWB + NWB - 1], acc[NWB - 1][t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[0][t] = mfma_p16(wsl[sl * NWB + 0], fah[j & 1][t], acc[0][t]);
        if constexpr (!CROSS) {
          acc[1][t] = mfma_p16(wsl[sl * NWB + 1], fah[j & 1][t], acc[1][t]);
          acc[NWB - 1][t] = mfma_p16(fah[j & 1][t], wsl[sl * NWB + NWB - 1], acc[NWB - 1][t]);
        }
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[0][t] = mfma_p16(wsh[sl * NWB + 0], fah[j & 1][t], acc[0][t]);
        if constexpr (!CROSS) {
          acc[1][t] = mfma_p16(wsh[sl * NWB + 1], fah[j & 1][t], acc[1][t]);
          acc[NWB - 1][t] = mfma_p16(fah[j & 1][t], wsh[sl * NWB + NWB - 1], acc[NWB - 1][t]);
        }
      }
#ifndef MDM_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      issue_w(std::integral_constant<int, sl>{}, c * SB_NSUB + j + WD);
    });
  }
  // the tail's re-fetches (W slots, the spare A buffer) land before registers / LDS are reused; the wait NAMES every slot register
  static_for<WD * NWB / 4>([&](auto q_tag) __attribute__((always_inline)) {
    constexpr int q = 4 * decltype(q_tag)::value;
    vmem_wait<0>(wsh[q], wsl[q], wsh[q + 1], wsl[q + 1], wsh[q + 2], wsl[q + 2], wsh[q + 3], wsl[q + 3]);
  });
  wg_barrier();             // every wave has retired its fragment reads and its last DMA: A buffer 0 becomes the Q image
  SB_STAMP(2);

  // ---- epilogue: fold / bias / scale / split -> the three fragment images
  const float accs = a.acc_scale;
  {
    // Q (wh = 0) and K (wh = 1): lane = token 32 t + r; register quad g = d 8 g + 4 h .. + 3 of the wave's 32-d block (k-block wid)
#pragma unroll
    for (int wh = 0; wh < (CROSS ? 1 : 2); ++wh) {
      unsigned char* img = lds + (wh == 0 ? SB_Q : SB_K);
      const float mult = wh == 0 ? a.qscale : 1.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float2 st = make_float2(0.f, 1.f);
        if constexpr (FOLD) st = stab[32 * t + r];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int n = wh * 128 + 32 * wid + 8 * g + 4 * h;
          const float4 bb = ld4(vec + n);
          float4 v4 = make_float4(acc[wh][t][4 * g + 0] * accs, acc[wh][t][4 * g + 1] * accs, acc[wh][t][4 * g + 2] * accs,
                                  acc[wh][t][4 * g + 3] * accs);
          if constexpr (FOLD) {
            const float4 cc = ld4(vec + 384 + n);
            v4.x = st.y * (v4.x - st.x * cc.x); v4.y = st.y * (v4.y - st.x * cc.y);
            v4.z = st.y * (v4.z - st.x * cc.z); v4.w = st.y * (v4.w - st.x * cc.w);
          }
          v4.x = (v4.x + bb.x) * mult; v4.y = (v4.y + bb.y) * mult; v4.z = (v4.z + bb.z) * mult; v4.w = (v4.w + bb.w) * mult;
          unsigned char* dst = img + ((wid * 2) * 4) * 1024 + t * 2048 + r * 64 + ((g ^ sw) * 16) + 8 * h;
          split4_store(reinterpret_cast<p16_t*>(dst), reinterpret_cast<p16_t*>(dst + 4 * 1024), v4);
        }
      }
    }
    // V (wh = 2): lane = d 32 wid + r; registers 8 s2 .. + 7 of sub-tile t = positions 8 h .. + 7 of 16-key group s2 of key tile t
    // (the accumulator's row order IS the V^T image's key order, attention_x3.h): one 16-byte store per plane
    if constexpr (!CROSS) {
    const float vb = vec[256 + 32 * wid + r];
    float vc = 0.f;
    if constexpr (FOLD) vc = vec[384 + 256 + 32 * wid + r];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if constexpr (FOLD) {
            const float2 st = stab[32 * t + mfma_row(8 * s2 + j, h)];
            vv[j] = st.y * (acc[NWB - 1][t][8 * s2 + j] * accs - st.x * vc) + vb;
          } else {
            vv[j] = acc[NWB - 1][t][8 * s2 + j] * accs + vb;
          }
        }
        p16x8 vh8, vl8;
        split8(vv, vh8, vl8);
        unsigned char* dst = lds + SB_V + ((t * 2) * 8) * 1024 + (32 * wid + r) * 64 + (((2 * s2 + h) ^ sw) * 16);
        *reinterpret_cast<p16x8*>(dst) = vh8;
        *reinterpret_cast<p16x8*>(dst + 8 * 1024) = vl8;
      }
    }   // !CROSS
  }
  wg_barrier();             // the three images (and the key mask) are complete
  SB_STAMP(3);

  // ================= attention: wave = (query tile qt, d half dh) =================
  const int qt = wid & 1, dh = wid >> 1;
  f32x16 p[2];
  {
    p16x8 qh[8], ql[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) {       // query 32 qt + r, d 16 st + 8 h .. + 7: k-block st / 2, sub-step st % 2
      const unsigned char* src = lds + SB_Q + (((st / 2) * 2) * 4) * 1024 + qt * 2048 + ((st & 1) ? fr1 : fr0);
      qh[st] = *reinterpret_cast<const p16x8*>(src);
      ql[st] = *reinterpret_cast<const p16x8*>(src + 4 * 1024);
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) p[kt][e] = 0.f;
      if (kt < nkt)
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const unsigned char* src = lds + SB_K + (((st / 2) * 2) * 4) * 1024 + kt * 2048 + ((st & 1) ? fr1 : fr0);
        const p16x8 kh = *reinterpret_cast<const p16x8*>(src), kl = *reinterpret_cast<const p16x8*>(src + 4 * 1024);
        p[kt] = mfma_p16(kl, qh[st], p[kt]);
        p[kt] = mfma_p16(kh, ql[st], p[kt]);
        p[kt] = mfma_p16(kh, qh[st], p[kt]);
      }
    }
  }
  // softmax over the keys: lane-local over its 16 keys per tile + one cross-half exchange (attention_x3.h)
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 m4 = ld4(kmask + 32 * kt + 8 * g + 4 * h);
      const float s0 = p[kt][4 * g] + m4.x, s1 = p[kt][4 * g + 1] + m4.y, s2 = p[kt][4 * g + 2] + m4.z, s3 = p[kt][4 * g + 3] + m4.w;
      p[kt][4 * g] = s0; p[kt][4 * g + 1] = s1; p[kt][4 * g + 2] = s2; p[kt][4 * g + 3] = s3;
      mx = fmaxf(mx, fmaxf(fmaxf(s0, s1), fmaxf(s2, s3)));
    }
  mx = fmaxf(mx, shfl_xor_f32(mx, 32));
  if constexpr (kSplitF16) mx -= 6.931471805599453f;      // probabilities as hi / lo of p * 2^10: cancels in 1 / sum
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
#ifdef MDM_EMU
      const float pe = expf(p[kt][e] - mx);
#else
      const float pe = __expf(p[kt][e] - mx);
#endif
      p[kt][e] = pe;
      sum += pe;
    }
  sum += shfl_xor_f32(sum, 32);
  const float inv = 1.0f / sum;

  // Ot[d][query] for the d blocks 2 dh, 2 dh + 1: A = V^T fragments (row = d, k = keys in accumulator order), B = split(P)
  f32x16 o[2];
#pragma unroll
  for (int dd = 0; dd < 2; ++dd)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dd][e] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
    if (kt < nkt)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = p[kt][8 * s2 + j];
      p16x8 ph, pl;
      split8(pv, ph, pl);
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) {
        const unsigned char* src = lds + SB_V + ((kt * 2) * 8) * 1024 + (2 * dh + dd) * 2048 + (s2 ? fr1 : fr0);
        const p16x8 vh = *reinterpret_cast<const p16x8*>(src), vl = *reinterpret_cast<const p16x8*>(src + 8 * 1024);
        o[dd] = mfma_p16(vl, ph, o[dd]);
        o[dd] = mfma_p16(vh, pl, o[dd]);
        o[dd] = mfma_p16(vh, ph, o[dd]);
      }
    }
  SB_STAMP(4);
  // normalised output -> the attention planes: query 32 qt + r, d 32 (2 dh + dd) + 8 g + 4 h .. + 3
  const int tok = 32 * qt + r;
  if (tok < S && m0 + tok < M) {
    const size_t obase = (size_t)(m0 + tok) * D + head * 128;
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const size_t oo = obase + 32 * (2 * dh + dd) + 8 * g + 4 * h;
        split4_store(a.oh + oo, a.ol + oo, make_float4(o[dd][4 * g + 0] * inv, o[dd][4 * g + 1] * inv, o[dd][4 * g + 2] * inv,
                                                       o[dd][4 * g + 3] * inv));
      }
  }
  SB_STAMP(5);
}

#ifndef MDM_X3_KERNEL_ONLY
#if defined(MDM_PROBES) && !defined(MDM_EMU)
inline int& sb_tl_target() { static int v = -1; return v; }   // mdm_debug_set(11, n); < 0: off
inline int& sb_tl_count() { static int v = 0; return v; }
#endif
inline bool selfattn_block_supported(int D, int S) { return S >= 1 && S <= SB_TR && D % 128 == 0 && D >= 128; }
inline bool crossattn_block_supported(int D, int S, int ntok) { return selfattn_block_supported(D, S) && ntok >= 1 && ntok <= SB_TR; }

template <int MODE>
inline int launch_seqhead_block_t(const SelfAttnArgs& a, hipStream_t stream) {
  auto kfn = &selfattn_block_kernel<MODE>;
  const int total = (a.M / a.S) * a.H;
#ifndef MDM_EMU
  {
    static bool configured[kMaxDevices] = {};
    if (const int rc = rt_dyn_lds_once(kfn, SB_LDS, configured, stream)) return rc;
  }
#endif
#if defined(MDM_PROBES) && !defined(MDM_EMU)
  if (sb_tl_target() >= 0) {   // timeline probe: stamps on for exactly the selected launch (stream-ordered switch)
    static int on_v[2] = {0, 1};
    const int on = (sb_tl_count()++ == sb_tl_target()) ? 1 : 0;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(g_sb_tl_on), &on_v[on], sizeof(int), 0, hipMemcpyHostToDevice, stream) != hipSuccess) return -1;
  }
#endif
  MDM_LAUNCH(kfn, dim3(total), dim3(64 * SB_WAVES), SB_LDS, stream, a, total);
  return 0;
}

// -1: hipFuncSetAttribute failed; -2: unsupported shape (callers check *_supported first).  mode: 0 self-attention with a plain
// in_proj, 1 self-attention with the previous LayerNorm folded, 2 cross-attention over the memory projections
inline int launch_seqhead_block(const SelfAttnArgs& a, int mode, hipStream_t stream) {
  if (!selfattn_block_supported(a.D, a.S) || a.M % a.S != 0 || a.H * 128 != a.D) return -2;
  if (mode == 2 && (a.ntok < 1 || a.ntok > SB_TR)) return -2;
  if (mode == 0) return launch_seqhead_block_t<0>(a, stream);
  if (mode == 1) return launch_seqhead_block_t<1>(a, stream);
  return launch_seqhead_block_t<2>(a, stream);
}
#endif

}  // namespace mdm

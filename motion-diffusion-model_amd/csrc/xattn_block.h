// xattn_block.h -- the cross-attention block of one nn.TransformerDecoderLayer as ONE kernel (DiP, split precision).
//
// Replaces, per decoder layer, the middle third of model/mdm.py:85-93's nn.TransformerDecoderLayer (torch transformer.py
// _mha_block under norm_first=False):   x = norm2(x + multihead_attn(x, memory, memory))   with x = norm1(y) on the way in, i.e.
//     q   = norm1(y) . Wq^T + bq                    (norm1 folded: Wq' = Wq diag(gamma1), rstd (Wq'.y - mean colsum) + bq')
//     att = softmax(q K^T / sqrt(128) + memory_key_padding_mask) V        per head; K | V = the projected text memory
//     x'  = att . Wo^T + bo + norm1(y)              (pre-norm2 sum, written as operand planes + row statistics)
// which round 4 ran as THREE dependent launches on 3,840 rows (decoder.h decoder_layers_planes: gemm_x3s kind 4 -> fp32 q,
// attention_f32_kernel<1> at 7 % matrix-pipe duty, gemm_x3s kind 2): 10.7 + 16.1 + 14.8 us per layer, of which ~5 us per launch are
// entry -> first MFMA and drain (profiles/r04j_x3s_timeline.md).  Everything between the two GEMMs is row-local given the
// sequence's memory, so one workgroup carries a 32-row tile of ONE sequence through all three stages:
//   * tile = 32 token rows of one sequence (group_rows = S: the memory is tile-uniform) x ALL D columns; 4 waves, wave w owns
//     columns [w D/4, (w+1) D/4) of both GEMMs = head w's 128 d (D = 512: NCB = 4 column blocks per wave);
//   * the tile's y rows (hi | lo planes, 32 x D) arrive ONCE by LDS-DMA as the k-blocked, XOR-swizzled fragment image of gemm_x3s.h
//     and stay resident: A operand of the q projection AND the plane residual of the epilogue (64 KB at D = 512);
//   * GEMM 1 is computed TRANSPOSED (acc = Wq . y^T: a lane holds one row's 4 consecutive columns per register quad), so the
//     folded-LayerNorm epilogue writes q, pre-scaled and split, straight into a second image with 8-byte LDS stores: no patch
//     round trips; wave w's columns are head w's queries;
//   * attention per wave = per head, computed transposed like attention_x3.h (St = K Q^T, exact softmax in registers, Ot = V^T P^T):
//     the hoisted fp32 memory K | V (+ the step's projected time row) is split into fragments on the fly from global memory -- 24
//     tokens x 128 d per head; the normalised output overwrites the wave's own q columns of the image (the A operand of GEMM 2);
//   * GEMM 2 (transposed too) reads that image; its epilogue is gemm_x3s.h's kind 2 without the patch round trips (residual =
//     LayerNorm rebuilt from the resident y image + the row statistics, output planes by 8-byte stores, partial statistics per 128
//     columns in the producer format every consumer already merges);
//   * W (fragment-ordered hi | lo planes of Wq' then Wo) streams to registers through ONE ring of two sub-step slots that runs
//     through both GEMMs (Wo's first sub-steps are fetched behind Wq's last), refilled in place, retired by counted vmcnt
//     waits that name the slot registers (common.h gload16_refill: the hazard class of profiles/r03b_pipe_determinism.md); it is
//     drained once, in front of the attention phase, so that no slot is in flight while that phase's register pressure may move it.
// Cost model at DiP's per-GPU shape (2 x 32 sequences x 60 tokens, D = 512): 128 workgroups; per workgroup 2 MB of W through the
// 64 B/clk vector-memory path = 15.6 us (the bound: 32 rows per W fragment), 768 MFMAs per wave = 11.7 us of matrix pipe.
#pragma once
#include "gemm_x3s.h"

namespace mdm {

struct XattnArgs {
  X3Operand y;            // [M][D] planes of the pre-norm1 sum
  const float* ystat;     // [M][D/128][2] its partial row statistics (gemm_x3s.h OSTAT format)
  X3Weights wq;           // Wq' = Wq diag(gamma1), fragment-ordered planes [D][D]
  const float* cq;        // [D] column sums of Wq'
  const float* bq;        // [D] bq + Wq beta1
  float qscale;           // 1 / sqrt(128)
  const float* k;         // projected memory keys: row (kseq * ntok + tok) * ldkv, head h at + h * 128
  const float* v;
  int ldkv;
  const float* kadd;      // [D] added to every key row (the step's projected time embedding) or null
  const float* vadd;
  const int* text_lengths;   // [B] valid memory tokens (memory_key_padding_mask as counts: the tokenizer pads on the right)
  int ntok, B, kv_B, kv_b0;
  X3Weights wo;           // cross-attention out_proj, fragment-ordered planes [D][D]
  const float* bo;        // [D]
  const float* gamma;     // [D] norm1.weight / bias: the residual is norm1(y)
  const float* beta;
  p16_t* oh;              // [M][D] planes of x' (the pre-norm2 sum)
  p16_t* ol;
  float* ostat;           // [M][D/128][2]
  int M, S;               // rows; tokens per sequence (tiles never straddle sequences)
  float inv_dim, acc_scale;
};

constexpr int XB_WAVES = 4, XB_TR = 32;
#if defined(MDM_PROBES) && !defined(MDM_EMU)
// PROBE BUILD ONLY: wave 0's shader-clock stamps of ONE selected launch (mdm_debug_set(10, n): the n-th xattn_block launch after the
// call; mdm_debug_get(200000 + 8 * workgroup + i)): i = 0 kernel entry, 1 y image / vectors / table visible, 2 GEMM 1 retired,
// 3 q image complete, 4 attention image complete, 5 GEMM 2 retired, 6 last plane store issued, 7 statistics written.
constexpr int XB_TL_WGS = 1024;
__device__ unsigned long long g_xb_tl[8 * XB_TL_WGS];
__device__ int g_xb_tl_on;
#define XB_STAMP(i)                                                                                    \
  do {                                                                                                 \
    if (tl_on && tid == 0 && blockIdx.x < XB_TL_WGS) g_xb_tl[8 * blockIdx.x + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define XB_STAMP(i) do { } while (0)
#endif
constexpr int xb_img_bytes(int ncb) { return ncb * 128 * 128; }               // 32 rows x D x hi|lo x 2 B, D = 128 ncb
constexpr int xb_vec_base(int ncb) { return 2 * xb_img_bytes(ncb); }          // five [D] fp32 vectors
constexpr int xb_tab_base(int ncb) { return xb_vec_base(ncb) + 5 * 128 * ncb * 4; }
constexpr int xb_part_base(int ncb) { return xb_tab_base(ncb) + XB_TR * 8; }
constexpr int xb_lds_bytes(int ncb) { return xb_part_base(ncb) + XB_WAVES * ncb * XB_TR * 8; }

// NCB = D / 128 (column blocks per wave = heads): 2 or 4.  NKT = 32-key tiles of the memory (ntok <= 32 NKT).
template <int NCB, int NKT>
__global__ __launch_bounds__(64 * XB_WAVES, 1) void xattn_block_kernel(XattnArgs a, int tiles_per_group, int total) {
  MDM_DYN_SMEM(unsigned char, lds);
  constexpr int D = 128 * NCB, H = NCB, KB = D / 32, NSUBT = D / 16;
  constexpr int IMG = xb_img_bytes(NCB);
  constexpr int WD = 4;                       // W ring: four sub-step slots per wave (hi + lo fragment of each of its NCB blocks: 32 KB
                                              // in flight per wave at D = 512 -- two slots covered 0.37 us of matrix work, less than an L2
                                              // round trip under load: 36 us per launch, profiles/r05b_xattn_block.md)
  constexpr int LW = 2 * NCB;                 // W loads per wave and sub-step
  static_assert(NCB == 2 || NCB == 4, "latent_dim 256 or 512");
  static_assert(NSUBT % 8 == 0 && 8 % WD == 0 && LW * WD <= 63, "chunks of eight sub-steps; slot <-> sub-step map; vmcnt range");

  const int tid = threadIdx.x, lane = tid & 63;
#if defined(MDM_PROBES) && !defined(MDM_EMU)
  const bool tl_on = g_xb_tl_on != 0;
#endif
  XB_STAMP(0);
#ifdef MDM_EMU
  const int wid = tid >> 6;
#else
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int r = lane & 31, h = lane >> 5;
  const int lid = xcd_remap((int)blockIdx.x, total);
  const int grp = lid / tiles_per_group, tig = lid - grp * tiles_per_group;
  const int m0 = grp * a.S + tig * XB_TR;
  const int rows_valid = min(XB_TR, a.S - tig * XB_TR);
  const int M = a.M;

  unsigned char* const yimg = lds;
  unsigned char* const qimg = lds + IMG;
  float* const vec = reinterpret_cast<float*>(lds + xb_vec_base(NCB));     // cq | bq | bo | gamma | beta
  float2* const stab = reinterpret_cast<float2*>(lds + xb_tab_base(NCB));

  // ---- the tile's y rows: KB k-blocks x (hi, lo) x two 16-row groups, 1 KB each (gemm_x3s.h's image: lane -> (row = lane >> 2,
  // stored 16-byte chunk = lane & 3) fetches the logical chunk (lane & 3) ^ ((row >> 2) & 3)); rows past the matrix are clamped
  {
    const int schunk = (lane & 3) ^ ((lane >> 4) & 3);
#pragma unroll
    for (int i = 0; i < KB; ++i) {            // KB * 4 pieces over 4 waves
      const int q = wid + XB_WAVES * i;
      const int g = q & 1, p = (q >> 1) & 1, ms = q >> 2;
      const int arow = min(m0 + g * 16 + (lane >> 2), M - 1);
      const p16_t* src = (p ? a.y.lo : a.y.hi) + (size_t)arow * D + ms * 32 + schunk * 8;
      glds16(src, yimg + ((ms * 2 + p) * 2 + g) * 1024);
    }
  }
  // ---- W stream: sub-step gj of the CONCATENATED contraction (gj < NSUBT: Wq', else Wo), this wave's NCB column blocks
  uint32_t wbase[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) wbase[cb] = (uint32_t)(wid * NCB + cb) * (uint32_t)NSUBT * 512u + (uint32_t)lane * 8u;
  p16x8 wsh[WD * NCB] = {}, wsl[WD * NCB] = {};     // slot d, block cb: [d * NCB + cb]  (zero: the first refill formally reads its slot)
  auto issue_w = [&](auto slot_tag, int gj) __attribute__((always_inline)) {
    constexpr int sl = decltype(slot_tag)::value;
    const int g2 = gj < 2 * NSUBT ? gj : gj - 2 * NSUBT;       // past the end: a harmless re-fetch keeps the wait counts uniform
    const bool second = g2 >= NSUBT;
    const p16_t* wh = second ? a.wo.hi : a.wq.hi;
    const p16_t* wl = second ? a.wo.lo : a.wq.lo;
    const uint32_t gg = (uint32_t)(second ? g2 - NSUBT : g2) * 512u;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      gload16_refill(wsh[sl * NCB + cb], wh + wbase[cb] + gg);
      gload16_refill(wsl[sl * NCB + cb], wl + wbase[cb] + gg);
    }
  };
  static_for<WD>([&](auto s_tag) __attribute__((always_inline)) { issue_w(s_tag, decltype(s_tag)::value); });

  // ---- per-column vectors -> LDS (read as float4 broadcasts by both epilogues), (mean, rstd) of the tile's rows
  for (int i = tid; i < 5 * D / 4; i += 64 * XB_WAVES) {
    const int which = i / (D / 4), c = (i - which * (D / 4)) * 4;
    const float* src = which == 0 ? a.cq : which == 1 ? a.bq : which == 2 ? a.bo : which == 3 ? a.gamma : a.beta;
    st4(vec + which * D + c, ld4(src + c));
  }
  if (tid < XB_TR) {
    const int m = m0 + tid;
    float2 v = make_float2(0.f, 0.f);         // pad rows: (0, 0) -> every folded value is a finite constant
    if (tid < rows_valid && m < M) {
      constexpr int np = D / 128;             // partials per row: one per 128 columns (gemm_x3s.h OSTAT), 2 or 4
      const float* q = a.ystat + (size_t)m * np * 2;
      const float4 p01 = ld4(q), p23 = np > 2 ? ld4(q + 4) : zero4();
      const float cols = 128.f, icols = 1.0f / 128.f;
      const float mean = ((p01.x + p01.z) + (p23.x + p23.z)) * a.inv_dim;
      const float d0 = p01.x * icols - mean, d1 = p01.z * icols - mean;
      const float d2 = np > 2 ? p23.x * icols - mean : 0.f, d3 = np > 2 ? p23.z * icols - mean : 0.f;
      const float m2 = (p01.y + p01.w) + (p23.y + p23.w) + cols * ((d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3));
      v = make_float2(mean, 1.0f / sqrtf(m2 * a.inv_dim + 1e-5f));
    }
    stab[tid] = v;
  }
  wait_vmem_all();          // this wave's pieces of the y image (and, as it happens, its first W slots)
  wg_barrier();             // image, vectors and table visible to every wave
  XB_STAMP(1);

  // ---- fragment reads of the k-blocked images: row r, 16-byte chunk (ks * 2 + h) ^ sw of its 64-byte row
  const int sw = (r >> 2) & 3;
  const uint32_t fr0 = (uint32_t)(r * 64 + ((h ^ sw) * 16)), fr1 = (uint32_t)(r * 64 + (((2 + h) ^ sw) * 16));
#ifndef MDM_EMU
  const uint32_t lds_base = lds_addr_of(lds);
#endif
  p16x8 fah[2], fal[2];
  // sub-step j (0..7) of chunk c (eight sub-steps = four k-blocks = 16 KB of image) of the image at byte offset `img_off`
  auto read_frags = [&](auto j_tag, int img_off, int c) __attribute__((always_inline)) {
    constexpr int j = decltype(j_tag)::value, ms = j / 2, ks = j % 2;
    constexpr uint32_t OH = (uint32_t)((ms * 2 + 0) * 2 * 1024), OL = (uint32_t)((ms * 2 + 1) * 2 * 1024);
#ifdef MDM_EMU
    lds_read16(fah[j & 1], lds + img_off + c * 16384, OH + (ks ? fr1 : fr0));
    lds_read16(fal[j & 1], lds + img_off + c * 16384, OL + (ks ? fr1 : fr0));
#else
    const uint32_t ad = lds_base + (uint32_t)img_off + (uint32_t)c * 16384u + (ks ? fr1 : fr0);
    lds_read16<(int)OH>(fah[j & 1], ad);
    lds_read16<(int)OL>(fal[j & 1], ad);
#endif
  };
  auto wait_w = [&](auto slot_tag) __attribute__((always_inline)) {      // slot retired: younger = the other slot's loads
    constexpr int sl = decltype(slot_tag)::value;
    if constexpr (NCB == 2) vmem_wait<LW*(WD - 1)>(wsh[sl * 2], wsl[sl * 2], wsh[sl * 2 + 1], wsl[sl * 2 + 1]);
    else vmem_wait<LW*(WD - 1)>(wsh[sl * 4], wsl[sl * 4], wsh[sl * 4 + 1], wsl[sl * 4 + 1], wsh[sl * 4 + 2], wsl[sl * 4 + 2],
                                 wsh[sl * 4 + 3], wsl[sl * 4 + 3]);
  };

  f32x16 acc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;

  // ================= GEMM 1, transposed: acc[cb][e] = q[row r][column nb + mfma_row(e, h)] (before the fold) =================
  for (int c = 0; c < NSUBT / 8; ++c) {
    read_frags(std::integral_constant<int, 0>{}, 0, c);
    static_for<8>([&](auto j_tag) __attribute__((always_inline)) {
      constexpr int j = decltype(j_tag)::value, sl = j % WD;
      if constexpr (j + 1 < 8) read_frags(std::integral_constant<int, j + 1>{}, 0, c);
      wait_w(std::integral_constant<int, sl>{});
      lds_wait<(j + 1 < 8) ? 2 : 0>(fah[j & 1], fal[j & 1]);
#ifndef MDM_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_p16(wsh[sl * NCB + cb], fal[j & 1], acc[cb]);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_p16(wsl[sl * NCB + cb], fah[j & 1], acc[cb]);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_p16(wsh[sl * NCB + cb], fah[j & 1], acc[cb]);
#ifndef MDM_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      issue_w(std::integral_constant<int, sl>{}, c * 8 + j + WD);      // (the last WD refills are Wo's first sub-steps)
    });
  }
  // Wo's first sub-steps (the loop's last refills) have LANDED before the attention phase: a slot register that hipcc spills
  // or copies under that phase's register pressure must hold data, not a load in flight (to hipcc the refill wrote it already)
  static_for<WD * NCB / 4>([&](auto q_tag) __attribute__((always_inline)) {
    constexpr int q = 4 * decltype(q_tag)::value;
    vmem_wait<0>(wsh[q], wsl[q], wsh[q + 1], wsl[q + 1], wsh[q + 2], wsl[q + 2], wsh[q + 3], wsl[q + 3]);
  });
  XB_STAMP(2);
  // ---- fold + scale + split -> q image: element (row r, k = n): k-block n / 32, chunk (n % 32) / 8 (swizzled), 8 bytes per quad
  const float accs = a.acc_scale;
  const uint32_t wr_lane = (uint32_t)(r * 64 + 8 * h);
  {
    const float2 st = stab[r];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int nb = (wid * NCB + cb) * 32;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 cc = ld4(vec + nb + 8 * g + 4 * h), bb = ld4(vec + D + nb + 8 * g + 4 * h);
        float4 v4;
        v4.x = (st.y * (acc[cb][4 * g + 0] * accs - st.x * cc.x) + bb.x) * a.qscale;
        v4.y = (st.y * (acc[cb][4 * g + 1] * accs - st.x * cc.y) + bb.y) * a.qscale;
        v4.z = (st.y * (acc[cb][4 * g + 2] * accs - st.x * cc.z) + bb.z) * a.qscale;
        v4.w = (st.y * (acc[cb][4 * g + 3] * accs - st.x * cc.w) + bb.w) * a.qscale;
        unsigned char* dst = qimg + ((wid * NCB + cb) * 2 * 2) * 1024 + wr_lane + ((g ^ sw) * 16);
        split4_store(reinterpret_cast<p16_t*>(dst), reinterpret_cast<p16_t*>(dst + 2 * 1024), v4);
      }
    }
  }
  wg_barrier();             // q image complete (D = 256: a head's columns come from two waves)
  XB_STAMP(3);

  // ================= attention over the memory, wave = head (transposed like attention_x3.h) =================
  if (wid < H) {
    const int hd = wid;
    const int seq = grp;                                  // the tile's sequence; local (branch, sample) -> memory sequence
    const int br = seq / a.B, bl = seq - br * a.B;
    const int kseq = a.kv_B > 0 ? br * a.kv_B + a.kv_b0 + bl : seq;
    const int ntok = a.ntok;
    const int nvalid = min(ntok, a.text_lengths[bl]);
    const float* kbase = a.k + (size_t)kseq * ntok * a.ldkv + hd * 128;
    const float* vbase = a.v + (size_t)kseq * ntok * a.ldkv + hd * 128;
    f32x16 p[NKT];
    {
      p16x8 qh[8], ql[8];
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const unsigned char* src = qimg + (((hd * 4 + st / 2) * 2) * 2) * 1024 + ((st & 1) ? fr1 : fr0);
        qh[st] = *reinterpret_cast<const p16x8*>(src);
        ql[st] = *reinterpret_cast<const p16x8*>(src + 2 * 1024);
      }
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) p[kt][e] = 0.f;
        const float* krow = kbase + (size_t)min(32 * kt + r, ntok - 1) * a.ldkv + 8 * h;     // pad keys: the last real row (finite)
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          float4 k0 = ld4(krow + 16 * st), k1 = ld4(krow + 16 * st + 4);
          if (a.kadd != nullptr) {
            k0 = add4(k0, ld4(a.kadd + hd * 128 + 16 * st + 8 * h));
            k1 = add4(k1, ld4(a.kadd + hd * 128 + 16 * st + 8 * h + 4));
          }
          const float kv8[8] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w};
          p16x8 kh, kl;
          split8(kv8, kh, kl);
          p[kt] = mfma_p16(kl, qh[st], p[kt]);
          p[kt] = mfma_p16(kh, ql[st], p[kt]);
          p[kt] = mfma_p16(kh, qh[st], p[kt]);
        }
      }
    }
    // softmax over the memory tokens: lane-local over its 16 keys per tile + one cross-half exchange; 1 / sum goes into the output
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float s = (32 * kt + mfma_row(e, h) < nvalid) ? p[kt][e] : -INFINITY;
        p[kt][e] = s;
        mx = fmaxf(mx, s);
      }
    mx = fmaxf(mx, shfl_xor_f32(mx, 32));
    if constexpr (kSplitF16) mx -= 6.931471805599453f;    // probabilities as hi / lo of p * 2^10 (attention_x3.h): cancels in 1 / sum
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
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

    // Ot[d][query] = V^T . P^T: the A fragment of (key tile kt, k-step s2, d block dt) is gathered from the fp32 memory in the
    // accumulator's key order -- slot 8 h + j of the step is key 16 s2 + 4 h + (j & 3) + 8 (j >> 2) (attention_x3.h ax_key_of_pos)
    f32x16 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
    float va[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) va[dt] = a.vadd != nullptr ? a.vadd[hd * 128 + 32 * dt + r] : 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        float pv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) pv[j] = p[kt][8 * s2 + j];
        p16x8 ph, pl;
        split8(pv, ph, pl);
        size_t koff[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          koff[j] = (size_t)min(32 * kt + 16 * s2 + 4 * h + (j & 3) + 8 * (j >> 2), ntok - 1) * a.ldkv + r;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          float vv[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) vv[j] = vbase[koff[j] + 32 * dt] + va[dt];
          p16x8 vh, vl;
          split8(vv, vh, vl);
          o[dt] = mfma_p16(vl, ph, o[dt]);
          o[dt] = mfma_p16(vh, pl, o[dt]);
          o[dt] = mfma_p16(vh, ph, o[dt]);
        }
      }
    // normalised output -> the head's own columns of the image (its q is dead): the A operand of the out_proj contraction
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 v4 = make_float4(o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        unsigned char* dst = qimg + ((hd * 4 + dt) * 2 * 2) * 1024 + wr_lane + ((g ^ sw) * 16);
        split4_store(reinterpret_cast<p16_t*>(dst), reinterpret_cast<p16_t*>(dst + 2 * 1024), v4);
      }
  }
  wg_barrier();             // attention image complete
  XB_STAMP(4);

  // ================= GEMM 2, transposed like GEMM 1: acc[cb][e] = x'[row r][column nb + mfma_row(e, h)] =================
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[cb][e] = 0.f;
  for (int c = 0; c < NSUBT / 8; ++c) {
    read_frags(std::integral_constant<int, 0>{}, IMG, c);
    static_for<8>([&](auto j_tag) __attribute__((always_inline)) {
      constexpr int j = decltype(j_tag)::value, sl = j % WD;
      if constexpr (j + 1 < 8) read_frags(std::integral_constant<int, j + 1>{}, IMG, c);
      wait_w(std::integral_constant<int, sl>{});
      lds_wait<(j + 1 < 8) ? 2 : 0>(fah[j & 1], fal[j & 1]);
#ifndef MDM_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_p16(wsh[sl * NCB + cb], fal[j & 1], acc[cb]);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_p16(wsl[sl * NCB + cb], fah[j & 1], acc[cb]);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = mfma_p16(wsh[sl * NCB + cb], fah[j & 1], acc[cb]);
#ifndef MDM_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      issue_w(std::integral_constant<int, sl>{}, NSUBT + c * 8 + j + WD);
    });
  }
  // the tail's re-fetches land before the registers are reused -- and the wait NAMES every slot register (gemm_x3s.h)
  static_for<WD * NCB / 4>([&](auto q_tag) __attribute__((always_inline)) {
    constexpr int q = 4 * decltype(q_tag)::value;
    vmem_wait<0>(wsh[q], wsl[q], wsh[q + 1], wsl[q + 1], wsh[q + 2], wsl[q + 2], wsh[q + 3], wsl[q + 3]);
  });
  XB_STAMP(5);

  // ---- epilogue: a lane holds row r's columns nb + 8 g + 4 h .. + 3 per register quad (no patch round trips): bias + the
  // LayerNorm(y) residual rebuilt from the resident y image (the same 8-byte slots the q image is written through) -> x', its
  // partial statistics per 32-column block (sum, then centred squares about the block mean: lane-local + one cross-half exchange),
  // the planes by 8-byte stores
  float2* const part_all = reinterpret_cast<float2*>(lds + xb_part_base(NCB));     // [block][row] partials
  {
    const float2 st = stab[r];
    const int m = m0 + r;
    const bool row_ok = r < rows_valid && m < M;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int blk = wid * NCB + cb, nb = blk * 32;
      float s1 = 0.f;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = nb + 8 * g + 4 * h;
        const float4 bb = ld4(vec + 2 * D + n), gg = ld4(vec + 3 * D + n), be = ld4(vec + 4 * D + n);
        const unsigned char* ysrc = yimg + (blk * 2 * 2) * 1024 + wr_lane + ((g ^ sw) * 16);
        const uint2 ya = *reinterpret_cast<const uint2*>(ysrc), yb = *reinterpret_cast<const uint2*>(ysrc + 2 * 1024);
        const float y0 = p16_to_f32((p16_t)(ya.x & 0xffffu)) + p16_to_f32((p16_t)(yb.x & 0xffffu));
        const float y1 = p16_to_f32((p16_t)(ya.x >> 16)) + p16_to_f32((p16_t)(yb.x >> 16));
        const float y2 = p16_to_f32((p16_t)(ya.y & 0xffffu)) + p16_to_f32((p16_t)(yb.y & 0xffffu));
        const float y3 = p16_to_f32((p16_t)(ya.y >> 16)) + p16_to_f32((p16_t)(yb.y >> 16));
        const float v0 = (acc[cb][4 * g + 0] * accs + bb.x) + ((y0 - st.x) * st.y * gg.x + be.x);
        const float v1 = (acc[cb][4 * g + 1] * accs + bb.y) + ((y1 - st.x) * st.y * gg.y + be.y);
        const float v2 = (acc[cb][4 * g + 2] * accs + bb.z) + ((y2 - st.x) * st.y * gg.z + be.z);
        const float v3 = (acc[cb][4 * g + 3] * accs + bb.w) + ((y3 - st.x) * st.y * gg.w + be.w);
        acc[cb][4 * g + 0] = v0; acc[cb][4 * g + 1] = v1; acc[cb][4 * g + 2] = v2; acc[cb][4 * g + 3] = v3;
        s1 += (v0 + v1) + (v2 + v3);
        if (row_ok) {
          const size_t o = (size_t)m * D + n;
          split4_store(a.oh + o, a.ol + o, make_float4(v0, v1, v2, v3));
        }
      }
      s1 += shfl_xor_f32(s1, 32);
      const float mw = s1 * (1.0f / 32.0f);
      float m2 = 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) { const float d = acc[cb][e] - mw; m2 += d * d; }
      m2 += shfl_xor_f32(m2, 32);
      if (h == 0) part_all[blk * XB_TR + r] = make_float2(s1, m2);
    }
  }
  XB_STAMP(6);
  wg_barrier();
  // rows x blocks partials -> one (sum, M2) pair per row and 128 columns: the producer format of gemm_x3s.h (four blocks, in order)
  for (int i = tid; i < XB_TR * (D / 128); i += 64 * XB_WAVES) {
    const int row = i % XB_TR, q = i / XB_TR;
    if (row < rows_valid && m0 + row < M) {
      float s1 = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) s1 += part_all[(4 * q + w4) * XB_TR + row].x;
      const float mt = s1 * (1.0f / 128.f);
      float m2 = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        const float2 v = part_all[(4 * q + w4) * XB_TR + row];
        const float dm = v.x * (1.0f / 32.0f) - mt;
        m2 += v.y + 32.0f * dm * dm;
      }
      *reinterpret_cast<float2*>(a.ostat + ((size_t)(m0 + row) * (D / 128) + q) * 2) = make_float2(s1, m2);
    }
  }
  XB_STAMP(7);
}

#ifndef MDM_X3_KERNEL_ONLY
#if defined(MDM_PROBES) && !defined(MDM_EMU)
inline int& xb_tl_target() { static int v = -1; return v; }   // mdm_debug_set(10, n); < 0: off
inline int& xb_tl_count() { static int v = 0; return v; }
#endif
inline bool xattn_block_supported(int D, int ntok) { return (D == 256 || D == 512) && ntok >= 1 && ntok <= 96; }

template <int NCB, int NKT>
inline int launch_xattn_block_t(const XattnArgs& a, hipStream_t stream) {
  auto kfn = &xattn_block_kernel<NCB, NKT>;
  constexpr int LDS = xb_lds_bytes(NCB);
#ifndef MDM_EMU
  if (LDS > 65536) {
    static bool configured[kMaxDevices] = {};
    if (const int rc = rt_dyn_lds_once(kfn, LDS, configured, stream)) return rc;
  }
#endif
  const int tpg = (a.S + XB_TR - 1) / XB_TR, total = (a.M / a.S) * tpg;
#if defined(MDM_PROBES) && !defined(MDM_EMU)
  if (xb_tl_target() >= 0) {   // timeline probe: stamps on for exactly the selected launch (stream-ordered switch)
    static int on_v[2] = {0, 1};
    const int on = (xb_tl_count()++ == xb_tl_target()) ? 1 : 0;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(g_xb_tl_on), &on_v[on], sizeof(int), 0, hipMemcpyHostToDevice, stream) != hipSuccess) return -1;
  }
#endif
  MDM_LAUNCH(kfn, dim3(total), dim3(64 * XB_WAVES), LDS, stream, a, tpg, total);
  return 0;
}

// -1: hipFuncSetAttribute failed; -2: unsupported shape (callers check xattn_block_supported first)
inline int launch_xattn_block(const XattnArgs& a, int D, hipStream_t s) {
  if (!xattn_block_supported(D, a.ntok) || a.M % a.S != 0) return -2;
  const int nkt = (a.ntok + 31) / 32;
  if (D == 512) {
    if (nkt == 1) return launch_xattn_block_t<4, 1>(a, s);
    if (nkt == 2) return launch_xattn_block_t<4, 2>(a, s);
    return launch_xattn_block_t<4, 3>(a, s);
  }
  if (nkt == 1) return launch_xattn_block_t<2, 1>(a, s);
  if (nkt == 2) return launch_xattn_block_t<2, 2>(a, s);
  return launch_xattn_block_t<2, 3>(a, s);
}
#endif

}  // namespace mdm

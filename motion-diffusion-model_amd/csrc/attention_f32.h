// Fused multi-head self-attention for the MDM encoder (exact fp32 on v_mfma_f32_32x32x2_f32).
//
// Replaces, per layer, torch's in-proj split / head transposes / scaled_dot_product_attention / head
// merge (F.multi_head_attention_forward as driven by nn.TransformerEncoderLayer, model/mdm.py:77-84,
// :253) and the key-padding mask of mdm.py:241-247.
//
//   qkv  [nseq*S, 3*D]  row = (sequence, token);  Q cols [0,D) already scaled by 1/sqrt(hd), K [D,2D), V [2D,3D)
//        (or separate query / key / value sources with their own lengths: AttnF32Args)
//   out  [nseq*S, D]    head h occupies cols [h*hd, (h+1)*hd)
//
// One workgroup per (sequence, head); hd = 128; S <= 32*NKT <= 224.  Wave w owns query rows
// [32w, 32w+32).  Everything is computed TRANSPOSED so that the softmax axis is lane-local and the
// probabilities never move:
//   phase 1  St[key][query] = K . Q^T      A-operand = K rows (LDS, conflict-free b128 reads),
//                                           B-operand = this wave's Q rows (registers)
//            -> lane (query = l&31, half = l>>5) holds, for every 32-key tile, the 16 keys
//               mfma_row(reg, half): softmax = in-register max/sum + ONE cross-half shuffle.
//   phase 2  Ot[d][query]  = V^T . P^T     B-operand = the probability registers AS THEY ARE
//               (MFMA #reg of a tile consumes key mfma_row(reg, half) from each half),
//            A-operand = V[key][d-tile*32 + lane&31] (LDS, ds_read_b32, conflict-free).
// K and V share one LDS buffer (K for phase 1, V for phase 2, then the output tile is staged
// through it for coalesced 16-byte stores).  Padded keys (>= S, or masked frames) get p = 0 and
// their K/V rows are zero-filled, so no NaN/Inf can leak from uninitialised memory.
#pragma once
#include "common.h"

namespace mdm {

constexpr int ATT_HD = 128;
constexpr int ATT_KLD = ATT_HD + 4;  // K / O staging row stride (floats)

// Operands.  Self-attention of the encoder: q = qkv, k = qkv + D, v = qkv + 2D, ldq = ldkv = 3D, Sq = Sk = S, lead = 1
// (token 0, the condition token, is never masked; frame j-1 must be < length: mdm.py:241-247).  DiP decoder
// (mdm.py:255-270): the same with lead = 0 for its self-attention, and for the cross-attention over the text memory
// q = projected tokens [nseq*Sq][D], k / v = the two halves of the projected memory [nseq*Sk][2D], lengths = text
// token counts (memory_key_padding_mask).
struct AttnF32Args {
  const float* q;
  int ldq;
  const float* k;
  const float* v;
  int ldkv;
  int Sq, Sk;
  const int* lengths;  // [B] or null: valid keys = min(Sk, lead + lengths[seq % B])
  int lead;
  int B;
  // optional rows [D] added to every key / value row while it is staged: the DiP window loop keeps the text part of the
  // projected memory (constant over a window's steps) and adds the step's projected time embedding here (decoder.h)
  const float* kadd = nullptr;
  const float* vadd = nullptr;
  // the K / V source may hold MORE samples than this launch covers (a sample group of a larger batch, mdm_sample_loop_dec):
  // local sequence (branch br, sample bl) reads source sequence br * kv_B + kv_b0 + bl; kv_B = 0: the source is local
  int kv_B = 0;
  int kv_b0 = 0;
  // `lengths` may likewise describe MORE samples than this launch covers: local sample bl is entry len_b0 + bl of an array
  // laid out for len_B samples (counts [len_B], then -- ABI 7 bitmap form -- eight words per sample); len_B = 0: B samples
  int len_B = 0;
  int len_b0 = 0;
};

// blockDim.x = 64 * ceil(Sq / 32): wave w owns query rows [32w, 32w + 32); NKT = ceil(Sk / 32) key tiles.
template <int NKT>
__global__ __launch_bounds__(448) void attention_f32_kernel(AttnF32Args a, float* __restrict__ out, int D, int H,
                                                             p16_t* __restrict__ oh, p16_t* __restrict__ ol) {
  MDM_DYN_SMEM(float, smem);  // max(NKT, query tiles) * 32 rows x ATT_KLD floats
  const int NT = (int)blockDim.x;
  constexpr int ROWS = 32 * NKT;
  const int S = a.Sk, Sq = a.Sq;

  const int tid = threadIdx.x;
  const int lane = tid & 63, w = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int seq = blockIdx.x / H, head = blockIdx.x - seq * H;
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
  const uint32_t* kbits = nullptr;   // arbitrary frame mask of this sequence (common.h key_valid_bits), else a count
  if (a.lengths != nullptr) {
    const int lb = a.len_b0 + seq % a.B, LB = a.len_B > 0 ? a.len_B : a.B;
    const int cnt = a.lengths[lb];
    if (cnt >= 0) nvalid = min(S, a.lead + cnt);
    else kbits = reinterpret_cast<const uint32_t*>(a.lengths + LB + 8 * lb);
  }

  // ---- Q fragment: query row q = 32w + r, this lane-half's 64 d's
  const int q = 32 * w + r;
  float qf[64];
  {
    const float* qp = qbase + (size_t)q * a.ldq + 64 * h;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float4 v = (q < Sq) ? ld4(qp + 4 * j) : zero4();
      qf[4 * j + 0] = v.x; qf[4 * j + 1] = v.y; qf[4 * j + 2] = v.z; qf[4 * j + 3] = v.w;
    }
  }
  // ---- stage K (rows >= S zero-filled)
  for (int idx = tid; idx < ROWS * 32; idx += NT) {
    const int key = idx >> 5, c4 = idx & 31;
    float4 v = (key < S) ? ld4(kbase + (size_t)key * ld + 4 * c4) : zero4();
    if (a.kadd != nullptr && key < S) v = add4(v, ld4(a.kadd + head * ATT_HD + 4 * c4));
    st4(&smem[key * ATT_KLD + 4 * c4], v);
  }
  __syncthreads();

  // ---- phase 1: St tiles
  f32x16 p[NKT];
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* kp = &smem[(kt * 32 + r) * ATT_KLD + 64 * h];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const float4 kf = ld4(kp + 4 * c);
      acc = mfma_f32(kf.x, qf[4 * c + 0], acc);
      acc = mfma_f32(kf.y, qf[4 * c + 1], acc);
      acc = mfma_f32(kf.z, qf[4 * c + 2], acc);
      acc = mfma_f32(kf.w, qf[4 * c + 3], acc);
    }
    p[kt] = acc;
  }

  // ---- softmax over keys (lane-local + one cross-half exchange)
  float mx = -INFINITY;
  if (kbits == nullptr) {
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = kt * 32 + mfma_row(e, h);
        const float s = (key < nvalid) ? p[kt][e] : -INFINITY;
        p[kt][e] = s;
        mx = fmaxf(mx, s);
      }
  } else {
    uint32_t wb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wb[i] = kbits[i];
    static_for<NKT>([&](auto kt_tag) __attribute__((always_inline)) {
      constexpr int kt = decltype(kt_tag)::value;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = mfma_row(e, h), key = kt * 32 + row;
        const float s = (key < S && key_valid_bits<kt>(wb, row, a.lead)) ? p[kt][e] : -INFINITY;
        p[kt][e] = s;
        mx = fmaxf(mx, s);
      }
    });
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
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
    for (int e = 0; e < 16; ++e) p[kt][e] *= inv;

  __syncthreads();  // every wave is done reading K
  // ---- stage V into the same buffer, row stride ATT_HD (rows >= S zero-filled)
  for (int idx = tid; idx < ROWS * 32; idx += NT) {
    const int key = idx >> 5, c4 = idx & 31;
    float4 v = (key < S) ? ld4(vbase + (size_t)key * ld + 4 * c4) : zero4();
    if (a.vadd != nullptr && key < S) v = add4(v, ld4(a.vadd + head * ATT_HD + 4 * c4));
    st4(&smem[key * ATT_HD + 4 * c4], v);
  }
  __syncthreads();

  // ---- phase 2: Ot[d][query]
  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[dt][e] = 0.f;
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const float* vp = &smem[(kt * 32 + mfma_row(e, h)) * ATT_HD + r];
      const float pv = p[kt][e];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] = mfma_f32(vp[32 * dt], pv, o[dt]);
    }
  }
  __syncthreads();  // every wave is done reading V

  // ---- stage O[q][d] (row stride ATT_KLD) and store coalesced
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d0 = dt * 32 + 8 * g + 4 * h;  // rows mfma_row(4g..4g+3, h) are 4 consecutive d's
      st4(&smem[q * ATT_KLD + d0], make_float4(o[dt][4 * g + 0], o[dt][4 * g + 1], o[dt][4 * g + 2], o[dt][4 * g + 3]));
    }
  __syncthreads();
  const size_t obase = (size_t)seq * Sq * D + head * ATT_HD;
  for (int idx = tid; idx < NT / 2 * 32; idx += NT) {   // NT / 2 = 32 * query tiles rows
    const int qq = idx >> 5, c4 = idx & 31;
    if (qq < Sq) {
      const float4 v = ld4(&smem[qq * ATT_KLD + 4 * c4]);
      const size_t o = obase + (size_t)qq * D + 4 * c4;
      if (out != nullptr) st4(out + o, v);
      if (oh != nullptr) split4_store(oh + o, ol + o, v);  // planes for the out_proj f16x3 GEMM
    }
  }
}

inline size_t attention_lds_bytes(int nkt, int nqt) { return (size_t)(nkt > nqt ? nkt : nqt) * 32 * ATT_KLD * sizeof(float); }

}  // namespace mdm

// Exact-fp32 MFMA GEMM for the MDM denoiser:  C[m][n] = sum_k A[m][k] * B[n][k]   ("NT": both
// operands are K-contiguous, which is what nn.Linear's [out, in] weights and row-major token
// matrices give us for free).
//
// Replaces the reference's `addmm` calls: in_proj / out_proj /
// linear1 / linear2 of torch's TransformerEncoderLayer (model/mdm.py:77-84), InputProcess
// (mdm.py:343-349), OutputProcess (mdm.py:372-386), TimestepEmbedder (mdm.py:329-330), embed_text
// (mdm.py:218).  Operand gathering and the epilogue are policy structs so the layout shuffles the
// reference performs as separate `copy_` kernels (35 % of its CPU time) disappear into the GEMM.
//
// Machine mapping (gfx950): 256 threads = 4 waves in a 2x2 grid, block tile 128x128, wave tile
// 64x64 = 2x2 v_mfma_f32_32x32x2_f32 accumulators (64 acc VGPRs), BK = 32.
//   * LDS tiles are [rows][32+4] floats: the +4 pad makes the per-lane 16-byte fragment reads
//     (row = lane&31, 16 consecutive k starting at 16*(lane>>5)) hit 16 distinct 16-B slots per
//     ds_read_b128 lane group -> conflict-free.
//   * lane-half h owns k in [16h, 16h+16) of every BK tile for BOTH operands, so the MFMA's two
//     k-slots always pair matching k's; the k summation order differs from a sequential loop only
//     in association, and every product/accumulate is an exact fp32 fma.
//   * global->LDS: register-staged; the next K tile's global loads are issued before the MFMAs of
//     the current tile (latency hidden behind 32 MFMAs x 64 cycles per wave).
//   * linear block id is remapped per XCD so the n-tiles that share an A row-panel share an L2.
#pragma once
#include "common.h"

namespace mdm {

constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_BK = 32;
constexpr int GEMM_LDS_LD = GEMM_BK + 4;  // floats per LDS row
constexpr int GEMM_THREADS = 256;

// ------------------------------------------------------------------------------------------------
// Operand loaders.  load4(row, k) returns logical elements (row, k..k+3); out-of-range -> 0.
// kColumnStaging selects the thread->element map used while staging: false = 8 threads sweep the
// 32 k's of one row (row-major sources), true = consecutive threads take consecutive rows (sources
// that are contiguous along the row index, e.g. the [B, J, 1, T] pose tensor).
// ------------------------------------------------------------------------------------------------
// LayerNorm folded into the GEMMs around it (the DiP decoder's post-norm layers; the same idea as gemm_x3.h's FOLD / OSTAT /
// RES 3 on this skeleton): the GEMM that produces a pre-norm sum y = x + f(x) (out_proj, linear2) writes y itself plus, per
// row and 32-column block, the partial statistics (sum y, sum (y - block mean)^2); whoever reads LN(y) merges the partials of
// its tile's rows once (Chan's update: no E[y^2] - mean^2 cancellation):
//   * the next GEMM, whose A operand is LN(y), runs on y itself with weights pre-multiplied by gamma (mdm_prepare):
//       W . LN(y) + b = rstd * (W' . y - mean * colsum(W')) + (b + W . beta)          applied per row in the epilogue;
//   * the next residual rebuilds (y - mean) * rstd * gamma + beta from the y it was loading anyway.
// No LayerNorm launch, no normalised copy of the residual stream (LnLinearEpilogue).
constexpr int LN_PART_COLS = 32;
struct LnFold {
  const float* stat = nullptr;   // [M][parts][2]; null = the tensor is not pre-norm (read as it is)
  const float* gamma = nullptr;  // [D]
  const float* beta = nullptr;
  int parts = 0;                 // D / 32
  float inv_dim = 0.f;           // 1 / D
};
__device__ __forceinline__ float2 ln_row_stats(const LnFold& f, int row, int M) {   // -> (mean, rstd), eps 1e-5
  if (row >= M) return make_float2(0.f, 1.f);
  const float* q = f.stat + (size_t)row * f.parts * 2;
  float sum = 0.f;
  for (int i = 0; i < f.parts; i += 2) {
    const float4 v = ld4(q + 2 * i);
    sum += v.x + v.z;
  }
  const float mean = sum * f.inv_dim;
  float m2 = 0.f;
  for (int i = 0; i < f.parts; i += 2) {
    const float4 v = ld4(q + 2 * i);
    const float d0 = v.x * (1.0f / LN_PART_COLS) - mean, d1 = v.z * (1.0f / LN_PART_COLS) - mean;
    m2 += (v.y + LN_PART_COLS * d0 * d0) + (v.w + LN_PART_COLS * d1 * d1);
  }
  return make_float2(mean, 1.0f / sqrtf(m2 * f.inv_dim + 1e-5f));
}
__device__ __forceinline__ float4 ln_apply4(float4 y, float2 st, float4 g, float4 b) {
  return make_float4((y.x - st.x) * st.y * g.x + b.x, (y.y - st.x) * st.y * g.y + b.y, (y.z - st.x) * st.y * g.z + b.z,
                     (y.w - st.x) * st.y * g.w + b.w);
}

struct RowMajorLoader {
  static constexpr bool kColumnStaging = false;
  static constexpr bool kGather = false;
  static constexpr bool kFragments = false;
  const float* p;
  int ld;    // floats between rows (multiple of 4)
  int rows;  // valid rows
  int K;     // valid k (multiple of 4)
  __device__ __forceinline__ float4 load4(int row, int k) const {
    if (row < rows && k < K) return ld4(p + (size_t)row * ld + k);
    return zero4();
  }
};

// B operand (weights) of the small X3 GEMM as PRE-SPLIT planes in MFMA-fragment order (gemm_x3.h header: Wp[plane][n/32][k/16]
// [lane][8], values w * 2^8, packed once by mdm_prepare): a wave's B fragment of one 16-deep k sub-step is ONE coalesced 1 KB
// load straight into registers.  The weights then never pass through LDS (half of the kernel's LDS traffic: 32 KB written and
// 64 KB read per 128-deep k tile) and are not re-split by every one of the M/64 row tiles.
struct X3FragB {
  static constexpr bool kColumnStaging = false;
  static constexpr bool kGather = false;
  static constexpr bool kFragments = true;
  const p16_t* hi;
  const p16_t* lo;
  int nblocks;   // 32-row blocks (rows padded)
  int K;
  __device__ __forceinline__ float4 load4(int, int) const { return zero4(); }
};

// A-operand of InputProcess: logical row m = b*T + t, logical k = feature jf in [0, J*F):
// element = x[b][jf][t] of the contiguous [B, J*F, T] pose tensor (mdm.py:345 permute+reshape fused away).
// Prefix completion (mdm.py:203-206, DiP): the first C of the T frames of a row batch come from `prefix` [B, J*F, C],
// the remaining T - C from x [B, J*F, T - C] -- the torch.cat along the frame axis is fused away.
struct PoseGatherLoader {
  static constexpr bool kColumnStaging = true;
  static constexpr bool kGather = true;
  static constexpr bool kFragments = false;
  const float* x;
  int T, JF, rows;
  const float* prefix = nullptr;
  int C = 0;
  __device__ __forceinline__ float4 load4(int row, int k) const {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < rows) {
      const int b = row / T, t = row - b * T;
      const bool in_prefix = t < C;
      const int Tx = in_prefix ? C : T - C;
      const float* base = (in_prefix ? prefix : x) + ((size_t)b * JF) * Tx + (in_prefix ? t : t - C);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (k + i < JF) v[i] = base[(size_t)(k + i) * Tx];
    }
    return make_float4(v[0], v[1], v[2], v[3]);
  }
};

// B-operand of OutputProcess (computed transposed, see OutProjEpilogue): logical row n = b*T + t maps to
// token row (b*S + 1 + t) of the final encoder output -- the `[1:]` slice of mdm.py:253 -- and, when
// classifier-free guidance is on, combines the cond / uncond branches *before* the projection:
//   W (u + s (c - u)) + b  ==  (W u + b) + s ((W c + b) - (W u + b))      (utils/sampler_util.py:34)
// which halves this GEMM and removes the separate combine pass.
struct CfgTokenLoader {
  static constexpr bool kColumnStaging = false;
  static constexpr bool kGather = true;
  static constexpr bool kFragments = false;
  const float* tok;    // [nbranch*B*S, D]
  const float* scale;  // [B] or nullptr (single branch)
  int B, T, S, D, rows;
  int lead = 1;        // tokens of a sequence in front of the T output frames: the condition token (trans_enc), or the
                       // context_len prefix frames of the DiP decoder (mdm.py:278-279)
  __device__ __forceinline__ float4 load4(int row, int k) const {
    if (row >= rows || k >= D) return zero4();
    const int b = row / T, t = row - b * T;
    const float4 c = ld4(tok + ((size_t)b * S + lead + t) * D + k);
    if (scale == nullptr) return c;
    const float4 u = ld4(tok + ((size_t)(B + b) * S + lead + t) * D + k);
    const float s = scale[b];
    return make_float4(u.x + s * (c.x - u.x), u.y + s * (c.y - u.y), u.z + s * (c.z - u.z), u.w + s * (c.w - u.w));
  }
};

// ------------------------------------------------------------------------------------------------
// Epilogues.  A lane owns 2 output columns (n) and 32 output rows (m) of its wave tile; the kernel
// asks the policy for a per-row and a per-column context once and then calls store() per element, so
// the index arithmetic (divisions by T, base offsets) is not repeated 64 times.  Lanes 0..31 of a wave
// hold 32 consecutive n for a fixed m, so n-contiguous destinations are written in 128-byte runs.
// ------------------------------------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_GELU = 1, ACT_SILU = 2 };

// v = act(acc + bias[n]) * (n < scale_cols ? col_scale : 1) + (res ? res[m][n] : 0);
// out[m][n] = v (if out) and/or the 16-bit split planes oh/ol[m][n] = hi/lo(v) (if oh) for a following f16x3 GEMM.
struct LinearEpilogue {
  static constexpr bool kVec4 = true;   // has the row-major 4-column form (pre4 / store4) the kernel prefers when ld % 4 == 0
  static constexpr bool kLn = false;
  float* out;        // may be null when only the planes are wanted
  const float* bias;  // [N] or null (no bias)
  const float* res;  // may alias out (each element is read then written by the same lane)
  int ld;
  int act;
  int scale_cols;   // columns [0, scale_cols) are multiplied by col_scale (q * 1/sqrt(hd) for in_proj)
  float col_scale;
  p16_t* oh;       // optional split planes, same [M][ld] shape
  p16_t* ol;
  struct Row { size_t base; };
  struct Col { int n; float bias, mult; };
  __device__ __forceinline__ Row row(int m) const { return Row{(size_t)m * ld}; }
  __device__ __forceinline__ Col col(int n) const { return Col{n, bias != nullptr ? bias[n] : 0.f, n < scale_cols ? col_scale : 1.f}; }
  // pre(): what store() needs from memory for this element, fetched for ALL of a lane's elements before the first store --
  // `res` may alias `out` (in-place residual), so a load behind a store cannot be hoisted by the compiler and the epilogue
  // degenerated into 16 serial load -> store round trips per lane (the DiP decoder's small GEMMs spent most of their time there)
  __device__ __forceinline__ float pre(const Row& r, const Col& c) const { return res != nullptr ? res[r.base + c.n] : 0.f; }
  __device__ __forceinline__ void store(const Row& r, const Col& c, float acc, float resv) const {
    float v = acc + c.bias;
    if (act == ACT_GELU) v = gelu_erf(v);
    else if (act == ACT_SILU) v = silu(v);
    v *= c.mult;
    const size_t o = r.base + c.n;
    v += resv;
    if (out != nullptr) out[o] = v;
    if (oh != nullptr) split_p16(v, oh[o], ol[o]);
  }
  // the same for 4 consecutive columns n .. n+3 of row m (n % 4 == 0, ld % 4 == 0): 16-byte loads / stores
  __device__ __forceinline__ bool vec4_ok() const { return (ld & 3) == 0; }
  __device__ __forceinline__ float4 pre4(int m, int n) const { return res != nullptr ? ld4(res + (size_t)m * ld + n) : zero4(); }
  __device__ __forceinline__ void store4(int m, int n, float4 a, float4 rv) const {
    const float4 b4 = bias != nullptr ? ld4(bias + n) : zero4();
    float v[4] = {a.x + b4.x, a.y + b4.y, a.z + b4.z, a.w + b4.w};
    const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (act == ACT_GELU) v[q] = gelu_erf(v[q]);
      else if (act == ACT_SILU) v[q] = silu(v[q]);
      if (n + q < scale_cols) v[q] *= col_scale;
      v[q] += rr[q];
    }
    const size_t o = (size_t)m * ld + n;
    const float4 v4 = make_float4(v[0], v[1], v[2], v[3]);
    if (out != nullptr) st4(out + o, v4);
    if (oh != nullptr) split4_store(oh + o, ol + o, v4);
  }
};

// v = act(A-fold(acc) + bias[n]) * (n < scale_cols ? col_scale : 1) + R[m][n]
//   A-fold(acc) = rstd[m] * (acc - mean[m] * a_colsum[n])   when a_ln.stat is set (the A operand was a pre-norm sum, see LnFold)
//   R = LN(res) (res_ln.stat set: res is itself a pre-norm sum), res as it is, or nothing (res == null)
// v -> out (which may alias res: each element is read, then written, by the same lane) and, when ostat is set, per row and
// 32-column block (sum v, sum (v - block mean)^2) -> ostat [M][ld/32][2] for the readers of LN(v).  Row-major form only
// (ld % 32 == 0); the kernel builds the (mean, rstd) tables of the tile's rows and performs the 8-lane reductions.
struct LnLinearEpilogue {
  static constexpr bool kVec4 = true;
  static constexpr bool kLn = true;
  float* out;
  const float* bias;
  int ld;
  int act;
  int scale_cols;
  float col_scale;
  LnFold a_ln;             // gamma / beta unused: they live in the weights and the bias
  const float* a_colsum;   // [N]
  const float* res;
  LnFold res_ln;
  float* ostat;
  struct Row { size_t base; };
  struct Col { int n; };
  __device__ __forceinline__ Row row(int m) const { return Row{(size_t)m * ld}; }          // (the per-element form is
  __device__ __forceinline__ Col col(int n) const { return Col{n}; }                        //  never taken: vec4_ok())
  __device__ __forceinline__ float pre(const Row&, const Col&) const { return 0.f; }
  __device__ __forceinline__ void store(const Row&, const Col&, float, float) const {}
  __device__ __forceinline__ bool vec4_ok() const { return true; }
  __device__ __forceinline__ float4 pre4(int m, int n) const { return res != nullptr ? ld4(res + (size_t)m * ld + n) : zero4(); }
  __device__ __forceinline__ void store4(int, int, float4, float4) const {}
  // the per-column vectors of columns n .. n+3, fetched once per 32-row sub-tile (the stores to `out` in between would
  // otherwise keep the compiler from hoisting them)
  struct Cols { float4 bias, colsum, gamma, beta; };
  __device__ __forceinline__ Cols cols4(int n) const {
    Cols c;
    c.bias = ld4(bias + n);
    c.colsum = a_ln.stat != nullptr ? ld4(a_colsum + n) : zero4();
    const bool rl = res != nullptr && res_ln.stat != nullptr;
    c.gamma = rl ? ld4(res_ln.gamma + n) : zero4();
    c.beta = rl ? ld4(res_ln.beta + n) : zero4();
    return c;
  }
  // the value of (m, n .. n+3); sa / sr = (mean, rstd) of A row m / residual row m
  __device__ __forceinline__ float4 value4(int n, const Cols& c, float4 a, float4 rv, float2 sa, float2 sr) const {
    const float4 b4 = c.bias;
    float v[4] = {a.x, a.y, a.z, a.w};
    if (a_ln.stat != nullptr) {
      const float4 c4 = c.colsum;
      const float ms = sa.x * sa.y;
      v[0] = v[0] * sa.y - ms * c4.x; v[1] = v[1] * sa.y - ms * c4.y; v[2] = v[2] * sa.y - ms * c4.z; v[3] = v[3] * sa.y - ms * c4.w;
    }
    v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
    if (res != nullptr && res_ln.stat != nullptr) rv = ln_apply4(rv, sr, c.gamma, c.beta);
    const float rr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (act == ACT_GELU) v[q] = gelu_erf(v[q]);
      else if (act == ACT_SILU) v[q] = silu(v[q]);
      if (n + q < scale_cols) v[q] *= col_scale;
      v[q] += rr[q];
    }
    return make_float4(v[0], v[1], v[2], v[3]);
  }
};

// InputProcess epilogue: token (b, s = 1 + t) of every branch gets  acc + b_in[n] + pe[s][n]
// (mdm.py:348, :251-252); the frame tokens are identical in the cond and uncond branches.
struct EmbedEpilogue {
  static constexpr bool kVec4 = false;
  static constexpr bool kLn = false;
  float* tok;          // [nbranch*B*S, D]
  const float* bias;   // [D]
  const float* pe;     // [max_len, D]
  int B, T, S, D, nbranch;
  p16_t* th;          // optional split planes of tok (f16x3 mode)
  p16_t* tl;
  int lead = 1;        // token rows in front of the frames: 1 (condition token, trans_enc) or 0 (trans_dec: S == T)
  struct Row { size_t tok_off, pe_off; };
  struct Col { int n; float bias; };
  __device__ __forceinline__ Row row(int m) const {
    const int b = m / T, t = m - b * T;
    return Row{((size_t)b * S + lead + t) * D, (size_t)(lead + t) * D};
  }
  __device__ __forceinline__ Col col(int n) const { return Col{n, bias[n]}; }
  __device__ __forceinline__ float pre(const Row& r, const Col& c) const { return pe[r.pe_off + c.n]; }
  __device__ __forceinline__ void store(const Row& r, const Col& c, float acc, float pev) const {
    const float v = acc + c.bias + pev;
    tok[r.tok_off + c.n] = v;
    if (nbranch == 2) tok[r.tok_off + (size_t)B * S * D + c.n] = v;
    if (th != nullptr) {
      p16_t hi, lo;
      split_p16(v, hi, lo);
      th[r.tok_off + c.n] = hi; tl[r.tok_off + c.n] = lo;
      if (nbranch == 2) { th[r.tok_off + (size_t)B * S * D + c.n] = hi; tl[r.tok_off + (size_t)B * S * D + c.n] = lo; }
    }
  }
};

// Per-step scalars of the fused sampler update  x_prev = a_x0 * x0 + a_xt * x_t + sigma * eps  (DDPM
// posterior mean + noise: gaussian_diffusion.py:246-268, :525-540;  DDIM: :729-779, folded on host).
struct StepCoefs {
  float a_x0, a_xt, sigma;
  int clip_denoised;
};

struct NoiseSource {
  const float* noise;  // [B, JF, T] injected noise for this step, or nullptr -> Philox
  uint64_t seed;
  uint32_t sample_base;  // global index of local sample 0 (sharding-invariant streams)
  uint32_t draw;         // draw index: 0 = x_T, 1 + k = k-th step
  uint32_t const_noise;  // p_sample(const_noise=True), gaussian_diffusion.py:527-528: every sample gets the eps of GLOBAL sample 0
                         // (the reference repeats row 0 of its batch; keyed by the global index so that shards agree)
  __device__ __forceinline__ float get(int b, uint32_t elem, size_t off) const {
    if (noise != nullptr) return noise[off];   // injected noise: the caller hands over the already-repeated tensor
    return philox_normal(seed, elem, const_noise ? 0u : sample_base + (uint32_t)b, draw);
  }
};

// OutputProcess computed transposed (m = feature jf, n = b*T + t) so that lanes run along t, the
// contiguous axis of the [B, J, F, T] pose tensors.  mode 0: plain model output (MDM.forward seam);
// mode 1: fused p_sample/ddim_sample tail (inpainting blend, clamp, posterior mean, noise add); the
// step's noise is a buffer (injected by the caller, or filled by randn_kernel just before).
struct OutProjEpilogue {
  static constexpr bool kVec4 = false;
  static constexpr bool kLn = false;
  const float* bias;   // [JF]
  float* out;          // mode 0: model output [nb, JF, T];  mode 1: x_prev [B, JF, T]
  float* x0_out;       // mode 1: optional pred_xstart [B, JF, T]
  const float* x_t;    // mode 1
  const float* noise;  // mode 1: [B, JF, T] or nullptr when sigma == 0
  const unsigned char* inpaint_mask;  // mode 1 optional [B, JF, T] (1 = take inpainted_motion)
  const float* inpaint_motion;
  int T, JF, mode;
  StepCoefs co;
  struct Row { size_t off; float bias; };
  struct Col { size_t base; };
  __device__ __forceinline__ Row row(int m) const { return Row{(size_t)m * T, bias[m]}; }
  __device__ __forceinline__ Col col(int n) const {
    const int b = n / T, t = n - b * T;
    return Col{(size_t)b * JF * T + t};
  }
  __device__ __forceinline__ float pre(const Row& r, const Col& c) const {
    return (mode != 0) ? x_t[c.base + r.off] : 0.f;     // x_t aliases out in the fused loop (in-place update)
  }
  __device__ __forceinline__ void store(const Row& r, const Col& c, float acc, float xt) const {
    const size_t off = c.base + r.off;
    float x0 = acc + r.bias;
    if (mode == 0) { out[off] = x0; return; }
    if (inpaint_mask != nullptr && inpaint_mask[off]) x0 = inpaint_motion[off];
    if (co.clip_denoised) x0 = fminf(1.f, fmaxf(-1.f, x0));
    float v = co.a_x0 * x0 + co.a_xt * xt;
    if (noise != nullptr) v += co.sigma * noise[off];
    if (x0_out != nullptr) x0_out[off] = x0;
    out[off] = v;
  }
};

// ------------------------------------------------------------------------------------------------
// BT = block tile edge: 128 (wave tile 64x64 = 2x2 accumulators) or 64 (wave tile 32x32, one accumulator) -- the small
// tile exists for problems whose 128x128 tiling leaves most of the 256 CUs idle (the DiP decoder's 3840-row GEMMs).
// X3 = the split-precision arithmetic of gemm_x3.h on THIS skeleton (same loaders, same epilogues, fp32 operands in memory):
// each staged fp32 value is split into fp16 hi + lo on its way into LDS and a 32-deep k tile costs 6 v_mfma_f32_32x32x16_f16
// per accumulator (hi*hi + hi*lo + lo*hi, two k sub-steps) instead of 16 v_mfma_f32_32x32x2_f32 -- 5.3x less matrix-pipe time.
// It is what the DiP decoder runs in the default `f16x3` mode: its GEMMs are too small for gemm_x3.h's 208-row tiles and
// their operands are not worth a planes round trip.
// These GEMMs are small (M = 3840 rows at 32 motions): a 64x64 tile's matrix work per 32-deep k tile is 6 MFMAs (0.1 us), so
// the kernel's time is the chain of global-load -> LDS -> barrier round trips, one per k tile.  The X3 form therefore
// stages BK = 128 k per step (4 steps for K = 512 instead of 16): 8 float4 per operand and thread in flight.
// LDS image of a plane: [rows][BK + 8] halfs (row stride = 4 dwords mod 64: the 16-byte fragment reads of 16 consecutive
// rows fall on 16 distinct bank quads).
#ifdef MDM_X3S_KSPLIT1      // A/B builds: four waves per workgroup
constexpr int GEMM_X3_KSPLIT = 1;
#else
constexpr int GEMM_X3_KSPLIT = 2;
#endif
constexpr int gemm_x3_bk(int bt) { return bt == 64 ? 128 : 64; }      // k per staging step (8 float4 per operand and thread either way)
constexpr int gemm_x3_ld(int bt) { return gemm_x3_bk(bt) + 8; }       // halfs per LDS row of a split plane
constexpr int gemm_f32_lds_bytes(int bt, bool x3) { return 2 * (x3 ? 2 * bt * gemm_x3_ld(bt) * 2 : bt * (GEMM_BK + 4) * 4); }
// + behind the operand images: (mean, rstd) of the tile's A rows and of its residual rows (LayerNorm fold, LnFold)
constexpr int gemm_f32_lds_total(int bt, bool x3, bool wfrag = false) {
  return gemm_f32_lds_bytes(bt, x3) / (wfrag ? 2 : 1) + 2 * bt * 8;   // (fragment-ordered weights bypass the LDS: X3FragB)
}
// KS = 2 (X3, 64x64 tiles): EIGHT waves per workgroup -- waves 4-7 take the odd 16-deep k sub-steps of every staged tile (a
// two-way split-K inside the workgroup, summed through LDS before the epilogue) and every thread stages half as much: the
// serial per-thread work of a step (loads, split conversions, LDS writes, MFMAs) halves.
template <class AL, class BL, class EP, int BT, bool X3 = false, int KS = 1>
__global__ __launch_bounds__(GEMM_THREADS * KS, 2 * KS) void gemm_f32_kernel(AL al, BL bl, EP ep, int M, int N, int K,
                                                                         int tiles_n, int weight_is_a) {
  constexpr int NT = GEMM_THREADS * KS;
  constexpr int BK = X3 ? gemm_x3_bk(BT) : GEMM_BK;
  constexpr int GEMM_X3_LD = gemm_x3_ld(BT);
  constexpr int WT = BT / 2;          // wave tile edge
  constexpr int NA = WT / 32;         // 32x32 accumulators per wave tile edge
  constexpr int NST = BT * BK / 4 / NT;   // float4 per operand per thread while staging
  constexpr int TPR = BK / 4;         // threads sweeping the k's of one row (row-major staging)
  constexpr int RPP = NT / TPR;       // rows per staging pass
  static_assert(KS == 1 || (X3 && BT == 64), "the in-workgroup split-K form exists for the small X3 tile only");
  constexpr bool WF = BL::kFragments;   // B = fragment-ordered weight planes, global -> registers (X3FragB)
  static_assert(!WF || (X3 && BT == 64), "fragment-ordered weights: the small X3 tile only");
  // fp32 tiles (exact mode) or hi | lo fp16 planes of the same tiles (X3)
  constexpr int OP_BYTES = gemm_f32_lds_bytes(BT, X3) / 2;
  constexpr int IMG_BYTES = WF ? OP_BYTES : 2 * OP_BYTES;
  MDM_DYN_SMEM(unsigned char, lds_raw);   // 2 * OP_BYTES (the X3 image of a 64-row tile pair is 68 KB: beyond static LDS)
  float* const As = reinterpret_cast<float*>(lds_raw);
  float* const Bs = reinterpret_cast<float*>(lds_raw + OP_BYTES);
  p16_t* const Ah = reinterpret_cast<p16_t*>(lds_raw);
  p16_t* const Al = Ah + BT * GEMM_X3_LD;
  p16_t* const Bh = reinterpret_cast<p16_t*>(lds_raw + OP_BYTES);
  p16_t* const Bl = Bh + BT * GEMM_X3_LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int wm = (wid & 3) >> 1, wn = wid & 1;
  const int kgrp = wid >> 2;          // 0, or 1 for the second wave quartet of the KS = 2 form

  const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int m0 = tile_m * BT, n0 = tile_n * BT;

  // staging coordinates: NST float4 per operand per thread
  int a_row[NST], a_k[NST], b_row[NST], b_k[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    if (AL::kColumnStaging) { a_row[i] = tid & (BT - 1); a_k[i] = (tid / BT + (NT / BT) * i) * 4; }
    else { a_row[i] = tid / TPR + RPP * i; a_k[i] = (tid % TPR) * 4; }
    if (BL::kColumnStaging) { b_row[i] = tid & (BT - 1); b_k[i] = (tid / BT + (NT / BT) * i) * 4; }
    else { b_row[i] = tid / TPR + RPP * i; b_k[i] = (tid % TPR) * 4; }
  }

  f32x16 acc[NA][NA];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int j = 0; j < NA; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Register staging TWO k tiles ahead: the loads of tile kt + 2 are issued right after tile kt has been handed to LDS, so
  // each has a whole step to land (one tile ahead, a load had only the few MFMAs of a step to hide behind: the small X3
  // tiles spent most of a step waiting for memory).  Tile kt lives in register set kt & 1.
  // (the gathering loaders keep ONE tile in flight instead of two: with their address arithmetic two did not fit the 128 VGPRs
  // that two resident workgroups leave a wave -- InputProcess spilled 263 registers, OutputProcess 101)
  constexpr int AHEAD = (AL::kGather || BL::kGather) ? 1 : 2;
  float4 ra[AHEAD][NST], rb[AHEAD][NST];
  const int nk = (K + BK - 1) / BK;
  auto fetch = [&](auto set_tag, int kt) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
    const int kb = kt * BK;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      ra[SET][i] = al.load4(m0 + a_row[i], kb + a_k[i]);
      if constexpr (!WF) rb[SET][i] = bl.load4(n0 + b_row[i], kb + b_k[i]);
    }
  };
  // WF: this wave's B fragments of the k tile in flight, one (hi, lo) pair per 16-deep sub-step it owns; pair kq is re-fetched
  // for the NEXT tile right behind the MFMAs that consumed it (a rolling prefetch one tile deep in the same registers)
  constexpr int NQ = X3 ? BK / 16 / KS : 1;
  p16x8 wfh[WF ? NQ : 1], wfl[WF ? NQ : 1];
  auto fetch_w = [&](int kq, int kt) __attribute__((always_inline)) {
    if constexpr (WF) {
      const int nb = (n0 + wn * WT) >> 5, kk = kt * (BK / 16) + KS * kq + kgrp;
      if (nb < bl.nblocks && kk * 16 < K) {
        const size_t o = (((size_t)nb * (size_t)(bl.K / 16) + kk) * 64 + lane) * 8;
        wfh[kq] = *reinterpret_cast<const p16x8*>(bl.hi + o);
        wfl[kq] = *reinterpret_cast<const p16x8*>(bl.lo + o);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { wfh[kq][j] = 0; wfl[kq][j] = 0; }
      }
    }
  };
  auto step = [&](auto set_tag, int kt) __attribute__((always_inline)) {
    constexpr int SET = decltype(set_tag)::value;
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      if constexpr (X3) {
        // the WEIGHT operand (B; A for the transposed OutputProcess) is split as hi / lo of w * 2^8 (common.h kX3WeightScale)
        const float sa = weight_is_a ? kX3WeightScale : 1.f, sb = weight_is_a ? 1.f : kX3WeightScale;
        split4_store(&Ah[a_row[i] * GEMM_X3_LD + a_k[i]], &Al[a_row[i] * GEMM_X3_LD + a_k[i]],
                     make_float4(ra[SET][i].x * sa, ra[SET][i].y * sa, ra[SET][i].z * sa, ra[SET][i].w * sa));
        if constexpr (!WF)
          split4_store(&Bh[b_row[i] * GEMM_X3_LD + b_k[i]], &Bl[b_row[i] * GEMM_X3_LD + b_k[i]],
                       make_float4(rb[SET][i].x * sb, rb[SET][i].y * sb, rb[SET][i].z * sb, rb[SET][i].w * sb));
      } else {
        st4(&As[a_row[i] * GEMM_LDS_LD + a_k[i]], ra[SET][i]);
        st4(&Bs[b_row[i] * GEMM_LDS_LD + b_k[i]], rb[SET][i]);
      }
    }
    __syncthreads();
    if (kt + AHEAD < nk) fetch(set_tag, kt + AHEAD);
    if constexpr (X3) {
#pragma unroll
      for (int kq = 0; kq < BK / 16 / KS; ++kq) {   // 16-deep k sub-steps: lane (r, h) holds k = 16 ks + 8 h .. + 7 of row r
        const int ks = KS * kq + kgrp;              // (KS = 2: even sub-steps for waves 0-3, odd ones for waves 4-7)
        p16x8 ah[NA], al_[NA], bh[NA], bl_[NA];
#pragma unroll
        for (int t = 0; t < NA; ++t) {
          const int ao = (wm * WT + t * 32 + r) * GEMM_X3_LD + 16 * ks + 8 * h;
          const int bo = (wn * WT + t * 32 + r) * GEMM_X3_LD + 16 * ks + 8 * h;
          ah[t] = *reinterpret_cast<const p16x8*>(&Ah[ao]);
          al_[t] = *reinterpret_cast<const p16x8*>(&Al[ao]);
          if constexpr (WF) { bh[t] = wfh[kq]; bl_[t] = wfl[kq]; }
          else {
            bh[t] = *reinterpret_cast<const p16x8*>(&Bh[bo]);
            bl_[t] = *reinterpret_cast<const p16x8*>(&Bl[bo]);
          }
        }
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
          for (int j = 0; j < NA; ++j) {
            acc[i][j] = mfma_p16(al_[i], bh[j], acc[i][j]);
            acc[i][j] = mfma_p16(ah[i], bl_[j], acc[i][j]);
            acc[i][j] = mfma_p16(ah[i], bh[j], acc[i][j]);
          }
        if constexpr (WF) { if (kt + 1 < nk) fetch_w(kq, kt + 1); }
      }
    } else
#pragma unroll
    for (int c = 0; c < 4; ++c) {  // 4 chunks of 4 k-pairs each
      float4 fa[NA], fb[NA];
#pragma unroll
      for (int t = 0; t < NA; ++t) {
        fa[t] = ld4(&As[(wm * WT + t * 32 + r) * GEMM_LDS_LD + 16 * h + 4 * c]);
        fb[t] = ld4(&Bs[(wn * WT + t * 32 + r) * GEMM_LDS_LD + 16 * h + 4 * c]);
      }
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          acc[i][j] = mfma_f32(fa[i].x, fb[j].x, acc[i][j]);
          acc[i][j] = mfma_f32(fa[i].y, fb[j].y, acc[i][j]);
          acc[i][j] = mfma_f32(fa[i].z, fb[j].z, acc[i][j]);
          acc[i][j] = mfma_f32(fa[i].w, fb[j].w, acc[i][j]);
        }
    }
    __syncthreads();
  };
  fetch(std::integral_constant<int, 0>{}, 0);
  if constexpr (WF) {
#pragma unroll
    for (int kq = 0; kq < NQ; ++kq) fetch_w(kq, 0);
  }
  if constexpr (AHEAD == 2) { if (nk > 1) fetch(std::integral_constant<int, 1>{}, 1); }
  // LayerNorm fold (LnLinearEpilogue): (mean, rstd) of the tile's rows, merged once from the producers' per-block partial sums
  // by one thread per row, under the first operand loads; first read in the epilogue, i.e. behind the barriers of the k loop
  float2* const ln_tab = reinterpret_cast<float2*>(lds_raw + IMG_BYTES);   // [0, BT) A rows, [BT, 2 BT) residual rows
  if constexpr (EP::kLn) {
    static_assert(NT >= 2 * BT, "one thread per table row");
    if (ep.a_ln.stat != nullptr && tid < BT) ln_tab[tid] = ln_row_stats(ep.a_ln, m0 + tid, M);
    if (ep.res_ln.stat != nullptr && tid >= BT && tid < 2 * BT) ln_tab[tid] = ln_row_stats(ep.res_ln, m0 + tid - BT, M);
  }
  if constexpr (AHEAD == 2) {
    for (int kt = 0; kt < nk; kt += 2) {
      step(std::integral_constant<int, 0>{}, kt);
      if (kt + 1 < nk) step(std::integral_constant<int, 1>{}, kt + 1);
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) step(std::integral_constant<int, 0>{}, kt);
  }

  static_assert(!WF || (KS == 2 ? 4 * 16 * 64 * 4 : 0) + 4 * 32 * 36 * 4 <= IMG_BYTES, "split-K buffer + epilogue patches must fit the A image");
  if constexpr (KS == 2) {
    // the second quartet's partial tile -> LDS (the staging image is dead: the k loop ended with a barrier) -> added by the
    // first quartet, which owns the epilogue; lane-major [wave][reg][lane] so that both sides are conflict-free
    float* const part = reinterpret_cast<float*>(lds_raw);
    if (kgrp == 1) {
#pragma unroll
      for (int e = 0; e < 16; ++e) part[((wid & 3) * 16 + e) * 64 + lane] = acc[0][0][e];
    }
    __syncthreads();
    if (kgrp == 1) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][0][e] += part[(wid * 16 + e) * 64 + lane];
  }

  if constexpr (EP::kLn) {
    // LnLinearEpilogue: the row-major form below with the folded LayerNorms + the per-row partial statistics of what it writes.  A lane holds 4 columns
    // of row 8 p + (lane >> 3); the 8 lanes of a row cover the 32 columns of one statistics block.
    float* const patch = reinterpret_cast<float*>(lds_raw) + (KS == 2 ? 4 * 16 * 64 : 0) + (wid & 3) * (32 * 36);
    const int prow = lane >> 3, pc4 = (lane & 7) * 4;
    const int parts = ep.ld / LN_PART_COLS;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int mb = m0 + wm * WT + i * 32, nb = n0 + wn * WT + j * 32 + pc4;
        float4 rv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int m = mb + 8 * p + prow;
          rv[p] = (m < M && nb < N) ? ep.pre4(m, nb) : zero4();
        }
        const typename EP::Cols cv = ep.cols4(nb < N ? nb : 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) patch[mfma_row(e, h) * 36 + r] = X3 ? acc[i][j][e] * kX3AccScale : acc[i][j][e];
        wave_lds_fence();
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int lrow = wm * WT + i * 32 + 8 * p + prow, m = m0 + lrow;
          float4 a4 = ld4(&patch[(8 * p + prow) * 36 + pc4]);
#if defined(MDM_F32_EPI_NOP) && !defined(MDM_EMU)
          // (round-4 experiment, profiles/r04c_packed_math.md: the patch values retired by an explicit wait, then MDM_F32_EPI_NOP x 8
          // idle issue slots in front of their first -- with SLP vectorisation: packed -- consumer)
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" : "+v"(a4.x), "+v"(a4.y), "+v"(a4.z), "+v"(a4.w));
#pragma unroll
          for (int nn = 0; nn < MDM_F32_EPI_NOP; ++nn) asm volatile("s_nop 7" : "+v"(a4.x), "+v"(a4.y), "+v"(a4.z), "+v"(a4.w));
#endif
          const bool ok = m < M && nb < N;
          const float4 y = ok ? ep.value4(nb, cv, a4, rv[p], ln_tab[lrow], ln_tab[BT + lrow]) : zero4();
          if (ok) st4(ep.out + (size_t)m * ep.ld + nb, y);
          if (ep.ostat != nullptr) {
            float sum = (y.x + y.y) + (y.z + y.w);
#pragma unroll
            for (int msk = 1; msk <= 4; msk <<= 1) sum += shfl_xor_f32(sum, msk);
            const float bm = sum * (1.0f / LN_PART_COLS);
            float m2 = ((y.x - bm) * (y.x - bm) + (y.y - bm) * (y.y - bm)) + ((y.z - bm) * (y.z - bm) + (y.w - bm) * (y.w - bm));
#pragma unroll
            for (int msk = 1; msk <= 4; msk <<= 1) m2 += shfl_xor_f32(m2, msk);
            if (ok && (lane & 7) == 0)
              *reinterpret_cast<float2*>(ep.ostat + ((size_t)m * parts + (nb / LN_PART_COLS)) * 2) = make_float2(sum, m2);
          }
        }
        wave_lds_fence();
      }
    return;
  }

  if constexpr (EP::kVec4) {
    if (ep.vec4_ok()) {
      // Row-major epilogue: in the accumulator layout a lane owns ONE column, i.e. 16 four-byte stores per 32x32 tile -- a
      // store-ISSUE-bound tail that was ~6 of a small GEMM's ~24 us.  Each wave turns its tiles through a private LDS patch
      // ([32][36] floats; the staging image is dead) into (row, 4 consecutive columns) per lane: 4 sixteen-byte stores.
      // (no barrier: the k loop ended with one, and the patches sit behind the split-K buffer of the KS = 2 form)
      float* const patch = reinterpret_cast<float*>(lds_raw) + (KS == 2 ? 4 * 16 * 64 : 0) + (wid & 3) * (32 * 36);
      const int prow = lane >> 3, pc4 = (lane & 7) * 4;
#pragma unroll
      for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NA; ++j) {
          const int mb = m0 + wm * WT + i * 32, nb = n0 + wn * WT + j * 32 + pc4;
          float4 rv[4];
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int m = mb + 8 * p + prow;
            rv[p] = (m < M && nb < N) ? ep.pre4(m, nb) : zero4();
          }
#pragma unroll
          for (int e = 0; e < 16; ++e) patch[mfma_row(e, h) * 36 + r] = X3 ? acc[i][j][e] * kX3AccScale : acc[i][j][e];
          wave_lds_fence();
#pragma unroll
          for (int p = 0; p < 4; ++p) {
            const int m = mb + 8 * p + prow;
            const float4 a4 = ld4(&patch[(8 * p + prow) * 36 + pc4]);
            if (m < M && nb < N) ep.store4(m, nb, a4, rv[p]);
          }
          wave_lds_fence();
        }
      return;
    }
  }

  typename EP::Col cc[NA];
  bool nv[NA];
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    const int n = n0 + wn * WT + j * 32 + r;
    nv[j] = n < N;
    cc[j] = ep.col(nv[j] ? n : 0);
  }
  // two passes: everything the stores need from memory first (see LinearEpilogue::pre), then the stores
  float pv[NA][16][NA];
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + wm * WT + i * 32 + mfma_row(e, h);
#pragma unroll
      for (int j = 0; j < NA; ++j) pv[i][e][j] = (m < M && nv[j]) ? ep.pre(ep.row(m), cc[j]) : 0.f;
    }
#pragma unroll
  for (int i = 0; i < NA; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = m0 + wm * WT + i * 32 + mfma_row(e, h);
      if (m < M) {
        const typename EP::Row rc = ep.row(m);
#pragma unroll
        for (int j = 0; j < NA; ++j)
          if (nv[j]) ep.store(rc, cc[j], X3 ? acc[i][j][e] * kX3AccScale : acc[i][j][e], pv[i][e][j]);
      }
    }
}

// 128x128 tiles unless they would leave more than half of the chip's workgroup slots (2 per CU) empty
template <class KF>
inline void gemm_f32_allow_lds(KF kfn, int bytes, hipStream_t stream) {
#ifndef MDM_EMU
  if (bytes > 65536) {
    static bool configured[kMaxDevices] = {};   // per instantiation (KF) and device
    (void)rt_dyn_lds_once(kfn, bytes, configured, stream);   // (a failure -- or a first use inside a stream capture -- surfaces as the launch's own error)
  }
#else
  (void)kfn; (void)bytes; (void)stream;
#endif
}
// (X3 tile choice, measured on the DiP bench: taking 128x128 tiles from 100 / 200 workgroups on instead of 512 is 30 % / 23 %
// SLOWER -- 227.6 / 248.1 vs 322.0 motions/s on one box: these GEMMs are per-workgroup latency chains, not L2-traffic-bound,
// and fewer, longer chains lose; profiles/r02_ab.md.)
template <bool X3, class AL, class BL, class EP>
inline void launch_gemm_f32_t(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, hipStream_t stream,
                              int weight_is_a) {
  const int tiles_m = (M + GEMM_BM - 1) / GEMM_BM, tiles_n = (N + GEMM_BN - 1) / GEMM_BN;
  // X3: ALWAYS the 64-row form (in-workgroup split-K, KS = 2).  The two forms associate the k-sum differently, and a tile shape that
  // follows the row count makes a sample's result depend on how many OTHER samples share the launch: DiP's hoisted memory
  // projection re-associated when a batch was sharded over ranks (SURVEY 8e promises output invariant to the
  // number of GPUs).  The exact-fp32 form accumulates in one k order under either tile shape and keeps the size rule.
  if (BL::kFragments || X3 || (tiles_m * tiles_n < 512 && (size_t)M * N >= 64 * 64 * 4)) {
    const int tm = (M + 63) / 64, tn = (N + 63) / 64;
    constexpr int KS = X3 ? GEMM_X3_KSPLIT : 1;
    constexpr int LDS = gemm_f32_lds_total(64, X3, BL::kFragments);
    auto kfn = &gemm_f32_kernel<AL, BL, EP, 64, X3, KS>;
    gemm_f32_allow_lds(kfn, LDS, stream);
    MDM_LAUNCH(kfn, dim3(tm * tn), dim3(GEMM_THREADS * KS), LDS, stream, al, bl, ep, M, N, K, tn, weight_is_a);
    return;
  }
  if constexpr (!BL::kFragments) {
  auto kfn = &gemm_f32_kernel<AL, BL, EP, 128, X3>;
  gemm_f32_allow_lds(kfn, gemm_f32_lds_total(128, X3), stream);
  MDM_LAUNCH(kfn, dim3(tiles_m * tiles_n), dim3(GEMM_THREADS), gemm_f32_lds_total(128, X3), stream, al, bl, ep, M, N, K, tiles_n, weight_is_a);
  }
}
template <class AL, class BL, class EP>
inline void launch_gemm_f32(const AL& al, const BL& bl, const EP& ep, int M, int N, int K, hipStream_t stream,
                            bool x3 = false, bool weight_is_a = false) {
  if constexpr (BL::kFragments) {
    (void)x3;   // pre-split weights: the split arithmetic by construction
    launch_gemm_f32_t<true>(al, bl, ep, M, N, K, stream, 0);
  } else {
    if (x3) launch_gemm_f32_t<true>(al, bl, ep, M, N, K, stream, (int)weight_is_a);
    else launch_gemm_f32_t<false>(al, bl, ep, M, N, K, stream, 0);
  }
}

}  // namespace mdm

// HBM-bound pieces of the MDM sampling step (gfx950): LayerNorm, condition-token assembly, the fused
// sampler update, Philox noise.  All are coalesced 16-byte-per-lane streams; none touch MFMA.
#pragma once
#include "common.h"
#include "gemm_f32.h"  // StepCoefs / NoiseSource

namespace mdm {

// LayerNorm over rows of D = 256*NV floats (in place when write_f32; planes-only otherwise) (norm1/norm2 of nn.TransformerEncoderLayer,
// eps = 1e-5, biased variance; torch transformer.py:951-956).  One wave per row, NV float4 per lane,
// two-pass (mean, then centred sum of squares) entirely in registers.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int rows, float eps,
                                                        p16_t* __restrict__ xh, p16_t* __restrict__ xl,
                                                        int write_f32) {
  constexpr int D = 256 * NV;
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;  // whole wave exits together (row is wave-uniform)
  float* xr = x + (size_t)row * D;
  float4 v[NV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i] = ld4(xr + 256 * i + 4 * lane);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += shfl_xor_f32(s, m);
  const float mean = s * (1.0f / D);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
    ss += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) ss += shfl_xor_f32(ss, m);
  const float rstd = 1.0f / sqrtf(ss * (1.0f / D) + eps);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const float4 g = ld4(gamma + 256 * i + 4 * lane), b = ld4(beta + 256 * i + 4 * lane);
    const float4 o = make_float4(v[i].x * rstd * g.x + b.x, v[i].y * rstd * g.y + b.y, v[i].z * rstd * g.z + b.z,
                                 v[i].w * rstd * g.w + b.w);
    if (write_f32) st4(xr + 256 * i + 4 * lane, o);
    if (xh != nullptr) {  // split planes for the next f16x3 GEMM (and, in that mode, the residual stream itself)
      const size_t off = (size_t)row * D + 256 * i + 4 * lane;
      split4_store(xh + off, xl + off, o);
    }
  }
}

// Token 0 of every sequence: embed_text(cond) (or its bias for the uncond branch) + time-MLP(pe[t]) + pe[0]
// (mdm.py:195, :218-220, :251-252).  cond_emb may be null for a branch => bias only (mask_cond zeroes the
// input: mdm.py:155-156).  Grid = nbranch*B blocks, D/4 threads... each thread 4 consecutive channels.
// `tadd` (null or [B][D]): the target-location embedding the reference adds to the timestep embedding of sample b in EVERY branch
// (`time_emb += mask_cond(embed_target_cond(...))`, mdm.py:197-199; include/mdm_hip.h mdm_set_time_add).
__device__ __forceinline__ float4 time_row(const float* __restrict__ time_table, long long t, const float* __restrict__ tadd, int b, int D,
                                           int c) {
  float4 tt = ld4(time_table + (size_t)t * D + c);
  if (tadd != nullptr) {
    const float4 g = ld4(tadd + (size_t)b * D + c);
    tt = make_float4(tt.x + g.x, tt.y + g.y, tt.z + g.z, tt.w + g.w);
  }
  return tt;
}
__global__ __launch_bounds__(256) void cond_token_kernel(float* __restrict__ tok, const float* __restrict__ cond_emb,
                                                         const float* __restrict__ text_bias,
                                                         const float* __restrict__ time_table,
                                                         const long long* __restrict__ timesteps,  // [B] or null
                                                         int t_uniform,  // used when timesteps == null
                                                         const float* __restrict__ pe, int B, int S, int D,
                                                         int uncond_from_branch, int table_rows,
                                                         p16_t* __restrict__ th, p16_t* __restrict__ tl,
                                                         const float* __restrict__ tadd) {
  const int seq = blockIdx.x, b = seq % B, br = seq / B;
  long long t = (timesteps != nullptr) ? timesteps[b] : (long long)t_uniform;
  if (t < 0) t = 0;
  if (t >= table_rows) t = table_rows - 1;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
    // (text_bias == null: a token that is the timestep embedding alone -- the class token of a trans_dec sequence, mdm.py:256-257)
    const float4 e = (br >= uncond_from_branch || cond_emb == nullptr) ? (text_bias != nullptr ? ld4(text_bias + c) : zero4())
                                                                      : ld4(cond_emb + (size_t)b * D + c);
    const float4 tt = time_row(time_table, t, tadd, b, D, c);
    const float4 p0 = ld4(pe + c);
    const float4 o = make_float4(e.x + tt.x + p0.x, e.y + tt.y + p0.y, e.z + tt.z + p0.z, e.w + tt.w + p0.w);
    st4(tok + (size_t)seq * S * D + c, o);
    if (th != nullptr) split4_store(th + (size_t)seq * S * D + c, tl + (size_t)seq * S * D + c, o);
  }
}

// Text memory of the DiP decoder (mdm.py:217-219, emb_policy 'add'): mem[seq][j] = embed_text(enc_text[j][b]) + time_emb[b]
// for the conditional branch, bias + time_emb[b] for the unconditional one (mask_cond zeroes the input, mdm.py:155-156).
// `proj` [ntok*B][D] holds embed_text(enc_text) token-major (row j*B + b), as bert_encode_text lays the tokens out.
// Grid = nbranch*B*ntok blocks; each thread 4 consecutive channels.
__global__ __launch_bounds__(256) void text_memory_kernel(float* __restrict__ mem, const float* __restrict__ proj,
                                                          const float* __restrict__ text_bias,
                                                          const float* __restrict__ time_table,
                                                          const long long* __restrict__ timesteps, int B, int ntok, int D,
                                                          int uncond_from_branch, int table_rows,
                                                          const float* __restrict__ tadd) {
  const int row = blockIdx.x, seq = row / ntok, j = row - seq * ntok, b = seq % B, br = seq / B;
  // timesteps == null: the text part alone (the window loop adds the projected time row inside the attention kernel)
  long long t = timesteps != nullptr ? timesteps[b] : 0;
  if (t < 0) t = 0;
  if (t >= table_rows) t = table_rows - 1;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
    const float4 e = (br >= uncond_from_branch || proj == nullptr) ? ld4(text_bias + c)
                                                                  : ld4(proj + ((size_t)j * B + b) * D + c);
    // (`tadd`, mdm.py:197-199: part of the timestep embedding; in a window loop it rides with the text part, which is built once)
    float4 tt = timesteps != nullptr ? ld4(time_table + (size_t)t * D + c) : zero4();
    if (tadd != nullptr) {
      const float4 g = ld4(tadd + (size_t)b * D + c);
      tt = make_float4(tt.x + g.x, tt.y + g.y, tt.z + g.z, tt.w + g.w);
    }
    st4(mem + (size_t)row * D + c, make_float4(e.x + tt.x, e.y + tt.y, e.z + tt.z, e.w + tt.w));
  }
}

// The UNCONDITIONAL branch's cross-attention over the text memory needs no attention.  mask_cond zeroes the text (model/mdm.py:155-156),
// so every memory row of such a sequence is the same vector -- embed_text.bias + time_emb (+ target) (mdm.py:217-219) --, its projected
// keys are identical, the scores of a query are equal over the valid tokens, the softmax is uniform and the output of EVERY query is the
// one projected value row (sum_j p_j v = v).  This kernel writes that row (v of the sequence's first memory token, + the step's
// projected time row in a window loop) into all S rows of the sequence's attention-output planes: the q projection and the attention
// kernel then run on the conditional half of a guided batch only (csrc/decoder.h, sequence-tile route).
// Grid = nseq_u * ceil(S / 16) blocks of D / 4 threads.
__global__ __launch_bounds__(256) void uncond_xattn_rows_kernel(p16_t* __restrict__ oh, p16_t* __restrict__ ol, const float* __restrict__ v,
                                                                size_t seq_stride, const float* __restrict__ vadd, int S, int D,
                                                                int dst_seq0, int src_seq0) {
  const int chunks = (S + 15) / 16, u = (int)blockIdx.x / chunks, r0 = ((int)blockIdx.x - u * chunks) * 16;
  const float* src = v + (size_t)(src_seq0 + u) * seq_stride;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
    float4 val = ld4(src + c);
    if (vadd != nullptr) {
      const float4 g = ld4(vadd + c);
      val = make_float4(val.x + g.x, val.y + g.y, val.z + g.z, val.w + g.w);
    }
    for (int r = r0; r < min(r0 + 16, S); ++r) {
      const size_t o = ((size_t)(dst_seq0 + u) * S + r) * D + c;
      split4_store(oh + o, ol + o, val);
    }
  }
}

// v [+ vadd] of sequences src_seq0 .. + n - 1 (first memory token each) -> dst [n][D]  (the one value row of an unconditional sequence)
__global__ __launch_bounds__(128) void gather_value_rows_kernel(float* __restrict__ dst, const float* __restrict__ v, size_t seq_stride,
                                                                const float* __restrict__ vadd, int D, int src_seq0) {
  const float* src = v + (size_t)(src_seq0 + (int)blockIdx.x) * seq_stride;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) {
    float4 val = ld4(src + c);
    if (vadd != nullptr) {
      const float4 g = ld4(vadd + c);
      val = make_float4(val.x + g.x, val.y + g.y, val.z + g.z, val.w + g.w);
    }
    st4(dst + (size_t)blockIdx.x * D + c, val);
  }
}

// ... and therefore the unconditional half's whole cross-attention BLOCK is a row-constant: x' = norm1(y) + (Wo . v + bo) with ONE
// vector o = Wo . v + bo per sequence (mdm.py:85-93, torch _mha_block + the residual).  This kernel is that block for the rows of the
// unconditional sequences on the sequence-tile route: norm1(y) rebuilt from y's planes and its partial row statistics (256 columns per
// partial, merged by Chan's formula exactly as gemm_x3.h does), + o, written as operand planes with the row's partial statistics in
// the producer format every consumer merges ((sum, centred sum of squares) per 256 columns).  One wave = one 256-column partial;
// block = D / 4 threads = one row at a time over 8 rows; grid = rows / 8.
__global__ __launch_bounds__(256) void uncond_xblock_rows_kernel(const p16_t* __restrict__ yh, const p16_t* __restrict__ yl,
                                                                 const float* __restrict__ ystat, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ o,
                                                                 const float* __restrict__ o2,   // null or [D]: added to every o row
                                                                 p16_t* __restrict__ oh, p16_t* __restrict__ ol, float* __restrict__ ostat,
                                                                 int row0, int rows, int S, int D, float inv_dim) {
  const int parts = D / 256, c = threadIdx.x * 4, part = threadIdx.x >> 6;
  const float4 g4 = ld4(gamma + c);
  float4 b4 = ld4(beta + c);
  if (o2 != nullptr) {
    const float4 t4 = ld4(o2 + c);
    b4 = make_float4(b4.x + t4.x, b4.y + t4.y, b4.z + t4.z, b4.w + t4.w);
  }
  for (int i = 0; i < 8; ++i) {
    const int r = (int)blockIdx.x * 8 + i;
    if (r >= rows) break;
    const size_t row = (size_t)row0 + r;
    float s1 = 0.f;
    for (int p = 0; p < parts; ++p) s1 += ystat[(row * parts + p) * 2];
    const float mean = s1 * inv_dim;
    float m2 = 0.f;
    for (int p = 0; p < parts; ++p) {
      const float dm = ystat[(row * parts + p) * 2] * (1.0f / 256.0f) - mean;
      m2 += ystat[(row * parts + p) * 2 + 1] + 256.0f * dm * dm;
    }
    const float rstd = 1.0f / sqrtf(m2 * inv_dim + 1e-5f);
    const size_t off = row * D + c;
    const uint2 a = *reinterpret_cast<const uint2*>(yh + off), b = *reinterpret_cast<const uint2*>(yl + off);
    const float x0 = p16_to_f32((p16_t)(a.x & 0xffff)) + p16_to_f32((p16_t)(b.x & 0xffff));
    const float x1 = p16_to_f32((p16_t)(a.x >> 16)) + p16_to_f32((p16_t)(b.x >> 16));
    const float x2 = p16_to_f32((p16_t)(a.y & 0xffff)) + p16_to_f32((p16_t)(b.y & 0xffff));
    const float x3 = p16_to_f32((p16_t)(a.y >> 16)) + p16_to_f32((p16_t)(b.y >> 16));
    const float4 o4 = ld4(o + (size_t)(r / S) * D + c);
    const float4 v = make_float4((x0 - mean) * rstd * g4.x + b4.x + o4.x, (x1 - mean) * rstd * g4.y + b4.y + o4.y,
                                 (x2 - mean) * rstd * g4.z + b4.z + o4.z, (x3 - mean) * rstd * g4.w + b4.w + o4.w);
    split4_store(oh + off, ol + off, v);
    float t = (v.x + v.y) + (v.z + v.w);
    for (int k = 1; k < 64; k <<= 1) t += shfl_xor_f32(t, k);
    const float mw = t * (1.0f / 256.0f);
    const float dx = v.x - mw, dy = v.y - mw, dz = v.z - mw, dw = v.w - mw;
    float q = (dx * dx + dy * dy) + (dz * dz + dw * dw);
    for (int k = 1; k < 64; k <<= 1) q += shfl_xor_f32(q, k);
    if ((threadIdx.x & 63) == 0) *reinterpret_cast<float2*>(ostat + (row * parts + part) * 2) = make_float2(t, q);
  }
}

// Rows idx[0 .. n) of `table` [rows][D] -> dst [n][D]: the time-embedding rows of a window loop's steps (mdm_sample_loop_dec) in ONE
// launch instead of one device-to-device copy per step (round 6).  Up to 64 rows per launch (the indices travel in the kernel argument).
struct RowGather {
  int idx[64];
};
__global__ __launch_bounds__(128) void gather_rows_kernel(float* __restrict__ dst, const float* __restrict__ table, RowGather g, int D) {
  const int row = blockIdx.x;
  const float* src = table + (size_t)g.idx[row] * D;
  for (int c = threadIdx.x * 4; c < D; c += blockDim.x * 4) st4(dst + (size_t)row * D + c, ld4(src + c));
}

// Stand-alone fused sampler update: classifier-free-guidance combine
// (utils/sampler_util.py:34), inpainting blend (gaussian_diffusion.py:300-304), optional clamp (:347-353),
// posterior mean / DDIM mean with host-folded coefficients, noise add with the t != 0 mask folded into
// sigma (:525-540, :729-779).  Used when the model output comes from outside the fused loop; inside the
// loop the same arithmetic runs in the OutputProcess GEMM epilogue (OutProjEpilogue).
__global__ __launch_bounds__(256) void sampler_step_kernel(const float* __restrict__ x_t,
                                                           const float* __restrict__ out_cond,
                                                           const float* __restrict__ out_uncond,  // may be null
                                                           const float* __restrict__ scale,       // [B] or null
                                                           const unsigned char* __restrict__ inpaint_mask,
                                                           const float* __restrict__ inpaint_motion,
                                                           float* __restrict__ x_prev, float* __restrict__ x0_out,
                                                           int per_sample, int B, StepCoefs co, NoiseSource ns) {
  const size_t total = (size_t)per_sample * B;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_sample);
    const uint32_t e = (uint32_t)(i - (size_t)b * per_sample);
    float x0 = out_cond[i];
    if (out_uncond != nullptr) {
      const float u = out_uncond[i];
      x0 = u + scale[b] * (x0 - u);
    }
    if (inpaint_mask != nullptr && inpaint_mask[i]) x0 = inpaint_motion[i];
    if (co.clip_denoised) x0 = fminf(1.f, fmaxf(-1.f, x0));
    float v = co.a_x0 * x0 + co.a_xt * x_t[i];
    if (co.sigma != 0.f) v += co.sigma * ns.get(b, e, i);
    if (x0_out != nullptr) x0_out[i] = x0;
    x_prev[i] = v;
  }
}

// x = N(0, I) from the counter-based stream (draw index ns.draw), optionally q_sample'd onto an init image:
// out = a * init + s * eps   (gaussian_diffusion.py:226-244, :693-700).  init == null -> out = eps.
__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, const float* __restrict__ init,
                                                    const float* __restrict__ eps_in, float a, float s,
                                                    int per_sample, int B, NoiseSource ns) {
  const size_t total = (size_t)per_sample * B;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_sample);
    const uint32_t e = (uint32_t)(i - (size_t)b * per_sample);
    const float eps = (eps_in != nullptr) ? eps_in[i] : ns.get(b, e, i);
    out[i] = (init != nullptr) ? a * init[i] + s * eps : eps;
  }
}

// Tail of the split-precision OutputProcess: the f16x3 GEMM leaves every token's 263 output features as a fp32 ROW
// (out_tok [nseq*S][ldo]); this kernel drops the S - T leading tokens (the condition token, mdm.py:253; the DiP
// decoder's context_len prefix tokens, mdm.py:278-282), transposes 32x32 tiles through LDS into the
// reference's [.., JF, T] pose layout (mdm.py:385) and fuses, per element, what OutProjEpilogue fuses in the fp32 path:
// mode 0 plain model output for every sequence; mode 1 classifier-free-guidance combine of the two branches
// (utils/sampler_util.py:34 -- here AFTER the projection, which is linear), inpainting blend, clamp, posterior /
// DDIM update with this step's noise -- injected, or drawn inline from the counter-based stream, bit-identical to
// randn_kernel's values (gaussian_diffusion.py:300-304, :347-353, :525-540).
// grid (ceil(T/32), ceil(JF/32), sequences or samples), 256 threads = 32 x 8.
__global__ __launch_bounds__(256) void outproj_finish_kernel(const float* __restrict__ out_tok, int ldo, int S, int T,
                                                             int JF, int B, const float* __restrict__ scale, int mode,
                                                             float* __restrict__ out, float* __restrict__ x0_out,
                                                             const float* __restrict__ x_t, NoiseSource ns,
                                                             const unsigned char* __restrict__ inpaint_mask,
                                                             const float* __restrict__ inpaint_motion, StepCoefs co) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int t0 = blockIdx.x * 32, j0 = blockIdx.y * 32, b = blockIdx.z;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int t = t0 + ty + 8 * i, j = j0 + tx;
    float v = 0.f;
    if (t < T && j < JF) {
      v = out_tok[((size_t)b * S + (S - T) + t) * ldo + j];
      if (mode == 1 && scale != nullptr) {
        const float u = out_tok[((size_t)(B + b) * S + (S - T) + t) * ldo + j];
        v = u + scale[b] * (v - u);
      }
    }
    tile[ty + 8 * i][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = j0 + ty + 8 * i, t = t0 + tx;
    if (j < JF && t < T) {
      const size_t off = ((size_t)b * JF + j) * T + t;
      float x0 = tile[tx][ty + 8 * i];
      if (mode == 0) { out[off] = x0; continue; }
      if (inpaint_mask != nullptr && inpaint_mask[off]) x0 = inpaint_motion[off];
      if (co.clip_denoised) x0 = fminf(1.f, fmaxf(-1.f, x0));
      float v = co.a_x0 * x0 + co.a_xt * x_t[off];
      if (co.sigma != 0.f) v += co.sigma * ns.get(b, (uint32_t)(j * T + t), off);   // injected, or the Philox stream inline
      if (x0_out != nullptr) x0_out[off] = x0;
      out[off] = v;
    }
  }
}

// InputProcess operand for the split-precision GEMM: poses x [B][JF][T] (frames contiguous) -> 16-bit hi/lo planes
// [B*T][KP] (row = (b, t), features contiguous, zero-padded from JF to KP) -- the permute of mdm.py:345 as a 32x32 LDS
// tile transpose.  grid (ceil(T/32), KP/32, B), 256 threads.
__global__ __launch_bounds__(256) void pose_to_planes_kernel(const float* __restrict__ x, p16_t* __restrict__ ph,
                                                             p16_t* __restrict__ pl, int T, int JF, int KP) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int t0 = blockIdx.x * 32, j0 = blockIdx.y * 32, b = blockIdx.z;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = j0 + ty + 8 * i, t = t0 + tx;
    tile[ty + 8 * i][tx] = (j < JF && t < T) ? x[((size_t)b * JF + j) * T + t] : 0.f;
  }
  __syncthreads();
  const int tl = threadIdx.x >> 3, jq = (threadIdx.x & 7) * 4;
  const int t = t0 + tl;
  if (t < T) {
    const size_t o = ((size_t)b * T + t) * KP + j0 + jq;
    split4_store(ph + o, pl + o, make_float4(tile[jq + 0][tl], tile[jq + 1][tl], tile[jq + 2][tl], tile[jq + 3][tl]));
  }
}

// pose_to_planes_kernel AND cond_token_kernel in one launch (round 6: the two are independent and back to back in every step of the
// f16x3 encoder loop; a dependent launch costs 2.2-2.7 us + the kernel's own latency chain, profiles/r06a_phase_barrier.md).  Linear grid:
// blocks [0, nt * nj * B) transpose a 32 x 32 tile of the poses, blocks [nt * nj * B, ... + nseq) write token 0 of a sequence.
struct CondTokArgs {
  float* tok;
  const float* cond_emb;
  const float* text_bias;
  const float* time_table;
  const long long* timesteps;   // [B] or null: then every sample is at t_uniform
  int t_uniform;
  const float* pe;
  int B, S, D, uncond_from_branch, table_rows;
  p16_t *th, *tl;
  const float* tadd;            // null or [B][D]: added to the timestep embedding (mdm.py:197-199)
};
__global__ __launch_bounds__(256) void pose_planes_cond_kernel(const float* __restrict__ x, p16_t* __restrict__ ph, p16_t* __restrict__ pl,
                                                               int T, int JF, int KP, int nB, CondTokArgs ct) {
  __shared__ float tile[32][33];
  const int nt = (T + 31) / 32, nj = KP / 32, pose_blocks = nt * nj * nB;
  if ((int)blockIdx.x >= pose_blocks) {          // ---- cond_token_kernel's body (block-uniform branch: no barrier on this side)
    const int seq = (int)blockIdx.x - pose_blocks, b = seq % ct.B, br = seq / ct.B;
    long long t = (ct.timesteps != nullptr) ? ct.timesteps[b] : (long long)ct.t_uniform;
    if (t < 0) t = 0;
    if (t >= ct.table_rows) t = ct.table_rows - 1;
    for (int c = threadIdx.x * 4; c < ct.D; c += blockDim.x * 4) {
      const float4 e = (br >= ct.uncond_from_branch || ct.cond_emb == nullptr) ? ld4(ct.text_bias + c) : ld4(ct.cond_emb + (size_t)b * ct.D + c);
      const float4 tt = time_row(ct.time_table, t, ct.tadd, b, ct.D, c);
      const float4 p0 = ld4(ct.pe + c);
      const float4 o = make_float4(e.x + tt.x + p0.x, e.y + tt.y + p0.y, e.z + tt.z + p0.z, e.w + tt.w + p0.w);
      st4(ct.tok + (size_t)seq * ct.S * ct.D + c, o);
      if (ct.th != nullptr) split4_store(ct.th + (size_t)seq * ct.S * ct.D + c, ct.tl + (size_t)seq * ct.S * ct.D + c, o);
    }
    return;
  }
  // ---- pose_to_planes_kernel's body
  const int bx = (int)blockIdx.x % nt, by = ((int)blockIdx.x / nt) % nj, b = (int)blockIdx.x / (nt * nj);
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int t0 = bx * 32, j0 = by * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int j = j0 + ty + 8 * i, t = t0 + tx;
    tile[ty + 8 * i][tx] = (j < JF && t < T) ? x[((size_t)b * JF + j) * T + t] : 0.f;
  }
  __syncthreads();
  const int tl = threadIdx.x >> 3, jq = (threadIdx.x & 7) * 4;
  const int t = t0 + tl;
  if (t < T) {
    const size_t o = ((size_t)b * T + t) * KP + j0 + jq;
    split4_store(ph + o, pl + o, make_float4(tile[jq + 0][tl], tile[jq + 1][tl], tile[jq + 2][tl], tile[jq + 3][tl]));
  }
}

// LayerNorm folded into the linear layer that consumes it (mdm_prepare; gemm_x3.h X3Epilogue):
//   wf[n][k] = w[n][k] * gamma[k];  colsum[n] = sum_k wf[n][k];  biasf[n] = bias[n] + sum_k w[n][k] * beta[k]
// so that  W.LN(x) + b = rstd * (Wf.x - mean * colsum) + biasf.  One wave per output row; rows N..Npad-1 of the
// vectors are zeroed.  fp64 accumulation of the two sums (they are constants of the model).
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ bias, float* __restrict__ wf,
                                                             float* __restrict__ colsum, float* __restrict__ biasf,
                                                             int N, int K, int Npad) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= Npad) return;
  if (n >= N) {
    if (lane == 0) { colsum[n] = 0.f; biasf[n] = 0.f; }
    return;
  }
  double cs = 0.0, bs = 0.0;
  for (int k = lane; k < K; k += 64) {
    const float wv = w[(size_t)n * K + k], f = wv * gamma[k];
    wf[(size_t)n * K + k] = f;
    cs += (double)f;
    bs += (double)wv * (double)beta[k];
  }
  float csf = (float)cs, bsf = (float)bs;   // wave reduction in fp32 of the 64 fp64 partials (values are O(1))
#pragma unroll
  for (int msk = 32; msk >= 1; msk >>= 1) {
    csf += shfl_xor_f32(csf, msk);
    bsf += shfl_xor_f32(bsf, msk);
  }
  if (lane == 0) { colsum[n] = csf; biasf[n] = bias[n] + bsf; }
}

// hi/lo 16-bit planes of a fp32 array (weights at mdm_prepare; test inputs).  n must be a multiple of 4.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ src, p16_t* __restrict__ hi,
                                                           p16_t* __restrict__ lo, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    split4_store(hi + 4 * i, lo + 4 * i, ld4(src + 4 * i));
}

// dst[r][0..ld_dst) = src[r][0..cols) zero-padded (16-byte-aligns the 263-wide poseEmbedding weight rows).
__global__ __launch_bounds__(256) void pad_rows_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                       int rows, int cols, int ld_dst) {
  const size_t total = (size_t)rows * ld_dst;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / ld_dst), c = (int)(i - (size_t)r * ld_dst);
    dst[i] = (c < cols) ? src[(size_t)r * cols + c] : 0.f;
  }
}

}  // namespace mdm

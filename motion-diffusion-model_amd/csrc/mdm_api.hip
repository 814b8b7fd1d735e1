// libmdm_hip.so -- host orchestration + C ABI (include/mdm_hip.h) for the MDM sampling hot path.
// Native counterpart of: MDM.forward (model/mdm.py:189-283), ClassifierFreeSampleModel.forward
// (utils/sampler_util.py:27-34), GaussianDiffusion.p_sample_loop / ddim_sample_loop
// (diffusion/gaussian_diffusion.py:591-727, :876-990).  No torch, no allocation: caller-owned pointers.
#include "../../include/mdm_hip.h"
#ifdef MDM_PROBES
#include "../../include/mdm_hip_probe.h"
#endif

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "attention_x3.h"
#include "attention_long.h"
#include "attention_f32.h"
#include "common.h"
#include "elementwise.h"
#include "gemm_f32.h"
#include "gemm_x3.h"
#include "gemm_x3s.h"
#include "xattn_block.h"
#include "selfattn_block.h"
#ifdef MDM_PROBES   // the rejected fp16 + MX-FP6 GEMM arithmetic of round 1: an experiment of the probe library, kept under lab/
#include "../../lab/csrc_probe/gemm_f16f6.h"
#endif
#include "motion_recover.h"

using namespace mdm;

#include "api_runtime.h"
#include "api_launch.h"
#include "encoder.h"

extern "C" {

int mdm_abi_version(void) { return MDM_ABI_VERSION; }

// How this binary was built (include/mdm_hip.h): the loaders refuse a product library whose string lacks "slp=off".
const char* mdm_build_info(void) {
  return "slp="
#ifdef MDM_NO_SLP
         "off"
#else
         "on"
#endif
         ";probes="
#ifdef MDM_PROBES
         "1"
#else
         "0"
#endif
         ";emu="
#ifdef MDM_EMU
         "1"
#else
         "0"
#endif
         ";planes="
#ifdef MDM_SPLIT_BF16
         "bf16"
#else
         "f16"
#endif
      ;
}
const char* mdm_last_error(void) { return g_err.c_str(); }

int mdm_create(const mdm_config_t* cfg, mdm_model_t** out) {
  if (cfg == nullptr || out == nullptr) return fail(MDM_EINVAL, "mdm_create: null argument");
  const int D = cfg->latent_dim, H = cfg->num_heads;
  if (D <= 0 || D % 256 != 0 || D > 1024) return fail(MDM_EUNSUPPORTED, "latent_dim must be 256, 512, 768 or 1024");
  if (H <= 0 || D != H * ATT_HD) return fail(MDM_EUNSUPPORTED, "latent_dim / num_heads must be 128");
  if (cfg->ff_size <= 0 || cfg->ff_size % 32) return fail(MDM_EUNSUPPORTED, "ff_size must be a positive multiple of 32");
  if (cfg->clip_dim <= 0 || cfg->clip_dim % 4) return fail(MDM_EUNSUPPORTED, "clip_dim must be a positive multiple of 4");
  if (cfg->njoints <= 0 || cfg->nfeats <= 0 || cfg->num_layers <= 0 || cfg->max_len < 2)
    return fail(MDM_EINVAL, "mdm_create: non-positive dimension");
  if (cfg->arch != MDM_ARCH_TRANS_ENC && cfg->arch != MDM_ARCH_TRANS_DEC) return fail(MDM_EUNSUPPORTED, "arch must be trans_enc or trans_dec");
  if (cfg->context_len < 0 || (cfg->arch == MDM_ARCH_TRANS_ENC && cfg->context_len != 0))
    return fail(MDM_EUNSUPPORTED, "context_len (prefix completion) belongs to the trans_dec (DiP) configuration");
  mdm_model* m = new (std::nothrow) mdm_model();
  if (m == nullptr) return fail(MDM_EINVAL, "out of host memory");
  m->cfg = *cfg;
  m->jf = cfg->njoints * cfg->nfeats;
  m->jf_pad = (m->jf + 3) / 4 * 4;
  m->jf_out = m->jf_pad;
  m->jf_k = (m->jf + 31) / 32 * 32;
  const int64_t d = D, ff = cfg->ff_size, jf = m->jf;
  auto& e = m->expect;
  e["input_process.poseEmbedding.weight"] = d * jf;
  e["input_process.poseEmbedding.bias"] = d;
  for (int l = 0; l < cfg->num_layers; ++l) {
    const std::string p = (cfg->arch == MDM_ARCH_TRANS_DEC ? "seqTransDecoder.layers." : "seqTransEncoder.layers.") +
                          std::to_string(l) + ".";
    if (cfg->arch == MDM_ARCH_TRANS_DEC) {   // nn.TransformerDecoderLayer: + cross-attention over the memory, + norm3
      e[p + "multihead_attn.in_proj_weight"] = 3 * d * d;
      e[p + "multihead_attn.in_proj_bias"] = 3 * d;
      e[p + "multihead_attn.out_proj.weight"] = d * d;
      e[p + "multihead_attn.out_proj.bias"] = d;
      e[p + "norm3.weight"] = d;
      e[p + "norm3.bias"] = d;
    }
    e[p + "self_attn.in_proj_weight"] = 3 * d * d;
    e[p + "self_attn.in_proj_bias"] = 3 * d;
    e[p + "self_attn.out_proj.weight"] = d * d;
    e[p + "self_attn.out_proj.bias"] = d;
    e[p + "linear1.weight"] = ff * d;
    e[p + "linear1.bias"] = ff;
    e[p + "linear2.weight"] = d * ff;
    e[p + "linear2.bias"] = d;
    e[p + "norm1.weight"] = d;
    e[p + "norm1.bias"] = d;
    e[p + "norm2.weight"] = d;
    e[p + "norm2.bias"] = d;
  }
  e["embed_timestep.time_embed.0.weight"] = d * d;
  e["embed_timestep.time_embed.0.bias"] = d;
  e["embed_timestep.time_embed.2.weight"] = d * d;
  e["embed_timestep.time_embed.2.bias"] = d;
  e["embed_text.weight"] = d * cfg->clip_dim;
  e["embed_text.bias"] = d;
  e["output_process.poseFinal.weight"] = jf * d;
  e["output_process.poseFinal.bias"] = jf;
  e["sequence_pos_encoder.pe"] = (int64_t)cfg->max_len * d;
#ifdef MDM_PROBES   // the probe library's whole-bench A/B scripts (tools/) preset a handle's options from the environment, once, here
  if (const char* e = getenv("MDM_X3S_MAX_SEQS")) m->x3s.max_seqs = atoi(e);
  if (const char* e = getenv("MDM_X3S_RT")) m->x3s.row_tiles = atoi(e);
  if (const char* e = getenv("MDM_X3S_NCB")) m->x3s.ncb = atoi(e);
#endif
  *out = m;
  return MDM_OK;
}

void mdm_destroy(mdm_model_t* m) { delete m; }

// Run-time options of a handle (include/mdm_hip.h, ABI 9).  The library reads NO environment variable (rounds 3-4 did, on the
// launch path); a value takes effect with the next call -- every call resolves its kernel route once, from the handle.
int mdm_set_option(mdm_model_t* m, int32_t key, int32_t value) {
  if (m == nullptr) return fail(MDM_EINVAL, "mdm_set_option: null model");
  switch (key) {
    case MDM_OPT_SMALL_GEMM_MAX_SEQS:
      if (value < 0) return fail(MDM_EINVAL, "mdm_set_option: MDM_OPT_SMALL_GEMM_MAX_SEQS must be >= 0");
      m->x3s.max_seqs = value;
      return MDM_OK;
    case MDM_OPT_SMALL_GEMM_ROW_TILES:
      if (value < 0 || value > 2) return fail(MDM_EINVAL, "mdm_set_option: MDM_OPT_SMALL_GEMM_ROW_TILES must be 0 (by size), 1 or 2");
      m->x3s.row_tiles = value;
      return MDM_OK;
    case MDM_OPT_DEC_FUSED_XATTN:
      if (value < 0 || value > 3) return fail(MDM_EINVAL, "mdm_set_option: MDM_OPT_DEC_FUSED_XATTN must be 0, 1, 2 or 3");
      m->fused_xattn = value;
      return MDM_OK;
    case MDM_OPT_DEC_FUSED_SELFATTN:
      if (value != 0 && value != 1) return fail(MDM_EINVAL, "mdm_set_option: MDM_OPT_DEC_FUSED_SELFATTN must be 0 or 1");
      m->fused_selfattn = value != 0;
      return MDM_OK;
    case MDM_OPT_ATTN_DIRECT_OUT:
      if (value != 0 && value != 1) return fail(MDM_EINVAL, "mdm_set_option: MDM_OPT_ATTN_DIRECT_OUT must be 0 or 1");
      m->attn_direct = value != 0;
      return MDM_OK;
    case MDM_OPT_DEC_TIME_TOKEN:
      if (value != 0 && value != 1) return fail(MDM_EINVAL, "mdm_set_option: MDM_OPT_DEC_TIME_TOKEN must be 0 or 1");
      if (value == 1 && (m->cfg.arch != MDM_ARCH_TRANS_DEC || m->cfg.context_len != 1))
        return fail(MDM_EINVAL, "mdm_set_option: MDM_OPT_DEC_TIME_TOKEN needs a trans_dec model created with context_len = 1 (the class-token row)");
      m->dec_time_token = value != 0;
      return MDM_OK;
    default:
      return fail(MDM_EINVAL, "mdm_set_option: unknown key " + std::to_string(key));
  }
}

// ABI 10: y['target_cond'] (model/mdm.py:197-199).  One-shot, caller-owned; see include/mdm_hip.h.
int mdm_set_time_add(mdm_model_t* m, const float* add_dev, int32_t B) {
  if (m == nullptr) return fail(MDM_EINVAL, "mdm_set_time_add: null model");
  if (add_dev != nullptr && B <= 0) return fail(MDM_EINVAL, "mdm_set_time_add: B must be >= 1");
  m->time_add_next = add_dev;
  m->time_add_B = add_dev != nullptr ? B : 0;
  return MDM_OK;
}

int mdm_get_option(const mdm_model_t* m, int32_t key, int32_t* value) {
  if (m == nullptr || value == nullptr) return fail(MDM_EINVAL, "mdm_get_option: null argument");
  switch (key) {
    case MDM_OPT_SMALL_GEMM_MAX_SEQS: *value = m->x3s.max_seqs; return MDM_OK;
    case MDM_OPT_SMALL_GEMM_ROW_TILES: *value = m->x3s.row_tiles; return MDM_OK;
    case MDM_OPT_DEC_FUSED_XATTN: *value = m->fused_xattn; return MDM_OK;
    case MDM_OPT_DEC_FUSED_SELFATTN: *value = m->fused_selfattn ? 1 : 0; return MDM_OK;
    case MDM_OPT_ATTN_DIRECT_OUT: *value = m->attn_direct ? 1 : 0; return MDM_OK;
    case MDM_OPT_DEC_TIME_TOKEN: *value = m->dec_time_token ? 1 : 0; return MDM_OK;
    default: return fail(MDM_EINVAL, "mdm_get_option: unknown key " + std::to_string(key));
  }
}

int mdm_set_weight(mdm_model_t* m, const char* name, const float* dev_ptr, int64_t numel) {
  if (m == nullptr || name == nullptr || dev_ptr == nullptr) return fail(MDM_EINVAL, "mdm_set_weight: null argument");
  auto it = m->expect.find(name);
  if (it == m->expect.end()) return fail(MDM_EINVAL, std::string("unexpected state-dict key: ") + name);
  if (it->second != numel)
    return fail(MDM_EINVAL, std::string("size mismatch for ") + name + ": expected " + std::to_string(it->second) +
                                " elements, got " + std::to_string(numel));
  if ((reinterpret_cast<uintptr_t>(dev_ptr) & 15) != 0) return fail(MDM_EINVAL, std::string(name) + ": pointer must be 16-byte aligned");
  m->w[name] = dev_ptr;
  m->prepared = false;
  return MDM_OK;
}

size_t mdm_const_bytes(const mdm_model_t* m) {
  if (m == nullptr) return 0;
  const size_t D = m->cfg.latent_dim;
  const size_t FF = m->cfg.ff_size;
  if (m->cfg.arch == MDM_ARCH_TRANS_DEC) {
    // padded poseEmbedding, time table + its MLP scratch; per layer the gamma-folded in_proj / q / linear1 (fp32 + 2 vectors
    // each) and the fragment-ordered planes of in_proj, out_proj, q, cross out_proj, linear1, linear2 (4 bytes per weight)
    const size_t fold = align_up(3 * D * D * 4, 256) + align_up(D * D * 4, 256) + align_up(FF * D * 4, 256) +
                        2 * (align_up(3 * D * 4, 256) + align_up(D * 4, 256) + align_up(FF * 4, 256));
    const size_t planes = align_up(3 * D * D * 4, 256) + 3 * align_up(D * D * 4, 256) + 2 * align_up(FF * D * 4, 256);
    // OutputProcess with the last norm3 folded in (the plane path of decoder_pass): fp32 copy, 2 vectors, planes
    const size_t outp = align_up((size_t)m->jf * D * 4, 256) + 2 * align_up((size_t)32 * ((m->jf_out + 31) / 32) * 4, 256) +
                        align_up(x3_packed_weight_elems(m->jf, (int)D) * 4, 256);
    const size_t kv_all = align_up((size_t)m->cfg.num_layers * 2 * D * D * 4, 256) + align_up((size_t)m->cfg.num_layers * 2 * D * 4, 256);
    return 256 /* range flag */ + align_up(D * m->jf_pad * sizeof(float), 256) +
           2 * align_up((size_t)m->cfg.max_len * D * sizeof(float), 256) + (size_t)m->cfg.num_layers * (fold + planes) + outp + kv_all;
  }
  const size_t per_layer = align_up(3 * D * D * 4, 256) + align_up(D * D * 4, 256) + 2 * align_up(FF * D * 4, 256);
  return 256 /* range flag */ + align_up(D * m->jf_pad * sizeof(float), 256) +
         2 * align_up((size_t)m->cfg.max_len * D * sizeof(float), 256) + (size_t)m->cfg.num_layers * per_layer + align_up(x3_packed_weight_elems(m->jf, (int)D) * 4, 256) +
         align_up((size_t)m->jf_out * sizeof(float), 256) +
         // folded-LayerNorm constants: per layer gamma-scaled in_proj / linear1 planes + 2 vectors each; OutputProcess;
         // one fp32 scratch matrix for the scaled weights before they are packed
         (size_t)m->cfg.num_layers * (align_up(3 * D * D * 4, 256) + align_up(FF * D * 4, 256) +
                                      2 * align_up(3 * D * 4, 256) + 2 * align_up(FF * 4, 256)) +
         align_up(x3_packed_weight_elems(m->jf, (int)D) * 4, 256) + 2 * align_up((size_t)32 * ((m->jf_out + 31) / 32) * 4, 256) +
         align_up(std::max<size_t>(3 * D, FF) * D * 4, 256) + align_up((size_t)D * m->jf_k * 4, 256);
}

int mdm_prepare(mdm_model_t* m, void* const_ws, size_t const_ws_bytes, void* stream) {
  ChainGuard chain_guard(stream);
  if (m == nullptr || const_ws == nullptr) return fail(MDM_EINVAL, "mdm_prepare: null argument");
  for (const auto& kv : m->expect)
    if (m->w.find(kv.first) == m->w.end()) return fail(MDM_ESTATE, "missing weight: " + kv.first);
  if (const_ws_bytes < mdm_const_bytes(m)) return fail(MDM_ENOSPC, "mdm_prepare: const workspace too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int D = m->cfg.latent_dim, R = m->cfg.max_len;
  char* base = static_cast<char*>(const_ws);
  // word 0: "a weight left the range of the 16-bit operand planes" (set by the pack kernels below; mdm_weights_in_range)
  m->range_flag = reinterpret_cast<int*>(base);
  base += 256;
#ifdef MDM_EMU
  *m->range_flag = 0;
#else
  if (hipMemsetAsync(m->range_flag, 0, 4, s) != hipSuccess) return fail(MDM_EHIP, "mdm_prepare: hipMemsetAsync failed");
#endif
  m->w_in_pad = reinterpret_cast<float*>(base);
  base += align_up((size_t)D * m->jf_pad * sizeof(float), 256);
  m->time_table = reinterpret_cast<float*>(base);
  base += align_up((size_t)R * D * sizeof(float), 256);
  float* hidden = reinterpret_cast<float*>(base);
  MDM_LAUNCH(pad_rows_kernel, dim3(256), dim3(256), 0, s, m->w_in_pad, m->W("input_process.poseEmbedding.weight"), D,
             m->jf, m->jf_pad);
  if (int rc = rt_launch_status()) return rc;
  // TimestepEmbedder for every possible t (model/mdm.py:323-330): table[t] = W2 silu(W0 pe[t] + b0) + b2
  if (int rc = launch_linear(nullptr, m->W("sequence_pos_encoder.pe"), D, m->W("embed_timestep.time_embed.0.weight"),
                             m->W("embed_timestep.time_embed.0.bias"), nullptr, hidden, R, D, D, ACT_SILU, 0, 1.f, s))
    return rc;
  if (int rc = launch_linear(nullptr, hidden, D, m->W("embed_timestep.time_embed.2.weight"),
                             m->W("embed_timestep.time_embed.2.bias"), nullptr, m->time_table, R, D, D, ACT_NONE, 0,
                             1.f, s))
    return rc;
  if (m->cfg.arch == MDM_ARCH_TRANS_DEC) {   // the DiP decoder: gamma-folded fp32 copies (fp32 skeleton) AND fragment-ordered planes of
                                             // every layer weight (the operand-plane route of the default f16x3 mode)
    // LayerNorm folded into the consumers of its output (gemm_f32.h LnFold)
    base += align_up((size_t)R * D * sizeof(float), 256);
    const int L = m->cfg.num_layers, FFd = m->cfg.ff_size;
    auto take = [&](size_t n) { float* p = reinterpret_cast<float*>(base); base += align_up(n * 4, 256); return p; };
    auto fold_one = [&](const float* w, const float* bias, const float* gamma, const float* beta, int N, float*& wf, float*& cvec,
                        float*& bvec) -> int {
      wf = take((size_t)N * D);
      cvec = take(N);
      bvec = take(N);
      MDM_LAUNCH(fold_layernorm_kernel, dim3((N + 3) / 4), dim3(256), 0, s, w, gamma, beta, bias, wf, cvec, bvec, N, D, N);
      return rt_launch_status();
    };
    m->dec_fold.assign(L, mdm_model::DecFold{});
    for (int l = 0; l < L; ++l) {
      mdm_model::DecFold& F = m->dec_fold[l];
      if (l >= 1)
        if (int rc = fold_one(m->L(l, "self_attn.in_proj_weight"), m->L(l, "self_attn.in_proj_bias"), m->L(l - 1, "norm3.weight"),
                              m->L(l - 1, "norm3.bias"), 3 * D, F.w_in, F.c_in, F.b_in)) return rc;
      if (int rc = fold_one(m->L(l, "multihead_attn.in_proj_weight"), m->L(l, "multihead_attn.in_proj_bias"), m->L(l, "norm1.weight"),
                            m->L(l, "norm1.bias"), D, F.w_q, F.c_q, F.b_q)) return rc;
      if (int rc = fold_one(m->L(l, "linear1.weight"), m->L(l, "linear1.bias"), m->L(l, "norm2.weight"), m->L(l, "norm2.bias"),
                            FFd, F.w_1, F.c_1, F.b_1)) return rc;
    }
    auto make_planes = [&](const float* src, int N, int K, X3Weights& op) -> int {
      const size_t n = x3_packed_weight_elems(N, K);
      p16_t* hi = reinterpret_cast<p16_t*>(base);
      base += align_up(n * 4, 256);
      op = X3Weights{hi, hi + n};
      return launch_pack_weights(src, hi, hi + n, N, K, s, m->range_flag);
    };
    m->dec_planes.assign(L, mdm_model::DecPlanes{});
    for (int l = 0; l < L; ++l) {
      const mdm_model::DecFold& F = m->dec_fold[l];
      mdm_model::DecPlanes& P = m->dec_planes[l];
      if (int rc = make_planes(l >= 1 ? F.w_in : m->L(l, "self_attn.in_proj_weight"), 3 * D, D, P.in_proj)) return rc;
      if (int rc = make_planes(m->L(l, "self_attn.out_proj.weight"), D, D, P.out_proj)) return rc;
      if (int rc = make_planes(F.w_q, D, D, P.q)) return rc;
      if (int rc = make_planes(m->L(l, "multihead_attn.out_proj.weight"), D, D, P.out_proj2)) return rc;
      if (int rc = make_planes(F.w_1, FFd, D, P.linear1)) return rc;
      if (int rc = make_planes(m->L(l, "linear2.weight"), D, FFd, P.linear2)) return rc;
    }
    {  // OutputProcess <- norm3(L-1) (decoder_pass on operand planes: no LayerNorm kernel in front of poseFinal)
      const int jf32 = (m->jf_out + 31) / 32 * 32;
      float* wf = take((size_t)m->jf * D);
      m->c_out = take(jf32);
      m->b_out = take(jf32);
      MDM_LAUNCH(fold_layernorm_kernel, dim3((jf32 + 3) / 4), dim3(256), 0, s, m->W("output_process.poseFinal.weight"),
                 m->L(L - 1, "norm3.weight"), m->L(L - 1, "norm3.bias"), m->W("output_process.poseFinal.bias"), wf, m->c_out,
                 m->b_out, m->jf, D, jf32);
      if (int rc = rt_launch_status()) return rc;
      if (int rc = make_planes(wf, m->jf, D, m->out_planes_f)) return rc;
    }
    // the cross-attention key | value projections of all layers as one [L * 2D][D] matrix (mdm_sample_loop_dec's hoisted projections)
    m->wkv_all = take((size_t)L * 2 * D * D);
    m->bkv_all = take((size_t)L * 2 * D);
    for (int l = 0; l < L; ++l) {
      if (int rc = rt_copy(m->wkv_all + (size_t)l * 2 * D * D, m->L(l, "multihead_attn.in_proj_weight") + (size_t)D * D,
                           (size_t)2 * D * D * sizeof(float), s)) return rc;
      if (int rc = rt_copy(m->bkv_all + (size_t)l * 2 * D, m->L(l, "multihead_attn.in_proj_bias") + D, (size_t)2 * D * sizeof(float), s)) return rc;
    }
    if ((size_t)(base - static_cast<char*>(const_ws)) > const_ws_bytes) return fail(MDM_ENOSPC, "mdm_prepare: const workspace too small");
    m->prepared = true;
    return MDM_OK;
  }
  // hi/lo planes of the encoder weights (always built: the precision mode can be switched afterwards)
  base += align_up((size_t)R * D * sizeof(float), 256);
  const size_t FF = m->cfg.ff_size;
  m->planes.assign(m->cfg.num_layers, mdm_model::LayerPlanes{});
  auto make_planes = [&](const float* src, int N, int K, X3Weights& op) -> int {
    const size_t n = x3_packed_weight_elems(N, K);
    p16_t* hi = reinterpret_cast<p16_t*>(base);
    p16_t* lo = hi + n;
    base += align_up(n * 4, 256);
    op = X3Weights{hi, lo};
    return launch_pack_weights(src, hi, lo, N, K, s, m->range_flag);
  };
  for (int l = 0; l < m->cfg.num_layers; ++l) {
    if (int rc = make_planes(m->L(l, "self_attn.in_proj_weight"), 3 * D, D, m->planes[l].in_proj)) return rc;
    if (int rc = make_planes(m->L(l, "self_attn.out_proj.weight"), D, D, m->planes[l].out_proj)) return rc;
    if (int rc = make_planes(m->L(l, "linear1.weight"), (int)FF, D, m->planes[l].linear1)) return rc;
    if (int rc = make_planes(m->L(l, "linear2.weight"), D, (int)FF, m->planes[l].linear2)) return rc;
  }
  // OutputProcess in split precision: weight rows / bias padded to jf_out (the pad rows are zero)
  if (int rc = make_planes(m->W("output_process.poseFinal.weight"), m->jf, D, m->out_planes)) return rc;
  m->out_bias_pad = reinterpret_cast<float*>(base);
  base += align_up((size_t)m->jf_out * sizeof(float), 256);
  MDM_LAUNCH(pad_rows_kernel, dim3(1), dim3(256), 0, s, m->out_bias_pad, m->W("output_process.poseFinal.bias"), 1, m->jf,
             m->jf_out);
  if (int rc = rt_launch_status()) return rc;
  // ---- LayerNorm folded into its consumers: in_proj(l >= 1) <- norm2(l-1), linear1(l) <- norm1(l), OutputProcess <- norm2(L-1)
  {
    const int L = m->cfg.num_layers;
    float* scratch_w = reinterpret_cast<float*>(base);
    base += align_up(std::max<size_t>(3 * (size_t)D, FF) * D * 4, 256);
    auto take_vec = [&](size_t n) { float* p = reinterpret_cast<float*>(base); base += align_up(n * 4, 256); return p; };
    auto fold_one = [&](const float* w, const float* bias, const float* gamma, const float* beta, int N, int Npad,
                        X3Weights& op, float*& cvec, float*& bvec) -> int {
      cvec = take_vec(Npad);
      bvec = take_vec(Npad);
      MDM_LAUNCH(fold_layernorm_kernel, dim3((Npad + 3) / 4), dim3(256), 0, s, w, gamma, beta, bias, scratch_w, cvec, bvec,
                 N, D, Npad);
      if (int rc = rt_launch_status()) return rc;
      return make_planes(scratch_w, N, D, op);   // stream order: the pack kernel reads scratch_w after the fold kernel
    };
    m->fold.assign(L, mdm_model::LayerFold{});
    for (int l = 0; l < L; ++l) {
      mdm_model::LayerFold& F = m->fold[l];
      if (l >= 1) {
        if (int rc = fold_one(m->L(l, "self_attn.in_proj_weight"), m->L(l, "self_attn.in_proj_bias"),
                              m->L(l - 1, "norm2.weight"), m->L(l - 1, "norm2.bias"), 3 * D, 3 * D, F.in_proj, F.c_qkv, F.b_qkv)) return rc;
      }
      if (int rc = fold_one(m->L(l, "linear1.weight"), m->L(l, "linear1.bias"), m->L(l, "norm1.weight"),
                            m->L(l, "norm1.bias"), (int)FF, (int)FF, F.linear1, F.c_1, F.b_1)) return rc;
    }
    const int jf32 = (m->jf_out + 31) / 32 * 32;
    if (int rc = fold_one(m->W("output_process.poseFinal.weight"), m->W("output_process.poseFinal.bias"),
                          m->L(L - 1, "norm2.weight"), m->L(L - 1, "norm2.bias"), m->jf, jf32, m->out_planes_f, m->c_out,
                          m->b_out)) return rc;
    // InputProcess in split precision: weight [D][jf] zero-padded along K to jf_k, then fragment-ordered planes
    MDM_LAUNCH(pad_rows_kernel, dim3(256), dim3(256), 0, s, scratch_w, m->W("input_process.poseEmbedding.weight"), D, m->jf,
               m->jf_k);
    if (int rc = rt_launch_status()) return rc;
    {
      const size_t n = x3_packed_weight_elems(D, m->jf_k);
      p16_t* hi = reinterpret_cast<p16_t*>(base);
      base += align_up(n * 4, 256);
      m->in_planes = X3Weights{hi, hi + n};
      if (int rc = launch_pack_weights(scratch_w, hi, hi + n, D, m->jf_k, s, m->range_flag)) return rc;
    }
    m->lnfold = true;
#ifdef MDM_PROBES   // A/B switch of the probe library: MDM_LNFOLD=0 runs the LayerNorms as kernels on the planes again
    if (const char* e = getenv("MDM_LNFOLD")) m->lnfold = e[0] != '0';
#endif
  }
  m->prepared = true;
  return MDM_OK;
}

int mdm_weights_in_range(mdm_model_t* m, int32_t* in_range, void* stream) {
  if (int rc = check_ready(m)) return rc;
  if (in_range == nullptr) return fail(MDM_EINVAL, "mdm_weights_in_range: null argument");
  int flag = 0;
#ifdef MDM_EMU
  (void)stream;
  flag = *m->range_flag;
#else
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemcpyAsync(&flag, m->range_flag, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
    return fail(MDM_EHIP, "mdm_weights_in_range: reading the flag back failed");
#endif
  *in_range = flag == 0 ? 1 : 0;
  return MDM_OK;
}

int mdm_set_precision(mdm_model_t* m, int32_t mode) {
  if (m == nullptr) return fail(MDM_EINVAL, "mdm_set_precision: null model");
  if (mode != MDM_PREC_F32 && mode != MDM_PREC_F16X3) return fail(MDM_EINVAL, "mdm_set_precision: unknown mode");
  if (mode == MDM_PREC_F16X3 && (m->cfg.latent_dim % X3_BK != 0 || m->cfg.ff_size % X3_BK != 0))
    return fail(MDM_EUNSUPPORTED, "f16x3 needs latent_dim and ff_size to be multiples of 32");
  m->precision = mode;
  return MDM_OK;
}

size_t mdm_workspace_bytes(const mdm_model_t* m, int32_t nseq, int32_t nframes) {
  if (m == nullptr || nseq <= 0 || nframes <= 0) return 0;
  return carve(m, nseq, nframes, nullptr).bytes;
}

int mdm_forward(mdm_model_t* m, const float* x, const int64_t* timesteps, const float* text_embed,
                const int32_t* lengths, int32_t B, int32_t T, int32_t branches, float* out, void* ws_dev,
                size_t ws_bytes, void* stream) {
  ChainGuard chain_guard(stream);
  if (int rc = check_ready(m)) return rc;
  TimeAddScope time_add(m, B, "mdm_forward");
  if (time_add.rc) return time_add.rc;
  if (m->cfg.arch != MDM_ARCH_TRANS_ENC) return fail(MDM_ESTATE, "mdm_forward: trans_dec models go through mdm_forward_dec");
  if (x == nullptr || timesteps == nullptr || out == nullptr || ws_dev == nullptr) return fail(MDM_EINVAL, "mdm_forward: null pointer");
  if (B <= 0 || T <= 0 || T + 1 > m->cfg.max_len) return fail(MDM_EINVAL, "mdm_forward: need B >= 1 and 1 <= T < the positional table's length");
  if (branches < 0 || branches > 2) return fail(MDM_EINVAL, "mdm_forward: bad branches");
  if (branches != MDM_BRANCH_UNCOND && text_embed == nullptr) return fail(MDM_EINVAL, "mdm_forward: text_embed required");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nbranch = (branches == MDM_BRANCH_BOTH) ? 2 : 1;
  const int nseq = nbranch * B, S = T + 1, D = m->cfg.latent_dim;
  Workspace ws = carve(m, nseq, T, ws_dev);
  if (ws_bytes < ws.bytes) return fail(MDM_ENOSPC, "mdm_forward: workspace too small");
  const int* len = m->cfg.mask_frames ? lengths : nullptr;
  if (branches != MDM_BRANCH_UNCOND)
    if (int rc = launch_linear(nullptr, text_embed, m->cfg.clip_dim, m->W("embed_text.weight"), m->W("embed_text.bias"), nullptr,
                               ws.cond, B, D, m->cfg.clip_dim, ACT_NONE, 0, 1.f, s)) return rc;
  const int uncond_from = (branches == MDM_BRANCH_UNCOND) ? 0 : 1;
  if (int rc = embed_tokens(m, ws, x, reinterpret_cast<const long long*>(timesteps), 0, ws.cond, B, T, nbranch,
                            uncond_from, s)) return rc;
  if (int rc = encoder(m, ws, nseq, B, S, len, s)) return rc;
  // OutputProcess, plain: every branch's tokens -> [nseq, JF, T]
  if (m->precision == MDM_PREC_F16X3)
    return outproj_x3(m, ws, nseq, B, T, nullptr, 0, out, nullptr, nullptr, NoiseSource{}, nullptr, nullptr, StepCoefs{}, s);
  RowMajorLoader al{m->W("output_process.poseFinal.weight"), D, m->jf, D};
  CfgTokenLoader bl{ws.tok, nullptr, nseq, T, S, D, nseq * T};
  OutProjEpilogue ep{};
  ep.bias = m->W("output_process.poseFinal.bias");
  ep.out = out;
  ep.T = T; ep.JF = m->jf; ep.mode = 0;
  ProfScope ps(&m->prof, MDM_PROF_OUTPROJ, 2.0 * nseq * T * (double)D * m->jf, s);
  launch_gemm_f32(al, bl, ep, m->jf, nseq * T, D, s);
  return rt_launch_status();
}

#include "decoder.h"

size_t mdm_workspace_bytes_dec(const mdm_model_t* m, int32_t nseq, int32_t pred_len, int32_t ntok) {
  if (m == nullptr || nseq <= 0 || pred_len <= 0 || ntok <= 0) return 0;
  return carve_dec(m, nseq, m->cfg.context_len + pred_len, ntok, nseq, nullptr).bytes;
}

size_t mdm_workspace_bytes_dec_loop(const mdm_model_t* m, int32_t nseq, int32_t pred_len, int32_t ntok, int32_t nsteps) {
  if (m == nullptr || nseq <= 0 || pred_len <= 0 || ntok <= 0 || nsteps <= 0) return 0;
  return carve_dec(m, nseq, m->cfg.context_len + pred_len, ntok, nseq, nullptr, nsteps, pred_len).bytes;
}

int mdm_forward_dec(mdm_model_t* m, const float* x, const float* prefix, const int64_t* timesteps, const float* text_tokens,
                    const int32_t* text_lengths, const int32_t* lengths, int32_t B, int32_t pred_len, int32_t ntok,
                    int32_t branches, float* out, void* ws_dev, size_t ws_bytes, void* stream) {
  ChainGuard chain_guard(stream);
  if (int rc = check_ready(m)) return rc;
  TimeAddScope time_add(m, B, "mdm_forward_dec");
  if (time_add.rc) return time_add.rc;
  if (x == nullptr || timesteps == nullptr || out == nullptr || ws_dev == nullptr || text_lengths == nullptr)
    return fail(MDM_EINVAL, "mdm_forward_dec: null pointer");
  if (int rc = check_dec_shapes(m, "mdm_forward_dec", prefix, B, pred_len, ntok)) return rc;
  if (branches < 0 || branches > 2) return fail(MDM_EINVAL, "mdm_forward_dec: bad branches");
  if (branches != MDM_BRANCH_UNCOND && text_tokens == nullptr) return fail(MDM_EINVAL, "mdm_forward_dec: text tokens required");
  const int nseq = ((branches == MDM_BRANCH_BOTH) ? 2 : 1) * B;
  DecWorkspace ws = carve_dec(m, nseq, m->cfg.context_len + pred_len, ntok, B, ws_dev);
  if (ws_bytes < ws.bytes) return fail(MDM_ENOSPC, "mdm_forward_dec: workspace too small");
  return decoder_pass(m, ws, x, prefix, timesteps, text_tokens, text_lengths, lengths, B, pred_len, ntok, branches, out,
                      static_cast<hipStream_t>(stream), DecHoist{});
}

int mdm_sampler_step(const float* x_t, const float* out_cond, const float* out_uncond, const float* scale,
                     const uint8_t* inpaint_mask, const float* inpaint_motion, const float* noise, float* x_prev,
                     float* x0, int32_t B, int32_t per_sample, const mdm_step_t* st, void* stream) {
  ChainGuard chain_guard(stream);
  if (x_t == nullptr || out_cond == nullptr || x_prev == nullptr || st == nullptr) return fail(MDM_EINVAL, "mdm_sampler_step: null pointer");
  if (out_uncond != nullptr && scale == nullptr) return fail(MDM_EINVAL, "mdm_sampler_step: scale required with out_uncond");
  if ((inpaint_mask == nullptr) != (inpaint_motion == nullptr)) return fail(MDM_EINVAL, "mdm_sampler_step: inpainting needs mask and motion");
  if (B <= 0 || per_sample <= 0) return fail(MDM_EINVAL, "mdm_sampler_step: bad shape");
  StepCoefs co{st->a_x0, st->a_xt, st->sigma, st->clip_denoised};
  NoiseSource ns{noise, st->seed, st->sample_base, st->draw, (uint32_t)(st->const_noise != 0)};
  const size_t total = (size_t)B * per_sample;
  const int grid = (int)std::min<size_t>((total + 255) / 256, 2048);
  MDM_LAUNCH(sampler_step_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), x_t, out_cond,
             out_uncond, scale, inpaint_mask, inpaint_motion, x_prev, x0, per_sample, B, co, ns);
  return rt_launch_status();
}

static int launch_randn(float* out, const float* init, const float* eps, float a, float s, int32_t B, int32_t per_sample,
                        uint64_t seed, uint32_t sample_base, uint32_t draw, uint32_t const_noise, void* stream) {
  if (out == nullptr || B <= 0 || per_sample <= 0) return fail(MDM_EINVAL, "mdm_randn: bad argument");
  NoiseSource ns{nullptr, seed, sample_base, draw, const_noise};
  const size_t total = (size_t)B * per_sample;
  const int grid = (int)std::min<size_t>((total + 255) / 256, 2048);
  MDM_LAUNCH(randn_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream), out, init, eps, a, s,
             per_sample, B, ns);
  return rt_launch_status();
}

int mdm_randn(float* out, const float* init, const float* eps, float a, float s, int32_t B, int32_t per_sample,
              uint64_t seed, uint32_t sample_base, uint32_t draw, void* stream) {
  ChainGuard chain_guard(stream);
  return launch_randn(out, init, eps, a, s, B, per_sample, seed, sample_base, draw, 0u, stream);
}
#include "loops.h"

#ifdef MDM_PROBES
int mdm_debug_set(int what, int value) {
  if (what == 0) g_x3_ablate = value;
  if (what == 1) g_x3_reuse_planes = value;
  if (what == 2 && (value == 4 || value == 8)) x3_waves_setting() = value;
  if (what == 3) g_ax_ablate = value;
  if (what == 4) g_f6_reference = value;
  if (what == 8) g_x3_delay = value;
  if (what == 5) g_f6_linear = value;
  if (what == 6) x3_pipe_probe() = value;
#ifndef MDM_EMU
  if (what == 9) { x3s_tl_target() = value; x3s_tl_count() = 0; }   // gemm_x3s.h timeline probe: stamp the value-th launch from now
  if (what == 10) { xb_tl_target() = value; xb_tl_count() = 0; }    // xattn_block.h timeline probe
  if (what == 11) { sb_tl_target() = value; sb_tl_count() = 0; }    // selfattn_block.h timeline probe (self and cross launches)
#endif
  return MDM_OK;
}

int mdm_debug_get(int idx, double* out) {   // ABL & 128 cycle counters of gemm_x3.h; idx < 0 resets them
#ifndef MDM_EMU
  if (idx >= 300000) {   // selfattn_block.h timeline stamps (read once at idx == 300000, then served from the host copy)
    static std::vector<unsigned long long> tl(8 * SB_TL_WGS);
    if (idx - 300000 >= 8 * SB_TL_WGS || out == nullptr) return fail(MDM_EINVAL, "mdm_debug_get: bad timeline index");
    if (idx == 300000 && (hipDeviceSynchronize() != hipSuccess ||
                          hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(g_sb_tl), tl.size() * sizeof(unsigned long long)) != hipSuccess))
      return fail(MDM_EHIP, "mdm_debug_get: reading the timeline failed");
    *out = (double)tl[idx - 300000];
    return MDM_OK;
  }
  if (idx >= 200000) {   // xattn_block.h timeline stamps (read once at idx == 200000, then served from the host copy)
    static std::vector<unsigned long long> tl(8 * XB_TL_WGS);
    if (idx - 200000 >= 8 * XB_TL_WGS || out == nullptr) return fail(MDM_EINVAL, "mdm_debug_get: bad timeline index");
    if (idx == 200000 && (hipDeviceSynchronize() != hipSuccess ||
                          hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(g_xb_tl), tl.size() * sizeof(unsigned long long)) != hipSuccess))
      return fail(MDM_EHIP, "mdm_debug_get: reading the timeline failed");
    *out = (double)tl[idx - 200000];
    return MDM_OK;
  }
  if (idx >= 100) {   // gemm_x3s.h timeline stamps (read once at idx == 100, then served from the host copy)
    static std::vector<unsigned long long> tl(4 * X3S_TL_WGS);
    if (idx - 100 >= 4 * X3S_TL_WGS || out == nullptr) return fail(MDM_EINVAL, "mdm_debug_get: bad timeline index");
    if (idx == 100 && (hipDeviceSynchronize() != hipSuccess ||
                       hipMemcpyFromSymbol(tl.data(), HIP_SYMBOL(g_x3s_tl), tl.size() * sizeof(unsigned long long)) != hipSuccess))
      return fail(MDM_EHIP, "mdm_debug_get: reading the timeline failed");
    *out = (double)tl[idx - 100];
    return MDM_OK;
  }
  unsigned long long v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (idx < 0) return hipMemcpyToSymbol(HIP_SYMBOL(g_x3_dbg), v, sizeof(v)) == hipSuccess ? MDM_OK : fail(MDM_EHIP, "mdm_debug_get: reset failed");
  if (idx >= 8 || out == nullptr) return fail(MDM_EINVAL, "mdm_debug_get: bad argument");
  if (hipMemcpyFromSymbol(v, HIP_SYMBOL(g_x3_dbg), sizeof(v)) != hipSuccess) return fail(MDM_EHIP, "mdm_debug_get: read failed");
  *out = (double)v[idx];
#else
  if (out != nullptr) *out = 0.0;
  (void)idx;
#endif
  return MDM_OK;
}
#endif

int mdm_profile_enable(mdm_model_t* m, int on) {
  if (m == nullptr) return fail(MDM_EINVAL, "mdm_profile_enable: null model");
  m->prof.on = on != 0;
  return MDM_OK;
}

int mdm_profile_read(mdm_model_t* m, int32_t category, double* total_ms, int64_t* launches, double* flops) {
  if (m == nullptr || category < 0 || category >= MDM_PROF_NUM) return fail(MDM_EINVAL, "mdm_profile_read: bad argument");
  double ms = 0.0, fl = 0.0;
  int64_t n = 0;
#ifndef MDM_EMU
  for (const auto& r : m->prof.recs) {
    if (r.cat != category) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) return fail(MDM_EHIP, "mdm_profile_read: hipEventSynchronize failed");
    float dt = 0.f;
    if (hipEventElapsedTime(&dt, r.a, r.b) != hipSuccess) return fail(MDM_EHIP, "mdm_profile_read: hipEventElapsedTime failed");
    ms += dt; fl += r.flops; ++n;
  }
#endif
  if (total_ms) *total_ms = ms;
  if (launches) *launches = n;
  if (flops) *flops = fl;
  return MDM_OK;
}

int mdm_profile_reset(mdm_model_t* m) {
  if (m == nullptr) return fail(MDM_EINVAL, "mdm_profile_reset: null model");
#ifndef MDM_EMU
  for (auto& r : m->prof.recs) { m->prof.pool.push_back(r.a); m->prof.pool.push_back(r.b); }
  m->prof.recs.clear();
#endif
  return MDM_OK;
}

int mdm_linear(const float* in, const float* w, const float* bias, const float* res, float* out, int32_t M, int32_t N,
               int32_t K, int32_t act, void* stream) {
  ChainGuard chain_guard(stream);
  if (!in || !w || !bias || !out || M <= 0 || N <= 0 || K <= 0) return fail(MDM_EINVAL, "mdm_linear: bad argument");
  return launch_linear(nullptr, in, K, w, bias, res, out, M, N, K, act, 0, 1.f, static_cast<hipStream_t>(stream));
}

size_t mdm_linear_x3_scratch_bytes(int32_t M, int32_t N, int32_t K) {
  return align_up((size_t)M * K * 4, 256) + align_up(x3_packed_weight_elems(N, K) * 4, 256);
}

int mdm_linear_x3(const float* in, const float* w, const float* bias, const float* res, float* out, int32_t M,
                      int32_t N, int32_t K, int32_t act, void* scratch, size_t scratch_bytes, void* stream) {
  ChainGuard chain_guard(stream);
  if (!in || !w || !bias || !out || !scratch || M <= 0 || N <= 0 || K <= 0) return fail(MDM_EINVAL, "mdm_linear_x3: bad argument");
  if (K % X3_BK != 0) return fail(MDM_EINVAL, "mdm_linear_x3: K must be a multiple of 32");
  if (scratch_bytes < mdm_linear_x3_scratch_bytes(M, N, K)) return fail(MDM_ENOSPC, "mdm_linear_x3: scratch too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  p16_t* ah = static_cast<p16_t*>(scratch);
  p16_t* al = ah + (size_t)M * K;
  p16_t* wh = reinterpret_cast<p16_t*>(static_cast<char*>(scratch) + align_up((size_t)M * K * 4, 256));
  p16_t* wl = wh + x3_packed_weight_elems(N, K);
  if (!g_x3_reuse_planes) {
    if (int rc = launch_split(in, ah, al, (size_t)M * K, s)) return rc;
    if (int rc = launch_pack_weights(w, wh, wl, N, K, s)) return rc;
  }
  return launch_linear_x3(nullptr, X3Operand{ah, al}, X3Weights{wh, wl}, bias, res, out, nullptr, nullptr, M, N, K, act, 0,
                          1.f, 0, s);
}

#ifdef MDM_PROBES
size_t mdm_linear_f16f6_scratch_bytes(int32_t M, int32_t N, int32_t K) {
  if (M <= 0 || N <= 0 || K <= 0 || K % 32 != 0) return 0;
  // A planes | W as fragment-ordered planes (fast kernel) | W as row-major planes (reference kernel)
  return f6_plane_bytes(M, K) + 2 * align_up(x3_packed_weight_elems(N, K) * 2, 256) + f6_plane_bytes(N, K);
}

int mdm_linear_f16f6(const float* in, const float* w, const float* bias, const float* res, float* out, int32_t M,
                     int32_t N, int32_t K, int32_t act, void* scratch, size_t scratch_bytes, void* stream) {
  ChainGuard chain_guard(stream);
  if (!in || !w || !bias || !out || !scratch || M <= 0 || N <= 0 || K <= 0) return fail(MDM_EINVAL, "mdm_linear_f16f6: bad argument");
  if (K % 32 != 0) return fail(MDM_EINVAL, "mdm_linear_f16f6: K must be a multiple of 32");
  if (act != ACT_NONE && act != ACT_GELU && act != ACT_SILU) return fail(MDM_EINVAL, "mdm_linear_f16f6: bad activation");
  if (scratch_bytes < mdm_linear_f16f6_scratch_bytes(M, N, K)) return fail(MDM_ENOSPC, "mdm_linear_f16f6: scratch too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* base = static_cast<char*>(scratch);
  const F6Planes pa = f6_carve(base, M, K);
  p16_t* wfh = reinterpret_cast<p16_t*>(base + f6_plane_bytes(M, K));
  p16_t* wfl = reinterpret_cast<p16_t*>(base + f6_plane_bytes(M, K) + align_up(x3_packed_weight_elems(N, K) * 2, 256));
  const F6Planes pw = f6_carve(base + f6_plane_bytes(M, K) + 2 * align_up(x3_packed_weight_elems(N, K) * 2, 256), N, K);
  // the production skeleton (gemm_x3_kernel<..., F6>) where its epilogues exist; else the one-wave-per-tile reference
  const bool fast = !g_f6_reference && N % 4 == 0 && ((act == ACT_NONE) || (act == ACT_GELU && res == nullptr));
  if (!g_x3_reuse_planes) {
    MDM_LAUNCH(pack_f16f6_kernel, dim3((M * (K / 32) + 255) / 256), dim3(256), 0, s, in, pa, M, K, K);
    if (int rc = rt_launch_status()) return rc;
    if (fast) {
      const int npad = (N + 31) / 32 * 32;
      MDM_LAUNCH(pack_weight_f16f6_kernel, dim3((npad * (K / 32) + 255) / 256), dim3(256), 0, s, w, wfh, wfl, N, K);
    } else {
      MDM_LAUNCH(pack_f16f6_kernel, dim3((N * (K / 32) + 255) / 256), dim3(256), 0, s, w, pw, N, K, K);
    }
    if (int rc = rt_launch_status()) return rc;
  }
  if (fast) {
    X3Epilogue ep{out, bias, res, nullptr, nullptr, nullptr, nullptr, N, 0, 1.f, QkvPlanes{}, 0, 0,
                  nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1.f, 1, 1, 1, 1.f};
    const X3Operand a{reinterpret_cast<const p16_t*>(pa.h16), reinterpret_cast<const p16_t*>(pa.rec)};
    const int rc = launch_gemm_f16f6(a, X3Weights{wfh, wfl}, ep, M, N, K, act, s);
    if (rc == -1 || rc == -3) return lds_fail(rc, "mdm_linear_f16f6");
    if (rc == -2) return fail(MDM_EUNSUPPORTED, "mdm_linear_f16f6: unsupported (activation, residual) combination");
    return rt_launch_status();
  }
  const dim3 grid((N + 31) / 32, (M + 31) / 32);
  if (act == ACT_GELU) MDM_LAUNCH(gemm_f16f6_ref_kernel<ACT_GELU>, grid, dim3(64), 0, s, pa, pw, bias, res, out, M, N, K);
  else if (act == ACT_SILU) MDM_LAUNCH(gemm_f16f6_ref_kernel<ACT_SILU>, grid, dim3(64), 0, s, pa, pw, bias, res, out, M, N, K);
  else MDM_LAUNCH(gemm_f16f6_ref_kernel<ACT_NONE>, grid, dim3(64), 0, s, pa, pw, bias, res, out, M, N, K);
  return rt_launch_status();
}
#endif

int mdm_layernorm(float* x, const float* gamma, const float* beta, int32_t rows, int32_t D, void* stream) {
  ChainGuard chain_guard(stream);
  if (!x || !gamma || !beta || rows <= 0 || D % 256 != 0) return fail(MDM_EINVAL, "mdm_layernorm: bad argument");
  return launch_layernorm(nullptr, x, gamma, beta, rows, D, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int mdm_attention(const float* qkv, float* out, const int32_t* lengths, int32_t nseq, int32_t B, int32_t S, int32_t D,
                  int32_t H, void* stream) {
  ChainGuard chain_guard(stream);
  if (!qkv || !out || nseq <= 0 || B <= 0) return fail(MDM_EINVAL, "mdm_attention: bad argument");
  return launch_attention(nullptr, qkv, out, lengths, nseq, B, S, D, H, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

size_t mdm_attention_x3_scratch_bytes(int32_t nseq, int32_t S, int32_t D) {
  if (nseq <= 0 || S <= 0 || D <= 0) return 0;
  const size_t SP = (size_t)(S + 31) / 32 * 32;
  return (size_t)nseq * SP * D * 12;
}

int mdm_attention_x3(const float* qkv, float* out, const int32_t* lengths, int32_t nseq, int32_t B, int32_t S,
                         int32_t D, int32_t H, void* scratch, size_t scratch_bytes, void* stream) {
  ChainGuard chain_guard(stream);
  if (!qkv || !out || !scratch || nseq <= 0 || B <= 0 || S <= 0 || H <= 0) return fail(MDM_EINVAL, "mdm_attention_x3: bad argument");
  if (D != H * AX_HD) return fail(MDM_EUNSUPPORTED, "attention: head_dim must be 128");
  if (scratch_bytes < mdm_attention_x3_scratch_bytes(nseq, S, D)) return fail(MDM_ENOSPC, "mdm_attention_x3: scratch too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int NKT = (S + 31) / 32, SP = 32 * NKT;
  const size_t plane = (size_t)nseq * SP * D;
  p16_t* q = static_cast<p16_t*>(scratch);
  QkvPlanes qp{q, q + plane, q + 2 * plane, q + 3 * plane, q + 4 * plane, q + 5 * plane, SP, NKT, H};
  const int grid = (int)std::min<size_t>((plane + 255) / 256, 4096);
  if (!g_x3_reuse_planes) {   // mdm_debug_set(1, 1): kernel-only timing, the planes of the previous call are reused
    MDM_LAUNCH(qkv_pack_kernel, dim3(grid), dim3(256), 0, s, qkv, qp, nseq, S, D);
    if (int rc = rt_launch_status()) return rc;
  }
  return launch_attention_x3(nullptr, qp, lengths, nseq, B, S, D, out, nullptr, nullptr, s);
}

#ifdef MDM_PROBES
// in_proj alone (the non-folded instantiation of layer 0): fp32 tokens [nseq * S][D] and in_proj weights [3D][D] -> the
// attention operand planes, written to `planes_dev` (6 planes of nseq * SP * D 16-bit elements: qh ql kh kl vh vl).
// `scratch_dev`: 4 * nseq * S * D + 12 * D * D bytes.  tools/in_proj_determinism.py compares repeated runs bit for bit.
int mdm_probe_in_proj(const float* tokens, const float* w, const float* bias, void* planes_dev, int32_t nseq, int32_t S,
                      int32_t D, void* scratch_dev, void* stream) {
  ChainGuard chain_guard(stream);
  if (!tokens || !w || !bias || !planes_dev || !scratch_dev || nseq <= 0 || S <= 0 || D % 256 != 0) return fail(MDM_EINVAL, "mdm_probe_in_proj");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t n = (size_t)nseq * S * D;
  p16_t* ah = static_cast<p16_t*>(scratch_dev);
  p16_t* al = ah + n;
  p16_t* wh = al + n;
  p16_t* wl = wh + (size_t)3 * D * D;
  if (!g_x3_reuse_planes) {
    if (int rc = launch_split(tokens, ah, al, n, s)) return rc;
    if (int rc = launch_pack_weights(w, wh, wl, 3 * D, D, s)) return rc;
  }
  const int H = D / AX_HD, NKT = (S + 31) / 32, SP = 32 * NKT;
  const size_t plane = (size_t)nseq * SP * D;
  p16_t* q = static_cast<p16_t*>(planes_dev);
  QkvPlanes qp{q, q + plane, q + 2 * plane, q + 3 * plane, q + 4 * plane, q + 5 * plane, SP, NKT, H};
  return launch_in_proj_x3(nullptr, X3Operand{ah, al}, X3Weights{wh, wl}, bias, qp, nseq, S, D, 0.08838834764831845f, s);
}
#endif

int mdm_recover_from_ric(const float* x, const float* mean, const float* stdv, float* out, int32_t B, int32_t T,
                         int32_t njoints_feat, int32_t joints, void* stream) {
  ChainGuard chain_guard(stream);
  if (!x || !mean || !stdv || !out || B <= 0 || T <= 0 || joints < 1) return fail(MDM_EINVAL, "mdm_recover_from_ric: bad argument");
  if (njoints_feat < 4 + 3 * (joints - 1)) return fail(MDM_EINVAL, "mdm_recover_from_ric: feature width too small for the joint count");
  if (T > 1024) return fail(MDM_EUNSUPPORTED, "mdm_recover_from_ric: at most 1024 frames");
  auto k = &recover_from_ric_kernel;
  MDM_LAUNCH(k, dim3(B), dim3(256), (size_t)3 * T * sizeof(float), static_cast<hipStream_t>(stream), x, mean, stdv, out, T,
             njoints_feat, joints);
  return rt_launch_status();
}

}  // extern "C"

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

namespace {

thread_local std::string g_err;
#ifdef MDM_PROBES   // libmdm_hip_probe.so only (include/mdm_hip_probe.h): process-global experiment switches
int g_x3_ablate = 0;        // gemm_x3.h ABL code
int g_x3_reuse_planes = 0;  // mdm_linear_x3 skips the operand split and reuses the planes in scratch
int g_f6_reference = 0;     // mdm_linear_f16f6 on the one-wave-per-tile reference kernel
int g_x3_delay = 0;         // gemm_x3.h, 4-wave form: start delay (x 64 cycles) of every CU's second workgroup
#else
constexpr int g_x3_ablate = 0, g_x3_reuse_planes = 0;
#endif

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#ifdef MDM_EMU
inline int rt_launch_status() { return 0; }
inline int rt_copy(void* dst, const void* src, size_t bytes, hipStream_t) { memcpy(dst, src, bytes); return 0; }
template <class K> inline int rt_allow_lds(K, size_t) { return 0; }
#else
inline int rt_launch_status() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(MDM_EHIP, std::string("kernel launch failed: ") + hipGetErrorString(e));
  return 0;
}
inline int rt_copy(void* dst, const void* src, size_t bytes, hipStream_t s) {
  hipError_t e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s);
  if (e != hipSuccess) return fail(MDM_EHIP, std::string("hipMemcpyAsync failed: ") + hipGetErrorString(e));
  return 0;
}
template <class K> inline int rt_allow_lds(K kernel, size_t bytes) {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return fail(MDM_EHIP, std::string("hipFuncSetAttribute failed: ") + hipGetErrorString(e));
  return 0;
}
#endif

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// launcher return codes -1 / -3 of the dynamic-LDS opt-in (common.h rt_dyn_lds_once)
inline int lds_fail(int rc, const char* what) {
  if (rc == -3)
    return fail(MDM_EUNSUPPORTED, std::string(what) + ": first use of this kernel instantiation while the stream is being captured into a "
                "hipGraph -- run one warm-up call of the SAME shapes (batch, frames, text tokens) outside the capture first");
  return fail(MDM_EHIP, std::string(what) + ": hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
}

}  // namespace

// Opt-in per-launch timing (mdm_profile_enable): one hipEvent pair per kernel launch, bucketed by kernel class.
struct Profiler {
  bool on = false;
#ifndef MDM_EMU
  struct Rec { int cat; hipEvent_t a, b; double flops; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  ~Profiler() {
    for (auto& r : recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : pool) (void)hipEventDestroy(e);
  }
#endif
};

struct ProfScope {   // records start on construction, stop on destruction (both on the launch stream)
#ifndef MDM_EMU
  Profiler* p; size_t idx; hipStream_t s;
  ProfScope(Profiler* prof, int cat, double flops, hipStream_t st) : p(prof && prof->on ? prof : nullptr), idx(0), s(st) {
    if (!p) return;
    Profiler::Rec r{cat, p->get(), p->get(), flops};
    (void)hipEventRecord(r.a, s);
    idx = p->recs.size();
    p->recs.push_back(r);
  }
  ~ProfScope() { if (p) (void)hipEventRecord(p->recs[idx].b, s); }
#else
  ProfScope(Profiler*, int, double, hipStream_t) {}
#endif
};

// Side streams of a model handle (probe build: the DiP window loop's concurrent sample groups, see mdm_sample_loop_dec): created
// on first use on the handle's device, joined back into the caller's stream with events before the call returns.
struct AuxStreams {
  static constexpr int kMax = 3;
#ifndef MDM_EMU
  hipStream_t s[kMax] = {};
  hipEvent_t fork = nullptr, join[kMax] = {};
  int n = 0;
  int ensure(int want) {
    if (fork == nullptr && hipEventCreateWithFlags(&fork, hipEventDisableTiming) != hipSuccess) return fail(MDM_EHIP, "hipEventCreate failed");
    for (; n < want && n < kMax; ++n)
      if (hipStreamCreateWithFlags(&s[n], hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&join[n], hipEventDisableTiming) != hipSuccess) return fail(MDM_EHIP, "hipStreamCreate failed");
    return MDM_OK;
  }
  ~AuxStreams() {
    for (int i = 0; i < kMax; ++i) {
      if (join[i] != nullptr) (void)hipEventDestroy(join[i]);
      if (s[i] != nullptr) (void)hipStreamDestroy(s[i]);
    }
    if (fork != nullptr) (void)hipEventDestroy(fork);
  }
#endif
};

struct mdm_model {
  mdm_config_t cfg;
  Profiler prof;
  AuxStreams aux;
  std::map<std::string, const float*> w;
  std::map<std::string, int64_t> expect;  // name -> numel
  bool prepared = false;
  int* range_flag = nullptr;    // device word in the const workspace: a weight left the 16-bit planes' range (mdm_prepare)
  float* w_in_pad = nullptr;    // [D][JFpad]
  float* time_table = nullptr;  // [max_len][D]
  int jf = 0, jf_pad = 0;
  int precision = MDM_PREC_F16X3;
  struct LayerPlanes { X3Weights in_proj, out_proj, linear1, linear2; };
  std::vector<LayerPlanes> planes;  // fragment-ordered hi/lo planes of the encoder weights (mdm_prepare)
  // LayerNorm folded into its consumers (gemm_x3.h X3Epilogue): gamma-scaled weight planes, column sums, folded biases
  struct LayerFold { X3Weights in_proj, linear1; float *c_qkv, *b_qkv, *c_1, *b_1; };
  std::vector<LayerFold> fold;
  // trans_dec: the same fold on fp32 weights (gemm_f32.h LnFold): in_proj(l >= 1) <- norm3(l-1), cross-attention q <- norm1(l),
  // linear1 <- norm2(l); w = W . diag(gamma), c = row sums of w, b = bias + W . beta
  struct DecFold { float *w_in, *c_in, *b_in, *w_q, *c_q, *b_q, *w_1, *c_1, *b_1; };
  std::vector<DecFold> dec_fold;
  // trans_dec: fragment-ordered fp16 hi/lo planes of the layer weights for the small X3 GEMM (gemm_f32.h X3FragB); in_proj,
  // q and linear1 from the gamma-folded copies where a LayerNorm is folded (in_proj of layer 0: the plain weight)
  struct DecPlanes { X3Weights in_proj, out_proj, q, out_proj2, linear1, linear2; };
  std::vector<DecPlanes> dec_planes;
  X3Weights in_planes{nullptr, nullptr};   // poseEmbedding.weight, K zero-padded to jf_k (f16x3 InputProcess)
  int jf_k = 0;                             // njoints*nfeats rounded up to a multiple of 32
  X3Weights out_planes_f{nullptr, nullptr};
  float *c_out = nullptr, *b_out = nullptr;
  bool lnfold = false;                      // f16x3 mode without LayerNorm kernels (set by mdm_prepare)
  X3sOptions x3s;                           // which forwards run on gemm_x3s.h's small tiles (mdm_set_option)
  int fused_xattn = 3;                      // trans_dec plane route, the cross-attention block: 2 = q projection + memory attention per
                                            // (sequence, head) (selfattn_block.h CROSS) + out_proj GEMM; 1 = one kernel (xattn_block.h);
                                            // 0 = q projection, exact-fp32 attention kernel, out_proj: three launches; 3 = by size
  bool fused_selfattn = true;               // ... and in_proj + self-attention of a (sequence, head) as one kernel (selfattn_block.h)
  bool attn_direct = false;                 // attention_x3.h DIRECT: planes from the accumulators, next item's tiles 1, 2 in front of the stores
  X3Weights out_planes{nullptr, nullptr};  // poseFinal.weight, rows padded to jf_out (f16x3 OutputProcess)
  float* out_bias_pad = nullptr;            // poseFinal.bias padded to jf_out
  int jf_out = 0;                           // njoints*nfeats rounded up to a multiple of 4

  const float* W(const std::string& k) const { return w.at(k); }
  const float* L(int layer, const char* suffix) const {
    return w.at((cfg.arch == MDM_ARCH_TRANS_DEC ? "seqTransDecoder.layers." : "seqTransEncoder.layers.") +
                std::to_string(layer) + "." + suffix);
  }
};

namespace {

struct Workspace {
  float *tok, *qkv, *att, *ffn, *cond;
  QkvPlanes qp;         // f16x3 mode: the in_proj epilogue writes Q/K/V^T planes over the qkv region
  p16_t *xah, *xal;    // folded-LayerNorm mode: planes of the post-attention pre-norm sum (alias tok)
  float *stat1, *stat2; // folded-LayerNorm mode: per-row partial (sum, sum^2) of xa / of tokh|tokl
  p16_t *tokh, *tokl;  // split planes of tok (f16x3 mode)
  p16_t *atth, *attl;  // alias att: the attention output is only consumed by the out_proj GEMM
  p16_t *ffnh, *ffnl;  // alias ffn: the GELU output is only consumed by the linear2 GEMM
  size_t bytes;
};

Workspace carve(const mdm_model* m, int nseq, int T, void* base) {
  const size_t D = m->cfg.latent_dim, FF = m->cfg.ff_size, S = (size_t)T + 1, M = (size_t)nseq * S;
  size_t off = 0;
  auto take = [&](size_t floats) {
    size_t o = off;
    off += align_up(floats * sizeof(float), 256);
    return base ? reinterpret_cast<float*>(static_cast<char*>(base) + o) : nullptr;
  };
  Workspace w;
  w.tok = take(M * D);
  const size_t NKT = (S + 31) / 32, SP = 32 * NKT;
  w.qkv = take((size_t)nseq * SP * 3 * D);  // fp32 [M][3D] (f32 mode) or six 16-bit planes of nseq*SP*D (f16x3 mode)
  w.att = take(M * D);
  w.ffn = take(M * FF);
  w.cond = take((size_t)nseq * D);
  float* tp = take(M * D);  // two 16-bit planes = one fp32 array's worth of bytes
  w.tokh = reinterpret_cast<p16_t*>(tp);
  w.tokl = tp ? w.tokh + M * D : nullptr;
  w.atth = reinterpret_cast<p16_t*>(w.att);
  w.attl = w.att ? w.atth + M * D : nullptr;
  w.ffnh = reinterpret_cast<p16_t*>(w.ffn);
  w.ffnl = w.ffn ? w.ffnh + M * FF : nullptr;
  w.xah = reinterpret_cast<p16_t*>(w.tok);
  w.xal = w.tok ? w.xah + M * D : nullptr;
  const size_t parts = (D + 127) / 128;   // per-row partial statistics: per 256 columns (gemm_x3.h) or per 128 (gemm_x3s.h)
  w.stat1 = take(M * parts * 2 + 4);   // (+ 16 bytes: gemm_x3.h stats_dma's last unit of a tile that starts 8-byte aligned)
  w.stat2 = take(M * parts * 2 + 4);
  {
    const size_t plane = (size_t)nseq * SP * D;
    p16_t* q = reinterpret_cast<p16_t*>(w.qkv);
    w.qp = QkvPlanes{q, q ? q + plane : nullptr, q ? q + 2 * plane : nullptr, q ? q + 3 * plane : nullptr,
                     q ? q + 4 * plane : nullptr, q ? q + 5 * plane : nullptr, (int)SP, (int)NKT, m->cfg.num_heads};
  }
  w.bytes = off;
  return w;
}

int launch_layernorm(Profiler* pf, float* x, const float* g, const float* b, int rows, int D, p16_t* xh, p16_t* xl,
                     hipStream_t s, bool write_f32 = true) {
  ProfScope ps(pf, MDM_PROF_LAYERNORM, 0.0, s);
  const dim3 grid((rows + 3) / 4), block(256);
  switch (D / 256) {
    case 1: { auto k = &layernorm_kernel<1>; MDM_LAUNCH(k, grid, block, 0, s, x, g, b, rows, 1e-5f, xh, xl, (int)write_f32); break; }
    case 2: { auto k = &layernorm_kernel<2>; MDM_LAUNCH(k, grid, block, 0, s, x, g, b, rows, 1e-5f, xh, xl, (int)write_f32); break; }
    case 3: { auto k = &layernorm_kernel<3>; MDM_LAUNCH(k, grid, block, 0, s, x, g, b, rows, 1e-5f, xh, xl, (int)write_f32); break; }
    case 4: { auto k = &layernorm_kernel<4>; MDM_LAUNCH(k, grid, block, 0, s, x, g, b, rows, 1e-5f, xh, xl, (int)write_f32); break; }
    default: return fail(MDM_EUNSUPPORTED, "layernorm: D must be 256, 512, 768 or 1024");
  }
  return rt_launch_status();
}

template <int NKT>
int launch_attention_t(const AttnF32Args& a, float* out, int nseq, int D, int H, p16_t* oh, p16_t* ol, hipStream_t s) {
  auto k = &attention_f32_kernel<NKT>;
  const int nqt = (a.Sq + 31) / 32;
  const size_t lds = attention_lds_bytes(NKT, nqt);
  if (int rc = rt_allow_lds(k, lds)) return rc;
  MDM_LAUNCH(k, dim3(nseq * H), dim3(64 * nqt), lds, s, a, out, D, H, oh, ol);
  return rt_launch_status();
}

// exact-fp32 attention with separate query / key-value sources (attention_f32.h AttnF32Args)
int launch_attention_args(Profiler* pf, const AttnF32Args& a, float* out, int nseq, int D, int H, p16_t* oh, p16_t* ol,
                          hipStream_t s) {
  ProfScope ps(pf, MDM_PROF_ATTENTION, 4.0 * nseq * H * (double)a.Sq * a.Sk * ATT_HD, s);
  if (D != H * ATT_HD) return fail(MDM_EUNSUPPORTED, "attention: head_dim must be 128");
  if (a.Sq < 1 || a.Sk < 1) return fail(MDM_EUNSUPPORTED, "attention: no tokens");
  if (a.Sq > 224 || a.Sk > 224) {   // streaming softmax over 32-key tiles (attention_long.h): any length
    const int nqb = al_query_blocks(a.Sq);
    MDM_LAUNCH(attention_f32_long_kernel, dim3(nseq * H * nqb), dim3(256), al_f32_lds_bytes(), s, a, out, D, H, oh, ol, nqb);
    return rt_launch_status();
  }
  switch ((a.Sk + 31) / 32) {
    case 1: return launch_attention_t<1>(a, out, nseq, D, H, oh, ol, s);
    case 2: return launch_attention_t<2>(a, out, nseq, D, H, oh, ol, s);
    case 3: return launch_attention_t<3>(a, out, nseq, D, H, oh, ol, s);
    case 4: return launch_attention_t<4>(a, out, nseq, D, H, oh, ol, s);
    case 5: return launch_attention_t<5>(a, out, nseq, D, H, oh, ol, s);
    case 6: return launch_attention_t<6>(a, out, nseq, D, H, oh, ol, s);
    default: return launch_attention_t<7>(a, out, nseq, D, H, oh, ol, s);
  }
}

// self-attention over packed qkv rows [nseq*S][3D]; `lead` tokens in front of the frames are never masked
int launch_attention(Profiler* pf, const float* qkv, float* out, const int* lengths, int nseq, int B, int S, int D,
                     int H, p16_t* oh, p16_t* ol, hipStream_t s, int lead = 1, int len_B = 0, int len_b0 = 0) {
  AttnF32Args a{qkv, 3 * D, qkv + D, qkv + 2 * D, 3 * D, S, S, lengths, lead, B};
  a.len_B = len_B;     // `lengths` covers len_B samples, this launch samples len_b0 .. len_b0 + B - 1 of them (0: exactly B)
  a.len_b0 = len_b0;
  return launch_attention_args(pf, a, out, nseq, D, H, oh, ol, s);
}

#ifdef MDM_PROBES
int g_ax_ablate = 0;   // mdm_debug_set(3, code): timing experiments on the NKT = 7 attention kernel (attention_x3.h ABL)
#endif
template <int NKT, int ABL = 0, bool DIRECT = false>
int launch_attention_x3_t(const QkvPlanes& qp, const int* lengths, int nseq, int B, int S, int D, float* out, p16_t* oh,
                          p16_t* ol, hipStream_t s, int lead) {
  auto k = &attention_x3_kernel<NKT, ABL, DIRECT>;
  const size_t lds = attention_x3_lds_bytes(NKT);
  if (int rc = rt_allow_lds(k, lds)) return rc;
  // two workgroups (query halves) per (sequence, head); the item <-> block mapping pairs blocks b and b + 8 (same XCD),
  // so the number of items is rounded up to a multiple of 8 and surplus workgroups exit
  // persistent workgroups, two per CU (the grid stays a multiple of 16 so that a workgroup keeps its query half)
  const int items = nseq * qp.H, groups = (items + 7) / 8;
  const int grid = std::min(groups * 16, std::max(16, x3_grid_limit(2) / 16 * 16));
  MDM_LAUNCH(k, dim3(grid), dim3(256), lds, s, qp, lengths, S, D, B, lead, out, oh, ol, items);
  return rt_launch_status();
}

// split-precision attention on the operand planes written by the in_proj epilogue (or qkv_pack_kernel)
// `lead` tokens in front of the frames are never masked (trans_enc: the condition token; trans_dec: none -- its `lengths` count the
// context_len prefix frames as frames)
int launch_attention_x3(Profiler* pf, const QkvPlanes& qp, const int* lengths, int nseq, int B, int S, int D, float* out,
                        p16_t* oh, p16_t* ol, hipStream_t s, int lead = 1, bool direct = false) {
  ProfScope ps(pf, MDM_PROF_ATTENTION, 4.0 * nseq * qp.H * (double)S * S * AX_HD, s);
  if (D != qp.H * AX_HD) return fail(MDM_EUNSUPPORTED, "attention: head_dim must be 128");
  if (S < 1) return fail(MDM_EUNSUPPORTED, "attention: no tokens");
  if (S > 224) {   // streaming softmax over the same operand planes (attention_long.h): any length
    if (qp.NKT != (S + 31) / 32 || qp.SP != 32 * qp.NKT) return fail(MDM_EINVAL, "attention: the operand planes do not match the sequence length");
    const int nqb = al_query_blocks(S);
    MDM_LAUNCH(attention_x3_long_kernel, dim3(nseq * qp.H * nqb), dim3(256), al_x3_lds_bytes(), s, qp, lengths, S, D, B, lead, out, oh, ol, nqb);
    return rt_launch_status();
  }
  if (direct && out == nullptr && oh != nullptr) {   // planes straight from the accumulators (attention_x3.h DIRECT)
    switch (qp.NKT) {
      case 1: return launch_attention_x3_t<1, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
      case 2: return launch_attention_x3_t<2, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
      case 3: return launch_attention_x3_t<3, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
      case 4: return launch_attention_x3_t<4, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
      case 5: return launch_attention_x3_t<5, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
      case 6: return launch_attention_x3_t<6, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
      default: return launch_attention_x3_t<7, 0, true>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    }
  }
  switch (qp.NKT) {
    case 1: return launch_attention_x3_t<1>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    case 2: return launch_attention_x3_t<2>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    case 3: return launch_attention_x3_t<3>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    case 4: return launch_attention_x3_t<4>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    case 5: return launch_attention_x3_t<5>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    case 6: return launch_attention_x3_t<6>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
    default:
#ifdef MDM_PROBES
      static const int env_abl = [] { const char* e = getenv("MDM_AX_ABL"); return e != nullptr ? atoi(e) : 0; }();   // whole-bench A/B runs
      switch (g_ax_ablate != 0 ? g_ax_ablate : env_abl) {
        case 1: return launch_attention_x3_t<7, 1>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 2: return launch_attention_x3_t<7, 2>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 3: return launch_attention_x3_t<7, 3>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 4: return launch_attention_x3_t<7, 4>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 8: return launch_attention_x3_t<7, 8>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 16: return launch_attention_x3_t<7, 16>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 32: return launch_attention_x3_t<7, 32>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 48: return launch_attention_x3_t<7, 48>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 63: return launch_attention_x3_t<7, 63>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 64: return launch_attention_x3_t<7, 64>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 128: return launch_attention_x3_t<7, 128>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        case 192: return launch_attention_x3_t<7, 192>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
        default: break;
      }
#endif
      return launch_attention_x3_t<7>(qp, lengths, nseq, B, S, D, out, oh, ol, s, lead);
  }
}

#ifdef MDM_PROBES
// TEST-ONLY mode (mdm_debug_set(5, 1)): the `f32` mode's encoder GEMMs on the f16f6 kernel, UNFUSED -- operands packed per
// call into a library-owned scratch (the one exception to "the caller owns every buffer": a debug path) -- so that the
// f16f6 arithmetic can be held against the reference's golden trajectories through the product's own seams before the fused
// path exists.  Not a performance path.
int g_f6_linear = 0;
void* g_f6_dbg_scratch = nullptr;
size_t g_f6_dbg_bytes = 0;
int launch_linear_f6_debug(Profiler* pf, const float* in, int ld_in, const float* w, const float* bias, const float* res,
                           float* out, int M, int N, int K, int act, int scale_cols, float col_scale, hipStream_t s) {
  const size_t wfrag = align_up(x3_packed_weight_elems(N, K) * 2, 256);
  const size_t need = f6_plane_bytes(M, K) + 2 * wfrag;
  if (need > g_f6_dbg_bytes) {
#ifdef MDM_EMU
    free(g_f6_dbg_scratch);
    g_f6_dbg_scratch = malloc(need);
#else
    if (hipDeviceSynchronize() != hipSuccess) return fail(MDM_EHIP, "f16f6 debug mode: synchronize failed");
    if (g_f6_dbg_scratch != nullptr) (void)hipFree(g_f6_dbg_scratch);
    if (hipMalloc(&g_f6_dbg_scratch, need) != hipSuccess) { g_f6_dbg_scratch = nullptr; g_f6_dbg_bytes = 0; return fail(MDM_EHIP, "f16f6 debug mode: hipMalloc failed"); }
#endif
    g_f6_dbg_bytes = need;
  }
  ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * M * (double)N * K, s);
  char* base = static_cast<char*>(g_f6_dbg_scratch);
  const F6Planes pa = f6_carve(base, M, K);
  p16_t* wfh = reinterpret_cast<p16_t*>(base + f6_plane_bytes(M, K));
  p16_t* wfl = reinterpret_cast<p16_t*>(base + f6_plane_bytes(M, K) + wfrag);
  MDM_LAUNCH(pack_f16f6_kernel, dim3((M * (K / 32) + 255) / 256), dim3(256), 0, s, in, pa, M, K, ld_in);
  if (int rc = rt_launch_status()) return rc;
  const int npad = (N + 31) / 32 * 32;
  MDM_LAUNCH(pack_weight_f16f6_kernel, dim3((npad * (K / 32) + 255) / 256), dim3(256), 0, s, w, wfh, wfl, N, K);
  if (int rc = rt_launch_status()) return rc;
  X3Epilogue ep{out, bias, res, nullptr, nullptr, nullptr, nullptr, N, scale_cols, col_scale, QkvPlanes{}, 0, 0,
                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1.f, 1, 1, 1, 1.f};
  const X3Operand a{reinterpret_cast<const p16_t*>(pa.h16), reinterpret_cast<const p16_t*>(pa.rec)};
  const int rc = launch_gemm_f16f6(a, X3Weights{wfh, wfl}, ep, M, N, K, act, s);
  if (rc != 0) return fail(MDM_EUNSUPPORTED, "f16f6 debug mode: launch failed");
  return rt_launch_status();
}
#endif

// x3: the split-precision arithmetic on this (fp32-in-memory) skeleton -- the DiP decoder's GEMMs in the f16x3 mode
int launch_linear(Profiler* pf, const float* in, int ld_in, const float* w, const float* bias, const float* res,
                  float* out, int M, int N, int K, int act, int scale_cols, float col_scale, hipStream_t s, bool x3 = false) {
#ifdef MDM_PROBES
  if (g_f6_linear && K % 32 == 0 && N % 4 == 0 && ld_in % 4 == 0 && (scale_cols % 256 == 0) &&
      (act == ACT_NONE || (act == ACT_GELU && res == nullptr)))
    return launch_linear_f6_debug(pf, in, ld_in, w, bias, res, out, M, N, K, act, scale_cols, col_scale, s);
#endif
  ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * M * (double)N * K, s);
  if (K % 4 != 0 || ld_in % 4 != 0) return fail(MDM_EINVAL, "linear: K and the row stride must be multiples of 4");
  RowMajorLoader al{in, ld_in, M, K};
  RowMajorLoader bl{w, K, N, K};
  LinearEpilogue ep{out, bias, res, N, act, scale_cols, col_scale, nullptr, nullptr};
  launch_gemm_f32(al, bl, ep, M, N, K, s, x3);
  return rt_launch_status();
}

// linear with LayerNorms folded in (gemm_f32.h LnLinearEpilogue): `a_ln` set = the A operand is a pre-norm sum and w / bias are
// the gamma-folded ones with column sums `colsum`; res_ln set = the residual is LN(res); ostat = where the partial statistics
// of the written rows go (or null)
// (x3: the weights come as the fragment-ordered planes `wp` of the same matrix, gemm_f32.h X3FragB)
int launch_linear_lnfold(Profiler* pf, const float* in, int ld_in, const LnFold& a_ln, const float* w, X3Weights wp,
                         const float* bias, const float* colsum, const float* res, const LnFold& res_ln, float* out,
                         float* ostat, int M, int N, int K, int act, int scale_cols, float col_scale, hipStream_t s, bool x3) {
  ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * M * (double)N * K, s);
  if (K % 16 != 0 || ld_in % 4 != 0 || N % LN_PART_COLS != 0) return fail(MDM_EINVAL, "linear (LayerNorm fold): bad K / N");
  RowMajorLoader al{in, ld_in, M, K};
  LnLinearEpilogue ep{out, bias, N, act, scale_cols, col_scale, a_ln, colsum, res, res_ln, ostat};
#ifdef MDM_PROBES   // A/B switch of the probe library: MDM_DEC_FRAGB=0 sends the layer weights through the fp32 loader again
  static const bool fragb = [] { const char* e = getenv("MDM_DEC_FRAGB"); return e == nullptr || e[0] != '0'; }();
#else
  constexpr bool fragb = true;
#endif
  if (x3 && wp.hi != nullptr && fragb) {
    X3FragB bl{wp.hi, wp.lo, (N + 31) / 32, K};
    launch_gemm_f32(al, bl, ep, M, N, K, s, true);
  } else {
    RowMajorLoader bl{w, K, N, K};
    launch_gemm_f32(al, bl, ep, M, N, K, s, x3);
  }
  return rt_launch_status();
}

// f16x3 GEMM on pre-split operands; writes fp32 `out` and/or split planes oh/ol.  seq_len > 0 tells the tiler that
// the M rows are token sequences of that length (tile = whole sequences).
int launch_linear_x3(Profiler* pf, X3Operand a, X3Weights w, const float* bias, const float* res, float* out,
                     p16_t* oh, p16_t* ol, int M, int N, int K, int act, int scale_cols, float col_scale, int seq_len,
                     hipStream_t s, X3Operand res_planes = X3Operand{nullptr, nullptr}) {
  if (K % X3_BK != 0) return fail(MDM_EINVAL, "f16x3 linear: K must be a multiple of 32");
  if (N % 4 != 0) return fail(MDM_EINVAL, "f16x3 linear: N must be a multiple of 4");
  ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * M * (double)N * K, s);
  X3Epilogue ep{out, bias, res, res_planes.hi, res_planes.lo, oh, ol, N, scale_cols, col_scale, QkvPlanes{}, 0, 0,
                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1.f, 1, 1, 1};
#ifdef MDM_PROBES
  if (g_x3_delay > 1) ep.emb_B = g_x3_delay;
#endif
  const int rc = launch_gemm_x3(a, w, ep, M, N, K, act, seq_len, s, g_x3_ablate);
  if (rc == -2) return fail(MDM_EUNSUPPORTED, "f16x3 linear: unsupported (activation, residual, output) combination");
  return rt_launch_status();
}

// in_proj in split precision: tokens -> Q (pre-scaled) / K / V^T operand planes of attention_x3.h
int launch_in_proj_x3(Profiler* pf, X3Operand a, X3Weights w, const float* bias, const QkvPlanes& qp, int nseq, int S,
                      int D, float qscale, hipStream_t s) {
  if (D % X3_BK != 0) return fail(MDM_EINVAL, "f16x3 in_proj: latent_dim must be a multiple of 32");
  ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * nseq * S * 3.0 * D * (double)D, s);
  X3Epilogue ep{nullptr, bias, nullptr, nullptr, nullptr, nullptr, nullptr, 3 * D, D, qscale, qp, S, D,
                nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 1.f, 1, 1, 1};
  const int rc = launch_gemm_x3_qkv(a, w, ep, nseq, S, D, s);
  if (rc == -2) return fail(MDM_EUNSUPPORTED, "f16x3 in_proj: sequences longer than 224 tokens");
  return rt_launch_status();
}

// One GEMM of the folded-LayerNorm encoder (gemm_x3.h launch_gemm_x3_ln kinds)
struct LnArgs {
  const float* astat = nullptr; const float* colsum = nullptr;                                   // FOLD
  X3Operand res{nullptr, nullptr}; const float* rstat = nullptr; const float* rgamma = nullptr; const float* rbeta = nullptr;  // residual
  float* ostat = nullptr;                                                                       // OSTAT
  int parts = 1; float inv_dim = 1.f;
  const float* res_f32 = nullptr; int emb_T = 1, emb_B = 1, emb_nbranch = 1;                     // EMBED (kind 5)
  bool small = false;      // the small-row-count kernel (gemm_x3s.h): the whole forward runs on one of the two kernels
  X3sShape shape{1, 1};    // ... and on ONE tile shape of it (x3s_shape(m->x3s, nseq))
  int stat_cols = 256;     // columns per partial of astat / rstat (what the PRODUCER's kernel wrote)
};
// The latency regime (gemm_x3s.h): a forward of at most MDM_OPT_SMALL_GEMM_MAX_SEQS sequences runs its GEMMs on 32 / 64-row tiles --
// and so does EVERY forward whose sequences are longer than gemm_x3.h's 224-row sequence tile (round 6: the row tiles do not care how
// long a sequence is; attention_long.h takes the attention)
inline bool use_small_gemm(const mdm_model* m, int nseq, int S) {
  return m->precision == MDM_PREC_F16X3 && m->lnfold && (nseq <= m->x3s.max_seqs || S > X3_TM) &&
         m->cfg.latent_dim % 128 == 0 && m->cfg.latent_dim % 256 == 0 && m->cfg.ff_size % 256 == 0;
}
int launch_x3_ln(Profiler* pf, int prof_cat, int kind, X3Operand a, X3Weights w, const float* bias, const LnArgs& ln,
                 float* out, p16_t* oh, p16_t* ol, const QkvPlanes* qp, int M, int N, int K, int S, int D,
                 int scale_cols, float col_scale, hipStream_t s) {
  if (K % X3_BK != 0 || N % 4 != 0) return fail(MDM_EINVAL, "f16x3 linear: K % 32 and N % 4 must be 0");
  // partial statistics per row: D / 256 on gemm_x3.h's tiles (<= 4), D / 128 on gemm_x3s.h's (<= 8); D <= 1024 (mdm_create)
  if (ln.parts < 1 || ln.parts > (ln.small ? 8 : 4)) return fail(MDM_EUNSUPPORTED, "folded LayerNorm: too many partial sums per row (D <= 1024)");
  ProfScope ps(pf, prof_cat, 2.0 * M * (double)N * K, s);
  X3Epilogue ep{out, bias, ln.res_f32, ln.res.hi, ln.res.lo, oh, ol, N, scale_cols, col_scale, qp ? *qp : QkvPlanes{}, S, D,
                ln.astat, ln.colsum, ln.rstat, ln.rgamma, ln.rbeta, ln.ostat, ln.parts, ln.inv_dim, ln.emb_T, ln.emb_B,
                ln.emb_nbranch};
  ep.stat_cols = ln.stat_cols;
  bool small = ln.small;
#ifdef MDM_PROBES   // (bisection of a misbehaving instantiation: bit k = GEMM kind k may run on the small kernel; results are wrong
                    // when producer and consumer of a row-statistics array disagree about their geometry)
  if (const char* e = getenv("MDM_X3S_KINDS")) small = small && ((atoi(e) >> kind) & 1);
#endif
  if (small) {
    // rows are grouped by sequence only where the epilogue needs (sequence, token) -- in_proj's Q / K / V^T planes, InputProcess's
    // (sample, frame); every other GEMM tiles its M rows CONTIGUOUSLY: 197 tokens are three 64-row tiles plus one of 5 rows, i.e.
    // a quarter of the workgroups of a sequence-aligned launch would do 8 % of a tile's work (B = 6: 37 row tiles instead of 48)
    const int group_rows = (kind == 0 || kind == 6) ? S : (kind == 5 ? ln.emb_T : M);
    const int rc = launch_gemm_x3s(kind, ln.shape, a, w, ep, M, N, K, group_rows, s);
    if (rc == -1 || rc == -3) return lds_fail(rc, "f16x3 linear (small tiles)");
    if (rc == -2) return fail(MDM_EUNSUPPORTED, "f16x3 linear (small tiles): unsupported shape (K must be 288 or a multiple of 256)");
#if defined(MDM_PROBES) && !defined(MDM_EMU)
    if (getenv("MDM_X3S_TRACE")) {      // bring-up: which launch faults
      fprintf(stderr, "[x3s] kind %d M %d N %d K %d launched\n", kind, M, N, K); fflush(stderr);
      const hipError_t e = hipStreamSynchronize(s);
      fprintf(stderr, "[x3s] kind %d done: %s\n", kind, hipGetErrorString(e)); fflush(stderr);
    }
#endif
    return rt_launch_status();
  }
  if (kind == 6) {   // layer 0's in_proj without a folded LayerNorm on the sequence-tile kernel (only reached by the bisection switch)
    const int rc6 = launch_gemm_x3_qkv(a, w, ep, M / S, S, D, s);
    if (rc6 != 0) return fail(MDM_EUNSUPPORTED, "f16x3 in_proj: launch failed");
    return rt_launch_status();
  }
  const int rpt = (kind == 0) ? S : x3_rows_per_tile(M, kind == 5 ? ln.emb_T : S);
  const int rc = launch_gemm_x3_ln(kind, a, w, ep, M, N, K, rpt, s);
  if (rc == -1 || rc == -3) return lds_fail(rc, "f16x3 linear");
  if (rc == -2) return fail(MDM_EUNSUPPORTED, "f16x3 linear: unsupported folded-LayerNorm GEMM kind");
  return rt_launch_status();
}

// fp32 [N][K] weights -> fragment-ordered hi/lo planes (gemm_x3.h header); K % 16 == 0
int launch_pack_weights(const float* src, p16_t* hi, p16_t* lo, int N, int K, hipStream_t s, int* overflow = nullptr) {
  if (K % 16 != 0) return fail(MDM_EINVAL, "pack_weights: K must be a multiple of 16");
  const size_t n = x3_packed_weight_elems(N, K) / 8;
  const int grid = (int)std::min<size_t>((n + 255) / 256, 4096);
  MDM_LAUNCH(pack_weight_planes_kernel, dim3(grid), dim3(256), 0, s, src, hi, lo, N, K, overflow);
  return rt_launch_status();
}

int launch_split(const float* src, p16_t* hi, p16_t* lo, size_t n, hipStream_t s) {
  if (n % 4 != 0) return fail(MDM_EINVAL, "split: element count must be a multiple of 4");
  const size_t n4 = n / 4;
  const int grid = (int)std::min<size_t>((n4 + 255) / 256, 4096);
  MDM_LAUNCH(split_planes_kernel, dim3(grid), dim3(256), 0, s, src, hi, lo, n4);
  return rt_launch_status();
}

// InputProcess in split precision (8-wave kernel): poses -> planes [B*T][jf_k] (in the dead ffn region) -> GEMM whose epilogue
// adds the positional rows and writes the frame tokens of every branch as planes.
int embed_frames_x3(mdm_model* m, const Workspace& ws, const float* x, int B, int T, int nbranch, hipStream_t s) {
  const int D = m->cfg.latent_dim, KP = m->jf_k;
  ProfScope ps(&m->prof, MDM_PROF_EMBED, 2.0 * B * T * (double)D * m->jf, s);
  p16_t* ph = reinterpret_cast<p16_t*>(ws.ffn);
  p16_t* pl = ph + (size_t)B * T * KP;
  MDM_LAUNCH(pose_to_planes_kernel, dim3((T + 31) / 32, KP / 32, B), dim3(256), 0, s, x, ph, pl, T, m->jf, KP);
  if (int rc = rt_launch_status()) return rc;
  LnArgs a;
  a.res_f32 = m->W("sequence_pos_encoder.pe");
  a.emb_T = T; a.emb_B = B; a.emb_nbranch = nbranch;
  a.small = use_small_gemm(m, nbranch * B, T + 1) && KP == 288;
  a.shape = x3s_shape(m->x3s, nbranch * B);
  return launch_x3_ln(nullptr, MDM_PROF_EMBED, 5, X3Operand{ph, pl}, m->in_planes, m->W("input_process.poseEmbedding.bias"), a,
                      nullptr, ws.tokh, ws.tokl, nullptr, B * T, D, KP, T + 1, D, 0, 1.f, s);
}
inline bool use_embed_x3(const mdm_model* m, int T) {
  // (longer sequences: the row-tile form of the same GEMM where it exists -- 263 features -- else the fp32-operand embedding below)
  return m->precision == MDM_PREC_F16X3 && x3_waves_setting() == 8 &&
         (T + 1 <= X3_TM || (use_small_gemm(m, 1, T + 1) && m->jf_k == 288));
}

// Tokens for every sequence: frame tokens via the InputProcess GEMM, token 0 via cond_token_kernel.
int embed_tokens(mdm_model* m, const Workspace& ws, const float* x, const long long* timesteps,
                 long long t_uniform_unused, const float* cond_emb, int B, int T, int nbranch,
                 int uncond_from_branch, hipStream_t s) {
  (void)t_uniform_unused;
  const int D = m->cfg.latent_dim, S = T + 1;
  PoseGatherLoader al{x, T, m->jf, B * T};
  RowMajorLoader bl{m->w_in_pad, m->jf_pad, D, m->jf_pad};
  const bool x3 = m->precision == MDM_PREC_F16X3;
  EmbedEpilogue ep{ws.tok, m->W("input_process.poseEmbedding.bias"), m->W("sequence_pos_encoder.pe"), B, T, S, D,
                   nbranch, x3 ? ws.tokh : nullptr, x3 ? ws.tokl : nullptr};
  if (use_embed_x3(m, T)) {
    if (int rc = embed_frames_x3(m, ws, x, B, T, nbranch, s)) return rc;
  } else {
    ProfScope ps(&m->prof, MDM_PROF_EMBED, 2.0 * B * T * (double)D * m->jf, s);
    launch_gemm_f32(al, bl, ep, B * T, D, m->jf_pad, s);
  }
  if (int rc = rt_launch_status()) return rc;
  ProfScope ps(&m->prof, MDM_PROF_ELEMENTWISE, 0.0, s);
  MDM_LAUNCH(cond_token_kernel, dim3(nbranch * B), dim3(128), 0, s, ws.tok, cond_emb, m->W("embed_text.bias"),
             (const float*)m->time_table, timesteps, 0, m->W("sequence_pos_encoder.pe"), B, S, D, uncond_from_branch,
             (int)m->cfg.max_len, x3 ? ws.tokh : (p16_t*)nullptr, x3 ? ws.tokl : (p16_t*)nullptr);
  return rt_launch_status();
}

// seqTransEncoder: num_layers post-norm layers over ws.tok [nseq*S, D] (in place).
int encoder(mdm_model* m, const Workspace& ws, int nseq, int B, int S, const int* lengths, hipStream_t s) {
  Profiler* pf = &m->prof;
  const int D = m->cfg.latent_dim, FF = m->cfg.ff_size, H = m->cfg.num_heads, M = nseq * S;
  const float qscale = 1.0f / sqrtf((float)(D / H));
  if (m->precision == MDM_PREC_F16X3 && m->lnfold && x3_waves_setting() == 8 && (S <= X3_TM || use_small_gemm(m, nseq, S))) {
    // No LayerNorm kernels: xb = tokh|tokl holds the layer input / the post-FFN PRE-norm sum, xa the post-attention
    // pre-norm sum, each with per-row partial (sum, sum^2) written by its producer; consumers fold the normalisation
    // (gemm_x3.h X3Epilogue).  Layer 0's input (the embedding) is not normalised: plain in_proj, plain residual.
    const X3Operand xb{ws.tokh, ws.tokl}, xa{ws.xah, ws.xal}, attp{ws.atth, ws.attl}, ffnp{ws.ffnh, ws.ffnl};
    // few sequences: the latency regime -- every GEMM of the stack on gemm_x3s.h's 32 / 64-row tiles (row statistics per 128
    // columns); else gemm_x3.h's sequence-sized tiles (per 256)
    const bool small = use_small_gemm(m, nseq, S);
    const X3sShape shape = x3s_shape(m->x3s, nseq);
    const int scols = small ? x3s_tn(shape.ncb) : 256;
    const int parts = (D + scols - 1) / scols;
    const float inv_dim = 1.0f / (float)D;
    auto LN = [&]() { LnArgs a; a.small = small; a.shape = shape; a.stat_cols = scols; a.parts = parts; a.inv_dim = inv_dim; return a; };
    // (Running the stack over two half-batches, so that every producer -> consumer hand-over stays inside the 256 MB Infinity
    // Cache, was built and measured: 1.5 % SLOWER on the same box -- profiles/r02_ab.md -- and removed.)
    // (Running in_proj -> attention and / or linear1 -> linear2 one guidance branch at a time, so that the 352 MB of Q / K / V^T
    // planes or the 207 MB of GELU planes stay inside the 256 MB Infinity Cache between producer and consumer, was built and
    // measured in round 4: attention 2 x 59.4 us against 111.5, in_proj 2 x 132.1 against 257.3, whole loop 1.0-1.5 % SLOWER on the
    // same box -- profiles/r04i_halves.md -- and removed.)
    // (Running the batch as TWO concurrent half-batch chains on two streams, each GEMM launch on half the CUs, so that one chain's
    // epilogue store bursts fall into the other's k-loops: round 5, probe-library hooks MDM_CHAIN_FREE / MDM_X3_GRID_DIV,
    // lab/probes/two_chains.py -- 3.5 % SLOWER, bit-identical results: profiles/r05l_two_chains.md.)
    for (int l = 0; l < m->cfg.num_layers; ++l) {
      const mdm_model::LayerPlanes& P = m->planes[l];
      const mdm_model::LayerFold& F = m->fold[l];
      if (l == 0 && small) {
        LnArgs a = LN();
        if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 6, xb, P.in_proj, m->L(l, "self_attn.in_proj_bias"), a, nullptr, nullptr,
                                  nullptr, &ws.qp, M, 3 * D, D, S, D, D, qscale, s)) return rc;
      } else if (l == 0) {
        if (int rc = launch_in_proj_x3(pf, xb, P.in_proj, m->L(l, "self_attn.in_proj_bias"), ws.qp, nseq, S, D, qscale, s)) return rc;
      } else {
        LnArgs a = LN(); a.astat = ws.stat2; a.colsum = F.c_qkv;
        if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 0, xb, F.in_proj, F.b_qkv, a, nullptr, nullptr, nullptr, &ws.qp, M,
                                  3 * D, D, S, D, D, qscale, s)) return rc;
      }
      if (int rc = launch_attention_x3(pf, ws.qp, lengths, nseq, B, S, D, nullptr, ws.atth, ws.attl, s, 1, m->attn_direct)) return rc;
      {  // xa = att.Wo + bo + layer input (normalised on the fly for l >= 1), + row statistics
        LnArgs a = LN(); a.res = xb; a.ostat = ws.stat1;
        if (l >= 1) { a.rstat = ws.stat2; a.rgamma = m->L(l - 1, "norm2.weight"); a.rbeta = m->L(l - 1, "norm2.bias"); }
        if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, l == 0 ? 1 : 2, attp, P.out_proj, m->L(l, "self_attn.out_proj.bias"),
                                  a, nullptr, ws.xah, ws.xal, nullptr, M, D, D, S, D, 0, 1.f, s)) return rc;
      }
      {  // ffn = gelu(LN1(xa).W1 + b1), LN1 folded
        LnArgs a = LN(); a.astat = ws.stat1; a.colsum = F.c_1;
        if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 3, xa, F.linear1, F.b_1, a, nullptr, ws.ffnh, ws.ffnl, nullptr, M, FF,
                                  D, S, D, 0, 1.f, s)) return rc;
      }
      {  // xb = ffn.W2 + b2 + LN1(xa), + row statistics
        LnArgs a = LN(); a.res = xa; a.rstat = ws.stat1; a.rgamma = m->L(l, "norm1.weight"); a.rbeta = m->L(l, "norm1.bias");
        a.ostat = ws.stat2;
        if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 2, ffnp, P.linear2, m->L(l, "linear2.bias"), a, nullptr, ws.tokh,
                                  ws.tokl, nullptr, M, D, FF, S, D, 0, 1.f, s)) return rc;
      }
    }
    return 0;   // the encoder's output is LN2(L-1)(xb): folded into OutputProcess (outproj_x3)
  }
  if (m->precision == MDM_PREC_F16X3) {
    // tok (fp32, residual stream) travels with its split planes tokh/tokl; attention and GELU outputs exist only as planes
    const X3Operand tokp{ws.tokh, ws.tokl}, attp{ws.atth, ws.attl}, ffnp{ws.ffnh, ws.ffnl};
    for (int l = 0; l < m->cfg.num_layers; ++l) {
      const mdm_model::LayerPlanes& P = m->planes[l];
      if (int rc = launch_in_proj_x3(pf, tokp, P.in_proj, m->L(l, "self_attn.in_proj_bias"), ws.qp, nseq, S, D, qscale, s)) return rc;
      if (int rc = launch_attention_x3(pf, ws.qp, lengths, nseq, B, S, D, nullptr, ws.atth, ws.attl, s, 1, m->attn_direct)) return rc;
      // the residual stream lives as planes only (value = hi + lo): the GEMM writes the pre-norm sum as fp32, LayerNorm
      // turns it back into planes and does NOT write fp32 (one 103 MB stream less per LayerNorm)
      if (int rc = launch_linear_x3(pf, attp, P.out_proj, m->L(l, "self_attn.out_proj.bias"), nullptr, ws.tok, nullptr,
                                    nullptr, M, D, D, ACT_NONE, 0, 1.f, S, s, tokp)) return rc;
      if (int rc = launch_layernorm(pf, ws.tok, m->L(l, "norm1.weight"), m->L(l, "norm1.bias"), M, D, ws.tokh, ws.tokl, s, false)) return rc;
      if (int rc = launch_linear_x3(pf, tokp, P.linear1, m->L(l, "linear1.bias"), nullptr, nullptr, ws.ffnh, ws.ffnl, M,
                                    FF, D, ACT_GELU, 0, 1.f, S, s)) return rc;
      if (int rc = launch_linear_x3(pf, ffnp, P.linear2, m->L(l, "linear2.bias"), nullptr, ws.tok, nullptr, nullptr, M, D,
                                    FF, ACT_NONE, 0, 1.f, S, s, tokp)) return rc;
      if (int rc = launch_layernorm(pf, ws.tok, m->L(l, "norm2.weight"), m->L(l, "norm2.bias"), M, D, ws.tokh, ws.tokl, s, false)) return rc;
    }
    return 0;
  }
  for (int l = 0; l < m->cfg.num_layers; ++l) {
    if (int rc = launch_linear(pf, ws.tok, D, m->L(l, "self_attn.in_proj_weight"), m->L(l, "self_attn.in_proj_bias"),
                               nullptr, ws.qkv, M, 3 * D, D, ACT_NONE, D, qscale, s)) return rc;
    if (int rc = launch_attention(pf, ws.qkv, ws.att, lengths, nseq, B, S, D, H, nullptr, nullptr, s)) return rc;
    if (int rc = launch_linear(pf, ws.att, D, m->L(l, "self_attn.out_proj.weight"), m->L(l, "self_attn.out_proj.bias"),
                               ws.tok, ws.tok, M, D, D, ACT_NONE, 0, 1.f, s)) return rc;
    if (int rc = launch_layernorm(pf, ws.tok, m->L(l, "norm1.weight"), m->L(l, "norm1.bias"), M, D, nullptr, nullptr, s)) return rc;
    if (int rc = launch_linear(pf, ws.tok, D, m->L(l, "linear1.weight"), m->L(l, "linear1.bias"), nullptr, ws.ffn, M,
                               FF, D, ACT_GELU, 0, 1.f, s)) return rc;
    if (int rc = launch_linear(pf, ws.ffn, FF, m->L(l, "linear2.weight"), m->L(l, "linear2.bias"), ws.tok, ws.tok, M,
                               D, FF, ACT_NONE, 0, 1.f, s)) return rc;
    if (int rc = launch_layernorm(pf, ws.tok, m->L(l, "norm2.weight"), m->L(l, "norm2.bias"), M, D, nullptr, nullptr, s)) return rc;
  }
  return 0;
}

// OutputProcess, split precision: every sequence's tokens x poseFinal -> fp32 rows in the (dead) qkv region, then the
// transposing / fusing tail kernel (elementwise.h outproj_finish_kernel).
int outproj_x3(mdm_model* m, const Workspace& ws, int nseq, int B, int T, const float* scale, int mode, float* out,
               float* x0_out, const float* x_t, NoiseSource noise, const unsigned char* inpaint_mask,
               const float* inpaint_motion, StepCoefs co, hipStream_t s) {
  const int D = m->cfg.latent_dim, S = T + 1, ldo = m->jf_out;
  float* out_tok = ws.qkv;
  ProfScope ps(&m->prof, MDM_PROF_OUTPROJ, 2.0 * nseq * T * (double)D * m->jf, s);
  if (m->lnfold && x3_waves_setting() == 8 && (S <= X3_TM || use_small_gemm(m, nseq, S))) {   // the final LayerNorm is folded into this GEMM
    LnArgs a; a.astat = ws.stat2; a.colsum = m->c_out; a.inv_dim = 1.0f / (float)D;
    a.small = use_small_gemm(m, nseq, S);            // (the same decision the encoder took: who wrote stat2)
    a.shape = x3s_shape(m->x3s, nseq);
    a.stat_cols = a.small ? x3s_tn(a.shape.ncb) : 256;
    a.parts = (D + a.stat_cols - 1) / a.stat_cols;
    if (int rc = launch_x3_ln(nullptr, MDM_PROF_OUTPROJ, 4, X3Operand{ws.tokh, ws.tokl}, m->out_planes_f, m->b_out, a,
                              out_tok, nullptr, nullptr, nullptr, nseq * S, ldo, D, S, D, 0, 1.f, s)) return rc;
  } else if (int rc = launch_linear_x3(nullptr, X3Operand{ws.tokh, ws.tokl}, m->out_planes, m->out_bias_pad, nullptr, out_tok,
                                       nullptr, nullptr, nseq * S, ldo, D, ACT_NONE, 0, 1.f, S, s)) return rc;
  const int nb = (mode == 1) ? B : nseq;
  MDM_LAUNCH(outproj_finish_kernel, dim3((T + 31) / 32, (m->jf + 31) / 32, nb), dim3(256), 0, s, (const float*)out_tok,
             ldo, S, T, m->jf, B, scale, mode, out, x0_out, x_t, noise, inpaint_mask, inpaint_motion, co);
  return rt_launch_status();
}

// ONE chain of this library's kernels per device (include/mdm_hip.h, "CONCURRENCY").  Why the guard exists: in the f16x3 mode
// the DiP path's small eight-wave GEMM returned rare wrong values when a workgroup of a DIFFERENT LDS-using kernel was
// co-resident on its CU -- this library's own chains on side streams, or another library's attention kernels on a foreign
// stream.  NOT cache coherence and not kernel ordering (round 2's "stale cache lines" reading was disproved in round 3:
// profiles/r03g_dip_groups.md); cause unknown; what cures it is the build without packed fp32 VALU math (mdm_build_info()).
// The guard keeps this library's own calls from overlapping each other: every exported call that enqueues kernels (i) takes a
// per-device lock for the duration of the host-side enqueue and (ii) when the previous call on this device used ANOTHER
// stream, records an event behind that stream's work and makes the caller's stream wait for it.  Same-stream callers --
// every caller the reference has -- pay one uncontended mutex and NO HIP call (round 3 recorded an event per call), so a
// single-stream loop may be captured into a hipGraph.  It cannot, of course, keep FOREIGN kernels off the device.
#ifdef MDM_EMU
struct ChainGuard { explicit ChainGuard(void*) {} };
#else
struct DeviceChain {
  std::mutex mu;
  hipEvent_t ev = nullptr;
  hipStream_t last = nullptr;
  bool has = false;
};
DeviceChain g_chain[kMaxDevices];
struct ChainGuard {
  DeviceChain& c;
  hipStream_t s;
  explicit ChainGuard(void* stream) : c(g_chain[rt_device_ordinal()]), s(static_cast<hipStream_t>(stream)) {
    c.mu.lock();
#ifdef MDM_PROBES   // lab/probes/two_chains.py: chains on different streams are NOT ordered against each other (probe library only)
    static const bool chain_free = [] { const char* e = getenv("MDM_CHAIN_FREE"); return e != nullptr && e[0] == '1'; }();
#else
    constexpr bool chain_free = false;
#endif
    if (c.has && c.last != s && !chain_free) {
      // A stream that is being CAPTURED into a hipGraph (torch.cuda.graph captures on a side stream of its own, so the warm-up
      // ran on another one) must not wait for an event recorded outside the capture: that invalidates the capture.
      // Nothing is enqueued while capturing, so there is nothing to order here; ordering the REPLAYS against other users of the
      // device is the caller's business, as for any graph (include/mdm_hip.h "hipGraph CAPTURE").
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      const bool capturing = hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
      if (!capturing) {
        // everything the previous caller's stream holds so far (its call's kernels, and whatever it enqueued since) first
        if (c.ev == nullptr && hipEventCreateWithFlags(&c.ev, hipEventDisableTiming) != hipSuccess) c.ev = nullptr;
        if (c.ev != nullptr && hipEventRecord(c.ev, c.last) == hipSuccess) (void)hipStreamWaitEvent(s, c.ev, 0);
        else (void)hipGetLastError();   // (the other stream no longer exists: its work has drained)
      }
    }
  }
  ~ChainGuard() {
    c.last = s;
    c.has = true;
    c.mu.unlock();
  }
  ChainGuard(const ChainGuard&) = delete;
  ChainGuard& operator=(const ChainGuard&) = delete;
};
#endif

int check_ready(const mdm_model* m) {
  if (m == nullptr) return fail(MDM_EINVAL, "null model");
  if (!m->prepared) return fail(MDM_ESTATE, "mdm_prepare has not been called (or weights changed since)");
  return 0;
}

}  // namespace

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
    default:
      return fail(MDM_EINVAL, "mdm_set_option: unknown key " + std::to_string(key));
  }
}

int mdm_get_option(const mdm_model_t* m, int32_t key, int32_t* value) {
  if (m == nullptr || value == nullptr) return fail(MDM_EINVAL, "mdm_get_option: null argument");
  switch (key) {
    case MDM_OPT_SMALL_GEMM_MAX_SEQS: *value = m->x3s.max_seqs; return MDM_OK;
    case MDM_OPT_SMALL_GEMM_ROW_TILES: *value = m->x3s.row_tiles; return MDM_OK;
    case MDM_OPT_DEC_FUSED_XATTN: *value = m->fused_xattn; return MDM_OK;
    case MDM_OPT_DEC_FUSED_SELFATTN: *value = m->fused_selfattn ? 1 : 0; return MDM_OK;
    case MDM_OPT_ATTN_DIRECT_OUT: *value = m->attn_direct ? 1 : 0; return MDM_OK;
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
    return 256 /* range flag */ + align_up(D * m->jf_pad * sizeof(float), 256) +
           2 * align_up((size_t)m->cfg.max_len * D * sizeof(float), 256) + (size_t)m->cfg.num_layers * (fold + planes) + outp;
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

// ---- DiP: trans_dec denoiser: the fp32 skeleton (f32 mode; MDM_OPT_SMALL_GEMM_MAX_SEQS = 0) and, in the
// default f16x3 mode, the operand-plane route with its (sequence, head) attention blocks (decoder_layers_planes) ---------------
namespace {
struct DecWorkspace {
  float *tok, *qkv, *att, *ffn, *mem, *kv, *proj;
  float *stat[2];    // [M][D/32][2] partial LayerNorm statistics of the residual stream (gemm_f32.h LnFold), ping-pong
  // the plane path (decoder_pass_planes): the residual stream as two ping-pong pairs of hi | lo operand planes; the attention
  // outputs and the GELU output as planes over att / ffn; Q / K / V^T planes over qkv
  p16_t *xh[2], *xl[2];
  p16_t *atth, *attl, *ffnh, *ffnl;
  QkvPlanes qp;
  // window loop only (nsteps > 0): what is constant over the steps of one p_sample_loop
  float *out;        // [nseq][J*F*pred_len] model output of the current step
  float *kv_text;    // [L][nseq*ntok][2D]   Wkv_l . (text part of the memory)            (no bias)
  float *kv_time;    // [L][nsteps][2D]      Wkv_l . time_table[timestep of step k] + b_kv_l
  float *time_rows;  // [nsteps][D]          the gathered time-embedding rows
  size_t bytes;
};
DecWorkspace carve_dec(const mdm_model* m, int nseq, int S, int ntok, int B, void* base, int nsteps = 0, int pred_len = 0) {
  const size_t D = m->cfg.latent_dim, FF = m->cfg.ff_size, M = (size_t)nseq * S, Mm = (size_t)nseq * ntok;
  size_t off = 0;
  auto take = [&](size_t floats) {
    size_t o = off;
    off += align_up(floats * sizeof(float), 256);
    return base ? reinterpret_cast<float*>(static_cast<char*>(base) + o) : nullptr;
  };
  DecWorkspace w;
  w.tok = take(M * D);
  const size_t NKT = ((size_t)S + 31) / 32, SP = 32 * NKT;
  w.qkv = take((size_t)nseq * SP * 3 * D);   // self-attention: packed q|k|v rows [M][3D] (or six 16-bit planes of nseq*SP*D: the
                                             // plane path); cross-attention: the projected queries [M][D]
  w.att = take(M * D);
  w.ffn = take(M * FF);
  for (int i = 0; i < 2; ++i) {   // two 16-bit planes = one fp32 array's worth of bytes
    float* tp = take(M * D);
    w.xh[i] = reinterpret_cast<p16_t*>(tp);
    w.xl[i] = tp ? w.xh[i] + M * D : nullptr;
  }
  w.atth = reinterpret_cast<p16_t*>(w.att);
  w.attl = w.att ? w.atth + M * D : nullptr;
  w.ffnh = reinterpret_cast<p16_t*>(w.ffn);
  w.ffnl = w.ffn ? w.ffnh + M * FF : nullptr;
  {
    const size_t plane = (size_t)nseq * SP * D;
    p16_t* q = reinterpret_cast<p16_t*>(w.qkv);
    w.qp = QkvPlanes{q, q ? q + plane : nullptr, q ? q + 2 * plane : nullptr, q ? q + 3 * plane : nullptr,
                     q ? q + 4 * plane : nullptr, q ? q + 5 * plane : nullptr, (int)SP, (int)NKT, m->cfg.num_heads};
  }
  w.mem = take(Mm * D);           // text memory [nseq][ntok][D]
  w.kv = take(Mm * 2 * D);        // its key | value projections of the current layer
  w.proj = take((size_t)ntok * B * D);   // embed_text(enc_text), token-major
  w.stat[0] = take(M * (D / LN_PART_COLS) * 2);
  w.stat[1] = take(M * (D / LN_PART_COLS) * 2);
  w.out = w.kv_text = w.kv_time = w.time_rows = nullptr;
  if (nsteps > 0) {
    const size_t L = m->cfg.num_layers;
    w.out = take((size_t)nseq * m->jf * pred_len);
    w.kv_text = take(L * Mm * 2 * D);
    w.kv_time = take(L * nsteps * 2 * D);
    w.time_rows = take((size_t)nsteps * D);
  }
  w.bytes = off;
  return w;
}
}  // namespace

size_t mdm_workspace_bytes_dec(const mdm_model_t* m, int32_t nseq, int32_t pred_len, int32_t ntok) {
  if (m == nullptr || nseq <= 0 || pred_len <= 0 || ntok <= 0) return 0;
  return carve_dec(m, nseq, m->cfg.context_len + pred_len, ntok, nseq, nullptr).bytes;
}

size_t mdm_workspace_bytes_dec_loop(const mdm_model_t* m, int32_t nseq, int32_t pred_len, int32_t ntok, int32_t nsteps) {
  if (m == nullptr || nseq <= 0 || pred_len <= 0 || ntok <= 0 || nsteps <= 0) return 0;
  return carve_dec(m, nseq, m->cfg.context_len + pred_len, ntok, nseq, nullptr, nsteps, pred_len).bytes;
}

namespace {
int check_dec_shapes(const mdm_model_t* m, const char* who, const float* prefix, int B, int pred_len, int ntok) {
  const int C = m->cfg.context_len, S = C + pred_len;
  const std::string w(who);
  if (m->cfg.arch != MDM_ARCH_TRANS_DEC) return fail(MDM_ESTATE, w + ": the model was created as trans_enc");
  if ((C > 0) != (prefix != nullptr)) return fail(MDM_EINVAL, w + ": prefix must be given iff context_len > 0");
  if (B <= 0 || pred_len <= 0 || S > m->cfg.max_len) return fail(MDM_EINVAL, w + ": need B >= 1 and context_len + pred_len <= the positional table's length");
  if (ntok <= 0 || ntok > 512) return fail(MDM_EINVAL, w + ": 1 <= text tokens <= 512");
  if (S > m->cfg.max_len) return fail(MDM_EINVAL, w + ": window longer than the positional table");
  return MDM_OK;
}

// One evaluation of the trans_dec denoiser.  hoist_step < 0: the stand-alone forward (memory = text + time built here from
// `timesteps`, projected per layer).  hoist_step = k >= 0: step k of a window loop -- ws.kv_text / ws.kv_time are filled,
// the memory is never materialised and the per-layer memory projection is skipped.
struct DecHoist {         // step k of a window loop: where the hoisted projections of the (whole) batch live
  int step = -1;          // < 0: not hoisted
  int nsteps = 0;
  const float* kv_text = nullptr;   // [L][nbranch * kv_B * ntok][2D]
  const float* kv_time = nullptr;   // [L][nsteps][2D]
  int kv_B = 0, kv_b0 = 0;          // this pass covers samples kv_b0 .. kv_b0 + B - 1 of kv_B
};
// The sampler update of a window-loop step, handed DOWN to the plane route: its transposing tail kernel (outproj_finish_kernel
// mode 1) then performs guidance combine + inpainting blend + clamp + posterior / DDIM update + inline Philox in place on x, exactly
// as the encoder loop's tail does -- one launch and one [nseq, J, P] round trip through memory fewer per step than
// OutputProcess -> sampler_step_kernel (same arithmetic, element for element).  `done` says whether the route applied it.
constexpr int kXattnOneKernelWgs = 144;    // MDM_OPT_DEC_FUSED_XATTN = 3: from this many 32-row tiles on, xattn_block.h's one-kernel block
struct DecTail {
  const float* scale = nullptr;      // [B] or null (single branch)
  float* x = nullptr;                // [B, J, F, P]: x_t in, x_{t-1} out
  float* x0_out = nullptr;
  NoiseSource ns{};
  const unsigned char* inpaint_mask = nullptr;
  const float* inpaint_motion = nullptr;
  StepCoefs co{};
  bool done = false;
};
// The decoder stack on 16-bit operand planes: the f16x3 mode at the sizes the reference's DiP callers run (model/mdm.py:255-270
// under sample/generate.py's autoregressive windows: 2 x 32 sequences of 20 + 40 tokens = 3,840 token rows).  That is the row
// count of the encoder's latency regime, so the six GEMMs of a layer run on gemm_x3s.h's 32 / 64-row tiles straight from planes
// (the fp32 skeleton of gemm_f32.h splits its operands inside the k-loop: 28 us per GEMM at this size), the self-attention on
// attention_x3.h's Q / K / V^T planes, and only the cross-attention -- 24 memory tokens whose keys / values are hoisted fp32 --
// stays on attention_f32.h (fp32 queries in, planes out).  All three LayerNorms of a layer are folded exactly as in the encoder
// (row statistics per 128 columns from the producer, merged by the consumer); the residual stream ping-pongs between two plane
// pairs because a GEMM cannot write the array its residual's statistics are read from.
// Frame masks (tgt_key_padding_mask, model/mdm.py:241-247 -- what DiP.md:181's `--mask_frames` recipe hands over on every call)
// travel as counts / bitmaps into attention_x3.h with lead = 0 since round 5.
// Not taken (the fp32 skeleton below stays): f32 mode, sample groups of the probe build; mdm_set_option(MDM_OPT_SMALL_GEMM_MAX_SEQS,
// 0) forces the skeleton (tests, A/B runs).  There is no upper row count: the alternative
// is not gemm_x3.h's sequence tiles (a 60-token sequence fills a quarter of one) but the skeleton, and the planes win at every
// size measured (B = 32: 544 vs 391 motions/s, B = 64: 660 vs 448; profiles/r04h_dip_planes.md).
inline bool dec_on_planes(const mdm_model* m, int M, int S, const DecHoist& hz, int B) {
  (void)M;
  return m->precision == MDM_PREC_F16X3 && m->x3s.max_seqs > 0 &&
         m->cfg.latent_dim % 256 == 0 && m->cfg.ff_size % 256 == 0 && m->out_planes_f.hi != nullptr &&
         (hz.step < 0 || (hz.kv_b0 == 0 && hz.kv_B == B));
}

int decoder_layers_planes(mdm_model_t* m, const DecWorkspace& ws, const float* x, const float* prefix, const int32_t* text_lengths,
                          const int32_t* len, int B, int pred_len, int ntok, int nbranch, float* out, hipStream_t s,
                          const DecHoist& hz, DecTail* tail) {
  const int C = m->cfg.context_len, S = C + pred_len, D = m->cfg.latent_dim, H = m->cfg.num_heads, FF = m->cfg.ff_size;
  const int nseq = nbranch * B, M = nseq * S, Mm = nseq * ntok;
  Profiler* pf = &m->prof;
  const float qscale = 1.0f / sqrtf((float)ATT_HD);
  const bool hoisted = hz.step >= 0;
  int cur = 0;   // which plane pair holds the layer input
  {  // tgt tokens: InputProcess over cat(prefix, x) + positional rows, written as planes (the fp32 copy in ws.tok is not read)
    PoseGatherLoader al{x, S, m->jf, B * S, prefix, C};
    RowMajorLoader bl{m->w_in_pad, m->jf_pad, D, m->jf_pad};
    EmbedEpilogue ep{ws.tok, m->W("input_process.poseEmbedding.bias"), m->W("sequence_pos_encoder.pe"), B, S, S, D, nbranch,
                     ws.xh[cur], ws.xl[cur], 0};
    ProfScope ps(pf, MDM_PROF_EMBED, 2.0 * B * S * (double)D * m->jf, s);
    launch_gemm_f32(al, bl, ep, B * S, D, m->jf_pad, s, true);
    if (int rc = rt_launch_status()) return rc;
  }
  const X3sShape shape = x3s_shape(m->x3s, (M + 196) / 197);   // the encoder's 32- / 64-row threshold, in its token rows
  const int scols = x3s_tn(shape.ncb), parts = (D + scols - 1) / scols;
  const float inv_dim = 1.0f / (float)D;
  auto LN = [&]() { LnArgs a; a.small = true; a.shape = shape; a.stat_cols = scols; a.parts = parts; a.inv_dim = inv_dim; return a; };
  const X3Operand attp{ws.atth, ws.attl}, ffnp{ws.ffnh, ws.ffnl};
  float* q32 = ws.tok;   // the projected cross-attention queries [M][D]
  for (int l = 0; l < m->cfg.num_layers; ++l) {
    const mdm_model::DecFold& F = m->dec_fold[l];
    const mdm_model::DecPlanes& P = m->dec_planes[l];
    p16_t *Xh = ws.xh[cur], *Xl = ws.xl[cur], *Yh = ws.xh[cur ^ 1], *Yl = ws.xl[cur ^ 1];
    const X3Operand X{Xh, Xl}, Y{Yh, Yl};
    float *sX = ws.stat[cur], *sY = ws.stat[cur ^ 1];
    // ---- Y = X' + self_attn(X'), X' = norm3(l-1)(X) (the embedded tokens for l = 0).  Sequences of at most 64 tokens (DiP: 20 + 40):
    // in_proj + attention of a (sequence, head) in one kernel (selfattn_block.h: Q / K / V^T never leave the CU); else in_proj into
    // operand planes + attention_x3.h
    if (m->fused_selfattn && selfattn_block_supported(D, S)) {
      SelfAttnArgs sa{};
      sa.x = X; sa.xstat = l == 0 ? nullptr : sX; sa.w = P.in_proj;
      sa.bias = l == 0 ? m->L(l, "self_attn.in_proj_bias") : F.b_in; sa.colsum = l == 0 ? nullptr : F.c_in;
      sa.qscale = qscale; sa.lengths = len; sa.lead = 0; sa.B = B; sa.oh = ws.atth; sa.ol = ws.attl;
      sa.M = M; sa.S = S; sa.D = D; sa.H = H; sa.stat_parts = parts; sa.stat_cols = scols; sa.inv_dim = inv_dim; sa.acc_scale = kX3AccScale;
      ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * M * 3.0 * D * (double)D + 4.0 * nseq * H * (double)S * S * ATT_HD, s);
      const int rc = launch_seqhead_block(sa, l != 0 ? 1 : 0, s);
      if (rc == -1 || rc == -3) return lds_fail(rc, "self-attention block");
      if (rc != 0) return fail(MDM_EUNSUPPORTED, "self-attention block: unsupported shape");
      if (int rc2 = rt_launch_status()) return rc2;
    } else {
    if (l == 0) {
      LnArgs a = LN();
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 6, X, P.in_proj, m->L(l, "self_attn.in_proj_bias"), a, nullptr, nullptr,
                                nullptr, &ws.qp, M, 3 * D, D, S, D, D, qscale, s)) return rc;
    } else {
      LnArgs a = LN(); a.astat = sX; a.colsum = F.c_in;
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 0, X, P.in_proj, F.b_in, a, nullptr, nullptr, nullptr, &ws.qp, M, 3 * D, D,
                                S, D, D, qscale, s)) return rc;
    }
    if (int rc = launch_attention_x3(pf, ws.qp, len, nseq, B, S, D, nullptr, ws.atth, ws.attl, s, /*lead=*/0, m->attn_direct)) return rc;
    }   // !fused self-attention
    {
      LnArgs a = LN(); a.res = X; a.ostat = sY;
      if (l >= 1) { a.rstat = sX; a.rgamma = m->L(l - 1, "norm3.weight"); a.rbeta = m->L(l - 1, "norm3.bias"); }
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, l == 0 ? 1 : 2, attp, P.out_proj, m->L(l, "self_attn.out_proj.bias"), a,
                                nullptr, Yh, Yl, nullptr, M, D, D, S, D, 0, 1.f, s)) return rc;
    }
    // ---- X = norm1(Y) + multihead_attn(norm1(Y), memory, memory).  One kernel (xattn_block.h: q projection with norm1 folded ->
    // attention over the memory -> out_proj + norm1 residual + row statistics) where its shapes are covered; else three launches:
    // fp32 queries (pre-scaled) from the small GEMM, the exact-fp32 attention kernel over k | v of the memory, the small GEMM again
    // by size (3): the one-kernel block re-reads all of Wq | Wo per 32-row tile -- it pays once its nseq * ceil(S / 32) workgroups
    // fill the chip (same-box, motions/s, one kernel vs (sequence, head) form: B = 32 per GPU / 128 tiles 598 vs 630, B = 48 / 192 tiles
    // 641 vs 605, B = 64 / 256 tiles 736 vs 707: profiles/r05c section 5)
    const int xb_wgs = nseq * ((S + XB_TR - 1) / XB_TR);
    const int xmode = m->fused_xattn == 3 ? ((xb_wgs >= kXattnOneKernelWgs && xattn_block_supported(D, ntok) && scols == 128) ? 1 : 2)
                                          : m->fused_xattn;
    // a form whose shapes are not covered falls to the OTHER fused form before the three-launch one (an explicit 1 at
    // latent_dim 768 / 1024 used to drop straight to 0 although 2 applies)
    const bool can_sh = crossattn_block_supported(D, S, ntok), can_one = xattn_block_supported(D, ntok) && scols == 128;
    const bool seqhead = (xmode == 2 && can_sh) || (xmode == 1 && !can_one && can_sh);
    const bool fused = !seqhead && xmode != 0 && can_one;
    if (!hoisted) {
      const float* wc = m->L(l, "multihead_attn.in_proj_weight");
      const float* bc = m->L(l, "multihead_attn.in_proj_bias");
      if (int rc = launch_linear(pf, ws.mem, D, wc + (size_t)D * D, bc + D, nullptr, ws.kv, Mm, 2 * D, D, ACT_NONE, 0, 1.f, s, true)) return rc;
    }
    if (seqhead) {
      SelfAttnArgs ca{};
      ca.x = Y; ca.xstat = sY; ca.w = P.q; ca.bias = F.b_q; ca.colsum = F.c_q; ca.qscale = qscale;
      ca.lengths = nullptr; ca.lead = 0; ca.B = B; ca.oh = ws.atth; ca.ol = ws.attl;
      ca.M = M; ca.S = S; ca.D = D; ca.H = H; ca.stat_parts = parts; ca.stat_cols = scols; ca.inv_dim = inv_dim; ca.acc_scale = kX3AccScale;
      if (!hoisted) {
        ca.mk = ws.kv; ca.mv = ws.kv + D; ca.kadd = ca.vadd = nullptr; ca.kv_B = 0; ca.kv_b0 = 0;
      } else {
        const float* kvt = hz.kv_text + (size_t)l * ((size_t)nbranch * hz.kv_B * ntok) * 2 * D;
        const float* row = hz.kv_time + ((size_t)l * hz.nsteps + hz.step) * 2 * D;
        ca.mk = kvt; ca.mv = kvt + D; ca.kadd = row; ca.vadd = row + D; ca.kv_B = hz.kv_B; ca.kv_b0 = hz.kv_b0;
      }
      ca.ldkv = 2 * D; ca.text_lengths = text_lengths; ca.ntok = ntok;
      {
        ProfScope ps(pf, MDM_PROF_LINEAR, 2.0 * M * (double)D * D + 4.0 * M * (double)ntok * D, s);
        const int rc = launch_seqhead_block(ca, 2, s);
        if (rc == -1 || rc == -3) return lds_fail(rc, "cross-attention (sequence, head) kernel");
        if (rc != 0) return fail(MDM_EUNSUPPORTED, "cross-attention (sequence, head) kernel: unsupported shape");
        if (int rc2 = rt_launch_status()) return rc2;
      }
      LnArgs a = LN(); a.res = Y; a.rstat = sY; a.rgamma = m->L(l, "norm1.weight"); a.rbeta = m->L(l, "norm1.bias"); a.ostat = sX;
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 2, attp, P.out_proj2, m->L(l, "multihead_attn.out_proj.bias"), a, nullptr,
                                Xh, Xl, nullptr, M, D, D, S, D, 0, 1.f, s)) return rc;
    } else if (fused) {
      XattnArgs xa{};
      xa.y = Y; xa.ystat = sY; xa.wq = P.q; xa.cq = F.c_q; xa.bq = F.b_q; xa.qscale = qscale;
      if (!hoisted) {
        xa.k = ws.kv; xa.v = ws.kv + D; xa.kadd = xa.vadd = nullptr; xa.kv_B = 0; xa.kv_b0 = 0;
      } else {
        const float* kvt = hz.kv_text + (size_t)l * ((size_t)nbranch * hz.kv_B * ntok) * 2 * D;
        const float* row = hz.kv_time + ((size_t)l * hz.nsteps + hz.step) * 2 * D;
        xa.k = kvt; xa.v = kvt + D; xa.kadd = row; xa.vadd = row + D; xa.kv_B = hz.kv_B; xa.kv_b0 = hz.kv_b0;
      }
      xa.ldkv = 2 * D; xa.text_lengths = text_lengths; xa.ntok = ntok; xa.B = B;
      xa.wo = P.out_proj2; xa.bo = m->L(l, "multihead_attn.out_proj.bias");
      xa.gamma = m->L(l, "norm1.weight"); xa.beta = m->L(l, "norm1.bias");
      xa.oh = Xh; xa.ol = Xl; xa.ostat = sX; xa.M = M; xa.S = S; xa.inv_dim = inv_dim; xa.acc_scale = kX3AccScale;
      // (profiled as ONE launch of the GEMM class: 2 D^2 per row twice + the attention contractions)
      ProfScope ps(pf, MDM_PROF_LINEAR, 4.0 * M * (double)D * D + 4.0 * M * (double)ntok * D, s);
      const int rc = launch_xattn_block(xa, D, s);
      if (rc == -1 || rc == -3) return lds_fail(rc, "cross-attention block");
      if (rc != 0) return fail(MDM_EUNSUPPORTED, "cross-attention block: unsupported shape");
      if (int rc2 = rt_launch_status()) return rc2;
    } else {
    {
      LnArgs a = LN(); a.astat = sY; a.colsum = F.c_q;
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 4, Y, P.q, F.b_q, a, q32, nullptr, nullptr, nullptr, M, D, D, S, D, D, qscale,
                                s)) return rc;
    }
    if (!hoisted) {
      const AttnF32Args a{q32, D, ws.kv, ws.kv + D, 2 * D, S, ntok, text_lengths, 0, B};
      if (int rc = launch_attention_args(pf, a, nullptr, nseq, D, H, ws.atth, ws.attl, s)) return rc;
    } else {
      const float* kvt = hz.kv_text + (size_t)l * ((size_t)nbranch * hz.kv_B * ntok) * 2 * D;
      const float* row = hz.kv_time + ((size_t)l * hz.nsteps + hz.step) * 2 * D;
      AttnF32Args a{q32, D, kvt, kvt + D, 2 * D, S, ntok, text_lengths, 0, B};
      a.kadd = row;
      a.vadd = row + D;
      a.kv_B = hz.kv_B;
      a.kv_b0 = hz.kv_b0;
      if (int rc = launch_attention_args(pf, a, nullptr, nseq, D, H, ws.atth, ws.attl, s)) return rc;
    }
    {
      LnArgs a = LN(); a.res = Y; a.rstat = sY; a.rgamma = m->L(l, "norm1.weight"); a.rbeta = m->L(l, "norm1.bias"); a.ostat = sX;
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 2, attp, P.out_proj2, m->L(l, "multihead_attn.out_proj.bias"), a, nullptr,
                                Xh, Xl, nullptr, M, D, D, S, D, 0, 1.f, s)) return rc;
    }
    }   // !fused
    // ---- Y = norm2(X) + linear2(gelu(linear1(norm2(X))))
    {
      LnArgs a = LN(); a.astat = sX; a.colsum = F.c_1;
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 3, X, P.linear1, F.b_1, a, nullptr, ws.ffnh, ws.ffnl, nullptr, M, FF, D, S, D,
                                0, 1.f, s)) return rc;
    }
    {
      LnArgs a = LN(); a.res = X; a.rstat = sX; a.rgamma = m->L(l, "norm2.weight"); a.rbeta = m->L(l, "norm2.bias"); a.ostat = sY;
      if (int rc = launch_x3_ln(pf, MDM_PROF_LINEAR, 2, ffnp, P.linear2, m->L(l, "linear2.bias"), a, nullptr, Yh, Yl, nullptr, M, D, FF, S, D, 0, 1.f, s)) return rc;
    }
    cur ^= 1;   // the layer's output (pre-norm3) is the next layer's X
  }
  // ---- OutputProcess over the completed suffix (mdm.py:278-282) with the last norm3 folded in: every token's features as an
  // fp32 row (in the dead qkv region), then the transposing tail over token rows context_len .. S-1
  float* out_tok = ws.qkv;
  const int ldo = m->jf_out;
  ProfScope ps(pf, MDM_PROF_OUTPROJ, 2.0 * nseq * pred_len * (double)D * m->jf, s);
  {
    LnArgs a = LN(); a.astat = ws.stat[cur]; a.colsum = m->c_out;
    if (int rc = launch_x3_ln(nullptr, MDM_PROF_OUTPROJ, 4, X3Operand{ws.xh[cur], ws.xl[cur]}, m->out_planes_f, m->b_out, a, out_tok,
                              nullptr, nullptr, nullptr, M, ldo, D, S, D, 0, 1.f, s)) return rc;
  }
  if (tail != nullptr) {   // window loop: the step's sampler update in the tail kernel (the guidance branches are rows b and B + b)
    MDM_LAUNCH(outproj_finish_kernel, dim3((pred_len + 31) / 32, (m->jf + 31) / 32, B), dim3(256), 0, s, (const float*)out_tok,
               ldo, S, pred_len, m->jf, B, tail->scale, 1, tail->x, tail->x0_out, (const float*)tail->x, tail->ns,
               tail->inpaint_mask, tail->inpaint_motion, tail->co);
    tail->done = true;
    return rt_launch_status();
  }
  MDM_LAUNCH(outproj_finish_kernel, dim3((pred_len + 31) / 32, (m->jf + 31) / 32, nseq), dim3(256), 0, s, (const float*)out_tok,
             ldo, S, pred_len, m->jf, B, (const float*)nullptr, 0, out, (float*)nullptr, (const float*)nullptr, NoiseSource{},
             (const unsigned char*)nullptr, (const float*)nullptr, StepCoefs{});
  return rt_launch_status();
}

int decoder_pass(mdm_model_t* m, const DecWorkspace& ws, const float* x, const float* prefix, const int64_t* timesteps,
                 const float* text_tokens, const int32_t* text_lengths, const int32_t* lengths, int B, int pred_len,
                 int ntok, int branches, float* out, hipStream_t s, const DecHoist& hz, DecTail* tail = nullptr) {
  const int C = m->cfg.context_len, S = C + pred_len, D = m->cfg.latent_dim, H = m->cfg.num_heads, FF = m->cfg.ff_size;
  const int nbranch = (branches == MDM_BRANCH_BOTH) ? 2 : 1;
  const int nseq = nbranch * B, M = nseq * S, Mm = nseq * ntok;
  Profiler* pf = &m->prof;
  const int* len = m->cfg.mask_frames ? lengths : nullptr;
  const float qscale = 1.0f / sqrtf((float)ATT_HD);
  const bool x3 = m->precision == MDM_PREC_F16X3;   // GEMM arithmetic (gemm_f32.h X3); attention and LayerNorm statistics stay fp32
  const bool hoisted = hz.step >= 0;

  // ---- text memory: embed_text over every token (cond branch), + time embedding (mdm.py:217-219)
  if (!hoisted && branches != MDM_BRANCH_UNCOND)
    if (int rc = launch_linear(nullptr, text_tokens, m->cfg.clip_dim, m->W("embed_text.weight"), m->W("embed_text.bias"),
                               nullptr, ws.proj, ntok * B, D, m->cfg.clip_dim, ACT_NONE, 0, 1.f, s)) return rc;
  if (!hoisted) {
    ProfScope ps(pf, MDM_PROF_ELEMENTWISE, 0.0, s);
    MDM_LAUNCH(text_memory_kernel, dim3(Mm), dim3(128), 0, s, ws.mem, (const float*)ws.proj, m->W("embed_text.bias"),
               (const float*)m->time_table, reinterpret_cast<const long long*>(timesteps), B, ntok, D,
               (branches == MDM_BRANCH_UNCOND) ? 0 : 1, (int)m->cfg.max_len);
    if (int rc = rt_launch_status()) return rc;
  }
  if (dec_on_planes(m, M, S, hz, B))
    return decoder_layers_planes(m, ws, x, prefix, text_lengths, len, B, pred_len, ntok, nbranch, out, s, hz,
                                 (tail != nullptr && (nbranch == 1) == (tail->scale == nullptr)) ? tail : nullptr);
  // ---- tgt tokens: InputProcess over cat(prefix, x) + positional rows (mdm.py:203-206, :239, :259-260); both branches
  {
    PoseGatherLoader al{x, S, m->jf, B * S, prefix, C};
    RowMajorLoader bl{m->w_in_pad, m->jf_pad, D, m->jf_pad};
    EmbedEpilogue ep{ws.tok, m->W("input_process.poseEmbedding.bias"), m->W("sequence_pos_encoder.pe"), B, S, S, D, nbranch,
                     nullptr, nullptr, 0};
    ProfScope ps(pf, MDM_PROF_EMBED, 2.0 * B * S * (double)D * m->jf, s);
    launch_gemm_f32(al, bl, ep, B * S, D, m->jf_pad, s, x3);
    if (int rc = rt_launch_status()) return rc;
  }
  // ---- nn.TransformerDecoder (mdm.py:265; post-norm layers, no final norm).  The three LayerNorms of a layer are folded
  // into the GEMMs around them (gemm_f32.h LnFold): ws.tok holds the PRE-norm sums y, `pend` says which LayerNorm its readers
  // have to apply (none for the embedded tokens entering layer 0); only the last norm3 runs as a kernel, for OutputProcess.
  LnFold pend{};
  int sp = 0;
  auto fold_of = [&](int l, const char* norm) {
    LnFold f;
    f.stat = ws.stat[sp];
    f.gamma = m->L(l, (std::string(norm) + ".weight").c_str());
    f.beta = m->L(l, (std::string(norm) + ".bias").c_str());
    f.parts = D / LN_PART_COLS;
    f.inv_dim = 1.0f / (float)D;
    return f;
  };
  const LnFold none{};
  for (int l = 0; l < m->cfg.num_layers; ++l) {
    const mdm_model::DecFold& F = m->dec_fold[l];
    const mdm_model::DecPlanes& P = m->dec_planes[l];
    const bool folded = pend.stat != nullptr;   // false only for the embedded tokens entering layer 0
    // x = norm1(x + self_attn(x))
    if (int rc = launch_linear_lnfold(pf, ws.tok, D, pend, folded ? F.w_in : m->L(l, "self_attn.in_proj_weight"), P.in_proj,
                                      folded ? F.b_in : m->L(l, "self_attn.in_proj_bias"), folded ? F.c_in : nullptr, nullptr,
                                      none, ws.qkv, nullptr, M, 3 * D, D, ACT_NONE, D, qscale, s, x3)) return rc;
    // (hoisted: `lengths` is the WHOLE batch's array -- counts, then the ABI-7 bitmaps -- and this pass covers samples kv_b0 ..)
    if (int rc = launch_attention(pf, ws.qkv, ws.att, len, nseq, B, S, D, H, nullptr, nullptr, s, /*lead=*/0,
                                  hoisted ? hz.kv_B : 0, hoisted ? hz.kv_b0 : 0)) return rc;
    if (int rc = launch_linear_lnfold(pf, ws.att, D, none, m->L(l, "self_attn.out_proj.weight"), P.out_proj,
                                      m->L(l, "self_attn.out_proj.bias"), nullptr, ws.tok, pend, ws.tok, ws.stat[sp ^ 1], M, D, D,
                                      ACT_NONE, 0, 1.f, s, x3)) return rc;
    sp ^= 1;
    pend = fold_of(l, "norm1");
    // x = norm2(x + multihead_attn(x, memory, memory)): q from the tokens, k | v from the memory (packed in_proj rows)
    const float* wc = m->L(l, "multihead_attn.in_proj_weight");
    const float* bc = m->L(l, "multihead_attn.in_proj_bias");
    if (int rc = launch_linear_lnfold(pf, ws.tok, D, pend, F.w_q, P.q, F.b_q, F.c_q, nullptr, none, ws.qkv, nullptr, M, D, D,
                                      ACT_NONE, D, qscale, s, x3)) return rc;
    if (!hoisted) {
      if (int rc = launch_linear(pf, ws.mem, D, wc + (size_t)D * D, bc + D, nullptr, ws.kv, Mm, 2 * D, D, ACT_NONE, 0, 1.f, s, x3)) return rc;
      const AttnF32Args a{ws.qkv, D, ws.kv, ws.kv + D, 2 * D, S, ntok, text_lengths, 0, B};
      if (int rc = launch_attention_args(pf, a, ws.att, nseq, D, H, nullptr, nullptr, s)) return rc;
    } else {
      const float* kvt = hz.kv_text + (size_t)l * ((size_t)nbranch * hz.kv_B * ntok) * 2 * D;
      const float* row = hz.kv_time + ((size_t)l * hz.nsteps + hz.step) * 2 * D;
      AttnF32Args a{ws.qkv, D, kvt, kvt + D, 2 * D, S, ntok, text_lengths, 0, B};
      a.kadd = row;
      a.vadd = row + D;
      a.kv_B = hz.kv_B;
      a.kv_b0 = hz.kv_b0;
      if (int rc = launch_attention_args(pf, a, ws.att, nseq, D, H, nullptr, nullptr, s)) return rc;
    }
    if (int rc = launch_linear_lnfold(pf, ws.att, D, none, m->L(l, "multihead_attn.out_proj.weight"), P.out_proj2,
                                      m->L(l, "multihead_attn.out_proj.bias"), nullptr, ws.tok, pend, ws.tok, ws.stat[sp ^ 1], M,
                                      D, D, ACT_NONE, 0, 1.f, s, x3)) return rc;
    sp ^= 1;
    pend = fold_of(l, "norm2");
    // x = norm3(x + linear2(gelu(linear1(x))))
    if (int rc = launch_linear_lnfold(pf, ws.tok, D, pend, F.w_1, P.linear1, F.b_1, F.c_1, nullptr, none, ws.ffn, nullptr, M, FF,
                                      D, ACT_GELU, 0, 1.f, s, x3)) return rc;
    if (int rc = launch_linear_lnfold(pf, ws.ffn, FF, none, m->L(l, "linear2.weight"), P.linear2, m->L(l, "linear2.bias"), nullptr,
                                      ws.tok, pend, ws.tok, ws.stat[sp ^ 1], M, D, FF, ACT_NONE, 0, 1.f, s, x3)) return rc;
    sp ^= 1;
    pend = fold_of(l, "norm3");
  }
  if (pend.stat != nullptr)
    if (int rc = launch_layernorm(pf, ws.tok, pend.gamma, pend.beta, M, D, nullptr, nullptr, s)) return rc;
  // ---- OutputProcess over the completed suffix (mdm.py:278-282): token rows context_len .. S-1 of every sequence
  RowMajorLoader al{m->W("output_process.poseFinal.weight"), D, m->jf, D};
  CfgTokenLoader bl{ws.tok, nullptr, nseq, pred_len, S, D, nseq * pred_len, C};
  OutProjEpilogue ep{};
  ep.bias = m->W("output_process.poseFinal.bias");
  ep.out = out;
  ep.T = pred_len; ep.JF = m->jf; ep.mode = 0;
  ProfScope ps(pf, MDM_PROF_OUTPROJ, 2.0 * nseq * pred_len * (double)D * m->jf, s);
  launch_gemm_f32(al, bl, ep, m->jf, nseq * pred_len, D, s, x3, /*weight_is_a=*/true);
  return rt_launch_status();
}
}  // namespace

int mdm_forward_dec(mdm_model_t* m, const float* x, const float* prefix, const int64_t* timesteps, const float* text_tokens,
                    const int32_t* text_lengths, const int32_t* lengths, int32_t B, int32_t pred_len, int32_t ntok,
                    int32_t branches, float* out, void* ws_dev, size_t ws_bytes, void* stream) {
  ChainGuard chain_guard(stream);
  if (int rc = check_ready(m)) return rc;
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

int mdm_sample_loop(mdm_model_t* m, const mdm_sample_params_t* p, float* x, void* ws_dev, size_t ws_bytes,
                    void* stream) {
  ChainGuard chain_guard(stream);
  if (int rc = check_ready(m)) return rc;
  if (m->cfg.arch != MDM_ARCH_TRANS_ENC) return fail(MDM_ESTATE, "mdm_sample_loop: the fused loop drives the trans_enc denoiser");
  if (p == nullptr || x == nullptr || ws_dev == nullptr) return fail(MDM_EINVAL, "mdm_sample_loop: null pointer");
  const int B = p->B, T = p->T;
  if (B <= 0 || T <= 0 || T + 1 > m->cfg.max_len) return fail(MDM_EINVAL, "mdm_sample_loop: need B >= 1 and 1 <= T < the positional table's length");
  if (p->num_timesteps <= 0 || p->start_index < 0 || p->start_index >= p->num_timesteps)
    return fail(MDM_EINVAL, "mdm_sample_loop: bad start_index / num_timesteps");
  if (!p->a_x0 || !p->a_xt || !p->sigma || !p->timestep_map) return fail(MDM_EINVAL, "mdm_sample_loop: null schedule table");
  if ((p->inpaint_mask_dev == nullptr) != (p->inpaint_motion_dev == nullptr))
    return fail(MDM_EINVAL, "mdm_sample_loop: inpainting needs mask and motion");
  const bool cfg = p->scale_dev != nullptr;
  const bool uncond_only = !cfg && (p->force_uncond || p->text_embed_dev == nullptr);
  if (cfg && p->text_embed_dev == nullptr) return fail(MDM_EINVAL, "mdm_sample_loop: CFG needs text_embed");
  if (p->num_dump > 0 && (p->dump_steps == nullptr || p->dump_dev == nullptr)) return fail(MDM_EINVAL, "mdm_sample_loop: dump buffers missing");
  for (int i = 0; i <= p->start_index; ++i)
    if (p->timestep_map[i] < 0 || p->timestep_map[i] >= m->cfg.max_len) return fail(MDM_EINVAL, "mdm_sample_loop: timestep outside the positional table");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int nbranch = cfg ? 2 : 1, nseq = nbranch * B, S = T + 1, D = m->cfg.latent_dim;
  Workspace ws = carve(m, nseq, T, ws_dev);
  if (ws_bytes < ws.bytes) return fail(MDM_ENOSPC, "mdm_sample_loop: workspace too small");
  const int* len = m->cfg.mask_frames ? p->lengths_dev : nullptr;
  const size_t per_sample = (size_t)m->jf * T;

  // step-invariant: embed_text(cond) once per loop (gaussian_diffusion.py:633-635 caches the encoder side;
  // the Linear on top is also constant across steps)
  if (!uncond_only)
    if (int rc = launch_linear(nullptr, p->text_embed_dev, m->cfg.clip_dim, m->W("embed_text.weight"), m->W("embed_text.bias"),
                               nullptr, ws.cond, B, D, m->cfg.clip_dim, ACT_NONE, 0, 1.f, s)) return rc;
  const int uncond_from = uncond_only ? 0 : 1;

  int dump_i = 0, k = 0;
  for (int i = p->start_index; i >= 0; --i, ++k) {
    // frame tokens + condition token for model timestep timestep_map[i]
    {
      PoseGatherLoader al{x, T, m->jf, B * T};
      RowMajorLoader bl{m->w_in_pad, m->jf_pad, D, m->jf_pad};
      const bool x3 = m->precision == MDM_PREC_F16X3;
      EmbedEpilogue ep{ws.tok, m->W("input_process.poseEmbedding.bias"), m->W("sequence_pos_encoder.pe"), B, T, S, D, nbranch,
                       x3 ? ws.tokh : nullptr, x3 ? ws.tokl : nullptr};
      if (use_embed_x3(m, T)) {
        if (int rc = embed_frames_x3(m, ws, x, B, T, nbranch, s)) return rc;
      } else {
        ProfScope ps(&m->prof, MDM_PROF_EMBED, 2.0 * B * T * (double)D * m->jf, s);
        launch_gemm_f32(al, bl, ep, B * T, D, m->jf_pad, s);
      }
      if (int rc = rt_launch_status()) return rc;
      ProfScope ps(&m->prof, MDM_PROF_ELEMENTWISE, 0.0, s);
      MDM_LAUNCH(cond_token_kernel, dim3(nseq), dim3(128), 0, s, ws.tok, (const float*)ws.cond,
                 m->W("embed_text.bias"), (const float*)m->time_table, (const long long*)nullptr,
                 (int)p->timestep_map[i], m->W("sequence_pos_encoder.pe"), B, S, D, uncond_from,
                 (int)m->cfg.max_len, x3 ? ws.tokh : (p16_t*)nullptr, x3 ? ws.tokl : (p16_t*)nullptr);
      if (int rc = rt_launch_status()) return rc;
    }
    if (int rc = encoder(m, ws, nseq, B, S, len, s)) return rc;
    // this step's eps: injected, or the counter-based stream -- drawn inline by the split-precision tail kernel, into the
    // (now dead) attention buffer for the exact-fp32 OutputProcess epilogue
    const bool x3mode = m->precision == MDM_PREC_F16X3;
    const float* step_noise = nullptr;
    if (p->sigma[i] != 0.f) {
      if (p->noise_dev != nullptr) step_noise = p->noise_dev + (size_t)k * B * per_sample;
      else if (!x3mode) {
        ProfScope ps(&m->prof, MDM_PROF_ELEMENTWISE, 0.0, s);
        if (int rc = launch_randn(ws.att, nullptr, nullptr, 0.f, 1.f, B, (int)per_sample, p->seed, p->sample_base,
                                  (uint32_t)(1 + k), (uint32_t)(p->const_noise != 0), stream)) return rc;
        step_noise = ws.att;
      }
    }
    // OutputProcess + CFG combine + sampler update, in place on x
    if (x3mode) {
      if (int rc = outproj_x3(m, ws, nseq, B, T, cfg ? p->scale_dev : nullptr, 1, x, (i == 0) ? p->x0_dev : nullptr, x,
                              NoiseSource{step_noise, p->seed, p->sample_base, (uint32_t)(1 + k), (uint32_t)(p->const_noise != 0)},
                              p->inpaint_mask_dev,
                              p->inpaint_motion_dev,
                              StepCoefs{p->a_x0[i], p->a_xt[i], p->sigma[i], p->clip_denoised}, s)) return rc;
    } else {
      RowMajorLoader al{m->W("output_process.poseFinal.weight"), D, m->jf, D};
      CfgTokenLoader bl{ws.tok, cfg ? p->scale_dev : nullptr, B, T, S, D, B * T};
      OutProjEpilogue ep{};
      ep.bias = m->W("output_process.poseFinal.bias");
      ep.out = x;
      ep.x0_out = (i == 0) ? p->x0_dev : nullptr;
      ep.x_t = x;
      ep.inpaint_mask = p->inpaint_mask_dev;
      ep.inpaint_motion = p->inpaint_motion_dev;
      ep.T = T; ep.JF = m->jf; ep.mode = 1;
      ep.co = StepCoefs{p->a_x0[i], p->a_xt[i], p->sigma[i], p->clip_denoised};
      ep.noise = step_noise;
      ProfScope ps(&m->prof, MDM_PROF_OUTPROJ, 2.0 * B * T * (double)D * m->jf, s);
      launch_gemm_f32(al, bl, ep, m->jf, B * T, D, s);
      if (int rc = rt_launch_status()) return rc;
    }
    if (dump_i < p->num_dump && p->dump_steps[dump_i] == k) {
      if (int rc = rt_copy(p->dump_dev + (size_t)dump_i * B * per_sample, x, (size_t)B * per_sample * sizeof(float), s)) return rc;
      ++dump_i;
    }
  }
  return MDM_OK;
}

int mdm_sample_loop_dec(mdm_model_t* m, const mdm_sample_dec_params_t* pd, float* x, void* ws_dev, size_t ws_bytes,
                        void* stream) {
  ChainGuard chain_guard(stream);
  if (int rc = check_ready(m)) return rc;
  if (pd == nullptr || x == nullptr || ws_dev == nullptr) return fail(MDM_EINVAL, "mdm_sample_loop_dec: null pointer");
  const mdm_sample_params_t* p = &pd->loop;
  const int B = p->B, P = p->T, ntok = pd->ntok;
  if (int rc = check_dec_shapes(m, "mdm_sample_loop_dec", pd->prefix_dev, B, P, ntok)) return rc;
  if (pd->text_lengths_dev == nullptr) return fail(MDM_EINVAL, "mdm_sample_loop_dec: text_lengths required");
  if (p->num_timesteps <= 0 || p->start_index < 0 || p->start_index >= p->num_timesteps)
    return fail(MDM_EINVAL, "mdm_sample_loop_dec: bad start_index / num_timesteps");
  if (!p->a_x0 || !p->a_xt || !p->sigma || !p->timestep_map) return fail(MDM_EINVAL, "mdm_sample_loop_dec: null schedule table");
  if ((p->inpaint_mask_dev == nullptr) != (p->inpaint_motion_dev == nullptr))
    return fail(MDM_EINVAL, "mdm_sample_loop_dec: inpainting needs mask and motion");
  const bool cfg = p->scale_dev != nullptr;
  const bool uncond_only = !cfg && (p->force_uncond || p->text_embed_dev == nullptr);
  if (cfg && p->text_embed_dev == nullptr) return fail(MDM_EINVAL, "mdm_sample_loop_dec: CFG needs the text tokens");
  if (p->num_dump > 0 && (p->dump_steps == nullptr || p->dump_dev == nullptr)) return fail(MDM_EINVAL, "mdm_sample_loop_dec: dump buffers missing");
  for (int i = 0; i <= p->start_index; ++i)
    if (p->timestep_map[i] < 0 || p->timestep_map[i] >= m->cfg.max_len) return fail(MDM_EINVAL, "mdm_sample_loop_dec: timestep outside the positional table");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int branches = cfg ? MDM_BRANCH_BOTH : (uncond_only ? MDM_BRANCH_UNCOND : MDM_BRANCH_COND);
  const int nbranch = cfg ? 2 : 1, nseq = nbranch * B, D = m->cfg.latent_dim, L = m->cfg.num_layers;
  const int nsteps = p->start_index + 1, Mm = nseq * ntok;
  DecWorkspace ws = carve_dec(m, nseq, m->cfg.context_len + P, ntok, B, ws_dev, nsteps, P);
  if (ws_bytes < ws.bytes) return fail(MDM_ENOSPC, "mdm_sample_loop_dec: workspace too small");
  const size_t per_sample = (size_t)m->jf * P;
  const bool x3 = m->precision == MDM_PREC_F16X3;
  Profiler* pf = &m->prof;

  // ---- once per window: what the steps share.  memory = embed_text(tokens) (cond) | bias (uncond)  +  time_emb[t]
  // (mdm.py:217-219, :262); its key | value projection of layer l is linear in the two parts:
  //   Wkv_l . memory + b = [Wkv_l . text part]  +  [Wkv_l . time_emb[t] + b]      (per token)   (per step)
  if (branches != MDM_BRANCH_UNCOND)
    if (int rc = launch_linear(nullptr, p->text_embed_dev, m->cfg.clip_dim, m->W("embed_text.weight"), m->W("embed_text.bias"),
                               nullptr, ws.proj, ntok * B, D, m->cfg.clip_dim, ACT_NONE, 0, 1.f, s)) return rc;
  {
    ProfScope ps(pf, MDM_PROF_ELEMENTWISE, 0.0, s);
    MDM_LAUNCH(text_memory_kernel, dim3(Mm), dim3(128), 0, s, ws.mem, (const float*)ws.proj, m->W("embed_text.bias"),
               (const float*)m->time_table, (const long long*)nullptr, B, ntok, D,
               (branches == MDM_BRANCH_UNCOND) ? 0 : 1, (int)m->cfg.max_len);
    if (int rc = rt_launch_status()) return rc;
  }
  for (int k = 0; k < nsteps; ++k)
    if (int rc = rt_copy(ws.time_rows + (size_t)k * D, m->time_table + (size_t)p->timestep_map[p->start_index - k] * D,
                         (size_t)D * sizeof(float), s)) return rc;
  for (int l = 0; l < L; ++l) {
    const float* wkv = m->L(l, "multihead_attn.in_proj_weight") + (size_t)D * D;
    const float* bkv = m->L(l, "multihead_attn.in_proj_bias") + D;
    if (int rc = launch_linear(pf, ws.mem, D, wkv, nullptr, nullptr, ws.kv_text + (size_t)l * Mm * 2 * D, Mm, 2 * D, D,
                               ACT_NONE, 0, 1.f, s, x3)) return rc;
    if (int rc = launch_linear(pf, ws.time_rows, D, wkv, bkv, nullptr, ws.kv_time + (size_t)l * nsteps * 2 * D, nsteps, 2 * D, D,
                               ACT_NONE, 0, 1.f, s, x3)) return rc;
  }

  // ---- the steps.  The loop is written over G sample groups (each owns the rows [g * Mg, (g + 1) * Mg) of the activation
  // buffers and reads the hoisted text K / V of the whole batch through the attention kernel's (branch, sample) remap);
  // production runs ONE group on the caller's stream.
  // PROBE BUILD ONLY (MDM_DIP_GROUPS=G): the groups' loops run CONCURRENTLY on side streams (forked behind the hoisted
  // projections, joined before returning).  Samples are independent chains and a launch at these sizes is mostly fixed cost,
  // so one group's dispatch floor / cold loads / tail hide behind another's matrix work: +3 % (two groups) on the bench.
  // It is NOT in the product because in the f16x3 mode two or four concurrent chains intermittently (a few % of the window
  // loops at four groups, more with a split-precision attention kernel) return one sequence off by 1e-4 .. 1e-1 -- never in the
  // f32 mode, never with one chain, never with the groups serialised on one stream; not root-caused (profiles/r02e_dip.md,
  // reproducer tools/repro_dip_groups.py).
  int G = 1;
#ifdef MDM_PROBES
  {
    const char* e = getenv("MDM_DIP_GROUPS");
    const int want = e != nullptr ? atoi(e) : 1;
    for (int g = std::min(std::max(want, 1), AuxStreams::kMax + 1); g >= 1; --g)
      if (B % g == 0) { G = g; break; }
  }
#endif
  hipStream_t gs[AuxStreams::kMax + 1] = {s, s, s, s};
#if !defined(MDM_EMU) && defined(MDM_PROBES)
  if (G > 1) {
    if (int rc = m->aux.ensure(G - 1)) return rc;
    if (hipEventRecord(m->aux.fork, s) != hipSuccess) return fail(MDM_EHIP, "mdm_sample_loop_dec: hipEventRecord failed");
    for (int g = 1; g < G; ++g) {
      gs[g] = m->aux.s[g - 1];
      if (hipStreamWaitEvent(gs[g], m->aux.fork, 0) != hipSuccess) return fail(MDM_EHIP, "mdm_sample_loop_dec: hipStreamWaitEvent failed");
    }
  }
#endif
  const int Bg = B / G, nseq_g = nbranch * Bg, S = m->cfg.context_len + P;
  const size_t Mg = (size_t)nseq_g * S, FFs = m->cfg.ff_size;
  int dump_i = 0, k = 0, rc_loop = MDM_OK;
  for (int i = p->start_index; i >= 0 && rc_loop == MDM_OK; --i, ++k) {
    const bool dump = dump_i < p->num_dump && p->dump_steps[dump_i] == k;
    for (int g = 0; g < G && rc_loop == MDM_OK; ++g) {
      const int b0 = g * Bg;
      const size_t xo = (size_t)b0 * per_sample;
      DecWorkspace wg = ws;
      wg.tok += g * Mg * D; wg.qkv += g * Mg * 3 * D; wg.att += g * Mg * D; wg.ffn += g * Mg * FFs;
      wg.stat[0] += g * Mg * (D / LN_PART_COLS) * 2; wg.stat[1] += g * Mg * (D / LN_PART_COLS) * 2;
      wg.out += (size_t)g * nseq_g * per_sample;
      DecHoist hz;
      hz.step = k; hz.nsteps = nsteps; hz.kv_text = ws.kv_text; hz.kv_time = ws.kv_time; hz.kv_B = B; hz.kv_b0 = b0;
      const float* prefix_g = pd->prefix_dev != nullptr ? pd->prefix_dev + (size_t)b0 * m->jf * m->cfg.context_len : nullptr;
      // CFG combine + posterior / DDIM update, in place on x (each element is read, then written, by the same lane): inside the
      // plane route's tail kernel (DecTail), else as a kernel of its own behind the denoiser
      StepCoefs co{p->a_x0[i], p->a_xt[i], p->sigma[i], p->clip_denoised};
      const float* step_noise = (p->noise_dev != nullptr && p->sigma[i] != 0.f) ? p->noise_dev + (size_t)k * B * per_sample + xo : nullptr;
      NoiseSource ns{step_noise, p->seed, p->sample_base + (uint32_t)b0, (uint32_t)(1 + k), (uint32_t)(p->const_noise != 0)};
      DecTail tail;
      tail.scale = cfg ? p->scale_dev + b0 : nullptr;
      tail.x = x + xo;
      tail.x0_out = (i == 0 && p->x0_dev != nullptr) ? p->x0_dev + xo : nullptr;
      tail.ns = ns;
      tail.inpaint_mask = p->inpaint_mask_dev != nullptr ? p->inpaint_mask_dev + xo : nullptr;
      tail.inpaint_motion = p->inpaint_motion_dev != nullptr ? p->inpaint_motion_dev + xo : nullptr;
      tail.co = co;
      rc_loop = decoder_pass(m, wg, x + xo, prefix_g, nullptr, p->text_embed_dev, pd->text_lengths_dev + b0,
                             p->lengths_dev, Bg, P, ntok, branches, wg.out, gs[g], hz, G == 1 ? &tail : nullptr);
      if (rc_loop != MDM_OK) break;
      const size_t total = (size_t)Bg * per_sample;
      const int grid = (int)std::min<size_t>((total + 255) / 256, 2048);
      if (!tail.done) {
        ProfScope ps(pf, MDM_PROF_ELEMENTWISE, 0.0, gs[g]);
        MDM_LAUNCH(sampler_step_kernel, dim3(grid), dim3(256), 0, gs[g], (const float*)(x + xo), (const float*)wg.out,
                   cfg ? (const float*)(wg.out + (size_t)Bg * per_sample) : (const float*)nullptr,
                   cfg ? p->scale_dev + b0 : (const float*)nullptr,
                   p->inpaint_mask_dev != nullptr ? p->inpaint_mask_dev + xo : (const uint8_t*)nullptr,
                   p->inpaint_motion_dev != nullptr ? p->inpaint_motion_dev + xo : (const float*)nullptr, x + xo,
                   (i == 0 && p->x0_dev != nullptr) ? p->x0_dev + xo : (float*)nullptr, (int)per_sample, Bg, co, ns);
        rc_loop = rt_launch_status();
      }
      if (rc_loop == MDM_OK && dump)
        rc_loop = rt_copy(p->dump_dev + (size_t)dump_i * B * per_sample + xo, x + xo, (size_t)Bg * per_sample * sizeof(float), gs[g]);
    }
    if (dump) ++dump_i;
  }
#if !defined(MDM_EMU) && defined(MDM_PROBES)
  for (int g = 1; g < G; ++g)   // join, also on the error path: the caller's stream must not run ahead of the side streams
    if (hipEventRecord(m->aux.join[g - 1], gs[g]) != hipSuccess || hipStreamWaitEvent(s, m->aux.join[g - 1], 0) != hipSuccess)
      return fail(MDM_EHIP, "mdm_sample_loop_dec: joining the side streams failed");
#endif
  return rc_loop;
}

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

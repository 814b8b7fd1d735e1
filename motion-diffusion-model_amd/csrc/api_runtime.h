// api_runtime.h -- part of the ONE translation unit csrc/mdm_api.hip (the C ABI of libmdm_hip.so); split out of it in round 6
// (source health, no behaviour change).  Error reporting, HIP runtime helpers, the opt-in profiler, the model handle, the one-chain-per-device guard.
#pragma once
// (included by mdm_api.hip after the kernel headers; relies on its includes and `using namespace mdm`)

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
  // mdm_set_time_add (ABI 10): `time_add_next` is the one-shot binding ([time_add_B][D], caller-owned), `time_add` what the call
  // in flight adds to the timestep embedding of its samples (TimeAddScope; null outside a call and for calls without a binding)
  const float* time_add_next = nullptr;
  int time_add_B = 0;
  const float* time_add = nullptr;
  bool dec_time_token = false;   // MDM_OPT_DEC_TIME_TOKEN: row 0 of a trans_dec sequence is the timestep embedding (emb_trans_dec)
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
  // trans_dec: the key | value rows of every layer's cross-attention in_proj stacked into ONE matrix [L * 2D][D] (+ bias [L * 2D]), so
  // that a window loop projects its text memory / its steps' time rows for all layers in one launch each (round 6: 16 launches -> 2)
  float *wkv_all = nullptr, *bkv_all = nullptr;
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

// The one-shot target embedding of mdm_set_time_add: consumed by the entry point that constructs this scope (its batch must be the
// bound one), visible to the call's kernels as m->time_add, cleared on every way out -- a failed call consumes it too.
struct TimeAddScope {
  mdm_model* m;
  int rc = 0;
  TimeAddScope(mdm_model* m_, int B, const char* who) : m(m_) {
    if (m->time_add_next == nullptr) return;
    if (m->time_add_B != B)
      rc = fail(MDM_EINVAL, std::string(who) + ": mdm_set_time_add bound " + std::to_string(m->time_add_B) + " samples, this call has " +
                                std::to_string(B));
    else
      m->time_add = m->time_add_next;
    m->time_add_next = nullptr;
    m->time_add_B = 0;
  }
  ~TimeAddScope() { m->time_add = nullptr; }
  TimeAddScope(const TimeAddScope&) = delete;
  TimeAddScope& operator=(const TimeAddScope&) = delete;
};

}  // namespace

// gemm_x3s.h -- the split-precision GEMM of gemm_x3.h for SMALL row counts: the latency regime of the encoder.
//
// Why a second kernel.  gemm_x3.h tiles the row space by whole token sequences (208 x 256 tiles, one persistent 8-wave
// workgroup per CU): at the batch sizes the reference's own callers use -- sample/generate.py:76,98 defaults to
// `--num_samples 6`, i.e. 12 sequences under classifier-free guidance; README.md:13 quotes per-call latency -- a launch is one
// tile deep on 5-30 % of the CUs and its time is that tile's serial k-loop: 33 us per GEMM launch, 73-77 ms per 50-step loop at
// B = 1 ... 10 (profiles/r04a_small_batch.md).  This kernel cuts the same contraction into 32- or 64-row x 128-column tiles,
// one NON-persistent 4-wave workgroup each, so that M = 394 rows (B = 1) already give 168 workgroups for in_proj, and keeps
// everything a tile needs in flight at once:
//   * A (activation planes [rows][K], hi | lo): a K-CHUNK of 16 (18) sub-steps x 16 k of the tile's rows lives in LDS
//     (32 KB per 32 rows), fetched by global_load_lds_dwordx4 in the 64-byte-row XOR-swizzled image of gemm_x3.h; chunk
//     c + 1 is requested into the other buffer as soon as the barrier of chunk c has passed: ONE rendezvous per 256 k;
//   * W (fragment-ordered hi | lo planes, gemm_x3.h header): straight to registers, an eight-sub-step ring of slots
//     (16 KB per wave in flight) refilled in place behind the MFMAs that consumed them (common.h gload16_refill), retired by
//     counted vmcnt waits -- the vector-memory queue retires in order across LDS-DMA pieces and register loads
//     (tools/vmcnt_order);
//   * fragment reads one sub-step ahead through untracked ds_reads with counted lgkmcnt waits (common.h lds_read16).
// Same operands, same weight planes, same epilogue algebra (X3Epilogue: folded LayerNorm, Q / K / V^T operand planes,
// GELU, plane residuals, row statistics, InputProcess / OutputProcess forms) as gemm_x3.h, so a forward may run on either
// kernel; the row statistics a producer leaves are per 128 columns here (X3Epilogue::stat_cols).
// Replaces the same reference calls as gemm_x3.h (model/mdm.py:77-84 `addmm`s under nn.TransformerEncoderLayer,
// :343-349 InputProcess, :372-386 OutputProcess) for nseq <= x3s_max_seqs().
#pragma once
#include "gemm_x3.h"

namespace mdm {

constexpr int X3S_TN = 128;          // columns per tile: 4 waves x 32
constexpr int X3S_WAVES = 4;
constexpr int X3S_WDEPTH = 8;        // W sub-steps in flight per wave (hi + lo fragment each: 64 VGPRs)
constexpr int x3s_buf_bytes(int rt, int nsub) { return nsub * 32 * rt * 64; }        // one K-chunk of A: hi | lo, 64-byte rows per 32 k
constexpr int x3s_patch_base(int rt, int nsub, bool multi) { return (multi ? 2 : 1) * x3s_buf_bytes(rt, nsub); }
constexpr int x3s_tab_base(int rt, int nsub, bool multi) { return x3s_patch_base(rt, nsub, multi) + X3S_WAVES * X3_PATCH_BYTES; }
constexpr int x3s_part_base(int rt, int nsub, bool multi) { return x3s_tab_base(rt, nsub, multi) + 32 * rt * 8; }
constexpr int x3s_lds_bytes(int rt, int nsub, bool multi) { return x3s_part_base(rt, nsub, multi) + X3S_WAVES * 32 * rt * 8; }

// RT: 32-row sub-tiles per tile (1 or 2).  NSUB: 16-deep k sub-steps per chunk (even); K = NSUB * 16 * nchunks.
// MULTI: more than one chunk (double-buffered A).  The other flags are gemm_x3_kernel's.
// Rows are GROUPED (group_rows = tokens of a sequence; InputProcess: frames of a sample): a tile never straddles two groups, so
// that the in_proj epilogue's (sequence, token) and the EMBED epilogue's (sample, frame) are tile-uniform / row-affine.
template <int RT, int NSUB, bool MULTI, int ACT, int RES, bool OUT_F32, bool OUT_PLANES, bool OUT_QKV, bool FOLD, bool OSTAT,
          bool EMBED>
__global__ __launch_bounds__(64 * X3S_WAVES, 2) void gemm_x3s_kernel(X3Operand A, X3Weights W, X3Epilogue ep, int M, int N, int K,
                                                                    int group_rows, int tiles_per_group, int tiles_n,
                                                                    int total) {
  MDM_DYN_SMEM(unsigned char, lds);
  static_assert(NSUB % 2 == 0 && (RT == 1 || RT == 2), "tile shape");
  constexpr int TR = 32 * RT, D = X3S_WDEPTH;
  constexpr int BUF = x3s_buf_bytes(RT, NSUB);
  constexpr int PW = NSUB * RT / 2;                      // LDS-DMA pieces (1 KB) per wave and chunk
  static_assert(2 * NSUB * RT % X3S_WAVES == 0, "pieces must divide among the waves");
  static_assert(2 * (D - 1) + PW <= 63, "vmcnt range");
  constexpr bool LN_TABS = FOLD || RES == 3;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef MDM_EMU
  const int wid = tid >> 6;
#else
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int r = lane & 31, h = lane >> 5;

  const int lid = xcd_remap((int)blockIdx.x, total);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int grp = tile_m / tiles_per_group, tig = tile_m - grp * tiles_per_group;
  const int m0 = grp * group_rows + tig * TR;                       // first row of the tile
  const int rows_valid = min(TR, group_rows - tig * TR);            // rows of the tile inside its group
  const int n0 = tile_n * X3S_TN;
  const int nchunks = K / (NSUB * 16);

  // ---- A stream: chunk image = for 32-k block ms, plane p, 16-row group g: 1 KB (16 rows x 64 B); lane -> (row = lane >> 2,
  // stored 16-byte chunk = lane & 3) fetches the logical chunk (lane & 3) ^ ((row >> 2) & 3) (gemm_x3.h: the swizzle is applied
  // to the per-lane SOURCE address and to the fragment reads).  Piece q = wid + 4 i: g = q % (2 RT), p = (q / (2 RT)) % 2,
  // ms = q / (4 RT).  Rows past the matrix are clamped (loaded, never stored).
  const int schunk = (lane & 3) ^ ((lane >> 4) & 3);
  auto issue_chunk = [&](int c, int buf) {
#pragma unroll
    for (int i = 0; i < PW; ++i) {
      const int q = wid + X3S_WAVES * i;
      const int g = q % (2 * RT), p = (q / (2 * RT)) % 2, ms = q / (4 * RT);
      const int arow = min(m0 + g * 16 + (lane >> 2), M - 1);
      const p16_t* src = (p ? A.lo : A.hi) + (size_t)arow * K + (size_t)c * (NSUB * 16) + ms * 32 + schunk * 8;
      glds16(src, lds + buf * BUF + ((ms * 2 + p) * 2 * RT + g) * 1024);
    }
  };
  // ---- W stream: this wave's fragments of 16-deep sub-step `gj` (global index over the whole K): hi and lo, 1 KB each
  // (a wave whose 32 columns lie past the padded weight rows -- OutputProcess: N = 264 -> 288 packed rows, 384 tile columns --
  // re-reads the last block: its results are never stored)
  const uint32_t wbase = (uint32_t)min((n0 >> 5) + wid, (N + 31) / 32 - 1) * (uint32_t)(K / 16) * 512u + (uint32_t)lane * 8u;
  const int nsub_total = K / 16;
  p16x8 wsh[D] = {}, wsl[D] = {};     // (zero: the first refill formally reads its slot)
  auto issue_w = [&](auto slot_tag, int gj) __attribute__((always_inline)) {
    constexpr int s = decltype(slot_tag)::value;
    const int gg = gj < nsub_total ? gj : gj - nsub_total;          // past the end: a harmless re-fetch keeps the wait counts uniform
    gload16_refill(wsh[s], W.hi + wbase + (uint32_t)gg * 512u);
    gload16_refill(wsl[s], W.lo + wbase + (uint32_t)gg * 512u);
  };

  issue_chunk(0, 0);
  static_for<D>([&](auto s_tag) __attribute__((always_inline)) { issue_w(s_tag, decltype(s_tag)::value); });

  // ---- everything the epilogue needs from memory is requested NOW, under the k-loop: per-column vectors, and the tile's
  // residual (16 bytes per lane and round) -- fetched where it is used, each of the 4 RT rounds paid an L2 round trip
  const int prow = lane >> 3, pc4 = (lane & 7) * 4;
  const int ncol0 = n0 + wid * 32, n4 = ncol0 + pc4;
  const bool col_ok = n4 < N;                       // N % 4 == 0
  const float4 b4 = col_ok ? ld4(ep.bias + n4) : zero4();
  float4 c4 = zero4(), g4 = zero4(), be4 = zero4();
  if constexpr (FOLD) { if (col_ok) c4 = ld4(ep.colsum + n4); }
  if constexpr (RES == 3) {
    if (col_ok) { g4 = ld4(ep.rgamma + n4); be4 = ld4(ep.rbeta + n4); }
  }
  float vbias = 0.f, vcsum = 0.f;                   // accumulator layout (lane -> column r of the wave's 32): the V^T path
  if constexpr (OUT_QKV) {
    if (ncol0 + r < N) {
      vbias = ep.bias[ncol0 + r];
      if constexpr (FOLD) vcsum = ep.colsum[ncol0 + r];
    }
  }
  float4 rres[RES == 1 ? 4 * RT : 1];
  uint2 rrh[(RES == 2 || RES == 3) ? 4 * RT : 1], rrl[(RES == 2 || RES == 3) ? 4 * RT : 1];
  if constexpr (RES != 0) {
#pragma unroll
    for (int q = 0; q < 4 * RT; ++q) {
      const int rit = 8 * q + prow, m = m0 + rit;                    // round q = 4 t + g covers tile rows 8 q .. 8 q + 7
      const bool ok = rit < rows_valid && m < M && col_ok;
      if constexpr (RES == 1) {
        const size_t o = (size_t)(EMBED ? 1 + (m - grp * group_rows) : m) * ep.ld + n4;   // EMBED: the positional row of the frame
        rres[q] = ok ? ld4(ep.res + o) : zero4();
      } else {
        const size_t o = (size_t)m * ep.ld + n4;
        rrh[q] = ok ? *reinterpret_cast<const uint2*>(ep.resh + o) : make_uint2(0u, 0u);
        rrl[q] = ok ? *reinterpret_cast<const uint2*>(ep.resl + o) : make_uint2(0u, 0u);
      }
    }
  }
  // ---- (mean, rstd) of the tile's rows from the producer's per-row partial statistics (FOLD: of the A rows, RES == 3: of
  // the residual rows; a kernel has one of the two): built BEHIND the prologue's requests: its (compiler-tracked) loads are the youngest of the queue, so what hipcc waits for
  // in front of the build is what step 0 needs anyway
  float2* const stab = reinterpret_cast<float2*>(lds + x3s_tab_base(RT, NSUB, MULTI));
  if constexpr (LN_TABS) {
    if (tid < TR) {
      const float* st = FOLD ? ep.astat : ep.rstat;
      const int m = m0 + tid;
      float2 v = make_float2(0.f, 0.f);         // pad rows: (0, 0) -> every folded value is the finite constant b' / beta
      if (tid < rows_valid && m < M) {
        const float* q = st + (size_t)m * ep.stat_parts * 2;
        const float cols = (float)ep.stat_cols;
        float s1 = 0.f;
        for (int p = 0; p < ep.stat_parts; ++p) s1 += q[2 * p];
        const float mean = s1 * ep.inv_dim;
        float m2 = 0.f;
        for (int p = 0; p < ep.stat_parts; ++p) {                  // Chan's merge of the centred partials
          const float dm = q[2 * p] / cols - mean;
          m2 += q[2 * p + 1] + cols * dm * dm;
        }
        v = make_float2(mean, 1.0f / sqrtf(m2 * ep.inv_dim + 1e-5f));
      }
      stab[tid] = v;
    }
  }


  f32x16 acc[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  // fragment read addresses: row r of sub-tile t, 16-byte chunk (ks * 2 + h) ^ sw of its 64-byte row
  const int sw = (r >> 2) & 3;
  const uint32_t fr0 = (uint32_t)(r * 64 + ((h ^ sw) * 16)), fr1 = (uint32_t)(r * 64 + (((2 + h) ^ sw) * 16));
#ifndef MDM_EMU
  const uint32_t lds_base = lds_addr_of(lds);
#endif
  p16x8 fah[2][RT], fal[2][RT];      // fragments of sub-steps j (set j & 1)
  auto read_frags = [&](auto j_tag, int buf) __attribute__((always_inline)) {
    constexpr int j = decltype(j_tag)::value, ms = j / 2, ks = j % 2;
#pragma unroll
    for (int t = 0; t < RT; ++t) {
#ifdef MDM_EMU
      lds_read16(fah[j & 1][t], lds + buf * BUF, (uint32_t)(((ms * 2 + 0) * 2 * RT) * 1024 + t * 2048) + (ks ? fr1 : fr0));
      lds_read16(fal[j & 1][t], lds + buf * BUF, (uint32_t)(((ms * 2 + 1) * 2 * RT) * 1024 + t * 2048) + (ks ? fr1 : fr0));
#else
      // (the immediate of a DS instruction is 16 bits: the part of a 64 / 72 KB chunk image beyond 32 KB goes into the address)
      constexpr uint32_t OH = (uint32_t)(((ms * 2 + 0) * 2 * RT) * 1024), OL = (uint32_t)(((ms * 2 + 1) * 2 * RT) * 1024);
      const uint32_t ad = lds_base + (uint32_t)buf * BUF + (ks ? fr1 : fr0) + (uint32_t)t * 2048u;
      lds_read16<(int)(OH & 32767u)>(fah[j & 1][t], ad + (OH & ~32767u));
      lds_read16<(int)(OL & 32767u)>(fal[j & 1][t], ad + (OL & ~32767u));
#endif
    }
  };
  auto wait_frags = [&](auto j_tag, auto younger_tag) __attribute__((always_inline)) {
    constexpr int j = decltype(j_tag)::value, n = decltype(younger_tag)::value;
    if constexpr (RT == 1) lds_wait<n>(fah[j & 1][0], fal[j & 1][0]);
    else lds_wait<n>(fah[j & 1][0], fal[j & 1][0], fah[j & 1][1], fal[j & 1][1]);
  };

  for (int c = 0; c < nchunks; ++c) {
    const int buf = MULTI ? (c & 1) : 0;
    // chunk c landed (this wave's pieces): c == 0 -- the D W sub-steps of the prologue are younger; c > 0 -- the counted W waits
    // of chunk c - 1 (sub-steps >= D, all issued behind the pieces) have already retired them
    if (c == 0) wait_vmem_upto<15>();     // (2 D = 16 younger loads; 15 is the encoding's reach here: one W load more retired)
    wg_barrier_nodrain();                 // every wave's pieces visible; every wave is past chunk c - 1, whose buffer refills now
    if constexpr (MULTI) issue_chunk(min(c + 1, nchunks - 1), buf ^ 1);   // (last chunk: a harmless re-fetch keeps the counts uniform)
    read_frags(std::integral_constant<int, 0>{}, buf);
    static_for<NSUB>([&](auto j_tag) __attribute__((always_inline)) {
      constexpr int j = decltype(j_tag)::value;
      if constexpr (j + 1 < NSUB) read_frags(std::integral_constant<int, j + 1>{}, buf);
      // W(c, j): younger = the D - 1 sub-steps behind it (+ the next chunk's pieces when it was issued in front of them)
      constexpr int NW = 2 * (D - 1) + ((MULTI && j < D) ? PW : 0);
      vmem_wait<NW>(wsh[j % D], wsl[j % D]);
      wait_frags(j_tag, std::integral_constant<int, (j + 1 < NSUB) ? 2 * RT : 0>{});
#ifndef MDM_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
      for (int t = 0; t < RT; ++t) {
        acc[t] = mfma_p16(fal[j & 1][t], wsh[j % D], acc[t]);
        acc[t] = mfma_p16(fah[j & 1][t], wsl[j % D], acc[t]);
        acc[t] = mfma_p16(fah[j & 1][t], wsh[j % D], acc[t]);
      }
#ifndef MDM_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      issue_w(std::integral_constant<int, j % D>{}, c * NSUB + j + D);
    });
  }
  // The tail's re-fetches land before the registers / LDS they target are reused -- and the wait NAMES the sixteen slot registers:
  // to hipcc an in-place refill writes its slot at the asm statement, so behind a slot's last MFMA the register is free, and a
  // bare s_waitcnt vmcnt(0) here let it hand slots with a load still in flight to the epilogue's lane indices (MI355X: rows of
  // fp16 weight bits as `lane >> 3` in a few lanes, stores into the void -- "Write access to a read-only page"; invisible to
  // the emulator, which executes variables, not registers: the hazard class of profiles/r03b_pipe_determinism.md)
  static_assert(D == 8, "the closing wait names eight slot pairs");
  vmem_wait<0>(wsh[0], wsl[0], wsh[1], wsl[1], wsh[2], wsl[2], wsh[3], wsl[3]);
  vmem_wait<0>(wsh[4], wsl[4], wsh[5], wsl[5], wsh[6], wsl[6], wsh[7], wsl[7]);

  // ---- epilogue: each wave turns its RT 32 x 32 accumulators through a private 1 KB LDS patch, 8 rows x 32 columns per round,
  // into (row = lane >> 3, 4 consecutive columns) per lane -> 16-byte fp32 / 8-byte plane accesses (gemm_x3.h).
  float* patch = reinterpret_cast<float*>(lds + x3s_patch_base(RT, NSUB, MULTI)) + wid * (X3_PATCH_BYTES / 4);
  const float accs = ep.acc_scale;
  constexpr bool COL_SCALE = x3_has_col_scale(ACT, RES);
  const float mult4 = (COL_SCALE && n4 < ep.scale_cols) ? ep.col_scale : 1.f;
  auto finish4 = [&](float4 v4, float2 st) __attribute__((always_inline)) {
    v4.x *= accs; v4.y *= accs; v4.z *= accs; v4.w *= accs;
    if constexpr (FOLD) {
      v4.x = st.y * (v4.x - st.x * c4.x) + b4.x; v4.y = st.y * (v4.y - st.x * c4.y) + b4.y;
      v4.z = st.y * (v4.z - st.x * c4.z) + b4.z; v4.w = st.y * (v4.w - st.x * c4.w) + b4.w;
    } else {
      v4.x += b4.x; v4.y += b4.y; v4.z += b4.z; v4.w += b4.w;
    }
    if (ACT == ACT_GELU) { v4.x = gelu_erf_fast(v4.x); v4.y = gelu_erf_fast(v4.y); v4.z = gelu_erf_fast(v4.z); v4.w = gelu_erf_fast(v4.w); }
    else if (ACT == ACT_SILU) { v4.x = silu(v4.x); v4.y = silu(v4.y); v4.z = silu(v4.z); v4.w = silu(v4.w); }
    if constexpr (COL_SCALE) { v4.x *= mult4; v4.y *= mult4; v4.z *= mult4; v4.w *= mult4; }
    return v4;
  };

  if constexpr (OUT_QKV) {
    // in_proj -> the attention operand planes of attention_x3.h.  group == sequence, row of the group == token; the tile's 128
    // columns are ONE head of ONE of Q / K / V (D % 128 == 0).
    const int Dm = ep.D, SPq = ep.qkv.SP, Hq = ep.qkv.H, nkt = ep.qkv.NKT;
    const int which = n0 / Dm, hcol = n0 - which * Dm, head = hcol >> 7, d0 = wid * 32;
    const size_t shq = (size_t)grp * Hq + head;
    if (n0 < N) {
      if (which == 2) {
        // V^T: accumulator registers 8 s2 .. 8 s2 + 7 of a lane ARE positions 8 h .. 8 h + 7 of 16-key group s2 of key tile kt
        const float bias = vbias, csum = vcsum;
#pragma unroll
        for (int t = 0; t < RT; ++t) {
          const int kt = tig * RT + t;
          if (kt < nkt) {
            p16_t* vhp = ep.qkv.vh + ((shq * nkt + kt) * AX_HD + d0 + r) * 32 + 8 * h;
            p16_t* vlp = ep.qkv.vl + ((shq * nkt + kt) * AX_HD + d0 + r) * 32 + 8 * h;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
              float vv[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                if constexpr (FOLD) {
                  const float2 st = stab[32 * t + mfma_row(8 * s2 + j, h)];
                  vv[j] = st.y * (acc[t][8 * s2 + j] * accs - st.x * csum) + bias;
                } else {
                  vv[j] = acc[t][8 * s2 + j] * accs + bias;
                }
              }
              p16x8 vh8, vl8;
              split8(vv, vh8, vl8);
              *reinterpret_cast<p16x8*>(vhp + 16 * s2) = vh8;
              *reinterpret_cast<p16x8*>(vlp + 16 * s2) = vl8;
            }
          }
        }
      } else {
        p16_t* dh = (which == 0 ? ep.qkv.qh : ep.qkv.kh) + shq * SPq * AX_HD + d0 + pc4;
        p16_t* dl = (which == 0 ? ep.qkv.ql : ep.qkv.kl) + shq * SPq * AX_HD + d0 + pc4;
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
#pragma unroll
            for (int e = 0; e < 4; ++e) patch[((e + 4 * h) << 5) + r] = acc[t][4 * g + e];
            wave_lds_fence();
            float4 v4 = ld4(&patch[prow * 32 + pc4]);
            wave_lds_fence();
            const int rit = t * 32 + 8 * g + prow, tok = tig * TR + rit;
            float2 st = make_float2(0.f, 1.f);
            if constexpr (FOLD) st = stab[rit];
            v4 = finish4(v4, st);
            if (rit < rows_valid && tok < ep.S) split4_store(dh + (size_t)tok * AX_HD, dl + (size_t)tok * AX_HD, v4);
          }
      }
    }
    return;
  } else {
    float2* part = reinterpret_cast<float2*>(lds + x3s_part_base(RT, NSUB, MULTI)) + wid * TR;   // OSTAT: this wave's partials
#pragma unroll
    for (int t = 0; t < RT; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e) patch[((e + 4 * h) << 5) + r] = acc[t][4 * g + e];
        wave_lds_fence();
        float4 v4 = ld4(&patch[prow * 32 + pc4]);
        wave_lds_fence();
        const int rit = t * 32 + 8 * g + prow, m = m0 + rit;
        const bool row_ok = rit < rows_valid && m < M;
        float2 st = make_float2(0.f, 1.f);
        if constexpr (LN_TABS) st = stab[rit];
        v4 = finish4(v4, FOLD ? st : make_float2(0.f, 1.f));
        if constexpr (RES == 1) {
          v4 = add4(v4, rres[4 * t + g]);
        } else if constexpr (RES == 2 || RES == 3) {
          {
            const uint2 a = rrh[4 * t + g], b = rrl[4 * t + g];
            float4 x4 = make_float4(p16_to_f32((p16_t)(a.x & 0xffffu)) + p16_to_f32((p16_t)(b.x & 0xffffu)),
                                    p16_to_f32((p16_t)(a.x >> 16)) + p16_to_f32((p16_t)(b.x >> 16)),
                                    p16_to_f32((p16_t)(a.y & 0xffffu)) + p16_to_f32((p16_t)(b.y & 0xffffu)),
                                    p16_to_f32((p16_t)(a.y >> 16)) + p16_to_f32((p16_t)(b.y >> 16)));
            if constexpr (RES == 3) {   // the residual is LayerNorm(x), rebuilt from x's planes and its row statistics
              x4.x = (x4.x - st.x) * st.y * g4.x + be4.x; x4.y = (x4.y - st.x) * st.y * g4.y + be4.y;
              x4.z = (x4.z - st.x) * st.y * g4.z + be4.z; x4.w = (x4.w - st.x) * st.y * g4.w + be4.w;
            }
            v4 = add4(v4, x4);
          }
        }
        if constexpr (OSTAT) {   // partial (sum, centred sum of squares) of this row over the wave's 32 columns
          const float s1 = sum_lanes8((v4.x + v4.y) + (v4.z + v4.w));
          const float mw = s1 * (1.0f / 32.0f);
          const float dx = v4.x - mw, dy = v4.y - mw, dz = v4.z - mw, dw = v4.w - mw;
          const float m2 = sum_lanes8((dx * dx + dy * dy) + (dz * dz + dw * dw));
          if ((lane & 7) == 0) part[rit] = make_float2(s1, m2);
        }
        if (row_ok && col_ok) {
          if constexpr (EMBED) {
            const int bb = grp, tt = m - grp * group_rows;      // group == sample, row of the group == frame
            for (int br = 0; br < ep.emb_nbranch; ++br) {
              const size_t o = ((size_t)(br * ep.emb_B + bb) * (ep.emb_T + 1) + 1 + tt) * ep.ld + n4;
              split4_store(ep.oh + o, ep.ol + o, v4);
            }
          } else {
            const size_t o = (size_t)m * ep.ld + n4;
            if constexpr (OUT_PLANES) split4_store(ep.oh + o, ep.ol + o, v4);
            if constexpr (OUT_F32) st4(ep.out + o, v4);
          }
        }
      }
    if constexpr (OSTAT) {   // rows x waves partials -> one (sum, M2) pair per row and 128-column tile (OSTAT launches: N % 128 == 0)
      wg_barrier();
      if (tid < rows_valid && m0 + tid < M) {
        const float2* pp = reinterpret_cast<const float2*>(lds + x3s_part_base(RT, NSUB, MULTI));
        float s1 = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < X3S_WAVES; ++w4) s1 += pp[w4 * TR + tid].x;
        const float mt = s1 * (1.0f / X3S_TN);
        float m2 = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < X3S_WAVES; ++w4) {
          const float2 v = pp[w4 * TR + tid];
          const float dm = v.x * (1.0f / 32.0f) - mt;
          m2 += v.y + 32.0f * dm * dm;
        }
        *reinterpret_cast<float2*>(ep.ostat + ((size_t)(m0 + tid) * tiles_n + tile_n) * 2) = make_float2(s1, m2);
      }
    }
  }
}

#ifndef MDM_X3_KERNEL_ONLY
// Up to how many token sequences a launch takes this kernel (default 32, i.e. 16 motions under guidance; MDM_X3S_MAX_SEQS=0
// disables it for same-box A/B runs): measured cross-over with gemm_x3.h's sequence-sized tiles, profiles/r04a_small_batch.md.
inline int x3s_max_seqs() {      // (read per call: the test suites switch kernels inside one process)
  const char* e = getenv("MDM_X3S_MAX_SEQS");
  return e != nullptr ? atoi(e) : 32;
}
// 32-row tiles while they leave the chip under-filled, 64-row tiles above (MDM_X3S_RT=1|2 pins it for A/B runs)
inline int x3s_rows_setting(int groups) {
  const char* e = getenv("MDM_X3S_RT");
  const int pin = e != nullptr ? atoi(e) : 0;
  if (pin == 1 || pin == 2) return pin;
  return groups <= 12 ? 1 : 2;
}

template <int RT, int NSUB, bool MULTI, int ACT, int RES, bool OUT_F32, bool OUT_PLANES, bool OUT_QKV, bool FOLD, bool OSTAT,
          bool EMBED>
inline int launch_gemm_x3s_t(const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N, int K, int group_rows,
                             hipStream_t stream) {
  auto kfn = &gemm_x3s_kernel<RT, NSUB, MULTI, ACT, RES, OUT_F32, OUT_PLANES, OUT_QKV, FOLD, OSTAT, EMBED>;
  if (K % (NSUB * 16) != 0 || (!MULTI && K != NSUB * 16)) return -2;   // (MULTI with one chunk works: the spare buffer is re-fetched)
  if (M % group_rows != 0) return -2;
  if (!x3_has_col_scale(ACT, RES) && ep.scale_cols > 0) return -2;
  constexpr int LDS = x3s_lds_bytes(RT, NSUB, MULTI);
#ifndef MDM_EMU
  if (LDS > 65536) {
    static bool configured[kMaxDevices] = {};
    bool& done = configured[rt_device_ordinal()];
    if (!done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
      done = true;
    }
  }
#endif
  const int TR = 32 * RT;
  const int tpg = (group_rows + TR - 1) / TR, tiles_m = (M / group_rows) * tpg, tiles_n = (N + X3S_TN - 1) / X3S_TN;
  const int total = tiles_m * tiles_n;
  MDM_LAUNCH(kfn, dim3(total), dim3(64 * X3S_WAVES), LDS, stream, A, W, ep, M, N, K, group_rows, tpg, tiles_n, total);
  return 0;
}

// the GEMM kinds of launch_gemm_x3_ln (gemm_x3.h), plus kind 6 = layer 0's in_proj (no folded LayerNorm)
template <int RT>
inline int launch_gemm_x3s_rt(int kind, const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N, int K,
                              int group_rows, hipStream_t s) {
  switch (kind) {
    case 0: return launch_gemm_x3s_t<RT, 16, true, ACT_NONE, 0, false, false, true, true, false, false>(A, W, ep, M, N, K, group_rows, s);
    case 6: return launch_gemm_x3s_t<RT, 16, true, ACT_NONE, 0, false, false, true, false, false, false>(A, W, ep, M, N, K, group_rows, s);
    case 1: return launch_gemm_x3s_t<RT, 16, true, ACT_NONE, 2, false, true, false, false, true, false>(A, W, ep, M, N, K, group_rows, s);
    case 2: return launch_gemm_x3s_t<RT, 16, true, ACT_NONE, 3, false, true, false, false, true, false>(A, W, ep, M, N, K, group_rows, s);
    case 3: return launch_gemm_x3s_t<RT, 16, true, ACT_GELU, 0, false, true, false, true, false, false>(A, W, ep, M, N, K, group_rows, s);
    case 4: return launch_gemm_x3s_t<RT, 16, true, ACT_NONE, 0, true, false, false, true, false, false>(A, W, ep, M, N, K, group_rows, s);
    case 5:   // InputProcess: K = 288 (263 features padded to 9 x 32) is one chunk of 18 sub-steps
      return launch_gemm_x3s_t<RT, 18, false, ACT_NONE, 1, false, true, false, false, false, true>(A, W, ep, M, N, K, group_rows, s);
    default: return -2;
  }
}
inline int launch_gemm_x3s(int kind, const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N, int K,
                           int group_rows, hipStream_t s) {
  if (x3s_rows_setting(M / group_rows) == 1) return launch_gemm_x3s_rt<1>(kind, A, W, ep, M, N, K, group_rows, s);
  return launch_gemm_x3s_rt<2>(kind, A, W, ep, M, N, K, group_rows, s);
}
#endif  // MDM_X3_KERNEL_ONLY

}  // namespace mdm

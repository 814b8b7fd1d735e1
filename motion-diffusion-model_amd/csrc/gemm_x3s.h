// gemm_x3s.h -- the split-precision GEMM of gemm_x3.h for SMALL row counts: the latency regime of the encoder.
//
// Why a second kernel.  gemm_x3.h tiles the row space by whole token sequences (208 x 256 tiles, one persistent 8-wave
// workgroup per CU): at the batch sizes the reference's own callers use -- sample/generate.py:76,98 defaults to
// `--num_samples 6`, i.e. 12 sequences under classifier-free guidance; README.md:13 quotes per-call latency -- a launch is one
// tile deep on 5-30 % of the CUs and its time is that tile's serial k-loop: 33 us per GEMM launch, 73-77 ms per 50-step loop at
// B = 1 ... 10 (profiles/r04a_small_batch.md).  This kernel cuts the same contraction into 32- or 64-row x 128- or 256-column
// tiles, one NON-persistent 4-wave workgroup each (a wave owns 32 or 64 columns: NCB column blocks sharing every A fragment it
// reads), so that M = 394 rows (B = 1) already give 168 workgroups for in_proj, and keeps everything a tile needs in flight:
//   * A (activation planes [rows][K], hi | lo): a K-CHUNK of 16 (18) sub-steps x 16 k of the tile's rows lives in LDS
//     (32 KB per 32 rows), fetched by global_load_lds_dwordx4 in the 64-byte-row XOR-swizzled image of gemm_x3.h; chunk
//     c + 1 is requested into the other buffer as soon as the barrier of chunk c has passed: ONE rendezvous per 256 k;
//   * W (fragment-ordered hi | lo planes, gemm_x3.h header): straight to registers, a ring of sub-step slots
//     (16 KB per wave in flight) refilled in place behind the MFMAs that consumed them (common.h gload16_refill), retired by
//     counted vmcnt waits -- the vector-memory queue retires in order across LDS-DMA pieces and register loads
//     (tools/vmcnt_order);
//   * fragment reads one sub-step ahead through untracked ds_reads with counted lgkmcnt waits (common.h lds_read16).
// Same operands, same weight planes, same epilogue algebra (X3Epilogue: folded LayerNorm, Q / K / V^T operand planes,
// GELU, plane residuals, row statistics, InputProcess / OutputProcess forms) as gemm_x3.h, so a forward may run on either
// kernel; the row statistics a producer leaves are per tile width (128 or 256 columns: X3Epilogue::stat_cols).
// Replaces the same reference calls as gemm_x3.h (model/mdm.py:77-84 `addmm`s under nn.TransformerEncoderLayer,
// :343-349 InputProcess, :372-386 OutputProcess) for nseq <= x3s_max_seqs().
#pragma once
#include "gemm_x3.h"

namespace mdm {

// MDM_X3S_EPI_AHEAD (default 1; -DMDM_X3S_EPI_AHEAD=0: one round trip per round): the epilogue reads round j+1's patch before it finishes
// round j (a wave's LDS operations execute in order: read j+1, then write j+2 behind it, one patch).  Same-box A/B, 8 of 8 pairs
// positive: DiP B = 32 +0.45 %, the 50-step loop at B = 1 / 6 / 10 -0.45 / -0.3 / -0.3 % (profiles/r05j_epilogue_ahead.md); the same
// change in gemm_x3.h (MDM_X3_EPI_AHEAD) measured neutral on the headline and stays off
#ifndef MDM_X3S_EPI_AHEAD
#define MDM_X3S_EPI_AHEAD 1
#endif
constexpr bool X3S_EPI_AHEAD = MDM_X3S_EPI_AHEAD != 0;

constexpr int X3S_WAVES = 4;
constexpr int x3s_tn(int ncb) { return 128 * ncb; }   // columns per tile: 4 waves x NCB blocks of 32
// W sub-steps in flight per wave (hi + lo fragment per column block, 8 VGPRs each): 16 KB per wave.  Twice the depth for the
// 32-row / 128-column tiles (a whole 256-k chunk ahead, 64 more VGPRs) measured SLOWER on the same box: 24.1 vs 23.3 ms per
// 50-step loop at B = 1, 40.4 vs 37.6 at B = 6 (profiles/r04a_small_batch.md) -- the tiles are not waiting for the W stream's depth
constexpr int x3s_wdepth(int ncb) { return 8 / ncb; }
constexpr int x3s_buf_bytes(int rt, int nsub) { return nsub * 32 * rt * 64; }        // one K-chunk of A: hi | lo, 64-byte rows per 32 k
constexpr int x3s_patch_base(int rt, int nsub, bool multi) { return (multi ? 2 : 1) * x3s_buf_bytes(rt, nsub); }
constexpr int x3s_tab_base(int rt, int nsub, bool multi) { return x3s_patch_base(rt, nsub, multi) + X3S_WAVES * X3_PATCH_BYTES; }
constexpr int x3s_part_base(int rt, int nsub, bool multi) { return x3s_tab_base(rt, nsub, multi) + 32 * rt * 8; }
constexpr int x3s_lds_bytes(int rt, int ncb, int nsub, bool multi) { return x3s_part_base(rt, nsub, multi) + X3S_WAVES * ncb * 32 * rt * 8; }

// RT: 32-row sub-tiles per tile (1 or 2).  NCB: 32-column blocks per wave (1 or 2).  NSUB: 16-deep k sub-steps per chunk (even);
// K = NSUB * 16 * nchunks.  MULTI: double-buffered A (more than one chunk).  The other flags are gemm_x3_kernel's.
// Rows are GROUPED (group_rows = tokens of a sequence; InputProcess: frames of a sample): a tile never straddles two groups, so
// that the in_proj epilogue's (sequence, token) and the EMBED epilogue's (sample, frame) are tile-uniform / row-affine.
#if defined(MDM_PROBES) && !defined(MDM_EMU)
// PROBE BUILD ONLY: wave 0's shader-clock stamps of ONE selected launch (mdm_debug_set(9, n): the n-th gemm_x3s launch after the
// call; mdm_debug_get(100 + 4 * workgroup + i)): i = 0 kernel entry, 1 chunk 0 visible to the workgroup (first barrier passed),
// 2 k-loop and its trailing re-fetches retired, 3 last store issued.  tools/x3s_timeline.py
constexpr int X3S_TL_WGS = 4096;
__device__ unsigned long long g_x3s_tl[4 * X3S_TL_WGS];
__device__ int g_x3s_tl_on;
#define X3S_STAMP(i)                                                                                     \
  do {                                                                                                   \
    if (tl_on && tid == 0 && blockIdx.x < X3S_TL_WGS) g_x3s_tl[4 * blockIdx.x + (i)] = __builtin_readcyclecounter(); \
  } while (0)
#else
#define X3S_STAMP(i) do { } while (0)
#endif

template <int RT, int NCB, int NSUB, bool MULTI, int ACT, int RES, bool OUT_F32, bool OUT_PLANES, bool OUT_QKV, bool FOLD, bool OSTAT,
          bool EMBED>
__global__ __launch_bounds__(64 * X3S_WAVES, 2) void gemm_x3s_kernel(X3Operand A, X3Weights W, X3Epilogue ep, int M, int N, int K,
                                                                    int group_rows, int tiles_per_group, int tiles_n,
                                                                    int total) {
  MDM_DYN_SMEM(unsigned char, lds);
  static_assert(NSUB % 2 == 0 && (RT == 1 || RT == 2) && (NCB == 1 || NCB == 2), "tile shape");
  constexpr int TR = 32 * RT, TN = x3s_tn(NCB), D = x3s_wdepth(NCB), NBLK = X3S_WAVES * NCB;
  constexpr int BUF = x3s_buf_bytes(RT, NSUB);
  constexpr int PW = NSUB * RT / 2;                      // LDS-DMA pieces (1 KB) per wave and chunk
  constexpr int LW = 2 * NCB;                            // W loads per wave and sub-step
  static_assert(2 * NSUB * RT % X3S_WAVES == 0, "pieces must divide among the waves");
  static_assert(LW * (D - 1) + PW <= 63 && (NSUB > D ? LW * D : LW * NSUB) <= 63 && NSUB >= D && (!MULTI || NSUB % D == 0) && (D * NCB) % 4 == 0,
                "vmcnt range / slot <-> sub-step map across chunks / closing wait");
  constexpr bool LN_TABS = FOLD || RES == 3;

  const int tid = threadIdx.x;
#if defined(MDM_PROBES) && !defined(MDM_EMU)
  const bool tl_on = g_x3s_tl_on != 0;
#endif
  X3S_STAMP(0);
  MDM_KERNARGS_NOW("s"(A.hi), "s"(A.lo), "s"(W.hi), "s"(W.lo), "s"(M), "s"(N), "s"(K), "s"(group_rows), "s"(tiles_per_group), "s"(tiles_n), "s"(total));
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
  const int n0 = tile_n * TN;
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
  // ---- W stream: this wave's fragments of 16-deep sub-step `gj` (global index over the whole K): hi and lo of each of its
  // NCB column blocks, 1 KB each.  (A block whose 32 columns lie past the padded weight rows -- OutputProcess: N = 264 -> 288
  // packed rows -- re-reads the last block: its results are never stored.)
  uint32_t wbase[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
    wbase[cb] = (uint32_t)min((n0 >> 5) + wid * NCB + cb, (N + 31) / 32 - 1) * (uint32_t)(K / 16) * 512u + (uint32_t)lane * 8u;
  const int nsub_total = K / 16;
  p16x8 wsh[D * NCB] = {}, wsl[D * NCB] = {};     // slot d, column block cb: [d * NCB + cb]  (zero: the first refill formally reads its slot)
  auto issue_w = [&](auto slot_tag, int gj) __attribute__((always_inline)) {
    constexpr int sl = decltype(slot_tag)::value;
    const int gg = gj < nsub_total ? gj : gj - nsub_total;          // past the end: a harmless re-fetch keeps the wait counts uniform
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      gload16_refill(wsh[sl * NCB + cb], W.hi + wbase[cb] + (uint32_t)gg * 512u);
      gload16_refill(wsl[sl * NCB + cb], W.lo + wbase[cb] + (uint32_t)gg * 512u);
    }
  };

  issue_chunk(0, 0);
  static_for<D>([&](auto s_tag) __attribute__((always_inline)) { issue_w(s_tag, decltype(s_tag)::value); });

  // ---- everything the epilogue needs from memory is requested NOW, under the k-loop: per-column vectors, and the tile's
  // residual (8 / 16 bytes per lane, round and column block) -- fetched where it is used, each round paid an L2 round trip
  const int prow = lane >> 3, pc4 = (lane & 7) * 4;
  int n4[NCB];
  bool col_ok[NCB];
  float4 b4[NCB], c4[NCB], g4[NCB], be4[NCB];
  float vbias[NCB], vcsum[NCB];                     // accumulator layout (lane -> column r of a 32-column block): the V^T path
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int nb = n0 + (wid * NCB + cb) * 32;
    n4[cb] = nb + pc4;
    col_ok[cb] = n4[cb] < N;                        // N % 4 == 0
    b4[cb] = col_ok[cb] ? ld4(ep.bias + n4[cb]) : zero4();
    c4[cb] = g4[cb] = be4[cb] = zero4();
    if constexpr (FOLD) { if (col_ok[cb]) c4[cb] = ld4(ep.colsum + n4[cb]); }
    if constexpr (RES == 3) {
      if (col_ok[cb]) { g4[cb] = ld4(ep.rgamma + n4[cb]); be4[cb] = ld4(ep.rbeta + n4[cb]); }
    }
    vbias[cb] = vcsum[cb] = 0.f;
    if constexpr (OUT_QKV) {
      if (nb + r < N) {
        vbias[cb] = ep.bias[nb + r];
        if constexpr (FOLD) vcsum[cb] = ep.colsum[nb + r];
      }
    }
  }
  constexpr int NRND = 4 * RT * NCB;                // epilogue rounds: round (cb, t, g) -> index (cb * RT + t) * 4 + g
  float4 rres[RES == 1 ? NRND : 1];
  uint2 rrh[(RES == 2 || RES == 3) ? NRND : 1], rrl[(RES == 2 || RES == 3) ? NRND : 1];
  // (Round 4 also built the plane residual as an LDS-DMA of the tile into the buffer that is spare during the last chunk -- same
  // piece count as the re-fetch it replaces, 32 VGPRs fewer: kernel entry -> first chunk visible 8.3 k -> 5.6 k cycles as
  // predicted, but the k-loop and the epilogue took the time back (10.4 k -> 11.2 k, 4.4 k -> 5.1 k): +-0.5 % on every bench,
  // tools/x3s_res_lds.patch, profiles/r04j_x3s_timeline.md; not kept.)
  if constexpr (RES != 0) {
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int q = 0; q < 4 * RT; ++q) {
        const int rit = 8 * q + prow, m = m0 + rit;                    // q = 4 t + g covers tile rows 8 q .. 8 q + 7
        const bool ok = rit < rows_valid && m < M && col_ok[cb];
        if constexpr (RES == 1) {
          const size_t o = (size_t)(EMBED ? 1 + (m - grp * group_rows) : m) * ep.ld + n4[cb];   // EMBED: the positional row of the frame
          rres[cb * 4 * RT + q] = ok ? ld4(ep.res + o) : zero4();
        } else {
          const size_t o = (size_t)m * ep.ld + n4[cb];
          rrh[cb * 4 * RT + q] = ok ? *reinterpret_cast<const uint2*>(ep.resh + o) : make_uint2(0u, 0u);
          rrl[cb * 4 * RT + q] = ok ? *reinterpret_cast<const uint2*>(ep.resl + o) : make_uint2(0u, 0u);
        }
      }
  }
  // ---- (mean, rstd) of the tile's rows from the producer's per-row partial statistics (FOLD: of the A rows, RES == 3: of
  // the residual rows; a kernel has one of the two): built BEHIND the prologue's requests -- its (compiler-tracked) loads are the
  // youngest of the queue, so what hipcc waits for in front of the build is what step 0 needs anyway
  float2* const stab = reinterpret_cast<float2*>(lds + x3s_tab_base(RT, NSUB, MULTI));
  if constexpr (LN_TABS) {
    if (tid < TR) {
      const float* st = FOLD ? ep.astat : ep.rstat;
      const int m = m0 + tid;
      float2 v = make_float2(0.f, 0.f);         // pad rows: (0, 0) -> every folded value is the finite constant b' / beta
      if (tid < rows_valid && m < M) {
        // the row's partials (sum, M2) -- D / stat_cols of them, at most 8 (latent_dim 1024 on 128-column tiles) -- in up to FOUR
        // 16-byte loads issued together: a loop over them was a chain of dependent round trips (load, wait, add) -- 3-4 us of a
        // tile's 9 at B = 1 (profiles/r04a_small_batch.md).  (Round 4 handled 1, 2, 4 and "else = 3" partials only: latent_dim 768
        // / 1024 have 6 / 8 on this kernel -- ADVICE r04.)  An odd count leaves odd rows 8-byte aligned: float2 loads there.
        const int np = ep.stat_parts;
        const float* q = st + (size_t)m * np * 2;
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
        const float cols = (float)ep.stat_cols, icols = 1.0f / cols;
        const float mean = ((pp[0].x + pp[0].z) + (pp[1].x + pp[1].z) + ((pp[2].x + pp[2].z) + (pp[3].x + pp[3].z))) * ep.inv_dim;
        // Chan's merge of the centred partials (no E[x^2] - mean^2 cancellation); absent partials contribute nothing
        float m2 = 0.f, dd = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float d0 = 2 * i < np ? pp[i].x * icols - mean : 0.f, d1 = 2 * i + 1 < np ? pp[i].z * icols - mean : 0.f;
          m2 += pp[i].y + pp[i].w;
          dd += d0 * d0 + d1 * d1;
        }
        m2 += cols * dd;
        v = make_float2(mean, 1.0f / sqrtf(m2 * ep.inv_dim + 1e-5f));
      }
      stab[tid] = v;
    }
  }

  f32x16 acc[NCB * RT];                               // column block cb, row sub-tile t: [cb * RT + t]
#pragma unroll
  for (int t = 0; t < NCB * RT; ++t)
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
    // chunk c landed (this wave's pieces): c == 0 -- the D W sub-steps of the prologue are younger; c > 0, NSUB > D -- the counted
    // W waits of chunk c - 1 (sub-steps >= D, all issued behind the pieces) have already retired them
    if constexpr (NSUB > D) {
      if (c == 0) vmem_wait<LW * D>(wsh[0], wsl[0]);   // (the prologue's W loads are younger than chunk 0's pieces)
    } else {
      // short chunks (NSUB == D): no W wait of chunk c - 1 lies behind the pieces of chunk c -- wait for them here: younger
      // = the LW * NSUB W loads issued since
      vmem_wait<LW * NSUB>(wsh[0], wsl[0]);
    }
    wg_barrier_nodrain();                 // every wave's pieces visible; every wave is past chunk c - 1, whose buffer refills now
    if (c == 0) X3S_STAMP(1);
    if constexpr (MULTI) issue_chunk(min(c + 1, nchunks - 1), buf ^ 1);   // (last chunk: a harmless re-fetch keeps the counts uniform)
    read_frags(std::integral_constant<int, 0>{}, buf);
    static_for<NSUB>([&](auto j_tag) __attribute__((always_inline)) {
      constexpr int j = decltype(j_tag)::value, sl = j % D;
      if constexpr (j + 1 < NSUB) read_frags(std::integral_constant<int, j + 1>{}, buf);
      // W(c, j): younger = the D - 1 sub-steps behind it (+ the next chunk's pieces when it was issued in front of them)
      constexpr int NW = LW * (D - 1) + ((MULTI && j < D) ? PW : 0);
      if constexpr (NCB == 1) vmem_wait<NW>(wsh[sl], wsl[sl]);
      else vmem_wait<NW>(wsh[sl * 2], wsl[sl * 2], wsh[sl * 2 + 1], wsl[sl * 2 + 1]);
      wait_frags(j_tag, std::integral_constant<int, (j + 1 < NSUB) ? 2 * RT : 0>{});
#ifndef MDM_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      // (accumulators interleaved: consecutive MFMAs on different accumulators; every A fragment feeds NCB column blocks)
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[cb * RT + t] = mfma_p16(fal[j & 1][t], wsh[sl * NCB + cb], acc[cb * RT + t]);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[cb * RT + t] = mfma_p16(fah[j & 1][t], wsl[sl * NCB + cb], acc[cb * RT + t]);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int t = 0; t < RT; ++t) acc[cb * RT + t] = mfma_p16(fah[j & 1][t], wsh[sl * NCB + cb], acc[cb * RT + t]);
#ifndef MDM_EMU
      __builtin_amdgcn_sched_barrier(0);
#endif
      issue_w(std::integral_constant<int, sl>{}, c * NSUB + j + D);
    });
  }
  // The tail's re-fetches land before the registers / LDS they target are reused -- and the wait NAMES every slot register:
  // to hipcc an in-place refill writes its slot at the asm statement, so behind a slot's last MFMA the register is free, and a
  // bare s_waitcnt vmcnt(0) here let it hand slots with a load still in flight to the epilogue's lane indices (MI355X: rows of
  // fp16 weight bits as `lane >> 3` in a few lanes, stores into the void -- "Write access to a read-only page"; invisible to
  // the emulator, which executes variables, not registers: the hazard class of profiles/r03b_pipe_determinism.md)
  static_for<D * NCB / 4>([&](auto q_tag) __attribute__((always_inline)) {
    constexpr int q = 4 * decltype(q_tag)::value;
    vmem_wait<0>(wsh[q], wsl[q], wsh[q + 1], wsl[q + 1], wsh[q + 2], wsl[q + 2], wsh[q + 3], wsl[q + 3]);
  });

  X3S_STAMP(2);
  // ---- epilogue: each wave turns its NCB x RT 32 x 32 accumulators through a private 1 KB LDS patch, 8 rows x 32 columns per
  // round, into (row = lane >> 3, 4 consecutive columns) per lane -> 16-byte fp32 / 8-byte plane accesses (gemm_x3.h).
  float* patch = reinterpret_cast<float*>(lds + x3s_patch_base(RT, NSUB, MULTI)) + wid * (X3_PATCH_BYTES / 4);
  const float accs = ep.acc_scale;
  constexpr bool COL_SCALE = x3_has_col_scale(ACT, RES);
  auto finish4 = [&](float4 v4, float2 st, int cb) __attribute__((always_inline)) {
    const float mult4 = (COL_SCALE && n4[cb] < ep.scale_cols) ? ep.col_scale : 1.f;
    const float4 bb = b4[cb], cc = c4[cb];
    v4.x *= accs; v4.y *= accs; v4.z *= accs; v4.w *= accs;
    if constexpr (FOLD) {
      v4.x = st.y * (v4.x - st.x * cc.x) + bb.x; v4.y = st.y * (v4.y - st.x * cc.y) + bb.y;
      v4.z = st.y * (v4.z - st.x * cc.z) + bb.z; v4.w = st.y * (v4.w - st.x * cc.w) + bb.w;
    } else {
      v4.x += bb.x; v4.y += bb.y; v4.z += bb.z; v4.w += bb.w;
    }
    if (ACT == ACT_GELU) { v4.x = gelu_erf_fast(v4.x); v4.y = gelu_erf_fast(v4.y); v4.z = gelu_erf_fast(v4.z); v4.w = gelu_erf_fast(v4.w); }
    else if (ACT == ACT_SILU) { v4.x = silu(v4.x); v4.y = silu(v4.y); v4.z = silu(v4.z); v4.w = silu(v4.w); }
    if constexpr (COL_SCALE) { v4.x *= mult4; v4.y *= mult4; v4.z *= mult4; v4.w *= mult4; }
    return v4;
  };
  // round j of column block cb: accumulator registers 4g .. 4g+3 of both lane halves -> the wave's patch (no-op past the last round)
  auto patch_write = [&](auto cb_tag, auto j_tag) __attribute__((always_inline)) {
    constexpr int cb = decltype(cb_tag)::value, j = decltype(j_tag)::value, t = j / 4, g = j % 4;
    if constexpr (j < 4 * RT) {
#pragma unroll
      for (int e = 0; e < 4; ++e) patch[((e + 4 * h) << 5) + r] = acc[cb * RT + t][4 * g + e];
    }
  };
  float2* const part_all = reinterpret_cast<float2*>(lds + x3s_part_base(RT, NSUB, MULTI));   // OSTAT: [block][row] partials

  static_for<NCB>([&](auto cb_tag) __attribute__((always_inline)) {
    constexpr int cb = decltype(cb_tag)::value;
    const int nb = n0 + (wid * NCB + cb) * 32;      // first column of this (wave, block)
    if constexpr (OUT_QKV) {
      // in_proj -> the attention operand planes of attention_x3.h.  group == sequence, row of the group == token; a 32-column
      // block lies inside ONE head of ONE of Q / K / V (D % 128 == 0).
      const int Dm = ep.D, SPq = ep.qkv.SP, Hq = ep.qkv.H, nkt = ep.qkv.NKT;
      const int which = nb / Dm, hcol = nb - which * Dm, head = hcol >> 7, d0 = hcol & 127;
      const size_t shq = (size_t)grp * Hq + head;
      if (nb < N) {
        if (which == 2) {
          // V^T: accumulator registers 8 s2 .. 8 s2 + 7 of a lane ARE positions 8 h .. 8 h + 7 of 16-key group s2 of key tile kt
          const float bias = vbias[cb], csum = vcsum[cb];
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
                    vv[j] = st.y * (acc[cb * RT + t][8 * s2 + j] * accs - st.x * csum) + bias;
                  } else {
                    vv[j] = acc[cb * RT + t][8 * s2 + j] * accs + bias;
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
          float4 v_cur = zero4();
          if constexpr (X3S_EPI_AHEAD) {
            patch_write(std::integral_constant<int, cb>{}, std::integral_constant<int, 0>{});
            wave_lds_fence();
            v_cur = ld4(&patch[prow * 32 + pc4]);
            wave_lds_fence();
            patch_write(std::integral_constant<int, cb>{}, std::integral_constant<int, 1>{});
          }
          static_for<4 * RT>([&](auto j_tag) __attribute__((always_inline)) {
              constexpr int j = decltype(j_tag)::value, t = j / 4, g = j % 4;
              float4 v4;
              if constexpr (X3S_EPI_AHEAD) {
                wave_lds_fence();
                float4 v_next = zero4();
                if constexpr (j + 1 < 4 * RT) v_next = ld4(&patch[prow * 32 + pc4]);
                wave_lds_fence();
                patch_write(std::integral_constant<int, cb>{}, std::integral_constant<int, j + 2>{});
                v4 = v_cur;
                v_cur = v_next;
              } else {
                patch_write(std::integral_constant<int, cb>{}, j_tag);
                wave_lds_fence();
                v4 = ld4(&patch[prow * 32 + pc4]);
                wave_lds_fence();
              }
              const int rit = t * 32 + 8 * g + prow, tok = tig * TR + rit;
              float2 st = make_float2(0.f, 1.f);
              if constexpr (FOLD) st = stab[rit];
              v4 = finish4(v4, st, cb);
              if (rit < rows_valid && tok < ep.S) split4_store(dh + (size_t)tok * AX_HD, dl + (size_t)tok * AX_HD, v4);
          });
        }
      }
    } else {
      float2* part = part_all + (wid * NCB + cb) * TR;
      float4 v_cur = zero4();
      if constexpr (X3S_EPI_AHEAD) {
        patch_write(std::integral_constant<int, cb>{}, std::integral_constant<int, 0>{});
        wave_lds_fence();
        v_cur = ld4(&patch[prow * 32 + pc4]);
        wave_lds_fence();
        patch_write(std::integral_constant<int, cb>{}, std::integral_constant<int, 1>{});
      }
      static_for<4 * RT>([&](auto j_tag) __attribute__((always_inline)) {
          constexpr int j = decltype(j_tag)::value, t = j / 4, g = j % 4;
          float4 v4;
          if constexpr (X3S_EPI_AHEAD) {
            wave_lds_fence();
            float4 v_next = zero4();
            if constexpr (j + 1 < 4 * RT) v_next = ld4(&patch[prow * 32 + pc4]);
            wave_lds_fence();
            patch_write(std::integral_constant<int, cb>{}, std::integral_constant<int, j + 2>{});
            v4 = v_cur;
            v_cur = v_next;
          } else {
            patch_write(std::integral_constant<int, cb>{}, j_tag);
            wave_lds_fence();
            v4 = ld4(&patch[prow * 32 + pc4]);
            wave_lds_fence();
          }
          const int rit = t * 32 + 8 * g + prow, m = m0 + rit;
          const bool row_ok = rit < rows_valid && m < M;
          float2 st = make_float2(0.f, 1.f);
          if constexpr (LN_TABS) st = stab[rit];
          v4 = finish4(v4, FOLD ? st : make_float2(0.f, 1.f), cb);
          if constexpr (RES == 1) {
            v4 = add4(v4, rres[(cb * RT + t) * 4 + g]);
          } else if constexpr (RES == 2 || RES == 3) {
            const uint2 a = rrh[(cb * RT + t) * 4 + g], b = rrl[(cb * RT + t) * 4 + g];
            float4 x4 = make_float4(p16_to_f32((p16_t)(a.x & 0xffffu)) + p16_to_f32((p16_t)(b.x & 0xffffu)),
                                    p16_to_f32((p16_t)(a.x >> 16)) + p16_to_f32((p16_t)(b.x >> 16)),
                                    p16_to_f32((p16_t)(a.y & 0xffffu)) + p16_to_f32((p16_t)(b.y & 0xffffu)),
                                    p16_to_f32((p16_t)(a.y >> 16)) + p16_to_f32((p16_t)(b.y >> 16)));
            if constexpr (RES == 3) {   // the residual is LayerNorm(x), rebuilt from x's planes and its row statistics
              const float4 gg = g4[cb], be = be4[cb];
              x4.x = (x4.x - st.x) * st.y * gg.x + be.x; x4.y = (x4.y - st.x) * st.y * gg.y + be.y;
              x4.z = (x4.z - st.x) * st.y * gg.z + be.z; x4.w = (x4.w - st.x) * st.y * gg.w + be.w;
            }
            v4 = add4(v4, x4);
          }
          if constexpr (OSTAT) {   // partial (sum, centred sum of squares) of this row over the block's 32 columns
            const float s1 = sum_lanes8((v4.x + v4.y) + (v4.z + v4.w));
            const float mw = s1 * (1.0f / 32.0f);
            const float dx = v4.x - mw, dy = v4.y - mw, dz = v4.z - mw, dw = v4.w - mw;
            const float m2 = sum_lanes8((dx * dx + dy * dy) + (dz * dz + dw * dw));
            if ((lane & 7) == 0) part[rit] = make_float2(s1, m2);
          }
          if (row_ok && col_ok[cb]) {
            if constexpr (EMBED) {
              const int bb = grp, tt = m - grp * group_rows;      // group == sample, row of the group == frame
              for (int br = 0; br < ep.emb_nbranch; ++br) {
                const size_t o = ((size_t)(br * ep.emb_B + bb) * (ep.emb_T + 1) + 1 + tt) * ep.ld + n4[cb];
                split4_store(ep.oh + o, ep.ol + o, v4);
              }
            } else {
              const size_t o = (size_t)m * ep.ld + n4[cb];
              if constexpr (OUT_PLANES) split4_store(ep.oh + o, ep.ol + o, v4);
              if constexpr (OUT_F32) st4(ep.out + o, v4);
            }
          }
      });
    }
  });
  if constexpr (OSTAT && !OUT_QKV) {   // rows x blocks partials -> one (sum, M2) pair per row and tile (OSTAT launches: N % TN == 0)
    wg_barrier();
    if (tid < rows_valid && m0 + tid < M) {
      float s1 = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < NBLK; ++w4) s1 += part_all[w4 * TR + tid].x;
      const float mt = s1 * (1.0f / TN);
      float m2 = 0.f;
#pragma unroll
      for (int w4 = 0; w4 < NBLK; ++w4) {
        const float2 v = part_all[w4 * TR + tid];
        const float dm = v.x * (1.0f / 32.0f) - mt;
        m2 += v.y + 32.0f * dm * dm;
      }
      *reinterpret_cast<float2*>(ep.ostat + ((size_t)(m0 + tid) * tiles_n + tile_n) * 2) = make_float2(s1, m2);
    }
  }
  X3S_STAMP(3);
}

#ifndef MDM_X3_KERNEL_ONLY
// Which forwards take this kernel, and on which tile height: per-model options (include/mdm_hip.h mdm_set_option, ABI 9;
// rounds 3-4 read environment variables here, on every launch).
//   max_seqs   up to how many token sequences a forward runs on these tiles (default 80, i.e. 40 motions under guidance; 0
//              disables the kernel).  Round 4 put the cross-over with gemm_x3.h's sequence-sized tiles at 40 sequences
//              (profiles/r04a_small_batch.md); re-measured on the final kernels of round 5 (profiles/r05k_crossovers.md, ms per
//              50-step loop, row tiles / sequence tiles): 48 sequences 88.2 / 101.1, 64: 114.2 / 112.9, 80: 135.7 / 137.2, 96:
//              159.7 / 154.5, 128: 212.1 / 178.3 -- and on a box whose power-limited clocks ran the big kernel 20 % slower the row
//              tiles won everything up to 80 sequences by 16-28 %;
//   row_tiles  0 = by size, 1 / 2 = pin 32- / 64-row tiles.
// ONE shape for all GEMMs of a forward, because the row statistics a producer leaves (per tile width) are what its consumer
// merges.  By size: 32-row tiles up to 12 sequences (B <= 6 under guidance: 22.0 vs 26.5 ms per 50-step loop at B = 1, 30.7 vs
// 32.4 at B = 5, 36.2 vs 37.4 at B = 6), 64-row tiles above (39.2 vs 42.1 at B = 8, 61.9 vs 72.5 at B = 16; r4lat8); always
// 128 columns: the 256-column form (NCB = 2: every A fragment feeds two column blocks, half the LDS reads and half the
// activation traffic per MFMA) measured no faster anywhere -- 28.7 / 37.0 / 55.7 / 68.1 / 123.7 ms at B = 1 / 6 / 10 / 16 / 32
// for 32 x 256 against 23.0 / 38.7 / 53.7 / 74.1 / 136.1 for 32 x 128 and 27.3 / 38.4 / 53.2 / 66.5 / 121.6 for 64 x 128 --
// and is compiled into the probe library only (X3sOptions::ncb).
#if defined(MDM_PROBES) && !defined(MDM_EMU)
inline int& x3s_tl_target() { static int v = -1; return v; }   // mdm_debug_set(9, n); < 0: off
inline int& x3s_tl_count() { static int v = 0; return v; }
#endif
struct X3sOptions { int max_seqs = 80; int row_tiles = 0; int ncb = 1; };
struct X3sShape { int rt, ncb; };
inline X3sShape x3s_shape(const X3sOptions& o, int nseq) {
  X3sShape sh{nseq <= 12 ? 1 : 2, 1};
  if (o.row_tiles == 1 || o.row_tiles == 2) sh.rt = o.row_tiles;
#ifdef MDM_PROBES
  if (o.ncb == 2) sh.ncb = 2;
#endif
  return sh;
}

template <int RT, int NCB, int NSUB, bool MULTI, int ACT, int RES, bool OUT_F32, bool OUT_PLANES, bool OUT_QKV, bool FOLD, bool OSTAT,
          bool EMBED>
inline int launch_gemm_x3s_t(const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N, int K, int group_rows,
                             hipStream_t stream) {
  auto kfn = &gemm_x3s_kernel<RT, NCB, NSUB, MULTI, ACT, RES, OUT_F32, OUT_PLANES, OUT_QKV, FOLD, OSTAT, EMBED>;
  if (K % (NSUB * 16) != 0 || (!MULTI && K != NSUB * 16)) return -2;   // (MULTI with one chunk works: the spare buffer is re-fetched)
  if (M % group_rows != 0) return -2;
  if (!x3_has_col_scale(ACT, RES) && ep.scale_cols > 0) return -2;
  if (OSTAT && N % x3s_tn(NCB) != 0) return -2;
  constexpr int LDS = x3s_lds_bytes(RT, NCB, NSUB, MULTI);
#ifndef MDM_EMU
  if (LDS > 65536) {
    static bool configured[kMaxDevices] = {};
    if (const int rc = rt_dyn_lds_once(kfn, LDS, configured, stream)) return rc;
  }
#endif
  const int TR = 32 * RT, TN = x3s_tn(NCB);
  const int tpg = (group_rows + TR - 1) / TR, tiles_m = (M / group_rows) * tpg, tiles_n = (N + TN - 1) / TN;
  const int total = tiles_m * tiles_n;
#if defined(MDM_PROBES) && !defined(MDM_EMU)
  if (x3s_tl_target() >= 0) {   // timeline probe: stamps on for exactly the selected launch (stream-ordered switch)
    static int on_v[2] = {0, 1};
    const int on = (x3s_tl_count()++ == x3s_tl_target()) ? 1 : 0;
    if (hipMemcpyToSymbolAsync(HIP_SYMBOL(g_x3s_tl_on), &on_v[on], sizeof(int), 0, hipMemcpyHostToDevice, stream) != hipSuccess) return -1;
    if (on) fprintf(stderr, "[x3s timeline] launch: RT %d NSUB %d ACT %d RES %d F32 %d PLANES %d QKV %d FOLD %d OSTAT %d | M %d N %d K %d group_rows %d workgroups %d\n",
                    RT, NSUB, ACT, RES, (int)OUT_F32, (int)OUT_PLANES, (int)OUT_QKV, (int)FOLD, (int)OSTAT, M, N, K, group_rows, total);
  }
#endif
  MDM_LAUNCH(kfn, dim3(total), dim3(64 * X3S_WAVES), LDS, stream, A, W, ep, M, N, K, group_rows, tpg, tiles_n, total);
  return 0;
}

// the GEMM kinds of launch_gemm_x3_ln (gemm_x3.h), plus kind 6 = layer 0's in_proj (no folded LayerNorm)
template <int RT, int NCB>
inline int launch_gemm_x3s_rt(int kind, const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N, int K,
                              int group_rows, hipStream_t s) {
  // K-chunks of 128 k (8 sub-steps): 16 / 32 KB per buffer.  (256-k chunks for the 32-row tiles measured 1.5-4 % slower: 22.2 /
  // 37.3 / 52.7 vs 21.9 / 36.6 / 50.5 ms at B = 1 / 6 / 10; a third resident workgroup per CU changed nothing: r4lat5)
  constexpr int NS = 8;
  switch (kind) {
    case 0: return launch_gemm_x3s_t<RT, NCB, NS, true, ACT_NONE, 0, false, false, true, true, false, false>(A, W, ep, M, N, K, group_rows, s);
    case 6: return launch_gemm_x3s_t<RT, NCB, NS, true, ACT_NONE, 0, false, false, true, false, false, false>(A, W, ep, M, N, K, group_rows, s);
    case 1: return launch_gemm_x3s_t<RT, NCB, NS, true, ACT_NONE, 2, false, true, false, false, true, false>(A, W, ep, M, N, K, group_rows, s);
    case 2: return launch_gemm_x3s_t<RT, NCB, NS, true, ACT_NONE, 3, false, true, false, false, true, false>(A, W, ep, M, N, K, group_rows, s);
    case 3: return launch_gemm_x3s_t<RT, NCB, NS, true, ACT_GELU, 0, false, true, false, true, false, false>(A, W, ep, M, N, K, group_rows, s);
    case 4: return launch_gemm_x3s_t<RT, NCB, NS, true, ACT_NONE, 0, true, false, false, true, false, false>(A, W, ep, M, N, K, group_rows, s);
    case 5:   // InputProcess: K = 288 (263 features padded to 9 x 32) is one chunk of 18 sub-steps
      return launch_gemm_x3s_t<RT, NCB, 18, false, ACT_NONE, 1, false, true, false, false, false, true>(A, W, ep, M, N, K, group_rows, s);
    default: return -2;
  }
}
inline int launch_gemm_x3s(int kind, X3sShape sh, const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N, int K,
                           int group_rows, hipStream_t s) {
#ifdef MDM_PROBES
  if (sh.rt == 1 && sh.ncb == 2) return launch_gemm_x3s_rt<1, 2>(kind, A, W, ep, M, N, K, group_rows, s);
  if (sh.rt == 2 && sh.ncb == 2) return launch_gemm_x3s_rt<2, 2>(kind, A, W, ep, M, N, K, group_rows, s);
#endif
  if (sh.rt == 1) return launch_gemm_x3s_rt<1, 1>(kind, A, W, ep, M, N, K, group_rows, s);
  return launch_gemm_x3s_rt<2, 1>(kind, A, W, ep, M, N, K, group_rows, s);
}
#endif  // MDM_X3_KERNEL_ONLY

}  // namespace mdm

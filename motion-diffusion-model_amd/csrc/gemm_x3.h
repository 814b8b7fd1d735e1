// Split-precision ("f16x3"; round 1: "bf16x3") MFMA GEMM for the encoder's dense contractions:
//     C[m][n] = sum_k A[m][k] * W[n][k],   A = Ah + Al,  W = Wh + Wl  (16-bit planes: fp16 by default, common.h kSplitF16 / split_p16)
//             ~ sum_k  Ah*Wh + Ah*Wl + Al*Wh            three v_mfma_f32_32x32x16_f16 passes, fp32 accumulate.
//
// Why: the reference computes these `addmm`s in fp32 (model/mdm.py:77-84 -> torch TransformerEncoderLayer) and
// BASELINE's parity bar is 1e-3 max-abs over a 50-step guided trajectory.  gfx950 has no TF32; exact-fp32 MFMA
// peaks at 157 TFLOP/s, fp16 / bf16 MFMA at 2.5 PFLOP/s, so three 16-bit passes carry a ~2^-22- (fp16) / ~2^-16- (bf16) relative fp32 product at up to
// ~5x the fp32 rate.  Replaces in_proj / out_proj / linear1 / linear2 and,
// with K / N padded, InputProcess and OutputProcess (rows a12, a16); the LayerNorms between them are folded into the
// epilogues (X3Epilogue).  The exact-fp32 kernel (gemm_f32.h) remains the `f32` mode.
//
// Data layout.  Activations: two 16-bit planes [rows][K] (hi, lo), K contiguous, written by the PRODUCING kernel's
// epilogue.  Weights: split ONCE (mdm_prepare), as hi / lo of w * 2^8 (common.h kX3WeightScale; the epilogue scales the
// accumulators back), and stored in MFMA-FRAGMENT order
//     Wp[plane][n/32][k/16][lane 0..63][8]      lane = (n%32) + 32*((k%16)/8),  element j = k%8
// so the B-operand fragment of one wave for one 16-deep k sub-step is ONE contiguous, perfectly coalesced 1 KB
// global_load_dwordx4 -- weights never pass through LDS (a wave's 32 output columns are private to it, so staging them
// in LDS bought nothing and cost an LDS-DMA write plus an LDS read per byte).
//
// Machine mapping (gfx950).  WAVES waves per PERSISTENT workgroup, each wave owning 32 output columns x all 7 row
// sub-tiles of the tile (7 accumulators = 112 VGPRs):
//     WAVES = 8 (default)  224 x 256 tiles, one workgroup per CU (66-97 KB of LDS, <= 256 VGPRs)
//     WAVES = 4            224 x 128 tiles, two independent workgroups per CU, which hide each other's barriers, LDS-DMA
//                          latency and epilogues -- and re-read the activation panels twice as often from L2.
// Whole-bench A/B on one box: 307 vs 302 motions/s.  The two shapes, four instruction schedules and two wave-priority
// schemes all measure within 0.1-3 % of each other: while this kernel runs the chip sits at its power limit (zero-filled
// operands: +16-22 %), and the costs of the parts add up instead of overlapping -- MFMA-only 141 us + epilogue stores
// 25-40 + fragment reads / barriers 18 + loads 32 = 230 us for in_proj (profiles/r01c_final.md).  What pays is less
// work and fewer bytes.
//   * the row extent of a tile is a whole number of token sequences (S = 197 -> one sequence per tile), so the headline
//     shape (256 sequences, N in {512, 1024, 1536}) gives every CU exactly N/256 equal tiles;
//   * A tile: global -> LDS by global_load_lds_dwordx4, BK = 32, two stages of Ah|Al [224][32] = 28 KB; the pieces a
//     wave issues per step ride BETWEEN the MFMA units; the stream has its own (tile, k) cursor one step ahead and
//     rolls over into the workgroup's next tile, so the pipeline never drains at a tile boundary;
//   * LDS image: row-major, 64-byte rows, 16-byte chunk index XOR-swizzled with (row>>2)&3 (conflict-free ds_read_b128
//     groups); LDS-DMA writes lane-linearly, so the swizzle is applied to the per-lane SOURCE address and to the reads
//     (cdna_hip_programming.md rule 21).  Fragment reads are issued through untracked inline-asm ds_reads retired by
//     counted lgkmcnt waits (common.h lds_read16): hipcc only ever emits lgkmcnt(0) beside an LDS-DMA;
//   * W fragments for step g+1 are fetched into registers during step g (plain loads, waited with the step's vmcnt);
//   * XCD-aware tile order keeps the workgroups that share an activation row panel on one XCD's L2.
#pragma once
#include "attention_x3.h"  // QkvPlanes: the in_proj epilogue writes the attention kernel's operand planes
#include "common.h"
#include "gemm_f32.h"  // ACT_* enums
#include <cstdlib>

namespace mdm {

constexpr int X3_TM = 224, X3_BK = 32;
// WAVES waves per workgroup, each owning 32 output columns: 4 -> 224x128 tiles, two workgroups per CU;
//                                                            8 -> 224x256 tiles, one workgroup per CU (half the
//                                                                 activation re-reads from L2, waves in lock-step)
constexpr int x3_tn(int waves) { return 32 * waves; }
constexpr int X3_MSUB = X3_TM / 32;                              // 7 row sub-tiles
constexpr int X3_A_BYTES = X3_TM * X3_BK * 2;                    // one A plane tile: 14336
constexpr int X3_A_STAGE = 2 * X3_A_BYTES;                       // Ah|Al: 28672
constexpr int X3_A_RING = 2;                                     // stages of the step-synchronous k-loop
constexpr int X3_PIPE_RING = 4;                                  // stages of the pipelined k-loop (PIPE): A runs 3 steps ahead
// -DMDM_X3_EPI_AHEAD=1: the epilogue reads round j+1's patch BEFORE it finishes round j (a wave's LDS operations execute in order, so the
// one patch is enough: read j+1, then write j+2 behind it) -- the patch round trip is two rounds of VALU work away from its use
// instead of one wait away (0-4 more VGPRs).  Measured NEUTRAL on the headline (GEMM class 291.9, 291.5 vs 291.2, 292.7 ms per loop,
// profiles/r05j_epilogue_ahead.md): this kernel's epilogue does not wait for its 28 LDS round trips -- which also closes the 16-row-round
// variant of VERDICT r04 -- so the switch stays off; gemm_x3s.h's twin gains 0.3-0.5 % and is on
#ifndef MDM_X3_EPI_AHEAD
#define MDM_X3_EPI_AHEAD 0
#endif
constexpr bool X3_EPI_AHEAD = MDM_X3_EPI_AHEAD != 0;
// -DMDM_X3_PIPE_BADWAIT: a deliberately too lenient middle-of-step wait -- the check that the emulator's LATE mode
// (tests/emu/hip_emu.h) really catches a wrong count (profiles/r03a_pipe_emulator.md); never defined in a product build
#ifdef MDM_X3_PIPE_BADWAIT
constexpr int X3P_MID_SLACK = 8;
#else
constexpr int X3P_MID_SLACK = 0;
#endif
// Counted vmcnt waits of the pipelined loop.  ORDERED: the queue retires in issue order across both kinds of operation it
// holds (LDS-DMA pieces of A, VGPR loads of W) -- the count is the number of YOUNGER operations of either kind.  STRICT
// (-DMDM_X3_PIPE_STRICT): only operations of the SAME kind are assumed to retire in order -- the count is the number of younger
// operations of the awaited kind alone, i.e. the wait also holds if every operation of the other kind has already retired.
#if defined(MDM_X3_PIPE_WDRAIN)     // (bisection build: both waits drain the whole queue)
constexpr int X3P_WAIT_WS = 0, X3P_WAIT_MID = 0;
#elif defined(MDM_X3_PIPE_STRICT)
constexpr int X3P_WAIT_WS = 6, X3P_WAIT_MID = 3;
#else
constexpr int X3P_WAIT_WS = 12, X3P_WAIT_MID = 7;
#endif
constexpr int x3_patch_base(int ring) { return ring * X3_A_STAGE; }   // 57344 (2 stages) / 114688 (4)
constexpr int X3_PATCH_BYTES = 8 * 32 * 4;                       // per wave: 8 rows x 32 columns fp32
constexpr int X3_TAB_BYTES = X3_TM * 8;                          // one (mean, rstd) table of the tile's rows
// LDS after the patches: the (mean, rstd) table of the tile's rows (FOLD or RES == 3: a kernel has one of them) and the
// raw partial sums it is built from (<= 4 partials per row: 7 KB, the LDS-DMA lands whole KBs), both double-buffered by
// tile parity, then the per-wave partial sums of OSTAT
constexpr int x3_tab_base(int waves, int ring = X3_A_RING) { return x3_patch_base(ring) + waves * X3_PATCH_BYTES; }
constexpr int X3_RAW_BYTES = 7 * 1024;
constexpr int x3_raw_base(int waves, int ring = X3_A_RING) { return x3_tab_base(waves, ring) + 2 * X3_TAB_BYTES; }      // tables: 2 (tile parity)
constexpr int x3_part_base(int waves, int ring = X3_A_RING) { return x3_raw_base(waves, ring) + 2 * X3_RAW_BYTES; }    // raw partials: 2 (parity)
// last: the epilogue's per-column vectors of the tile (bias, folded column sums, residual gamma, beta: 4 x 256 floats),
// fetched by LDS-DMA one tile ahead, double-buffered by tile parity
constexpr int X3_CVEC_BYTES = 4 * 1024;
constexpr int x3_cvec_base(int waves, bool ln, int ring = X3_A_RING) {
  return ln ? x3_part_base(waves, ring) + waves * X3_TAB_BYTES : x3_tab_base(waves, ring);
}
// 69632 (4 waves) / 73728 (8); with the LayerNorm tables 95232 (8); the 4-stage pipelined form 131072 / 152576
constexpr int x3_lds_bytes(int waves, bool ln, int ring = X3_A_RING) {
  return x3_cvec_base(waves, ln, ring) + 2 * X3_CVEC_BYTES;
}
constexpr int X3_A_GROUPS = X3_A_STAGE / 1024;                   // 28 LDS-DMA wave-instructions per stage
constexpr int x3_a_pieces(int waves) { return (X3_A_GROUPS + waves - 1) / waves; }  // 7 (4 waves) / 4 (8 waves)

struct X3Operand {   // activations: [rows][K] planes
  const p16_t* hi;
  const p16_t* lo;
};
struct X3Weights {   // fragment-ordered planes (see header); rows padded to a multiple of 32
  const p16_t* hi;
  const p16_t* lo;
};
inline size_t x3_packed_weight_elems(int N, int K) { return (size_t)((N + 31) / 32 * 32) * K; }

// v = act(acc + bias[n]) * (n < scale_cols ? col_scale : 1) + (res ? res[m][n] : 0) -> fp32 out and/or split planes,
// or (OUT_QKV) the attention operand planes of attention_x3.h.
struct X3Epilogue {
  float* out;        // [M][ld] or null
  const float* bias;
  const float* res;  // RES == 1: fp32 residual [M][ld]; may alias out
  const p16_t* resh;  // RES == 2: the residual as hi/lo planes [M][ld] (value = hi + lo)
  const p16_t* resl;
  p16_t* oh;        // [M][ld] split planes or null
  p16_t* ol;
  int ld;
  int scale_cols;
  float col_scale;
  QkvPlanes qkv;     // OUT_QKV only
  int S, D;          // OUT_QKV only: tokens per sequence (= rows per tile), model width (N = 3 D)
  // ---- LayerNorm folded into the GEMMs around it (no LayerNorm kernel, no normalised copy of the residual stream):
  // the producer of a pre-norm sum x (out_proj / linear2, OSTAT) writes x as planes plus, per row and column tile, the
  // partial statistics (sum x, sum (x - tile mean)^2) over its columns; every consumer merges them into mean / rstd
  // (Chan's pairwise update: no E[x^2] - mean^2 cancellation).  OSTAT needs N % 256 == 0.
  //   FOLD  the A operand is x itself and the weights were pre-multiplied by gamma (mdm_prepare), so
  //         W.LN(x) + b = rstd * (W'.x - mean * colsum) + b'       with colsum[n] = sum_k W'[n][k], b' = b + W.beta
  //   RES 3 the residual is LN(x) = (x - mean) * rstd * gamma + beta, rebuilt on the fly from x's planes
  const float* astat;    // FOLD: partial sums of the A rows [M][stat_parts][2]
  const float* colsum;   // FOLD: [N]
  const float* rstat;    // RES == 3: partial sums of the residual rows [M][stat_parts][2]
  const float* rgamma;   // RES == 3: [ld]
  const float* rbeta;
  float* ostat;          // OSTAT: [M][tiles_n][2] partial sums of the rows this launch writes
  int stat_parts;        // partials per row in astat / rstat
  float inv_dim;         // 1 / (normalised width) for astat / rstat
  // ---- EMBED (InputProcess, mdm.py:343-349 + :251-252): GEMM row m = (b, t) of [B*T]; the value gets the positional row
  // res[(1 + t) * ld + n] added and is written, as planes, to token row b*S + 1 + t of every branch (S = emb_T + 1)
  int emb_T, emb_B, emb_nbranch;
  float acc_scale = kX3AccScale;   // accumulators -> value: undoes the 2^8 the weight planes carry (common.h kX3WeightScale)
  int stat_cols = 256;             // columns each partial of astat / rstat covers: 256 (this kernel's OSTAT), 128 (gemm_x3s.h)
};

constexpr bool x3_has_col_scale(int act, int res) { return act == 0 && (res == 0 || res == 1); }

// exact-GELU with erf from Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7, i.e. fp32-rounding class): one v_exp, one v_rcp
// and a 5-term Horner chain instead of the ~40-instruction libm erff; used only in this split-precision path.  (Round 3 tried
// the odd rational x P(x^2) / Q(x^2) on |x| <= 4 -- ten packed FMAs and a single v_rcp per element, |err| < 4.5e-7: same-box
// A/B 0.35 % SLOWER over the whole sampling loop, profiles/r03c_ab.md; not kept.)
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
#ifdef MDM_EMU
  const float t = 1.0f / (1.0f + 0.3275911f * z);
#else
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);  // v_rcp_f32 (1 ulp) instead of the ~10-instruction IEEE divide
#endif
  float p = 1.061405429f;
  p = p * t - 1.453152027f;
  p = p * t + 1.421413741f;
  p = p * t - 0.284496736f;
  p = p * t + 0.254829592f;
#ifdef MDM_EMU
  const float erfc_z = p * t * expf(-z * z);
#else
  const float erfc_z = p * t * __expf(-z * z);
#endif
  const float erf_abs = 1.0f - erfc_z;
  return 0.5f * x * (1.0f + copysignf(erf_abs, x));
}

// fp32 [N][K] -> fragment-ordered hi/lo planes (rows >= N zero).  One thread per 8 consecutive k of one row.
// `overflow` (device int, may be null): set to 1 when a hi element is not a finite 16-bit number, i.e. |w * 2^8| left the
// plane format's range (fp16: |w| >= 255.9) -- such a matrix cannot be carried by the split arithmetic (mdm_weights_in_range).
__global__ __launch_bounds__(256) void pack_weight_planes_kernel(const float* __restrict__ w, p16_t* __restrict__ hi,
                                                                 p16_t* __restrict__ lo, int N, int K, int* overflow) {
  const int npad = (N + 31) / 32 * 32, k8n = K / 8;
  const size_t total = (size_t)npad * k8n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / k8n), k8 = (int)(i - (size_t)n * k8n);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (n < N) ? w[(size_t)n * K + 8 * k8 + j] * kX3WeightScale : 0.f;
    p16x8 h8, l8;
    split8(v, h8, l8);
    if (overflow != nullptr) {
      bool bad = false;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float back = p16_to_f32((p16_t)h8[j]);
        bad = bad || !(fabsf(back) <= 3.0e38f);   // inf or NaN (also when the fp32 weight itself is not finite)
      }
      if (bad) *overflow = 1;   // benign race: every writer stores the same value
    }
    const int kstep = k8 >> 1, half = k8 & 1, lane = (n & 31) + 32 * half;
    const size_t o = (((size_t)(n >> 5) * (K / 16) + kstep) * 64 + lane) * 8;
    *reinterpret_cast<p16x8*>(hi + o) = h8;
    *reinterpret_cast<p16x8*>(lo + o) = l8;
  }
}

// epilogue rounds (8 rows each) of row sub-tile t: 4, or 2 for the 16-row last sub-tile; and how many rounds' worth of
// residual loads are younger than sub-tile t's when RR sub-tiles are kept in flight + in use
constexpr int x3_res_rounds(int t, bool t16) { return (t16 && t == X3_MSUB - 1) ? 2 : 4; }
constexpr int x3_res_younger_rounds(int t, int rr, bool t16) {
  int n = 0;
  for (int q = t + 1; q <= t + rr - 1 && q < X3_MSUB; ++q) n += x3_res_rounds(q, t16);
  return n;
}

// ABL & 128 (timing experiment): cycles (s_memtime) wave 0 of every workgroup spends [0] in the end-of-step vmcnt(0) of a
// tile's FIRST k step -- which also drains the previous tile's epilogue stores --, [1] in the same wait of all other steps,
// [2] in whole k-loops, [3] in whole epilogues; [4] tiles, [5] k steps.  Read with mdm_debug_get.
#if !defined(MDM_EMU) && defined(MDM_PROBES)
#define MDM_X3_DBG 1
__device__ unsigned long long g_x3_dbg[8];
__device__ __forceinline__ unsigned long long x3_now() { return __builtin_readcyclecounter(); }
#endif

// The A load stream: which tile of this workgroup and which k step it fetches next, and the per-lane source element
// offsets of the wave's seven LDS-DMA pieces for that tile.
template <int PIECES>
struct X3Cursor {
  int v;        // virtual tile id (blockIdx.x + j * gridDim.x)
  int k;        // next k step
  uint32_t off[PIECES];
};

// RES: 0 = no residual, 1 = fp32 residual, 2 = residual held as 16-bit hi/lo planes.
// ABL (profiling experiments only, 0 in production): 1 = no epilogue stores, 2 = no loads after the prologue,
// 4 = no MFMAs, 8 = loads issued but not waited for, 16 / 32 / 64 / 256 = timing probes described where they are used.
// FOLD / OSTAT / RES == 3: LayerNorm folded into the GEMMs (X3Epilogue).
// T16: the tile is 208 rows -- six 32-row sub-tiles plus ONE 16-row sub-tile (rows 192-207) on v_mfma_f32_16x16x32_bf16 --
// for row extents <= 208 (S = 197: 11 pad rows instead of 27, i.e. 6.5 of 7 units of matrix work and 13 of 14 A groups).
// The 16-row sub-tile needs the wave's W fragments in the 16x16x32 operand layout; they are derived from the 32x32x16
// fragments in registers by two lane swaps per dword (common.h frag32_to_frag16), not fetched a second time.
// F6: the k-loop runs the "f16f6" arithmetic of gemm_f16f6.h on the same skeleton -- plane 0 holds fp16 values, plane 1 the MX-FP6
// records, the weight planes come from pack_weight_f16f6_kernel; units are ordered sub-tile-major so that a sub-tile's two
// 16-byte record reads (k sub-steps 0 and 1) meet in ONE scaled MFMA; 2 + 1 MFMAs per sub-tile and step instead of 6.
// NCB (round 4, "wide" form): 32-column blocks per wave.  NCB = 2 with WAVES = 4 is the one-wave-per-SIMD arrangement of the SAME
// 208 x 256 tile: four waves x 64 columns, ONE workgroup per CU, up to 512 registers per lane (accumulators 208, W slots 64), every
// A fragment read from LDS feeds two column blocks -- half the LDS fragment traffic and half the rendezvous partners of the
// 8-wave form.  Pipelined loop only.
template <int WAVES, int ACT, int RES, bool OUT_F32, bool OUT_PLANES, bool OUT_QKV, int ABL, bool FOLD = false,
          bool OSTAT = false, bool EMBED = false, bool T16 = false, bool F6 = false, bool PIPE = false, int NCB = 1>
__global__ __launch_bounds__(64 * WAVES, NCB == 2 ? 1 : 2) void gemm_x3_kernel(X3Operand A, X3Weights W, X3Epilogue ep, int M, int N,
                                                                     int K, int rows_per_tile, int tiles_n, int total) {
  MDM_DYN_SMEM(unsigned char, lds);
  static_assert(!PIPE || (T16 && (WAVES == 8 || (WAVES == 4 && NCB == 2)) && !F6), "the pipelined k-loop exists for 208-row (T16), 256-column tiles");
  static_assert(NCB == 1 || (NCB == 2 && PIPE && WAVES == 4), "two column blocks per wave: the pipelined 4-wave form only");
  constexpr int RINGN = PIPE ? X3_PIPE_RING : X3_A_RING;   // A stages in LDS
  constexpr int NBLK = WAVES * NCB;                        // 32-column blocks of a tile (the LDS layout is per block)
  constexpr int X3_WAVES = WAVES, X3_TN = 32 * NBLK, X3_A_PIECES = x3_a_pieces(WAVES);
  constexpr int NT32 = T16 ? X3_MSUB - 1 : X3_MSUB;     // 32-row sub-tiles
  constexpr int NROUNDS = 4 * NT32 + (T16 ? 2 : 0);     // epilogue rounds of 8 rows
  using Cursor = X3Cursor<X3_A_PIECES>;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
#ifdef MDM_EMU
  const int wid = tid >> 6;
#else
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
  const int r = lane & 31, h = lane >> 5;
  const int nk = K / X3_BK;
  const int gstride = (int)gridDim.x;

  auto tile_origin = [&](int v, int& m0, int& n0) {
    const int lid = xcd_remap(v, total);
    const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
    m0 = tile_m * rows_per_tile;
    n0 = tile_n * X3_TN;
  };

  // ---- LDS-DMA sources.  A stage image = 28 groups of 1 KB (16 rows x 64 B): groups 0-13 Ah, 14-27 Al.  Wave w issues
  // groups w, w+4, ..., w+24.  Lane -> (row = lane>>2, stored chunk = lane&3); the logical k-chunk it fetches is
  // stored ^ ((row>>2)&3) = (lane&3) ^ ((lane>>4)&3).  Rows past the tile / matrix are clamped (never stored).
  const int schunk = (lane & 3) ^ ((lane >> 4) & 3);
  auto aim_a = [&](Cursor& c) {
    int m0, n0;
    tile_origin(c.v, m0, n0);
#pragma unroll
    for (int i = 0; i < X3_A_PIECES; ++i) {
      // PIPE: the 26 groups a 208-row tile reads (13 per plane) are dealt round-robin: piece j = wid + 8 i -> group j (hi) /
      // j + 1 (lo, stage groups 14..26); waves 0 and 1 issue four pieces per step, the others three
      const int jj = min(wid + X3_WAVES * i, 25);
      const int q = PIPE ? (jj < 13 ? jj : jj + 1) : min(wid + X3_WAVES * i, X3_A_GROUPS - 1);
      const int ga = (q < 14) ? q : q - 14;
      const int arow = min(m0 + ga * 16 + (lane >> 2), M - 1);
      if constexpr ((ABL & 16) != 0) {
        // timing experiment: a k-BLOCKED plane layout ([row/16][k/32][16 rows][32 k]) would make every piece one
        // contiguous KB; the data fetched here is wrong, only the address pattern is representative
        c.off[i] = (uint32_t)(arow >> 4) * (uint32_t)(K / 32) * 512u + (uint32_t)(arow & 15) * 32u + schunk * 8;
      } else {
        c.off[i] = (uint32_t)arow * (uint32_t)K + schunk * 8;
      }
    }
  };
  auto piece_a = [&](const Cursor& c, int i, int buf) {
    if constexpr (PIPE) {
      const int j = wid + X3_WAVES * i;
      const int q = j < 13 ? j : j + 1;
      if (j < 26) glds16(((q < 14) ? A.hi : A.lo) + c.off[i] + c.k * X3_BK, lds + buf * X3_A_STAGE + q * 1024);
    } else {
      const int q = wid + X3_WAVES * i;
      if (q < X3_A_GROUPS && !(T16 && (q == 13 || q == 27)))   // T16: rows 208-223 of the stage are never read
        glds16(((q < 14) ? A.hi : A.lo) + c.off[i] + c.k * ((ABL & 16) ? 512 : X3_BK), lds + buf * X3_A_STAGE + q * 1024);
    }
  };
  // past its last tile the stream simply re-fetches that tile (one wasted stage per workgroup): no "anything left to
  // load" branches in the step body
  auto advance_a = [&](Cursor& c) {
    if (++c.k == nk) {
      c.k = 0;
      if (c.v + gstride < total) { c.v += gstride; aim_a(c); }
    }
  };

  // ---- W fragments of this wave for k step `k` of tile `v`: four 16-byte loads per lane, each a contiguous 1 KB per wave
  const size_t wk16 = (size_t)(K / 16);
  auto load_w = [&](int v, int k, p16x8 (&fh)[2], p16x8 (&fl)[2]) {
    int m0, n0;
    tile_origin(v, m0, n0);
    const size_t nb = (size_t)((n0 >> 5) + wid);
    const size_t o = ((nb * wk16 + 2 * (size_t)k) * 64 + lane) * 8;
    fh[0] = *reinterpret_cast<const p16x8*>(W.hi + o);
    fl[0] = *reinterpret_cast<const p16x8*>(W.lo + o);
    fh[1] = *reinterpret_cast<const p16x8*>(W.hi + o + 512);
    fl[1] = *reinterpret_cast<const p16x8*>(W.lo + o + 512);
  };

  // ---- fragment read offsets (bytes inside a plane tile): row*64 + ((ksub*2 + h) ^ sw)*16, sw = (row>>2)&3
  const int sw = (r >> 2) & 3;
  const int fa = r * 64;                 // + t*2048 per row sub-tile
  // 16-row sub-tile (T16): lane -> (row 192 + (lane&15), k chunk lane>>4)
  const int r16 = lane & 15, g16 = lane >> 4;
  const int fa16 = (192 + r16) * 64 + ((g16 ^ ((r16 >> 2) & 3)) * 16);
#ifndef MDM_EMU
  const uint32_t lds_base = lds_addr_of(lds);
#endif

  int v = (int)blockIdx.x;
  if (v >= total) return;
#if defined(MDM_PROBES) && !defined(MDM_EMU)
  // (probe build, mdm_debug_set(8, d): the SECOND workgroup of every CU -- the upper half of the grid in dispatch order -- starts
  // d * 64 cycles late, so that the two 4-wave workgroups of a CU run their k-loops and epilogues in anti-phase:
  // tools/gemm_dephase_probe.py, profiles/r04b_ab.md)
  if constexpr (WAVES == 4 && !EMBED) {
    if (ep.emb_B > 1 && (int)blockIdx.x >= gstride / 2) {
      const unsigned long long t0 = __builtin_readcyclecounter();
      while (__builtin_readcyclecounter() - t0 < (unsigned long long)ep.emb_B * 64ULL) __builtin_amdgcn_s_sleep(8);
    }
  }
#endif
  Cursor ca{v, 0, {}};
  aim_a(ca);
  int wv = v, wkk = 0;   // the W stream's (tile, k): one step ahead of the MFMAs, like the A stream

  // Pipeline invariant: at the top of global step g the LDS holds A(g) [landed, visible] and the registers wh/wl hold
  // W(g).  During the step A(g+1) is issued into the other stage (its previous content A(g-1) was last read before the
  // barrier that ended step g-1) and W(g+1) is fetched into wnh/wnl; the step ends with vmcnt(0) + one barrier.
  // LayerNorm row statistics: LDS-DMA of the producer's partial sums of the 224 rows starting at m0 (contiguous:
  // [row][part][2] floats) into raw buffer `par`; waves 0..npieces-1 move 1 KB each
  auto stats_dma = [&](int m0s, int par) {
    const float* st = FOLD ? ep.astat : ep.rstat;
    const int npieces = (X3_TM * ep.stat_parts * 8 + 1023) / 1024;
    for (int pc = wid; pc < npieces; pc += X3_WAVES) {      // (<= 7 pieces; four waves take two)
      // A 16-byte unit keeps its place in the table (unit u of the tile = floats 4u .. 4u+3 behind the tile's first row) as long
      // as it holds any float of the matrix; units entirely past it read a valid address instead.  With an odd number of
      // partials per row (D = 256, 768) a tile that starts on an odd row is only 8-byte aligned and its last unit may straddle
      // the end of the array by up to 12 bytes: the workspace carves every statistics array with 16 spare bytes for this
      // (api_launch.h carve).  (Round 3 clamped straddling units to the last 16 bytes of the array, which moved the last row's
      // partials to the wrong table slot: wrong last token row for D = 256 when the last tile starts on an odd row.)
      const long long total_f = (long long)M * ep.stat_parts * 2;
      long long fo = (long long)m0s * ep.stat_parts * 2 + 4LL * (64 * pc + lane);
      if (fo >= total_f) fo = 0;                                       // rows past the matrix: any valid address
      glds16(st + fo, lds + x3_raw_base(NBLK, RINGN) + par * X3_RAW_BYTES + pc * 1024);
    }
  };
  // the epilogue's per-column vectors of the tile starting at column n0c -> LDS buffer `par`: wave 0 bias, wave 1 folded
  // column sums (FOLD), waves 2 / 3 residual gamma / beta (RES == 3); 1 KB = the tile's 256 columns each (columns past N:
  // clamped source, never stored)
  constexpr bool LN_ANY = FOLD || OSTAT || RES == 3;
  auto cvec_dma = [&](int n0c, int par) {
    const float* src = nullptr;
    if (wid == 0) src = ep.bias;
    if constexpr (FOLD) { if (wid == 1) src = ep.colsum; }
    if constexpr (RES == 3) { if (wid == 2) src = ep.rgamma; if (wid == 3) src = ep.rbeta; }
    if (src != nullptr) glds16(src + min(n0c + 4 * lane, N - 4), lds + x3_cvec_base(NBLK, LN_ANY, RINGN) + par * X3_CVEC_BYTES + wid * 1024);
  };
  {
    int m0f, n0f;
    tile_origin(v, m0f, n0f);
    if constexpr (FOLD || RES == 3) stats_dma(m0f, 0);
    cvec_dma(n0f, 0);
  }
  p16x8 wh[2], wl[2], wnh[2], wnl[2];
  // PIPE: the W stream lives in four half-step slots (slot 2 * (step parity) + k sub-step; hi and lo plane fragment each),
  // refilled IN PLACE for two steps later as soon as their last MFMA has been issued -- 32 VGPRs, 1.5 steps of cover
  p16x8 wsh[4 * NCB] = {}, wsl[4 * NCB] = {};   // slot q, column block cb: [q * NCB + cb]  (zero: the first refill formally reads its slot)
  uint32_t wso = 0;           // element offset of the W stream's current step inside the fragment-ordered planes
  uint32_t wtile = 0;         // ... of its tile's first step (wave-uniform): changes only when the stream enters a new tile
  auto aim_w_tile = [&]() {
    int m0w, n0w;
    tile_origin(wv, m0w, n0w);
    wtile = (uint32_t)((n0w >> 5) + wid * NCB) * (uint32_t)wk16 * 512u;   // (column block cb: + cb * wk16 * 512)
  };
  auto aim_w = [&]() { wso = wtile + (uint32_t)wkk * 1024u + (uint32_t)lane * 8u; };
  auto advance_w = [&]() {
    if (++wkk == nk) {
      wkk = 0;
      if (wv + gstride < total) { wv += gstride; aim_w_tile(); }
    }
  };
  auto load_w_half = [&](int ks, auto q_tag) __attribute__((always_inline)) {   // in-place refill of slot q (common.h gload16_refill)
    constexpr int q = decltype(q_tag)::value;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const uint32_t o = wso + (uint32_t)cb * (uint32_t)wk16 * 512u + 512u * (uint32_t)ks;
#ifdef MDM_X3_PIPE_NOGUARD
      gload16_async(wsh[q * NCB + cb], W.hi + o);
      gload16_async(wsl[q * NCB + cb], W.lo + o);
#else
      gload16_refill(wsh[q * NCB + cb], W.hi + o);
      gload16_refill(wsl[q * NCB + cb], W.lo + o);
#endif
    }
  };
  // counted wait naming slot q's registers (all column blocks); closing wait over every slot
  auto wait_slot = [&](auto n_tag, auto q_tag) __attribute__((always_inline)) {
    constexpr int n = decltype(n_tag)::value, q = decltype(q_tag)::value;
    if constexpr (NCB == 1) vmem_wait<n>(wsh[q], wsl[q]);
    else vmem_wait<n>(wsh[2 * q], wsl[2 * q], wsh[2 * q + 1], wsl[2 * q + 1]);
  };
  auto wait_all_slots = [&]() __attribute__((always_inline)) {
    static_for<NCB>([&](auto c_tag) __attribute__((always_inline)) {
      constexpr int c4 = 4 * decltype(c_tag)::value;
      vmem_wait<0>(wsh[c4], wsl[c4], wsh[c4 + 1], wsl[c4 + 1], wsh[c4 + 2], wsl[c4 + 2], wsh[c4 + 3], wsl[c4 + 3]);
    });
  };
  int gs = 0;                 // PIPE: global k-step counter of this workgroup (stage of step g = g & 3)
#if defined(MDM_X3_PIPE_PRIO) && !defined(MDM_EMU)
  if constexpr (PIPE) { if (wid >= 4) __builtin_amdgcn_s_setprio(1); }   // (A/B build: static priority for the younger half)
#endif
  if constexpr (PIPE) {
    // A(0), A(1), A(2) into stages 0..2 and W(0), W(1) into the four slots; everything landed and visible before step 0
#pragma unroll
    for (int st = 0; st < 3; ++st) {
#pragma unroll
      for (int i = 0; i < X3_A_PIECES; ++i) piece_a(ca, i, st);
      advance_a(ca);
    }
    aim_w_tile();
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      aim_w();
      if (st == 0) { load_w_half(0, std::integral_constant<int, 0>{}); load_w_half(1, std::integral_constant<int, 1>{}); }
      else { load_w_half(0, std::integral_constant<int, 2>{}); load_w_half(1, std::integral_constant<int, 3>{}); }
      advance_w();
    }
    wait_all_slots();
    wg_barrier();
  } else {
#pragma unroll
    for (int i = 0; i < X3_A_PIECES; ++i) piece_a(ca, i, 0);
    advance_a(ca);
    load_w(wv, wkk, wh, wl);
    wait_vmem_all();
    wg_barrier();
  }

  int abuf = 0;
  int tile_parity = 0;
  for (; v < total; v += gstride, tile_parity ^= 1) {
    int m0, n0;
    tile_origin(v, m0, n0);
    f32x16 accs_[NCB][NT32];     // [column block of the wave][row sub-tile]
    f32x4 acc16s_[NCB][2];       // T16: rows 192-207 x columns 0-15 / 16-31 of each column block
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
#pragma unroll
      for (int t = 0; t < NT32; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) accs_[cb][t][e] = 0.f;
      acc16s_[cb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      acc16s_[cb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x16 (&acc)[NT32] = accs_[0];       // the one-block forms (step-synchronous loop) work on block 0
    f32x4 (&acc16)[2] = acc16s_[0];
    const int ncol0_w = n0 + wid * NCB * 32;             // this wave's first column (wave-uniform)
    // this tile's per-column epilogue vectors are in LDS buffer `tile_parity` (requested one tile ago / in the prologue: no
    // global-load latency in front of the epilogue); the next tile's are requested in this tile's SECOND k step (below),
    // i.e. behind a workgroup barrier every wave reaches only after its epilogue of the previous tile -- whose vectors
    // live in the buffer being refilled -- and land under the rest of this tile's k-loop
    const float* const cvec = reinterpret_cast<const float*>(lds + x3_cvec_base(NBLK, LN_ANY, RINGN) + tile_parity * X3_CVEC_BYTES);
    const int kt_cvec = nk > 1 ? 1 : 0;
    if (nk == 1) wg_barrier();   // single-step contractions: no k-step barrier in front of the request
    // row statistics (mean, rstd) of this tile's rows, built HERE -- where the accumulators are not live yet -- from the
    // producer's partial sums, which an LDS-DMA issued one tile ago (or in the kernel prologue) has already landed; the
    // next tile's partials are requested now and land under this tile's k-loop.  Tables and raw buffers alternate with
    // the tile parity: a fast wave may build table j+1 while a slow one still reads table j in its epilogue.
    constexpr bool LN_TABS = FOLD || RES == 3;
    float2* const stab = reinterpret_cast<float2*>(lds + x3_tab_base(NBLK, RINGN) + tile_parity * X3_TAB_BYTES);
    if constexpr (LN_TABS) {
      if (tid < X3_TM) {
        // rows of the tile past the matrix (the last sequence's pad rows) have no statistics: their raw slots hold whatever the
        // clamped DMA fetched -- for an odd M with one partial per row two floats BEHIND the buffer (uninitialised workspace:
        // a NaN there went through these rows' V^T pad keys, 0 * NaN, into the whole sequence) -- so they get (0, 0): every
        // folded value of such a row is then the finite constant b' / beta
        int m0t, n0t;
        tile_origin(v, m0t, n0t);
        const bool pad_row = m0t + tid >= M;
        const float* sraw = reinterpret_cast<const float*>(lds + x3_raw_base(NBLK, RINGN) + tile_parity * X3_RAW_BYTES);
        // partials are (sum, CENTRED sum of squares about the partial's own mean) of X3_TN columns each; merged by Chan's
        // formula -- no E[x^2] - mean^2 cancellation when a row's mean is large against its spread
        float s1 = 0.f;
        for (int p = 0; p < ep.stat_parts; ++p) s1 += sraw[(tid * ep.stat_parts + p) * 2];
        const float mean = s1 * ep.inv_dim;
        const float pcols = (float)ep.stat_cols, inv_pcols = 1.0f / pcols;   // columns per partial (256, or 128 from gemm_x3s.h)
        float m2 = 0.f;
        for (int p = 0; p < ep.stat_parts; ++p) {
          const float dm = sraw[(tid * ep.stat_parts + p) * 2] * inv_pcols - mean;
          m2 += sraw[(tid * ep.stat_parts + p) * 2 + 1] + pcols * dm * dm;
        }
        const float var = m2 * ep.inv_dim;
#ifdef MDM_EMU
        stab[tid] = pad_row ? make_float2(0.f, 0.f) : make_float2(mean, 1.0f / sqrtf(var + 1e-5f));
#else
        stab[tid] = pad_row ? make_float2(0.f, 0.f) : make_float2(mean, __builtin_amdgcn_rsqf(var + 1e-5f));   // v_rsq_f32, 1 ulp
#endif
      }
      // the NEXT tile's partials are requested BEHIND the table build: hipcc drains the vector-memory queue in front of the
      // build's LDS reads (an LDS-DMA may be pending), and issued first, this request -- a fresh HBM / L2 round trip -- was what
      // it waited for at every tile start (round 3; the other raw buffer is the target, the build does not touch it)
      if (v + gstride < total) {
        int m0n, n0n;
        tile_origin(v + gstride, m0n, n0n);
        stats_dma(m0n, tile_parity ^ 1);
      }
    }
    float2* const atab = stab;   // FOLD: statistics of the A rows;  RES == 3: of the residual rows (a kernel has one)
    float2* const rtab = stab;
#ifdef MDM_X3_DBG
    unsigned long long dbg_t0 = 0, dbg_w0 = 0, dbg_w1 = 0, dbg_bar = 0;
    if constexpr ((ABL & 128) != 0) dbg_t0 = x3_now();
#endif
    if constexpr (PIPE) {
      // ================= pipelined k-loop (round 3) =================
      // What the step-synchronous loop below pays per 32-deep step -- a full vmcnt(0) drain of loads issued at most one step
      // earlier, a rendezvous of all eight waves behind it, and a cold restart of the fragment-read pipeline (measured: a
      // step takes ~2.8 us against 1.3 us of matrix work) -- is removed by running every stream AHEAD of the matrix work:
      //   * A: four LDS stages.  During step g the pieces of A(g+3) are issued (into the stage A(g-1) lived in); each wave
      //     retires its own pieces of A(g+1) with a COUNTED vmcnt at the step's middle, then one bare s_barrier -- no drain --
      //     makes them visible, and from there on the fragment reads of step g+1 may begin: the unit pipeline never restarts
      //     inside a tile.  cover: 1.5-2 steps.
      //   * W: four half-step register slots refilled in place for step g+2 right behind their last MFMA (cover 1.5 steps),
      //     by untracked loads retired with counted waits that name the slot (hipcc would drain the LDS-DMA queue for a
      //     tracked load).
      //   * the 16-row sub-tile ("X") sits between the two k sub-steps, right behind the barrier, where both W halves of the
      //     step are valid; its two reads travel in the same in-order read queue as the units'.
      // Per step and wave the vector-memory queue therefore holds, in program order,
      //     W0(g+2) x2 | A(g+3) x nA | W1(g+2) x2          nA = 4 (waves 0, 1) or 3
      // and the waits are: step start, slot W0(g): 2 nA + 6 younger operations; middle, A(g+1) and W1(g): nA + 4.
      // Other operations that land in the queue (column vectors, row statistics, the previous tile's stores) only make
      // these waits stricter: the queue retires in order.
#ifndef MDM_X3_PIPE_DEPTH
#define MDM_X3_PIPE_DEPTH 2
#endif
      constexpr int NU = 2 * NT32, NE = NU + 1, XP = NT32, DEPTH = MDM_X3_PIPE_DEPTH, RING = DEPTH + 1;
      // counted waits, generalised: a half-step slot is LWH = 2 NCB loads, a wave issues at least NA_MIN = 26 / WAVES pieces per step
      //   step start, slot W0(g): younger = A(g+1) + W1(g) + W0(g+1) + A(g+2) + W1(g+1) = 2 NA_MIN + 3 LWH   (8 waves: 12)
      //   middle, A(g+1) and W1(g): younger = W0(g+1) + A(g+2) + W1(g+1) = NA_MIN + 2 LWH                  (8 waves: 7)
      constexpr int LWH = 2 * NCB, NA_MIN = 26 / X3_WAVES;
      constexpr int WAIT_WS = (NCB == 1 && X3_WAVES == 8) ? X3P_WAIT_WS : 2 * NA_MIN + 3 * LWH;
      constexpr int WAIT_MID = (NCB == 1 && X3_WAVES == 8) ? X3P_WAIT_MID : NA_MIN + 2 * LWH;
      static_assert((NCB == 1 && X3_WAVES == 8) || (WAIT_WS == 24 && WAIT_MID == 14), "wide form: 4 waves x 2 column blocks");
      static_assert(NU % RING == 0, "fragment ring slots must line up across steps");
      p16x8 ah[RING], al[RING], a16h, a16l;
#ifndef MDM_EMU
      const uint32_t lane_a = lds_base + fa, lane_a16 = lds_base + fa16;
      const uint32_t sw0 = (uint32_t)((h ^ sw) * 16), sw1 = (uint32_t)(((2 + h) ^ sw) * 16);
#endif
      // reads of element `ee` of the step whose A stage is `stg` (runtime, 0..3): a regular unit or the 16-row sub-tile
      auto issue_reads = [&](auto ee_tag, uint32_t stg) __attribute__((always_inline)) {
        constexpr int ee = decltype(ee_tag)::value;
        if constexpr (ee == XP) {
#ifdef MDM_EMU
          lds_read16(a16h, lds + stg * X3_A_STAGE, fa16);
          lds_read16(a16l, lds + stg * X3_A_STAGE, X3_A_BYTES + fa16);
#else
          lds_read16<0>(a16h, lane_a16 + stg * X3_A_STAGE);
          lds_read16<X3_A_BYTES>(a16l, lane_a16 + stg * X3_A_STAGE);
#endif
        } else {
          constexpr int u = ee < XP ? ee : ee - 1, ks = u / NT32, t = u - ks * NT32;
#ifdef MDM_EMU
          lds_read16(ah[u % RING], lds + stg * X3_A_STAGE, fa + t * 2048 + (((ks * 2 + h) ^ sw) * 16));
          lds_read16(al[u % RING], lds + stg * X3_A_STAGE, X3_A_BYTES + fa + t * 2048 + (((ks * 2 + h) ^ sw) * 16));
#else
          const uint32_t ad = lane_a + stg * X3_A_STAGE + (ks ? sw1 : sw0);
          lds_read16<t * 2048>(ah[u % RING], ad);
          lds_read16<X3_A_BYTES + t * 2048>(al[u % RING], ad);
#endif
        }
      };
      auto pipe_step = [&](auto par_tag) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_tag)::value;
        const uint32_t cur = (uint32_t)gs & 3u, nxt = (uint32_t)(gs + 1) & 3u, fill = (uint32_t)(gs + 3) & 3u;
        p16x8 w16h[NCB][2], w16l[NCB][2];
#ifdef MDM_X3_PIPE_NOLOOK
        static_for<DEPTH>([&](auto d_tag) __attribute__((always_inline)) { issue_reads(d_tag, cur); });
#endif
        // slot W0(g) landed?  (issued in the middle of step g-2)
        wait_slot(std::integral_constant<int, WAIT_WS>{}, std::integral_constant<int, 2 * PAR>{});
        static_for<NE>([&](auto e_tag) __attribute__((always_inline)) {
          constexpr int e = decltype(e_tag)::value;
          // ---- 1. reads of the element DEPTH ahead (past the step: units 0, 1 of step g+1, from the next stage)
          if constexpr (e + DEPTH < NE) issue_reads(std::integral_constant<int, e + DEPTH>{}, cur);
#ifndef MDM_X3_PIPE_NOLOOK   // (bisection build: the fragment pipeline restarts at every step)
          else issue_reads(std::integral_constant<int, e + DEPTH - NE>{}, nxt);
#endif
          // ---- 2. the middle of the step sits in front of the 16-row sub-tile
          if constexpr (e == XP) {
            // own pieces of A(g+1) and slot W1(g) (both issued during step g-2) landed
            wait_slot(std::integral_constant<int, WAIT_MID + X3P_MID_SLACK>{}, std::integral_constant<int, 2 * PAR + 1>{});
#if (defined(MDM_X3_PIPE_SNOP) || defined(MDM_X3_PIPE_SNOP2)) && !defined(MDM_EMU)   // (bisection build: the matrix pipe drains before slot W0 is rewritten)
            asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
#endif
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
              w16h[cb][0] = wsh[(2 * PAR) * NCB + cb]; w16h[cb][1] = wsh[(2 * PAR + 1) * NCB + cb];
              w16l[cb][0] = wsl[(2 * PAR) * NCB + cb]; w16l[cb][1] = wsl[(2 * PAR + 1) * NCB + cb];
              frag32_to_frag16(w16h[cb][0], w16h[cb][1]);
              frag32_to_frag16(w16l[cb][0], w16l[cb][1]);
            }
#ifndef MDM_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
            // slot W0 is free (its last 32-row MFMA was issued with unit NT32-1, its lane-swapped copy is taken): refill for g+2
            if constexpr (!(ABL & 2)) { aim_w(); load_w_half(0, std::integral_constant<int, 2 * PAR>{}); }
#ifdef MDM_X3_PIPE_DRAINBAR   // (bisection build: lgkmcnt(0) in front of the rendezvous)
            lds_wait<0>(a16h, a16l);
            wg_barrier();
#else
            wg_barrier_nodrain();   // A(g+1) visible to every wave; every wave is past step g-1, whose stage A(g+3) refills
#endif
#ifndef MDM_EMU
            __builtin_amdgcn_sched_barrier(0);
#endif
          }
          // ---- 3. this element's reads retired (those of the DEPTH younger elements may stay in flight)
#ifdef MDM_X3_PIPE_NOLOOK
          constexpr int issued_after = (NE - 1 - e) < DEPTH ? (NE - 1 - e) : DEPTH;
#else
          constexpr int issued_after = DEPTH;
#endif
          if constexpr (e == XP) {
            lds_wait<2 * issued_after>(a16h, a16l);
          } else {
            constexpr int u = e < XP ? e : e - 1;
            lds_wait<2 * issued_after>(ah[u % RING], al[u % RING]);
          }
#ifndef MDM_EMU
          __builtin_amdgcn_sched_barrier(0);  // the MFMAs below must not be hoisted above the wait (rule 18)
#endif
          // ---- 4. matrix work
          if constexpr (e == XP) {
            if constexpr ((ABL & 4) != 0) {
#ifndef MDM_EMU
              asm volatile("" ::"v"(a16h), "v"(a16l), "v"(w16h[0][0]), "v"(w16h[0][1]), "v"(w16l[0][0]), "v"(w16l[0][1]));
#endif
            } else {
#pragma unroll
              for (int wb = 0; wb < NCB; ++wb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                  acc16s_[wb][cb] = mfma16_p16(a16l, w16h[wb][cb], acc16s_[wb][cb]);
                  acc16s_[wb][cb] = mfma16_p16(a16h, w16l[wb][cb], acc16s_[wb][cb]);
                  acc16s_[wb][cb] = mfma16_p16(a16h, w16h[wb][cb], acc16s_[wb][cb]);
                }
            }
          } else {
            constexpr int u = e < XP ? e : e - 1, ks = u / NT32, t = u - ks * NT32;
            if constexpr ((ABL & 4) != 0) {
#ifndef MDM_EMU
              asm volatile("" ::"v"(al[u % RING]), "v"(ah[u % RING]), "v"(wsh[(2 * PAR + ks) * NCB]), "v"(wsl[(2 * PAR + ks) * NCB]));
#endif
            } else {
              // (the wave's column blocks interleaved: consecutive MFMAs on different accumulators, one A fragment pair for all)
#pragma unroll
              for (int wb = 0; wb < NCB; ++wb) accs_[wb][t] = mfma_p16(al[u % RING], wsh[(2 * PAR + ks) * NCB + wb], accs_[wb][t]);
#pragma unroll
              for (int wb = 0; wb < NCB; ++wb) accs_[wb][t] = mfma_p16(ah[u % RING], wsl[(2 * PAR + ks) * NCB + wb], accs_[wb][t]);
#pragma unroll
              for (int wb = 0; wb < NCB; ++wb) accs_[wb][t] = mfma_p16(ah[u % RING], wsh[(2 * PAR + ks) * NCB + wb], accs_[wb][t]);
            }
          }
#ifndef MDM_EMU
          __builtin_amdgcn_sched_barrier(0);  // keep the same-accumulator triple back to back (no filler inside)
#endif
          // ---- 5. one LDS-DMA piece of A(g+3) rides behind each of the four elements that follow the barrier
          if constexpr (!(ABL & 2) && e >= XP && e < XP + X3_A_PIECES) piece_a(ca, e - XP, (int)fill);
#ifndef MDM_EMU
          __builtin_amdgcn_sched_barrier(0);
#endif
        });
        // slot W1 is free: refill for g+2; then both streams move on
#if defined(MDM_X3_PIPE_SNOP2) && !defined(MDM_EMU)   // (bisection build: the matrix pipe drains before slot W1 is rewritten)
        asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
#endif
        if constexpr (!(ABL & 2)) load_w_half(1, std::integral_constant<int, 2 * PAR + 1>{});
        advance_w();
        advance_a(ca);
        ++gs;
      };
#ifdef MDM_X3_PIPE_SYNCTILE   // (bisection build: every tile starts from a drained, rendezvoused workgroup)
      wait_vmem_all();
      wg_barrier();
#endif
      // prime the fragment pipeline of this tile: its first stage was made visible by the previous step's barrier / the prologue
#ifndef MDM_X3_PIPE_NOLOOK
      static_for<DEPTH>([&](auto d_tag) __attribute__((always_inline)) { issue_reads(d_tag, (uint32_t)gs & 3u); });
#endif
      for (int kt = 0; kt < nk; kt += 2) {
        pipe_step(std::integral_constant<int, 0>{});
        // the next tile's per-column vectors: behind step 0's barrier, which every wave reaches only after its epilogue of
        // the previous tile (whose vectors live in the buffer being refilled)
        if (kt == 0 && v + gstride < total) {
          int m0n, n0n;
          tile_origin(v + gstride, m0n, n0n);
          cvec_dma(n0n, tile_parity ^ 1);
        }
        pipe_step(std::integral_constant<int, 1>{});
      }
      // the tile's last step has run two elements ahead like every other (ONE step body: a separate "last step" instance made
      // hipcc keep two copies of the accumulators, 192 VGPRs): those fragments belong to the next tile's first step, whose
      // pipeline is primed afresh behind the epilogue -- retire and drop them
#ifndef MDM_X3_PIPE_NOLOOK
      if constexpr (DEPTH == 2) lds_wait<0>(ah[0], al[0], ah[1], al[1]);
      else lds_wait<0>(ah[0], al[0], ah[1], al[1], ah[2], al[2]);
#endif
      // the epilogue must not meet a W slot whose load is still in flight (a spill would save the stale register); hipcc
      // drains the queue in front of the epilogue's first LDS read anyway (LDS-DMA pending)
      wait_all_slots();
    } else {
    for (int kt = 0; kt < nk; ++kt) {
      if (kt == kt_cvec && v + gstride < total) {
        int m0n, n0n;
        tile_origin(v + gstride, m0n, n0n);
        cvec_dma(n0n, tile_parity ^ 1);
      }
      // W(g+1): advance the W stream and fetch (past the last tile: re-fetch, like the A stream)
      if (++wkk == nk) {
        wkk = 0;
        if (wv + gstride < total) wv += gstride;
      }
      if (!(ABL & 2)) load_w(wv, wkk, wnh, wnl);
      // 14 units per stage (2 k sub-steps x 7 row sub-tiles), each = 2 A-fragment reads + 3 MFMAs, software-pipelined
      // DEPTH units deep: the reads of unit u+DEPTH are issued, then a COUNTED wait (2*DEPTH younger reads may stay in
      // flight) retires unit u's, then its 3 MFMAs go.  One LDS-DMA piece of A(g+1) rides behind each of the first seven.
      constexpr int DEPTH = 2, RING = DEPTH + 1;  // fragment-read lookahead in units
      p16x8 ah[RING], al[RING];
#ifdef MDM_EMU
      const unsigned char* sa = lds + abuf * X3_A_STAGE;
#define X3_RD_A(dst, plane, t, ks) lds_read16(dst, sa, (plane) * X3_A_BYTES + fa + (t) * 2048 + ((((ks) * 2 + h) ^ sw) * 16))
#else
      // per-lane LDS byte addresses of this stage's fragments for k sub-step 0 / 1 (the XOR swizzle moves with ks)
      const uint32_t sa0 = lds_base + abuf * X3_A_STAGE + fa;
      const uint32_t aaddr[2] = {sa0 + ((h ^ sw) * 16), sa0 + (((2 + h) ^ sw) * 16)};
#define X3_RD_A(dst, plane, t, ks) lds_read16<(plane) * X3_A_BYTES + (t) * 2048>(dst, aaddr[ks])
#endif
      // T16: the 16-row sub-tile's A fragments (one per plane covers the whole 32-deep step) are read FIRST, so they are
      // older than every unit's reads and retired by unit 0's wait; its W fragments come from wh / wl by lane swaps
      p16x8 a16h, a16l, w16h[2], w16l[2];
      if constexpr (T16) {
#ifdef MDM_EMU
        lds_read16(a16h, sa, fa16);
        lds_read16(a16l, sa, X3_A_BYTES + fa16);
#else
        const uint32_t a16addr = lds_base + abuf * X3_A_STAGE + fa16;
        lds_read16<0>(a16h, a16addr);
        lds_read16<X3_A_BYTES>(a16l, a16addr);
#endif
        w16h[0] = wh[0]; w16h[1] = wh[1];
        w16l[0] = wl[0]; w16l[1] = wl[1];
        frag32_to_frag16(w16h[0], w16h[1]);
        frag32_to_frag16(w16l[0], w16l[1]);
      }
      constexpr int NU = 2 * NT32;  // units per stage
      p16x8 f6_hold = {0, 0, 0, 0, 0, 0, 0, 0};   // F6 only
      static_for<NU + DEPTH>([&](auto u_tag) __attribute__((always_inline)) {
        constexpr int u = decltype(u_tag)::value;
        if constexpr (u < NU && (!(ABL & 64) || u % 2 == 0)) {   // 64: timing experiment -- half the fragment reads
          constexpr int ks = F6 ? (u & 1) : u / NT32, t = F6 ? (u >> 1) : u - ks * NT32;
          X3_RD_A(ah[u % RING], 0, t, ks);
          X3_RD_A(al[u % RING], 1, t, ks);
        }
        if constexpr (u >= DEPTH) {
          constexpr int uv = u - DEPTH, ks = F6 ? (uv & 1) : uv / NT32, t = F6 ? (uv >> 1) : uv - ks * NT32;
          // reads allowed to stay in flight: those of the (up to) DEPTH younger units
          constexpr int younger = 2 * ((NU - 1 - uv) < DEPTH ? (NU - 1 - uv) : DEPTH);
          if constexpr (T16 && uv == 0) lds_wait<younger>(ah[uv % RING], al[uv % RING], a16h, a16l);
          else lds_wait<younger>(ah[uv % RING], al[uv % RING]);
#ifndef MDM_EMU
          __builtin_amdgcn_sched_barrier(0);  // the MFMAs below must not be hoisted above the wait (rule 18)
#endif
          if constexpr ((ABL & 4) != 0 || ((ABL & 32) != 0 && uv == NU - 1)) {
            // 4: no MFMAs at all; 32: timing experiment -- one unit of 14 skipped = the MFMA work a 16-row last sub-tile
            // (208 instead of 224 rows per tile) would save
#ifndef MDM_EMU
            asm volatile("" ::"v"(al[uv % RING]), "v"(ah[uv % RING]), "v"(wh[ks]), "v"(wl[ks]));
#endif
          } else if constexpr (F6) {
            // main term on fp16; the record's first 16 bytes (code dwords c0-c3) wait in f6_hold for the second read
            // (c4, c5, scale) of the same sub-tile, then ONE scaled MFMA adds both cross terms of the 32-k block
            if constexpr (ks == 0) {
              f6_hold = al[uv % RING];
              acc[t] = mfma_f16(__builtin_bit_cast(f16x8, ah[uv % RING]), __builtin_bit_cast(f16x8, wh[0]), acc[t]);
            } else {
              acc[t] = mfma_f16(__builtin_bit_cast(f16x8, ah[uv % RING]), __builtin_bit_cast(f16x8, wh[1]), acc[t]);
              const u32x4 c0 = __builtin_bit_cast(u32x4, f6_hold), c1 = __builtin_bit_cast(u32x4, al[uv % RING]);
              const u32x4 w0 = __builtin_bit_cast(u32x4, wl[0]), w1 = __builtin_bit_cast(u32x4, wl[1]);
              const i32x8 a6 = {(int)c0[0], (int)c0[1], (int)c0[2], (int)c0[3], (int)c1[0], (int)c1[1], 0, 0};
              const i32x8 w6 = {(int)w0[0], (int)w0[1], (int)w0[2], (int)w0[3], (int)w1[0], (int)w1[1], 0, 0};
              acc[t] = mfma_mx_fp6(a6, w6, acc[t], (int)c1[2], (int)w1[2]);
            }
          } else if constexpr ((ABL & 256) != 0) {
            // 256: timing experiment -- the INSTRUCTION MIX of the f16f6 scheme (gemm_f16f6.h) inside this kernel's skeleton:
            // per row sub-tile and 32 k, two main-term MFMAs (fp16 and bf16 run at the same rate) and ONE scaled MX-FP6 MFMA
            // (both cross terms of 32 k fill its K = 64); loads, LDS traffic and barriers as in production (the planned plane
            // records have the bytes of today's lo plane).  Operands of the scaled MFMA are whatever bits the lo fragments
            // hold: results are garbage, only the time is representative.
            acc[t] = mfma_p16(ah[uv % RING], wh[ks], acc[t]);
#ifndef MDM_EMU
            if constexpr (ks == 0) {
              const u32x4 qa0 = __builtin_bit_cast(u32x4, al[uv % RING]), qa1 = __builtin_bit_cast(u32x4, ah[uv % RING]);
              const u32x4 qb0 = __builtin_bit_cast(u32x4, wl[0]), qb1 = __builtin_bit_cast(u32x4, wl[1]);
              const i32x8 qa = {(int)qa0[0], (int)qa0[1], (int)qa0[2], (int)qa0[3], (int)qa1[0], (int)qa1[1], 0, 0};
              const i32x8 qb = {(int)qb0[0], (int)qb0[1], (int)qb0[2], (int)qb0[3], (int)qb1[0], (int)qb1[1], 0, 0};
              acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[t], 2, 2, 0, 120 + (lane & 7), 0, 121 + (lane & 3));
            }
#endif
          } else {
            acc[t] = mfma_p16(al[uv % RING], wh[ks], acc[t]);
            acc[t] = mfma_p16(ah[uv % RING], wl[ks], acc[t]);
            acc[t] = mfma_p16(ah[uv % RING], wh[ks], acc[t]);
            if constexpr (T16 && uv == 0) {
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) {
                acc16[cb] = mfma16_p16(a16l, w16h[cb], acc16[cb]);
                acc16[cb] = mfma16_p16(a16h, w16l[cb], acc16[cb]);
                acc16[cb] = mfma16_p16(a16h, w16h[cb], acc16[cb]);
              }
            }
          }
#ifndef MDM_EMU
          __builtin_amdgcn_sched_barrier(0);  // keep the same-accumulator triple back to back (no filler inside)
#endif
          if constexpr (!(ABL & 2) && uv < X3_A_PIECES) piece_a(ca, uv, abuf ^ 1);
#ifndef MDM_EMU
          __builtin_amdgcn_sched_barrier(0);
#endif
        }
      });
#undef X3_RD_A
      advance_a(ca);
#ifdef MDM_X3_DBG
      if constexpr ((ABL & 128) != 0) {
        const unsigned long long a0 = x3_now();
        wait_vmem_all();
        const unsigned long long a1 = x3_now();
        if (kt == 0) dbg_w0 += a1 - a0; else dbg_w1 += a1 - a0;
      }
#endif
      if (!(ABL & 8)) wait_vmem_all();   // 8: experiment -- loads issued but never waited for (results are garbage)
#ifdef MDM_X3_DBG
      if constexpr ((ABL & 128) != 0) {
        const unsigned long long b0 = x3_now();
        wg_barrier();
        dbg_bar += x3_now() - b0;
      }
#endif
      wg_barrier();
      abuf ^= 1;
      if (!(ABL & 2)) {
        wh[0] = wnh[0]; wh[1] = wnh[1]; wl[0] = wnl[0]; wl[1] = wnl[1];
      }
    }
    }   // !PIPE

    // ---- epilogue.  In the accumulator layout a lane owns ONE column and 16 rows of each 32x32 sub-tile, which would
    // mean 4-byte (fp32) / 2-byte (planes) stores: the store tail is issue-bound (cdna_hip_programming.md T21).  So
    // each wave transposes 8 rows x 32 columns at a time (accumulator registers 4g..4g+3 of both lane halves) through
    // its private 1 KB LDS patch -- disjoint from the A stages, which already hold the next tile's first stage
    // -- and writes 16 bytes per lane: lane -> (row = lane>>3, 4 consecutive columns).
#ifdef MDM_X3_DBG
    unsigned long long dbg_t1 = 0;
    if constexpr ((ABL & 128) != 0) dbg_t1 = x3_now();
#endif
    static_for<NCB>([&](auto cbk_tag) __attribute__((always_inline)) {   // the wave's column blocks, one after the other
    constexpr int cbk = decltype(cbk_tag)::value;
    f32x16 (&acc)[NT32] = accs_[cbk];
    f32x4 (&acc16)[2] = acc16s_[cbk];
    const int wblk = wid * NCB + cbk;               // this pass's 32-column block inside the tile (wave-uniform)
    const int ncol0 = ncol0_w + cbk * 32;           // ... and its first column
    {   // epilogue scope: every lane-derived index below is rebuilt from an OPAQUE copy of the lane id, so that hipcc cannot
        // compute the epilogue's per-round offsets once, in front of the tile loop, and carry them (24 VGPRs of hoisted
        // store offsets, spilled in the in_proj instantiation) through every k-loop
    int lane_e = lane;
#ifndef MDM_EMU
    asm volatile("" : "+v"(lane_e));
#endif
    const int r = lane_e & 31, h = lane_e >> 5, r16 = lane_e & 15, g16 = lane_e >> 4;
    const int m_end = min(M, m0 + rows_per_tile);
    float* patch = reinterpret_cast<float*>(lds + x3_patch_base(RINGN)) + wid * (X3_PATCH_BYTES / 4);  // [8][32] fp32
    const int prow = lane_e >> 3, pc4 = (lane_e & 7) * 4;
    const int n4 = ncol0 + pc4;                          // first of this lane's 4 columns in the row layout
    // per-lane column vectors of the row-major side: bias (or folded bias), Q scale, folded column sums, residual gamma/beta
    const int cl4 = wblk * 32 + pc4;                      // this lane's first column inside the tile
    const float4 b4 = ld4(cvec + cl4);
    // column scale (in_proj's Q columns; scale_cols is a multiple of the tile width): only the instantiations without an
    // activation and without a plane residual carry one (the launcher refuses it elsewhere) -- two packed multiplies per round
    constexpr bool COL_SCALE = x3_has_col_scale(ACT, RES);
    const float mult4 = (COL_SCALE && n4 < ep.scale_cols) ? ep.col_scale : 1.f;
    float4 c4 = zero4(), g4 = zero4(), be4 = zero4();
    if constexpr (FOLD) c4 = ld4(cvec + 256 + cl4);
    if constexpr (RES == 3) {
      g4 = ld4(cvec + 512 + cl4);
      be4 = ld4(cvec + 768 + cl4);
    }
    // the same in the accumulator layout (lane -> column r of the wave's 32): V^T path
    const float bias = cvec[wblk * 32 + r];
    float csum = 0.f;
    if constexpr (FOLD) csum = cvec[256 + wblk * 32 + r];
    // accumulator values of one round, row-major, -> the GEMM's value:  fold / bias, activation, Q scale
    const float accs = ep.acc_scale;
    // RES == 3: the rebuilt LayerNorm residual's beta is a per-column constant like the bias -- added with it
    const float4 bb4 = RES == 3 ? make_float4(b4.x + be4.x, b4.y + be4.y, b4.z + be4.z, b4.w + be4.w) : b4;
    auto finish4 = [&](float4 v4, float2 st) __attribute__((always_inline)) {
      v4.x *= accs; v4.y *= accs; v4.z *= accs; v4.w *= accs;
      if constexpr (FOLD) {
        v4.x = st.y * (v4.x - st.x * c4.x) + bb4.x; v4.y = st.y * (v4.y - st.x * c4.y) + bb4.y;
        v4.z = st.y * (v4.z - st.x * c4.z) + bb4.z; v4.w = st.y * (v4.w - st.x * c4.w) + bb4.w;
      } else {
        v4.x += bb4.x; v4.y += bb4.y; v4.z += bb4.z; v4.w += bb4.w;
      }
      if (ACT == ACT_GELU) { v4.x = gelu_erf_fast(v4.x); v4.y = gelu_erf_fast(v4.y); v4.z = gelu_erf_fast(v4.z); v4.w = gelu_erf_fast(v4.w); }
      else if (ACT == ACT_SILU) { v4.x = silu(v4.x); v4.y = silu(v4.y); v4.z = silu(v4.z); v4.w = silu(v4.w); }
      if constexpr (COL_SCALE) { v4.x *= mult4; v4.y *= mult4; v4.z *= mult4; v4.w *= mult4; }
      return v4;
    };
    // row statistics of the rows a lane finishes in round j (FOLD: of the A rows, RES == 3: of the residual rows).  Read ONE
    // ROUND AHEAD, next to the patch read: fetched where it is used, each round paid a second, exposed LDS round trip
    // (the wave-level fences keep it behind the next round's patch writes)
    auto row_stats = [&](auto j_tag) __attribute__((always_inline)) {
      constexpr int j = decltype(j_tag)::value;
      if constexpr (LN_TABS && j < NROUNDS) return stab[(j / 4) * 32 + 8 * (j % 4) + prow];
      else return make_float2(0.f, 1.f);
    };
    // 28 rounds (row sub-tile t, register group g): raw accumulators -> patch -> 16-byte row-major read.  Round j+1's
    // patch writes are issued between round j's read and its stores, so the LDS round trip of one round hides under
    // the VALU work of the next (a wave's LDS operations execute in order).
    auto patch_write = [&](auto j_tag) __attribute__((always_inline)) {
      constexpr int j = decltype(j_tag)::value, t = j / 4, g = j % 4;
      if constexpr (j < 4 * NT32) {
#pragma unroll
        for (int e = 0; e < 4; ++e) patch[((e + 4 * h) << 5) + r] = acc[t][4 * g + e];
      } else if constexpr (T16 && j < NROUNDS) {
        // 16-row sub-tile, rows 8g .. 8g+7 = accumulator rows 4*(lane>>4) + e of the lanes with lane>>5 == g
        if (h == g) {
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 4; ++e) patch[((e + 4 * (g16 & 1)) << 5) + 16 * cb + r16] = acc16[cb][e];
        }
      }
    };

    if (OUT_QKV) {
      // in_proj -> attention operand planes (attention_x3.h).  rows_per_tile == S: tile row == token, tile_m == sequence.
      // Pad tokens (S <= token < SP): V^T pads are written with whatever finite values the neighbouring activation rows
      // produce (the attention kernel multiplies them by p == 0 exactly); Q / K pad rows are never read by it and are not
      // stored in the last sub-tile (masking every round cost 112 hoisted lane masks, 280 spilled SGPRs).
      const int Dm = ep.D, SPq = ep.qkv.SP, Hq = ep.qkv.H;
      int ncol_e = ncol0, m0_e = m0;
#ifndef MDM_EMU
      // opaque copies: keeps hipcc from computing the epilogue's 64-bit store addresses BEFORE the k-loop and carrying
      // them (spilled) across it
      asm volatile("" : "+s"(ncol_e), "+s"(m0_e));
#endif
      const int which = ncol_e / Dm, hcol = ncol_e - which * Dm, head = hcol >> 7, d0 = hcol & 127;
      const size_t shq = (size_t)(m0_e / rows_per_tile) * Hq + head;
      if (ncol0 < N && !(ABL & 1)) {
        if (which == 2) {
          // V^T: accumulator registers 8 s2 .. 8 s2 + 7 of a lane ARE positions 8h .. 8h+7 of 16-key group s2
          p16_t* vhp = ep.qkv.vh + ((shq * ep.qkv.NKT) * AX_HD + d0 + r) * 32 + 8 * h;
          p16_t* vlp = ep.qkv.vl + ((shq * ep.qkv.NKT) * AX_HD + d0 + r) * 32 + 8 * h;
          const int nkt = ep.qkv.NKT;
#pragma unroll
          for (int t = 0; t < NT32; ++t) {
            if (t < nkt) {
#pragma unroll
              for (int s2 = 0; s2 < 2; ++s2) {
                float vv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  if constexpr (FOLD) {
                    const float2 st = atab[32 * t + mfma_row(8 * s2 + j, h)];
                    vv[j] = st.y * (acc[t][8 * s2 + j] * accs - st.x * csum) + bias;
                  } else {
                    vv[j] = acc[t][8 * s2 + j] * accs + bias;
                  }
                }
                p16x8 vh8, vl8;
                split8(vv, vh8, vl8);
                *reinterpret_cast<p16x8*>(vhp + t * (AX_HD * 32) + 16 * s2) = vh8;
                *reinterpret_cast<p16x8*>(vlp + t * (AX_HD * 32) + 16 * s2) = vl8;
              }
            }
          }
          if constexpr (T16) {
            // keys 192 + 4*(lane>>4) + e of key tile 6, 16-key group 0: positions 4*((g16>>1) + 2*(g16&1)) + e; the group-1
            // half of the tile (keys 208-223) is never written -- the attention kernel skips it
            if (NT32 < nkt) {
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) {
                const int cl = 16 * cb + r16;                      // this lane's column inside the wave's 32
                const float b16 = lane_bcast(bias, cl);
                const float c16 = FOLD ? lane_bcast(csum, cl) : 0.f;
                float vv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  if constexpr (FOLD) {
                    const float2 st = atab[192 + 4 * g16 + e];
                    vv[e] = st.y * (acc16[cb][e] * accs - st.x * c16) + b16;
                  } else {
                    vv[e] = acc16[cb][e] * accs + b16;
                  }
                }
                const size_t o = ((shq * ep.qkv.NKT + NT32) * AX_HD + d0 + cl) * 32 + 4 * ((g16 >> 1) + 2 * (g16 & 1));
                split4_store(ep.qkv.vh + o, ep.qkv.vl + o, make_float4(vv[0], vv[1], vv[2], vv[3]));
              }
            }
          }
        } else {
          // Q / K rows: one base pointer per plane, 32-bit offsets
          p16_t* dh = (which == 0 ? ep.qkv.qh : ep.qkv.kh) + shq * SPq * AX_HD + d0 + pc4;
          p16_t* dl = (which == 0 ? ep.qkv.ql : ep.qkv.kl) + shq * SPq * AX_HD + d0 + pc4;
          const int nkt = ep.qkv.NKT;
          patch_write(std::integral_constant<int, 0>{});
          float2 st_cur = row_stats(std::integral_constant<int, 0>{});
          float4 v_cur = zero4();
          if constexpr (X3_EPI_AHEAD) {
            wave_lds_fence();
            v_cur = ld4(&patch[prow * 32 + pc4]);
            wave_lds_fence();
            patch_write(std::integral_constant<int, 1>{});
          }
          static_for<NROUNDS>([&](auto j_tag) __attribute__((always_inline)) {
            constexpr int j = decltype(j_tag)::value, t = j / 4, g = j % 4;
            float4 v4;
            float2 st_next;
            if constexpr (X3_EPI_AHEAD) {
              wave_lds_fence();
              float4 v_next = zero4();
              if constexpr (j + 1 < NROUNDS) v_next = ld4(&patch[prow * 32 + pc4]);
              st_next = row_stats(std::integral_constant<int, j + 1>{});
              wave_lds_fence();
              patch_write(std::integral_constant<int, j + 2>{});
              v4 = v_cur;
              v_cur = v_next;
            } else {
              wave_lds_fence();
              v4 = ld4(&patch[prow * 32 + pc4]);
              st_next = row_stats(std::integral_constant<int, j + 1>{});
              wave_lds_fence();
              patch_write(std::integral_constant<int, j + 1>{});
            }
            const float2 st = st_cur;
            st_cur = st_next;
            if (t < nkt) {
              const int tok = 32 * t + 8 * g + prow;
              v4 = finish4(v4, st);
              // tokens past the sequence (only the last sub-tile can hold any when the tile is one sequence of more than
              // 192 tokens) are not stored: the attention kernel never reads Q / K pad rows
              if (t < X3_MSUB - 1 || tok < ep.S) split4_store(dh + tok * AX_HD, dl + tok * AX_HD, v4);
            }
          });
        }
      }
    } else {
      // residual tile: streamed two row sub-tiles ahead of its use through untracked loads (common.h gload16_async);
      // rows past the matrix are clamped (loaded, never stored)
      constexpr bool HAS_RES = RES != 0;
      constexpr bool RES_PLANES = RES == 2 || RES == 3;
      // plane rows of this tile (output and residual) are addressed through clipped windows (common.h ClipWin): byte offset
      // of a lane's 4 columns in tile row `prow`, plus a wave-uniform row term per round
      static_assert(!(EMBED && RES_PLANES), "the EMBED epilogue takes an fp32 residual");
      constexpr bool CLIP_OUT = OUT_PLANES && !EMBED;
      const uint32_t pitch2 = (uint32_t)ep.ld * 2u;
      const uint32_t voff0 = n4 < N ? ((uint32_t)prow * (uint32_t)ep.ld + (uint32_t)n4) * 2u : CLIP_OFF;
      ClipWin w_oh = {}, w_ol = {}, w_rh = {}, w_rl = {};
      {
        int m0_e = m0;
#ifndef MDM_EMU
        asm volatile("" : "+s"(m0_e));   // (opaque: keeps the windows from being built, and carried, in front of the k-loop)
#endif
        const size_t row0 = (size_t)m0_e * ep.ld;
        if constexpr (CLIP_OUT) {
          const uint32_t bytes = (uint32_t)(m_end - m0_e) * pitch2;
          w_oh = clip_win(ep.oh + row0, bytes);
          w_ol = clip_win(ep.ol + row0, bytes);
        }
        if constexpr (RES_PLANES) {
          const uint32_t bytes = (uint32_t)min(M - m0_e, X3_TM) * pitch2;
          w_rh = clip_win(ep.resh + row0, bytes);
          w_rl = clip_win(ep.resl + row0, bytes);
        }
      }
      constexpr int RR = (RES == 3) ? 2 : 3;   // residual sub-tiles in flight + in use (RES == 3 sits at the VGPR limit)
      f32x4 rr[RR][4];       // RES == 1
      u32x2 rh[RR][4], rl[RR][4];  // RES == 2 / 3
      auto res_issue = [&](auto t_tag) __attribute__((always_inline)) {
        constexpr int t = decltype(t_tag)::value;
        if constexpr (HAS_RES && t < X3_MSUB) {
#pragma unroll
          for (int g = 0; g < x3_res_rounds(t, T16); ++g) {
            if constexpr (RES == 1) {
              const int m = min(m0 + t * 32 + 8 * g + prow, M - 1);
              const size_t o = (size_t)(EMBED ? 1 + m % ep.emb_T : m) * ep.ld + (n4 < N ? n4 : 0);   // EMBED: positional row
              gload16_async(rr[t % RR][g], ep.res + o);
            } else {
              const uint32_t o = voff0 + (uint32_t)(t * 32 + 8 * g) * pitch2;   // past the matrix: reads 0, never stored
              clip_load8_async(rh[t % RR][g], w_rh, o);
              clip_load8_async(rl[t % RR][g], w_rl, o);
            }
          }
        }
      };
      auto res_wait = [&](auto t_tag) __attribute__((always_inline)) {
        constexpr int t = decltype(t_tag)::value;
        // loads of the younger sub-tiles already requested (the only ones allowed to stay in flight)
        constexpr int ahead = x3_res_younger_rounds(t, RR, T16);
        if constexpr (RES == 1) {
          vmem_wait<ahead>(rr[t % RR][0], rr[t % RR][1], rr[t % RR][2], rr[t % RR][3]);
        } else if constexpr (RES_PLANES) {
          vmem_wait<2 * ahead>(rh[t % RR][0], rh[t % RR][1], rh[t % RR][2], rh[t % RR][3], rl[t % RR][0], rl[t % RR][1],
                               rl[t % RR][2], rl[t % RR][3]);
        }
      };
      if (!(ABL & 1)) {
        res_issue(std::integral_constant<int, 0>{});
        if constexpr (RR == 3) res_issue(std::integral_constant<int, 1>{});
      }
      float2* part = reinterpret_cast<float2*>(lds + x3_part_base(NBLK, RINGN)) + wblk * X3_TM;   // OSTAT: this column block's partials
      patch_write(std::integral_constant<int, 0>{});
      float2 st_cur = row_stats(std::integral_constant<int, 0>{});
      float4 v_cur = zero4();
      if constexpr (X3_EPI_AHEAD) {
        wave_lds_fence();
        v_cur = ld4(&patch[prow * 32 + pc4]);
        wave_lds_fence();
        patch_write(std::integral_constant<int, 1>{});
      }
      static_for<NROUNDS>([&](auto j_tag) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value, t = j / 4, g = j % 4;
        if constexpr (HAS_RES && g == 0) {
          if (!(ABL & 1)) {
            res_issue(std::integral_constant<int, t + RR - 1>{});
            res_wait(std::integral_constant<int, t>{});
          }
        }
        float4 v4;
        float2 st_next;
        if constexpr (X3_EPI_AHEAD) {
          wave_lds_fence();
          float4 v_next = zero4();
          if constexpr (j + 1 < NROUNDS) v_next = ld4(&patch[prow * 32 + pc4]);
          st_next = row_stats(std::integral_constant<int, j + 1>{});
          wave_lds_fence();
          patch_write(std::integral_constant<int, j + 2>{});
          v4 = v_cur;
          v_cur = v_next;
        } else {
          wave_lds_fence();
          v4 = ld4(&patch[prow * 32 + pc4]);
          st_next = row_stats(std::integral_constant<int, j + 1>{});
          wave_lds_fence();
          patch_write(std::integral_constant<int, j + 1>{});
        }
        const float2 st = st_cur;
        st_cur = st_next;
        const int row_in_tile = t * 32 + 8 * g + prow;
        v4 = finish4(v4, st);
        if (!(ABL & 1)) {
          if constexpr (RES == 1) {
            const f32x4 q4 = rr[t % RR][g];
            v4.x += q4[0]; v4.y += q4[1]; v4.z += q4[2]; v4.w += q4[3];
          } else if constexpr (RES_PLANES && kSplitF16) {
            // x = hi + lo (RES == 3: minus the row mean, times rstd * gamma; beta rides in the bias vector bb4): the planes are
            // converted inside the adds (common.h f16_half_plus), the mean leaves before anything is scaled
            const u32x2 a = rh[t % RR][g], b = rl[t % RR][g];
            const float c0 = RES == 3 ? -st.x : 0.f;
            const float d0 = f16_half_plus<0>(b[0], f16_half_plus<0>(a[0], c0));
            const float d1 = f16_half_plus<1>(b[0], f16_half_plus<1>(a[0], c0));
            const float d2 = f16_half_plus<0>(b[1], f16_half_plus<0>(a[1], c0));
            const float d3 = f16_half_plus<1>(b[1], f16_half_plus<1>(a[1], c0));
            if constexpr (RES == 3) {
              v4.x = fmaf(d0, st.y * g4.x, v4.x); v4.y = fmaf(d1, st.y * g4.y, v4.y);
              v4.z = fmaf(d2, st.y * g4.z, v4.z); v4.w = fmaf(d3, st.y * g4.w, v4.w);
            } else {
              v4.x += d0; v4.y += d1; v4.z += d2; v4.w += d3;
            }
          } else if constexpr (RES_PLANES) {
            const u32x2 a = rh[t % RR][g], b = rl[t % RR][g];
            float4 x4 = make_float4(
                p16_to_f32((p16_t)(a[0] & 0xffffu)) + p16_to_f32((p16_t)(b[0] & 0xffffu)),
                p16_to_f32((p16_t)(a[0] >> 16)) + p16_to_f32((p16_t)(b[0] >> 16)),
                p16_to_f32((p16_t)(a[1] & 0xffffu)) + p16_to_f32((p16_t)(b[1] & 0xffffu)),
                p16_to_f32((p16_t)(a[1] >> 16)) + p16_to_f32((p16_t)(b[1] >> 16)));
            if constexpr (RES == 3) {   // the residual is LayerNorm(x), rebuilt from x's planes and its row statistics
              x4.x = (x4.x - st.x) * st.y * g4.x; x4.y = (x4.y - st.x) * st.y * g4.y;   // (+ beta: in bb4)
              x4.z = (x4.z - st.x) * st.y * g4.z; x4.w = (x4.w - st.x) * st.y * g4.w;
            }
            v4.x += x4.x; v4.y += x4.y; v4.z += x4.z; v4.w += x4.w;
          }
        }
        if constexpr (OSTAT) {   // partial (sum, centred sum of squares) of this row over the wave's 32 columns
          const float s1 = sum_lanes8((v4.x + v4.y) + (v4.z + v4.w));
          const float mw = s1 * (1.0f / 32.0f);
          const float dx = v4.x - mw, dy = v4.y - mw, dz = v4.z - mw, dw = v4.w - mw;
          const float m2 = sum_lanes8((dx * dx + dy * dy) + (dz * dz + dw * dw));
          if ((lane_e & 7) == 0) part[row_in_tile] = make_float2(s1, m2);
        }
        if (!(ABL & 1)) {
          if constexpr (CLIP_OUT) split4_store_clip(w_oh, w_ol, voff0 + (uint32_t)(t * 32 + 8 * g) * pitch2, v4);
          if constexpr (EMBED || OUT_F32) {
            const int m = m0 + row_in_tile;
            if (m < m_end && n4 < N) {  // N % 4 == 0
              if constexpr (EMBED) {
                const int bb = m / ep.emb_T, tt = m - bb * ep.emb_T;
                for (int br = 0; br < ep.emb_nbranch; ++br) {
                  const size_t o = ((size_t)(br * ep.emb_B + bb) * (ep.emb_T + 1) + 1 + tt) * ep.ld + n4;
                  split4_store(ep.oh + o, ep.ol + o, v4);
                }
              } else {
                st4(ep.out + (size_t)m * ep.ld + n4, v4);
              }
            }
          }
        }
      });
    }
    }   // epilogue scope
    });
    if constexpr (OSTAT && !OUT_QKV) {   // rows x column blocks partials -> one (sum, M2) pair per row and column tile
        const int m_end = min(M, m0 + rows_per_tile);
        wg_barrier();
        if (tid < X3_TM && m0 + tid < m_end) {
          const float2* pp = reinterpret_cast<const float2*>(lds + x3_part_base(NBLK, RINGN));
          float s1 = 0.f;
#pragma unroll
          for (int w8 = 0; w8 < NBLK; ++w8) s1 += pp[w8 * X3_TM + tid].x;
          const float mt = s1 * (1.0f / X3_TN);     // OSTAT launches have N % X3_TN == 0: every block contributes 32 columns
          float m2 = 0.f;
#pragma unroll
          for (int w8 = 0; w8 < NBLK; ++w8) {
            const float2 v = pp[w8 * X3_TM + tid];
            const float dm = v.x * (1.0f / 32.0f) - mt;
            m2 += v.y + 32.0f * dm * dm;
          }
          *reinterpret_cast<float2*>(ep.ostat + ((size_t)(m0 + tid) * tiles_n + n0 / X3_TN) * 2) = make_float2(s1, m2);
        }
      }
#ifdef MDM_X3_DBG
    if constexpr ((ABL & 128) != 0) {
      const unsigned long long t2 = x3_now();
      if (tid == 0) {
        atomicAdd(&g_x3_dbg[0], dbg_w0); atomicAdd(&g_x3_dbg[1], dbg_w1); atomicAdd(&g_x3_dbg[2], dbg_t1 - dbg_t0);
        atomicAdd(&g_x3_dbg[3], t2 - dbg_t1); atomicAdd(&g_x3_dbg[4], 1ULL); atomicAdd(&g_x3_dbg[5], (unsigned long long)nk); atomicAdd(&g_x3_dbg[6], dbg_bar);
      }
    }
#endif
  }
  wait_vmem_all();  // the stream's last (unused) LDS-DMA stage must land before this workgroup's LDS is released
}

#ifndef MDM_X3_KERNEL_ONLY   // (register-pressure studies compile single instantiations of the kernel without the launchers)
// rows per block tile: a whole number of sequences when the row space is sequence-structured (keeps the tile count a
// multiple of the sequence count -> no ragged last wave of workgroups), else the full 224.
inline int x3_rows_per_tile(int M, int seq_len) {
  if (seq_len > 0 && seq_len <= X3_TM && M % seq_len == 0) return (X3_TM / seq_len) * seq_len;
  return X3_TM;
}

// persistent grid: `per_cu` workgroups per CU (4-wave: two -- 60 KB of LDS and <= 256 VGPRs each; 8-wave: one)
inline int x3_grid_limit(int per_cu) {
#ifdef MDM_EMU
  (void)per_cu;
  return 3;  // small, so that the emulator exercises the tile roll-over paths
#else
  static int cus_of[kMaxDevices] = {};   // per device ordinal: a process may drive several GPUs
  const int dev = rt_device_ordinal();
  int& cus = cus_of[dev];
  if (cus == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  int wgs = per_cu * cus / 8 * 8;  // xcd_remap keeps a workgroup on one XCD only if the stride is a multiple of 8
#ifdef MDM_PROBES   // lab/probes/two_chains.py: two half-batch chains share the chip, each launch takes 1 / div of the CUs
  static const int div = [] { const char* e = getenv("MDM_X3_GRID_DIV"); return e != nullptr && atoi(e) > 1 ? atoi(e) : 1; }();
  wgs = wgs / div / 8 * 8;
#endif
  return wgs > 0 ? wgs : 8;
#endif
}

// workgroup shape used by the launchers below: 8 waves (default: whole-bench A/B on one box 307 vs 302 motions/s) or 4
// (mdm_debug_set(2, waves) / MDM_X3_WAVES for A/B probes)
#ifdef MDM_PROBES
inline int& x3_waves_setting() {
  static int waves = [] {
    const char* e = getenv("MDM_X3_WAVES");   // A/B runs of whole benchmarks
    return (e != nullptr && e[0] == '4') ? 4 : 8;
  }();
  return waves;
}
#else
inline int x3_waves_setting() { return 8; }   // the 4-wave form is compiled into the probe library only
#endif

template <int WAVES, int ACT, int RES, bool OUT_F32, bool OUT_PLANES, bool OUT_QKV, int ABL, bool FOLD = false,
          bool OSTAT = false, bool EMBED = false, bool T16 = false, bool F6 = false, bool PIPE = false, int NCB = 1>
inline int launch_gemm_x3_w(const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N, int K,
                                int rpt, hipStream_t stream) {
  constexpr int NBLK = WAVES * NCB;          // 32-column blocks per tile: the LDS layout is per block
  constexpr int TN = 32 * NBLK;
  constexpr bool LN = FOLD || OSTAT || RES == 3;
  constexpr int RINGN = PIPE ? X3_PIPE_RING : X3_A_RING;
  const int tiles_m = (M + rpt - 1) / rpt, tiles_n = (N + TN - 1) / TN;
  const int total = tiles_m * tiles_n;
  static_assert(!(F6 && T16), "the f16f6 k-loop has no 16-row sub-tile yet");
  auto kfn = &gemm_x3_kernel<WAVES, ACT, RES, OUT_F32, OUT_PLANES, OUT_QKV, ABL, FOLD, OSTAT, EMBED, T16, F6, PIPE, NCB>;
  if (T16 && rpt > X3_TM - 16) return -2;
  if (PIPE && (K / X3_BK) % 2 != 0) return -2;   // the pipelined k-loop is unrolled over step pairs
  if (!x3_has_col_scale(ACT, RES) && ep.scale_cols > 0) return -2;   // (this instantiation compiles the column scale out)
#ifndef MDM_EMU
  if (x3_lds_bytes(NBLK, LN, RINGN) > 65536) {
    static bool configured[kMaxDevices] = {};  // per instantiation and device (the attribute belongs to the device's code object)
    if (const int rc = rt_dyn_lds_once(kfn, x3_lds_bytes(NBLK, LN, RINGN), configured, stream)) return rc;
  }
#endif
  const int grid = std::min(total, x3_grid_limit((WAVES == 4 && NCB == 1) ? 2 : 1));
  MDM_LAUNCH(kfn, dim3(grid), dim3(64 * WAVES), x3_lds_bytes(NBLK, LN, RINGN), stream, A, W, ep, M, N, K, rpt, tiles_n, total);
  return 0;
}

// MDM_X3_WIDE=1 (probe library only; read per launch) selects the four-wave, 64-columns-per-wave form of the pipelined loop --
// built, parity-green on the MI355X and 12 % slower over the whole loop (323 vs 367 motions/s): profiles/r04d_wide.md
inline bool x3_wide_setting() {
#ifdef MDM_PROBES
  const char* e = getenv("MDM_X3_WIDE");
  return e != nullptr && e[0] == '1';
#else
  return false;
#endif
}

// The pipelined k-loop (PIPE) is the default wherever it exists (208-row tiles, an even number of 32-deep k steps);
// MDM_X3_PIPE=0 selects the step-synchronous loop for same-box A/B runs.
inline bool x3_pipe_setting(int kind = 0) {
#ifndef MDM_PROBES
  (void)kind;
  return true;      // (the product library reads no environment variable)
#else
  static const bool on = [] {
    const char* e = getenv("MDM_X3_PIPE");
    return !(e != nullptr && e[0] == '0');
  }();
  // MDM_X3_PIPE_KINDS: bit k = GEMM kind k of launch_gemm_x3_ln (0 in_proj, 1 out_proj layer 0, 2 out_proj / linear2,
  // 3 linear1, 4 OutputProcess), bit 5 = layer 0's in_proj; default: all -- bisection of a misbehaving instantiation
  static const int kinds = [] {
    const char* e = getenv("MDM_X3_PIPE_KINDS");
    return e != nullptr ? atoi(e) : 0x3f;
  }();
  return on && ((kinds >> kind) & 1);
#endif
}

// The GEMMs of the folded-LayerNorm encoder (8-wave workgroups only):
//   kind 0  in_proj, A = pre-norm sum           FOLD -> attention operand planes
//   kind 1  out_proj of layer 0                 residual = plain planes, writes planes + row statistics
//   kind 2  out_proj (l >= 1) / linear2         residual = LayerNorm rebuilt from planes, writes planes + row statistics
//   kind 3  linear1                             FOLD + GELU -> planes
//   kind 4  OutputProcess                       FOLD -> fp32
//   kind 5  InputProcess                        + positional rows, planes to the token rows of every branch (EMBED)
// WIDE (round 4): the pipelined loop as four waves x 64 columns, one wave per SIMD (kernel header, NCB = 2)
template <bool T16, bool PIPE = false, bool WIDE = false>
inline int launch_gemm_x3_ln_t(int kind, const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N,
                                   int K, int rpt, hipStream_t s) {
#ifdef MDM_PROBES   // (measured 12-18 % slower per launch, profiles/r04d_wide.md: instantiated in the probe library only)
  if constexpr (WIDE) {
    switch (kind) {
      case 0: return launch_gemm_x3_w<4, ACT_NONE, 0, false, false, true, 0, true, false, false, T16, false, PIPE, 2>(A, W, ep, M, N, K, rpt, s);
      case 1: return launch_gemm_x3_w<4, ACT_NONE, 2, false, true, false, 0, false, true, false, T16, false, PIPE, 2>(A, W, ep, M, N, K, rpt, s);
      case 2: return launch_gemm_x3_w<4, ACT_NONE, 3, false, true, false, 0, false, true, false, T16, false, PIPE, 2>(A, W, ep, M, N, K, rpt, s);
      case 3: return launch_gemm_x3_w<4, ACT_GELU, 0, false, true, false, 0, true, false, false, T16, false, PIPE, 2>(A, W, ep, M, N, K, rpt, s);
      case 4: return launch_gemm_x3_w<4, ACT_NONE, 0, true, false, false, 0, true, false, false, T16, false, PIPE, 2>(A, W, ep, M, N, K, rpt, s);
      default: return -2;
    }
  }
#endif
  switch (kind) {
    case 0: return launch_gemm_x3_w<8, ACT_NONE, 0, false, false, true, 0, true, false, false, T16, false, PIPE>(A, W, ep, M, N, K, rpt, s);
    case 1: return launch_gemm_x3_w<8, ACT_NONE, 2, false, true, false, 0, false, true, false, T16, false, PIPE>(A, W, ep, M, N, K, rpt, s);
    case 2: return launch_gemm_x3_w<8, ACT_NONE, 3, false, true, false, 0, false, true, false, T16, false, PIPE>(A, W, ep, M, N, K, rpt, s);
    case 3: return launch_gemm_x3_w<8, ACT_GELU, 0, false, true, false, 0, true, false, false, T16, false, PIPE>(A, W, ep, M, N, K, rpt, s);
    case 4: return launch_gemm_x3_w<8, ACT_NONE, 0, true, false, false, 0, true, false, false, T16, false, PIPE>(A, W, ep, M, N, K, rpt, s);
    case 5:   // InputProcess: K = 288 is nine steps -- stays on the step-synchronous loop
      return launch_gemm_x3_w<8, ACT_NONE, 1, false, true, false, 0, false, false, true, T16>(A, W, ep, M, N, K, rpt, s);
    default: return -2;
  }
}
// 208-row tiles (T16) whenever the row extent of a tile fits (S = 197 does); probe library: MDM_X3_T16=0 keeps 224-row tiles
inline bool x3_t16_setting() {
#ifdef MDM_PROBES
  static const bool on = [] {
    const char* e = getenv("MDM_X3_T16");
    return !(e != nullptr && e[0] == '0');
  }();
  return on;
#else
  return true;
#endif
}
inline int launch_gemm_x3_ln(int kind, const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N,
                                 int K, int rpt, hipStream_t s) {
  if (rpt <= X3_TM - 16 && x3_t16_setting()) {
    if (kind != 5 && (K / X3_BK) % 2 == 0 && x3_pipe_setting(kind)) {
      if (x3_wide_setting()) return launch_gemm_x3_ln_t<true, true, true>(kind, A, W, ep, M, N, K, rpt, s);
      return launch_gemm_x3_ln_t<true, true>(kind, A, W, ep, M, N, K, rpt, s);
    }
    return launch_gemm_x3_ln_t<true>(kind, A, W, ep, M, N, K, rpt, s);
  }
  return launch_gemm_x3_ln_t<false>(kind, A, W, ep, M, N, K, rpt, s);
}

#ifdef MDM_PROBES
inline int& x3_pipe_probe() { static int on = 0; return on; }
#endif

template <int ACT, int RES, bool OUT_F32, bool OUT_PLANES, bool OUT_QKV, int ABL = 0>
inline int launch_gemm_x3_t(const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N, int K,
                                int rpt, hipStream_t stream) {
#ifdef MDM_PROBES
  if (x3_waves_setting() == 4)
    return launch_gemm_x3_w<4, ACT, RES, OUT_F32, OUT_PLANES, OUT_QKV, ABL>(A, W, ep, M, N, K, rpt, stream);
#endif
  return launch_gemm_x3_w<8, ACT, RES, OUT_F32, OUT_PLANES, OUT_QKV, ABL>(A, W, ep, M, N, K, rpt, stream);
}

// runtime (act, res, outputs) -> one of the instantiations the encoder needs
inline int launch_gemm_x3(const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N, int K, int act,
                              int seq_len, hipStream_t s, int ablate = 0) {
  const bool res = ep.res != nullptr, f32 = ep.out != nullptr, pl = ep.oh != nullptr;
  const int rpt = x3_rows_per_tile(M, seq_len);
  if (ep.resh != nullptr) {  // residual stream held as planes (the model's f16x3 mode)
    if (ablate == 0 && act == ACT_NONE && !res && f32 && !pl)
      return launch_gemm_x3_t<ACT_NONE, 2, true, false, false>(A, W, ep, M, N, K, rpt, s);
    return -2;
  }
#ifdef MDM_PROBES
  // mdm_debug_set(6, p): the plain fp32-out variant on 208-row (T16) tiles of 197-token sequences -- p = 1: the PIPELINED
  // k-loop, p = 2: the step-synchronous one -- with the ablation codes both support (1 no epilogue stores, 2 no loads after
  // the prologue, 4 no MFMAs, and their sums): tools/gemm_probe_pipe.py
  if (x3_pipe_probe() != 0 && act == ACT_NONE && !res && f32 && !pl && M % 197 == 0 && (K / X3_BK) % 2 == 0) {
#define X3_PIPE_PROBE_CASE(c) case c: return x3_pipe_probe() == 1 \
      ? launch_gemm_x3_w<8, ACT_NONE, 0, true, false, false, c, false, false, false, true, false, true>(A, W, ep, M, N, K, 197, s) \
      : launch_gemm_x3_w<8, ACT_NONE, 0, true, false, false, c, false, false, false, true, false, false>(A, W, ep, M, N, K, 197, s);
    switch (ablate) {
      X3_PIPE_PROBE_CASE(0) X3_PIPE_PROBE_CASE(1) X3_PIPE_PROBE_CASE(2) X3_PIPE_PROBE_CASE(3) X3_PIPE_PROBE_CASE(4)
      X3_PIPE_PROBE_CASE(5) X3_PIPE_PROBE_CASE(6) X3_PIPE_PROBE_CASE(7)
      default: return -2;
    }
#undef X3_PIPE_PROBE_CASE
  }
  if (ablate != 0) {  // profiling experiments (mdm_debug_set): only the plain fp32-out variant is instantiated
    if (!(act == ACT_NONE && !res && f32 && !pl)) return -2;
    switch (ablate) {
      case 1: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 1>(A, W, ep, M, N, K, rpt, s);
      case 2: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 2>(A, W, ep, M, N, K, rpt, s);
      case 3: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 3>(A, W, ep, M, N, K, rpt, s);
      case 4: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 4>(A, W, ep, M, N, K, rpt, s);
      case 8: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 8>(A, W, ep, M, N, K, rpt, s);
      case 16: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 16>(A, W, ep, M, N, K, rpt, s);
      case 32: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 32>(A, W, ep, M, N, K, rpt, s);
      case 64: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 64>(A, W, ep, M, N, K, rpt, s);
      case 128: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 128>(A, W, ep, M, N, K, rpt, s);
      case 256: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 256>(A, W, ep, M, N, K, rpt, s);
      // decomposition of the no-MFMA floor: 5 = 4|1 (also no stores), 6 = 4|2 (also no loads), 7 = 4|2|1, 68 = 4|64 (half the reads)
      case 5: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 5>(A, W, ep, M, N, K, rpt, s);
      case 6: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 6>(A, W, ep, M, N, K, rpt, s);
      case 7: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 7>(A, W, ep, M, N, K, rpt, s);
      case 68: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 68>(A, W, ep, M, N, K, rpt, s);
      case 9: return launch_gemm_x3_t<ACT_NONE, 0, true, false, false, 9>(A, W, ep, M, N, K, rpt, s);
      default: return -2;
    }
  }
#else
  if (ablate != 0) return -2;
#endif
  if (act == ACT_NONE && !res && f32 && !pl) return launch_gemm_x3_t<ACT_NONE, 0, true, false, false>(A, W, ep, M, N, K, rpt, s);
  if (act == ACT_NONE && res && f32 && !pl) return launch_gemm_x3_t<ACT_NONE, 1, true, false, false>(A, W, ep, M, N, K, rpt, s);
  if (act == ACT_GELU && !res && !f32 && pl) return launch_gemm_x3_t<ACT_GELU, 0, false, true, false>(A, W, ep, M, N, K, rpt, s);
  if (act == ACT_GELU && !res && f32 && !pl) return launch_gemm_x3_t<ACT_GELU, 0, true, false, false>(A, W, ep, M, N, K, rpt, s);
  if (act == ACT_GELU && res && f32 && !pl) return launch_gemm_x3_t<ACT_GELU, 1, true, false, false>(A, W, ep, M, N, K, rpt, s);
  if (act == ACT_SILU && !res && f32 && !pl) return launch_gemm_x3_t<ACT_SILU, 0, true, false, false>(A, W, ep, M, N, K, rpt, s);
  return -2;
}

#ifdef MDM_PROBES
// The f16f6 building block (mdm_linear_f16f6): plain fp32-out epilogues on the F6 k-loop, 224-row tiles
inline int launch_gemm_f16f6(const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int M, int N, int K, int act,
                             hipStream_t s) {
  const bool res = ep.res != nullptr;
  if (act == ACT_NONE && !res && x3_waves_setting() == 4)   // A/B probes: two independent 4-wave workgroups per CU
    return launch_gemm_x3_w<4, ACT_NONE, 0, true, false, false, 0, false, false, false, false, true>(A, W, ep, M, N, K, X3_TM, s);
  if (act == ACT_NONE && !res)
    return launch_gemm_x3_w<8, ACT_NONE, 0, true, false, false, 0, false, false, false, false, true>(A, W, ep, M, N, K, X3_TM, s);
  if (act == ACT_NONE && res)
    return launch_gemm_x3_w<8, ACT_NONE, 1, true, false, false, 0, false, false, false, false, true>(A, W, ep, M, N, K, X3_TM, s);
  if (act == ACT_GELU && !res)
    return launch_gemm_x3_w<8, ACT_GELU, 0, true, false, false, 0, false, false, false, false, true>(A, W, ep, M, N, K, X3_TM, s);
  return -2;
}
#endif

// in_proj: tokens [nseq*S][D] x W [3D][D] -> the attention operand planes; one sequence per tile (tile row == token)
inline int launch_gemm_x3_qkv(const X3Operand& A, const X3Weights& W, const X3Epilogue& ep, int nseq, int S, int D,
                                  hipStream_t s) {
  if (S > X3_TM) return -2;
  if (x3_waves_setting() == 8 && S <= X3_TM - 16 && x3_t16_setting()) {
    if ((D / X3_BK) % 2 == 0 && x3_pipe_setting(5)) {
#ifdef MDM_PROBES
      if (x3_wide_setting())
        return launch_gemm_x3_w<4, ACT_NONE, 0, false, false, true, 0, false, false, false, true, false, true, 2>(A, W, ep, nseq * S,
                                                                                                                   3 * D, D, S, s);
#endif
      return launch_gemm_x3_w<8, ACT_NONE, 0, false, false, true, 0, false, false, false, true, false, true>(A, W, ep, nseq * S,
                                                                                                              3 * D, D, S, s);
    }
    return launch_gemm_x3_w<8, ACT_NONE, 0, false, false, true, 0, false, false, false, true>(A, W, ep, nseq * S, 3 * D, D,
                                                                                                 S, s);
  }
  return launch_gemm_x3_t<ACT_NONE, 0, false, false, true>(A, W, ep, nseq * S, 3 * D, D, S, s);
}

#endif  // MDM_X3_KERNEL_ONLY

}  // namespace mdm

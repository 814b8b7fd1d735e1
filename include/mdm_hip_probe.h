/* mdm_hip_probe.h -- the EXPERIMENT surface of the library: NOT part of the production ABI.
 *
 * `python __graft_entry__.py build` compiles csrc/mdm_api.hip twice: libmdm_hip.so (the product: include/mdm_hip.h only, no
 * ablation instantiations, no process-global switches, never allocates) and libmdm_hip_probe.so (-DMDM_PROBES: the same
 * sources plus what this header declares).  the scripts under tools/ and the probe-only GPU tests bind the probe library explicitly; the
 * Python seams (motion-diffusion-model_amd/) never load it. */
#ifndef MDM_HIP_PROBE_H
#define MDM_HIP_PROBE_H

#include "mdm_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Process-global switches; value 0 = production behaviour.  what = 0: split-precision GEMM ablation code (gemm_x3.h ABL);
 * what = 1: mdm_linear_x3 reuses the operand planes already in scratch (kernel-only timing); what = 2: waves per GEMM
 * workgroup, 8 (default: 208/224 x 256 tiles, one workgroup per CU) or 4 (224 x 128 tiles, two per CU); what = 3: attention
 * ablation code; what = 4: mdm_linear_f16f6 on its reference kernel; what = 5: the `f32` mode's encoder GEMMs run unfused on
 * the f16f6 kernel, operands packed per call into a scratch this library allocates itself; what = 6: mdm_linear_x3's plain
 * fp32-out variant runs on the pipelined k-loop (gemm_x3.h PIPE) with ablation codes 0..7. */
int mdm_debug_set(int what, int value);
/* Cycle counters of the split-precision GEMM's ABL = 128 build (idx 0..7; idx < 0 resets). */
int mdm_debug_get(int idx, double* out);

/* mdm_linear's contract (res may be null) on the EXPERIMENTAL "f16f6" GEMM (csrc/gemm_f16f6.h: one fp16 MFMA pass + two cross
 * terms on block-scaled MX-FP6 operands, K % 32 == 0), through the production GEMM skeleton with the f16f6 k-loop (N % 4 == 0;
 * act none with or without res, gelu without) or a one-wave-per-tile reference kernel (otherwise, or after
 * mdm_debug_set(4, 1)).  Not used by the model: on trained-like weights its trajectory error is 3x the round-1 bf16 split's and
 * 30x the fp16 split's (tools/precision_probe.py --hostile).  `scratch_dev` receives both operands' planes. */
size_t mdm_linear_f16f6_scratch_bytes(int32_t M, int32_t N, int32_t K);
int mdm_linear_f16f6(const float* in_dev, const float* w_dev, const float* bias_dev, const float* res_dev,
                     float* out_dev, int32_t M, int32_t N, int32_t K, int32_t act, void* scratch_dev,
                     size_t scratch_bytes, void* stream);

/* in_proj alone (layer 0's instantiation: no folded LayerNorm): fp32 tokens [nseq * S][D], weights [3D][D] -> the six attention
 * operand planes (nseq * 32 * ceil(S / 32) * D 16-bit elements each: qh ql kh kl vh vl) in `planes_dev`; `scratch_dev`:
 * 4 * nseq * S * D + 12 * D * D bytes.  Determinism screens of the GEMM k-loops (tools/in_proj_determinism.py). */
int mdm_probe_in_proj(const float* tokens, const float* w, const float* bias, void* planes_dev, int32_t nseq, int32_t S,
                      int32_t D, void* scratch_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDM_HIP_PROBE_H */

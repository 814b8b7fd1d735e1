// EXPERIMENT build of the whole library with gemm_x3s8.h's 8-wave split-K tiles behind launch_gemm_x3s (tools/x3s8/build.sh).
// csrc/ is compiled as it is: gemm_x3s.h is included first (its own definitions keep their names), then every USE of
// launch_gemm_x3s in mdm_api.hip is redirected to the dispatcher of gemm_x3s8.h.
#include "gemm_x3s8.h"
#define launch_gemm_x3s launch_gemm_x3s_or_x3s8
#include "mdm_api.hip"

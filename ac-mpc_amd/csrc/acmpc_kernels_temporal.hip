// Translation unit of the mode-T step-major rollout with plain float32 arithmetic: the same kernel template as
// acmpc_kernels.hip, instantiated here so that it can be compiled with -fno-slp-vectorize (see
// launch_rollout_temporal_plain for the measurement behind that).
#define ACMPC_TEMPORAL_TU 1
#include "acmpc_kernels.hip"

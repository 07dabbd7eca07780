/*
 * cattus_hip_diag.h -- diagnostic entry point of libcattus_hip.so.  NOT part of the drop-in boundary (cattus_hip.h): a Cattus
 * host never calls it.  It exists so that the equality tests and the A/B timing scripts can force a code path that
 * cattus_hip_create would not choose, without the library reading switches from the process environment.
 */
#ifndef CATTUS_HIP_DIAG_H
#define CATTUS_HIP_DIAG_H

#include "cattus_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* cattus_hip_create with a list of switches "KEY=VALUE;KEY=VALUE" (NULL or "" = none = cattus_hip_create).  Unknown keys are
 * refused (CATTUS_E_INVALID).  Every switch selects among kernels whose per-leaf results the tests assert to be bit-identical,
 * or a memory plan; none changes what a leaf evaluates to:
 *   CATTUS_TOWER64=0        networks of <= 64 filters: per-layer launches instead of the resident tower
 *   CATTUS_T64_CH=2|4, CATTUS_T64_LS=0, CATTUS_T64S_SHAPE=1|2|9, CATTUS_T64S_HEADS=0   workgroup shapes of the resident towers
 *   CATTUS_SPLIT_W=0        f16x2 direct form: weights through the LDS ring (conv3x3_split_kernel) instead of the register ring
 *   CATTUS_CONV_CB=1|2, CATTUS_CONV_PBW=1|2   tile shapes of the per-layer conv kernels
 *   CATTUS_FUSED_STEM=0     plane pack as its own launch in front of the stem
 *   CATTUS_FORCE_GENERIC=1  the one-thread-per-output f32 path (a second checker of the MFMA kernels)
 *   CATTUS_WINO_KERNEL=k16|k4|k8   Winograd form: the 16-frequencies-per-wave kernel (conv3x3_wino_kernel), the 4-frequencies x
 *                               2x2-blocks one (conv3x3_wino4_kernel / tower_wino4_kernel) or the eight-wave one (conv3x3_wino8_kernel); same bits
 *   CATTUS_WINO_PERSIST=0   the Winograd tower as per-layer launches instead of one launch (tower_wino4_kernel); CATTUS_WINO_SPIN=<n>:
 *                           polls a hand-off wait of that launch may take before it gives up (a launch that gave up is run again,
 *                           per layer: the tests set 1 to walk that path)
 *   CATTUS_WINO_INPLACE=0, CATTUS_ARENA=0     memory plan of the Winograd tower (a third activation buffer; separate allocations) */
int cattus_hip_create_diag(const void* weights, size_t nbytes, const cattus_eval_config* cfg, const char* switches, cattus_eval** out);

#ifdef __cplusplus
}
#endif
#endif /* CATTUS_HIP_DIAG_H */

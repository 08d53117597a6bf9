/*
 * trajnet_hip_profile.h -- MEASUREMENT hooks of libtrajnet_hip.so.  Not part of the drop-in boundary (trajnet_hip.h):
 * the reference has no counterpart, no caller of the reference's API needs them; bench.py's roofline leg is the only
 * user.  Plain C99 like the main header.
 */
#ifndef TRAJNET_HIP_PROFILE_H
#define TRAJNET_HIP_PROFILE_H

#include "trajnet_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* -------------------------------------------------------------------------------------------
 * Kernel timing hook: when enabled, every launch of the chosen kernel class (`which`: 0 = first
 * pooling-embedding layer -- the sparse kernel or the dense GEMM --, 1 = all GEMM launches) is
 * timed by a pair of hipEvents on the stream the kernel is launched on; tnp_profile_read
 * synchronises those events and returns the summed milliseconds and the launch count since
 * tnp_profile_begin (at most 32768 launches are recorded, later ones are not timed).
 * The sparse first layer's register-accumulator kernel -- the dominant kernel of the step --
 * carries the two events IN ITS DISPATCH (hipExtLaunchKernelGGL start / stop events: the begin
 * and end timestamps of the kernel's own AQL packet, the quantity rocprofv3's kernel trace
 * reports); every other launch is bracketed by two hipEventRecord calls, whose markers add
 * about 3 us to the span.  tnp_profile_dispatch_timed: how many of the launches returned by the
 * last tnp_profile_read were timed by their dispatch.
 * ----------------------------------------------------------------------------------------- */
TNP_API int tnp_profile_begin(int which);
TNP_API int tnp_profile_read(double *total_ms, int *launches);
TNP_API int tnp_profile_dispatch_timed(void);
TNP_API int tnp_profile_end(void);

/* -------------------------------------------------------------------------------------------
 * Tile-selection knobs (tests and tools/diag/small_step_probe.py pin a tile to show that results do not
 * depend on it; a caller never needs them).  Process-wide, not thread-safe against running launches.
 *   "sparse_tile"     value = (egos_per_tile << 8) | column_sets: tile of the sparse first embedding layer
 *                     (64|2, 32|2, 32|1, 16|1, 8|1, 4|1); 0 = automatic (the largest tile that gives about
 *                     one workgroup per CU).  Every tile gives the same result bit for bit.
 *   "sparse_min_wg"   workgroups a tile must reach before the next smaller one is tried (0 = one per CU)
 *   "skinny_max_rows" / "skinny_gates_max_rows"
 *                     tracks up to which the 16-track register-operand GEMMs (csrc/gemm_skinny.hip) run the
 *                     step's dense layers / LSTM gates (defaults 160 / 512; 0 = never)
 *   "sparse_wgrad_plan"
 *                     sparse first layer's weight gradient: 16 * waves + batches per trip (waves 4 / 8 / 16,
 *                     batches 1 / 2); 0 = by size (8 x 1 up to 16384 stacked rows, 4 x 1 beyond)
 *   "wgrad_min_rows" / "wgrad_target_wgs"
 *                     dense weight gradients: fewest rows of K per split (default 128) and the workgroup
 *                     count a contraction is split towards (512).  Plans only change the order of the sums.
 *   "fuse_prepare_grid"
 *                     occupancy / directional grids: track_prepare and the step's grid in one launch: 1 = while the
 *                     repeated position work is small (tracks x largest scene <= 196608; default), 2 = always,
 *                     0 = two launches.  Bit-identical either way.
 * Initial values: environment TNP_SPARSE_TILE="te,ncs", TNP_SPARSE_MIN_WG, TNP_SKINNY_MAX_M (both),
 * TNP_SKINNY_GATES_MAX_M, TNP_SPARSE_WGRAD_PLAN, TNP_WGRAD_MIN_ROWS, TNP_WGRAD_TARGET,
 * TNP_FUSE_PREPARE_GRID.
 * ----------------------------------------------------------------------------------------- */
TNP_API int tnp_tuning_set(const char *key, long value);

#ifdef __cplusplus
}
#endif

#endif /* TRAJNET_HIP_PROFILE_H */

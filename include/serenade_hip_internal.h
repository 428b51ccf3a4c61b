/* serenade_hip_internal.h -- measurement and test aids exported by libserenade_hip.so that are NOT part of the drop-in boundary.
 *
 * include/serenade_hip.h is what a reference-side binding binds: index construction, predict, the shard group, serving.  The three entry
 * points below exist for this repository's own tools and tests (tools/*.py, tests/test_gpu_parity.py): per-phase shader cycles, the per-rank time split of a
 * shard group, and a re-read of the SRN_* experiment knobs.  A host application has no use for them and must not rely on them.
 */
#ifndef SERENADE_HIP_INTERNAL_H
#define SERENADE_HIP_INTERNAL_H

#include "serenade_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Profiling aid: returns (and clears) 16 counters of shader cycles summed over workgroups, one per
 * kernel phase (DESIGN.md "Kernel phases"), accumulated by predict calls made while enabled; then
 * switches the accounting on (enable != 0) or off.  Not for production use. */
int srn_debug_phase_cycles(const srn_index_t* idx, int enable, uint64_t* out16);

/* Measurement aid (environment variable SRN_GROUP_TIMING=1 when the group is created): milliseconds of local shard 0's own launches in the group's last batch --
 * neighbours pipeline: prep records + front end | back end | top-n merge; lists pipeline: head + count + copy | prep + unsharded launch sequence | top-n merge.
 * The call that recorded them synchronised the stream: measurement runs only (tools/shard_rank_time.py). */
int srn_debug_shard_group_times(const srn_shard_group_t* g, double* out_ms3);

/* Measurement aid: how many queries of the last predict call the lean fast kernel listed for its MID instantiation (sessions of <= 10 items with 5..8 posting lists or
 * similarity numerators above 15: DESIGN.md section 4.1); srn_last_path_counts' `general` counts what reached the general kernel after both.  0 when the call had no such tier. */
int srn_debug_last_mid_count(const srn_index_t* idx, uint32_t* out_listed);
/* the persistent latency path: 100 MHz ticks of the last session the lean form's first resident workgroup served -- [0] waited for the doorbell, [1] doorbell -> prep record
 * written, [2] doorbell -> answer posted */
int srn_debug_serve_stamps(const srn_index_t* idx, uint32_t* out4);
/* 32-bit words per query of the streaming back end's exchange record at this (k, m) -- 0: none (knobs, shape); the gather form ships k + 1 words */
uint32_t srn_debug_shard_nb_positions_stride(size_t k, size_t m);
/* ... and how many of those MID listed for its BIG form (80 KB of LDS: merged lists beyond the 53 KB layout's buffers). */
int srn_debug_last_big_count(const srn_index_t* idx, uint32_t* out_listed);

/* Test aid: batches of this item shard that went through the wave-per-query back end of srn_sback.hip (the shard group's neighbours pipeline), since the shard was attached. */
int srn_debug_sback_launches(const srn_index_t* idx, uint64_t* out_launches);

/* Test / experiment knobs (environment variables SRN_NO_FAST, SRN_NO_MID, SRN_NO_MASKS, SRN_NO_MERGE, SRN_DENSE, SRN_HOT_SLOTS,
 * SRN_SKETCH_SLOTS, SRN_LDS_BUDGET_KB, SRN_GRID_MULT, SRN_DEBUG) force individual kernel code paths.  They are read ONCE,
 * when the library is first used -- never on the launch path; this call re-reads them (the parity tests switch paths
 * between calls).  Not for production use: make sure no predict call is in flight. */
void srn_debug_reload_knobs(void);

#ifdef __cplusplus
}
#endif
#endif

// ffx_internal.h -- what the translation units of libffx share beyond the public ABI (include/ffx.h).
// C++ linkage, no stability promise, not installed: comm.hip (the multi-GPU half) fills a plan's candidate lists and a
// merge's owner-side state IN PLACE, on its own routing stream, when a batch is routed by the ranks that own its
// queries (ffx_sharded_stage_owned): the lists arrive from the other ranks straight into the buffers the kernels read.
#ifndef FFX_INTERNAL_H_
#define FFX_INTERNAL_H_

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/ffx.h"
#include "../../include/ffx_private.h"

int ffx_fail(int code, const char* fmt, ...);
bool ffx_debug_variant(const char* name, long long* value);  // the FFX_DEBUG_VARIANT list of A/B switches (ffx.hip)
// The switches of FFX_DEBUG_VARIANT come in two kinds.  ffx_debug_variant: a choice between two paths that BOTH ship (the
// dispatcher reaches either under some input), kept so that tests can compare them on one input.  ffx_experiment: tuning
// hooks -- kernels and schedules that lost their A/B and that no input reaches (tools/kernel_ab.py, slice_sweep.py,
// sort_ab.py, dims_bench.py) -- compiled only into the experiments build (make EXPERIMENTS=1 -> libffx_experiments.so,
// loaded with FFX_LIBRARY=...); in the default build the switch is a constant `false` and the code behind it is not there.
#ifdef FFX_EXPERIMENTS
inline bool ffx_experiment(const char* name, long long* value) { return ffx_debug_variant(name, value); }
#else
inline bool ffx_experiment(const char*, long long*) { return false; }
#endif

// Room in the plan for n_queries + 1 offsets and n_pairs ordinals; the caller fills both device buffers on any stream.
// (Buffers may be re-allocated here: call it while nothing of the plan is in flight.)
int ffx_plan_adopt_begin(ffx_plan* pl, int64_t n_queries, int64_t n_pairs, int mode, int64_t** d_pair_off, int32_t** d_pair_ids);
// The lists are in place once `stream` reaches this point; host_pair_off holds the same offsets on the host.  The
// ordinals are validated ON `stream` (get_indices, index/util.py:29-41), asynchronously: *d_verdict is the device word
// that will hold the smallest position of an ordinal without rows (~0: none) ...
int ffx_plan_adopt_check(ffx_plan* pl, const int64_t* host_pair_off, hipStream_t stream, const unsigned long long** d_verdict);
// ... and once the caller has SYNCHRONISED `stream`, the plan is sized and ready (or FFX_NOT_FOUND + *bad_pair).
int ffx_plan_adopt_finish(ffx_plan* pl, hipStream_t stream, int64_t* bad_pair);
// Device room for n_queries query vectors, which the caller fills before the plan runs.
int ffx_plan_queries_room(ffx_plan* pl, int64_t n_queries, float** d_queries);
// A plan that owns no stream of its own and issues its work on `stream`.
int ffx_plan_create_on(ffx_index* idx, hipStream_t stream, ffx_plan** out);

// A merge that owns no stream (ffx_merge_run is given one; the set forms use the default stream).
int ffx_merge_create_on(int device, ffx_merge** out);
// The owner-side state of a merge written in place: offsets of the owned queries, arrival slot and first-stage score
// of every owned candidate (d_sparse is NULL without first-stage scores).
int ffx_merge_adopt_begin(ffx_merge* m, int64_t n_queries, int64_t n_pairs, bool have_sparse, int64_t** d_pair_off, int32_t** d_slot,
                          float** d_sparse);
// order_base: added to the positions the per-query order is made of (a slice of a larger list)
int ffx_merge_adopt_finish(ffx_merge* m, int64_t longest_query, int32_t order_base);
// results of the last ffx_merge_run copied out on `stream` (asynchronously for page-locked destinations)
int ffx_merge_fetch_async(ffx_merge* m, float* out_ff, float* out_interp, int32_t* out_order, hipStream_t stream);
int64_t ffx_merge_pairs(const ffx_merge* m);

#endif  // FFX_INTERNAL_H_

/*
 * ffx_private.h -- entry points of libffx that are NOT part of the stable contract (include/ffx.h).
 *
 * Exported by the same library and bound by this repository's own Python layer (fast_forward_amd/native.py), its tools
 * and its tests; they may change or disappear with any commit.  Three kinds:
 *   - building blocks that ffx.h's calls are made of, exposed so that the Python layer and the tests can drive them one by
 *     one (the owner-side merge, device-resident candidate lists, the windows of a streamed index)
 *   - host natives over NumPy arrays that keep the pandas-facing side of the package (Ranking, OnDiskIndex, sharded id
 *     tables) out of Python loops: threaded gathers, orderings, joins, string numbering
 *   - test / bench data generators and the rehearsal transport's inbox
 * Each names the reference lines it stands for, as in ffx.h.
 */
#ifndef FFX_PRIVATE_H_
#define FFX_PRIVATE_H_

#include "ffx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- building blocks of the index and the plan ----------------------------------------------------------- */
/* the same, returning once every row is gathered and its copy queued (not landed): the caller's next gather overlaps the
 * tail of the copies; ffx_index_upload_wait before the rows are used (it also finishes the code check of PQ rows) */
int ffx_index_upload_rows_async(ffx_index* idx, const void* const* chunks, int64_t n_chunks, int64_t chunk_rows,
                                const int64_t* rows, int64_t n_rows, int64_t dst_row, int32_t n_threads);
/* the same rows pulled by the DEVICE itself: every chunk map is page-locked host memory (ffx_host_pin each map once; a
 * memory-mapped file may be page-locked), a kernel on the index's copy stream reads the rows over PCIe.  No staging
 * block and no host thread touches the rows.  Asynchronous; ffx_index_upload_wait before the rows are used. */
int ffx_index_pull_rows_async(ffx_index* idx, const void* const* chunks, int64_t n_chunks, int64_t chunk_rows,
                              const int64_t* rows, int64_t n_rows, int64_t dst_row);
int ffx_index_upload_wait(ffx_index* idx);
/* whatever is queued on idx's copy stream from now on runs after what `before`'s copy stream holds so far (the windows
 * of a streamed index arrive one after the other instead of sharing the link) */
int ffx_index_upload_after(ffx_index* idx, ffx_index* before);
/* fill rows with the counter-based synthetic generator shared with the oracle
 * (value of element e of global row g: see fast_forward_amd/synth.py); bench/test data only */
int ffx_index_fill_synthetic(ffx_index* idx, uint64_t seed, float scale, int64_t row_offset,
                             int64_t n_rows, int64_t first_global_row);
/* dst is to score the same candidate lists as src (two plans of ONE index, e.g. alternating send buffers): the
 * validated lists are taken over device to device instead of being uploaded and validated again. */
int ffx_plan_copy_pairs(ffx_plan* dst, const ffx_plan* src);
/* ffx_plan_set_pairs for lists that already are in DEVICE memory (copied device to device); host_pair_off holds the
 * same offsets on the host.  No first-stage scores.  Used by the sharded scorer, which routes a batch on the device. */
int ffx_plan_set_pairs_device(ffx_plan* plan, const int64_t* dev_pair_off, const int64_t* host_pair_off, int64_t n_queries,
                              const int32_t* dev_pair_ids, int mode, int64_t* bad_pair);
/* gather_dot's duration (HIP events on the plan's stream) of the last n runs, oldest first, for a caller that queues
 * runs back to back without waiting in between (n <= 256 and <= the runs so far); waits for the stream */
int ffx_plan_recent_gather_ms(ffx_plan* plan, int32_t n, float* out_ms);

/* How much of the traffic between caller memory and the device the library staged through its own page-locked blocks
 * since the process began (csrc/host_xfer.h: ordinary memory above 1 MiB never reaches the runtime, which would page-lock
 * it on the fly): transfers and bytes.  For tests and tools. */
int ffx_host_xfer_stats(uint64_t* staged_transfers, uint64_t* staged_bytes);
/* release the page-locked staging blocks of every device (up to 64 MiB per device and direction (four lanes of two 8 MiB blocks); they come back on the next
 * staged transfer): for a host program that is done with the GPU for a while */
int ffx_host_xfer_release(void);
/* 1 for the experiments build (make EXPERIMENTS=1: the tuning hooks of FFX_DEBUG_VARIANT and the kernels only they reach), 0
 * for the default build, which ships only what its dispatcher can reach */
int ffx_build_has_experiments(void);

/* ---- merge: the owner-side half of a row-sharded index --------------------------------------
 * With the index row-sharded by document over several GPUs (one process each), every rank scores
 * the candidates whose documents it holds, and the dense scores of one query arrive at the
 * query's owner rank from all ranks (ffx_comm_exchange below).  The owner then scatters them back
 * into first-stage order, interpolates (ranking.py:319) and orders (ranking.py:115-117) -- the same
 * arithmetic as ffx_plan_run's epilogue, on a device buffer. */
typedef struct ffx_merge ffx_merge;
int ffx_merge_create(int device, ffx_merge** out);
int ffx_merge_destroy(ffx_merge* m);
/* pair_off: CSR of the owned queries' candidates in first-stage order; recv_slot[p]: where the
 * dense score of candidate p sits in the received buffer; sparse may be NULL */
int ffx_merge_set(ffx_merge* m, const int64_t* pair_off, int64_t n_queries, const int32_t* recv_slot,
                  const float* sparse);
/* the same with the arrival slots in DEVICE memory (offsets and first-stage scores still come from the host) */
int ffx_merge_set_device(ffx_merge* m, const int64_t* pair_off, int64_t n_queries, const int32_t* dev_recv_slot, const float* sparse);
/* recv_dev: DEVICE pointer to n_pairs floats.  stream: hipStream_t to run on (NULL: own stream) */
int ffx_merge_run(ffx_merge* m, const float* recv_dev, double alpha, uint32_t flags, void* stream);
int ffx_merge_sync(ffx_merge* m);
int ffx_merge_fetch(ffx_merge* m, float* out_ff, float* out_interp, int32_t* out_order);
int ffx_merge_device_ptr(ffx_merge* m, int which, void** ptr, int64_t* n_elems);

/* ---- the rehearsal transport ----------------------------------------------------------------------------- */
/* FFX_COMM_REHEARSAL only: what the peers WOULD send on the routing lane (ffx_sharded_stage_owned) from now on -- n_words
 * 4-byte words of DEVICE memory made by the caller, who knows the whole synthetic batch: per-query counts, candidate
 * ordinals, query vectors, in the order the lane receives them.  NULL: zeros again. */
int ffx_comm_rehearsal_inbox(ffx_comm* comm, const void* dev_words, int64_t n_words);

/* ---- string ids and run files ---------------------------------------------------------------------------- */
/* ffx_ids_lookup for n strings handed over as one buffer with the byte `sep` between them */
int ffx_ids_lookup_packed(const ffx_ids* ids, const char* data, int64_t n_bytes, char sep, int64_t n,
                          int32_t* out_ordinals, int64_t* first_missing);
/* ffx_ids_intern for n strings handed over as one buffer with `sep` between them */
int ffx_ids_intern_packed(ffx_ids* table, const char* data, int64_t n_bytes, char sep, int64_t n, int32_t* out_ordinals);
/* all strings in ordinal order joined by `sep`; *needed = bytes required, copied when cap suffices */
int ffx_ids_blob(const ffx_ids* table, char sep, char* buf, int64_t cap, int64_t* needed);
/* Numbering the n strings of one joined buffer without keeping a table (the factorisation of a run's id column,
 * ranking.py:95-98): codes[i] = number of string i, equal strings get equal numbers, numbers are dense; *names_blob
 * receives the distinct strings in number order, joined by `sep` (*names_bytes bytes, *n_names strings; release it
 * with ffx_buffer_free).  By host threads (n_threads <= 0: every core this process may use); the numbering does not
 * depend on the thread count. */
int ffx_strings_code(const char* data, int64_t n_bytes, char sep, int64_t n, int32_t n_threads, int32_t* codes, char** names_blob,
                     int64_t* names_bytes, int64_t* n_names);
/* ffx_strings_code for strings lying back to back in `data`, string i = data[offsets[i] .. offsets[i+1]) (the layout of
 * an Arrow string array: nothing to join, nothing to scan for separators); `sep` only joins *names_blob */
int ffx_strings_code_offsets(const char* data, const int64_t* offsets, int64_t n, char sep, int32_t n_threads, int32_t* codes,
                             char** names_blob, int64_t* names_bytes, int64_t* n_names);
/* ... with 32-bit offsets (an Arrow `string` array), and for n NUL-padded fields of `width` bytes (numpy 'S<width>') */
int ffx_strings_code_offsets32(const char* data, const int32_t* offsets, int64_t n, char sep, int32_t n_threads, int32_t* codes,
                               char** names_blob, int64_t* names_bytes, int64_t* n_names);
int ffx_strings_code_fixed(const char* data, int32_t width, int64_t n, char sep, int32_t n_threads, int32_t* codes, char** names_blob,
                           int64_t* names_bytes, int64_t* n_names);
/* *group = the first group of rows (group_off[g] .. group_off[g+1]) that holds one code twice, -1 if none: the repeated
 * (query, id) pair of ranking.py:95-98 for rows grouped by query that hold id codes in [0, n_codes) */
int ffx_host_repeated_in_groups(const int32_t* codes, const int64_t* group_off, int64_t n_groups, int64_t n_codes, int32_t n_threads,
                                int64_t* group);
int ffx_buffer_free(void* p);
/* owner[i] = crc32 (zlib's) of row i's document id -- its passage id if it has no document id -- modulo world: the rank of a
 * sharded index that holds the row (255: the row has neither id).  Ids as NUL-padded fixed-width fields (the id columns of
 * an index file, index/disk.py:145-165); either column may be NULL.  Each rank keeps and interns only its own rows' ids. */
int ffx_host_owner_crc32(const char* doc_ids, int32_t doc_width, const char* psg_ids, int32_t psg_width, int64_t n, int32_t world,
                         int32_t n_threads, uint8_t* owner);
/* Numbering n 8-byte values (the key is the value itself): codes[i] = number of values[i], dense, ascending with the
 * values; first_row[c] = a row that holds number c (first_row must hold n entries at most).  A column of a pandas frame is an
 * array of pointers to string objects shared by many rows: numbering the pointers leaves only the distinct objects to
 * be read as strings (the Ranking constructor, ranking.py:95-116, at tens of millions of rows). */
int ffx_host_code_u64(const uint64_t* values, int64_t n, int32_t n_threads, int32_t* codes, int64_t* first_row, int64_t* n_distinct);
int ffx_run_flags(const ffx_run* run, int32_t* q_flags, int32_t* id_flags, int32_t* misc_flags);
int ffx_run_ids_blob(const ffx_run* run, char* buf, int64_t cap, int64_t* needed);
/* A run opened WITHOUT an id table numbers its candidate ids while it is parsed (as ffx_strings_code would): ffx_run_export
 * hands out the code of every row, and this the distinct ids in code order joined by '\n' -- a view into the run, valid until
 * ffx_run_close (Ranking.from_file on a large run: no string per row ever exists, ranking.py:388-409). */
int ffx_run_names(const ffx_run* run, const char** blob, int64_t* bytes, int64_t* n_names);
/* Where the time of the calling thread's last ffx_run_open / ffx_run_write / ffx_run_save* / ffx_strings_code* call went:
 * "stage=seconds;stage=seconds;..." (*needed = bytes required incl. the NUL; copied when cap suffices). */
int ffx_host_stage_report(char* buf, int64_t cap, int64_t* needed);
/* Ranking.save (ranking.py:348-366) of a frame grouped by query, without a parsed run: q_blob holds
 * the n_queries query ids and id_blob the n_pairs candidate ids (frame order), each joined by '\n';
 * pair_off[q] .. pair_off[q+1] are the rows of query q.  Scores are printed like pandas prints
 * float32. */
int ffx_run_save(const char* path, const char* q_blob, int64_t q_bytes, int64_t n_queries, const int64_t* pair_off,
                 const char* id_blob, int64_t id_bytes, int64_t n_pairs, const float* scores, const char* tag);
/* the same for rows that hold id CODES (what Index.__call__ / Index.rerank / Ranking.from_file keep for a large
 * run): row p is written with string number id_codes[p] of name_blob (n_names strings joined by '\n') */
int ffx_run_save_coded(const char* path, const char* q_blob, int64_t q_bytes, int64_t n_queries, const int64_t* pair_off,
                       const char* name_blob, int64_t name_bytes, int64_t n_names, const int32_t* id_codes, int64_t n_pairs,
                       const float* scores, const char* tag);

/* ---- threaded host passes over the arrays of a Ranking / a streamed batch -------------------------------- */
/* per-query order computed on the HOST (threads over queries): scores descending, exact ties by position,
 * -0.0 == +0.0 -- the same order FFX_FLAG_SORT gives on the device (ranking.py:115-117), for scores that were
 * combined on the host (Ranking.interpolate, ranking.py:293-326).  out_order[p]: position in the whole array. */
int ffx_host_order(const float* scores, const int64_t* pair_off, int64_t n_queries, int32_t n_threads, int32_t* out_order);
/* the same with the queries' blocks laid out anew: query q's order is written from out_order[dst_off[q]] on (every query
 * ordered and the queries put into the constructor's "q_id descending" sequence in one pass); dst_off NULL = in place */
int ffx_host_order_blocks(const float* scores, const int64_t* pair_off, const int64_t* dst_off, int64_t n_queries, int32_t n_threads,
                          int32_t* out_order);
/* the same pass with the columns moved along: out_scores / out_carry (a second 4-byte column: the id codes) receive every
 * query's rows in final order at the query's destination block; any output may be NULL */
int ffx_host_rank_blocks(const float* scores, const int32_t* carry, const int64_t* pair_off, const int64_t* dst_off, int64_t n_queries,
                         int32_t n_threads, int32_t* out_order, float* out_scores, int32_t* out_carry);
/* out[i] = src[index[i]] for 4-byte elements (float32 scores, int32 id codes), by host threads (n_threads <= 0:
 * every core this process may use): reorders a run's columns by the per-query order of ffx_plan_fetch /
 * ffx_score / ffx_host_order (the other half of ranking.py:115-117's sort).  FFX_BAD_ARG for an index
 * outside [0, n_src). */
int ffx_host_take32(const void* src, int64_t n_src, const int32_t* index, int64_t n, int32_t n_threads, void* out);
/* Ranking.interpolate (ranking.py:319) on the host when the dense ranking's rows are a known permutation of the
 * first-stage ranking's (row i of the dense one is row source_row[i] of the sparse one):
 *   out[source_row[i]] = f32(a * sparse[source_row[i]]) + f32(b * dense_of[i])      (no FMA)
 * by host threads.  source_row must be a permutation of 0..n-1 (FFX_BAD_ARG otherwise); *any_nan (may be NULL)
 * is set when a result is NaN. */
int ffx_host_interpolate(const float* sparse, const float* dense_of, const int32_t* source_row, int64_t n, float a, float b,
                         int32_t n_threads, float* out, int32_t* any_nan);
/* numpy.unique(rows, return_inverse=True) for row numbers in 0 .. n_index_rows-1 without a sort (a bitmap over the
 * index's rows, marked and ranked by host threads): the distinct rows a streamed batch reads from the index file, ascending,
 * and every gathered row's position among them.  uniq must hold min(n, n_index_rows) entries. */
int ffx_host_unique_rows(const int64_t* rows, int64_t n, int64_t n_index_rows, int32_t n_threads, int64_t* uniq, int64_t* n_uniq, int32_t* inverse);
/* get_indices (index/util.py:29-41) for candidates that contribute one row each, fused with the above:
 * row(i) = table2 ? table2[table1[ids[i]]] : table1[ids[i]] (PASSAGE: table1 = rows of the passages; FIRSTP / one-row
 * documents: table1 = first entry of each document, table2 = rows of the entries), then the distinct rows ascending and
 * every candidate's slot among them.  An id or entry outside its table, or a row outside the index, is refused. */
int ffx_host_rows_unique(const int32_t* ids, int64_t n, const int32_t* table1, int64_t n1, const int32_t* table2, int64_t n2,
                         int64_t n_index_rows, int32_t n_threads, int64_t* uniq, int64_t* n_uniq, int32_t* inverse);
/* the same for candidates that contribute ALL rows of their document (MaxP / AveP over multi-passage documents):
 * entry_off[id] .. entry_off[id + 1] (n_docs + 1 offsets) are the document's entries, entry_row[] (n_entries) their rows.
 * ffx_host_segment_offsets: seg_off[0 .. n], seg_off[n] = R gathered rows; ffx_host_segment_rows_unique (seg_off as
 * returned): the distinct rows ascending and, in inverse[0 .. R), the slot of every gathered row among them. */
int ffx_host_segment_offsets(const int32_t* ids, int64_t n, const int32_t* entry_off, int64_t n_docs, int32_t n_threads, int64_t* seg_off);
int ffx_host_segment_rows_unique(const int32_t* ids, int64_t n, const int32_t* entry_off, int64_t n_docs, const int32_t* entry_row,
                                 int64_t n_entries, const int64_t* seg_off, int64_t n_index_rows, int32_t n_threads, int64_t* uniq,
                                 int64_t* n_uniq, int32_t* inverse);
/* The outer join of Ranking.interpolate / + (ranking.py:293-326) on integer keys, for two rankings whose rows stand
 * grouped by query: rows off_a[g] .. off_a[g+1] of A and off_b[g] .. off_b[g+1] of B belong to query g, code_a / code_b
 * number the ids in one common table, a pair occurs at most once per side (FFX_BAD_ARG otherwise).  Per query the union
 * of the id sets in ascending code: u_code, the row of A / of B holding the pair (u_a / u_b, -1 where missing); union
 * rows of query g are u_off[g] .. u_off[g+1].  u_code / u_a / u_b must hold off_a[n] + off_b[n] entries. */
int ffx_host_outer_join(const int64_t* off_a, const int32_t* code_a, const int64_t* off_b, const int32_t* code_b, int64_t n_queries,
                        int32_t n_threads, int64_t* u_off, int32_t* u_code, int32_t* u_a, int32_t* u_b);

/* numpy.repeat(arange(n_runs), diff(off)) as int64 by host threads: out[off[r] .. off[r+1]) = r (the per-row query numbers of a run
 * whose rows stand grouped by query; index/base.py:418-421 numbers queries in order of first appearance). */
int ffx_host_fill_runs(const int64_t* off, int64_t n_runs, int32_t n_threads, int64_t* out);
/* The allocations the first ffx_score / ffx_score_ranked call of an index would make on the way (its three plans and their
 * page-locked staging blocks, the kernel and result streams, the device buffers of a full slice), made now: a caller with
 * host work to do before its first call (Index.rerank resolves the ranking's ids) runs this beside it on another thread.
 * Sizes follow ffx_score's slicing for n_pairs candidates of n_queries queries; what turns out too small grows in the
 * call.  Not while another thread uses the index. */
int ffx_index_prepare_score(ffx_index* idx, int64_t n_queries, int64_t n_pairs, int32_t has_sparse, int32_t ranked, int32_t page_locked_buffers);

/* the placement probe of ffx_index_reseat alone, over the block the index has: milliseconds of a random gather of whole rows
 * (tools/placement_probe.py compares its ranking of several indexes with the scoring kernel's) */
int ffx_index_probe_placement(ffx_index* idx, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* FFX_PRIVATE_H_ */

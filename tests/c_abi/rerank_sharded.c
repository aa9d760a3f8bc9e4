/* rerank_sharded.c (test infrastructure: it links the oracle) -- the multi-GPU half of the C ABI from
 * plain C99, no Python and no PyTorch anywhere: N processes (fork), one communicator (ffx_comm_*),
 * an index row-sharded by document (ffx_sharded_*), one all-to-all of scores per batch, and every
 * owner checks its queries against the CPU oracle bit for bit -- once with the batch set on every rank (ffx_sharded_set_pairs)
 * and once as a fresh batch of which every rank hands over only the queries it owns (ffx_sharded_score).
 *
 *   rerank_sharded rccl        1 rank over RCCL on device 0 (the same calls an 8-GPU node makes)
 *   rerank_sharded host N      N ranks sharing device 0, scores exchanged through shared memory
 *                              (RCCL refuses ranks that share a GPU); the communicator id travels
 *                              from rank 0 to the others through a pipe
 *
 *   gcc -std=c99 -D_POSIX_C_SOURCE=200809L -Iinclude tests/c_abi/rerank_sharded.c -o rerank_sharded \
 *       -Lfast-forward-indexes_amd/fast_forward_amd -lffx -Loracle -lffx_oracle -lm
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/types.h>
#include <sys/wait.h>
#include <unistd.h>

#include "ffx.h"

void ffo_synth_rows(uint64_t seed, float scale, int64_t first_row, int64_t n_rows, int64_t dim, float* out);
int ffo_score(const float* vectors, int64_t n_rows, int64_t d, const int64_t* doc_off, const int32_t* doc_rows,
              int64_t n_docs, const int32_t* psg_row, int64_t n_psgs, const float* queries, int64_t n_queries,
              const int64_t* pair_off, const int32_t* pair_ids, int mode, const float* sparse, double alpha,
              float* out_ff, float* out_interp, int64_t* bad_pair);

#define CHECK(call)                                                                          \
  do {                                                                                       \
    int rc_ = (call);                                                                        \
    if (rc_ != FFX_OK) {                                                                     \
      fprintf(stderr, "rank %d: %s failed (%d): %s\n", rank, #call, rc_, ffx_last_error());  \
      return 1;                                                                              \
    }                                                                                        \
  } while (0)

enum { DIM = 768, N_DOCS = 600, PSG_PER_DOC = 2, N_ROWS = N_DOCS * PSG_PER_DOC, N_Q = 7, K = 150 };

static int run_rank(int transport, int rank, int world, const unsigned char* id) {
  const double alpha = 0.2;
  const float scale = 0.036f;
  ffx_comm* comm = NULL;
  CHECK(ffx_comm_init(transport, 0, rank, world, id, &comm));

  /* the whole corpus is described on every rank (ids are small); each rank uploads only its shard */
  float* rows = malloc(sizeof(float) * N_ROWS * DIM);
  float* queries = malloc(sizeof(float) * N_Q * DIM);
  ffo_synth_rows(21, scale, 0, N_ROWS, DIM, rows);
  ffo_synth_rows(22, scale, 0, N_Q, DIM, queries);
  static int64_t doc_off[N_DOCS + 1];
  static int32_t doc_rows[N_ROWS], psg_row[N_ROWS];
  for (int i = 0; i <= N_DOCS; ++i) doc_off[i] = (int64_t)i * PSG_PER_DOC;
  for (int i = 0; i < N_ROWS; ++i) doc_rows[i] = psg_row[i] = i;
  int64_t doc_bounds[65];
  for (int r = 0; r <= world; ++r) doc_bounds[r] = (int64_t)N_DOCS * r / world; /* contiguous document ranges */
  const int64_t d_lo = doc_bounds[rank], d_hi = doc_bounds[rank + 1], n_local = (d_hi - d_lo) * PSG_PER_DOC;

  ffx_index* shard = NULL;
  CHECK(ffx_index_create(0, n_local, DIM, &shard));
  CHECK(ffx_index_upload(shard, rows + (size_t)d_lo * PSG_PER_DOC * DIM, 0, n_local));
  static int64_t local_off[N_DOCS + 1];
  for (int64_t i = 0; i <= d_hi - d_lo; ++i) local_off[i] = i * PSG_PER_DOC;
  CHECK(ffx_index_set_doc_rows(shard, local_off, d_hi - d_lo, doc_rows, n_local)); /* doc_rows[i] = i: local rows */
  CHECK(ffx_index_set_psg_rows(shard, psg_row, n_local));

  static int64_t pair_off[N_Q + 1];
  static int32_t pair_ids[N_Q * K];
  static float sparse[N_Q * K];
  pair_off[0] = 0;
  for (int q = 0; q < N_Q; ++q) pair_off[q + 1] = pair_off[q] + (q == 3 ? 0 : K - 11 * q); /* ragged; query 3 is empty */
  for (int q = 0; q < N_Q; ++q)
    for (int64_t p = pair_off[q]; p < pair_off[q + 1]; ++p) {
      const int i = (int)(p - pair_off[q]);
      pair_ids[p] = (int32_t)((i * 7 + q * 13) % N_DOCS); /* GLOBAL document ordinals */
      sparse[p] = 25.0f - 0.1f * (float)i;
    }

  ffx_sharded* s = NULL;
  int64_t bad = -1;
  CHECK(ffx_sharded_create(shard, comm, doc_bounds, &s));
  CHECK(ffx_sharded_set_queries(s, queries, N_Q));
  CHECK(ffx_sharded_set_pairs(s, pair_off, N_Q, pair_ids, sparse, FFX_MODE_MAXP, &bad));
  for (int step = 0; step < 3; ++step) /* back-to-back steps alternate between the two plans */
    CHECK(ffx_sharded_run(s, alpha, FFX_FLAG_EXACT_SUM | FFX_FLAG_SORT));
  int64_t q_lo, q_hi, first, n_own;
  CHECK(ffx_sharded_owned(s, &q_lo, &q_hi, &first, &n_own));
  static float ff[N_Q * K], interp[N_Q * K], want_ff[N_Q * K], want_interp[N_Q * K];
  static int32_t order[N_Q * K];
  CHECK(ffx_sharded_fetch(s, ff, interp, order));

  if (ffo_score(rows, N_ROWS, DIM, doc_off, doc_rows, N_DOCS, psg_row, N_ROWS, queries, N_Q, pair_off, pair_ids, FFX_MODE_MAXP,
                sparse, alpha, want_ff, want_interp, &bad) != 0) {
    fprintf(stderr, "oracle failed\n");
    return 1;
  }
  if (memcmp(ff, want_ff + first, sizeof(float) * (size_t)n_own) != 0 || memcmp(interp, want_interp + first, sizeof(float) * (size_t)n_own) != 0) {
    fprintf(stderr, "rank %d: scores of the owned queries differ from the oracle\n", rank);
    return 1;
  }
  for (int64_t q = q_lo; q < q_hi; ++q)
    for (int64_t p = pair_off[q] + 1; p < pair_off[q + 1]; ++p) {
      const int32_t a = order[p - first - 1], b = order[p - first]; /* positions inside the owned block */
      if (interp[a] < interp[b] || (interp[a] == interp[b] && a > b)) {
        fprintf(stderr, "rank %d: query %lld is not in order\n", rank, (long long)q);
        return 1;
      }
    }
  /* The same batch as a FRESH one, the way the reference meets a batch (a new set of queries on every call, index/base.py:418-429):
   * every rank hands over ONLY the block of queries it owns -- their vectors, candidate lists (offsets from 0), first-stage
   * scores -- and ffx_sharded_score routes the candidates to their holders, scores, exchanges, merges, orders and returns this
   * rank's block, host buffers to host buffers.  Same bits as above. */
  {
    int64_t q_bounds[65];
    CHECK(ffx_query_partition(N_Q, world, q_bounds));
    if (q_bounds[rank] != q_lo || q_bounds[rank + 1] != q_hi) {
      fprintf(stderr, "rank %d: ffx_query_partition disagrees with ffx_sharded_owned\n", rank);
      return 1;
    }
    static int64_t own_off[N_Q + 1];
    for (int64_t q = q_lo; q <= q_hi; ++q) own_off[q - q_lo] = pair_off[q] - first;
    static float ff2[N_Q * K], interp2[N_Q * K];
    static int32_t order2[N_Q * K];
    CHECK(ffx_sharded_score(s, q_bounds, queries + (size_t)q_lo * DIM, own_off, pair_ids + first, NULL, sparse + first, FFX_MODE_MAXP, alpha,
                            FFX_FLAG_EXACT_SUM | FFX_FLAG_SORT, 2, ff2, interp2, order2, &bad));
    if (memcmp(ff2, want_ff + first, sizeof(float) * (size_t)n_own) != 0 || memcmp(interp2, want_interp + first, sizeof(float) * (size_t)n_own) != 0 ||
        memcmp(order2, order, sizeof(int32_t) * (size_t)n_own) != 0) {
      fprintf(stderr, "rank %d: the fresh-batch form (ffx_sharded_score) differs from the set-once form\n", rank);
      return 1;
    }
  }
  ffx_shard_stats st;
  CHECK(ffx_sharded_stats(s, &st));
  double checked = (double)n_own;
  CHECK(ffx_comm_allreduce_f64(comm, &checked, 1, FFX_OP_SUM));
  CHECK(ffx_comm_barrier(comm));
  if (rank == 0)
    printf("rerank_sharded.c ok: %d rank(s), transport %d, %.0f owned pairs checked bit for bit, rank 0 gathered %lld rows, sent %lld bytes\n",
           world, st.transport, checked, (long long)st.local.n_rows, (long long)st.send_bytes);
  /* an ordinal outside every shard is reported on every rank, before any device work */
  pair_ids[5] = N_DOCS + 9;
  if (ffx_sharded_set_pairs(s, pair_off, N_Q, pair_ids, sparse, FFX_MODE_MAXP, &bad) != FFX_NOT_FOUND || bad != 5) {
    fprintf(stderr, "rank %d: expected FFX_NOT_FOUND at pair 5\n", rank);
    return 1;
  }
  CHECK(ffx_sharded_destroy(s));
  CHECK(ffx_index_destroy(shard));
  CHECK(ffx_comm_destroy(comm));
  free(rows);
  free(queries);
  return checked == (double)pair_off[N_Q] ? 0 : 1;
}

int main(int argc, char** argv) {
  const int rccl = argc > 1 && strcmp(argv[1], "rccl") == 0;
  const int world = rccl ? 1 : (argc > 2 ? atoi(argv[2]) : 2);
  const int transport = rccl ? FFX_COMM_RCCL : FFX_COMM_HOST;
  unsigned char id[FFX_COMM_ID_BYTES];
  if (world < 1 || world > 6) return 2;
  if (world == 1) {
    if (ffx_comm_unique_id(transport, id) != FFX_OK) {
      fprintf(stderr, "ffx_comm_unique_id: %s\n", ffx_last_error());
      return 1;
    }
    return run_rank(transport, 0, 1, id);
  }
  /* fork BEFORE any GPU call: rank 0 (the parent) draws the id afterwards and pipes it to the children */
  int pipes[6][2];
  pid_t kids[6];
  for (int r = 1; r < world; ++r) {
    if (pipe(pipes[r]) != 0) return 2;
    kids[r] = fork();
    if (kids[r] == 0) {
      close(pipes[r][1]);
      if (read(pipes[r][0], id, sizeof id) != (ssize_t)sizeof id) return 3;
      return run_rank(transport, r, world, id);
    }
    close(pipes[r][0]);
  }
  if (ffx_comm_unique_id(transport, id) != FFX_OK) return 1;
  for (int r = 1; r < world; ++r)
    if (write(pipes[r][1], id, sizeof id) != (ssize_t)sizeof id) return 3;
  int status = run_rank(transport, 0, world, id);
  for (int r = 1; r < world; ++r) {
    int st = 0;
    waitpid(kids[r], &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) status = status ? status : 1;
  }
  return status;
}

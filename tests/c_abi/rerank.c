/* rerank.c (test infrastructure: it links the oracle) -- the C ABI of libffx (include/ffx.h) used from plain C99, no Python anywhere:
 * build an HBM-resident index, score a batch of (query, candidate) pairs with MaxP aggregation,
 * interpolate with first-stage scores, get the per-query order back -- and check every number
 * against the CPU oracle (oracle/ffx_oracle.c, test infrastructure) bit for bit.
 *
 *   gcc -std=c99 -Iinclude tests/c_abi/rerank.c -o rerank \
 *       -Lfast-forward-indexes_amd/fast_forward_amd -lffx -Loracle -lffx_oracle -lm
 *
 * tests/test_gpu_parity.py::test_c_example_runs_against_the_oracle builds and runs it on the GPU box.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ffx.h"

/* the oracle's entry points used here (oracle/ffx_oracle.c) */
void ffo_synth_rows(uint64_t seed, float scale, int64_t first_row, int64_t n_rows, int64_t dim, float* out);
int ffo_score(const float* vectors, int64_t n_rows, int64_t d, const int64_t* doc_off, const int32_t* doc_rows,
              int64_t n_docs, const int32_t* psg_row, int64_t n_psgs, const float* queries, int64_t n_queries,
              const int64_t* pair_off, const int32_t* pair_ids, int mode, const float* sparse, double alpha,
              float* out_ff, float* out_interp, int64_t* bad_pair);

#define CHECK(call)                                                              \
  do {                                                                           \
    int rc_ = (call);                                                            \
    if (rc_ != FFX_OK) {                                                         \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, ffx_last_error());     \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main(void) {
  enum { DIM = 768, N_DOCS = 500, PSG_PER_DOC = 3, N_ROWS = N_DOCS * PSG_PER_DOC, N_Q = 6, K = 200 };
  const double alpha = 0.2;
  const float scale = 0.036f;

  /* rows: the counter-based generator shared by the oracle and the device (the device could
   * also fill itself: ffx_index_fill_synthetic); documents own 3 consecutive rows each */
  float* rows = malloc(sizeof(float) * N_ROWS * DIM);
  float* queries = malloc(sizeof(float) * N_Q * DIM);
  ffo_synth_rows(11, scale, 0, N_ROWS, DIM, rows);
  ffo_synth_rows(12, scale, 0, N_Q, DIM, queries);
  int64_t doc_off[N_DOCS + 1];
  int32_t doc_rows[N_ROWS], psg_row[N_ROWS];
  for (int i = 0; i <= N_DOCS; ++i) doc_off[i] = (int64_t)i * PSG_PER_DOC;
  for (int i = 0; i < N_ROWS; ++i) doc_rows[i] = psg_row[i] = i;

  /* candidates: K distinct documents per query, first-stage scores descending */
  int64_t pair_off[N_Q + 1];
  static int32_t pair_ids[N_Q * K];
  static float sparse[N_Q * K];
  for (int q = 0; q <= N_Q; ++q) pair_off[q] = (int64_t)q * K;
  for (int q = 0; q < N_Q; ++q)
    for (int i = 0; i < K; ++i) {
      pair_ids[q * K + i] = (int32_t)((i * 7 + q * 13) % N_DOCS); /* 7 and 500 are coprime: distinct */
      sparse[q * K + i] = 25.0f - 0.1f * (float)i;
    }

  ffx_index* index = NULL;
  CHECK(ffx_index_create(0, N_ROWS, DIM, &index));
  CHECK(ffx_index_upload(index, rows, 0, N_ROWS));
  CHECK(ffx_index_set_doc_rows(index, doc_off, N_DOCS, doc_rows, N_ROWS));
  CHECK(ffx_index_set_psg_rows(index, psg_row, N_ROWS));

  static float ff[N_Q * K], interp[N_Q * K], want_ff[N_Q * K], want_interp[N_Q * K];
  static int32_t order[N_Q * K];
  int64_t bad = -1;
  ffx_stats stats;
  CHECK(ffx_score(index, queries, N_Q, pair_off, pair_ids, sparse, FFX_MODE_MAXP, alpha,
                  FFX_FLAG_EXACT_SUM | FFX_FLAG_SORT, ff, interp, order, &bad, &stats));

  if (ffo_score(rows, N_ROWS, DIM, doc_off, doc_rows, N_DOCS, psg_row, N_ROWS, queries, N_Q, pair_off, pair_ids,
                FFX_MODE_MAXP, sparse, alpha, want_ff, want_interp, &bad) != 0) {
    fprintf(stderr, "oracle failed\n");
    return 1;
  }
  if (memcmp(ff, want_ff, sizeof ff) != 0 || memcmp(interp, want_interp, sizeof interp) != 0) {
    fprintf(stderr, "scores differ from the oracle\n");
    return 1;
  }
  for (int q = 0; q < N_Q; ++q)
    for (int i = 1; i < K; ++i) {
      const int32_t a = order[q * K + i - 1], b = order[q * K + i];
      if (a / K != q || b / K != q || interp[a] < interp[b] || (interp[a] == interp[b] && a > b)) {
        fprintf(stderr, "query %d is not in order at rank %d\n", q, i);
        return 1;
      }
    }

  /* The reference's documented two-call pipeline -- dense = index(first), then first.interpolate(dense, alpha)
   * (index/base.py:389-469, ranking.py:293-326) -- without scoring twice: the first call returns the dense scores in final
   * order and leaves them in HBM (FFX_FLAG_KEEP_DENSE) with the carried words (here: the candidate ordinals themselves); the
   * second hands over first-stage scores and alpha for the SAME candidate lists and gets interpolated scores + carried words in
   * final order -- the interpolation and the ordering alone.  Checked against the oracle's scores in the order found above. */
  {
    static float dense_ranked[N_Q * K], ranked[N_Q * K];
    static int32_t dense_carry[N_Q * K], carry[N_Q * K];
    CHECK(ffx_score_ranked(index, queries, N_Q, pair_off, pair_ids, NULL, pair_ids, FFX_MODE_MAXP, 0.0, FFX_FLAG_EXACT_SUM | FFX_FLAG_KEEP_DENSE,
                           dense_ranked, dense_carry, NULL, &bad, NULL));
    uint64_t token = 0;
    int64_t kept_q = 0, kept_p = 0;
    CHECK(ffx_index_kept(index, &token, &kept_q, &kept_p));
    if (token == 0 || kept_q != N_Q || kept_p != pair_off[N_Q]) {
      fprintf(stderr, "nothing kept after FFX_FLAG_KEEP_DENSE\n");
      return 1;
    }
    for (int q = 0; q < N_Q; ++q)
      for (int i = 1; i < K; ++i)
        if (dense_ranked[q * K + i - 1] < dense_ranked[q * K + i]) {
          fprintf(stderr, "dense scores of query %d are not in order\n", q);
          return 1;
        }
    CHECK(ffx_rank_kept(index, token, N_Q, pair_off, sparse, alpha, FFX_FLAG_EXACT_SUM, ranked, carry, NULL, NULL));
    for (int64_t p = 0; p < pair_off[N_Q]; ++p)
      if (memcmp(&ranked[p], &want_interp[order[p]], sizeof(float)) != 0 || carry[p] != pair_ids[order[p]]) {
        fprintf(stderr, "ffx_rank_kept differs from the oracle at rank %lld\n", (long long)p);
        return 1;
      }
    if (ffx_rank_kept(index, token + 1, N_Q, pair_off, sparse, alpha, 0, ranked, carry, NULL, NULL) == FFX_OK) {
      fprintf(stderr, "a stale token was accepted\n");
      return 1;
    }
  }

  /* an id the index does not hold is reported with its position */
  pair_ids[321] = N_DOCS + 5;
  int rc = ffx_score(index, queries, N_Q, pair_off, pair_ids, sparse, FFX_MODE_MAXP, alpha, FFX_FLAG_EXACT_SUM, ff, interp,
                     NULL, &bad, NULL);
  if (rc != FFX_NOT_FOUND || bad != 321) {
    fprintf(stderr, "expected FFX_NOT_FOUND at pair 321, got %d at %lld\n", rc, (long long)bad);
    return 1;
  }

  printf("rerank.c ok: %lld pairs, %lld rows gathered, kernel variant %d, best of query 0: doc %d score %.6f\n",
         (long long)stats.n_pairs, (long long)stats.n_rows, stats.kernel_variant, pair_ids[order[0]], (double)interp[order[0]]);
  CHECK(ffx_index_destroy(index));
  free(rows);
  free(queries);
  return 0;
}

/* Plain C caller of the drop-in boundary (include/probly_search_amd.h): the reference's README
 * example (README.md:99-141 of probly-search 2.0.1) - two documents, title + description fields,
 * the query "abc" scored with BM25 - through ps_index_* / ps_index_query.
 * Prints one "key score-bits" line per result; exit code = the ps_status of the query
 * (PS_ENODEVICE on a machine without a HIP device: there is no CPU scoring fallback). */
#include <inttypes.h>
#include <stdio.h>
#include <string.h>

#include "probly_search_amd.h"

static ps_str S(const char* s) {
  ps_str r;
  r.ptr = s;
  r.len = strlen(s);
  return r;
}

int main(void) {
  ps_index* idx = NULL;
  if (ps_index_new(2, &idx) != PS_OK) return 100;
  /* doc 0: title "abc", description "dfg"; doc 1: title "dfgh", description "abcd" */
  const size_t one_each[2] = {1, 1};
  ps_str d0[2], d1[2];
  d0[0] = S("abc"); d0[1] = S("dfg");
  d1[0] = S("dfgh"); d1[1] = S("abcd");
  if (ps_index_add_document(idx, 0, d0, one_each, NULL, NULL) != PS_OK) return 101;
  if (ps_index_add_document(idx, 1, d1, one_each, NULL, NULL) != PS_OK) return 102;
  printf("docs %zu nodes %zu\n", ps_index_docs_len(idx), ps_index_count_nodes(idx));

  ps_scorer_desc bm25;
  memset(&bm25, 0, sizeof bm25);
  bm25.kind = PS_SCORER_BM25;
  bm25.bm25_k1 = 1.2;
  bm25.bm25_b = 0.75;
  const double boosts[2] = {1.0, 1.0};
  ps_result* out = NULL;
  size_t n = 0;
  const char* q = "abc";
  ps_status rc = ps_index_query(idx, &bm25, q, strlen(q), boosts, 2, NULL, NULL, 0 /* every match */, &out, &n);
  if (rc != PS_OK) {
    printf("query status %d: %s\n", (int)rc, ps_last_error());
    ps_index_free(idx);
    return (int)rc;
  }
  for (size_t i = 0; i < n; ++i) {
    uint64_t bits;
    memcpy(&bits, &out[i].score, 8);
    printf("result %" PRIu64 " %016" PRIx64 "\n", out[i].key, bits);
  }
  ps_free(out);
  ps_index_free(idx);
  return 0;
}

/* A ScoreCalculator plugin written in C99 against include/probly_search_amd.h: BM25
 * (src/score/default/bm25.rs:35-93 of the reference) as before_each / score callbacks, run by
 * the library's host walk (PS_SCORER_HOST_CALLBACKS).  Mode "host": print the results (the test
 * compares them with the oracle).  Mode "gpu": also run the built-in GPU BM25 and compare. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "probly_search_amd.h"

typedef struct { double idf, eb; } calc_t;
static unsigned long g_calls = 0;

static int before_each(void* user, const ps_term_data* te, size_t df, size_t n_docs, const ps_index* idx, void** memory) {
  (void)user; (void)idx;
  ++g_calls;
  calc_t* m = (calc_t*)malloc(sizeof(calc_t));
  size_t f = df < n_docs ? df : n_docs;
  m->idf = log(1.0 + ((double)(n_docs - f) + 0.5) / ((double)f + 0.5));
  if (te->query_term.len == te->query_term_expanded.len &&
      memcmp(te->query_term.ptr, te->query_term_expanded.ptr, te->query_term.len) == 0)
    m->eb = 1.0;
  else
    m->eb = log(1.0 + (1.0 / (1.0 + (double)te->query_term_expanded.len - (double)te->query_term.len)));
  *memory = m;
  return 1;
}

static int score(void* user, const void* memory, const ps_document_pointer* dp, const ps_document_details* dd,
                 uint64_t node, const ps_field_data* fd, const ps_term_data* te, double* out) {
  (void)user; (void)node; (void)te;
  ++g_calls;
  const calc_t* m = (const calc_t*)memory;
  const double k1 = 1.2, b = 0.75;
  double s = 0.0;
  for (size_t x = 0; x < fd->n_fields; ++x) {
    double tf = (double)dp->term_frequency[x];
    if (tf > 0.0) {
      double tfn = ((k1 + 1.0) * tf) / (k1 * ((1.0 - b) + b * ((double)dd->field_length[x] / fd->fields[x].avg)) + tf);
      s += tfn * m->idf * fd->fields_boost[x] * m->eb;
    }
  }
  if (s > 0.0) { *out = s; return 1; }
  return 0;
}

static void drop_memory(void* user, void* memory) { (void)user; free(memory); }

static void add(ps_index* idx, uint64_t key, const char* f0, const char* f1) {
  ps_str vals[2] = {{f0, strlen(f0)}, {f1, strlen(f1)}};
  size_t counts[2] = {1, 1};
  if (ps_index_add_document(idx, key, vals, counts, NULL, NULL) != PS_OK) { printf("add failed: %s\n", ps_last_error()); exit(2); }
}

int main(int argc, char** argv) {
  const int gpu = argc > 1 && strcmp(argv[1], "gpu") == 0;
  ps_index* idx = NULL;
  if (ps_index_new(2, &idx) != PS_OK) return 2;
  add(idx, 0, "abc", "dfg");
  add(idx, 1, "dfgh abc", "abcd");
  add(idx, 2, "x", "abc abc q");
  const double boosts[2] = {1.0, 2.0};
  ps_score_callbacks cb = {before_each, score, NULL, drop_memory, NULL};
  ps_scorer_desc plugin = {PS_SCORER_HOST_CALLBACKS, 0, 0.0, 0.0, &cb};
  ps_scorer_desc builtin = {PS_SCORER_BM25, 0, 1.2, 0.75, NULL};
  const char* queries[3] = {"abc", "ab dfg", "q x abc"};
  int same = 0;
  unsigned long calls_builtin = 0;
  for (int i = 0; i < 3; ++i) {
    ps_result* r = NULL;
    size_t n = 0;
    ps_status st = ps_index_query(idx, &plugin, queries[i], strlen(queries[i]), boosts, 2, NULL, NULL, 0, &r, &n);
    if (st != PS_OK) { printf("plugin query failed (%d): %s\n", (int)st, ps_last_error()); return (int)st; }
    for (size_t k = 0; k < n; ++k) {
      unsigned long long bits;
      memcpy(&bits, &r[k].score, 8);
      printf("cb [%s] %llu %016llx\n", queries[i], (unsigned long long)r[k].key, bits);
    }
    if (gpu) {
      ps_result* g = NULL;
      size_t gn = 0;
      const unsigned long before = g_calls;
      st = ps_index_query(idx, &builtin, queries[i], strlen(queries[i]), boosts, 2, NULL, NULL, 0, &g, &gn);
      calls_builtin += g_calls - before;
      if (st != PS_OK) { printf("gpu query failed (%d): %s\n", (int)st, ps_last_error()); return (int)st; }
      if (gn == n && (n == 0 || memcmp(g, r, n * sizeof(ps_result)) == 0)) ++same;
      ps_free(g);
    }
    ps_free(r);
  }
  if (gpu) {
    printf("gpu == callbacks: %d queries bit-identical\n", same);
    printf("callback invocations during built-in queries: %lu\n", calls_builtin);
    if (same != 3) return 1;
  }
  ps_index_free(idx);
  return 0;
}

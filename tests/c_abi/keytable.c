/* A plain C99 caller of the key table (include/probly_search_amd.h, ps_keytable_*): what a binding for
 * `Index<T>` with a non-integer T (src/index.rs:19-33 of the reference) does around the u64 entry points.
 * Host code only: runs without a HIP device.  Prints "ok" and returns 0, or the failing line. */
#include <stdio.h>
#include <string.h>

#include "probly_search_amd.h"

#define CHECK(c) do { if (!(c)) { printf("failed at line %d: %s (%s)\n", __LINE__, #c, ps_last_error()); return 1; } } while (0)

typedef struct { unsigned char uuid[16]; } doc_key; /* T: Eq + Hash + Copy */

int main(void) {
  ps_keytable* kt = NULL;
  CHECK(ps_keytable_new(&kt) == PS_OK && kt);
  doc_key a, b;
  memset(&a, 0x11, sizeof a);
  memset(&b, 0x22, sizeof b);
  a.uuid[7] = 0; /* NUL bytes inside a key are fine: keys are (pointer, length) */
  uint64_t ia = 99, ib = 99, again = 99;
  int fresh = -1;
  CHECK(ps_keytable_intern(kt, &a, sizeof a, &ia, &fresh) == PS_OK && ia == 0 && fresh == 1);
  CHECK(ps_keytable_intern(kt, &b, sizeof b, &ib, &fresh) == PS_OK && ib == 1 && fresh == 1);
  CHECK(ps_keytable_intern(kt, &a, sizeof a, &again, &fresh) == PS_OK && again == 0 && fresh == 0);
  CHECK(ps_keytable_len(kt) == 2);
  uint64_t found = 99;
  CHECK(ps_keytable_find(kt, &b, sizeof b, &found) == 1 && found == 1);
  doc_key c;
  memset(&c, 0x33, sizeof c);
  CHECK(ps_keytable_find(kt, &c, sizeof c, &found) == 0); /* remove_document of an unknown key: skip the call */

  /* ids are the u64 keys of the index; results come back as ids and are resolved in one call */
  ps_index* idx = NULL;
  CHECK(ps_index_new(1, &idx) == PS_OK);
  ps_str va = {"abc dfg", 7}, vb = {"dfgh abcd", 9};
  size_t one = 1;
  CHECK(ps_index_add_document(idx, ia, &va, &one, NULL, NULL) == PS_OK);
  CHECK(ps_index_add_document(idx, ib, &vb, &one, NULL, NULL) == PS_OK);
  CHECK(ps_index_docs_len(idx) == 2);
  ps_result res[2] = {{1, 0.5}, {0, 0.25}}; /* (what a query entry point would have returned) */
  ps_str keys[2];
  CHECK(ps_keytable_resolve(kt, res, 2, keys) == PS_OK);
  CHECK(keys[0].len == sizeof b && memcmp(keys[0].ptr, &b, sizeof b) == 0);
  CHECK(keys[1].len == sizeof a && memcmp(keys[1].ptr, &a, sizeof a) == 0);
  ps_result bad = {7, 0.0};
  CHECK(ps_keytable_resolve(kt, &bad, 1, keys) == PS_EINVAL);

  /* bulk form: a column of string keys -> the u64 column ps_index_add_documents_flat takes */
  const char* bytes = "k-1k-2k-1";
  const uint64_t offsets[4] = {0, 3, 6, 9};
  uint64_t ids[3];
  CHECK(ps_keytable_intern_flat(kt, 3, bytes, offsets, ids) == PS_OK);
  CHECK(ids[0] == 2 && ids[1] == 3 && ids[2] == 2 && ps_keytable_len(kt) == 4);
  ps_str k;
  CHECK(ps_keytable_key(kt, 3, &k) == PS_OK && k.len == 3 && memcmp(k.ptr, "k-2", 3) == 0);

  ps_index_free(idx);
  ps_keytable_free(kt);
  printf("ok\n");
  return 0;
}

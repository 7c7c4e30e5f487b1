/* Two ranks (processes) drive the sharded-batch path through the C ABI only:
 *   ps_comm_get_unique_id (rank 0) -> pipe -> ps_comm_init_rank (both) ->
 *   ps_snapshot_query_batch_allgather_flat (each rank scores its shard, the blocks are all-gathered).
 * usage: two_ranks <corpus file> <world> <top_k>
 * corpus file lines:  "D <key>\t<field0>\t<field1>"  |  "Q <query>"
 * Every rank prints "rank r query i n=<count> key:scorebits ..." for the WHOLE batch.
 * With PS_COMM_TRANSPORT=hostshm both ranks may share one GPU (debug transport); without it the
 * exchange is ncclAllGather and rank r uses device r. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

#include "probly_search_amd.h"

#define MAXQ 64
static char* qtext[MAXQ];
static int nq = 0;

static void die(const char* what) {
  printf("FAILED %s: %s\n", what, ps_last_error());
  exit(3);
}

static int run_rank(const char* path, const unsigned char* uid, int world, int rank, size_t top_k) {
  const int shared_gpu = getenv("PS_COMM_TRANSPORT") && strcmp(getenv("PS_COMM_TRANSPORT"), "hostshm") == 0;
  const int device = shared_gpu ? 0 : rank;
  ps_index* idx = NULL;
  if (ps_index_new(2, &idx) != PS_OK) die("ps_index_new");
  FILE* f = fopen(path, "r");
  if (!f) { printf("cannot open %s\n", path); return 2; }
  char line[4096];
  while (fgets(line, sizeof(line), f)) {
    size_t n = strlen(line);
    while (n && (line[n - 1] == '\n' || line[n - 1] == '\r')) line[--n] = 0;
    if (line[0] == 'D') {
      char* key = line + 2;
      char* f0 = strchr(key, '\t'); *f0++ = 0;
      char* f1 = strchr(f0, '\t'); *f1++ = 0;
      ps_str vals[2] = {{f0, strlen(f0)}, {f1, strlen(f1)}};
      size_t counts[2] = {1, 1};
      if (ps_index_add_document(idx, strtoull(key, NULL, 10), vals, counts, NULL, NULL) != PS_OK) die("add_document");
    } else if (line[0] == 'Q' && nq < MAXQ) {
      qtext[nq++] = strdup(line + 2);
    }
  }
  fclose(f);
  ps_snapshot* snap = NULL;
  ps_status st = ps_index_snapshot(idx, device, 256, &snap);
  if (st != PS_OK) { printf("snapshot failed (%d): %s\n", (int)st, ps_last_error()); return (int)st; }
  ps_comm* comm = NULL;
  st = ps_comm_init_rank(uid, world, rank, device, &comm);
  if (st != PS_OK) { printf("comm init failed (%d): %s\n", (int)st, ps_last_error()); return (int)st; }
  if (ps_comm_world_size(comm) != world || ps_comm_rank(comm) != rank) die("comm identity");
  /* contiguous shards of equal size (pad with empty queries) */
  const int per = (nq + world - 1) / world;
  char text[8192];
  uint64_t offsets[MAXQ + 1];
  size_t pos = 0;
  for (int i = 0; i < per; ++i) {
    const int q = rank * per + i;
    offsets[i] = pos;
    if (q < nq) { memcpy(text + pos, qtext[q], strlen(qtext[q])); pos += strlen(qtext[q]); }
  }
  offsets[per] = pos;
  const size_t bb = ps_topk_block_bytes((size_t)per, top_k);
  void *d_local = NULL, *d_all = NULL;
  if (hipMalloc(&d_local, bb) != hipSuccess || hipMalloc(&d_all, bb * (size_t)world) != hipSuccess) die("hipMalloc");
  const double boosts[2] = {1.0, 1.0};
  ps_scorer_desc sc = {PS_SCORER_BM25, 0, 1.2, 0.75, NULL};
  st = ps_snapshot_query_batch_allgather_flat(snap, comm, &sc, text, offsets, (size_t)per, boosts, 2, NULL, NULL, top_k,
                                              d_local, d_all, NULL);
  if (st != PS_OK) { printf("allgather failed (%d): %s\n", (int)st, ps_last_error()); return (int)st; }
  unsigned char* h = (unsigned char*)malloc(bb * (size_t)world);
  if (hipMemcpy(h, d_all, bb * (size_t)world, hipMemcpyDeviceToHost) != hipSuccess) die("hipMemcpy");
  for (int r = 0; r < world; ++r) {
    const unsigned char* blk = h + (size_t)r * bb;
    const uint64_t* keys = (const uint64_t*)blk;
    const uint64_t* bits = (const uint64_t*)(blk + (size_t)per * top_k * 8);
    const uint32_t* cnt = (const uint32_t*)(blk + (size_t)per * top_k * 16);
    for (int i = 0; i < per && r * per + i < nq; ++i) {
      printf("rank %d query %d n=%u", rank, r * per + i, cnt[i]);
      for (uint32_t k = 0; k < cnt[i]; ++k)
        printf(" %llu:%016llx", (unsigned long long)keys[(size_t)i * top_k + k], (unsigned long long)bits[(size_t)i * top_k + k]);
      printf("\n");
    }
  }
  free(h);
  hipFree(d_local);
  hipFree(d_all);
  ps_comm_free(comm);
  ps_snapshot_free(snap);
  ps_index_free(idx);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 4) { printf("usage: two_ranks <corpus> <world> <top_k>\n"); return 2; }
  const int world = atoi(argv[2]);
  const size_t top_k = (size_t)atoi(argv[3]);
  /* rank 0's id travels to the other ranks through pipes; no HIP call happens before the fork */
  int pipes[8][2];
  for (int r = 1; r < world; ++r) if (pipe(pipes[r]) != 0) return 2;
  pid_t kids[8];
  for (int r = 1; r < world; ++r) {
    kids[r] = fork();
    if (kids[r] == 0) {
      unsigned char uid[PS_COMM_ID_BYTES];
      if (read(pipes[r][0], uid, sizeof(uid)) != (ssize_t)sizeof(uid)) return 2;
      return run_rank(argv[1], uid, world, r, top_k);
    }
  }
  unsigned char uid[PS_COMM_ID_BYTES];
  if (ps_comm_get_unique_id(uid) != PS_OK) { printf("unique id failed: %s\n", ps_last_error()); return 6; }
  for (int r = 1; r < world; ++r) if (write(pipes[r][1], uid, sizeof(uid)) != (ssize_t)sizeof(uid)) return 2;
  int rc = run_rank(argv[1], uid, world, 0, top_k);
  for (int r = 1; r < world; ++r) {
    int status = 0;
    waitpid(kids[r], &status, 0);
    if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) rc = rc ? rc : 10 + r;
  }
  fflush(stdout);
  return rc;
}

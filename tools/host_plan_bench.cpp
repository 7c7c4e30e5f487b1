// Dev tool: times the host planner of a 1024-query batch (thread pool + merge) on a saved C2 snapshot, with an idle gap
// between batches as in a serving loop.  usage: host_plan_bench <threads> <gap_us>   (tools/host_plan_bench.sh builds and runs it)
#include "../probly-search_amd/csrc/ps_snapshot.hpp"
#include "../probly-search_amd/csrc/ps_pool.hpp"
#include "../probly-search_amd/csrc/ps_capi_internal.hpp"
#include <chrono>
#include <cstdio>
#include <fstream>
#include <string>
#include <thread>
#include <vector>
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main(int argc, char** argv) {
  unsigned want = argc > 1 ? atoi(argv[1]) : 16;
  int gap_us = argc > 2 ? atoi(argv[2]) : 300;
  ps_snapshot* s = nullptr;
  if (ps_snapshot_load("/tmp/ps_plan_bench/c2.snap", -1, &s) != PS_OK) { printf("load failed\n"); return 1; }
  std::vector<std::string> qs; std::ifstream f("/tmp/ps_plan_bench/queries.txt"); std::string l; while (std::getline(f, l)) qs.push_back(l);
  ps_scorer_desc sc{}; sc.kind = PS_SCORER_BM25; sc.bm25_k1 = 1.2; sc.bm25_b = 0.75;
  ps::Pool pool(want - 1);
  const ps::Snapshot& snap = *s->snap;
  for (int rep = 0; rep < 64; ++rep) {
    const size_t n = 1024, base = (size_t)(rep % 64) * 1024;
    auto t0 = clk::now();
    struct alignas(256) PP { ps::Plan p; }; std::vector<PP> pparts(want);
    auto t1 = clk::now();
    pool.run([&](unsigned part, unsigned nparts) {
      size_t b = n * part / nparts, e = n * (part + 1) / nparts;
      ps::Plan& pl = pparts[part].p;
      pl.qbeg.assign(1, 0);
      for (size_t i = b; i < e; ++i) snap.plan_query(sc, qs[base + i], nullptr, nullptr, pl);
    });
    auto t2 = clk::now();
    ps::Plan plan; plan.qbeg.assign(1, 0);
    for (PP& pp : pparts) { ps::Plan& pl = pp.p;
      const uint32_t b0 = (uint32_t)plan.entries.size();
      plan.entries.insert(plan.entries.end(), pl.entries.begin(), pl.entries.end());
      for (size_t i = 1; i < pl.qbeg.size(); ++i) plan.qbeg.push_back(b0 + pl.qbeg[i]);
      plan.qterms_len.insert(plan.qterms_len.end(), pl.qterms_len.begin(), pl.qterms_len.end());
    }
    auto t3 = clk::now();
    if (rep >= 56) printf("alloc %.1f  run %.1f  merge %.1f  total %.1f us\n", us(t0, t1), us(t1, t2), us(t2, t3), us(t0, t3));
    std::this_thread::sleep_for(std::chrono::microseconds(gap_us));
  }
}

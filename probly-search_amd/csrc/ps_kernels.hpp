// ps_kernels.hpp — device code of the query-scoring path (gfx950 / CDNA4), one file per kernel family:
//   ps_kernels_common.hpp  KParams, K1d work descriptors, work counters, wave helpers, wave top-K
//   ps_kernels_score.hpp   K0 k_bm25_lut, K0b k_dense_rows, K1 k_score, K2 k_z21, K3 k_merge          (streaming kernels)
//   ps_kernels_daat.hpp    K1d k_daat / k_daat_small, filters, score planes, K3d k_merge_items         (pruning kernels, BM25)
//   ps_kernels_plan.hpp    k_plan / k_plan_scan (device planner), k_upload, k_pack_tfl, k_pack_results
// (the device-side preparation of a K1d batch: ps_prep_kernels.hpp; zero_to_one's pruning kernel: ps_z21_daat.hpp).  Included by
// ps_engine.hip only (one translation unit); see that file's header comment for the kernel overview and DESIGN.md section 3.
#pragma once
#include "ps_kernels_common.hpp"
#include "ps_kernels_score.hpp"
#include "ps_kernels_daat.hpp"
#include "ps_kernels_plan.hpp"

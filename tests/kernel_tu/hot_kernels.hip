// Instantiates the kernels of the hot path for tests/test_kernel_resources.py (register / scratch budget check with
// hipcc -Rpass-analysis=kernel-resource-usage; a few seconds instead of the whole engine translation unit).
#include "../../probly-search_amd/csrc/ps_z21_daat.hpp"
namespace ps {
template __global__ void k_daat_small<2, true>(const KParams);
template __global__ void k_daat_small<2, false>(const KParams);
template __global__ void k_daat_small<1, false>(const KParams);
template __global__ void k_daat_small<2, true, 3>(const KParams);
template __global__ void k_daat_small<2, false, 3>(const KParams);
template __global__ void k_daat<2, true>(const KParams);
template __global__ void k_daat<2, false>(const KParams);
template __global__ void k_daat_z<2, true>(const KParams);
template __global__ void k_daat_z<2, false>(const KParams);
template __global__ void k_daat_z<1, false>(const KParams);
template __global__ void k_daat_z<2, false, 8>(const KParams);
template __global__ void k_zprep_query<4>(const ZPrepParams);
template __global__ void k_zprep_query<8>(const ZPrepParams);
template __global__ void k_score<MODE_BM25, 2, false, false, 8>(const KParams);
template __global__ void k_score<MODE_BM25, 1, false, false, 8>(const KParams);
template __global__ void k_score<MODE_BM25, 1, false, false, 4>(const KParams);
template __global__ void k_score<MODE_Z21S, 2, false, false, 8>(const KParams);
}

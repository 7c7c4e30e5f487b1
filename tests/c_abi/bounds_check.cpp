// CPU check of ps_bounds.hpp (the host side of the two-field joint bound): for any positive finite fields_boost b the cone
// coefficients (lo, a, c) satisfy a * w_lo + c * w_(lo+1) >= b componentwise, and the interpolated bound a * H[lo] + c * H[lo+1]
// dominates b . v for every point of random non-negative point sets whose direction supports H were computed the way
// k_list_bounds computes them.  Prints "ok" or the first violation.
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../probly-search_amd/csrc/ps_bounds.hpp"

int main() {
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> u01(0.0, 1.0);
  auto boost = [&](int kind) {
    switch (kind % 6) {
      case 0: return std::pow(10.0, -300.0 + 600.0 * u01(rng));           // anywhere in the double range
      case 1: return 0.5 + u01(rng);                                      // around 1
      case 2: return 5e-324 * (double)(1 + (rng() % 1000));               // subnormal
      case 3: return 1e300 * (0.5 + u01(rng));
      case 4: return 1.0;
      default: return std::pow(2.0, (double)((int)(rng() % 41) - 20));
    }
  };
  double dirs[ps::BOUND_NDIR][2];
  for (int d = 0; d < ps::BOUND_NDIR; ++d) ps::bound_dir(d, dirs[d][0], dirs[d][1]);
  if (dirs[0][0] != 1.0 || dirs[0][1] != 0.0 || dirs[ps::BOUND_NDIR - 1][0] != 0.0 || dirs[ps::BOUND_NDIR - 1][1] != 1.0) {
    printf("the end directions are not the axes\n");
    return 1;
  }
  for (int it = 0; it < 200000; ++it) {
    const double b[2] = {boost(it), boost(it / 6 + it)};
    uint32_t lo = 99;
    double a = -1, c = -1;
    ps::boost_cone(b, lo, a, c);
    if (lo > (uint32_t)(ps::BOUND_NDIR - 2) || !(a >= 0.0) || !(c >= 0.0)) { printf("bad cone for (%a, %a): lo %u a %a c %a\n", b[0], b[1], lo, a, c); return 1; }
    const double r0 = a * dirs[lo][0] + c * dirs[lo + 1][0], r1 = a * dirs[lo][1] + c * dirs[lo + 1][1];
    // (the fallback of boost_cone - never taken for finite positive boosts - makes c huge: still a valid dominating combination)
    if (!(r0 >= b[0]) || !(r1 >= b[1])) { printf("cone does not dominate (%a, %a): lo %u -> (%a, %a)\n", b[0], b[1], lo, r0, r1); return 1; }
    if (it % 50 == 0) {
      // a random "list": points (tfn_0, tfn_1) >= 0, many on the axes (a posting that holds the term in one field only)
      const int n = 1 + (int)(rng() % 200);
      std::vector<double> v0(n), v1(n);
      for (int i = 0; i < n; ++i) {
        const int kind = (int)(rng() % 3);
        v0[i] = kind == 1 ? 0.0 : 2.2 * u01(rng);
        v1[i] = kind == 0 ? 0.0 : 2.2 * u01(rng);
      }
      double H[ps::BOUND_NDIR];
      for (int d = 0; d < ps::BOUND_NDIR; ++d) {
        H[d] = 0.0;
        for (int i = 0; i < n; ++i) H[d] = std::fmax(H[d], dirs[d][0] * v0[i] + dirs[d][1] * v1[i]);
      }
      const double bound = (a * H[lo] + c * H[lo + 1]) * (1.0 + 1e-12) + 0x1p-1066;  // (prep_entry_ub's inflation, relative and absolute)
      for (int i = 0; i < n; ++i) {
        const double s = b[0] * v0[i] + b[1] * v1[i];
        if (std::isfinite(s) && !(bound >= s)) { printf("bound %a below b.v %a for (%a, %a), point (%a, %a)\n", bound, s, b[0], b[1], v0[i], v1[i]); return 1; }
      }
    }
  }
  printf("ok\n");
  return 0;
}

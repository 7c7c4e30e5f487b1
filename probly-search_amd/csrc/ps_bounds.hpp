// ps_bounds.hpp - host side of the two-field joint bound (ps_prep_kernels.hpp, "direction supports"): the fixed directions and the
// decomposition of a fields_boost vector into the two directions around it.  Plain C++ (no HIP): the engine uses it
// (ensure_list_bounds), tests/c_abi/bounds_check.cpp checks its guarantees on the CPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>

namespace ps {

constexpr int BOUND_NDIR = 17;  // directions of the two-field joint bound: angles 0 .. 90 degrees in equal steps (ends = the per-field maxima);
                                // == PREP_NDIR of the device code (ps_prep_kernels.hpp; static_assert in ps_engine.hip)

// The direction w_d = (cos, sin) of angle d * 90 / (BOUND_NDIR - 1) degrees; the ends are exactly the axes.
inline void bound_dir(int d, double& c, double& sn) {
  if (d == 0) { c = 1.0; sn = 0.0; return; }
  if (d == BOUND_NDIR - 1) { c = 0.0; sn = 1.0; return; }
  const double th = (3.14159265358979323846 / 2.0) * (double)d / (double)(BOUND_NDIR - 1);
  c = std::cos(th); sn = std::sin(th);
}
// boosts (two positive finite numbers) as a conic combination of the two directions around them: lo, a, b with
// a * w_lo + b * w_(lo+1) >= boosts componentwise (verified below; inflated by what the solve may have rounded away), so that
// a * H[lo] + b * H[lo+1] bounds boosts . v for every point v >= 0 of a list.
inline void boost_cone(const double* boosts, uint32_t& lo, double& a, double& b) {
  const double b0 = boosts[0], b1 = boosts[1];
  const double th = std::atan2(b1, b0), step = (3.14159265358979323846 / 2.0) / (double)(BOUND_NDIR - 1);
  int d = (int)std::floor(th / step);
  d = std::max(0, std::min(BOUND_NDIR - 2, d));
  for (int tries = 0; tries < 3; ++tries) {
    double c0, s0, c1, s1;
    bound_dir(d, c0, s0); bound_dir(d + 1, c1, s1);
    const double det = c0 * s1 - s0 * c1;
    double al = (b0 * s1 - b1 * c1) / det, be = (c0 * b1 - s0 * b0) / det;
    if (al < 0.0 && d > 0 && tries < 2) { --d; continue; }                  // (rounding put the angle one sector off)
    if (be < 0.0 && d < BOUND_NDIR - 2 && tries < 2) { ++d; continue; }
    al = std::max(al, 0.0); be = std::max(be, 0.0);
    const double r0 = al * c0 + be * c1, r1 = al * s0 + be * s1;
    double scale = 1.0;
    if (!(r0 >= b0)) scale = std::max(scale, b0 / r0);
    if (!(r1 >= b1)) scale = std::max(scale, b1 / r1);
    if (!(scale >= 1.0) || !std::isfinite(scale)) break;
    scale *= 1.0 + 1e-12;
    double fa = al * scale, fb = be * scale;
    // the guarantee is checked on the FINAL coefficients, with the products the device forms (among subnormal boosts the
    // inflation above rounds away: a few ulps more, one at a time)
    bool ok = false;
    for (int bump = 0; bump < 256; ++bump) {
      if (fa * c0 + fb * c1 >= b0 && fa * s0 + fb * s1 >= b1) { ok = true; break; }
      fa = std::nextafter(fa, INFINITY);
      fb = std::nextafter(fb, INFINITY);
    }
    if (!ok || !std::isfinite(fa) || !std::isfinite(fb)) break;
    lo = (uint32_t)d; a = fa; b = fb;
    return;
  }
  // (cannot happen for the boosts K1d admits - positive and finite; the per-field sum of maxima is always valid)
  lo = 0; a = b0; b = b1 * 1e308;  // b * H[1] = +inf for any list with field-1 postings: min(ub_m, ub_j) keeps ub_m
}

}  // namespace ps

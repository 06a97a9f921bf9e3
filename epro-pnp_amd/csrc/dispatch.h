// dispatch.h -- turn run-time (dof, points-per-lane, bounds, block-size class) into template arguments.
#pragma once
#include <type_traits>

namespace pnp {

template <int V>
using ic = std::integral_constant<int, V>;

// f(DOF, PPL, BND, MAXW) with integral_constant / bool_constant arguments; returns int.
template <class F>
inline int dispatch_shape(int dof, int ppl, bool bnd, int waves, F&& f) {
  auto d3 = [&](auto DOF, auto PPL, auto BND) -> int {
    // block-size class: the register budget per lane is 512 / (waves per SIMD) -> 512 / 256 / 128 VGPRs
    if (waves <= 4) return f(DOF, PPL, BND, ic<4>{});
    if (waves <= 8) return f(DOF, PPL, BND, ic<8>{});
    return f(DOF, PPL, BND, ic<16>{});
  };
  auto d2 = [&](auto DOF, auto PPL) -> int {
    return bnd ? d3(DOF, PPL, std::true_type{}) : d3(DOF, PPL, std::false_type{});
  };
  auto d1 = [&](auto DOF) -> int {
    switch (ppl) {
      case 1: return d2(DOF, ic<1>{});
      case 2: return d2(DOF, ic<2>{});
      case 4: return d2(DOF, ic<4>{});
      default: return d2(DOF, ic<8>{});
    }
  };
  return (dof == 6) ? d1(ic<6>{}) : d1(ic<4>{});
}

template <class F>
inline int dispatch_dof_bounds(int dof, bool bnd, F&& f) {
  auto d2 = [&](auto DOF) -> int { return bnd ? f(DOF, std::true_type{}) : f(DOF, std::false_type{}); };
  return (dof == 6) ? d2(ic<6>{}) : d2(ic<4>{});
}

}  // namespace pnp

// stencil_math.hpp -- the 7-point operator at one cell, shared by the streaming kernels
// (stencil.hip) and the single-workgroup small-level kernel (tail.hip) so both evaluate
// exactly the same expression tree (reference operators.7pt.c:49-89).
#pragma once
#include "common.hpp"

namespace hpgmg {

// A x at one cell, exactly as the reference's macro evaluates it.
template <int V>
__device__ __forceinline__ double apply_op_7pt(double xc, double xim, double xip, double xjm, double xjp, double xkm, double xkp,
                                               double bi0, double bi1, double bj0, double bj1, double bk0, double bk1,
                                               double alpha, double a, double b, double h2inv) {
  if (V == HPGMG_HIP_7PT_CC) {
    double s = xip + xim;
    s = s + xjp;
    s = s + xjm;
    s = s + xkp;
    s = s + xkm;
    s = s - xc * 6.0;
    return a * xc - (b * h2inv) * s;
  } else {
    double s = bi1 * (xip - xc);
    s = s + bi0 * (xim - xc);
    s = s + bj1 * (xjp - xc);
    s = s + bj0 * (xjm - xc);
    s = s + bk1 * (xkp - xc);
    s = s + bk0 * (xkm - xc);
    if (V == HPGMG_HIP_7PT_VC_HELMHOLTZ) return (a * alpha) * xc - (b * h2inv) * s;
    return ((-b) * h2inv) * s;
  }
}


}  // namespace hpgmg

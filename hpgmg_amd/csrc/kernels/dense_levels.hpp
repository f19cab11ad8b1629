// dense_levels.hpp -- small levels addressed by GLOBAL cell coordinate (tail.hip: a whole level on one workgroup; brick_visit.hip: a level as
// bricks of 16^3 cells, one workgroup each): where a global cell lives in the boxed layout, and the per-cell coefficient record the lanes keep in registers.
#pragma once
#include "common.hpp"

namespace hpgmg {

// dimensions are almost always powers of two: divide by shifting then (a wave-uniform choice), since an
// integer division costs ~40 instructions and a level visit needs a dozen per cell
struct IDiv { int d, sh; };
__device__ __forceinline__ IDiv idiv_of(int d) { IDiv r; r.d = d; r.sh = (d > 0 && (d & (d - 1)) == 0) ? __builtin_ctz(d) : -1; return r; }
__device__ __forceinline__ int operator/(int x, const IDiv &D) { return D.sh >= 0 ? (x >> D.sh) : (x / D.d); }
__device__ __forceinline__ int operator%(int x, const IDiv &D) { return D.sh >= 0 ? (x & (D.d - 1)) : (x % D.d); }

struct CellRef { int box, ijk; };                    // where a global cell lives in the boxed layout
struct LevelGeom { IDiv D, bd, nb; int jS, kS; };
__device__ __forceinline__ LevelGeom geom_of(const hpgmg_hip_level &L) {
  LevelGeom G; G.D = idiv_of(L.dim_i); G.bd = idiv_of(L.dim); G.nb = idiv_of(L.dim_i / G.bd); G.jS = L.jStride; G.kS = L.kStride; return G;
}
__device__ __forceinline__ CellRef locate(const LevelGeom &G, int gi, int gj, int gk) {
  const int bi = gi / G.bd, bj = gj / G.bd, bk = gk / G.bd;
  CellRef r;
  r.box = bi + G.nb.d * (bj + G.nb.d * bk);
  r.ijk = (gi - bi * G.bd.d) + (gj - bj * G.bd.d) * G.jS + (gk - bk * G.bd.d) * G.kS;
  return r;
}

template <int V>
struct CellCoef { double bi0, bi1, bj0, bj1, bk0, bk1, al, dinv, rhs; };

}  // namespace hpgmg

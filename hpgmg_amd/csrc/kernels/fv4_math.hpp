// fv4_math.hpp -- the bracket of the 4th-order operator (reference operators.fv4.c:87-108) in the pieces the LDS-tiled kernels evaluate it in:
// the 25 values of the iterate and the 30 coefficient values of one cell are READ FIRST (each exactly once, through volatile LDS pointers: the
// backend then issues ds_read_b64 and does not merge pairs of them into ds_read2_b64, which moves half as many bytes per LDS cycle on gfx950),
// then combined in the reference macro's order: T * (six face terms) + (0.25 T) * (twelve mixed terms), every group summed left to right,
// a mixed term = (beta+ - beta-) * (((x1 - x2) - x3) + x4).  Shared by fv4_tile.hpp and fv4_rb.hpp: one expression tree, bit-identical results.
#pragma once
#include "common.hpp"

namespace hpgmg {
#ifndef FV4_TWELFTH
#define FV4_TWELFTH ( 0.0833333333333333333)
#endif
namespace fv4rb {
typedef const volatile double __attribute__((address_space(3))) *ldsr;
struct B18 { double f[6], d[12]; };    // what the stencil takes from the coefficients at one cell: six face values, twelve differences
// operators.fv4.c:87-108 read at a cell of three beta_i / beta_j planes and two beta_k faces (row stride WB)
template <int W>
__device__ __forceinline__ void beta18(B18 &o, const double *I0, const double *Im, const double *Ip, const double *J0, const double *Jm, const double *Jp,
                                       const double *K0, const double *K1) {
  o.f[0] = I0[0]; o.f[1] = I0[1]; o.f[2] = J0[0]; o.f[3] = J0[W]; o.f[4] = K0[0]; o.f[5] = K1[0];
  o.d[0] = I0[W] - I0[-W];         o.d[1] = Ip[0] - Im[0];
  o.d[2] = J0[1] - J0[-1];         o.d[3] = Jp[0] - Jm[0];
  o.d[4] = K0[1] - K0[-1];         o.d[5] = K0[W] - K0[-W];
  o.d[6] = I0[1 + W] - I0[1 - W];  o.d[7] = Ip[1] - Im[1];
  o.d[8] = J0[W + 1] - J0[W - 1];  o.d[9] = Jp[W] - Jm[W];
  o.d[10] = K1[1] - K1[-1];        o.d[11] = K1[W] - K1[-W];
}
// The same in pieces, so that a stage issues its LDS reads in a few large groups ahead of the arithmetic (the wave has one other wave to hide
// behind, so every round trip to the LDS that the instruction stream exposes is paid in full) without holding all 55 operands at once:
// group 1 = what the six face terms and the first five mixed terms take, group 2 = the rest
struct BG1 { double i00, i01, j00, jw0, k00, k10, i0p, i0m, ip0, im0, j01, j0m, jp0, jm0, k0p, k0m; };
struct BG2 { double k0w, k0mw, i0pp, i0mp, ip1, im1, jwp, jwm, jpw, jmw, k1p, k1m, k1w, k1mw; };
template <int W>
__device__ __forceinline__ void beta_g1(BG1 &r, ldsr I0, ldsr Im, ldsr Ip, ldsr J0, ldsr Jm, ldsr Jp, ldsr K0, ldsr K1) {
  r.i00 = I0[0]; r.i01 = I0[1]; r.j00 = J0[0]; r.jw0 = J0[W]; r.k00 = K0[0]; r.k10 = K1[0];
  r.i0p = I0[W]; r.i0m = I0[-W]; r.ip0 = Ip[0]; r.im0 = Im[0]; r.j01 = J0[1]; r.j0m = J0[-1]; r.jp0 = Jp[0]; r.jm0 = Jm[0]; r.k0p = K0[1]; r.k0m = K0[-1];
}
template <int W>
__device__ __forceinline__ void beta_g2(BG2 &r, ldsr I0, ldsr Im, ldsr Ip, ldsr J0, ldsr Jm, ldsr Jp, ldsr K0, ldsr K1) {
  r.k0w = K0[W]; r.k0mw = K0[-W]; r.i0pp = I0[1 + W]; r.i0mp = I0[1 - W]; r.ip1 = Ip[1]; r.im1 = Im[1];
  r.jwp = J0[W + 1]; r.jwm = J0[W - 1]; r.jpw = Jp[W]; r.jmw = Jm[W]; r.k1p = K1[1]; r.k1m = K1[-1]; r.k1w = K1[W]; r.k1mw = K1[-W];
}
// the same read at a cell of a box's own arrays in memory (p: the cell in the box's level vectors): what the reference reads for a cell of that box
__device__ __forceinline__ void beta18_global(B18 &o, gcptr p, size_t vol, int jS, int kS) {
  gcptr I = p + (size_t)VECTOR_BETA_I * vol, J = p + (size_t)VECTOR_BETA_J * vol, K = p + (size_t)VECTOR_BETA_K * vol;
  o.f[0] = I[0]; o.f[1] = I[1]; o.f[2] = J[0]; o.f[3] = J[jS]; o.f[4] = K[0]; o.f[5] = K[kS];
  o.d[0] = I[jS] - I[-jS];          o.d[1] = I[kS] - I[-kS];
  o.d[2] = J[1] - J[-1];            o.d[3] = J[kS] - J[-kS];
  o.d[4] = K[1] - K[-1];            o.d[5] = K[jS] - K[-jS];
  o.d[6] = I[1 + jS] - I[1 - jS];   o.d[7] = I[1 + kS] - I[1 - kS];
  o.d[8] = J[jS + 1] - J[jS - 1];   o.d[9] = J[jS + kS] - J[jS - kS];
  o.d[10] = K[kS + 1] - K[kS - 1];  o.d[11] = K[kS + jS] - K[kS - jS];
}
// the 25 values of the iterate the stencil reads: centre, +-1 / +-2 along each axis, the four in-plane diagonals, the four in-plane
// neighbours on the planes below (m_) and above (p_)
struct X25 { double c, im1, ip1, im2, ip2, jm1, jp1, jm2, jp2, km1, kp1, km2, kp2, mm, pm, mp, pp, m_im, m_ip, m_jm, m_jp, p_im, p_ip, p_jm, p_jp; };
// the bracket of operators.fv4.c:87-108 in fv4_tile.hpp's (= the reference's) order: T * (six face terms) + (0.25 T) * (twelve mixed terms)
__device__ __forceinline__ double fv4_sum(const X25 &x, const B18 &b) {
  double s1 = b.f[0] * (15.0 * (x.im1 - x.c) - (x.im2 - x.ip1));
  s1 = s1 + b.f[1] * (15.0 * (x.ip1 - x.c) - (x.ip2 - x.im1));
  s1 = s1 + b.f[2] * (15.0 * (x.jm1 - x.c) - (x.jm2 - x.jp1));
  s1 = s1 + b.f[3] * (15.0 * (x.jp1 - x.c) - (x.jp2 - x.jm1));
  s1 = s1 + b.f[4] * (15.0 * (x.km1 - x.c) - (x.km2 - x.kp1));
  s1 = s1 + b.f[5] * (15.0 * (x.kp1 - x.c) - (x.kp2 - x.km1));
  double s2 = b.d[0] * (x.mp - x.jp1 - x.mm + x.jm1);
  s2 = s2 + b.d[1] * (x.p_im - x.kp1 - x.m_im + x.km1);
  s2 = s2 + b.d[2] * (x.pm - x.ip1 - x.mm + x.im1);
  s2 = s2 + b.d[3] * (x.p_jm - x.kp1 - x.m_jm + x.km1);
  s2 = s2 + b.d[4] * (x.m_ip - x.ip1 - x.m_im + x.im1);
  s2 = s2 + b.d[5] * (x.m_jp - x.jp1 - x.m_jm + x.jm1);
  s2 = s2 + b.d[6] * (x.pp - x.jp1 - x.pm + x.jm1);
  s2 = s2 + b.d[7] * (x.p_ip - x.kp1 - x.m_ip + x.km1);
  s2 = s2 + b.d[8] * (x.pp - x.ip1 - x.mp + x.im1);
  s2 = s2 + b.d[9] * (x.p_jp - x.kp1 - x.m_jp + x.km1);
  s2 = s2 + b.d[10] * (x.p_ip - x.ip1 - x.p_im + x.im1);
  s2 = s2 + b.d[11] * (x.p_jp - x.jp1 - x.p_jm + x.jm1);
  return FV4_TWELFTH * s1 + (0.25 * FV4_TWELFTH) * s2;
}
// fv4_sum in two steps (the same expression tree): what depends on the iterate only, then the products with the coefficients
struct Br18 { double a[6], m[12]; };
__device__ __forceinline__ void fv4_brackets(Br18 &o, const X25 &x) {
  o.a[0] = 15.0 * (x.im1 - x.c) - (x.im2 - x.ip1);
  o.a[1] = 15.0 * (x.ip1 - x.c) - (x.ip2 - x.im1);
  o.a[2] = 15.0 * (x.jm1 - x.c) - (x.jm2 - x.jp1);
  o.a[3] = 15.0 * (x.jp1 - x.c) - (x.jp2 - x.jm1);
  o.a[4] = 15.0 * (x.km1 - x.c) - (x.km2 - x.kp1);
  o.a[5] = 15.0 * (x.kp1 - x.c) - (x.kp2 - x.km1);
  o.m[0] = x.mp - x.jp1 - x.mm + x.jm1;
  o.m[1] = x.p_im - x.kp1 - x.m_im + x.km1;
  o.m[2] = x.pm - x.ip1 - x.mm + x.im1;
  o.m[3] = x.p_jm - x.kp1 - x.m_jm + x.km1;
  o.m[4] = x.m_ip - x.ip1 - x.m_im + x.im1;
  o.m[5] = x.m_jp - x.jp1 - x.m_jm + x.jm1;
  o.m[6] = x.pp - x.jp1 - x.pm + x.jm1;
  o.m[7] = x.p_ip - x.kp1 - x.m_ip + x.km1;
  o.m[8] = x.pp - x.ip1 - x.mp + x.im1;
  o.m[9] = x.p_jp - x.kp1 - x.m_jp + x.km1;
  o.m[10] = x.p_ip - x.ip1 - x.p_im + x.im1;
  o.m[11] = x.p_jp - x.jp1 - x.p_jm + x.jm1;
}
// ... in the two parts the coefficient groups allow: s1 and the first five terms of s2, then the other seven and the total
__device__ __forceinline__ void fv4_combine_a(double &s1, double &s2, const Br18 &r, const BG1 &g) {
  s1 = g.i00 * r.a[0];
  s1 = s1 + g.i01 * r.a[1];
  s1 = s1 + g.j00 * r.a[2];
  s1 = s1 + g.jw0 * r.a[3];
  s1 = s1 + g.k00 * r.a[4];
  s1 = s1 + g.k10 * r.a[5];
  s2 = (g.i0p - g.i0m) * r.m[0];
  s2 = s2 + (g.ip0 - g.im0) * r.m[1];
  s2 = s2 + (g.j01 - g.j0m) * r.m[2];
  s2 = s2 + (g.jp0 - g.jm0) * r.m[3];
  s2 = s2 + (g.k0p - g.k0m) * r.m[4];
}
__device__ __forceinline__ double fv4_combine_b(double s1, double s2, const Br18 &r, const BG2 &g) {
  s2 = s2 + (g.k0w - g.k0mw) * r.m[5];
  s2 = s2 + (g.i0pp - g.i0mp) * r.m[6];
  s2 = s2 + (g.ip1 - g.im1) * r.m[7];
  s2 = s2 + (g.jwp - g.jwm) * r.m[8];
  s2 = s2 + (g.jpw - g.jmw) * r.m[9];
  s2 = s2 + (g.k1p - g.k1m) * r.m[10];
  s2 = s2 + (g.k1w - g.k1mw) * r.m[11];
  return FV4_TWELFTH * s1 + (0.25 * FV4_TWELFTH) * s2;
}
__device__ __forceinline__ double fv4_combine(const Br18 &r, const B18 &b) {
  double s1 = b.f[0] * r.a[0];
  s1 = s1 + b.f[1] * r.a[1];
  s1 = s1 + b.f[2] * r.a[2];
  s1 = s1 + b.f[3] * r.a[3];
  s1 = s1 + b.f[4] * r.a[4];
  s1 = s1 + b.f[5] * r.a[5];
  double s2 = b.d[0] * r.m[0];
  s2 = s2 + b.d[1] * r.m[1];
  s2 = s2 + b.d[2] * r.m[2];
  s2 = s2 + b.d[3] * r.m[3];
  s2 = s2 + b.d[4] * r.m[4];
  s2 = s2 + b.d[5] * r.m[5];
  s2 = s2 + b.d[6] * r.m[6];
  s2 = s2 + b.d[7] * r.m[7];
  s2 = s2 + b.d[8] * r.m[8];
  s2 = s2 + b.d[9] * r.m[9];
  s2 = s2 + b.d[10] * r.m[10];
  s2 = s2 + b.d[11] * r.m[11];
  return FV4_TWELFTH * s1 + (0.25 * FV4_TWELFTH) * s2;
}
}  // namespace fv4rb
}  // namespace hpgmg

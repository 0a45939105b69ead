// device_common.hpp -- grid descriptor, indexing, WENO and TEOS-10 device functions (gfx950).
//
// Index convention on the device: 0-based interior indices.  Cell i spans faces i (west) and
// i+1 (east); likewise j (south/north) and k (bottom/top).  Halo cells have negative indices
// or indices >= N.  Arrays are Oceananigans `parent(field)` layouts: i fastest, then j, then k.
#pragma once
#include <hip/hip_runtime.h>

namespace gb25 {

struct Grid {
  int Nx, Ny, Nz, H;      // LOCAL interior size and halo
  int sx;                 // row pitch          = Nx + 2H
  int pl_c, pl_v;         // plane strides      = sx*(Ny+2H), sx*(Ny+2H+1)
  int sy_c, sy_v;         // parent y extents
  int x_periodic;         // 1: single slab, x halos are filled by local periodic copy
  float dy, g, rho0, Lz;
  // metric tables, pointers are pre-offset so that index 0 is the first interior cell/face
  const float *dxc, *dxf, *azc, *azf, *fcor, *phic;  // by j   (valid j: -H-2 .. Ny+H+2)
  const float *rdxc, *razc, *razf;                   // reciprocals (host-computed in fp64, rounded once)
  const float *zc, *dzc, *dzf, *rdzc;                // by k   (valid k: -H-2 .. Nz+H+2)
  float rdy, rLz;
  // TEOS-10 folded per level: rho'(s,t) = sum_{i+j<=6} eos[k][idx(i,j)] s^i t^j, k = 0..Nz (Nz = mirrored halo level)
  const double* eos;
  const double* dzf_d;                               // dzf in fp64 for the hydrostatic integral, by k (0..Nz)
};

// element offsets
__device__ __forceinline__ int ic(const Grid& g, int i, int j, int k) {
  return (i + g.H) + g.sx * (j + g.H) + g.pl_c * (k + g.H);
}
__device__ __forceinline__ int iv(const Grid& g, int i, int j, int k) {
  return (i + g.H) + g.sx * (j + g.H) + g.pl_v * (k + g.H);
}
__device__ __forceinline__ int i2(const Grid& g, int i, int j) { return (i + g.H) + g.sx * (j + g.H); }

__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// ---------------------------------------------------------------------------------------------
// WENO reconstruction, Oceananigans flavour: uniform coefficients, Z-weights
// alpha_s = C_s (1 + (tau/(beta_s+eps))^2), eps = 1e-8, integer-scaled smoothness indicators
// (3x the Jiang-Shu ones), written here in the factored (cancellation-free, non-negative) form.
// Arguments run from the most-upwind value `a` to the most-downwind value `e`.
// ---------------------------------------------------------------------------------------------
// Instruction-count notes (these functions are ~75 % of the tendency kernels' VALU work):
//  * the WENO5 indicators are carried as beta/0.75 = (13/3) d1^2 + d2^2 (3 instructions instead of 4); every
//    indicator, tau and eps are scaled alike, so tau/(beta+eps) and hence the weights are unchanged;
//  * the candidate polynomials are carried as 6 p_s (integer coefficients) and the 1/6 is applied once at the end.
constexpr float kWenoEps = 1e-8f;
constexpr float kWenoEps5 = 1e-8f / 0.75f;   // eps in the scaled WENO5 indicator units

__device__ __forceinline__ float beta5_0(float c, float d, float e) {
  float d1 = c - 2.f * d + e, d2 = 3.f * c - 4.f * d + e;
  return (d1 * (13.f / 3.f)) * d1 + d2 * d2;
}
__device__ __forceinline__ float beta5_1(float b, float c, float d) {
  float d1 = b - 2.f * c + d, d2 = b - d;
  return (d1 * (13.f / 3.f)) * d1 + d2 * d2;
}
__device__ __forceinline__ float beta5_2(float a, float b, float c) {
  float d1 = a - 2.f * b + c, d2 = a - 4.f * b + 3.f * c;
  return (d1 * (13.f / 3.f)) * d1 + d2 * d2;
}
// Z-weights alpha_s = C_s (1 + (tau/b_s)^2), b_s = beta_s + eps, evaluated as C_s (1 + (q rho_s)^2) with
// q = min(tau/b_min, 1e9) and rho_s = b_min/b_s <= 1.  Identical in exact arithmetic; in fp32 the plain form
// overflows (tau/b ~ 1e18 when one indicator cancels to zero next to area-weighted divergences ~1e8).
constexpr float kZCap = 1e9f;
__device__ __forceinline__ float weno5_combine(float a, float b, float c, float d, float e, float b0, float b1,
                                               float b2) {
  float p0 = 2.f * c + 5.f * d - e;            // 6 x the candidate polynomials
  float p1 = 5.f * c + 2.f * d - b;
  float p2 = 2.f * a - 7.f * b + 11.f * c;
  float tau = fabsf(b0 - b2);
  b0 += kWenoEps5;
  b1 += kWenoEps5;
  b2 += kWenoEps5;
  float bmin = fminf(b0, fminf(b1, b2));
  float qb = fminf(tau * rcp(bmin), kZCap) * bmin;
  float r0 = qb * rcp(b0), r1 = qb * rcp(b1), r2 = qb * rcp(b2);
  float a0 = 0.3f * r0 * r0 + 0.3f, a1 = 0.6f * r1 * r1 + 0.6f, a2 = 0.1f * r2 * r2 + 0.1f;
  return (a0 * p0 + a1 * p1 + a2 * p2) * (rcp(a0 + a1 + a2) * (1.f / 6.f));
}
__device__ __forceinline__ float beta3(float x, float y) {
  float d = x - y;
  return d * d;
}
__device__ __forceinline__ float weno3_combine(float b, float c, float d, float b0, float b1) {
  float p0 = c + d;                            // 2 x the candidate polynomials
  float p1 = 3.f * c - b;
  float tau = fabsf(b0 - b1);
  b0 += kWenoEps;
  b1 += kWenoEps;
  float bmin = fminf(b0, b1);
  float qb = fminf(tau * rcp(bmin), kZCap) * bmin;
  float r0 = qb * rcp(b0), r1 = qb * rcp(b1);
  float a0 = (2.f / 3.f) * r0 * r0 + (2.f / 3.f), a1 = (1.f / 3.f) * r1 * r1 + (1.f / 3.f);
  return (a0 * p0 + a1 * p1) * (rcp(a0 + a1) * 0.5f);
}

// Self-smoothness WENO5 of upwind-ordered values.
__device__ __forceinline__ float weno5(float a, float b, float c, float d, float e) {
  return weno5_combine(a, b, c, d, e, beta5_0(c, d, e), beta5_1(b, c, d), beta5_2(a, b, c));
}

// Upwind-biased reconstruction from six consecutive values q[0..5] (positions p..p+5).
// Face target f:   p = f-3.   Centre target c: p = c-2.
// left: use q[0..4]; right: use q[5..1] mirrored.  order in {5,3,1} (wall-adjacent reduction).
// s: smoothness inputs (FunctionStencil) or nullptr-equivalent (pass q); t: second smoothness
// set (VelocityStencil) averaged with s when TWO is true.
template <bool TWO>
__device__ __forceinline__ float biased6(int order, bool left, const float* q, const float* s, const float* t) {
  float c = left ? q[2] : q[3];
  if (order == 1) return c;
  float b = left ? q[1] : q[4], d = left ? q[3] : q[2];
  float sb = left ? s[1] : s[4], sc = left ? s[2] : s[3], sd = left ? s[3] : s[2];
  float tb = 0, tc = 0, td = 0;
  if (TWO) {
    tb = left ? t[1] : t[4];
    tc = left ? t[2] : t[3];
    td = left ? t[3] : t[2];
  }
  if (order == 3) {
    float b0 = beta3(sc, sd), b1 = beta3(sb, sc);
    if (TWO) {
      b0 = 0.5f * (b0 + beta3(tc, td));
      b1 = 0.5f * (b1 + beta3(tb, tc));
    }
    return weno3_combine(b, c, d, b0, b1);
  }
  float a = left ? q[0] : q[5], e = left ? q[4] : q[1];
  float sa = left ? s[0] : s[5], se = left ? s[4] : s[1];
  float b0 = beta5_0(sc, sd, se), b1 = beta5_1(sb, sc, sd), b2 = beta5_2(sa, sb, sc);
  if (TWO) {
    float ta = left ? t[0] : t[5], te = left ? t[4] : t[1];
    b0 = 0.5f * (b0 + beta5_0(tc, td, te));
    b1 = 0.5f * (b1 + beta5_1(tb, tc, td));
    b2 = 0.5f * (b2 + beta5_2(ta, tb, tc));
  }
  return weno5_combine(a, b, c, d, e, b0, b1, b2);
}

// wall-adjacent order reduction in a bounded direction of extent N (0-based target index)
__device__ __forceinline__ int biased_order_face(int f, int N) {
  return (f >= 3 && f <= N - 3) ? 5 : ((f >= 2 && f <= N - 2) ? 3 : 1);
}
__device__ __forceinline__ int biased_order_center(int c, int N) {
  return (c >= 2 && c <= N - 3) ? 5 : ((c >= 1 && c <= N - 2) ? 3 : 1);
}
__device__ __forceinline__ bool sym4_face(int f, int N) { return f >= 3 && f <= N - 3; }
__device__ __forceinline__ bool sym4_center(int c, int N) { return c >= 2 && c <= N - 3; }
// centred interpolation from four consecutive values (target sits between q1 and q2)
__device__ __forceinline__ float sym_interp(bool fourth, float q0, float q1, float q2, float q3) {
  return fourth ? (7.f * (q1 + q2) - (q0 + q3)) * (1.f / 12.f) : 0.5f * (q1 + q2);
}

// ---------------------------------------------------------------------------------------------
// TEOS-10 55-term polynomial (Roquet et al. 2015) as used by SeawaterPolynomials' TEOS10EquationOfState:
// rho(Theta, S_A, Z) = r0(zeta) + r'(tau, s, zeta), tau = Theta/40, s = sqrt((S_A+32) 0.875/35.16504),
// zeta = -Z/1e4.  The host folds the zeta dependence per model level (gb25_api.hip: build_eos_tables).
// ---------------------------------------------------------------------------------------------
// rho - rho0 at one level from the folded table: 28 coefficients ordered j-major (t-power), i ascending (s-power):
// [P_0(s): 7][P_1: 6][P_2: 5][P_3: 4][P_4: 3][P_5: 2][P_6: 1];  rho' = sum_j t^j P_j(s).  All in fp64.
__device__ __forceinline__ double teos10_level(const double* __restrict__ c, double s, double t) {
  double p0 = c[0] + s * (c[1] + s * (c[2] + s * (c[3] + s * (c[4] + s * (c[5] + s * c[6])))));
  double p1 = c[7] + s * (c[8] + s * (c[9] + s * (c[10] + s * (c[11] + s * c[12]))));
  double p2 = c[13] + s * (c[14] + s * (c[15] + s * (c[16] + s * c[17])));
  double p3 = c[18] + s * (c[19] + s * (c[20] + s * c[21]));
  double p4 = c[22] + s * (c[23] + s * c[24]);
  double p5 = c[25] + s * c[26];
  double p6 = c[27];
  return p0 + t * (p1 + t * (p2 + t * (p3 + t * (p4 + t * (p5 + t * p6)))));
}

// XCD-aware remap of a linear block id: blocks b and b+8 share an XCD (round-robin dispatch),
// so give each XCD one contiguous chunk of the logical tile sequence (bijective for any n).
__device__ __forceinline__ int xcd_remap(int b, int n) {
  int q = n >> 3, r = n & 7, x = b & 7, s = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}

}  // namespace gb25

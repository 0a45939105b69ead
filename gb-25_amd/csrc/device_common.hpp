// device_common.hpp -- grid descriptor, indexing, WENO and TEOS-10 device functions (gfx950).
//
// Index convention on the device: 0-based interior indices.  Cell i spans faces i (west) and
// i+1 (east); likewise j (south/north) and k (bottom/top).  Halo cells have negative indices
// or indices >= N.  Arrays are Oceananigans `parent(field)` layouts: i fastest, then j, then k.
#pragma once
#include <hip/hip_runtime.h>

#ifndef GB25_REAL
#define GB25_REAL float   // the model's float type; libgb25hip_f64.so is this same source built with double
#endif

namespace gb25 {

using real = GB25_REAL;
struct alignas(2 * sizeof(real)) real2 { real x, y; };
struct alignas(4 * sizeof(real)) real4 { real x, y, z, w; };

__device__ __forceinline__ float rabs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double rabs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float rmin(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double rmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float rmax(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double rmax(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float rtanh(float x) { return tanhf(x); }
__device__ __forceinline__ double rtanh(double x) { return tanh(x); }

// ImmersedBoundaryGrid(grid, GridFittedBottom(bottom_height); active_cells_map = false) -- GB-25 src/model_utils.jl:
// 134-146.  A cell (i,j,k) is active from level kc(i,j) on (kc = number of immersed cells of the column).  Everything
// the kernels need is folded per column on the host into small 2-D tables (laid out like the parent of a (c,f) field:
// pitch sx, Ny+2H+1 rows), so that no kernel ever looks at a neighbour's bottom:
//   first level from which a reconstruction stencil is fully active -- 255 = never; walls count as inactive, so the
//   wall-adjacent order reduction of the plain grid is the special case kc = 0:
//     ordA = kc | KX5 << 8 | KX3 << 16 | KY5 << 24      face target in x / y: cells i-3..i+2 (order 5), i-2..i+1 (order 3)
//     ordB = KY3 | KXC5 << 8 | KXC3 << 16 | KYC5 << 24   centre target: face nodes i-2..i+3 / i-1..i+2, a face node being
//     ordC = KYC3 | KPU << 8 | KPV << 16                  inactive when BOTH its cells are
//   KPU / KPV: first level at which the u / v face is not an immersed peripheral node (max of the two columns' kc)
//   static column depths at the faces (min of the two columns) and their reciprocals (0 where there is no depth)
struct Immersed {
  const unsigned *ordA, *ordB, *ordC;
  const unsigned* ordD;   // WENO(order = 7) tracer advection: first level from which the eight-cell stencil of the x face /
                          // the y face is fully active (bits 0-7 / 8-15); bits 16-23 / 24-31: first level from which the u face /
                          // the v face is an ACTIVE node (min of the two columns' kc: CATKE's conditional vertical differences)
  const real *Hfc, *Hcf, *rHfc, *rHcf;
};
__device__ __forceinline__ int order_from(int k, int K5, int K3) { return k >= K5 ? 5 : (k >= K3 ? 3 : 1); }
__device__ __forceinline__ int order_from7(int k, int K7, int K5, int K3) { return k >= K7 ? 7 : order_from(k, K5, K3); }

// Orthogonal curvilinear grid (the TripolarGrid of GB-25 src/model_utils.jl:134-137): horizontal metrics by location,
// 2-D arrays with the parent layout of a (c,f) field (pitch sx, Ny+2H+1 rows), Oceananigans' names (dxfc = Δxᶠᶜᵃ ...),
// reciprocals computed on the host in fp64 and rounded once, Coriolis parameter averaged to the u and v points.
// north_fold: the northern edge is the zipper fold instead of a wall; the y faces on the fold line (row Ny) are stepped.
struct Curv {
  const real *dxfc, *dxcf, *dyfc, *dycf, *azcc;
  const real *rdxfc, *rdycf, *razcc, *razfc, *razcf, *razff;
  const real *fbar_u, *fbar_v, *phicc;
  int on, north_fold;
  int pivot_slaved;   // option FOLD_PIVOT_SLAVED: the fold fill also writes the eastern half of the pivot row (image of its western half)
};

// The barotropic correction of the current step when it is applied inside the kernels that read u and v instead of by a
// sweep over them (k_corrector_2d, kernels.hpp): 2-D, parent layout of a (c,f) field; u = u_mem + du, v = v_mem + dv on
// the levels the fills write.
struct LazyCorr {
  const real *du, *dv;
  // w ON THE FLY (option W_ON_THE_FLY, only together with the above): the tendency kernels do not read w at all -- within a
  // chunk of levels they carry it up from the divergence of the very transports they hold (w(k+1) = w(k) - div(k) / Az), and
  // take w at the chunk's first level from `wbase` (k_w_bases, kernels.hpp): [chunk][parent layout of a 2-D (c,f) field]
  const real* wbase;
  int wplane;
};

struct Grid {
  int Nx, Ny, Nz, H;      // LOCAL interior size and halo
  int sx;                 // row pitch          = Nx + 2H
  int pl_c, pl_v;         // plane strides      = sx*(Ny+2H), sx*(Ny+2H+1)
  int sy_c, sy_v;         // parent y extents
  int x_periodic;         // 1: single slab, x halos are filled by local periodic copy
  // y: local row indices of the GLOBAL southern and northern wall faces.  Single domain and x slabs: 0 and Ny.  A rank of a
  // 2-D (x, y) decomposition owns the rows [j0, j0 + Ny) of the global grid: jws = -j0, jwn = Ny_global - j0, so that every
  // wall test and every wall-adjacent order reduction is the global one wherever it is evaluated (halo rows included); a side
  // without a wall gets its halo rows from the neighbour.  Folded grid: no northern wall, jwn = 1 << 20.
  int jws, jwn;
  real dy, g, rho0, Lz;
  // metric tables, pointers are pre-offset so that index 0 is the first interior cell/face
  const real *dxc, *dxf, *azc, *azf, *fcor, *phic;  // by j   (valid j: -H-2 .. Ny+H+2)
  const real *rdxc, *razc, *razf;                   // reciprocals (host-computed in fp64, rounded once)
  const real *zc, *dzc, *dzf, *rdzc, *rdzf;         // by k   (valid k: -H-2 .. Nz+H+2)
  real rdy, rLz;
  // TEOS-10 folded per level: rho'(s,t) = sum_{i+j<=6} eos[k][idx(i,j)] s^i t^j, k = 0..Nz (Nz = mirrored halo level)
  const double* eos;
  const double* eosf;                                // the same folded at the depths of the faces k = 0..Nz (CATKE: alpha, beta of N^2)
  const double* dzf_d;                               // dzf in fp64 for the hydrostatic integral, by k (0..Nz)
  Immersed im;                                       // (null pointers on a grid without bathymetry)
  // FluxBoundaryCondition at the top of u, v, T, S (null: the default no-flux): 2-D arrays with the parent layout of a
  // 2-D field of the same location; enter the tendency of the top cell as -J / dz (apply_z_top_bc!)
  const real* top_flux[4];
  // quadratic bottom drag (gb25_set_bottom_drag): the bottom flux boundary condition of u and v, 2-D like top_flux, added to
  // the tendency of the face's first free level as +J / dz (null: no drag)
  const real* bottom_flux[2];
  Curv cv;                                           // (on = 0: the LatitudeLongitudeGrid with its row tables)
};

// element offsets
__device__ __forceinline__ int ic(const Grid& g, int i, int j, int k) {
  return (i + g.H) + g.sx * (j + g.H) + g.pl_c * (k + g.H);
}
__device__ __forceinline__ int iv(const Grid& g, int i, int j, int k) {
  return (i + g.H) + g.sx * (j + g.H) + g.pl_v * (k + g.H);
}
__device__ __forceinline__ int i2(const Grid& g, int i, int j) { return (i + g.H) + g.sx * (j + g.H); }

// A wave-uniform element of a table no kernel writes (the vertical spacings): through the constant address space it is a scalar
// load (s_load_dword, lgkmcnt).  As a plain load inside a loop that also stores, clang makes it a VECTOR load -- and the wait for
// it (vmcnt counts in order, and loads issued under an exec mask may not have been issued at all) waits for every load ahead of
// it too: in the tendency kernels that was the whole batch of the next level's loads, every level.
__device__ __forceinline__ real uniform_at(const real* p, int k) {
  return *(const __attribute__((address_space(4))) real*)(p + k);
}
__device__ __forceinline__ float rfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double rfma(double a, double b, double c) { return __builtin_fma(a, b, c); }
// ab2_step_field!: phi + dt (C1 G^n - C2 G^-).  Spelled with explicit FMAs so that every kernel that advances a
// field (the stand-alone AXPY kernels and the tendency kernels that pre-advance the next step) rounds identically.
__device__ __forceinline__ real ab2_advance(real phi, real gn, real gm, real dt, real C1, real C2) {
  return rfma(dt, rfma(C1, gn, -(C2 * gm)), phi);
}
// reciprocal: the 1-ulp hardware approximation in fp32, a true division in fp64
__device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double rcp(double x) { return 1.0 / x; }

// Two-wide values: a register PAIR per lane that carries the same quantity of two independent cells / tracers.
// gfx950 issues one wave64 VALU instruction per 4 cycles per SIMD (tools/micro/valu_rate.hip: 2.0 ns per dependent
// v_fma_f32 with 2-4 waves per SIMD, i.e. 65 TFLOP/s), and v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do two lanes'
// worth of fp32 work in one such slot (112-122 TFLOP/s measured).  Arithmetic on real2v compiles to those; rcp, min,
// max, abs and selects have no packed form and cost one instruction per half.
using real2v = real __attribute__((ext_vector_type(2)));
__device__ __forceinline__ real2v v2(real a, real b) {
  real2v r;
  r.x = a;
  r.y = b;
  return r;
}
__device__ __forceinline__ real2v rcp(real2v x) { return v2(rcp(x.x), rcp(x.y)); }
__device__ __forceinline__ real2v rabs(real2v x) { return v2(rabs(x.x), rabs(x.y)); }
__device__ __forceinline__ real2v rmin(real2v a, real2v b) { return v2(rmin(a.x, b.x), rmin(a.y, b.y)); }
__device__ __forceinline__ real2v rmin(real2v a, real b) { return v2(rmin(a.x, b), rmin(a.y, b)); }
__device__ __forceinline__ real2v rmax(real2v a, real2v b) { return v2(rmax(a.x, b.x), rmax(a.y, b.y)); }

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
constexpr real kWenoEps = real(1e-8);
constexpr real kWenoEps5 = real(1e-8) / real(0.75);   // eps in the scaled WENO5 indicator units

//  * the indicators are formed from the differences of neighbours (ten instructions for the three instead of twelve, and
//    no cancellation of the values' common part), and the eps of b_s = beta_s + eps rides in their first FMA (`seed`).
template <class T>
__device__ __forceinline__ T beta5_0(T c, T d, T e, T seed) {
  T D1 = d - c, D2 = e - d;
  T d1 = D2 - D1, d2 = D2 - real(3.) * D1;
  return ((d1 * (real(13.) / real(3.))) * d1 + seed) + d2 * d2;
}
template <class T>
__device__ __forceinline__ T beta5_1(T b, T c, T d, T seed) {
  T D0 = c - b, D1 = d - c;
  T d1 = D1 - D0, d2 = D0 + D1;
  return ((d1 * (real(13.) / real(3.))) * d1 + seed) + d2 * d2;
}
template <class T>
__device__ __forceinline__ T beta5_2(T a, T b, T c, T seed) {
  T Dm = b - a, D0 = c - b;
  T d1 = D0 - Dm, d2 = real(3.) * D0 - Dm;
  return ((d1 * (real(13.) / real(3.))) * d1 + seed) + d2 * d2;
}
// Z-weights alpha_s = C_s (1 + (tau/b_s)^2), b_s = beta_s + eps, evaluated as C_s (1 + (q rho_s)^2) with
// q = min(tau/b_min, 1e9) and rho_s = b_min/b_s <= 1, i.e. q rho_s = min(tau, 1e9 b_min) / b_s.  Identical in exact
// arithmetic; in fp32 the plain form overflows (tau/b ~ 1e18 when one indicator cancels to zero next to area-weighted
// divergences ~1e8).
constexpr real kZCap = real(1e9);
// `self`: the indicators were formed from a..e themselves (a compile-time fact after inlining).  Then the result is taken
// as c + sum_s w_s (p_s - c) with 6 (p_s - c) from the neighbour differences the indicators already hold: two
// instructions fewer, and c's digits are not carried through the weighted sum.
template <class T>
__device__ __forceinline__ T weno5_combine(T a, T b, T c, T d, T e, T b0, T b1, T b2, bool self = false) {
  T tau = rabs(b0 - b2);                             // (b_s = beta_s + eps already: the eps cancels here)
  T bmin = rmin(b0, rmin(b1, b2));
  // q b_min = min(tau / b_min, 1e9) b_min = min(tau, 1e9 b_min): no reciprocal of b_min and no search for the largest
  // of the three that are needed anyway (v_rcp_f32, min and max have no packed form: one instruction per half each)
  T i0 = rcp(b0), i1 = rcp(b1), i2 = rcp(b2);
  T qb = rmin(tau, kZCap * bmin);
  T r0 = qb * i0, r1 = qb * i1, r2 = qb * i2;
  T a0 = real(0.3) * r0 * r0 + real(0.3), a1 = real(0.6) * r1 * r1 + real(0.6), a2 = real(0.1) * r2 * r2 + real(0.1);
  T w = rcp(a0 + a1 + a2) * (real(1.) / real(6.));
  if (self) {
    T Dm = b - a, D0 = c - b, D1 = d - c, D2 = e - d;
    T q0 = real(4.) * D1 - D2;                       // 6 (p_s - c)
    T q1 = real(2.) * D1 + D0;
    T q2 = real(5.) * D0 - real(2.) * Dm;
    return c + (a0 * q0 + a1 * q1 + a2 * q2) * w;
  }
  T p0 = real(2.) * c + real(5.) * d - e;            // 6 x the candidate polynomials
  T p1 = real(5.) * c + real(2.) * d - b;
  T p2 = real(2.) * a - real(7.) * b + real(11.) * c;
  return (a0 * p0 + a1 * p1 + a2 * p2) * w;
}
template <class T>
__device__ __forceinline__ T beta3(T x, T y) {
  T d = x - y;
  return d * d;
}
template <class T>
__device__ __forceinline__ T weno3_combine(T b, T c, T d, T b0, T b1) {
  T p0 = c + d;                            // 2 x the candidate polynomials
  T p1 = real(3.) * c - b;
  T tau = rabs(b0 - b1);
  b0 += kWenoEps;
  b1 += kWenoEps;
  T bmin = rmin(b0, b1);
  T i0 = rcp(b0), i1 = rcp(b1);
  T qb = rmin(tau, kZCap * bmin);
  T r0 = qb * i0, r1 = qb * i1;
  T a0 = (real(2.) / real(3.)) * r0 * r0 + (real(2.) / real(3.)), a1 = (real(1.) / real(3.)) * r1 * r1 + (real(1.) / real(3.));
  return (a0 * p0 + a1 * p1) * (rcp(a0 + a1) * real(0.5));
}

// Self-smoothness WENO5 of upwind-ordered values.
template <class T>
__device__ __forceinline__ T weno5(T a, T b, T c, T d, T e) {
  return weno5_combine(a, b, c, d, e, beta5_0(c, d, e, T(kWenoEps5)), beta5_1(b, c, d, T(kWenoEps5)), beta5_2(a, b, c, T(kWenoEps5)), true);
}

// Upwind-biased reconstruction from six consecutive values q[0..5] (positions p..p+5).
// Face target f:   p = f-3.   Centre target c: p = c-2.
// left: use q[0..4]; right: use q[5..1] mirrored.  order in {5,3,1} (wall-adjacent reduction).
// s: smoothness inputs (FunctionStencil) or nullptr-equivalent (pass q); t: second smoothness
// set (VelocityStencil) averaged with s when TWO is true.
template <bool TWO, class T = real>
__device__ __forceinline__ T biased6(int order, bool left, const T* q, const T* s, const T* t) {
  T c = left ? q[2] : q[3];
  if (order == 1) return c;
  T b = left ? q[1] : q[4], d = left ? q[3] : q[2];
  T sb = left ? s[1] : s[4], sc = left ? s[2] : s[3], sd = left ? s[3] : s[2];
  T tb = T(0), tc = T(0), td = T(0);
  if (TWO) {
    tb = left ? t[1] : t[4];
    tc = left ? t[2] : t[3];
    td = left ? t[3] : t[2];
  }
  if (order == 3) {
    T b0 = beta3(sc, sd), b1 = beta3(sb, sc);
    if (TWO) {
      b0 = real(0.5) * (b0 + beta3(tc, td));
      b1 = real(0.5) * (b1 + beta3(tb, tc));
    }
    return weno3_combine(b, c, d, b0, b1);
  }
  T a = left ? q[0] : q[5], e = left ? q[4] : q[1];
  T sa = left ? s[0] : s[5], se = left ? s[4] : s[1];
  // TWO: the SUM of the two indicator sets with 2 eps (every b_s, tau and the cap scale alike: the weights are those of
  // the average with eps), the second set accumulated onto the first inside its FMAs
  const T seed = T(TWO ? real(2.) * kWenoEps5 : kWenoEps5);
  T b0 = beta5_0(sc, sd, se, seed), b1 = beta5_1(sb, sc, sd, seed), b2 = beta5_2(sa, sb, sc, seed);
  if (TWO) {
    T ta = left ? t[0] : t[5], te = left ? t[4] : t[1];
    b0 = beta5_0(tc, td, te, b0);
    b1 = beta5_1(tb, tc, td, b1);
    b2 = beta5_2(ta, tb, tc, b2);
  }
  return weno5_combine(a, b, c, d, e, b0, b1, b2, !TWO && q == s);
}

// WENO(order = 7), self-smoothness (tracer fluxes of ClimaOcean's ocean_simulation): candidate polynomials, linear weights and
// smoothness indicators of Balsara & Shu (2000), Z-weights with tau_7 = |b0 + 3 b1 - 3 b2 - b3| (oracle: weno7_combine).
// a..g: upwind-most .. downwind-most, the face between d and e.  The indicators are translation invariant and are
// evaluated on the values minus d: in Float32 the expanded quadratic forms (coefficients up to 17246) would otherwise
// cancel eight digits of T^2 ~ 1e3.
template <class T>
__device__ __forceinline__ T weno7(T a, T b, T c, T d, T e, T f, T g) {
  const T p0 = real(3.) * d + real(13.) * e - real(5.) * f + g;          // 12 x the candidate polynomials
  const T p1 = real(7.) * (d + e) - (c + f);
  const T p2 = b - real(5.) * c + real(13.) * d + real(3.) * e;
  const T p3 = real(25.) * d - real(23.) * c + real(13.) * b - real(3.) * a;
  const T A = a - d, B = b - d, C = c - d, E = e - d, F = f - d, G = g - d;   // (D = 0)
  T b0 = E * (real(11003.) * E - real(17246.) * F + real(4642.) * G) + F * (real(7043.) * F - real(3882.) * G) + real(547.) * G * G;
  T b1 = C * (real(547.) * C + real(1922.) * E - real(494.) * F) + E * (real(2843.) * E - real(1642.) * F) + real(267.) * F * F;
  T b2 = B * (real(267.) * B - real(1642.) * C - real(494.) * E) + C * (real(2843.) * C + real(1922.) * E) + real(547.) * E * E;
  T b3 = A * (real(547.) * A - real(3882.) * B + real(4642.) * C) + B * (real(7043.) * B - real(17246.) * C) + real(11003.) * C * C;
  const T tau = rabs(b0 + real(3.) * b1 - real(3.) * b2 - b3);
  b0 = rmax(b0, T(real(0.))) + kWenoEps; b1 = rmax(b1, T(real(0.))) + kWenoEps;
  b2 = rmax(b2, T(real(0.))) + kWenoEps; b3 = rmax(b3, T(real(0.))) + kWenoEps;
  const T bmin = rmin(rmin(b0, b1), rmin(b2, b3));
  const T i0 = rcp(b0), i1 = rcp(b1), i2 = rcp(b2), i3 = rcp(b3);
  const T qb = rmin(tau, kZCap * bmin);
  const T r0 = qb * i0, r1 = qb * i1, r2 = qb * i2, r3 = qb * i3;
  const T a0 = real(4. / 35.) * r0 * r0 + real(4. / 35.), a1 = real(18. / 35.) * r1 * r1 + real(18. / 35.);
  const T a2 = real(12. / 35.) * r2 * r2 + real(12. / 35.), a3 = real(1. / 35.) * r3 * r3 + real(1. / 35.);
  return (a0 * p0 + a1 * p1 + a2 * p2 + a3 * p3) * (rcp(a0 + a1 + a2 + a3) * (real(1.) / real(12.)));
}
// Upwind-biased reconstruction to a face from eight consecutive values w[0..7] (cells f-4 .. f+3); order in {7, 5, 3, 1}: the
// lower orders are those of biased6 on the inner six.
template <class T = real>
__device__ __forceinline__ T biased8(int order, bool left, const T* w) {
  if (order != 7) return biased6<false, T>(order, left, w + 1, w + 1, w + 1);
  return left ? weno7(w[0], w[1], w[2], w[3], w[4], w[5], w[6]) : weno7(w[7], w[6], w[5], w[4], w[3], w[2], w[1]);
}

// Two reconstructions of the same order at once, one per half of the pair, each with its own upwind direction
// (l0 for .x, l1 for .y).  Used where two different quantities of one cell share stencil shape and order.
__device__ __forceinline__ real2v pick(bool l0, bool l1, real2v a, real2v b) { return v2(l0 ? a.x : b.x, l1 ? a.y : b.y); }
template <bool TWO>
__device__ __forceinline__ real2v biased6p(int order, bool l0, bool l1, const real2v* q, const real2v* s,
                                           const real2v* t) {
  real2v c = pick(l0, l1, q[2], q[3]);
  if (order == 1) return c;
  real2v b = pick(l0, l1, q[1], q[4]), d = pick(l0, l1, q[3], q[2]);
  real2v sb = pick(l0, l1, s[1], s[4]), sc = pick(l0, l1, s[2], s[3]), sd = pick(l0, l1, s[3], s[2]);
  real2v tb = real2v(0), tc = real2v(0), td = real2v(0);
  if (TWO) {
    tb = pick(l0, l1, t[1], t[4]);
    tc = pick(l0, l1, t[2], t[3]);
    td = pick(l0, l1, t[3], t[2]);
  }
  if (order == 3) {
    real2v b0 = beta3(sc, sd), b1 = beta3(sb, sc);
    if (TWO) {
      b0 = real(0.5) * (b0 + beta3(tc, td));
      b1 = real(0.5) * (b1 + beta3(tb, tc));
    }
    return weno3_combine(b, c, d, b0, b1);
  }
  real2v a = pick(l0, l1, q[0], q[5]), e = pick(l0, l1, q[4], q[1]);
  real2v sa = pick(l0, l1, s[0], s[5]), se = pick(l0, l1, s[4], s[1]);
  // TWO: the SUM of the two indicator sets with 2 eps (every b_s, tau and the cap scale alike: the weights are those of
  // the average with eps), the second set accumulated onto the first inside its FMAs
  const real2v seed = real2v(TWO ? real(2.) * kWenoEps5 : kWenoEps5);
  real2v b0 = beta5_0(sc, sd, se, seed), b1 = beta5_1(sb, sc, sd, seed), b2 = beta5_2(sa, sb, sc, seed);
  if (TWO) {
    real2v ta = pick(l0, l1, t[0], t[5]), te = pick(l0, l1, t[4], t[1]);
    b0 = beta5_0(tc, td, te, b0);
    b1 = beta5_1(tb, tc, td, b1);
    b2 = beta5_2(ta, tb, tc, b2);
  }
  return weno5_combine(a, b, c, d, e, b0, b1, b2, !TWO && q == s);
}

// ---------------------------------------------------------------------------------------------
// Buffer addressing.  A stencil kernel reads one array at many (i,j,k) offsets from the same cell.  With flat/global
// loads every such access carries its own 64-bit per-lane address (one v_lshl_add_u64 and a VGPR pair each: 9 % of the
// tracer kernel's VALU instructions and ~25 of its VGPRs).  A buffer access is  base(SGPR x4) + voffset(one VGPR,
// shared by every access of the cell) + soffset(SGPR: the wave-uniform j/k displacement) + immediate(the i
// displacement): no address arithmetic on the vector side at all.  Offsets are BYTES and must stay non-negative and
// below 2^31: kernels bias voffset to the (-3,-3,-3) corner of the stencil.  Reads past `bytes` return zero.
// ---------------------------------------------------------------------------------------------
struct Buf {
  __amdgpu_buffer_rsrc_t r;
};
__device__ __forceinline__ Buf make_buf(const real* p, long elems) {
  Buf b;
  // (num_records is a 32-bit byte count; a view never needs more than the planes a block touches)
  const long bytes = elems * (long)sizeof(real);
  b.r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)(bytes < 0x7fffffffL ? bytes : 0x7fffffffL), 0x00020000);
  return b;
}
__device__ __forceinline__ float bload_(const Buf& b, int voff, int soff, float) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(b.r, voff, soff, 0));
}
__device__ __forceinline__ double bload_(const Buf& b, int voff, int soff, double) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(b.r, voff, soff, 0));
}
__device__ __forceinline__ real bload(const Buf& b, int voff, int soff) { return bload_(b, voff, soff, real(0)); }
__device__ __forceinline__ void bstore(const Buf& b, int voff, int soff, float x) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x), b.r, voff, soff, 0);
}
__device__ __forceinline__ void bstore(const Buf& b, int voff, int soff, double x) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, x), b.r, voff, soff, 0);
}

// Two reconstructions to a face at once, one per half of the pair, each with its own upwind direction AND its own order
// (two different columns of one field: k_tracer_tendencies_single): the same order in both halves -- the usual case -- is one
// packed evaluation; different orders (next to bathymetry) evaluate both and keep a half of each.
template <int ORD>
__device__ __forceinline__ real2v biased_pair_same(int order, bool l0, bool l1, const real2v* w) {
  if (ORD == 7) {
    if (order == 7)
      return weno7(pick(l0, l1, w[0], w[7]), pick(l0, l1, w[1], w[6]), pick(l0, l1, w[2], w[5]), pick(l0, l1, w[3], w[4]),
                   pick(l0, l1, w[4], w[3]), pick(l0, l1, w[5], w[2]), pick(l0, l1, w[6], w[1]));
    return biased6p<false>(order, l0, l1, w + 1, w + 1, w + 1);
  }
  return biased6p<false>(order, l0, l1, w, w, w);
}
template <int ORD>
__device__ __forceinline__ real2v biased_pair(int o0, int o1, bool l0, bool l1, const real2v* w) {
  // ONE instance of the evaluation, run once or twice: a second inlined copy of which only one half is used gets
  // scalarised by the compiler, with its own choice of fused multiply-adds -- and a column's result then depended, in
  // the last bit, on which column it happened to be paired with (slabs against the single domain).
  real2v out = real2v(real(0.));
  const int n = o0 == o1 ? 1 : 2;
#pragma nounroll
  for (int h = 0; h < n; h++) {
    const real2v r = biased_pair_same<ORD>(h ? o1 : o0, l0, l1, w);
    if (h == 0) out = r;
    else out.y = r.y;
  }
  return out;
}

// wall-adjacent order reduction in a bounded direction of extent N (0-based target index)
__device__ __forceinline__ int biased_order_face(int f, int N) {
  return (f >= 3 && f <= N - 3) ? 5 : ((f >= 2 && f <= N - 2) ? 3 : 1);
}
__device__ __forceinline__ int biased_order_face7(int f, int N) { return (f >= 4 && f <= N - 4) ? 7 : biased_order_face(f, N); }
__device__ __forceinline__ int biased_order_center(int c, int N) {
  return (c >= 2 && c <= N - 3) ? 5 : ((c >= 1 && c <= N - 2) ? 3 : 1);
}
__device__ __forceinline__ bool sym4_face(int f, int N) { return f >= 3 && f <= N - 3; }
__device__ __forceinline__ bool sym4_center(int c, int N) { return c >= 2 && c <= N - 3; }
// centred interpolation from four consecutive values (target sits between q1 and q2)
__device__ __forceinline__ real sym_interp(bool fourth, real q0, real q1, real q2, real q3) {
  return fourth ? (real(7.) * (q1 + q2) - (q0 + q3)) * (real(1.) / real(12.)) : real(0.5) * (q1 + q2);
}

// ---------------------------------------------------------------------------------------------
// TEOS-10 55-term polynomial (Roquet et al. 2015) as used by SeawaterPolynomials' TEOS10EquationOfState:
// rho(Theta, S_A, Z) = r0(zeta) + r'(tau, s, zeta), tau = Theta/40, s = sqrt((S_A+32) 0.875/35.16504),
// zeta = -Z/1e4.  The host folds the zeta dependence per model level (gb25_api.hip: build_eos_tables).
// ---------------------------------------------------------------------------------------------
// rho - rho0 at one level from the folded table: 28 coefficients ordered j-major (t-power), i ascending (s-power):
// [P_0(s): 7][P_1: 6][P_2: 5][P_3: 4][P_4: 3][P_5: 2][P_6: 1];  rho' = sum_j t^j P_j(s).  All in fp64.
// sqrt of a positive, normal double: the seed and the Newton sequence of the compiler's own expansion of sqrt(double) (so the
// same bits) without its range scaling (two v_ldexp_f64) and its selects for 0 / inf / denormals -- a third of the expansion,
// 50 of the 421 instructions of a level of the pressure kernel (five evaluations), whose time is its dependent fp64 chains.  The argument of
// the equation of state, (S + 32) * 0.0249 with S in g/kg, is of order one.
__device__ __forceinline__ double sqrt_pos(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
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

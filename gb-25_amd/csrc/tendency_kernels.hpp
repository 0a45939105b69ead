// tendency_kernels.hpp -- the shipped tendency kernels (gfx950): momentum ("v5", LDS-staged tiles, packed fp32
// reconstructions, AB2 look-ahead of u, v) and tracers ("v5", wave-autonomous, T and S as one two-wide value,
// buffer addressing, AB2 look-ahead of T, S).  Each face flux / derived quantity is computed ONCE per (i,j,k) and
// shared through LDS (x, y), a wave shuffle (x faces of the tracers) or carried in registers while marching up the
// column (z).  The direct-stencil kernels of kernels.hpp (k_gu, k_gv, k_tracer_tendencies) evaluate the same
// expressions cell by cell and are kept as the cross-check generation (option "kernels" = 1).
//
// A wave64 VALU instruction costs ~4 cycles of a SIMD on gfx950 and scalar forms of these kernels were
// issue-saturated (VALUBusy ~100 %): see device_common.hpp (real2v), tools/micro/ and profiles/r01_tuning_log.md.
#pragma once
#include <type_traits>

#include "device_common.hpp"

namespace gb25 {

constexpr int V2_TX = 64;   // tile width = one wavefront; tile height TY (rows = waves per block) is a template parameter

// =============================================================================================
// Momentum tendencies G_u and G_v fused: block = (64 x 8) columns marching in k.  Per level the block stages
// the u, v (level k) and w (level k+1) tiles in LDS, derives vorticity zeta and its VelocityStencil inputs at the
// (f,f,c) points and the area-weighted divergence pieces at the (c,c,c) points ONCE per point (v1 recomputed each
// of them for 6-12 neighbouring cells), then every thread evaluates G_u and G_v of its cell from LDS.  The vertical
// momentum fluxes are carried from level to level.  Arithmetic per term is that of k_gu / k_gv (kernels.hpp).
// =============================================================================================
constexpr int MU_X = V2_TX + 6;   // u, v tiles: origin (i0-3, j0-3), TY+6 rows
constexpr int MW_X = V2_TX + 3;   // w tile:     origin (i0-2, j0-2), TY+3 rows
constexpr int MD_X = V2_TX + 5;   // derived tiles: ffc origin (i0-2, j0-2), ccc origin (i0-3, j0-3), TY+5 rows

template <int V2_TY>
struct MomentumLds {
  static constexpr int MU_Y = V2_TY + 6, MW_Y = V2_TY + 3, MD_Y = V2_TY + 5;
  real U[2][MU_Y][MU_X];
  real V[2][MU_Y][MU_X];
  real W[2][MW_Y][MW_X];
  real Z[MD_Y][MD_X], UQ[MD_Y][MD_X], VQ[MD_Y][MD_X];  // (f,f,c)
  real DU[MD_Y][MD_X], DV[MD_Y][MD_X];                 // (c,c,c)
};
// Orthogonal curvilinear grid: the face lengths every derived point needs, staged once per block in the geometry of the
// u / v tiles (origin (i0-3, j0-3)), and 1 / Az^ffc in the geometry of the (f,f,c) tile (origin (i0-2, j0-2)).
template <int V2_TY>
struct MomentumMetricLds {
  static constexpr int MU_Y = V2_TY + 6, MD_Y = V2_TY + 5;
  real dxfc[MU_Y][MU_X], dxcf[MU_Y][MU_X], dyfc[MU_Y - 1][MU_X], dycf[MU_Y][MU_X];   // (the last row of dyfc is never read)
  real razff[MD_Y][MD_X];
};
struct NoLds { char unused; };
// LAZY: the barotropic corrections of the u / v tile (k-independent), staged once per block
template <int V2_TY>
struct MomentumCorrLds {
  real du[V2_TY + 6][MU_X], dv[V2_TY + 6][MU_X];
};

// =============================================================================================
// Momentum tendencies, packed evaluation: the eight WENO reconstructions of a cell are evaluated as four two-wide ones (a G_u term paired with the G_v term of
// the same stencil shape and order), so their arithmetic issues as v_pk_fma/mul/add_f32.  Rows next to the walls,
// where the vorticity reconstruction of G_u drops below order 5 but G_v's does not, take the scalar form for that
// one pair.
// =============================================================================================
// AHEAD: the kernel also performs the NEXT step's ab2_step_field! of u and v into partner arrays (un, vn) and leaves
// the per-chunk column sums of (C1 G^n - C2 G^-) dz and of un dz, vn dz in P; k_ab2_velocities_finish adds the chunks
// up into G.U, G.V and the corrector's column integrals.  u, v and the fresh tendencies are in registers here, so the
// separate 6R + 2W sweep of k_ab2_velocities shrinks to one read (G^-) and one write per component.
// Which tile columns a launch covers.  The x axis is cut into columns of tiles bx = 0 .. ceil(Nx/64)-1; a launch owns
// `n` of them: the first `nlead` are bx0, bx0+1, ..., the rest bx1, bx1+1, ...  Whole domain: {n, n, 0, 0}.  A slab of
// a decomposition launches its interior tile columns (which read own columns only) while the x-halo bundle is still
// travelling, and the edge columns {0} + {last} afterwards (interior_tendency_kernel_parameters +
// complete_communication_and_compute_buffer!, GB-25 src/precompile.jl:67,72): same tiles, same arithmetic, same bits.
// ... and which chunks of levels: [first, first + count) of the kchunks chunks, with the array pointers rebased by kofs planes
// (arrays beyond 4 GB take two or more launches; one launch with kofs = 0 otherwise).
struct ChunkRange {
  int first, count, kofs;
};
struct TileCols {
  int n, nlead, bx0, bx1;
  ChunkRange cr;
};
__device__ __forceinline__ int tile_column(const TileCols& tc, int q) { return q < tc.nlead ? tc.bx0 + q : tc.bx1 + (q - tc.nlead); }

struct UvAhead {
  const real *GmU, *GmV;
  real *un, *vn, *P;
  real dt, C1, C2;
  int plane2;
  // single periodic domain: un, vn are written with the halo cells tupled_fill_halo_regions! derives from them (periodic
  // x images, the y layer of u, zero on the wall faces of v, the bottom / top layers and their x images), so that the
  // adopted buffers need no fill launch whatever the corrector does afterwards
  int fold;
};

// IMM: immersed boundary.  Orders and the 4th/2nd-order switches of the centred interpolations come per lane and per
// level from the folded tables; the tendencies of faces that touch the solid are zero (their velocities are masked and
// stay so).  The pairs that share direction and target keep their packed evaluation; the vorticity pair (y for G_u, x
// for G_v) is packed where both orders are 5 and evaluated one by one elsewhere, as next to the walls.
// CURV: orthogonal curvilinear grid (always with the tables, IMM): every face length, area and the Coriolis parameter per
// point -- k-independent, so the lengths of the tile sit in LDS (MomentumMetricLds), the own cell's reciprocals in
// registers, and Az^ccc is multiplied into the w tile when it is staged.  With the zipper fold the tiles cover one more
// row: the y faces ON the fold line have a G_v (and, AHEAD, a v of the next step) like any other row.
// LAZY: u, v in memory lack this step's barotropic correction (k_corrector_2d, kernels.hpp): it is added to every value
// as it is loaded -- tile elements (their du, dv are k-independent: registers) and the own column's vertical window.
// DRAG: the quadratic bottom drag's flux boundary condition (Grid.bottom_flux) enters the first free level (an instance of its
// own: the hook cost the default instances a few spilled registers)
// WFLY: w is not read -- the thread that derives the divergence pieces DU, DV of a (c,c,c) point in phase 1 also
// carries that point's w up the chunk, w(k+1) = w(k) - (DU + DV) / Az, and puts it into the w tile; w at the chunk's first
// level comes from lz.wbase (k_w_bases).  Saves the 4.6 B per cell of the w tile and, with the tracer kernel doing the same,
// the whole k_compute_w launch of a step.  With or without LAZY (round 4: the grids with a bottom and the curvilinear ones keep the
// corrector's sweep but not the w launch); CURV: the tile holds Az w, so Az w is what is carried: Az w(k+1) = Az w(k) - (DU + DV).
template <int MINW, int V2_TY, bool AHEAD, bool IMM, bool CURV = false, bool LAZY = false, bool DRAG = false, bool WFLY = false>
__global__ __launch_bounds__(V2_TX* V2_TY, MINW) void k_momentum_tendencies_v5(
    Grid g, const real* __restrict__ u, const real* __restrict__ v, const real* __restrict__ w,
    const real* __restrict__ dpx, const real* __restrict__ dpy, real* __restrict__ Gu, real* __restrict__ Gv,
    TileCols tc, int kchunks, int nb, UvAhead next, LazyCorr lz) {
  static_assert(!CURV || IMM, "the curvilinear variant takes its orders from the tables");
  static_assert(!LAZY || (!IMM && !CURV), "the corrector is applied inside its consumers on the flat lat-lon grid only");
  __shared__ MomentumLds<V2_TY> lds;
  __shared__ typename std::conditional<CURV, MomentumMetricLds<V2_TY>, NoLds>::type mt;
  __shared__ typename std::conditional<LAZY, MomentumCorrLds<V2_TY>, NoLds>::type cr;
  constexpr int MU_Y = V2_TY + 6, MW_Y = V2_TY + 3, MD_Y = V2_TY + 5;
  const int L = xcd_remap(blockIdx.x, nb);
  const int r = L / tc.n, bx = tile_column(tc, L - r * tc.n);
  const ChunkRange cr_ = tc.cr;                       // (the chunks of levels this launch covers: all of them, normally)
  const int kc = cr_.first + r % cr_.count, by = r / cr_.count;
  const int klen = (g.Nz + kchunks - 1) / kchunks;
  const int k0 = kc * klen, k1 = min(g.Nz, k0 + klen);
  const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * V2_TX + tx;
  const int i0 = bx * V2_TX, j0 = by * V2_TY;
  const int i = i0 + tx, j = j0 + ty;
  // rows of y faces that have a tendency: with the zipper fold the fold line (row Ny) is one of them
  const int jv_last = CURV ? g.Ny - 1 : g.Ny - 1;
  const bool inside_u = (i < g.Nx) && (j < g.Ny), inside_v = (i < g.Nx) && (j <= jv_last);
  const int ic_ = min(i, g.Nx - 1), jc_ = min(j, jv_last);   // ragged tiles: threads past the edge work on a duplicate
  const int sx = g.sx, pc = g.pl_c, pv = g.pl_v, H = g.H;
  const real dy = g.dy;

  // metrics of this thread's row (lat-lon) or cell (curvilinear: reciprocals at the u and the v point, 1/Az^fcc and
  // 1/Az^cfc for the vertical advection, the Coriolis parameter averaged to the two points)
  const int om = CURV ? i2(g, ic_, jc_) : 0;
  const real dxf_s = CURV ? real(0.) : g.dxf[j], dxf_n = CURV ? real(0.) : g.dxf[j + 1];
  const real rdxc_j = CURV ? g.cv.rdxfc[om] : g.rdxc[j], rdy_j = CURV ? g.cv.rdycf[om] : g.rdy;
  const real razc_j = CURV ? g.cv.razfc[om] : g.razc[j], razf_j = CURV ? g.cv.razcf[om] : g.razf[j];
  const real fcor_j = CURV ? g.cv.fbar_v[om] : g.fcor[j];
  const real fbar = CURV ? g.cv.fbar_u[om] : real(0.5) * (g.fcor[j] + g.fcor[j + 1]);
  // Az^ccc of the cell and its neighbours in x and y (the advecting Az w of the vertical momentum flux); the curvilinear
  // tiles carry Az w already, so the factors are needed for the flux through the chunk's bottom face only
  const real Az = CURV ? g.cv.azcc[om] : g.azc[j];
  const real az_m2 = CURV ? g.cv.azcc[om - 2 * sx] : g.azc[j - 2], az_m1 = CURV ? g.cv.azcc[om - sx] : g.azc[j - 1],
             az_p1 = CURV ? g.cv.azcc[om + sx] : g.azc[j + 1];
  // (rows counted from the global southern wall: a rank of a 2-D decomposition reduces orders where the single domain does)
  const int jw_ = j - g.jws, Nyw_ = g.jwn - g.jws;
  int oc_y = biased_order_center(jw_, Nyw_), of_y = biased_order_face(jw_, Nyw_);
  bool s4c_y = sym4_center(jw_, Nyw_), s4f_y = sym4_face(jw_, Nyw_);
  // orders / switches that are constants of the plain grid (x is periodic) and per-level quantities with a bottom
  int oc_x = 5, of_x = 5;
  bool s4c_x = true, s4f_x = true, s4f_xw = true, s4f_yw = s4f_y;
  int kbt = 0, KX5 = 0, KX3 = 0, KY5 = 0, KY3 = 0, KXC5 = 0, KXC3 = 0, KYC5 = 0, KYC3 = 0, KPU = 0, KPV = 0;
  if (IMM) {
    const int o2 = i2(g, ic_, jc_);
    const unsigned A = g.im.ordA[o2], B = g.im.ordB[o2], C = g.im.ordC[o2];
    kbt = A & 255; KX5 = (A >> 8) & 255; KX3 = (A >> 16) & 255; KY5 = A >> 24;
    KY3 = B & 255; KXC5 = (B >> 8) & 255; KXC3 = (B >> 16) & 255; KYC5 = B >> 24;
    KYC3 = C & 255; KPU = (C >> 8) & 255; KPV = (C >> 16) & 255;
  }
  const int Nzc = g.Nz - kbt;
  // per-level orders: plain grid -> nothing to do; immersed -> from the thresholds.  `kw` is the level whose activity
  // the horizontal stencil of Az w on the top face k+1 sees (the row above the face, the top face that of level Nz-1)
  auto level_orders = [&](int k) {
    if (!IMM) return;
    oc_y = order_from(k, KYC5, KYC3); of_y = order_from(k, KY5, KY3);
    oc_x = order_from(k, KXC5, KXC3); of_x = order_from(k, KX5, KX3);
    s4c_y = k >= KYC5; s4f_y = k >= KY5; s4c_x = k >= KXC5; s4f_x = k >= KX5;
    const int kw = min(k + 1, g.Nz - 1);
    s4f_xw = kw >= KX5; s4f_yw = kw >= KY5;
  };

  // threads of a ragged edge tile work on a clamped (duplicate) column so that every address stays in bounds
  // Arrays beyond 4 GB (config 5 as one domain: 4.4 GB per field): the per-lane offsets are 32-bit BYTE offsets, so the host
  // hands such a launch array pointers rebased by `kofs` planes and a sub-range of the chunks of levels that stays within
  // reach of them (momentum_impl); plane indices below count from the rebased pointers.  kofs = 0 otherwise.
  int o = ic(g, ic_, jc_, k0 - cr_.kofs), ov = iv(g, ic_, jc_, k0 - cr_.kofs);
  // (uniform base pointer + 32-bit per-lane byte offset: the accesses take the scalar-base addressing form and need no
  // 64-bit address arithmetic per lane; the own cell's byte offsets ob / obv serve every centre- / v-shaped array)
  auto at = [](const real* base, unsigned byte_off) {
    return *reinterpret_cast<const real*>(reinterpret_cast<const char*>(base) + byte_off);
  };
  auto put = [](real* base, unsigned byte_off, real x) {
    *reinterpret_cast<real*>(reinterpret_cast<char*>(base) + byte_off) = x;
  };
  constexpr unsigned SZ = (unsigned)sizeof(real);
  unsigned ob = (unsigned)o * SZ, obv = (unsigned)ov * SZ;
  real uz[7], vz[7];
#pragma unroll
  for (int m = 0; m < 7; m++) {
    uz[m] = u[o + (m - 3) * pc];
    vz[m] = v[ov + (m - 3) * pv];
  }
  if constexpr (LAZY) {   // the own column's correction (levels the fills write: deeper ones are never used)
    const int o2 = i2(g, ic_, jc_);
    const real du_o = lz.du[o2], dv_o = lz.dv[o2];
#pragma unroll
    for (int m = 0; m < 7; m++)
      if (k0 + m - 3 >= -1 && k0 + m - 3 <= g.Nz) {
        uz[m] = uz[m] + du_o;
        vz[m] = vz[m] + dv_o;
      }
  }
  // vertical momentum fluxes through the bottom face of the first level
  real fzu, fzv;
  {
    const int ord = biased_order_face(k0 - kbt, Nzc);
    // (the bottom face k0 of the chunk is the top face of level k0-1; face 0 carries w = 0 whatever the order)
    if (IMM) level_orders(max(k0 - 1, 0));
    const real ax_m2 = CURV ? g.cv.azcc[om - 2] : Az, ax_m1 = CURV ? g.cv.azcc[om - 1] : Az, ax_p1 = CURV ? g.cv.azcc[om + 1] : Az;
    // (WFLY: w of the chunk's first level from the 2-D bases; face 0 carries w = 0)
    const real* w0 = WFLY ? lz.wbase + (long)kc * lz.wplane + i2(g, ic_, jc_) : w + o;
    real wu = sym_interp(s4f_xw, ax_m2 * w0[-2], ax_m1 * w0[-1], Az * w0[0], ax_p1 * w0[1]);
    real wv = sym_interp(s4f_yw, az_m2 * w0[-2 * sx], az_m1 * w0[-sx], Az * w0[0], az_p1 * w0[sx]);
    fzu = wu * biased6<false>(ord, wu > real(0.), uz, uz, uz);
    fzv = wv * biased6<false>(ord, wv > real(0.), vz, vz, vz);
  }

  // Tile staging is software-pipelined: the global loads of level k+1 are issued before the arithmetic of level
  // k and land in LDS (other parity) after it, so their latency hides behind phases 1-2 instead of in front of a
  // barrier.  Each thread owns up to 2 elements of the u / v tiles and 2 of the w tile.
  constexpr int NT = V2_TX * V2_TY;
  constexpr int NEU = (MU_X * MU_Y + NT - 1) / NT, NEW = (MW_X * MW_Y + NT - 1) / NT;   // elements per thread
  const int tile_u = (i0 - 3 + H) + sx * (j0 - 3 + H), tile_w = (i0 - 2 + H) + sx * (j0 - 2 + H);
  int eu_off[NEU], eu_lds[NEU], ew_off[NEW], ew_lds[NEW];   // global offset within the tile plane / LDS index
#pragma unroll
  for (int q = 0; q < NEU; q++) {
    int e = tid + q * NT;
    int ey = e / MU_X, ex = e - ey * MU_X;
    // clamp to the parent array (ragged tiles): columns <= Nx+H-1, rows <= Ny+H-1 relative to the tile origin
    eu_off[q] = (e < MU_X * MU_Y) ? (min(ex, g.Nx + H + 2 - i0) + sx * min(ey, g.Ny + H + 2 - j0)) * (int)sizeof(real) : -1;
    eu_lds[q] = e;
  }
#pragma unroll
  for (int q = 0; q < NEW; q++) {
    int e = tid + q * NT;
    int ey = e / MW_X, ex = e - ey * MW_X;
    ew_off[q] = (e < MW_X * MW_Y) ? (min(ex, g.Nx + H + 1 - i0) + sx * min(ey, g.Ny + H + 1 - j0)) * (int)sizeof(real) : -1;
    ew_lds[q] = e;
  }
  real ru[NEU], rv[NEU], rw[NEW], rpw = real(0.), rps = real(0.);
  if constexpr (LAZY) {   // the corrections of the tile, in LDS (registers are what this kernel is short of)
#pragma unroll
    for (int q = 0; q < NEU; q++)
      if (eu_off[q] >= 0) {
        (&cr.du[0][0])[eu_lds[q]] = *reinterpret_cast<const real*>(reinterpret_cast<const char*>(lz.du + tile_u) + eu_off[q]);
        (&cr.dv[0][0])[eu_lds[q]] = *reinterpret_cast<const real*>(reinterpret_cast<const char*>(lz.dv + tile_u) + eu_off[q]);
      }
  }
  real azw[NEW];   // CURV: Az^ccc at this thread's elements of the w tile
  if constexpr (CURV) {
    const real* ab = g.cv.azcc + tile_w;
#pragma unroll
    for (int q = 0; q < NEW; q++) azw[q] = ew_off[q] >= 0 ? *reinterpret_cast<const real*>(reinterpret_cast<const char*>(ab) + ew_off[q]) : real(0.);
    const real *m0 = g.cv.dxfc + tile_u, *m1 = g.cv.dxcf + tile_u, *m2 = g.cv.dyfc + tile_u, *m3 = g.cv.dycf + tile_u;
#pragma unroll
    for (int q = 0; q < NEU; q++)
      if (eu_off[q] >= 0) {
        auto ld = [&](const real* b_) { return *reinterpret_cast<const real*>(reinterpret_cast<const char*>(b_) + eu_off[q]); };
        (&mt.dxfc[0][0])[eu_lds[q]] = ld(m0);
        (&mt.dxcf[0][0])[eu_lds[q]] = ld(m1);
        if (eu_lds[q] < (MU_Y - 1) * MU_X) (&mt.dyfc[0][0])[eu_lds[q]] = ld(m2);
        (&mt.dycf[0][0])[eu_lds[q]] = ld(m3);
      }
    for (int e = tid; e < MD_X * MD_Y; e += NT) {
      const int py = e / MD_X, px = e - py * MD_X;
      // (clamped like the tiles: the last rows / columns of a ragged tile are never used)
      (&mt.razff[0][0])[e] = g.cv.razff[(min(i0 - 2 + px, g.Nx + H - 1) + H) + sx * (min(j0 - 2 + py, g.Ny + H) + H)];
    }
  }
  // (uniform base pointer + 32-bit per-lane byte offset: the loads take the scalar-base addressing form and need no
  // 64-bit address arithmetic per lane)
  auto fetch = [&](int k, unsigned oob) {   // oob: byte offset of the own cell at level k
    const real* ub = u + (tile_u + pc * (k + H - cr_.kofs));
    const real* vb = v + (tile_u + pv * (k + H - cr_.kofs));
    const real* wb = w + (tile_w + pc * (k + 1 + H - cr_.kofs));
#pragma unroll
    for (int q = 0; q < NEU; q++)
      if (eu_off[q] >= 0) {
        ru[q] = at(ub, (unsigned)eu_off[q]);
        rv[q] = at(vb, (unsigned)eu_off[q]);
      }
    if constexpr (!WFLY) {
#pragma unroll
      for (int q = 0; q < NEW; q++)
        if (ew_off[q] >= 0) rw[q] = at(wb, (unsigned)ew_off[q]);
    }
    rpw = at(dpx, oob);   // p'(i) - p'(i-1) and p'(j) - p'(j-1), differenced in fp64 by k_compute_p
    rps = at(dpy, oob);
  };
  auto stash = [&](int par) {
    real* U0 = &lds.U[par][0][0];
    real* V0 = &lds.V[par][0][0];
    real* W0 = &lds.W[par][0][0];
#pragma unroll
    for (int q = 0; q < NEU; q++)
      if (eu_off[q] >= 0) {
        if constexpr (LAZY) {
          U0[eu_lds[q]] = ru[q] + (&cr.du[0][0])[eu_lds[q]];
          V0[eu_lds[q]] = rv[q] + (&cr.dv[0][0])[eu_lds[q]];
        } else {
          U0[eu_lds[q]] = ru[q];
          V0[eu_lds[q]] = rv[q];
        }
      }
    if constexpr (!WFLY) {
#pragma unroll
      for (int q = 0; q < NEW; q++)
        if (ew_off[q] >= 0) W0[ew_lds[q]] = CURV ? azw[q] * rw[q] : rw[q];
    }
  };
  real sAu = real(0.), sAv = real(0.), sIu = real(0.), sIv = real(0.);   // AHEAD: this chunk's column sums
  // AHEAD with next.fold: does this tile hold cells with a periodic x image or a y layer to write?
  const bool fold_tile = AHEAD && (j0 == 0 || j0 + V2_TY >= g.Ny || i0 < H || i0 + V2_TX > g.Nx - H);
  // per-block tables for phase 1: packed (row << 8 | column) of every derived point, and the metrics of the rows
  // mdxc[py] = dxc(j0-3+py), mrazf[py] = razf(j0-2+py), mdxf[py] = dxf(j0-3+py)
  // (the curvilinear variant does without them: its metric tiles need the LDS, and with 40 KB per block a fourth block
  // fits a CU; the point's row and column then cost a multiply-high per level)
  __shared__ int ptab[CURV ? 1 : MD_X * MD_Y];
  __shared__ real mdxc[CURV ? 1 : MD_Y + 1], mrazf[CURV ? 1 : MD_Y], mdxf[CURV ? 1 : MD_Y + 1];
  if (!CURV)
    for (int e = tid; e < MD_X * MD_Y; e += NT) {
      const int py = e / MD_X;
      ptab[e] = (py << 8) | (e - py * MD_X);
    }
  if (!CURV && tid <= MD_Y) {
    mdxc[tid] = g.dxc[j0 - 3 + tid];
    mdxf[tid] = g.dxf[j0 - 3 + tid];
    if (tid < MD_Y) mrazf[tid] = g.razf[j0 - 2 + tid];
  }
  // WFLY: this thread's (c,c,c) points of phase 1 (the same e = tid + q NT every level) carry their w; the w tile is the
  // (c,c,c) tile without its first row / column and its last row
  constexpr int NPT = (MD_X * MD_Y + NT - 1) / NT;
  __shared__ real mrazc[(WFLY && !CURV) ? MD_Y : 1];
  real wk[NPT];
  if constexpr (WFLY) {
    if (!CURV && tid < MD_Y) mrazc[tid] = g.razc[j0 - 3 + tid];
#pragma unroll
    for (int q = 0; q < NPT; q++) {
      const int e = tid + q * NT, py = e / MD_X, px = e - py * MD_X;
      // (clamped like the tiles: the last rows / columns of a ragged tile are never used)
      const int gi = min(i0 - 3 + px, g.Nx + H - 1), gj = min(j0 - 3 + py, g.Ny + H - 1);
      wk[q] = (e < MD_X * MD_Y) ? lz.wbase[(long)kc * lz.wplane + i2(g, gi, gj)] : real(0.);
      if constexpr (CURV) wk[q] = wk[q] * g.cv.azcc[i2(g, gi, gj)];   // (the curvilinear tile holds Az w)
    }
  }
  fetch(k0, ob);
  stash(k0 & 1);
  real pw_ = rpw, ps_ = rps;   // p'(i,j) - p'(i-1,j) and p'(i,j) - p'(i,j-1) of the current level
  __syncthreads();

  for (int k = k0; k < k1; k++) {
    const int par = k & 1;
    const real dz = uniform_at(g.dzc, k);
    // Wave priority by phase: the short phases that end in a barrier (the loads of the next level, the derived tiles; at the end of
    // the iteration the stash of the loaded tiles) go ahead of the long arithmetic phase of the three other blocks on the CU -- a
    // wave that reaches its barrier late holds up its whole block, a wave in the arithmetic phase that issues a few cycles later
    // holds up nobody.  Momentum launch 1.231 -> 1.203 ms at 1440x720x48 (-2.3 %; the reverse assignment +1.6 %), same box,
    // alternating runs (profiles/r03_tuning_log.md).
    __builtin_amdgcn_s_setprio(3);
    // ---- phase 0: issue the loads of the next level's tiles (consumed at the end of this iteration)
    const bool more = (k + 1 < k1);
    if (more) fetch(k + 1, ob + (unsigned)pc * SZ);
    real unew = at(u, ob + 4u * (unsigned)pc * SZ), vnew = at(v, obv + 4u * (unsigned)pv * SZ);
    real gmu = real(0.), gmv = real(0.);
    if (AHEAD) { gmu = at(next.GmU, ob); gmv = at(next.GmV, obv); }
    // (LAZY: their correction is added where they are consumed, at the end of the iteration -- added here, the wait for the two
    // loads just issued was a wait for the whole batch of the next level's tile loads: the staging's pipeline, undone)
    // ---- phase 1: derived quantities, once per point.  (row, column) of a point and the row metrics come from the
    // small LDS tables filled once per block: an integer division and five global loads per point and level otherwise
    auto derive = [&](int e, int& py, int& px) -> real {   // returns the point's divergence piece DU + DV
      if constexpr (CURV) {
        py = e / MD_X;
        px = e - py * MD_X;
      } else {
        const int pk = ptab[e];
        py = pk >> 8;
        px = pk & 255;
      }
      // (f,f,c) point (i0-2+px, j0-2+py)
      {
        real uc = lds.U[par][py + 1][px + 1], us = lds.U[par][py][px + 1];
        real vc = lds.V[par][py + 1][px + 1], vw = lds.V[par][py + 1][px];
        if constexpr (CURV)
          lds.Z[py][px] = ((mt.dycf[py + 1][px + 1] * vc - mt.dycf[py + 1][px] * vw) -
                           (mt.dxfc[py + 1][px + 1] * uc - mt.dxfc[py][px + 1] * us)) * mt.razff[py][px];
        else
          lds.Z[py][px] = ((dy * vc - dy * vw) - (mdxc[py + 1] * uc - mdxc[py] * us)) * mrazf[py];
        lds.UQ[py][px] = real(0.5) * (us + uc);
        lds.VQ[py][px] = real(0.5) * (vw + vc);
      }
      // (c,c,c) point (i0-3+px, j0-3+py)
      real du_, dv_;
      if constexpr (CURV) {
        du_ = mt.dyfc[py][px + 1] * dz * lds.U[par][py][px + 1] - mt.dyfc[py][px] * dz * lds.U[par][py][px];
        dv_ = mt.dxcf[py + 1][px] * dz * lds.V[par][py + 1][px] - mt.dxcf[py][px] * dz * lds.V[par][py][px];
      } else {
        const real Ax = dy * dz;
        du_ = Ax * lds.U[par][py][px + 1] - Ax * lds.U[par][py][px];
        dv_ = mdxf[py + 1] * dz * lds.V[par][py + 1][px] - mdxf[py] * dz * lds.V[par][py][px];
      }
      lds.DU[py][px] = du_;
      lds.DV[py][px] = dv_;
      return du_ + dv_;
    };
    if constexpr (WFLY) {
#pragma unroll
      for (int q = 0; q < NPT; q++) {
        const int e = tid + q * NT;
        if (e < MD_X * MD_Y) {
          int py, px;
          const real div = derive(e, py, px);
          if constexpr (CURV) wk[q] = wk[q] - div;       // Az w on the top face of level k
          else wk[q] = wk[q] - div * mrazc[py];          // w on the top face of level k
          if (py >= 1 && py <= MW_Y && px >= 1 && px <= MW_X) lds.W[par][py - 1][px - 1] = wk[q];
        }
      }
    } else {
      for (int e = tid; e < MD_X * MD_Y; e += NT) {
        int py, px;
        derive(e, py, px);
      }
    }
    __syncthreads();
    __builtin_amdgcn_s_setprio(0);
    // ---- phase 2: the two tendencies of cell (i,j,k)
    // tile accessors relative to (i,j)
#define UT(di, dj) lds.U[par][ty + 3 + (dj)][tx + 3 + (di)]
#define VT(di, dj) lds.V[par][ty + 3 + (dj)][tx + 3 + (di)]
#define WT(di, dj) lds.W[par][ty + 2 + (dj)][tx + 2 + (di)]
#define ZF(A, di, dj) lds.A[ty + 2 + (dj)][tx + 2 + (di)]
#define DC(A, di, dj) lds.A[ty + 3 + (dj)][tx + 3 + (di)]
    real gu, gv;
    const int ozt = biased_order_face(k + 1 - kbt, Nzc);
    const real rdz = uniform_at(g.rdzc, k);
    level_orders(k);
    {
      // Packed evaluation: the eight reconstructions of the cell are done as four PAIRS that share stencil shape
      // and order, (.x, .y) = (a term of G_u, a term of G_v); see real2v in device_common.hpp.
      const real vws = VT(-1, 0), vwn = VT(-1, 1), vcs = VT(0, 0), vcn = VT(0, 1);
#define MT(A, di, dj) mt.A[ty + 3 + (dj)][tx + 3 + (di)]
      real vhat_u, uhat_v;
      if constexpr (CURV) {
        vhat_u = (real(0.5) * (MT(dxcf, -1, 0) * vws + MT(dxcf, -1, 1) * vwn) + real(0.5) * (MT(dxcf, 0, 0) * vcs + MT(dxcf, 0, 1) * vcn)) *
                 real(0.5) * rdxc_j;
        uhat_v = (real(0.5) * (MT(dyfc, 0, -1) * UT(0, -1) + MT(dyfc, 1, -1) * UT(1, -1)) +
                  real(0.5) * (MT(dyfc, 0, 0) * UT(0, 0) + MT(dyfc, 1, 0) * UT(1, 0))) * real(0.5) * rdy_j;
      } else {
        vhat_u = (real(0.5) * (dxf_s * vws + dxf_n * vwn) + real(0.5) * (dxf_s * vcs + dxf_n * vcn)) * real(0.5) * rdxc_j;
        uhat_v = (real(0.5) * (dy * UT(0, -1) + dy * UT(1, -1)) + real(0.5) * (dy * UT(0, 0) + dy * UT(1, 0))) * real(0.5) * rdy_j;
      }
#undef MT
      const real uhat_u = uz[3], vhat_v = vz[3];

      // (1) vorticity flux: zeta reconstructed in y for G_u (centre order) and in x for G_v (order 5)
      real hadv_u, hadv_v;
      if (oc_y == 5 && oc_x == 5) {
        real2v zq[6], uq[6], vq[6];
#pragma unroll
        for (int m = 0; m < 6; m++) {
          zq[m] = v2(ZF(Z, 0, m - 2), ZF(Z, m - 2, 0));
          uq[m] = v2(ZF(UQ, 0, m - 2), ZF(UQ, m - 2, 0));
          vq[m] = v2(ZF(VQ, 0, m - 2), ZF(VQ, m - 2, 0));
        }
        const real2v z = biased6p<true>(5, vhat_u > real(0.), uhat_v > real(0.), zq, uq, vq);
        hadv_u = -vhat_u * z.x;
        hadv_v = uhat_v * z.y;
      } else {
        real zq[6], uq[6], vq[6];
#pragma unroll
        for (int m = 0; m < 6; m++) {
          zq[m] = ZF(Z, 0, m - 2);
          uq[m] = ZF(UQ, 0, m - 2);
          vq[m] = ZF(VQ, 0, m - 2);
        }
        hadv_u = -vhat_u * biased6<true>(oc_y, vhat_u > real(0.), zq, uq, vq);
#pragma unroll
        for (int m = 0; m < 6; m++) {
          zq[m] = ZF(Z, m - 2, 0);
          uq[m] = ZF(UQ, m - 2, 0);
          vq[m] = ZF(VQ, m - 2, 0);
        }
        hadv_v = uhat_v * biased6<true>(oc_x, uhat_v > real(0.), zq, uq, vq);
      }

      // (2) G_u: divergence flux and Bernoulli head, both upwinded in x by u (order 5, one direction for the pair)
      real duR, dKu_u;
      {
        real2v qq[6], ss[6];
        real u7[7];
#pragma unroll
        for (int m = 0; m < 7; m++) u7[m] = UT(m - 3, 0);
#pragma unroll
        for (int m = 0; m < 6; m++) {
          const real Du = DC(DU, m - 3, 0);
          qq[m] = v2(Du, real(0.5) * u7[m + 1] * u7[m + 1] - real(0.5) * u7[m] * u7[m]);
          ss[m] = v2(Du + DC(DV, m - 3, 0), real(0.5) * (u7[m] + u7[m + 1]));
        }
        const bool l = uhat_u > real(0.);
        const real2v rr = biased6p<false>(of_x, l, l, qq, ss, ss);
        duR = rr.x;
        dKu_u = rr.y;
      }
      // (3) G_v: the same two terms, upwinded in y by v (face order)
      real dvR, dKv_v;
      {
        real2v qq[6], ss[6];
        real v7[7];
#pragma unroll
        for (int m = 0; m < 7; m++) v7[m] = VT(0, m - 3);
#pragma unroll
        for (int m = 0; m < 6; m++) {
          const real Dv = DC(DV, 0, m - 3);
          qq[m] = v2(Dv, real(0.5) * v7[m + 1] * v7[m + 1] - real(0.5) * v7[m] * v7[m]);
          ss[m] = v2(DC(DU, 0, m - 3) + Dv, real(0.5) * (v7[m] + v7[m + 1]));
        }
        const bool l = vhat_v > real(0.);
        const real2v rr = biased6p<false>(of_y, l, l, qq, ss, ss);
        dvR = rr.x;
        dKv_v = rr.y;
      }
      // (4) vertical advection of u and v: same order, own directions
      // (curvilinear: the tile holds Az w)
      const real wt_u = CURV ? sym_interp(s4f_xw, WT(-2, 0), WT(-1, 0), WT(0, 0), WT(1, 0))
                             : sym_interp(s4f_xw, Az * WT(-2, 0), Az * WT(-1, 0), Az * WT(0, 0), Az * WT(1, 0));
      const real wt_v = CURV ? sym_interp(s4f_yw, WT(0, -2), WT(0, -1), WT(0, 0), WT(0, 1))
                             : sym_interp(s4f_yw, az_m2 * WT(0, -2), az_m1 * WT(0, -1), Az * WT(0, 0), az_p1 * WT(0, 1));
      real2v zz[6];
#pragma unroll
      for (int m = 0; m < 6; m++) zz[m] = v2(uz[m + 1], vz[m + 1]);
      const real2v ftp = v2(wt_u, wt_v) * biased6p<false>(ozt, wt_u > real(0.), wt_v > real(0.), zz, zz, zz);

      {  // ---------------- assemble G_u at (f,c,c)
        real Dv4[4];
#pragma unroll
        for (int m = 0; m < 4; m++) Dv4[m] = DC(DV, m - 2, 0);
        const real dvs = sym_interp(s4f_x, Dv4[0], Dv4[1], Dv4[2], Dv4[3]);
        const real phi = uhat_u * (dvs + duR);
        const real vadv = (phi + (ftp.x - fzu)) * (razc_j * rdz);
        fzu = ftp.x;
        real a4[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
          real vc = VT(0, m - 1), vw = VT(-1, m - 1);
          a4[m] = real(0.5) * vc * vc - real(0.5) * vw * vw;
        }
        const real dKv = sym_interp(s4c_y, a4[0], a4[1], a4[2], a4[3]);
        const real bern = (dKu_u + dKv) * rdxc_j;
        const real cor = -fbar * vhat_u;
        const real dpdx = pw_ * rdxc_j;
        gu = -(hadv_u + vadv + bern) - cor - dpdx;
      }
      {  // ---------------- assemble G_v at (c,f,c)
        real Du4[4];
#pragma unroll
        for (int m = 0; m < 4; m++) Du4[m] = DC(DU, 0, m - 2);
        const real dus = sym_interp(s4f_y, Du4[0], Du4[1], Du4[2], Du4[3]);
        const real phi = vhat_v * (dus + dvR);
        const real vadv = (phi + (ftp.y - fzv)) * (razf_j * rdz);
        fzv = ftp.y;
        real a4[4];
#pragma unroll
        for (int m = 0; m < 4; m++) {
          real un = UT(m - 1, 0), us = UT(m - 1, -1);
          a4[m] = real(0.5) * un * un - real(0.5) * us * us;
        }
        const real dKu = sym_interp(s4c_x, a4[0], a4[1], a4[2], a4[3]);
        const real bern = (dKv_v + dKu) * rdy_j;
        const real cor = fcor_j * uhat_v;
        const real dpdy = ps_ * rdy_j;
        gv = -(hadv_v + vadv + bern) - cor - dpdy;
      }
    }
#undef UT
#undef VT
#undef WT
#undef ZF
#undef DC
    if constexpr (LAZY) {
      if (k + 4 <= g.Nz) {
        unew = unew + cr.du[ty + 3][tx + 3];
        vnew = vnew + cr.dv[ty + 3][tx + 3];
      }
    }
    asm volatile("" : "+v"(unew), "+v"(vnew));
    if (k == g.Nz - 1 && (g.top_flux[0] || g.top_flux[1])) {
      // compute_hydrostatic_boundary_tendency_contributions!: top flux boundary conditions (wind stress)
      const int o2 = i2(g, ic_, jc_);
      if (g.top_flux[0]) gu = gu - g.top_flux[0][o2] * rdz;
      if (g.top_flux[1] && j != g.jws) gv = gv - g.top_flux[1][o2] * rdz;
    }
    if (DRAG && g.bottom_flux[0] != nullptr) {
      // quadratic bottom drag: the bottom flux boundary condition enters the first free level of the face's column
      const int o2 = i2(g, ic_, jc_);
      if (k == (IMM ? KPU : 0)) gu = gu + g.bottom_flux[0][o2] * rdz;
      if (k == (IMM ? KPV : 0) && j != g.jws) gv = gv + g.bottom_flux[1][o2] * rdz;
    }
    if (IMM) {   // faces that touch the solid: no tendency (their velocity is masked and stays zero)
      if (k < KPU) gu = real(0.);
      if (k < KPV) gv = real(0.);
    }
    if (inside_v) {
      if (inside_u) put(Gu, ob, gu);
      put(Gv, obv, gv);
      if (AHEAD) {
        const real au = rfma(next.C1, gu, -(next.C2 * gmu)), av = rfma(next.C1, gv, -(next.C2 * gmv));
        const real un = rfma(next.dt, au, uz[3]);
        const real vn = (next.fold && j == 0) ? real(0.) : rfma(next.dt, av, vz[3]);   // (the southern wall face)
        if (inside_u) put(next.un, ob, un);
        put(next.vn, obv, vn);
        if (next.fold && (fold_tile || k == 0 || k == g.Nz - 1)) {   // (wave-uniform: most tiles and levels skip all of it)
          const unsigned NxB = (unsigned)g.Nx * SZ, sxB = (unsigned)sx * SZ;
          const bool xw = i < H, xe = i >= g.Nx - H;
          auto images = [&](real* base, unsigned off, real x, bool self) {
            if (self) put(base, off, x);
            if (xw) put(base, off + NxB, x);
            if (xe) put(base, off - NxB, x);
          };
          images(next.un, ob, un, false);
          images(next.vn, obv, vn, false);
          if (j == 0) images(next.un, ob - sxB, un, true);                    // row -1 <- row 0
          if (j == g.Ny - 1) {
            images(next.un, ob + sxB, un, true);                             // row Ny <- row Ny-1
            images(next.vn, obv + sxB, real(0.), true);                      // v: face Ny is the northern wall
          }
          if (k == 0) {
            images(next.un, ob - (unsigned)pc * SZ, un, true);
            images(next.vn, obv - (unsigned)pv * SZ, vn, true);
          }
          if (k == g.Nz - 1) {
            images(next.un, ob + (unsigned)pc * SZ, un, true);
            images(next.vn, obv + (unsigned)pv * SZ, vn, true);
          }
        }
        sAu = (k == k0) ? dz * au : rfma(dz, au, sAu);
        sAv = (k == k0) ? dz * av : rfma(dz, av, sAv);
        sIu = (k == k0) ? dz * un : rfma(dz, un, sIu);
        sIv = (k == k0) ? dz * vn : rfma(dz, vn, sIv);
      }
    }
    ob += (unsigned)pc * SZ;
    obv += (unsigned)pv * SZ;
#pragma unroll
    for (int m = 0; m < 6; m++) {
      uz[m] = uz[m + 1];
      vz[m] = vz[m + 1];
    }
    uz[6] = unew;
    vz[6] = vnew;
    __builtin_amdgcn_s_setprio(3);
    if (more) {
      stash(par ^ 1);
      pw_ = rpw;
      ps_ = rps;
    }
    __syncthreads();   // next tiles visible; derived arrays free for the next phase 1
  }
  if (AHEAD && inside_v) {
    const long o2 = i2(g, i, j), q = (long)kchunks * next.plane2, c = (long)kc * next.plane2;
    next.P[q + c + o2] = sAv;
    next.P[3 * q + c + o2] = sIv;
    if (inside_u) {
      next.P[c + o2] = sAu;
      next.P[2 * q + c + o2] = sIu;
    }
  }
}

constexpr int V3_OUT = 63;   // outputs per wavefront

// AHEAD: the kernel also writes the tracers of the next time level, Tn = T + dt (C1 G^n - C2 G^-), into a second
// pair of arrays: at this point T and the fresh G^n are in registers, so the next step's ab2_step_field! for T and S
// costs one read (G^-) and one write per tracer here instead of a separate 3R + 1W sweep.  The host adopts Tn/Sn
// (pointer exchange) at the next ab2_step! if dt, chi and the inputs are still the ones used here.
struct Ab2Ahead {
  const real *GmT, *GmS;
  real *Tn, *Sn;
  real dt, C1, C2;
  // WCORR (with LAZY): the kernel also WRITES the corrected velocities it forms -- u + du on the west face, v + dv on the south
  // face of its cell -- into a second pair of arrays (interior cells; their halo cells by the fill that follows).  Every u, v is
  // some cell's west / south face, so this replaces the corrector's sweep over u and v (2R + 2W per cell) by two stores here,
  // and the momentum kernel that follows reads corrected velocities like any other.
  real *uc, *vc;
};

// =============================================================================================
// Tracer tendencies: wave-autonomous (no LDS, no barriers).  A wavefront owns 63 consecutive cells of one row (lane
// 63 only supplies the east face of lane 62), marches up the column, reconstructs the WEST face once (the east face
// arrives from lane+1 by a wave shuffle), the TOP face once (bottom = carried) and both y faces: 8 reconstructions per
// cell instead of 12.  T and S are carried as ONE two-wide value per stencil point.  Both tracers see the same advecting velocity, the same
// upwind direction and the same (wave-uniform) wall-adjacent order, so every reconstruction is evaluated once on
// register pairs: the smoothness indicators, polynomials and weights become v_pk_fma/mul/add_f32, which do two lanes'
// worth of fp32 work per issue slot (device_common.hpp, real2v).  Only rcp, min, abs and the upwind selects stay
// per-half.
// =============================================================================================
// IMM: the grid has an immersed boundary.  The orders of the reconstructions then come per lane from the folded
// per-column tables (device_common.hpp, Immersed) instead of per wave from the row index; everything else is unchanged:
// the fluxes through faces that touch the solid vanish because the velocities there are masked to zero.
// FOLD (with AHEAD, single periodic domain): T and S of the next time level are written with the halo cells
// tupled_fill_halo_regions! derives from them (periodic x images, y layer, z layers and their x images), so that the
// adopted buffers need no fill launch.
// The arithmetic of one tile (63 cells of a row x 4 rows x one chunk of levels); L = logical tile index.
// CURV: orthogonal curvilinear grid (with the tables): the three face lengths, the area and its reciprocal per lane.
// LAZY: the barotropic correction of this step is added to u and v as they are loaded (see k_momentum_tendencies_v5).
// ORD: 5 = WENO(order = 5) (baroclinic_instability_model); 7 = WENO(order = 7) (ClimaOcean's ocean_simulation): windows of
// 2R = 8 values per face (R = 4), the order-5 path next to walls and the immersed boundary (biased8, device_common.hpp).
// WFLY: w is not read; Az w on the top face follows from the divergence of the transports the lane holds anyway --
// Az w(k+1) = Az w(k) - [(Axu(i+1) - Axu(i)) + (Ayn - Ays)], the east face's transport from the next lane -- starting from
// lz.wbase at the chunk's first level (k_w_bases).  Continuity and advection then see the very same transports.
template <bool AHEAD, bool IMM, bool FOLD, bool CURV = false, bool LAZY = false, int ORD = 5, bool WFLY = false, bool WCORR = false>
__device__ __forceinline__ void tracer_tile(const Grid& g, const real* __restrict__ u, const real* __restrict__ v,
                                            const real* __restrict__ w, const real* __restrict__ T,
                                            const real* __restrict__ S, real* __restrict__ GT, real* __restrict__ GS,
                                            int nbx, int kchunks, const Ab2Ahead& next, const int L, const LazyCorr& lz) {
  const int bx = L % nbx, r = L / nbx;
  const int kc = r % kchunks, by = r / kchunks;
  const int klen = (g.Nz + kchunks - 1) / kchunks;
  const int k0 = kc * klen, k1 = min(g.Nz, k0 + klen);
  const int lane = threadIdx.x;
  const int i = bx * V3_OUT + lane, j = by * blockDim.y + threadIdx.y;
  if (j >= g.Ny) return;                       // whole wave (one row) leaves together: no barriers in this kernel
  const bool writes = (lane < V3_OUT) && (i < g.Nx);
  const int sx = g.sx, pc = g.pl_c, pv = g.pl_v;
  static_assert(!CURV || IMM, "the curvilinear variant takes its orders from the tables");
  const int om = i2(g, min(i, g.Nx), j);
  const real dy = CURV ? g.cv.dyfc[om] : g.dy, Az = CURV ? g.cv.azcc[om] : g.azc[j];
  const real dxf_s = CURV ? g.cv.dxcf[om] : g.dxf[j], dxf_n = CURV ? g.cv.dxcf[om + g.sx] : g.dxf[j + 1];
  const real razc_j = CURV ? g.cv.razcc[om] : g.razc[j];
  static_assert(!WCORR || LAZY, "the corrected velocities are what the lazy loads form");
  real du_l = real(0.), dv_s = real(0.), dv_n = real(0.);
  // (with a bottom the correction acts from the face's first free level on: the faces below touch the solid and stay zero)
  int KPUl = 0, KPVs = 0, KPVn = 0;
  if (LAZY) {
    du_l = lz.du[om];
    dv_s = lz.dv[om];
    dv_n = lz.dv[om + g.sx];
    if (IMM) {
      const unsigned C = g.im.ordC[om], Cn = g.im.ordC[om + g.sx];
      KPUl = (C >> 8) & 255; KPVs = (C >> 16) & 255; KPVn = (Cn >> 16) & 255;
    }
  }
  constexpr int R = ORD == 7 ? 4 : 3;   // reach of the reconstruction stencils
  static_assert(ORD == 5 || ORD == 7, "WENO(order = 5) or WENO(order = 7)");
  const int jw_ = j - g.jws, Nyw_ = g.jwn - g.jws;   // (rows counted from the global southern wall)
  int oys = ORD == 7 ? biased_order_face7(jw_, Nyw_) : biased_order_face(jw_, Nyw_);
  int oyn = ORD == 7 ? biased_order_face7(jw_ + 1, Nyw_) : biased_order_face(jw_ + 1, Nyw_), ox = ORD;
  int kbt = 0, KX5 = 0, KX3 = 0, KY5 = 0, KY3 = 0, KY5n = 0, KY3n = 0, KX7 = 0, KY7 = 0, KY7n = 0;
  if (IMM) {
    const int o2 = i2(g, min(i, g.Nx), j);
    const unsigned A = g.im.ordA[o2], B = g.im.ordB[o2], An = g.im.ordA[o2 + g.sx], Bn = g.im.ordB[o2 + g.sx];
    kbt = A & 255; KX5 = (A >> 8) & 255; KX3 = (A >> 16) & 255; KY5 = A >> 24; KY3 = B & 255;
    KY5n = An >> 24; KY3n = Bn & 255;
    if (ORD == 7) {
      const unsigned D = g.im.ordD[o2], Dn = g.im.ordD[o2 + g.sx];
      KX7 = D & 255; KY7 = (D >> 8) & 255; KY7n = (Dn >> 8) & 255;
    }
  }
  const int Nzc = g.Nz - kbt;   // (levels above the column's bottom: the bottom acts like the wall at k = 0, shifted)

  // buffer views (device_common.hpp): one per-lane byte offset for the centre-shaped arrays, one for v
  constexpr int SZ = (int)sizeof(real);
  const long nzp = g.Nz + 2 * g.H;
  // (arrays beyond 2 GB: the descriptors start at the plane of the stencil's lowest corner, level k0 - R, and the per-lane
  // byte offsets count from there: a chunk of levels spans (levels + 2R + 1) planes)
  const long pb = k0 + g.H - R;                 // first plane of the views (>= 0: H >= R)
  const Buf bT = make_buf(T + pc * pb, pc * (nzp - pb)), bS = make_buf(S + pc * pb, pc * (nzp - pb)),
            bu = make_buf(u + pc * pb, pc * (nzp - pb)), bw = make_buf(w + pc * pb, pc * (nzp + 1 - pb)),
            bv = make_buf(v + pv * (pb + R), pv * (nzp - pb - R)), bGT = make_buf(GT + pc * pb, pc * (nzp - pb)),
            bGS = make_buf(GS + pc * pb, pc * (nzp - pb));
  Buf buc = bu, bvc = bv;
  if (WCORR) {
    buc = make_buf(next.uc + pc * pb, pc * (nzp - pb));
    bvc = make_buf(next.vc + pv * (pb + R), pv * (nzp - pb - R));
  }
  Buf bGmT = bT, bGmS = bT, bTn = bT, bSn = bT;
  if (AHEAD) {
    bGmT = make_buf(next.GmT + pc * pb, pc * (nzp - pb)); bGmS = make_buf(next.GmS + pc * pb, pc * (nzp - pb));
    bTn = make_buf(next.Tn + pc * pb, pc * (nzp - pb));   bSn = make_buf(next.Sn + pc * pb, pc * (nzp - pb));
  }
  // lanes past the east edge work on a clamped (duplicate) column.  `vo` addresses the (-3,-3,-3) corner of the
  // cell's stencil so that every displacement below is a non-negative byte count (needs H >= 3, as WENO5 does).
  const int cc = (R * pc + R * sx + R) * SZ;                    // corner -> cell
  int vo = ic(g, min(i, g.Nx), j, R - g.H) * SZ - cc;   // (level k0 is plane R of the views)
  int vov = iv(g, min(i, g.Nx), j, -g.H) * SZ;           // (... and plane 0 of v's)
#define CZ(m) (((m) * pc + R * sx + R) * SZ)                     // (0, 0, m-R)
#define CY(m) ((R * pc + (m) * sx + R) * SZ)                     // (0, m-R, 0)
#define CX(m) ((R * pc + R * sx) * SZ), ((m) * SZ)               // (m-R, 0, 0): uniform part, immediate part
  auto recon = [](int order, bool left, const real2v* w) {      // reconstruction to the face between w[R-1] and w[R]
    if constexpr (ORD == 7) return biased8<real2v>(order, left, w);
    else return biased6<false, real2v>(order, left, w, w, w);
  };
  auto zorder = [](int f, int N) { return ORD == 7 ? biased_order_face7(f, N) : biased_order_face(f, N); };
  real2v cz[2 * R + 1];                        // vertical window of (T, S)
#pragma unroll
  for (int m = 0; m < 2 * R + 1; m++) cz[m] = v2(bload(bT, vo, CZ(m)), bload(bS, vo, CZ(m)));
  real2v fz;
  real Azw_cur;                                // WFLY: Az w on the bottom face of the current level
  {
    real Azw = WFLY ? Az * lz.wbase[(long)kc * lz.wplane + om] : Az * bload(bw, vo, cc);
    Azw_cur = Azw;
    int ord = zorder(k0 - kbt, Nzc);
    fz = Azw * recon(ord, Azw > real(0.), cz);
  }
  // FOLD: does this wave hold cells with a periodic x image or a y layer to write?
  const bool fold_row = FOLD && (j == 0 || j == g.Ny - 1 || bx * V3_OUT < g.H || bx * V3_OUT + V3_OUT > g.Nx - g.H);
  for (int k = k0; k < k1; k++) {
    const real dz = uniform_at(g.dzc, k), rdz = uniform_at(g.rdzc, k);   // (scalar loads: device_common.hpp)
    if (IMM) {
      ox = ORD == 7 ? order_from7(k, KX7, KX5, KX3) : order_from(k, KX5, KX3);
      oys = ORD == 7 ? order_from7(k, KY7, KY5, KY3) : order_from(k, KY5, KY3);
      oyn = ORD == 7 ? order_from7(k, KY7n, KY5n, KY3n) : order_from(k, KY5n, KY3n);
    }
    real u_l = bload(bu, vo, cc), v_s = bload(bv, vov, 0), v_n = bload(bv, vov, sx * SZ);
    // Every load of the level is issued HERE, ahead of the first wait: the newest value of the vertical window (needed by the next
    // level) and, AHEAD, the old tendencies the new T, S are advanced with.  Loaded where they are used -- the window value behind
    // the level's stores, G^- inside the `writes` branch, one after the other -- each cost the wave a memory latency of its own:
    // four sleeps per level instead of one (tools/kernel_isa.py: s_waitcnt vmcnt(0) x 4 in the loop).
    const real2v cz_next = v2(bload(bT, vo, CZ(2 * R) + pc * SZ), bload(bS, vo, CZ(2 * R) + pc * SZ));
    real gmT = real(0.), gmS = real(0.);
    if (AHEAD) {
      gmT = bload(bGmT, vo, cc);
      gmS = bload(bGmS, vo, cc);
    }
    if (LAZY) {
      u_l = (!IMM || k >= KPUl) ? u_l + du_l : u_l;
      v_s = (!IMM || k >= KPVs) ? v_s + dv_s : v_s;
      v_n = (!IMM || k >= KPVn) ? v_n + dv_n : v_n;
    }
    const real Axu = dy * dz * u_l;
    const real Ays = dxf_s * dz * v_s;
    const real Ayn = dxf_n * dz * v_n;
    real Azw;
    if constexpr (WFLY) {
      Azw = Azw_cur - ((__shfl_down(Axu, 1) - Axu) + (Ayn - Ays));
      Azw_cur = Azw;
    } else {
      Azw = Az * bload(bw, vo, cc + pc * SZ);
    }
    real2v q[2 * R + 1];
#pragma unroll
    for (int m = 0; m < 2 * R; m++) q[m] = v2(bload(bT, vo + m * SZ, (R * pc + R * sx) * SZ), bload(bS, vo + m * SZ, (R * pc + R * sx) * SZ));
    const real2v fx = Axu * recon(IMM ? ox : ORD, Axu > real(0.), q);
#pragma unroll
    for (int m = 0; m < 2 * R + 1; m++) q[m] = v2(bload(bT, vo, CY(m)), bload(bS, vo, CY(m)));
    // (the two ends of the window are used by the full-order branch of one reconstruction each: left alone, the compiler sinks
    // their loads INTO that branch -- the common one -- and waits for each separately: two more sleeps per level)
    asm volatile("" : "+v"(q[0]), "+v"(q[2 * R]));
    const real2v fs = Ays * recon(oys, Ays > real(0.), q);
    const real2v fn = Ayn * recon(oyn, Ayn > real(0.), q + 1);
    // top face from the vertical window
    const int ozt = zorder(k + 1 - kbt, Nzc);
    const real2v ft = Azw * recon(ozt, Azw > real(0.), cz + 1);
    // east faces = west faces of the next lane
    const real2v fe = v2(__shfl_down(fx.x, 1), __shfl_down(fx.y, 1));
    if (writes) {
      const real rV = razc_j * rdz;
      real2v G = -(((fe - fx) + (fn - fs) + (ft - fz)) * rV);
      if (k == g.Nz - 1 && (g.top_flux[2] || g.top_flux[3]) && (!IMM || kbt < g.Nz)) {
        // compute_hydrostatic_boundary_tendency_contributions!: top flux boundary conditions (heat, fresh water)
        const int o2 = i2(g, i, j);
        if (g.top_flux[2]) G.x = G.x - g.top_flux[2][o2] * rdz;
        if (g.top_flux[3]) G.y = G.y - g.top_flux[3][o2] * rdz;
      }
      bstore(bGT, vo, cc, G.x);
      bstore(bGS, vo, cc, G.y);
      if (WCORR) {   // (with the level's other stores: a store ahead of the window loads would sit in front of them in the memory counter)
        bstore(buc, vo, cc, u_l);
        bstore(bvc, vov, 0, v_s);
      }
      if (AHEAD) {   // T, S of the NEXT step while T, S (cz[3]) and the new tendency are in registers
        const real tn = ab2_advance(cz[R].x, G.x, gmT, next.dt, next.C1, next.C2);
        const real sn = ab2_advance(cz[R].y, G.y, gmS, next.dt, next.C1, next.C2);
        bstore(bTn, vo, cc, tn);
        bstore(bSn, vo, cc, sn);
        if (FOLD && (fold_row || k == 0 || k == g.Nz - 1)) {   // (wave-uniform: most waves and levels skip all of it)
          const int NxB = g.Nx * SZ;
          const bool xw = i < g.H, xe = i >= g.Nx - g.H;
          auto images = [&](int so, bool self) {   // the cell displaced by `so` bytes (a y / z layer) and its x images
            if (self) { bstore(bTn, vo, cc + so, tn); bstore(bSn, vo, cc + so, sn); }
            if (xw) { bstore(bTn, vo + NxB, cc + so, tn); bstore(bSn, vo + NxB, cc + so, sn); }
            if (xe) { bstore(bTn, vo - NxB, cc + so, tn); bstore(bSn, vo - NxB, cc + so, sn); }
          };
          images(0, false);
          if (j == 0) images(-sx * SZ, true);
          if (j == g.Ny - 1) images(sx * SZ, true);
          if (k == 0) images(-pc * SZ, true);
          if (k == g.Nz - 1) images(pc * SZ, true);
        }
      }
    }
    fz = ft;
    vo += pc * SZ;
    vov += pv * SZ;
#pragma unroll
    for (int m = 0; m < 2 * R; m++) cz[m] = cz[m + 1];
    cz[2 * R] = cz_next;
  }
#undef CZ
#undef CY
#undef CX
}
template <int MINW, bool AHEAD, bool IMM, bool FOLD = false, bool CURV = false, bool LAZY = false, int ORD = 5, bool WFLY = false, bool WCORR = false>
__global__ __launch_bounds__(256, MINW) void k_tracer_tendencies_v5(Grid g, const real* __restrict__ u,
                                                              const real* __restrict__ v,
                                                              const real* __restrict__ w,
                                                              const real* __restrict__ T, const real* __restrict__ S,
                                                              real* __restrict__ GT, real* __restrict__ GS, int nbx,
                                                              int kchunks, int nb, Ab2Ahead next, LazyCorr lz) {
  tracer_tile<AHEAD, IMM, FOLD, CURV, LAZY, ORD, WFLY, WCORR>(g, u, v, w, T, S, GT, GS, nbx, kchunks, next, xcd_remap(blockIdx.x, nb), lz);
}

// =============================================================================================
// Advection of ONE tracer (CATKE's e): the wave-autonomous scheme of tracer_tile with the two halves of every register pair
// holding the same field at two COLUMNS 63 apart instead of two tracers at one -- a wave owns 2 x 63 consecutive cells of a
// row.  Velocities, upwind directions, metrics and (with bathymetry) reconstruction orders are then per half; the arithmetic
// per half is that of tracer_tile.  (Sending e through the two-tracer kernel with T = S = e left half of every packed
// instruction idle: 1.73 ms for the third tracer against 1.95 ms for the first two at 1440 x 720 x 60 with WENO(order = 7).)
// No look-ahead, no halo writes, no flux boundary condition: e has none of them.
// =============================================================================================
// WFLY: w is not read (the field is stale in a step that carries w inside its tendency kernels): Az w of both columns from the
// divergence of their transports, the east faces' from the next lane, starting from lz.wbase -- as tracer_tile does.
constexpr int V3_PAIR = 2 * V3_OUT;   // outputs per wavefront
template <bool IMM, bool CURV, int ORD, bool WFLY = false>
__device__ __forceinline__ void tracer_tile_single(const Grid& g, const real* __restrict__ u, const real* __restrict__ v,
                                                   const real* __restrict__ w, const real* __restrict__ E,
                                                   real* __restrict__ GE, int nbx, int kchunks, const int L, const LazyCorr& lz) {
  constexpr int R = ORD == 7 ? 4 : 3;
  const int bx = L % nbx, r = L / nbx;
  const int kc = r % kchunks, by = r / kchunks;
  const int klen = (g.Nz + kchunks - 1) / kchunks;
  const int k0 = kc * klen, k1 = min(g.Nz, k0 + klen);
  const int lane = threadIdx.x;
  const int j = by * blockDim.y + threadIdx.y;
  if (j >= g.Ny) return;
  const int i0 = bx * V3_PAIR + lane, i1 = i0 + V3_OUT;          // the two columns of this lane
  const bool wr0 = (lane < V3_OUT) && (i0 < g.Nx), wr1 = (lane < V3_OUT) && (i1 < g.Nx);
  const int c0 = min(i0, g.Nx), c1 = min(i1, g.Nx);              // (lanes past the east edge work on a clamped column)
  const int sx = g.sx, pc = g.pl_c, pv = g.pl_v;
  static_assert(!CURV || IMM, "the curvilinear variant takes its orders from the tables");
  const int om0 = i2(g, c0, j), om1 = i2(g, c1, j);
  auto m2 = [&](const real* tab, int d) { return v2(tab[om0 + d], tab[om1 + d]); };
  const real2v dy = CURV ? m2(g.cv.dyfc, 0) : real2v(g.dy), Az = CURV ? m2(g.cv.azcc, 0) : real2v(g.azc[j]);
  const real2v dxf_s = CURV ? m2(g.cv.dxcf, 0) : real2v(g.dxf[j]), dxf_n = CURV ? m2(g.cv.dxcf, g.sx) : real2v(g.dxf[j + 1]);
  const real2v razc_j = CURV ? m2(g.cv.razcc, 0) : real2v(g.razc[j]);
  const int jw_ = j - g.jws, Nyw_ = g.jwn - g.jws;   // (rows counted from the global southern wall)
  const int oys_w = ORD == 7 ? biased_order_face7(jw_, Nyw_) : biased_order_face(jw_, Nyw_);
  const int oyn_w = ORD == 7 ? biased_order_face7(jw_ + 1, Nyw_) : biased_order_face(jw_ + 1, Nyw_);
  int kbt[2] = {0, 0}, KX5[2] = {0, 0}, KX3[2] = {0, 0}, KY5[2] = {0, 0}, KY3[2] = {0, 0}, KY5n[2] = {0, 0}, KY3n[2] = {0, 0};
  int KX7[2] = {0, 0}, KY7[2] = {0, 0}, KY7n[2] = {0, 0};
  if (IMM) {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int o2 = h ? om1 : om0;
      const unsigned A = g.im.ordA[o2], B = g.im.ordB[o2], An = g.im.ordA[o2 + g.sx], Bn = g.im.ordB[o2 + g.sx];
      kbt[h] = A & 255; KX5[h] = (A >> 8) & 255; KX3[h] = (A >> 16) & 255; KY5[h] = A >> 24; KY3[h] = B & 255;
      KY5n[h] = An >> 24; KY3n[h] = Bn & 255;
      if (ORD == 7) {
        const unsigned D = g.im.ordD[o2], Dn = g.im.ordD[o2 + g.sx];
        KX7[h] = D & 255; KY7[h] = (D >> 8) & 255; KY7n[h] = (Dn >> 8) & 255;
      }
    }
  }
  const int Nzc0 = g.Nz - kbt[0], Nzc1 = g.Nz - kbt[1];
  constexpr int SZ = (int)sizeof(real);
  const long nzp = g.Nz + 2 * g.H;
  const long pb = k0 + g.H - R;                 // (views from the plane of the stencil's lowest corner: see tracer_tile)
  const Buf bE = make_buf(E + pc * pb, pc * (nzp - pb)), bu = make_buf(u + pc * pb, pc * (nzp - pb)),
            bw = make_buf(w + pc * pb, pc * (nzp + 1 - pb)), bv = make_buf(v + pv * (pb + R), pv * (nzp - pb - R)),
            bG = make_buf(GE + pc * pb, pc * (nzp - pb));
  const int cc = (R * pc + R * sx + R) * SZ;
  int vo0 = ic(g, c0, j, R - g.H) * SZ - cc, vo1 = ic(g, c1, j, R - g.H) * SZ - cc;
  int vv0 = iv(g, c0, j, -g.H) * SZ, vv1 = iv(g, c1, j, -g.H) * SZ;
  auto ld2 = [&](const Buf& b, int a0, int a1, int so) { return v2(bload(b, a0, so), bload(b, a1, so)); };
  auto zorder = [](int f, int N) { return ORD == 7 ? biased_order_face7(f, N) : biased_order_face(f, N); };
  auto ord3 = [](int k, int K7_, int K5_, int K3_) { return ORD == 7 ? order_from7(k, K7_, K5_, K3_) : order_from(k, K5_, K3_); };
#define CZ(m) (((m) * pc + R * sx + R) * SZ)
#define CY(m) ((R * pc + (m) * sx + R) * SZ)
  real2v cz[2 * R + 1];
#pragma unroll
  for (int m = 0; m < 2 * R + 1; m++) cz[m] = ld2(bE, vo0, vo1, CZ(m));
  real2v fz;
  real2v Azw_cur;                              // WFLY: Az w on the bottom face of the current level
  {
    const real2v Azw = WFLY ? Az * v2(lz.wbase[(long)kc * lz.wplane + om0], lz.wbase[(long)kc * lz.wplane + om1]) : Az * ld2(bw, vo0, vo1, cc);
    Azw_cur = Azw;
    fz = Azw * biased_pair<ORD>(zorder(k0 - kbt[0], Nzc0), zorder(k0 - kbt[1], Nzc1), Azw.x > real(0.), Azw.y > real(0.), cz);
  }
  for (int k = k0; k < k1; k++) {
    const real dz = uniform_at(g.dzc, k), rdz = uniform_at(g.rdzc, k);
    int ox0 = ORD, ox1 = ORD, os0 = oys_w, os1 = oys_w, on0 = oyn_w, on1 = oyn_w;
    if (IMM) {
      ox0 = ord3(k, KX7[0], KX5[0], KX3[0]); ox1 = ord3(k, KX7[1], KX5[1], KX3[1]);
      os0 = ord3(k, KY7[0], KY5[0], KY3[0]); os1 = ord3(k, KY7[1], KY5[1], KY3[1]);
      on0 = ord3(k, KY7n[0], KY5n[0], KY3n[0]); on1 = ord3(k, KY7n[1], KY5n[1], KY3n[1]);
    }
    const real2v cz_next = ld2(bE, vo0, vo1, CZ(2 * R) + pc * SZ);   // (with the level's other loads: see tracer_tile)
    const real2v Axu = dy * dz * ld2(bu, vo0, vo1, cc);
    const real2v Ays = dxf_s * dz * ld2(bv, vv0, vv1, 0);
    const real2v Ayn = dxf_n * dz * ld2(bv, vv0, vv1, sx * SZ);
    real2v Azw;
    if constexpr (WFLY) {
      Azw = Azw_cur - ((v2(__shfl_down(Axu.x, 1), __shfl_down(Axu.y, 1)) - Axu) + (Ayn - Ays));
      Azw_cur = Azw;
    } else {
      Azw = Az * ld2(bw, vo0, vo1, cc + pc * SZ);
    }
    real2v q[2 * R + 1];
#pragma unroll
    for (int m = 0; m < 2 * R; m++) q[m] = ld2(bE, vo0 + m * SZ, vo1 + m * SZ, (R * pc + R * sx) * SZ);
    const real2v fx = Axu * biased_pair<ORD>(ox0, ox1, Axu.x > real(0.), Axu.y > real(0.), q);
#pragma unroll
    for (int m = 0; m < 2 * R + 1; m++) q[m] = ld2(bE, vo0, vo1, CY(m));
    asm volatile("" : "+v"(q[0]), "+v"(q[2 * R]));   // (no load sunk into a branch of one reconstruction: see tracer_tile)
    const real2v fs = Ays * biased_pair<ORD>(os0, os1, Ays.x > real(0.), Ays.y > real(0.), q);
    const real2v fn = Ayn * biased_pair<ORD>(on0, on1, Ayn.x > real(0.), Ayn.y > real(0.), q + 1);
    const real2v ft = Azw * biased_pair<ORD>(zorder(k + 1 - kbt[0], Nzc0), zorder(k + 1 - kbt[1], Nzc1), Azw.x > real(0.),
                                             Azw.y > real(0.), cz + 1);
    // east faces = west faces of the next lane; lane 62's second column borders the NEXT wave's first, supplied by lane 63
    const real2v fe = v2(__shfl_down(fx.x, 1), __shfl_down(fx.y, 1));
    const real2v G = -(((fe - fx) + (fn - fs) + (ft - fz)) * (razc_j * rdz));
    if (wr0) bstore(bG, vo0, cc, G.x);
    if (wr1) bstore(bG, vo1, cc, G.y);
    fz = ft;
    vo0 += pc * SZ; vo1 += pc * SZ;
    vv0 += pv * SZ; vv1 += pv * SZ;
#pragma unroll
    for (int m = 0; m < 2 * R; m++) cz[m] = cz[m + 1];
    cz[2 * R] = cz_next;
  }
#undef CZ
#undef CY
}
template <int MINW, bool IMM, bool CURV, int ORD, bool WFLY = false>
__global__ __launch_bounds__(256, MINW) void k_tracer_tendencies_single(Grid g, const real* __restrict__ u,
                                                                       const real* __restrict__ v,
                                                                       const real* __restrict__ w,
                                                                       const real* __restrict__ E, real* __restrict__ GE,
                                                                       int nbx, int kchunks, int nb, LazyCorr lz) {
  tracer_tile_single<IMM, CURV, ORD, WFLY>(g, u, v, w, E, GE, nbx, kchunks, xcd_remap(blockIdx.x, nb), lz);
}

}  // namespace gb25

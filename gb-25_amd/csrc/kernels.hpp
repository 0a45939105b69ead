// kernels.hpp -- hand-written HIP kernels (gfx950) for the HydrostaticFreeSurfaceModel time step.
//
// Thread mapping used throughout: lanes run along i (the fastest-varying, coalesced axis), so
// a wavefront is 64 consecutive i at one (j,k); everything that depends only on j or k
// (wall-adjacent order reduction, metrics) is wave-uniform.  3-D tendency kernels use a 1-D grid
// of (64 x 4) tiles ordered (i-tile, k, j-tile) and remapped so that every XCD sweeps one
// contiguous latitude band bottom-to-top: the +-3 vertical stencil planes stay in that XCD's L2.
// Column kernels (w, pressure, AB2 + vertical integral, corrector) give one thread a whole
// (i,j) column and march in k.
#pragma once
#include "device_common.hpp"

namespace gb25 {

// a value and its periodic x images (see the FOLD variants of k_corrector, k_barotropic_multi and the tracer kernel)
__device__ __forceinline__ void store_x_images(const Grid& g, real* a, int o, real x, bool xw, bool xe) {
  a[o] = x;
  if (xw) a[o + g.Nx] = x;   // column i < H is the periodic image of column i + Nx (east halo)
  if (xe) a[o - g.Nx] = x;   // column i >= Nx - H of column i - Nx (west halo)
}

constexpr int TX = 64, TY = 4;

struct TileIdx {
  int i, j, k;
  bool ok;
};
__device__ __forceinline__ TileIdx tile_index(const Grid& g, int nbx, int nb) {
  int L = xcd_remap(blockIdx.x, nb);
  int bx = L % nbx, r = L / nbx;
  int k = r % g.Nz, by = r / g.Nz;
  TileIdx t;
  t.i = bx * TX + threadIdx.x;
  t.j = by * TY + threadIdx.y;
  t.k = k;
  t.ok = (t.i < g.Nx) && (t.j < g.Ny);
  return t;
}

// =============================================================================================
// Halo filling: tupled_fill_halo_regions!(prognostic_fields) (GB-25 src/precompile.jl:44-46).
// Bounded y / z: ONE halo layer (zero-gradient; wall-normal velocity = 0).  Periodic x: all H
// columns over the whole parent (j,k) extent, filled last so corners are consistent.
// =============================================================================================
struct Halo3 {
  real* p[4];   // up to four 3-D fields ...
  int is_v[4];   // ... flagged when face-located in y (v-shaped parent, wall-normal velocity)
  int n;
  int xf[4], neg[4];   // zipper fold only: located on x faces (u); a vector component (changes sign across the fold)
};
struct Halo2 {
  real* p[3];  // centre-y fields first, then the face-y field (is_v[])
  int is_v[3];
  int n;
  int xf[3], neg[3];
};

// south/north layer of the four 3-D fields + all 2-D fields over columns [i0, i0+ni).
// grid: (ceil(ni/256), Nz + 1)
// (a side without a wall -- the fold line; the open sides of a rank of a 2-D decomposition -- has no layer to fill: its halo rows
// are images / the neighbour's rows)
__device__ __forceinline__ void fill_y_body(const Grid& g, const Halo3& f3, const Halo2& f2, int i0, int ni, int k) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ni) return;
  i += i0;
  const bool south = g.jws == 0, north = g.jwn == g.Ny && !g.cv.north_fold;
  if (k < g.Nz) {
    for (int q = 0; q < f3.n; q++) {
      real* c = f3.p[q];
      if (f3.is_v[q]) {  // v: faces 0 and Ny are walls (Ny: unless it is the fold line -- k_fill_fold)
        if (south) c[iv(g, i, 0, k)] = real(0.);
        if (north) c[iv(g, i, g.Ny, k)] = real(0.);
      } else {
        if (south) c[ic(g, i, -1, k)] = c[ic(g, i, 0, k)];
        if (north) c[ic(g, i, g.Ny, k)] = c[ic(g, i, g.Ny - 1, k)];
      }
    }
  } else {
    for (int q = 0; q < f2.n; q++) {
      real* c = f2.p[q];
      if (f2.is_v[q]) {
        if (south) c[i2(g, i, 0)] = real(0.);
        if (north) c[i2(g, i, g.Ny)] = real(0.);
      } else {
        if (south) c[i2(g, i, -1)] = c[i2(g, i, 0)];
        if (north) c[i2(g, i, g.Ny)] = c[i2(g, i, g.Ny - 1)];
      }
    }
  }
}
__global__ void k_fill_y(Grid g, Halo3 f3, Halo2 f2, int i0, int ni) { fill_y_body(g, f3, f2, i0, ni, blockIdx.y); }
// bottom/top layer of the four 3-D fields over columns [i0, i0+ni).  grid: (ceil(ni/256), rows): row j = jlo + blockIdx.y
// (rows [0, Ny); a rank of a 2-D decomposition also does the halo rows of its open sides once they have arrived)
__device__ __forceinline__ void fill_z_body(const Grid& g, const Halo3& f3, int i0, int ni, int j) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ni) return;
  i += i0;
  for (int q = 0; q < f3.n; q++) {
    real* c = f3.p[q];
    if (f3.is_v[q]) {
      // (row 0 is the wall face: the y fill zeroes it at every level, and copying that zero is spelled out here so
      // that this fill does not depend on the other one having run)
      c[iv(g, i, j, -1)] = (j == g.jws) ? real(0.) : c[iv(g, i, j, 0)];
      c[iv(g, i, j, g.Nz)] = (j == g.jws) ? real(0.) : c[iv(g, i, j, g.Nz - 1)];
    } else {
      c[ic(g, i, j, -1)] = c[ic(g, i, j, 0)];
      c[ic(g, i, j, g.Nz)] = c[ic(g, i, j, g.Nz - 1)];
    }
  }
}
__global__ void k_fill_z(Grid g, Halo3 f3, int i0, int ni, int jlo) { fill_z_body(g, f3, i0, ni, jlo + (int)blockIdx.y); }
// both in one launch: they touch disjoint cells and neither reads what the other writes.
// grid: (ceil(ni/256), Nz + 1 + rows of the z part)
__global__ void k_fill_yz(Grid g, Halo3 f3, Halo2 f2, int i0, int ni, int jlo) {
  const int b = blockIdx.y;
  if (b <= g.Nz) fill_y_body(g, f3, f2, i0, ni, b);
  else fill_z_body(g, f3, i0, ni, jlo + b - (g.Nz + 1));
}
// periodic x for one array of `rows` parent rows: thread = (q in 0..2H-1, row)
__device__ __forceinline__ void periodic_row(const Grid& g, real* c, long row, int q) {
  real* r = c + row * g.sx;
  if (q < g.H) r[q] = r[g.Nx + q];                  // west halo <- east interior
  else r[g.Nx + q] = r[q];                          // east halo (index H+Nx+(q-H)) <- west interior
}
// grid: (ceil(rows_max*2H/256), 4 + n2d)
__global__ void k_fill_x(Grid g, Halo3 f3, Halo2 f2, int rows_c, int rows_v) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  int q = t % (2 * g.H);
  long row = t / (2 * g.H);
  int f = blockIdx.y;
  if (f < 4) {
    if (f < f3.n) {
      long rows = f3.is_v[f] ? rows_v : rows_c;
      if (row < rows) periodic_row(g, f3.p[f], row, q);
    }
  } else {
    int s = f - 4;
    if (s < f2.n) {
      long rows = f2.is_v[s] ? g.sy_v : g.sy_c;
      if (row < rows) periodic_row(g, f2.p[s], row, q);
    }
  }
}

// Zipper fold along the northern edge of the tripolar grid: Oceananigans' fold_north_{center,face}_{center,face}!, restated
// from memory of v0.96 [UPSTREAM-UNVERIFIED] (oracle/gb25_oracle.c: fold_rows_levels).  The fold pivots on the CENTRES of the
// last row of cells, Ny-1 here (0-based), between the two poles, which sit on the x faces 0 and Nx/2:
//   cell (i, Ny-1+q)   <-  s  cell (Nx-1-i, Ny-1-q)          x face (i, Ny-1+q)  <-  s' x face ((Nx-i) mod Nx, Ny-1-q)
//   y face (i, Ny-1+q) <-  s  y face (Nx-1-i, Ny-q)          q = 1..H;  s = -1 for vector components; s' = s except on the
//                                                            x face that wraps (i = 0), which keeps its sign
// Row Ny-1 is held twice -- cell (i, Ny-1) IS cell (Nx-1-i, Ny-1) ("the Ny line is duplicated") -- and both copies are
// stepped; neither is overwritten with the other (as the fill functions of v0.96 are recalled; a later upstream fix that
// slaves one copy to the other is NOT restated: it would need the partner's pivot row BEFORE the barotropic corrector).
// The bottom / top layer of the rows written is filled too (the hydrostatic integral of the rows beyond the fold starts in the
// top halo level).  Reads interior rows and levels only: independent of the y / z fill, before the periodic x copy.
__device__ __forceinline__ int fold_src_column(int ig, int Nxg, bool xface) { return xface ? (ig == 0 ? 0 : Nxg - ig) : Nxg - 1 - ig; }
__device__ __forceinline__ real fold_sign(int ig, bool xface, bool neg) { return (neg && !(xface && ig == 0)) ? -real(1.) : real(1.); }
// grid: (ceil(Nx/256), H, Nz + 2 | 1); blockIdx.y = q - 1; blockIdx.z = level + 1 of the 3-D fields, the last slice does the 2-D fields
__global__ void k_fill_fold(Grid g, Halo3 f3, Halo2 f2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= g.Nx) return;
  // q = 1 .. H; with option FOLD_PIVOT_SLAVED one more slice, q = 0: the pivot row itself -- held twice, cell (i, Ny-1) IS cell
  // (Nx-1-i, Ny-1) -- whose eastern half becomes the image of its western half (cells i >= Nx/2; x faces i > Nx/2: the face
  // Nx/2 is a pole and its own image).  [the later upstream fix as recalled; UPSTREAM-UNVERIFIED]
  const int q = (int)blockIdx.y + 1 - g.cv.pivot_slaved;
  const bool twod = (int)blockIdx.z == (f3.n ? g.Nz + 2 : 0);
  const int k = (int)blockIdx.z - 1, ks = min(max(k, 0), g.Nz - 1);
  const int n = twod ? f2.n : f3.n;
  for (int f = 0; f < n; f++) {
    real* c = twod ? f2.p[f] : f3.p[f];
    const bool is_v = twod ? f2.is_v[f] : f3.is_v[f], xf = twod ? f2.xf[f] : f3.xf[f], neg = twod ? f2.neg[f] : f3.neg[f];
    if (q == 0 && (is_v || i < g.Nx / 2 + (xf ? 1 : 0))) continue;
    const int isrc = fold_src_column(i, g.Nx, xf);
    const real sg = fold_sign(i, xf, neg);
    const int jd = g.Ny - 1 + q, js = is_v ? g.Ny - q : g.Ny - 1 - q;
    if (twod) c[i2(g, i, jd)] = sg * c[i2(g, isrc, js)];
    else if (is_v) c[iv(g, i, jd, k)] = sg * c[iv(g, isrc, js, ks)];
    else c[ic(g, i, jd, k)] = sg * c[ic(g, isrc, js, ks)];
  }
}

// The same fold on a slab of a decomposition: the cells beyond the fold are the images of cells of the PARTNER rank
// P-1-r (mirrored in x), which sends the H rows south of its pivot row -- all its parent columns, every interior level --
// and receives ours.  Buffer layout per field: [level][q - 1][parent column], q = 1 .. H: cell rows Ny-1-q, y-face rows
// Ny-q; 3-D fields first, then the 2-D ones.
struct FoldFields {
  real* p[9];
  int is_v[9], xf[9], neg[9], nz[9];   // nz: interior levels (1: a 2-D field)
  long off[9];                          // element offset of the field in the exchange buffer
  int n;
};
// grid: (ceil(sx/256), H, sum of nz)
__global__ void k_fold_pack(Grid g, FoldFields F, real* __restrict__ buf) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y + 1;
  if (a >= g.sx) return;
  int f = 0, k = blockIdx.z;
  while (k >= F.nz[f]) k -= F.nz[f++];
  const bool twod = F.nz[f] == 1;
  const int row = (F.is_v[f] ? g.Ny - q : g.Ny - 1 - q) + g.H;
  const long pl = F.is_v[f] ? g.pl_v : g.pl_c;
  buf[F.off[f] + ((long)k * g.H + (q - 1)) * g.sx + a] = F.p[f][a + (long)g.sx * row + (twod ? 0 : pl * (k + g.H))];
}
// grid: (ceil(sx/256), H, sum of (nz + 2 | 1)): the 3-D fields also get the bottom / top layer of the rows written.
// ig0: global column of local column 0; Nxg: global Nx (the x face that wraps keeps its sign).
__global__ void k_fold_unpack(Grid g, FoldFields F, const real* __restrict__ buf, int ig0, int Nxg) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y + 1;
  if (a >= g.sx) return;
  int f = 0, kk = blockIdx.z;
  while (kk >= (F.nz[f] == 1 ? 1 : F.nz[f] + 2)) kk -= (F.nz[f] == 1 ? 1 : F.nz[f] + 2), f++;
  const bool twod = F.nz[f] == 1, is_v = F.is_v[f] != 0, xf = F.xf[f] != 0;
  const int k = twod ? 0 : kk - 1, ks = twod ? 0 : min(max(k, 0), g.Nz - 1);
  const int am = xf ? g.sx - a : g.sx - 1 - a;             // the partner's parent column of the mirrored cell / face
  if (am >= g.sx) return;                                   // (the westernmost x face of the halo: never read)
  int ig = ig0 + a - g.H;
  ig = ((ig % Nxg) + Nxg) % Nxg;
  const real sg = fold_sign(ig, xf, F.neg[f] != 0);
  const long pl = is_v ? g.pl_v : g.pl_c;
  const int jd = g.Ny - 1 + q + g.H;                        // destination parent row
  F.p[f][a + (long)g.sx * jd + (twod ? 0 : pl * (k + g.H))] = sg * buf[F.off[f] + ((long)ks * g.H + (q - 1)) * g.sx + am];
}

// The split-explicit sub-cycle on a folded grid runs on TALL arrays: the barotropic work arrays (widened in x on a slab) with
// Wy more rows beyond the pivot row, filled ONCE per step with the images of the rows south of it, as Oceananigans extends
// the halo of its free surface on the TripolarGrid to the number of substeps: eta, U, V and the forcing G.U, G.V; then the
// Ns substeps need nothing from beyond the fold (what the missing neighbour of the last row spoils moves one row per substep
// and never reaches the pivot row).  The images' sources belong to the partner rank P-1-r on a slab (pack -> exchange ->
// unpack); a single domain packs and unpacks its own buffer.  Buffer: [array 0..4][q = 1..Wy+1][array column].
struct TallRows {
  real* p[5];          // eta, U, V, G.U, G.V (geometry: pitch sx, first row -H)
  int sx, xo, yo, Wy;  // pitch, array column of i = 0, array row of j = 0, rows beyond the pivot row (the y-face arrays take one more)
  int wrap;            // single domain: the x face beyond the last column is face 0
};
// grid: (ceil(sx/256), Wy + 1, 5)
template <bool PACK>
__global__ void k_tall_rows(Grid g, TallRows T, real* __restrict__ buf, int ig0, int Nxg) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, q = blockIdx.y + 1, f = blockIdx.z;
  if (a >= T.sx) return;
  const bool is_v = f == 2 || f == 4, xf = f == 1 || f == 3;
  if (q > T.Wy && !is_v) return;
  real* c = T.p[f];
  const long bo = ((long)f * (T.Wy + 1) + (q - 1)) * T.sx;
  if (PACK) {
    buf[bo + a] = c[a + (long)T.sx * ((is_v ? g.Ny - q : g.Ny - 1 - q) + T.yo)];
    return;
  }
  int am = xf ? T.sx - a : T.sx - 1 - a;   // the source's array column in the sender's (mirrored) layout
  if (xf && T.wrap) am = (a == T.xo) ? T.xo : 2 * T.xo + g.Nx - a;   // single domain (pitch Nx + 2 xo): face i <- face (Nx - i) mod Nx
  else if (!xf && T.wrap) am = 2 * T.xo + g.Nx - 1 - a;
  if (am < 0 || am >= T.sx) return;
  int ig = ig0 + a - T.xo;
  ig = ((ig % Nxg) + Nxg) % Nxg;
  c[a + (long)T.sx * (g.Ny - 1 + q + T.yo)] = fold_sign(ig, xf, f != 0) * buf[bo + am];
}

// ---------------------------------------------------------------------------------------------
// All three fills in ONE launch.  The periodic x copy has to see the y / z layers of its source columns; instead of
// waiting for them it reads what they are made FROM: a cell of a y layer is the neighbouring interior row (or zero on a
// wall face of v), a cell of a z layer the neighbouring interior level, everything else is copied as it stands.  No
// cell that one part reads is written by another, so the parts need no order.
// 1-D grid: the first nb_yz blocks do the y and z layers (as k_fill_yz with nbx blocks per row of blocks), the rest
// the x copy (nbr blocks per field).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ real x_source_3d(const Grid& g, const real* c, bool is_v, int isrc, int jj, int kk) {
  // value at parent column isrc, parent row jj, parent level kk AFTER the y and z fills
  const int sy = is_v ? g.sy_v : g.sy_c, pl = is_v ? g.pl_v : g.pl_c;
  const int j = jj - g.H, k = kk - g.H;
  int js = jj, ks = kk;
  if (k >= 0 && k < g.Nz) {
    if (is_v) {
      if (j == 0 || j == g.Ny) return real(0.);
    } else {
      if (j == -1) js = g.H;
      else if (j == g.Ny) js = g.H + g.Ny - 1;
    }
  } else if ((k == -1 || k == g.Nz) && j >= 0 && j < g.Ny) {
    if (is_v && j == 0) return real(0.);
    ks = (k == -1) ? g.H : g.H + g.Nz - 1;
  }
  (void)sy;
  return c[isrc + g.sx * js + pl * ks];
}
__device__ __forceinline__ real x_source_2d(const Grid& g, const real* c, bool is_v, int isrc, int jj) {
  const int j = jj - g.H;
  int js = jj;
  if (is_v) {
    if (j == 0 || j == g.Ny) return real(0.);
  } else {
    if (j == -1) js = g.H;
    else if (j == g.Ny) js = g.H + g.Ny - 1;
  }
  return c[isrc + g.sx * js];
}
__global__ void k_fill_fused(Grid g, Halo3 f3, Halo2 f2, int nbx, int nb_yz, int nbr, int rows_c, int rows_v) {
  const int b = blockIdx.x;
  if (b < nb_yz) {
    const int by = b / nbx;
    const int i = (b - by * nbx) * blockDim.x + threadIdx.x;
    if (i >= g.Nx) return;
    // (fill_y_body / fill_z_body take their column from blockIdx.x: do the same work inline)
    if (by <= g.Nz) {
      const int k = by;
      if (k < g.Nz) {
        for (int q = 0; q < f3.n; q++) {
          real* c = f3.p[q];
          if (f3.is_v[q]) {
            c[iv(g, i, 0, k)] = real(0.);
            c[iv(g, i, g.Ny, k)] = real(0.);
          } else {
            c[ic(g, i, -1, k)] = c[ic(g, i, 0, k)];
            c[ic(g, i, g.Ny, k)] = c[ic(g, i, g.Ny - 1, k)];
          }
        }
      } else {
        for (int q = 0; q < f2.n; q++) {
          real* c = f2.p[q];
          if (f2.is_v[q]) {
            c[i2(g, i, 0)] = real(0.);
            c[i2(g, i, g.Ny)] = real(0.);
          } else {
            c[i2(g, i, -1)] = c[i2(g, i, 0)];
            c[i2(g, i, g.Ny)] = c[i2(g, i, g.Ny - 1)];
          }
        }
      }
    } else {
      const int j = by - (g.Nz + 1);
      for (int q = 0; q < f3.n; q++) {
        real* c = f3.p[q];
        if (f3.is_v[q]) {
          c[iv(g, i, j, -1)] = (j == 0) ? real(0.) : c[iv(g, i, j, 0)];
          c[iv(g, i, j, g.Nz)] = (j == 0) ? real(0.) : c[iv(g, i, j, g.Nz - 1)];
        } else {
          c[ic(g, i, j, -1)] = c[ic(g, i, j, 0)];
          c[ic(g, i, j, g.Nz)] = c[ic(g, i, j, g.Nz - 1)];
        }
      }
    }
    return;
  }
  // ---- periodic x copy, reading through the y / z fills
  const int bb = b - nb_yz;
  const int f = bb / nbr;
  const long t = (long)(bb - f * nbr) * blockDim.x + threadIdx.x;
  const int q = (int)(t % (2 * g.H));
  const long row = t / (2 * g.H);
  const int idst = (q < g.H) ? q : g.Nx + q;               // parent column written
  const int isrc = (q < g.H) ? g.Nx + q : q;               // parent column it is the periodic image of
  if (f < 4) {
    if (f >= f3.n) return;
    const bool is_v = f3.is_v[f] != 0;
    const long rows = is_v ? rows_v : rows_c;
    if (row >= rows) return;
    const int sy = is_v ? g.sy_v : g.sy_c;
    const int kk = (int)(row / sy), jj = (int)(row - (long)kk * sy);
    f3.p[f][idst + row * g.sx] = x_source_3d(g, f3.p[f], is_v, isrc, jj, kk);
  } else {
    const int s2 = f - 4;
    if (s2 >= f2.n) return;
    const bool is_v = f2.is_v[s2] != 0;
    if (row >= (is_v ? g.sy_v : g.sy_c)) return;
    f2.p[s2][idst + row * g.sx] = x_source_2d(g, f2.p[s2], is_v, isrc, (int)row);
  }
}

// =============================================================================================
// compute_auxiliaries!: w from continuity and the hydrostatic pressure anomaly
// (GB-25 src/precompile.jl:113-115).  One thread per column on the extended range
// [-H+1, N+H-2] so that w and p are valid in the halos without any exchange.
// =============================================================================================
// Columns [i_first, i_first + n_a) and, after them, [i_first_b, i_first_b + n_b): the whole extended range in one piece,
// or (slab of a decomposition) the own columns while the x-halo bundle travels and the two edge strips afterwards.
// CURV: orthogonal curvilinear grid, the four face lengths of the column from the 2-D metric arrays.
// LAZY: u, v in memory lack the barotropic correction of this step; it is added on the fly (LazyCorr, below).
template <bool CURV, bool LAZY = false>
__global__ __launch_bounds__(256) void k_compute_w(Grid g, const real* __restrict__ u, const real* __restrict__ v,
                                                   real* __restrict__ w, int i_first, int n_a, int i_first_b, int n_b,
                                                   LazyCorr lz) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t < n_a ? i_first + t : i_first_b + (t - n_a);
  int j = blockIdx.y * blockDim.y + threadIdx.y - g.H + 1;
  if (t >= n_a + n_b || j > g.Ny + g.H - 2) return;
  const int o2 = i2(g, i, j);
  const real dxs = CURV ? g.cv.dxcf[o2] : g.dxf[j], dxn = CURV ? g.cv.dxcf[o2 + g.sx] : g.dxf[j + 1];
  const real raz = CURV ? g.cv.razcc[o2] : g.razc[j];
  const real dyw = CURV ? g.cv.dyfc[o2] : g.dy, dye = CURV ? g.cv.dyfc[o2 + 1] : g.dy;
  int o = ic(g, i, j, 0), ov = iv(g, i, j, 0);
  real wk = real(0.);
  w[o] = real(0.);
  real due = real(0.), duw = real(0.), dvn = real(0.), dvs = real(0.);
  if (LAZY) {
    due = lz.du[o2 + 1]; duw = lz.du[o2];
    dvn = lz.dv[o2 + g.sx]; dvs = lz.dv[o2];
  }
#pragma unroll 8
  for (int k = 0; k < g.Nz; k++) {
    real dz = uniform_at(g.dzc, k);
    real div = LAZY ? (dye * dz * (u[o + 1] + due) - dyw * dz * (u[o] + duw)) + (dxn * dz * (v[ov + g.sx] + dvn) - dxs * dz * (v[ov] + dvs))
                    : (dye * dz * u[o + 1] - dyw * dz * u[o]) + (dxn * dz * v[ov + g.sx] - dxs * dz * v[ov]);
    wk = wk - div * raz;
    o += g.pl_c;
    ov += g.pl_v;
    w[o] = wk;
  }
}

// Hydrostatic pressure anomaly p'[k] = p'[k+1] - Iz(b)[k+1] dz^f[k+1], b = -g rho'(T,S,z)/rho0 (TEOS-10).
// The state stays fp32, but the equation of state, the vertical integral AND the horizontal differences the
// momentum tendencies need run in fp64: rho ~ 1e3 kg/m3 has an fp32 ulp of 1.2e-4 kg/m3 (after integration ~1e-3
// of the pressure-gradient signal), and p' ~ 1e2..1e3 m2/s2 stored as fp32 still costs ~1e-4 of dp/dx on a
// quarter-degree grid.  So the kernel emits, besides the fp32 pHY' field of the API, the two differences
// dpx = p'(i)-p'(i-1) and dpy = p'(j)-p'(j-1) (small numbers, fp32 storage is then harmless).
// Mapping: a wave is 64 consecutive columns of which lane 0 only feeds lane 1's west difference (tiles advance by
// 63); a thread marches R = 4 adjacent rows plus the row south of them at once, which also gives the fp64 EOS
// chains 5-way instruction-level parallelism.  (v_fma_f64 issues at the full rate on gfx950 -- tools/micro/pk_rate.hip -- the kernel's time is the latency of its chains); the depth dependence of the
// 55-term polynomial is folded per level on the host (28 fp64 FMAs per evaluation).
constexpr int PR = 4;   // rows per thread of the full-range launch
// Columns [i_first, i_last] are written (whole extended range: -H+1 .. Nx+H-2; a slab of a decomposition does
// its interior early and the strips next to the x halos once those have arrived).
// PR_: rows per thread.  4 for the full range (five evaluations per level share one helper row); 1 for the narrow launches of
// a decomposition, which have too few waves to hide latency behind each other and want short per-level chains (the narrowest of
// them on the lat-lon grid take k_compute_p_tile below).
// WRITE_P = false: only the two differences the momentum kernel consumes are stored; pHY' itself is a diagnostic
// that the host side then materialises on demand (gb25_api.hip, phy_stale).
#ifndef GB25_P_MINW
#define GB25_P_MINW 1   // (tools/build_variant.sh: waves per SIMD the register allocator of the pressure kernel is held to)
#endif
template <int PR_, bool WRITE_P>
__global__ __launch_bounds__(256, GB25_P_MINW) void k_compute_p(Grid g, const real* __restrict__ T, const real* __restrict__ S,
                                                   real* __restrict__ p, real* __restrict__ dpx,
                                                   real* __restrict__ dpy, int i_first, int i_last, int i_first_b,
                                                   int i_last_b, int tiles_a, real* __restrict__ n2, int j_first, int j_last) {
  // rows [j_first, j_last] are written (the whole extended range: -H+1 .. Ny+H-2; a rank of a 2-D decomposition redoes row 0
  // -- whose y difference reads the southern neighbour's row -- once that row has arrived)
  // (n2: unused since round 4 -- CATKE's N^2 is SeawaterBuoyancy's dz_b, alpha dzT - beta dzS: k_catke_n2)
  // an optional second column range [i_first_b, i_last_b] takes the tiles from tiles_a on (both strips of a slab in
  // one launch)
  const int lane = threadIdx.x;
  const bool second = (int)blockIdx.x >= tiles_a;
  if (second) {
    i_first = i_first_b;
    i_last = i_last_b;
  }
  const int i = i_first - 1 + ((int)blockIdx.x - (second ? tiles_a : 0)) * 63 + lane;   // lane 0: helper column
  const int jb = j_first + (blockIdx.y * blockDim.y + threadIdx.y) * PR_;     // first of this thread's PR rows
  const int imax = i_last, jmax = j_last;
  if (jb > jmax) return;                                                    // whole wave leaves together
  const int ic_ = min(i, g.Nx + g.H - 1);                                    // clamp: addresses stay in the parent
  const int Nz = g.Nz;
  const double gr = -(double)g.g / (double)g.rho0;
  const double sc = 0.875 / 35.16504;
  int o[PR_ + 1];
  double bup[PR_ + 1], pk[PR_ + 1];
#pragma unroll
  for (int r = 0; r <= PR_; r++) {
    const int j = min(jb - 1 + r, g.Ny + g.H - 1);    // r = 0 is the helper row south of the thread's rows
    o[r] = ic(g, ic_, j, Nz);
    // b in the first halo cell above the surface: mirrored geopotential height (table row Nz)
    bup[r] = gr * teos10_level(g.eos + 28 * Nz, sqrt_pos(((double)S[o[r]] + 32.0) * sc), (double)T[o[r]] * 0.025);
    pk[r] = 0.0;
  }
  for (int k = Nz - 1; k >= 0; k--) {
    const double* c = g.eos + 28 * k;
    const double dz = g.dzf_d[k + 1];
#pragma unroll
    for (int r = 0; r <= PR_; r++) {
      o[r] -= g.pl_c;
      double bk = gr * teos10_level(c, sqrt_pos(((double)S[o[r]] + 32.0) * sc), (double)T[o[r]] * 0.025);
      pk[r] = pk[r] - 0.5 * (bk + bup[r]) * dz;
      bup[r] = bk;
    }
#pragma unroll
    for (int r = 1; r <= PR_; r++) {
      const double pw = __shfl_up(pk[r], 1);
      const int j = jb - 1 + r;
      if (lane >= 1 && i <= imax && j <= jmax) {
        if (WRITE_P) p[o[r]] = (real)pk[r];
        dpx[o[r]] = (real)(pk[r] - pw);
        dpy[o[r]] = (real)(pk[r] - pk[r - 1]);
      }
    }
  }
}
// The same pressure for the narrow launches of a decomposition (a slab's own columns, the strips next to its halos), which have
// too few waves to hide the latency of their chains: a wave is a tile of 16 columns x 4 rows -- lane = 16 ly + lx -- whose first
// column and first row are the helpers of the two differences, both fetched by lane shuffles; ONE evaluation per thread and
// level, 1.42 per written cell (45 of 64 lanes write) against 2.03 for the one-row form of k_compute_p (a helper row per
// thread), and 1.4 x the waves.  Same operands, same operations per cell: the same bits.
template <bool WRITE_P>
__global__ __launch_bounds__(256) void k_compute_p_tile(Grid g, const real* __restrict__ T, const real* __restrict__ S,
                                                        real* __restrict__ p, real* __restrict__ dpx, real* __restrict__ dpy,
                                                        int i_first, int i_last, int i_first_b, int i_last_b, int tiles_a,
                                                        real* __restrict__ n2, int j_first, int j_last) {
  const int lane = threadIdx.x, lx = lane & 15, ly = lane >> 4;
  const bool second = (int)blockIdx.x >= tiles_a;
  if (second) {
    i_first = i_first_b;
    i_last = i_last_b;
  }
  const int i = i_first - 1 + ((int)blockIdx.x - (second ? tiles_a : 0)) * 15 + lx;            // lx = 0: helper column
  const int j = j_first - 1 + (int)(blockIdx.y * blockDim.y + threadIdx.y) * 3 + ly;           // ly = 0: helper row
  if (j_first + (int)(blockIdx.y * blockDim.y + threadIdx.y) * 3 > j_last) return;             // whole wave leaves together
  const int Nz = g.Nz;
  const double gr = -(double)g.g / (double)g.rho0;
  const double sc = 0.875 / 35.16504;
  int o = ic(g, min(i, g.Nx + g.H - 1), min(j, g.Ny + g.H - 1), Nz);                           // clamp: addresses stay in the parent
  const bool writes = lx >= 1 && ly >= 1 && i <= i_last && j <= j_last;
  // b in the first halo cell above the surface: mirrored geopotential height (table row Nz)
  double bup = gr * teos10_level(g.eos + 28 * Nz, sqrt_pos(((double)S[o] + 32.0) * sc), (double)T[o] * 0.025);
  double pk = 0.0;
  for (int k = Nz - 1; k >= 0; k--) {
    const double* c = g.eos + 28 * k;
    const double dz = g.dzf_d[k + 1];
    o -= g.pl_c;
    const double bk = gr * teos10_level(c, sqrt_pos(((double)S[o] + 32.0) * sc), (double)T[o] * 0.025);
    pk = pk - 0.5 * (bk + bup) * dz;
    bup = bk;
    const double pw = __shfl_up(pk, 1), ps = __shfl_up(pk, 16);
    if (writes) {
      if (WRITE_P) p[o] = (real)pk;
      dpx[o] = (real)(pk - pw);
      dpy[o] = (real)(pk - ps);
    }
  }
}
// GB25_OPT_PRESSURE_PRECISION = 32: the hydrostatic pressure exactly as a model whose float type is `real` evaluates
// it -- the 55-term TEOS-10 polynomial at every cell in `real`, Horner in (t, s) then in zeta, no contraction into FMAs,
// the integral and the stored pHY' in `real`, the differences taken from the stored values.  This is the arithmetic of
// the all-Float32 restatement (oracle/gb25_oracle.c: teos10_rho, compute_p), operation for operation; it exists so
// that the distance of the default (fp64-pressure) path from a Float32 reference run can be separated into "the
// reference's own round-off" and "everything else" (tests/test_gpu_fp32_story.py).
#pragma clang fp contract(off)
__device__ __forceinline__ real teos10_rho_literal(real Theta, real Sa, real Z) {
  const real t = Theta * real(0.025);
  const real s = (real)sqrt((double)((Sa + real(32.0)) * real(0.875 / 35.16504)));
  const real z = -Z * real(1e-4);
  const real R000 = 8.0189615746e+02, R100 = 8.6672408165e+02, R200 = -1.7864682637e+03, R300 = 2.0375295546e+03,
             R400 = -1.2849161071e+03, R500 = 4.3227585684e+02, R600 = -6.0579916612e+01, R010 = 2.6010145068e+01,
             R110 = -6.5281885265e+01, R210 = 8.1770425108e+01, R310 = -5.6888046321e+01, R410 = 1.7681814114e+01,
             R510 = -1.9193502195e+00, R020 = -3.7074170417e+01, R120 = 6.1548258127e+01, R220 = -6.0362551501e+01,
             R320 = 2.9130021253e+01, R420 = -5.4723692739e+00, R030 = 2.1661789529e+01, R130 = -3.3449108469e+01,
             R230 = 1.9717078466e+01, R330 = -3.1742946532e+00, R040 = -8.3627885467e+00, R140 = 1.1311538584e+01,
             R240 = -5.3563304045e+00, R050 = 5.4048723791e-01, R150 = 4.8169980163e-01, R060 = -1.9083568888e-01,
             R001 = 1.9681925209e+01, R101 = -4.2549998214e+01, R201 = 5.0774768218e+01, R301 = -3.0938076334e+01,
             R401 = 6.6051753097e+00, R011 = -1.3336301113e+01, R111 = -4.4870114575e+00, R211 = 5.0042598061e+00,
             R311 = -6.5399043664e-01, R021 = 6.7080479603e+00, R121 = 3.5063081279e+00, R221 = -1.8795372996e+00,
             R031 = -2.4649669534e+00, R131 = -5.5077101279e-01, R041 = 5.5927935970e-01, R002 = 2.0660924175e+00,
             R102 = -4.9527603989e+00, R202 = 2.5019633244e+00, R012 = 2.0564311499e+00, R112 = -2.1311365518e-01,
             R022 = -1.2419983026e+00, R003 = -2.3342758797e-02, R103 = -1.8507636718e-02, R013 = 3.7969820455e-01;
  const real R00 = 4.6494977072e+01, R01 = -5.2099962525e+00, R02 = 2.2601900708e-01, R03 = 6.4326772569e-02,
             R04 = 1.5616995503e-02, R05 = -1.7243708991e-03;
  real r3 = R013 * t + R103 * s + R003;
  real r2 = (R022 * t + R112 * s + R012) * t + (R202 * s + R102) * s + R002;
  real r1 = (((R041 * t + R131 * s + R031) * t + (R221 * s + R121) * s + R021) * t + ((R311 * s + R211) * s + R111) * s + R011) * t +
            (((R401 * s + R301) * s + R201) * s + R101) * s + R001;
  real r0 = (((((R060 * t + R150 * s + R050) * t + (R240 * s + R140) * s + R040) * t + ((R330 * s + R230) * s + R130) * s + R030) * t +
              (((R420 * s + R320) * s + R220) * s + R120) * s + R020) * t +
             ((((R510 * s + R410) * s + R310) * s + R210) * s + R110) * s + R010) * t +
            (((((R600 * s + R500) * s + R400) * s + R300) * s + R200) * s + R100) * s + R000;
  real rp = ((r3 * z + r2) * z + r1) * z + r0;
  real rz = (((((R05 * z + R04) * z + R03) * z + R02) * z + R01) * z + R00) * z;
  return rz + rp;
}
// one thread per column of the extended range; p is always stored (the differences are taken from the stored values)
__global__ __launch_bounds__(256) void k_compute_p_literal(Grid g, const real* __restrict__ T, const real* __restrict__ S,
                                                           real* __restrict__ p, int i_first, int n_a, int i_first_b,
                                                           int n_b) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int i = t < n_a ? i_first + t : i_first_b + (t - n_a);
  const int j = blockIdx.y * blockDim.y + threadIdx.y - g.H + 1;
  if (t >= n_a + n_b || j > g.Ny + g.H - 2) return;
  const int Nz = g.Nz;
  auto buoy = [&](int o, real Z) {
    real rho = teos10_rho_literal(T[o], S[o], Z);
    return -(g.g * (rho - g.rho0)) / g.rho0;
  };
  int o = ic(g, i, j, Nz);
  real bup = buoy(o, uniform_at(g.zc, Nz - 1) - real(1.) * uniform_at(g.dzf, Nz - 1));   // mirrored height of the first cell above the surface
  o -= g.pl_c;
  real bk = buoy(o, uniform_at(g.zc, Nz - 1));
  real pk = -((bk + bup) / real(2.)) * uniform_at(g.dzf, Nz);
  p[o] = pk;
  for (int k = Nz - 2; k >= 0; k--) {
    o -= g.pl_c;
    bup = bk;
    bk = buoy(o, uniform_at(g.zc, k));
    pk = pk - ((bk + bup) / real(2.)) * uniform_at(g.dzf, k + 1);
    p[o] = pk;
  }
}
#pragma clang fp contract(fast)

// the same two differences from an fp32 pHY' uploaded by the host (set_field): keeps the arrays consistent
__global__ void k_pressure_differences(Grid g, const real* __restrict__ p, real* __restrict__ dpx,
                                       real* __restrict__ dpy, long n) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < g.sx || t >= n) return;
  dpx[t] = p[t] - p[t - 1];
  dpy[t] = p[t] - p[t - g.sx];
}

// =============================================================================================
// Tracer tendencies: G_c = -div(U c), WENO(order=5) upwind-biased flux form
// (compute_hydrostatic_free_surface_Gc!, GB-25 src/precompile.jl:75-111).  T and S in one pass.
// =============================================================================================
__device__ __forceinline__ real tracer_div(const Grid& g, const real* __restrict__ c, int o, real Ax, real uw,
                                            real ue, real Ays, real Ayn, real vs, real vn, real Az, real wb,
                                            real wt, int oys, int oyn, int ozb, int ozt) {
  real q[7];
#pragma unroll
  for (int m = 0; m < 7; m++) q[m] = c[o + m - 3];
  real fw = Ax * uw * biased6<false>(5, uw > real(0.), q, q, q);
  real fe = Ax * ue * biased6<false>(5, ue > real(0.), q + 1, q + 1, q + 1);
#pragma unroll
  for (int m = 0; m < 7; m++) q[m] = c[o + (m - 3) * g.sx];
  real fs = Ays * vs * biased6<false>(oys, vs > real(0.), q, q, q);
  real fn = Ayn * vn * biased6<false>(oyn, vn > real(0.), q + 1, q + 1, q + 1);
#pragma unroll
  for (int m = 0; m < 7; m++) q[m] = c[o + (m - 3) * g.pl_c];
  real fb = Az * wb * biased6<false>(ozb, wb > real(0.), q, q, q);
  real ft = Az * wt * biased6<false>(ozt, wt > real(0.), q + 1, q + 1, q + 1);
  return (fe - fw) + (fn - fs) + (ft - fb);
}

__global__ __launch_bounds__(256) void k_tracer_tendencies(Grid g, const real* __restrict__ u,
                                                           const real* __restrict__ v, const real* __restrict__ w,
                                                           const real* __restrict__ T, const real* __restrict__ S,
                                                           real* __restrict__ GT, real* __restrict__ GS, int nbx,
                                                           int nb) {
  TileIdx t = tile_index(g, nbx, nb);
  if (!t.ok) return;
  const int i = t.i, j = t.j, k = t.k;
  const int o = ic(g, i, j, k), ov = iv(g, i, j, k);
  const real dz = uniform_at(g.dzc, k);
  const real Ax = g.dy * dz, Ays = g.dxf[j] * dz, Ayn = g.dxf[j + 1] * dz, Az = g.azc[j];
  const real uw = u[o], ue = u[o + 1], vs = v[ov], vn = v[ov + g.sx], wb = w[o], wt = w[o + g.pl_c];
  const int oys = biased_order_face(j - g.jws, g.jwn - g.jws), oyn = biased_order_face(j + 1 - g.jws, g.jwn - g.jws);
  const int ozb = biased_order_face(k, g.Nz), ozt = biased_order_face(k + 1, g.Nz);
  const real rV = g.razc[j] * uniform_at(g.rdzc, k);
  GT[o] = -(tracer_div(g, T, o, Ax, uw, ue, Ays, Ayn, vs, vn, Az, wb, wt, oys, oyn, ozb, ozt) * rV);
  GS[o] = -(tracer_div(g, S, o, Ax, uw, ue, Ays, Ayn, vs, vn, Az, wb, wt, oys, oyn, ozb, ozt) * rV);
}

// =============================================================================================
// Momentum tendencies (compute_hydrostatic_momentum_tendencies!, GB-25 src/precompile.jl:63-73):
// WENOVectorInvariant(order=5): vorticity flux upwinded with VelocityStencil smoothness,
// self-upwinded divergence flux and kinetic-energy gradient (OnlySelfUpwinding, cross terms
// centred 4th order), WENO5 vertical advection, enstrophy-conserving spherical Coriolis,
// hydrostatic pressure gradient.  The barotropic pressure gradient lives in the sub-cycle.
// =============================================================================================
__global__ __launch_bounds__(256) void k_gu(Grid g, const real* __restrict__ u, const real* __restrict__ v,
                                            const real* __restrict__ w, const real* __restrict__ dpx,
                                            real* __restrict__ Gu, int nbx, int nb) {
  TileIdx t = tile_index(g, nbx, nb);
  if (!t.ok) return;
  const int i = t.i, j = t.j, k = t.k;
  const int o = ic(g, i, j, k), ov = iv(g, i, j, k);
  const int sx = g.sx, pc = g.pl_c, pv = g.pl_v;
  const real dy = g.dy, dz = uniform_at(g.dzc, k);
  const real dxf_s = g.dxf[j], dxf_n = g.dxf[j + 1], rdxc_j = g.rdxc[j];

  // advecting v at (f,c,c)
  const real vhat =
      (real(0.5) * (dxf_s * v[ov - 1] + dxf_n * v[ov - 1 + sx]) + real(0.5) * (dxf_s * v[ov] + dxf_n * v[ov + sx])) * real(0.5) * rdxc_j;

  // vorticity at faces j-2 .. j+3 of column i, plus the VelocityStencil smoothness inputs
  real zq[6], uq[6], vq[6];
#pragma unroll
  for (int m = 0; m < 6; m++) {
    int jf = j - 2 + m;
    real vc = v[ov + (m - 2) * sx], vw = v[ov - 1 + (m - 2) * sx];
    real uc = u[o + (m - 2) * sx], us = u[o + (m - 3) * sx];
    zq[m] = ((dy * vc - dy * vw) - (g.dxc[jf] * uc - g.dxc[jf - 1] * us)) * g.razf[jf];
    uq[m] = real(0.5) * (us + uc);
    vq[m] = real(0.5) * (vw + vc);
  }
  const real zetaR = biased6<true>(biased_order_center(j - g.jws, g.jwn - g.jws), vhat > real(0.), zq, uq, vq);
  const real hadv = -vhat * zetaR;

  // self-upwinded divergence flux
  const real uhat = u[o];
  real u7[7];
#pragma unroll
  for (int m = 0; m < 7; m++) u7[m] = u[o + m - 3];
  const real Ax = dy * dz, Ays = dxf_s * dz, Ayn = dxf_n * dz;
  real Du[6], Dv[6], Dd[6];
#pragma unroll
  for (int m = 0; m < 6; m++) {
    Du[m] = Ax * u7[m + 1] - Ax * u7[m];
    Dv[m] = Ayn * v[ov + (m - 3) + sx] - Ays * v[ov + (m - 3)];
    Dd[m] = Du[m] + Dv[m];
  }
  const real dvs = sym_interp(true, Dv[1], Dv[2], Dv[3], Dv[4]);
  const real duR = biased6<false>(5, uhat > real(0.), Du, Dd, Dd);
  const real phi = uhat * (dvs + duR);

  // vertical advection of u
  const real Az = g.azc[j];
  real fz[2];
#pragma unroll
  for (int tt = 0; tt < 2; tt++) {
    int ow = o + tt * pc;
    real wt = sym_interp(true, Az * w[ow - 2], Az * w[ow - 1], Az * w[ow], Az * w[ow + 1]);
    real q[6];
#pragma unroll
    for (int m = 0; m < 6; m++) q[m] = u[o + (tt + m - 3) * pc];
    fz[tt] = wt * biased6<false>(biased_order_face(k + tt, g.Nz), wt > real(0.), q, q, q);
  }
  const real vadv = (phi + (fz[1] - fz[0])) * (g.razc[j] * uniform_at(g.rdzc, k));

  // Bernoulli head
  real Ku[6], su[6];
#pragma unroll
  for (int m = 0; m < 6; m++) {
    Ku[m] = real(0.5) * u7[m + 1] * u7[m + 1] - real(0.5) * u7[m] * u7[m];
    su[m] = real(0.5) * (u7[m] + u7[m + 1]);
  }
  const real dKu = biased6<false>(5, uhat > real(0.), Ku, su, su);
  real a4[4];
#pragma unroll
  for (int m = 0; m < 4; m++) {
    real vc = v[ov + (m - 1) * sx], vw = v[ov - 1 + (m - 1) * sx];
    a4[m] = real(0.5) * vc * vc - real(0.5) * vw * vw;
  }
  const real dKv = sym_interp(sym4_center(j - g.jws, g.jwn - g.jws), a4[0], a4[1], a4[2], a4[3]);
  const real bern = (dKu + dKv) * rdxc_j;

  const real cor = -real(0.5) * (g.fcor[j] + g.fcor[j + 1]) * vhat;
  const real dpdx = dpx[o] * rdxc_j;
  Gu[o] = -(hadv + vadv + bern) - cor - dpdx;
}

__global__ __launch_bounds__(256) void k_gv(Grid g, const real* __restrict__ u, const real* __restrict__ v,
                                            const real* __restrict__ w, const real* __restrict__ dpy,
                                            real* __restrict__ Gv, int nbx, int nb) {
  TileIdx t = tile_index(g, nbx, nb);
  if (!t.ok) return;
  const int i = t.i, j = t.j, k = t.k;
  const int o = ic(g, i, j, k), ov = iv(g, i, j, k);
  const int sx = g.sx, pc = g.pl_c, pv = g.pl_v;
  const real dy = g.dy, dz = uniform_at(g.dzc, k);

  // advecting u at (c,f,c)
  const real uhat = (real(0.5) * (dy * u[o - sx] + dy * u[o - sx + 1]) + real(0.5) * (dy * u[o] + dy * u[o + 1])) * real(0.5) * g.rdy;

  // vorticity at faces i-2 .. i+3 of row j
  const real dxc_j = g.dxc[j], dxc_s = g.dxc[j - 1], razf = g.razf[j];
  real zq[6], uq[6], vq[6];
#pragma unroll
  for (int m = 0; m < 6; m++) {
    real vc = v[ov + (m - 2)], vw = v[ov + (m - 3)];
    real uc = u[o + (m - 2)], us = u[o + (m - 2) - sx];
    zq[m] = ((dy * vc - dy * vw) - (dxc_j * uc - dxc_s * us)) * razf;
    uq[m] = real(0.5) * (us + uc);
    vq[m] = real(0.5) * (vw + vc);
  }
  const real zetaR = biased6<true>(5, uhat > real(0.), zq, uq, vq);
  const real hadv = uhat * zetaR;

  // self-upwinded divergence flux
  const real vhat = v[ov];
  real v7[7];
#pragma unroll
  for (int m = 0; m < 7; m++) v7[m] = v[ov + (m - 3) * sx];
  const real Ax = dy * dz;
  real Du[6], Dv[6], Dd[6];
#pragma unroll
  for (int m = 0; m < 6; m++) {
    int jc = j - 3 + m;
    Du[m] = Ax * u[o + 1 + (m - 3) * sx] - Ax * u[o + (m - 3) * sx];
    Dv[m] = g.dxf[jc + 1] * dz * v7[m + 1] - g.dxf[jc] * dz * v7[m];
    Dd[m] = Du[m] + Dv[m];
  }
  const int of = biased_order_face(j - g.jws, g.jwn - g.jws);
  const bool s4 = sym4_face(j - g.jws, g.jwn - g.jws);
  const real dus = sym_interp(s4, Du[1], Du[2], Du[3], Du[4]);
  const real dvR = biased6<false>(of, vhat > real(0.), Dv, Dd, Dd);
  const real phi = vhat * (dus + dvR);

  // vertical advection of v
  real fz[2];
#pragma unroll
  for (int tt = 0; tt < 2; tt++) {
    int ow = o + tt * pc;
    real wt = sym_interp(s4, g.azc[j - 2] * w[ow - 2 * sx], g.azc[j - 1] * w[ow - sx], g.azc[j] * w[ow],
                          g.azc[j + 1] * w[ow + sx]);
    real q[6];
#pragma unroll
    for (int m = 0; m < 6; m++) q[m] = v[ov + (tt + m - 3) * pv];
    fz[tt] = wt * biased6<false>(biased_order_face(k + tt, g.Nz), wt > real(0.), q, q, q);
  }
  const real vadv = (phi + (fz[1] - fz[0])) * (razf * uniform_at(g.rdzc, k));

  // Bernoulli head
  real Kv[6], sv[6];
#pragma unroll
  for (int m = 0; m < 6; m++) {
    Kv[m] = real(0.5) * v7[m + 1] * v7[m + 1] - real(0.5) * v7[m] * v7[m];
    sv[m] = real(0.5) * (v7[m] + v7[m + 1]);
  }
  const real dKv = biased6<false>(of, vhat > real(0.), Kv, sv, sv);
  real a4[4];
#pragma unroll
  for (int m = 0; m < 4; m++) {
    real un = u[o + (m - 1)], us = u[o + (m - 1) - sx];
    a4[m] = real(0.5) * un * un - real(0.5) * us * us;
  }
  const real dKu = sym_interp(true, a4[0], a4[1], a4[2], a4[3]);
  const real bern = (dKv + dKu) * g.rdy;

  const real cor = g.fcor[j] * uhat;
  const real dpdy = dpy[o] * g.rdy;
  Gv[ov] = -(hadv + vadv + bern) - cor - dpdy;
}

// =============================================================================================
// ab2_step!: velocities + vertically integrated AB2 tendencies (barotropic forcing), one thread
// per column (fuses ab2_step_field! x2 with _compute_integrated_ab2_tendencies!).
// =============================================================================================
__global__ __launch_bounds__(256) void k_ab2_velocities(Grid g, real* __restrict__ u, real* __restrict__ v,
                                                        const real* __restrict__ Gnu, const real* __restrict__ Gmu,
                                                        const real* __restrict__ Gnv, const real* __restrict__ Gmv,
                                                        real* __restrict__ GU, real* __restrict__ GV,
                                                        real* __restrict__ Usum, real* __restrict__ Vsum, real dt,
                                                        real chi, int kchunks) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  const bool urow = j < g.Ny;   // (zipper fold: the row of y faces on the fold line is stepped, v only)
  const real C1 = real(1.5) + chi, C2 = real(0.5) + chi;
  const real ne = (chi != -real(0.5)) ? real(1.) : real(0.);
  int o = ic(g, i, min(j, g.Ny - 1), 0), ov = iv(g, i, j, 0);
  // The column integrals are summed per chunk of levels and the chunk sums added in order: the association of
  // the momentum kernel's look-ahead (UvAhead, kernels_v2.hpp), whose blocks own one chunk of a column each, so
  // that both routes give the same bits.  Explicit FMAs for the same reason.
  const int klen = (g.Nz + kchunks - 1) / kchunks;
  real SU = real(0.), SV = real(0.), IU = real(0.), IV = real(0.);
  for (int k0 = 0; k0 < g.Nz; k0 += klen) {
    const int k1 = min(g.Nz, k0 + klen);
    real su = real(0.), sv = real(0.), iu = real(0.), iv_ = real(0.);
#pragma unroll 4
    for (int k = k0; k < k1; k++) {
      real dz = uniform_at(g.dzc, k);
      real gu = rfma(C1, Gnu[o], -((C2 * Gmu[o]) * ne));
      real gv = rfma(C1, Gnv[ov], -((C2 * Gmv[ov]) * ne));
      real un = rfma(dt, gu, u[o]), vn = rfma(dt, gv, v[ov]);
      if (urow) u[o] = un;
      v[ov] = vn;
      su = (k == k0) ? dz * gu : rfma(dz, gu, su);
      sv = (k == k0) ? dz * gv : rfma(dz, gv, sv);
      // column integrals of the UPDATED velocities: the barotropic corrector needs them after the sub-cycle and
      // they are in registers here (saves the corrector's first sweep over u and v)
      iu = (k == k0) ? dz * un : rfma(dz, un, iu);
      iv_ = (k == k0) ? dz * vn : rfma(dz, vn, iv_);
      o += g.pl_c;
      ov += g.pl_v;
    }
    SU = (k0 == 0) ? su : SU + su;
    SV = (k0 == 0) ? sv : SV + sv;
    IU = (k0 == 0) ? iu : IU + iu;
    IV = (k0 == 0) ? iv_ : IV + iv_;
  }
  const int o2 = i2(g, i, j);
  if (urow) {
    GU[o2] = SU;
    Usum[o2] = IU;
  }
  GV[o2] = (j == g.jws) ? real(0.) : SV;  // the wall face is a peripheral node
  Vsum[o2] = (j == g.jws) ? real(0.) : IV;  // v on the wall face is reset to zero by the halo fill before the corrector
}
// Second half of the velocity look-ahead: adds up the per-chunk column sums the momentum kernel left in P
// (layout [quantity 0..3][chunk][2-D parent]) in chunk order.
__global__ __launch_bounds__(256) void k_ab2_velocities_finish(Grid g, const real* __restrict__ P, int kchunks,
                                                               int plane2, real* __restrict__ GU,
                                                               real* __restrict__ GV, real* __restrict__ Usum,
                                                               real* __restrict__ Vsum) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  const int o2 = i2(g, i, j);
  real t[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const real* Pq = P + (long)q * kchunks * plane2 + o2;
    real a = Pq[0];
    for (int kc = 1; kc < kchunks; kc++) a = a + Pq[(long)kc * plane2];
    t[q] = a;
  }
  if (j < g.Ny) {
    GU[o2] = t[0];
    Usum[o2] = t[2];
  }
  GV[o2] = (j == g.jws) ? real(0.) : t[1];
  Vsum[o2] = (j == g.jws) ? real(0.) : t[3];
}

// =============================================================================================
// implicit_step!: vertically implicit diffusion after the explicit AB2 update (closure =
// VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), kappa, nu), GB-25 src/baroclinic_instability_model.jl:31;
// "batched tridiagonal per column").  (1 - dt d/dz K d/dz) phi = phi*, no flux through the bottom
// face of the column's first free level and through the top face:
//   lower_k = -dt K / (dz^c_k dz^f_k),  upper_k = -dt K / (dz^c_k dz^f_{k+1}),  diag_k = 1 - lower_k - upper_k.
// One thread per column and field, a PAIR of fields per launch (u with v, T with S; blockIdx.z).  Thomas algorithm; every
// global element is read once and written once -- 2 accesses per cell and field, the compulsory traffic.  The velocity
// launch also leaves the column integrals of the new u, v for the barotropic corrector, with the chunked association
// every other producer of those sums uses.  kind 0: (u, v); 1: (T, S).
// Two kernels: columns of up to 128 levels live in registers (k_implicit_vertical_reg, below: the one that runs at the
// sizes of BASELINE.json); deeper ones in LDS with per-thread elimination factors ([level][thread], conflict-free;
// dynamic LDS (2 blockDim.x + 2) Nz reals) -- this kernel.
// =============================================================================================
template <bool IMM>
__global__ __launch_bounds__(256) void k_implicit_vertical(Grid g, real* __restrict__ fa, real* __restrict__ fb, int kind,
                                                           real Ka, real Kb, real dt, real* __restrict__ sum_a,
                                                           real* __restrict__ sum_b, int kchunks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int T = blockDim.x, tid = threadIdx.x, Nz = g.Nz, f = blockIdx.z;
  // LDS: the column [Nz][T], its elimination factors [Nz][T], the per-level coupling tables [2][Nz]
  real* col = reinterpret_cast<real*>(lds_raw);
  real* gam = col + (size_t)Nz * T;
  real* cdn = gam + (size_t)Nz * T;    // cdn[k] = 1 / (dz^c_k dz^f_k): coupling of level k to the level below
  real* cup = cdn + Nz;                // cup[k] = 1 / (dz^c_k dz^f_{k+1}): to the level above
  for (int k = tid; k < Nz; k += T) {
    cdn[k] = real(1.) / (uniform_at(g.dzc, k) * uniform_at(g.dzf, k));
    cup[k] = real(1.) / (uniform_at(g.dzc, k) * uniform_at(g.dzf, k + 1));
  }
  __syncthreads();
  const int i = blockIdx.x * T + tid, j = blockIdx.y;
  const bool vsh = kind == 0 && f == 1;
  // rows: cells 0 .. Ny-1; y faces 1 .. Ny-1 and, with the zipper fold, the fold line Ny (the wall faces stay zero)
  if (i >= g.Nx || (vsh ? (j > g.Ny - 1) : (j >= g.Ny))) return;
  const int o2 = i2(g, i, j);
  int kf = 0;
  if (IMM) {
    const unsigned w = kind == 1 ? g.im.ordA[o2] : g.im.ordC[o2] >> (f == 0 ? 8 : 16);
    kf = min((int)(w & 255), Nz);
  }
  real* F = f ? fb : fa;
  real* S = f ? sum_b : sum_a;
  const real dK = dt * (f ? Kb : Ka);
  const bool solve = dK != real(0.) && kf < Nz && !(vsh && j == g.jws);
  if (!solve && S == nullptr) return;
  const int pl = vsh ? g.pl_v : g.pl_c, o0 = vsh ? iv(g, i, j, 0) : ic(g, i, j, 0);
#pragma unroll 8
  for (int k = 0; k < Nz; k++) col[k * T + tid] = F[o0 + k * pl];
  if (solve) {
    real rb = real(1.), p = real(0.);   // 1 / beta and the last eliminated value
    for (int k = kf; k < Nz; k++) {
      const real up = (k == Nz - 1) ? real(0.) : cup[k];
      const real lo = (k == kf) ? real(0.) : -(dK * cdn[k]);
      const real gk = (k == kf) ? real(0.) : -(dK * cup[k - 1]) * rb;       // gamma_k = upper_{k-1} / beta_{k-1}
      rb = rcp((real(1.) - lo + dK * up) - lo * gk);
      gam[k * T + tid] = gk;
      p = (col[k * T + tid] - lo * p) * rb;
      col[k * T + tid] = p;
    }
    // back substitution, the results leave for HBM as they are formed
    real nxt = col[(Nz - 1) * T + tid];
    F[o0 + (Nz - 1) * pl] = nxt;
    for (int k = Nz - 2; k >= kf; k--) {
      nxt = col[k * T + tid] - gam[(k + 1) * T + tid] * nxt;
      col[k * T + tid] = nxt;
      F[o0 + k * pl] = nxt;
    }
  }
  if (S != nullptr) {   // column integral of the new velocity (the corrector's), chunked like every other producer of it
    const int klen = (Nz + kchunks - 1) / kchunks;
    real tot = real(0.);
    for (int k0 = 0; k0 < Nz; k0 += klen) {
      const int k1 = min(Nz, k0 + klen);
      real q = real(0.);
      for (int k = k0; k < k1; k++) q = (k == k0) ? uniform_at(g.dzc, k) * col[k * T + tid] : rfma(uniform_at(g.dzc, k), col[k * T + tid], q);
      tot = (k0 == 0) ? q : tot + q;
    }
    S[o2] = (vsh && j == g.jws) ? real(0.) : tot;
  }
}

// The same solve with the column in REGISTERS (Nz <= NZT, loops fully unrolled: all of a column's loads are in flight at
// once, 8+ waves per SIMD hide the rest) and the elimination factors from tables: with a constant K they depend on the
// level and on the column's first free level only, so the host tabulates 1/beta and gamma for every (kfirst, k) once per
// (dt, K) -- flat bottom: every lane reads the same 2 Nz numbers.  One thread per column AND field (blockIdx.z).
struct ImplicitFields {
  real* f[2];
  int vshape[2];        // v-shaped parent (y faces)
  int first[2];         // first free level from: 0 = the u-face table, 1 = the v-face table, 2 = the cell table
  const real* lo[2];    // [Nz]: -dt K / (dz^c_k dz^f_k)
  const real* rb[2];    // [Nz][Nz]: 1 / beta_k of the chain that starts at level kfirst
  const real* gm[2];    // [Nz][Nz]: gamma_k of that chain
  real* sum[2];         // column integral of the result (null: not wanted)
};
template <int NZT, bool IMM>
__global__ __launch_bounds__(256) void k_implicit_vertical_reg(Grid g, ImplicitFields A, int kchunks) {
  const int f = blockIdx.z, Nz = g.Nz;
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  const bool vsh = A.vshape[f] != 0;
  if (i >= g.Nx || (vsh ? (j > g.Ny - 1) : (j >= g.Ny))) return;
  const int o2 = i2(g, i, j);
  int kf = 0;
  if (IMM) {
    const unsigned w = A.first[f] == 2 ? g.im.ordA[o2] : g.im.ordC[o2] >> (A.first[f] == 0 ? 8 : 16);
    kf = min((int)(w & 255), Nz);
  }
  real* F = A.f[f];
  const int pl = vsh ? g.pl_v : g.pl_c, o0 = vsh ? iv(g, i, j, 0) : ic(g, i, j, 0);
  real x[NZT];
#pragma unroll
  for (int k = 0; k < NZT; k++) x[k] = (k < Nz) ? F[o0 + k * pl] : real(0.);
  if (kf < Nz && !(vsh && j == g.jws)) {
    const real* rb = A.rb[f] + kf * Nz;
    const real* gm = A.gm[f] + kf * Nz;
    const real* lo = A.lo[f];
    real p = real(0.);
#pragma unroll
    for (int k = 0; k < NZT; k++)
      if (k < Nz && k >= kf) {
        p = (k == kf ? x[k] : x[k] - lo[k] * p) * rb[k];
        x[k] = p;
      }
#pragma unroll
    for (int k = NZT - 2; k >= 0; k--)
      if (k < Nz - 1 && k >= kf) x[k] = x[k] - gm[k + 1] * x[k + 1];
#pragma unroll
    for (int k = 0; k < NZT; k++)
      if (k < Nz && k >= kf) F[o0 + k * pl] = x[k];
  }
  if (A.sum[f] != nullptr) {
    const int klen = (Nz + kchunks - 1) / kchunks;
    real tot = real(0.), p = real(0.);
    int kk = 0;   // position inside the chunk
#pragma unroll
    for (int k = 0; k < NZT; k++)
      if (k < Nz) {
        p = (kk == 0) ? uniform_at(g.dzc, k) * x[k] : rfma(uniform_at(g.dzc, k), x[k], p);
        if (++kk == klen || k == Nz - 1) {
          tot = (k < klen) ? p : tot + p;
          kk = 0;
        }
      }
    A.sum[f][o2] = (vsh && j == g.jws) ? real(0.) : tot;
  }
}

#include "catke_kernels.hpp"     // closure = CATKEVerticalDiffusivity(): diffusivities and the implicit solves with them
#include "forcing_kernels.hpp"   // data-free forcing: similarity-theory fluxes

// tracers: flat AXPY over the interior planes of a parent array (G halos are identically zero)
using realx4 = __attribute__((ext_vector_type(4))) real;
template <bool NT>
__global__ void k_ab2_tracers4(real4* __restrict__ T_, real4* __restrict__ S_, const real4* __restrict__ GnT_,
                               const real4* __restrict__ GmT_, const real4* __restrict__ GnS_,
                               const real4* __restrict__ GmS_, long n4, real dt, real C1, real C2) {
  realx4* T = reinterpret_cast<realx4*>(T_);
  realx4* S = reinterpret_cast<realx4*>(S_);
  const realx4 *GnT = reinterpret_cast<const realx4*>(GnT_), *GmT = reinterpret_cast<const realx4*>(GmT_);
  const realx4 *GnS = reinterpret_cast<const realx4*>(GnS_), *GmS = reinterpret_cast<const realx4*>(GmS_);
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  // pure stream: every byte is touched once per step, so the tendency reads bypass the caches (nontemporal)
  for (; t < n4; t += stride) {
    realx4 a = T[t], b = S[t];
    realx4 gn, gm, hn, hm;
    if (NT) {
      gn = __builtin_nontemporal_load(&GnT[t]);
      gm = __builtin_nontemporal_load(&GmT[t]);
      hn = __builtin_nontemporal_load(&GnS[t]);
      hm = __builtin_nontemporal_load(&GmS[t]);
    } else {
      gn = GnT[t];
      gm = GmT[t];
      hn = GnS[t];
      hm = GmS[t];
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
      a[e] = ab2_advance(a[e], gn[e], gm[e], dt, C1, C2);
      b[e] = ab2_advance(b[e], hn[e], hm[e], dt, C1, C2);
    }
    T[t] = a;
    S[t] = b;
  }
}
__global__ void k_ab2_tracers1(real* __restrict__ T, real* __restrict__ S, const real* __restrict__ GnT,
                               const real* __restrict__ GmT, const real* __restrict__ GnS,
                               const real* __restrict__ GmS, long n, real dt, real C1, real C2) {
  long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long stride = (long)gridDim.x * blockDim.x;
  for (; t < n; t += stride) {
    T[t] = ab2_advance(T[t], GnT[t], GmT[t], dt, C1, C2);
    S[t] = ab2_advance(S[t], GnS[t], GmS[t], dt, C1, C2);
  }
}

// =============================================================================================
// Split-explicit free surface (step_free_surface!, forward-backward): one fused launch per
// substep.  eta is advanced with the old transports, then U,V with the NEW eta, exactly as the
// reference's two kernels do; because the launch is fused, the new eta at (i,j), (i-1,j), (i,j-1)
// is recomputed locally and the state is ping-ponged between two buffer sets.
// Periodic x / wall y are handled in-kernel (topology-aware operators): no halo fills inside
// the sub-cycle.
// =============================================================================================
struct Baro {
  const real *eta0, *U0, *V0;  // state at substep m
  real *eta1, *U1, *V1;        // state at substep m+1
  real *etab, *Ub, *Vb;        // running time averages
  const real *GU, *GV;
  const real *Hfc, *Hcf;       // static column depths at the U / V faces, same geometry (null: flat bottom, g.Lz)
  // geometry of these 2-D arrays: row pitch, array column of i = 0, computed range [ilo, ihi), and
  // whether i-1 / i+1 wrap around the periodic domain (single slab) or simply reach into the wide halo
  int sx, xo, ilo, ihi, wrap;
  // rows [jlo, jhi) are advanced: [0, Ny); up to Ny + Wy on the tall arrays of a folded grid (whose rows beyond the pivot row
  // are images); on a rank of a 2-D decomposition from -Wy / up to Ny + Wy on the sides where a neighbour's rows were received
  // (wide halos in y, as in x).  yo: array row of j = 0.  top_open: no wall behind row jhi - 1 -- the face row jhi exists and is
  // read (a fold's is an image and is carried through the ping-pong; a neighbour's is rim)
  int jlo, jhi, yo, top_open;
};
__device__ __forceinline__ int bi(const Grid& g, const Baro& b, int i, int j) { return (i + b.xo) + b.sx * (j + b.yo); }
__device__ __forceinline__ real eta_step(const Grid& g, const Baro& b, int i, int j, real dtau) {
  int ip = (b.wrap && i == g.Nx - 1) ? 0 : i + 1;
  real dxU = g.dy * b.U0[bi(g, b, ip, j)] - g.dy * b.U0[bi(g, b, i, j)];
  real dyV;
  if (j == g.jwn - 1) dyV = -(g.dxf[j] * b.V0[bi(g, b, i, j)]);
  else if (j == g.jws) dyV = g.dxf[j + 1] * b.V0[bi(g, b, i, j + 1)];
  else dyV = g.dxf[j + 1] * b.V0[bi(g, b, i, j + 1)] - g.dxf[j] * b.V0[bi(g, b, i, j)];
  return b.eta0[bi(g, b, i, j)] - dtau * (dxU + dyV) / g.azc[j];
}
// ORDER = 1 (option SUBSTEP_ORDER; SURVEY A.7: the order of the two halves of a substep changed between upstream releases): U, V
// first, from the old eta, then eta from the NEW U, V -- the thread recomputes the new transports on the four faces of its cell.
template <bool IMM>
__device__ __forceinline__ real U_step(const Grid& g, const Baro& b, int i, int j, real dtau) {
  const int im = (b.wrap && i == 0) ? g.Nx - 1 : i - 1, o = bi(g, b, i, j);
  const real dxe = (b.eta0[o] - b.eta0[bi(g, b, im, j)]) / g.dxc[j];
  return b.U0[o] + dtau * (-g.g * (IMM ? b.Hfc[o] : g.Lz) * dxe + b.GU[o]);
}
template <bool IMM>
__device__ __forceinline__ real V_step(const Grid& g, const Baro& b, int i, int j, real dtau) {
  const int o = bi(g, b, i, j);
  real dye = real(0.);
  if (j != g.jws) dye = (b.eta0[o] - b.eta0[bi(g, b, i, j - 1)]) / g.dy;
  return b.V0[o] + dtau * (-g.g * (IMM ? b.Hcf[o] : g.Lz) * dye + b.GV[o]);
}
template <bool IMM, int ORDER = 0>
__global__ __launch_bounds__(256) void k_barotropic_substep(Grid g, Baro b, real dtau, real wgt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x + b.ilo;
  int j = blockIdx.y * blockDim.y + threadIdx.y + b.jlo;
  if (i >= b.ihi || j >= b.jhi) return;
  if constexpr (ORDER == 1) {
    const int ip = (b.wrap && i == g.Nx - 1) ? 0 : i + 1, o = bi(g, b, i, j);
    const real Un = U_step<IMM>(g, b, i, j, dtau), Ue = U_step<IMM>(g, b, ip, j, dtau), Vn = V_step<IMM>(g, b, i, j, dtau);
    const real dxU = g.dy * Ue - g.dy * Un;
    real dyV;
    if (j == g.jwn - 1) dyV = -(g.dxf[j] * Vn);                                             // the northern wall face carries nothing
    else if (j == g.jws) dyV = g.dxf[j + 1] * V_step<IMM>(g, b, i, j + 1, dtau);
    else dyV = g.dxf[j + 1] * V_step<IMM>(g, b, i, j + 1, dtau) - g.dxf[j] * Vn;
    const real e = b.eta0[o] - dtau * (dxU + dyV) / g.azc[j];
    b.eta1[o] = e;
    b.U1[o] = Un;
    b.V1[o] = Vn;
    b.etab[o] += wgt * e;
    b.Ub[o] += wgt * Un;
    b.Vb[o] += wgt * Vn;
    return;
  }
  int im = (b.wrap && i == 0) ? g.Nx - 1 : i - 1;
  real e = eta_step(g, b, i, j, dtau);
  real ew = eta_step(g, b, im, j, dtau);
  real dxe = (e - ew) / g.dxc[j];
  real dye = real(0.);
  if (j != g.jws) dye = (e - eta_step(g, b, i, j - 1, dtau)) / g.dy;
  int o = bi(g, b, i, j);
  // (next to land the face has no depth: no pressure force, and G.U is zero there, so the transport stays zero)
  real Un = b.U0[o] + dtau * (-g.g * (IMM ? b.Hfc[o] : g.Lz) * dxe + b.GU[o]);
  real Vn = b.V0[o] + dtau * (-g.g * (IMM ? b.Hcf[o] : g.Lz) * dye + b.GV[o]);
  b.eta1[o] = e;
  b.U1[o] = Un;
  b.V1[o] = Vn;
  b.etab[o] += wgt * e;
  b.Ub[o] += wgt * Un;
  b.Vb[o] += wgt * Vn;
}
// Orthogonal curvilinear grid (tables always present): the same substep with the face lengths and areas from 2-D metric
// arrays laid out like the Baro arrays (canonical on a single domain, widened on a slab, tall on a folded grid: there the
// rows [Ny, jhi) beyond the pivot row are images, filled once per step -- k_tall_rows -- and advanced like any other row;
// the last row reads V of the face row beyond it, which is never advanced).
// (No contraction into FMAs in this kernel and in its temporally blocked sibling: the two are interchangeable bit for bit.)
#pragma clang fp contract(off)
struct CurvBaro {
  const real *dyfc, *dxcf, *razcc, *rdxfc, *rdycf;   // geometry of the Baro arrays (pitch b.sx, column offset b.xo)
};
__device__ __forceinline__ real eta_step_curv(const Grid& g, const Baro& b, const CurvBaro& c, int i, int j, real dtau) {
  const int ip = (b.wrap && i == g.Nx - 1) ? 0 : i + 1, o = bi(g, b, i, j), oe = bi(g, b, ip, j);
  const real dxU = c.dyfc[oe] * b.U0[oe] - c.dyfc[o] * b.U0[o];
  real dyV;
  if (j == g.jwn - 1 && !g.cv.north_fold) dyV = -(c.dxcf[o] * b.V0[o]);
  else if (j == g.jws) dyV = c.dxcf[o + b.sx] * b.V0[o + b.sx];
  else dyV = c.dxcf[o + b.sx] * b.V0[o + b.sx] - c.dxcf[o] * b.V0[o];
  return b.eta0[o] - dtau * (dxU + dyV) * c.razcc[o];
}
__global__ __launch_bounds__(256) void k_barotropic_substep_curv(Grid g, Baro b, CurvBaro c, real dtau, real wgt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x + b.ilo;
  const int j = blockIdx.y * blockDim.y + threadIdx.y + b.jlo;
  if (i >= b.ihi || j > b.jhi) return;
  const int o = bi(g, b, i, j);
  if (j == b.jhi) {   // no wall there: the face row behind the last advanced row is carried along unchanged
    if (b.top_open) b.V1[o] = b.V0[o];
    return;
  }
  const int im = (b.wrap && i == 0) ? g.Nx - 1 : i - 1;
  const real e = eta_step_curv(g, b, c, i, j, dtau);
  const real dxe = (e - eta_step_curv(g, b, c, im, j, dtau)) * c.rdxfc[o];
  real dye = real(0.);
  if (j != g.jws) dye = (e - eta_step_curv(g, b, c, i, j - 1, dtau)) * c.rdycf[o];
  const real Un = b.U0[o] + dtau * (-g.g * b.Hfc[o] * dxe + b.GU[o]);
  const real Vn = b.V0[o] + dtau * (-g.g * b.Hcf[o] * dye + b.GV[o]);
  b.eta1[o] = e;
  b.U1[o] = Un;
  b.V1[o] = Vn;
  b.etab[o] += wgt * e;
  b.Ub[o] += wgt * Un;
  b.Vb[o] += wgt * Vn;
}
#pragma clang fp contract(fast)
// ---------------------------------------------------------------------------------------------
// Temporally blocked sub-cycle: BT_S substeps per launch.  The one-substep kernel above moves ~60 MB per substep
// through L2/Infinity Cache (14 array sweeps of 4.3 MB at 1440x720) and is bound by that; here a block loads its
// (64 x 16) tile plus a ring of BT_S cells of eta, U, V, GU, GV into LDS once, advances BT_S substeps in LDS (the
// ring absorbs the one-cell-per-substep growth of the dependency cone), keeps the running averages in registers
// and writes everything back once.  Per-point arithmetic and summation order are those of k_barotropic_substep.
// ---------------------------------------------------------------------------------------------
constexpr int BT_TX = 64, BT_NT = 256, BT_SMAX = 24;   // (24: the whole sub-cycle in one launch, k_barotropic_whole)
struct BaroMulti {
  Baro b;
  real w[BT_SMAX];
  int ns;  // substeps in this launch (1..BT_S)
  // first launch of a sub-cycle: the running averages start from zero (no memset of the average arrays);
  // last launch: eta, U, V <- averages are written here (no finalize launch), into arrays of the canonical layout,
  // and so are the filtered-state arrays when the averages live in (wide) work arrays (no publish launch)
  int first, last;
  real *eta_out, *U_out, *V_out, *eb_out, *ub_out, *vb_out;
  // single periodic domain: the last launch also writes the halo cells tupled_fill_halo_regions! derives from the new
  // eta, U, V (periodic x images, y layer, zero on the wall faces of V): no fill launch for them in the step
  int fold;
  int out_halo;   // widened slab: the last launch writes eta, U, V of this many x halo columns too (nothing is exchanged after it)
  int out_js, out_jn;   // ... of the rows [out_js, out_jn): [0, Ny), with the halo rows of the open sides of a rank of a 2-D decomposition
  // k_barotropic_whole only: the own columns [0, Nx) of eta, U, V, G.U, G.V are read from these arrays of the canonical layout
  // instead of the work arrays (no interior-copy launch; null: the work arrays hold everything), and the y layers next to the
  // walls of the new eta, U, V are written with them (no fill launch after the sub-cycle)
  const real* own_src[5];
  int layers;
};
// (3 waves per SIMD: at 1440x720 the launch has 540 blocks; with the 173 VGPRs the 7-substep variant took when left alone
// only two blocks fit a CU, 512 on the chip, and the last 28 blocks were a second round that doubled the launch time)
template <int BT_S, int BT_TY, bool IMM>
__global__ __launch_bounds__(BT_NT, (BT_S <= 5 ? 4 : 3)) void k_barotropic_multi(Grid g, BaroMulti bm, real dtau) {
  constexpr int BT_RX = BT_TX + 2 * BT_S, BT_RY = BT_TY + 2 * BT_S, BT_NP = BT_RX * BT_RY;
  constexpr int BT_PPT = (BT_NP + BT_NT - 1) / BT_NT;
  __shared__ real E[BT_RY][BT_RX], U[BT_RY][BT_RX], V[BT_RY][BT_RX], GUs[BT_RY][BT_RX], GVs[BT_RY][BT_RX];
  __shared__ real Mdxf[BT_RY + 1], Mrazc[BT_RY], Mrdxc[BT_RY];   // row metrics: no global loads inside the sub-cycle
  // This kernel is a chain of short dependent phases (LDS, two barriers per substep) and runs BESIDE the tracer
  // tendency kernel, whose waves saturate the vector issue of every SIMD: with equal priority each of its instructions
  // queues behind theirs and the sub-cycle takes 0.32 ms per launch instead of 0.07 -- longer than the kernel it hides
  // behind.  Static high priority for its few waves costs the neighbour almost nothing.
  __builtin_amdgcn_s_setprio(3);
  const Baro& b = bm.b;
  const int tid = threadIdx.x;
  const int i0 = b.ilo + blockIdx.x * BT_TX, j0 = b.jlo + blockIdx.y * BT_TY;
  const real gH = g.g * g.Lz, rdy = g.rdy, dyc = g.dy;
  const int lo = -b.xo, hi = b.sx - b.xo - 1;   // valid array columns (slab mode: clamp; garbage stays in the rim)
  int pl[BT_PPT], pj[BT_PPT], po[BT_PPT];       // LDS index, global row, global element offset (-1: no such point)
  bool own[BT_PPT];
  real ae[BT_PPT], au[BT_PPT], av[BT_PPT];
  real ghf[BT_PPT], ghc[BT_PPT];                // g x static column depth at the point's U / V face
  real le[BT_PPT], lu[BT_PPT], lv[BT_PPT], lgu[BT_PPT], lgv[BT_PPT];   // (the loaded tile on its way to LDS)
#pragma unroll
  for (int q = 0; q < BT_PPT; q++) {
    const int p = tid + q * BT_NT;
    const int ly = p / BT_RX, lx = p - ly * BT_RX;
    const int ig = i0 - BT_S + lx, jg = j0 - BT_S + ly;
    pl[q] = p;
    pj[q] = jg;
    const bool exists = (p < BT_NP) && jg >= b.jlo && jg < b.jhi;
    int ii = ig;
    if (b.wrap) {
      ii = ii % g.Nx;
      if (ii < 0) ii += g.Nx;
    } else {
      ii = max(lo, min(hi, ii));
    }
    po[q] = exists ? bi(g, b, ii, jg) : -1;
    own[q] = exists && lx >= BT_S && lx < BT_S + BT_TX && ly >= BT_S && ly < BT_S + BT_TY && ig < b.ihi;
    // loads only: the values wait in registers until every load of the thread's points is in flight (stored to LDS
    // point by point, each point's five loads were waited for before the next point's were issued: eight round trips)
    le[q] = lu[q] = lv[q] = lgu[q] = lgv[q] = real(0.);
    if (exists) {
      le[q] = b.eta0[po[q]];
      lu[q] = b.U0[po[q]];
      lv[q] = b.V0[po[q]];
      lgu[q] = b.GU[po[q]];
      lgv[q] = b.GV[po[q]];
    }
    ae[q] = au[q] = av[q] = real(0.);
    ghf[q] = ghc[q] = gH;
    if (IMM && exists) {
      ghf[q] = g.g * b.Hfc[po[q]];
      ghc[q] = g.g * b.Hcf[po[q]];
    }
    if (own[q] && !bm.first) {
      ae[q] = b.etab[po[q]];
      au[q] = b.Ub[po[q]];
      av[q] = b.Vb[po[q]];
    }
  }
  // (the row metrics BEHIND the tile loads: ahead of them, the first wave of the block waited for three table loads one after the
  // other before it issued its share of the tile -- and the block's first barrier waited for that wave)
  if (tid <= BT_RY) {
    // rows beyond the metric tables do not exist in the domain; their entries are never used
    const int jg = max(b.jlo - (g.H + 2), min(b.jhi + g.H + 2, j0 - BT_S + tid));
    real m0 = g.dxf[jg], m1 = g.razc[jg], m2 = g.rdxc[jg];
    asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2));   // (one batch: the compiler sinks the last two into the branch below)
    Mdxf[tid] = m0;
    if (tid < BT_RY) {
      Mrazc[tid] = m1;
      Mrdxc[tid] = m2;
    }
  }
#pragma unroll
  for (int q = 0; q < BT_PPT; q++) {
    const int p = tid + q * BT_NT;
    if (p < BT_NP) {
      (&E[0][0])[p] = le[q];
      (&U[0][0])[p] = lu[q];
      (&V[0][0])[p] = lv[q];
      (&GUs[0][0])[p] = lgu[q];
      (&GVs[0][0])[p] = lgv[q];
    }
  }
  __syncthreads();
  for (int s = 0; s < bm.ns; s++) {
    const real wgt = bm.w[s];
    // ---- eta with the old transports (needs U(i+1), V(j+1))
#pragma unroll
    for (int q = 0; q < BT_PPT; q++) {
      const int p = pl[q], jg = pj[q];
      const int ly = p / BT_RX, lx = p - ly * BT_RX;
      if (po[q] >= 0 && lx < BT_RX - 1 && (ly < BT_RY - 1 || jg == g.jwn - 1)) {
        real dxU = dyc * U[ly][lx + 1] - dyc * U[ly][lx];
        real dyV;
        if (jg == g.jwn - 1) dyV = -(Mdxf[ly] * V[ly][lx]);
        else if (jg == g.jws) dyV = Mdxf[ly + 1] * V[ly + 1][lx];
        else dyV = Mdxf[ly + 1] * V[ly + 1][lx] - Mdxf[ly] * V[ly][lx];
        real e = E[ly][lx] - dtau * (dxU + dyV) * Mrazc[ly];
        E[ly][lx] = e;
        if (own[q]) ae[q] += wgt * e;
      }
    }
    __syncthreads();
    // ---- U, V with the new eta (needs eta(i-1), eta(j-1))
#pragma unroll
    for (int q = 0; q < BT_PPT; q++) {
      const int p = pl[q], jg = pj[q];
      const int ly = p / BT_RX, lx = p - ly * BT_RX;
      if (po[q] >= 0 && lx >= 1 && (ly >= 1 || jg == g.jws)) {
        real e = E[ly][lx];
        real dxe = (e - E[ly][lx - 1]) * Mrdxc[ly];
        real dye = real(0.);
        if (jg != g.jws) dye = (e - E[ly - 1][lx]) * rdy;
        real Un = U[ly][lx] + dtau * (GUs[ly][lx] - ghf[q] * dxe);
        real Vn = V[ly][lx] + dtau * (GVs[ly][lx] - ghc[q] * dye);
        U[ly][lx] = Un;
        V[ly][lx] = Vn;
        if (own[q] == 1) {
          au[q] += wgt * Un;
          av[q] += wgt * Vn;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < BT_PPT; q++)
    if (own[q]) {
      const int p = pl[q];
      b.eta1[po[q]] = (&E[0][0])[p];
      b.U1[po[q]] = (&U[0][0])[p];
      b.V1[po[q]] = (&V[0][0])[p];
      b.etab[po[q]] = ae[q];
      b.Ub[po[q]] = au[q];
      b.Vb[po[q]] = av[q];
      if (bm.last) {
        const int ly = p / BT_RX, lx = p - ly * BT_RX;
        const int ig = i0 - BT_S + lx;
        // (a widened slab also owns columns -- and, in a 2-D decomposition, rows -- outside the canonical array)
        if (ig >= -bm.out_halo && ig < g.Nx + bm.out_halo && pj[q] >= bm.out_js && pj[q] < bm.out_jn) {
          const int oc = i2(g, ig, pj[q]);
          if (bm.fold) {
            const int jg = pj[q];
            const bool xw = ig < g.H, xe = ig >= g.Nx - g.H;
            const real vv = jg == 0 ? real(0.) : av[q];                    // the southern wall face
            store_x_images(g, bm.eta_out, oc, ae[q], xw, xe);
            store_x_images(g, bm.U_out, oc, au[q], xw, xe);
            store_x_images(g, bm.V_out, oc, vv, xw, xe);
            if (jg == 0) {
              store_x_images(g, bm.eta_out, oc - g.sx, ae[q], xw, xe);
              store_x_images(g, bm.U_out, oc - g.sx, au[q], xw, xe);
            }
            if (jg == g.Ny - 1) {
              store_x_images(g, bm.eta_out, oc + g.sx, ae[q], xw, xe);
              store_x_images(g, bm.U_out, oc + g.sx, au[q], xw, xe);
              store_x_images(g, bm.V_out, oc + g.sx, real(0.), xw, xe);   // the northern wall face
            }
          } else {
            bm.eta_out[oc] = ae[q];
            bm.U_out[oc] = au[q];
            bm.V_out[oc] = av[q];
          }
          if (bm.eb_out) {
            bm.eb_out[oc] = ae[q];
            bm.ub_out[oc] = au[q];
            bm.vb_out[oc] = av[q];
          }
        }
      }
    }
}

// The WHOLE sub-cycle in one launch, for the narrow slabs of a decomposition: a 180-column rank has 172 tiles -- fewer blocks
// than the chip has CUs -- and its five blocked launches above (20 us each, dependent) are 100 us on the critical path of a
// 600 us step.  Here a block of 1024 threads holds its (64 x 17) tile with a ring of NS cells (NS = every substep: 21 with
// SplitExplicitFreeSurface(substeps = 30)) in 125 KB of LDS -- one block per CU -- and advances all NS substeps between
// barriers; substep s only touches the points the tile's final state still depends on (ring distance <= NS - 1 - s: half the
// point updates of the full ring).  Per-point arithmetic, summation order and outputs are those of k_barotropic_multi's
// first-and-last launch.  Dynamic LDS: 5 (64 + 2 NS) (BW_TY + 2 NS) reals.
// (BW_TY rows per tile: 17, or 24 -- 140 KB of LDS -- when that brings a wider slab's tile count under the number of CUs)
constexpr int BW_NT = 1024;
template <int NS, bool IMM, int BW_TY = 17>
__global__ __launch_bounds__(BW_NT) void k_barotropic_whole(Grid g, BaroMulti bm, real dtau) {
  constexpr int RX = BT_TX + 2 * NS, RY = BW_TY + 2 * NS, NP = RX * RY, PPT = (NP + BW_NT - 1) / BW_NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char bw_lds[];
  real* E = reinterpret_cast<real*>(bw_lds);
  real *U = E + NP, *V = U + NP, *GUs = V + NP, *GVs = GUs + NP;
  __shared__ real Mdxf[RY + 1], Mrazc[RY], Mrdxc[RY];
  __builtin_amdgcn_s_setprio(3);
  const Baro& b = bm.b;
  const int tid = threadIdx.x;
  const int i0 = b.ilo + blockIdx.x * BT_TX, j0 = b.jlo + blockIdx.y * BW_TY;
  const real gH = g.g * g.Lz, rdy = g.rdy, dyc = g.dy;
  const int lo = -b.xo, hi = b.sx - b.xo - 1;
  int po[PPT];            // global element offset of the point (-1: none)
  short dist[PPT];        // ring distance from the own tile (0: own); 127: never computed
  real ae[PPT], au[PPT], av[PPT], ghf[PPT], ghc[PPT];
  // (loads only in this loop, the values wait in registers until every load of the thread's points is in flight: stored to LDS
  // point by point, each point's five loads were waited for before the next point's were issued -- seven round trips at the
  // head of a 50 us kernel that a narrow rank has on its critical path; k_barotropic_multi does the same)
  real tle[PPT], tlu[PPT], tlv[PPT], tlgu[PPT], tlgv[PPT];
#pragma unroll
  for (int q = 0; q < PPT; q++) {
    const int p = tid + q * BW_NT;
    const int ly = p / RX, lx = p - ly * RX;
    const int ig = i0 - NS + lx, jg = j0 - NS + ly;
    const bool exists = (p < NP) && jg >= b.jlo && jg < b.jhi;
    int ii = ig;
    if (b.wrap) {
      ii = ii % g.Nx;
      if (ii < 0) ii += g.Nx;
    } else {
      ii = max(lo, min(hi, ii));
    }
    po[q] = exists ? bi(g, b, ii, jg) : -1;
    const int dx = lx < NS ? NS - lx : (lx >= NS + BT_TX ? lx - (NS + BT_TX - 1) : 0);
    const int dy = ly < NS ? NS - ly : (ly >= NS + BW_TY ? ly - (NS + BW_TY - 1) : 0);
    dist[q] = (short)(p < NP ? max(dx, dy) : 127);
    real le = real(0.), lu = real(0.), lv = real(0.), lgu = real(0.), lgv = real(0.);
    ghf[q] = ghc[q] = gH;
    if (exists) {
      if (bm.own_src[0] != nullptr && ig >= 0 && ig < g.Nx) {   // (own column: straight from the canonical arrays)
        const int oc = i2(g, ig, jg);
        le = bm.own_src[0][oc];
        lu = bm.own_src[1][oc];
        lv = bm.own_src[2][oc];
        lgu = bm.own_src[3][oc];
        lgv = bm.own_src[4][oc];
      } else {
        le = b.eta0[po[q]];
        lu = b.U0[po[q]];
        lv = b.V0[po[q]];
        lgu = b.GU[po[q]];
        lgv = b.GV[po[q]];
      }
      if (IMM) {
        ghf[q] = g.g * b.Hfc[po[q]];
        ghc[q] = g.g * b.Hcf[po[q]];
      }
    }
    tle[q] = le; tlu[q] = lu; tlv[q] = lv; tlgu[q] = lgu; tlgv[q] = lgv;
    ae[q] = au[q] = av[q] = real(0.);
  }
  if (tid <= RY) {   // (the row metrics behind the tile loads, in one batch)
    const int jg = max(b.jlo - (g.H + 2), min(b.jhi + g.H + 2, j0 - NS + tid));
    real m0 = g.dxf[jg], m1 = g.razc[jg], m2 = g.rdxc[jg];
    asm volatile("" : "+v"(m0), "+v"(m1), "+v"(m2));
    Mdxf[tid] = m0;
    if (tid < RY) {
      Mrazc[tid] = m1;
      Mrdxc[tid] = m2;
    }
  }
#pragma unroll
  for (int q = 0; q < PPT; q++) {
    const int p = tid + q * BW_NT;
    if (p < NP) {
      E[p] = tle[q]; U[p] = tlu[q]; V[p] = tlv[q]; GUs[p] = tlgu[q]; GVs[p] = tlgv[q];
    }
  }
  __syncthreads();
  for (int s = 0; s < bm.ns; s++) {
    const real wgt = bm.w[s];
    // Points farther out no longer matter to the tile: U, V of the state this substep produces are needed out to ring distance
    // `reach`, and they read the NEW eta one cell further west / south
    const int reach = NS - 1 - s;
    // ---- eta with the old transports (needs U(i+1), V(j+1))
#pragma unroll
    for (int q = 0; q < PPT; q++) {
      const int p = tid + q * BW_NT;
      const int ly = p / RX, lx = p - ly * RX, jg = j0 - NS + ly;
      if (po[q] >= 0 && dist[q] <= reach + 1 && lx < RX - 1 && (ly < RY - 1 || jg == g.jwn - 1)) {
        real dxU = dyc * U[p + 1] - dyc * U[p];
        real dyV;
        if (jg == g.jwn - 1) dyV = -(Mdxf[ly] * V[p]);
        else if (jg == g.jws) dyV = Mdxf[ly + 1] * V[p + RX];
        else dyV = Mdxf[ly + 1] * V[p + RX] - Mdxf[ly] * V[p];
        real e = E[p] - dtau * (dxU + dyV) * Mrazc[ly];
        E[p] = e;
        if (dist[q] == 0) ae[q] += wgt * e;
      }
    }
    __syncthreads();
    // ---- U, V with the new eta (needs eta(i-1), eta(j-1))
#pragma unroll
    for (int q = 0; q < PPT; q++) {
      const int p = tid + q * BW_NT;
      const int ly = p / RX, lx = p - ly * RX, jg = j0 - NS + ly;
      if (po[q] >= 0 && dist[q] <= reach && lx >= 1 && (ly >= 1 || jg == g.jws)) {
        real e = E[p];
        real dxe = (e - E[p - 1]) * Mrdxc[ly];
        real dye = real(0.);
        if (jg != g.jws) dye = (e - E[p - RX]) * rdy;
        real Un = U[p] + dtau * (GUs[p] - ghf[q] * dxe);
        real Vn = V[p] + dtau * (GVs[p] - ghc[q] * dye);
        U[p] = Un;
        V[p] = Vn;
        if (dist[q] == 0) {
          au[q] += wgt * Un;
          av[q] += wgt * Vn;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < PPT; q++)
    if (po[q] >= 0 && dist[q] == 0) {
      const int p = tid + q * BW_NT;
      const int ly = p / RX, lx = p - ly * RX;
      const int ig = i0 - NS + lx, jg = j0 - NS + ly;
      if (ig < b.ihi && ig >= -bm.out_halo && ig < g.Nx + bm.out_halo && jg >= bm.out_js && jg < bm.out_jn) {
        const int oc = i2(g, ig, jg);
        bm.eta_out[oc] = ae[q];
        bm.U_out[oc] = au[q];
        bm.V_out[oc] = av[q];
        if (bm.eb_out) {
          bm.eb_out[oc] = ae[q];
          bm.ub_out[oc] = au[q];
          bm.vb_out[oc] = av[q];
        }
        if (bm.layers) {   // what the y fill of the new eta, U, V would write next to the walls (fill_y_body)
          if (jg == g.jws) {
            bm.eta_out[oc - g.sx] = ae[q];
            bm.U_out[oc - g.sx] = au[q];
            bm.V_out[oc] = real(0.);
          }
          if (jg == g.jwn - 1) {
            bm.eta_out[oc + g.sx] = ae[q];
            bm.U_out[oc + g.sx] = au[q];
            bm.V_out[oc + g.sx] = real(0.);
          }
        }
      }
    }
}

// The temporally blocked sub-cycle on an orthogonal curvilinear grid: single domain (periodic in x), widened slab, and the
// tall arrays of a folded grid (rows [0, jhi) advanced, the face row jhi read only).
// Arithmetic per point is that of k_barotropic_substep_curv, operation for operation (no FMA contraction either), so the two
// are interchangeable bit for bit.  Differences from the lat-lon version above: the five metrics and the two depths a point
// needs are per-point registers; the products dyfc U and dxcf V that the eta update differences live in LDS (FU, FV) in place
// of G.U, G.V (registers here).
#pragma clang fp contract(off)
template <int BT_S, int BT_TY>
__global__ __launch_bounds__(BT_NT, 3) void k_barotropic_multi_curv(Grid g, BaroMulti bm, CurvBaro c, real dtau) {
  constexpr int BT_RX = BT_TX + 2 * BT_S, BT_RY = BT_TY + 2 * BT_S, BT_NP = BT_RX * BT_RY;
  constexpr int BT_PPT = (BT_NP + BT_NT - 1) / BT_NT;
  __shared__ real E[BT_RY][BT_RX], U[BT_RY][BT_RX], V[BT_RY][BT_RX], FU[BT_RY][BT_RX], FV[BT_RY][BT_RX];
  __builtin_amdgcn_s_setprio(3);
  const Baro& b = bm.b;
  const int tid = threadIdx.x, Nx = g.Nx, Ny = g.Ny;
  const int i0 = b.ilo + blockIdx.x * BT_TX, j0 = b.jlo + blockIdx.y * BT_TY;
  const bool open_north = b.top_open != 0;   // no wall behind the last advanced row: the face row jhi exists (read only)
  const int jtop = b.jhi + (open_north ? 1 : 0);
  const int lo = -b.xo, hi = b.sx - b.xo - 1;     // valid array columns (slab mode: clamp; garbage stays in the rim)
  int po[BT_PPT];                               // element offset of the point in the scratch / average arrays (-1: none)
  unsigned char own[BT_PPT];                    // this block writes the point back
  real ae[BT_PPT], au[BT_PPT], av[BT_PPT], gu[BT_PPT], gv[BT_PPT];
  real mrazcc[BT_PPT], mdyfc[BT_PPT], mdxcf[BT_PPT], mrdxfc[BT_PPT], mrdycf[BT_PPT], nghf[BT_PPT], nghc[BT_PPT];
  real le[BT_PPT], lu[BT_PPT], lv[BT_PPT];   // (the loaded tile on its way to LDS)
#pragma unroll
  for (int q = 0; q < BT_PPT; q++) {
    const int p = tid + q * BT_NT;
    const int ly = p / BT_RX, lx = p - ly * BT_RX;
    const int ig = i0 - BT_S + lx, jg = j0 - BT_S + ly;
    const bool exists = (p < BT_NP) && jg >= b.jlo && jg < jtop;
    int ii = ig;
    if (b.wrap) {
      ii = ii % Nx;
      if (ii < 0) ii += Nx;
    } else {
      ii = max(lo, min(hi, ii));
    }
    const bool in_tile = exists && lx >= BT_S && lx < BT_S + BT_TX && ly >= BT_S && ly < BT_S + BT_TY && ig < b.ihi;
    own[q] = (unsigned char)((in_tile && jg < b.jhi) ? 1 : (in_tile ? 2 : 0));   // 2: the face row jhi, V carried along unchanged
    po[q] = exists ? bi(g, b, ii, jg) : -1;
    // (loads only in this loop: the tile goes to LDS once every load of the thread's points is in flight)
    le[q] = lu[q] = lv[q] = real(0.);
    gu[q] = gv[q] = real(0.);
    mrazcc[q] = mdyfc[q] = mdxcf[q] = mrdxfc[q] = mrdycf[q] = nghf[q] = nghc[q] = real(0.);
    if (exists) {
      const int o = po[q];
      lv[q] = b.V0[o];
      mdxcf[q] = c.dxcf[o];
      if (jg < b.jhi) {
        le[q] = b.eta0[o];
        lu[q] = b.U0[o];
        gu[q] = b.GU[o];
        gv[q] = b.GV[o];
        mrazcc[q] = c.razcc[o];
        mdyfc[q] = c.dyfc[o];
        mrdxfc[q] = c.rdxfc[o];
        nghf[q] = -g.g * b.Hfc[o];
        mrdycf[q] = c.rdycf[o];
        nghc[q] = -g.g * b.Hcf[o];
      }
    }
    ae[q] = au[q] = av[q] = real(0.);
    if (!bm.first && own[q] == 1) {
      ae[q] = b.etab[po[q]];
      au[q] = b.Ub[po[q]];
      av[q] = b.Vb[po[q]];
    }
  }
#pragma unroll
  for (int q = 0; q < BT_PPT; q++) {
    const int p = tid + q * BT_NT;
    if (p < BT_NP) {
      (&E[0][0])[p] = le[q];
      (&U[0][0])[p] = lu[q];
      (&V[0][0])[p] = lv[q];
      (&FU[0][0])[p] = mdyfc[q] * lu[q];
      (&FV[0][0])[p] = mdxcf[q] * lv[q];
    }
  }
  __syncthreads();
  for (int s = 0; s < bm.ns; s++) {
    const real wgt = bm.w[s];
    // ---- eta with the old transports
#pragma unroll
    for (int q = 0; q < BT_PPT; q++) {
      const int p = tid + q * BT_NT;
      const int ly = p / BT_RX, lx = p - ly * BT_RX, jg = j0 - BT_S + ly;
      const bool wall = jg == g.jwn - 1 && !g.cv.north_fold;
      if (po[q] >= 0 && jg < b.jhi && lx < BT_RX - 1 && (ly < BT_RY - 1 || wall)) {
        const real dxU = FU[ly][lx + 1] - FU[ly][lx];
        real dyV;
        if (wall) dyV = -FV[ly][lx];
        else if (jg == g.jws) dyV = FV[ly + 1][lx];
        else dyV = FV[ly + 1][lx] - FV[ly][lx];
        const real e = E[ly][lx] - dtau * (dxU + dyV) * mrazcc[q];
        E[ly][lx] = e;
        if (own[q] == 1) ae[q] += wgt * e;
      }
    }
    __syncthreads();
    // ---- U, V with the new eta
#pragma unroll
    for (int q = 0; q < BT_PPT; q++) {
      const int p = tid + q * BT_NT;
      const int ly = p / BT_RX, lx = p - ly * BT_RX, jg = j0 - BT_S + ly;
      if (po[q] >= 0 && jg < b.jhi && lx >= 1 && (ly >= 1 || jg == g.jws)) {
        const real e = E[ly][lx];
        const real dxe = (e - E[ly][lx - 1]) * mrdxfc[q];
        real dye = real(0.);
        if (jg != g.jws) dye = (e - E[ly - 1][lx]) * mrdycf[q];
        const real Un = U[ly][lx] + dtau * (nghf[q] * dxe + gu[q]);
        const real Vn = V[ly][lx] + dtau * (nghc[q] * dye + gv[q]);
        U[ly][lx] = Un;
        V[ly][lx] = Vn;
        FU[ly][lx] = mdyfc[q] * Un;
        FV[ly][lx] = mdxcf[q] * Vn;
        if (own[q] == 1) {
          au[q] += wgt * Un;
          av[q] += wgt * Vn;
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int q = 0; q < BT_PPT; q++)
    if (own[q] == 2) {
      b.V1[po[q]] = (&V[0][0])[tid + q * BT_NT];
    } else if (own[q]) {
      const int p = tid + q * BT_NT, o = po[q];
      b.eta1[o] = (&E[0][0])[p];
      b.U1[o] = (&U[0][0])[p];
      b.V1[o] = (&V[0][0])[p];
      b.etab[o] = ae[q];
      b.Ub[o] = au[q];
      b.Vb[o] = av[q];
      if (bm.last) {   // eta, U, V <- the averages, in arrays of the canonical layout (and the filtered state, when the
        const int ly = p / BT_RX, lx = p - ly * BT_RX;   // averages live in work arrays of another geometry)
        const int ig = i0 - BT_S + lx, jg = j0 - BT_S + ly;
        if (jg >= bm.out_js && jg < bm.out_jn && ig >= -bm.out_halo && ig < Nx + bm.out_halo) {
          const int oc = i2(g, ig, jg);
          bm.eta_out[oc] = ae[q];
          bm.U_out[oc] = au[q];
          bm.V_out[oc] = av[q];
          if (bm.eb_out) {
            bm.eb_out[oc] = ae[q];
            bm.ub_out[oc] = au[q];
            bm.vb_out[oc] = av[q];
          }
        }
      }
    }
}
#pragma clang fp contract(fast)

// eta, U, V <- time averages on the interior (source arrays may be the wide work arrays)
// rows [js, jn): [0, Ny), with the halo rows of the open sides of a rank of a 2-D decomposition
__global__ void k_barotropic_finalize(Grid g, real* eta, real* U, real* V, const real* etab, const real* Ub,
                                      const real* Vb, int src_sx, int src_xo, int src_yo, int halo, int js, int jn) {
  int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - halo;   // (halo > 0: a widened slab, its x halo columns included)
  int j = (int)(blockIdx.y * blockDim.y + threadIdx.y) + js;
  if (i >= g.Nx + halo || j >= jn) return;
  int o = i2(g, i, j), q = (i + src_xo) + src_sx * (j + src_yo);
  V[o] = Vb[q];
  eta[o] = etab[q];
  U[o] = Ub[q];
}
// copy columns [0, Nx) of whole rows between 2-D arrays with different pitch / x-offset; up to five arrays per
// launch (blockIdx.z = array)
struct InteriorCopies {
  real* dst[5];
  const real* src[5];
  int dsx[5], dxo[5], ssx[5], sxo[5], rows[5];
  int n;
};
__global__ void k_copy_interior_columns(InteriorCopies C, int Nx) {
  const int f = blockIdx.z;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int r = blockIdx.y;
  if (i >= Nx || r >= C.rows[f]) return;
  C.dst[f][(i + C.dxo[f]) + C.dsx[f] * r] = C.src[f][(i + C.sxo[f]) + C.ssx[f] * r];
}

// =============================================================================================
// Barotropic mode and corrector (correct_velocities_and_cache_previous_tendencies!).
// =============================================================================================
__global__ __launch_bounds__(256) void k_barotropic_mode(Grid g, const real* __restrict__ u,
                                                         const real* __restrict__ v, real* __restrict__ U,
                                                         real* __restrict__ V) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  int o = ic(g, i, min(j, g.Ny - 1), 0), ov = iv(g, i, j, 0);
  real su = uniform_at(g.dzc, 0) * u[o], sv = uniform_at(g.dzc, 0) * v[ov];
  for (int k = 1; k < g.Nz; k++) {
    o += g.pl_c;
    ov += g.pl_v;
    su += uniform_at(g.dzc, k) * u[o];
    sv += uniform_at(g.dzc, k) * v[ov];
  }
  if (j < g.Ny) U[i2(g, i, j)] = su;
  V[i2(g, i, j)] = sv;
}
// The corrector of a slab's OWN columns when their column integrals are at hand (AB2 kernel / momentum look-ahead): no
// marching at all -- one thread per cell.  A 180-column slab has 2000 waves' worth of columns: the column-marching form
// runs at the latency of its 48 dependent steps (44 us for 25 MB), this one at the bandwidth of its 100 MB.
// Same operands, same additions as k_corrector -- where the correction is a rounded number before it is added, so no
// contraction of the product into the sum here: the same bits.
#pragma clang fp contract(off)
template <bool IMM>
__global__ __launch_bounds__(256) void k_corrector_cells(Grid g, real* __restrict__ u, real* __restrict__ v,
                                                         const real* __restrict__ U, const real* __restrict__ V,
                                                         real* __restrict__ Ub, real* __restrict__ Vb,
                                                         const real* __restrict__ Usum, const real* __restrict__ Vsum,
                                                         int i0, int ni, int skip_from, int skip, int jr0, int nj,
                                                         int jskip_from, int jskip) {
  // columns i0 .. i0+ni-1 with a gap of `skip` columns from index skip_from on (the two x-halo strips of a slab in one
  // launch: their column integrals arrived with the 3-D bundle, computed by the columns' owner); rows likewise (jr0, nj,
  // jskip_from, jskip: [0, Ny), or the halo rows of the open sides of a rank of a 2-D decomposition)
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, jy = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z;
  if (idx >= ni || jy >= nj) return;
  int i = i0 + idx, j = jr0 + jy;
  if (i >= skip_from) i += skip;
  if (j >= jskip_from) j += jskip;
  const int o2 = i2(g, i, j);
  const bool urow = true;
  const real su = Usum[o2], sv = Vsum[o2];
  if (k == 0) {
    if (urow) Ub[o2] = su;
    Vb[o2] = sv;
  }
  int KPU = 0, KPV = 0;
  if (IMM) {
    const unsigned C = g.im.ordC[o2];
    KPU = (C >> 8) & 255;
    KPV = (C >> 16) & 255;
  }
  if (urow && (!IMM || k >= KPU)) {
    const real du = (U[o2] - su) * (IMM ? g.im.rHfc[o2] : g.rLz);
    const int o = ic(g, i, j, k);
    u[o] = u[o] + du;
  }
  if (!IMM || k >= KPV) {
    const real dv = (V[o2] - sv) * (IMM ? g.im.rHcf[o2] : g.rLz);
    const int o = iv(g, i, j, k);
    v[o] = v[o] + dv;
  }
}
#pragma clang fp contract(fast)

// The corrector applied INSIDE its consumers (single periodic domain, flat lat-lon grid, composite steps): the
// correction u += (U - Ubar) / H is the same number for every level of a column, so instead of a sweep over u and v
// (2R + 2W per cell: 0.8 GB at 1440x720x48) this 2-D kernel leaves du = (U - Ubar) / H and dv (with the halo cells the
// fills would derive: periodic x images, the y layer of du, zero on the wall faces of dv), and the three kernels that read
// u, v in a step -- w, the momentum tendencies (which also produce the next u, v), the tracer tendencies -- add it as they
// load.  Memory then holds the uncorrected velocities until the composite call returns (k_apply_correction).
// Same operands, same additions: the same bits as the sweep.
// A slab of a decomposition (x_periodic = 0) runs it over its own columns first and, once the bundle has brought the
// neighbours' column integrals, over its x halo columns [i0, i0 + ni) with a gap of `skip` columns from `skip_from` on; no
// periodic images there.
// IMM (round 4: the grids with a bottom and the curvilinear ones): the static column depth at the face instead of Lz (zero where
// the face has no depth); the rows beyond a zipper fold and every other halo cell of du, dv come from the 2-D fill that follows.
template <bool IMM>
__global__ __launch_bounds__(256) void k_corrector_2d(Grid g, const real* __restrict__ U, const real* __restrict__ V,
                                                      const real* __restrict__ Usum, const real* __restrict__ Vsum,
                                                      real* __restrict__ Ub, real* __restrict__ Vb, real* __restrict__ du,
                                                      real* __restrict__ dv, int i0, int ni, int skip_from, int skip, int jr0,
                                                      int nj) {
  // rows [jr0, jr0 + nj): [0, Ny]; a rank of a 2-D decomposition: with the halo rows of its open sides (their column integrals and
  // their new U, V arrived like those of the halo columns).  Walls are where the GLOBAL grid has them (Grid::jws, jwn).
  const int ix = blockIdx.x * blockDim.x + threadIdx.x, jy = blockIdx.y * blockDim.y + threadIdx.y;
  if (ix >= ni || jy >= nj) return;
  int i = i0 + ix;
  if (i >= skip_from) i += skip;
  const int j = jr0 + jy;
  const int o2 = i2(g, i, j);
  const bool xw = g.x_periodic && i < g.H, xe = g.x_periodic && i >= g.Nx - g.H;
  if (j == g.jwn) {   // the northern wall face of v
    store_x_images(g, dv, o2, real(0.), xw, xe);
    return;
  }
  const real su = Usum[o2], sv = Vsum[o2];
  if (i >= 0 && i < g.Nx && j >= 0 && j < g.Ny) {
    Ub[o2] = su;
    Vb[o2] = sv;
  }
  const real a = (U[o2] - su) * (IMM ? g.im.rHfc[o2] : g.rLz);
  const real b = j == g.jws ? real(0.) : (V[o2] - sv) * (IMM ? g.im.rHcf[o2] : g.rLz);
  store_x_images(g, du, o2, a, xw, xe);
  store_x_images(g, dv, o2, b, xw, xe);
  if (j == g.jws) store_x_images(g, du, o2 - g.sx, a, xw, xe);
  if (j == g.jwn - 1) store_x_images(g, du, o2 + g.sx, a, xw, xe);
}
// The corrector through the tracer kernel (gb25_api.hip: uvc): that kernel reads the UNCORRECTED u, v of the adopted look-ahead,
// whose halo cells nobody has written -- and needs exactly two strips of them: u on the x face Nx (the east face of the last
// column: the periodic image of face 0) and, on a folded grid, v on the y faces Ny (beyond the pivot row: -v(Nx-1-i, Ny-1)).
// And v on the southern wall face: the look-ahead wrote v + dt G_v there like on any other face; it is zero (the fill's business).
// blockIdx.z = 0: the column of u; 1: the wall row of v; 2: the row of v beyond the fold.
__global__ void k_uncorrected_edges(Grid g, real* __restrict__ u, real* __restrict__ v) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (blockIdx.z == 0) {
    if (a < g.Ny && g.x_periodic) u[ic(g, g.Nx, a, k)] = u[ic(g, 0, a, k)];
  } else if (blockIdx.z == 1) {
    if (a < g.Nx && g.jws >= 0 && g.jws < g.Ny) v[iv(g, a, g.jws, k)] = real(0.);
  } else if (g.cv.north_fold && a < g.Nx) {
    v[iv(g, a, g.Ny, k)] = -v[iv(g, g.Nx - 1 - a, g.Ny - 1, k)];
  }
}
// w at the first level of every chunk of levels of the tendency kernels, for w ON THE FLY (LazyCorr::wbase): from the chunk
// integrals of u dz, v dz the momentum look-ahead left in P (of the uncorrected velocities: + du, dv times the chunk's
// thickness), differenced like the continuity equation.  One thread per column of the range the momentum kernel's w tile
// reaches: [-2, Nx + 2) x [-2, Ny + 2).  Halo columns are periodic images; in y the velocities hold one mirrored layer of u,
// zero on and beyond the wall faces of v, zeros deeper -- as the fills leave them.
// (w within a chunk then follows level by level inside the kernels: the association differs from k_compute_w's single
// march up the column, results agree to round-off, not to the last bit.)
// A slab of a decomposition (x_periodic = 0: no wrap, the halo columns of P and of du, dv are the neighbours', brought by the
// bundle) runs it over its columns [0, Nx - 2] before the interior momentum pass overwrites P there, and over the columns next
// to the x halos -- [i0, i0 + ni) with a gap of `skip` from `skip_from` on -- once the bundle has arrived.
// IMM: the correction of a face acts from its first free level KPU / KPV on (the faces below touch the solid and stay zero), so
// du, dv count with the thickness of the chunk's levels from there on; static column depths are in du, dv already.
// CURV: the face lengths and the area per point.  With the zipper fold (single domain) the rows beyond the pivot row are the
// images of the cells south of it: w(i, Ny-1+q) = w(Nx-1-i, Ny-1-q), evaluated AT the source cell -- whose northern y face,
// when it lies beyond the pivot row, is the image -v(Nx-1-i', Ny-1).
template <bool IMM, bool CURV>
__global__ __launch_bounds__(256) void k_w_bases(Grid g, const real* __restrict__ P, int kchunks, int plane2, LazyCorr lz,
                                                 real* __restrict__ wbase, int i0, int ni, int skip_from, int skip) {
  const int ix = blockIdx.x * blockDim.x + threadIdx.x, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 2;
  if (ix >= ni || j >= g.Ny + 2) return;
  int i = i0 + ix;
  if (i >= skip_from) i += skip;
  const long q = (long)kchunks * plane2;
  const int klen = (g.Nz + kchunks - 1) / kchunks;
  auto wrap = [&](int ii) { return !g.x_periodic ? ii : (ii < 0 ? ii + g.Nx : (ii >= g.Nx ? ii - g.Nx : ii)); };
  const int o2 = i2(g, i, j);
  const bool fold = CURV && g.cv.north_fold;
  // the cell whose w this is (beyond a zipper fold: the cell it is the image of)
  int ci = i, cj = j;
  if (fold && j > g.Ny - 1) {
    ci = g.Nx - 1 - wrap(i);
    cj = 2 * (g.Ny - 1) - j;
  }
  // (walls where the GLOBAL grid has them: Grid::jws, jwn; the rows of an open side of a rank of a 2-D decomposition are the
  // neighbour's and hold what it holds)
  const bool urow = cj >= g.jws - 1 && cj <= g.jwn;               // rows whose u is not identically zero (interior + one layer)
  const int ju = min(max(cj, g.jws), g.jwn - 1);
  const int ow = i2(g, wrap(ci), ju), oe = i2(g, wrap(ci + 1), ju);
  const bool vs_ok = cj >= g.jws + 1 && cj <= g.jwn - 1, vn_ok = cj + 1 >= g.jws + 1 && cj + 1 <= g.jwn - 1;
  int os = i2(g, wrap(ci), min(max(cj, g.jws), g.jwn)), on = i2(g, wrap(ci), min(max(cj + 1, g.jws), g.jwn));
  real sn = real(1.);
  if (fold && cj + 1 > g.Ny - 1) {   // the y face beyond the pivot row: (ci, Ny) <- -(Nx-1-ci, Ny-1)
    on = i2(g, g.Nx - 1 - wrap(ci), g.Ny - 1);
    sn = -real(1.);
  }
  const int oc = i2(g, wrap(ci), min(max(cj, -g.H), g.Ny + g.H - 1));
  const real dxs = CURV ? g.cv.dxcf[os] : g.dxf[cj], dxn = CURV ? g.cv.dxcf[on] : g.dxf[cj + 1];
  const real dye = CURV ? g.cv.dyfc[oe] : g.dy, dyw = CURV ? g.cv.dyfc[ow] : g.dy;
  const real raz = CURV ? g.cv.razcc[oc] : g.razc[cj];
  int Ke = 0, Kw = 0, Ks = 0, Kn = 0;
  if (IMM) {
    Ke = (int)((g.im.ordC[oe] >> 8) & 255); Kw = (int)((g.im.ordC[ow] >> 8) & 255);
    Ks = (int)((g.im.ordC[os] >> 16) & 255); Kn = (int)((g.im.ordC[on] >> 16) & 255);
  }
  const real due = lz.du ? lz.du[oe] : real(0.), duw = lz.du ? lz.du[ow] : real(0.);
  const real dvs = lz.dv ? lz.dv[os] : real(0.), dvn = lz.dv ? lz.dv[on] : real(0.);
  real w = real(0.);
  wbase[o2] = w;
  for (int c = 0; c + 1 < kchunks; c++) {
    real Ze = real(0.), Zw = real(0.), Zs = real(0.), Zn = real(0.);
    for (int k = c * klen; k < min(g.Nz, (c + 1) * klen); k++) {
      const real dz = uniform_at(g.dzc, k);
      if (!IMM || k >= Ke) Ze += dz;
      if (!IMM || k >= Kw) Zw += dz;
      if (!IMM || k >= Ks) Zs += dz;
      if (!IMM || k >= Kn) Zn += dz;
    }
    const real* Pu = P + 2 * q + (long)c * plane2;
    const real* Pv = P + 3 * q + (long)c * plane2;
    const real ue = urow ? Pu[oe] + due * Ze : real(0.), uw = urow ? Pu[ow] + duw * Zw : real(0.);
    const real vn = vn_ok ? sn * (Pv[on] + dvn * Zn) : real(0.), vs = vs_ok ? Pv[os] + dvs * Zs : real(0.);
    const real div = (dye * ue - dyw * uw) + (dxn * vn - dxs * vs);
    w = w - div * raz;
    wbase[(long)(c + 1) * plane2 + o2] = w;
  }
}
// u += du, v += dv over the whole parent extent in x and y, levels -1 .. Nz: what the consumers saw becomes what memory holds
__global__ __launch_bounds__(256) void k_apply_correction(Grid g, real* __restrict__ u, real* __restrict__ v, LazyCorr lz) {
  const int ip = blockIdx.x * blockDim.x + threadIdx.x, jp = blockIdx.y * blockDim.y + threadIdx.y;   // parent indices
  if (ip >= g.sx || jp >= g.sy_v) return;
  const int o2 = ip + g.sx * jp;
  const real a = lz.du[o2], b = lz.dv[o2];
  // (the y layer has no bottom / top layer of its own: the fills never write those corner cells)
  const bool ylayer = jp < g.H || jp >= g.H + g.Ny;
  const int klo = ylayer ? 0 : -1, khi = ylayer ? g.Nz - 1 : g.Nz;
  if (a != real(0.) && jp < g.sy_c) {
    int o = ip + g.sx * jp + g.pl_c * (g.H + klo);
    for (int k = klo; k <= khi; k++, o += g.pl_c) u[o] = u[o] + a;
  }
  if (b != real(0.)) {
    int o = ip + g.sx * jp + g.pl_v * (g.H + klo);
    for (int k = klo; k <= khi; k++, o += g.pl_v) v[o] = v[o] + b;
  }
}

// Ubar,Vbar <- column integrals of u,v (work arrays, as in the reference), then
// u += (U - Ubar)/H, v += (V - Vbar)/H.  The second sweep re-reads the column from L2.
// Columns [i0, i0+ni): a slab of a multi-GPU run also corrects its x-halo columns (same arithmetic as the
// owning neighbour, so no second halo exchange is needed); Ubar/Vbar are stored for interior columns only.
// IMM: divide by the static column depth at the face and leave the faces that touch the solid at zero.
// FOLD (single periodic domain): the corrector is the last writer of u and v in a step, so it also writes every halo
// cell tupled_fill_halo_regions! would derive from the cell it has in hand -- the periodic x image, the y layer (the
// wall faces of v: zero), the z layers, and the x images of those -- and the separate fill launch of u, v disappears
// from the critical path of the step.  Halo cells deeper than one layer in y / z never change while stepping; their x
// images were written by the last complete fill (first_time_step!, or the step after a host write).
template <bool IMM, bool FOLD>
__global__ __launch_bounds__(256) void k_corrector(Grid g, real* __restrict__ u, real* __restrict__ v,
                                                   const real* __restrict__ U, const real* __restrict__ V,
                                                   real* __restrict__ Ub, real* __restrict__ Vb,
                                                   const real* __restrict__ Usum, const real* __restrict__ Vsum,
                                                   int i0, int ni, int kchunks, int skip_from, int skip, int jr0, int nj,
                                                   int jskip_from, int jskip, real* __restrict__ du_out = nullptr,
                                                   real* __restrict__ dv_out = nullptr) {
  // du_out, dv_out (or null): the correction of the own columns as 2-D fields, for w on the fly beside the sweep (k_w_bases)
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= ni || j >= nj) return;
  i += i0;
  if (i >= skip_from) i += skip;   // (the two x-halo strips of a slab in one launch: skip the interior)
  j += jr0;                        // (rows likewise: [0, Ny), or the halo rows of the open sides of a rank of a 2-D decomposition)
  if (j >= jskip_from) j += jskip;
  const bool own = i >= 0 && i < g.Nx && j >= 0 && j < g.Ny;
  const int o0 = ic(g, i, j, 0), ov0 = iv(g, i, j, 0);
  const int o2 = i2(g, i, j);
  int o = o0, ov = ov0;
  real su = real(0.), sv = real(0.);
  if (Usum != nullptr && own) {
    // interior column: the integrals were accumulated by the AB2 kernel / the momentum kernel's look-ahead
    su = Usum[o2];
    sv = Vsum[o2];
  } else {
    // same association as theirs (per chunk of levels, chunks added in order, explicit FMAs): a halo column of a
    // slab must get the bits its owner computes
    const int klen = (g.Nz + kchunks - 1) / kchunks;
    for (int k0 = 0; k0 < g.Nz; k0 += klen) {
      const int k1 = min(g.Nz, k0 + klen);
      real pu = real(0.), pv = real(0.);
      for (int k = k0; k < k1; k++) {
        pu = (k == k0) ? uniform_at(g.dzc, k) * u[o] : rfma(uniform_at(g.dzc, k), u[o], pu);
        pv = (k == k0) ? uniform_at(g.dzc, k) * v[ov] : rfma(uniform_at(g.dzc, k), v[ov], pv);
        o += g.pl_c;
        ov += g.pl_v;
      }
      su = (k0 == 0) ? pu : su + pu;
      sv = (k0 == 0) ? pv : sv + pv;
    }
  }
  if (own) {
    Ub[o2] = su;
    Vb[o2] = sv;
  }
  const real du = (U[o2] - su) * (IMM ? g.im.rHfc[o2] : g.rLz), dv = (V[o2] - sv) * (IMM ? g.im.rHcf[o2] : g.rLz);
  if (own && du_out != nullptr) {
    du_out[o2] = du;
    dv_out[o2] = dv;
  }
  int KPU = 0, KPV = 0;
  if (IMM) {
    const unsigned C = g.im.ordC[o2];
    KPU = (C >> 8) & 255;
    KPV = (C >> 16) & 255;
  }
  o = o0;
  ov = ov0;
#pragma unroll 4
  for (int k = 0; k < g.Nz; k++) {
    if (!IMM || k >= KPU) u[o] = u[o] + du;
    if (!IMM || k >= KPV) v[ov] = v[ov] + dv;
    o += g.pl_c;
    ov += g.pl_v;
  }
  if (!FOLD) return;
  if (j == 0) {   // v on the southern wall face: zero (what the y fill writes there)
    ov = ov0;
#pragma unroll 8
    for (int k = 0; k < g.Nz; k++, ov += g.pl_v) v[ov] = real(0.);
  }
  // ---- the halo cells derived from this column (the streaming loop above stays as it was: the few threads with images
  // to write re-read their column from L2)
  const bool xw = i < g.H, xe = i >= g.Nx - g.H, ys = j == 0, yn = j == g.Ny - 1;
  {   // z layers: level -1 <- level 0, level Nz <- level Nz-1 (interior rows), with their x images
    const int ot = o0 + (g.Nz - 1) * g.pl_c, ovt = ov0 + (g.Nz - 1) * g.pl_v;
    store_x_images(g, u, o0 - g.pl_c, u[o0], xw, xe);
    store_x_images(g, v, ov0 - g.pl_v, v[ov0], xw, xe);
    store_x_images(g, u, ot + g.pl_c, u[ot], xw, xe);
    store_x_images(g, v, ovt + g.pl_v, v[ovt], xw, xe);
  }
  if (!(xw || xe || ys || yn)) return;
  o = o0;
  ov = ov0;
  // (few threads, one dependent chain each: unrolled so that eight levels' loads are in flight at once -- the row
  // blocks of the northern edge are dispatched last and their time is the tail of the kernel)
#pragma unroll 8
  for (int k = 0; k < g.Nz; k++) {
    const real un = u[o], vn = v[ov];
    if (xw) { u[o + g.Nx] = un; v[ov + g.Nx] = vn; }
    if (xe) { u[o - g.Nx] = un; v[ov - g.Nx] = vn; }
    if (ys) store_x_images(g, u, o - g.sx, un, xw, xe);                 // row -1 <- row 0
    if (yn) {
      store_x_images(g, u, o + g.sx, un, xw, xe);                       // row Ny <- row Ny-1
      store_x_images(g, v, ov + g.sx, real(0.), xw, xe);                // v: face Ny is the northern wall
    }
    o += g.pl_c;
    ov += g.pl_v;
  }
}

// set_baroclinic_instability!(model) (GB-25 src/model_utils.jl:83-87,99-110)
__global__ void k_set_baroclinic_instability(Grid g, real* __restrict__ T, real* __restrict__ S) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y;
  int k = blockIdx.z;
  if (i >= g.Nx) return;
  real phi = g.cv.on ? g.cv.phicc[i2(g, i, j)] : g.phic[j], z = uniform_at(g.zc, k);
  real step = (real(1.) - rtanh((rabs(phi) - real(40.)) / real(5.))) / real(2.);
  int o = ic(g, i, j, k);
  T[o] = (real(30.) + real(1e-3) * z) * step;
  S[o] = -real(5e-3) * z;
}

// mask_immersed_model_fields!(model, grid) (GB-25 src/precompile.jl:21,34): prognostic fields are zero at peripheral nodes
// of their location -- u, v on faces that touch an inactive cell (the wall faces of v included), T, S in inactive cells;
// the barotropic transports on faces without depth.  One thread per interior column.  Inside the composite steps the
// kernels keep these zeros themselves (no tendency, no correction on such faces; no flux into such cells), so this
// sweep runs in update_state! / initialize! and after host writes only.
__global__ __launch_bounds__(256) void k_mask_immersed(Grid g, real* __restrict__ u, real* __restrict__ v,
                                                       real* __restrict__ T, real* __restrict__ S,
                                                       real* __restrict__ U, real* __restrict__ V) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y * blockDim.y + threadIdx.y;
  // (the y faces beyond the pivot row are halo cells; so is face row Ny of a rank whose northern neighbour owns it)
  if (i >= g.Nx || j > g.Ny || (j == g.Ny && (g.cv.north_fold || g.jwn != g.Ny))) return;
  const int o2 = i2(g, i, j);
  const unsigned A = g.im.ordA[o2], C = g.im.ordC[o2];
  // v faces: j = 0 and j = Ny are walls (peripheral on the underlying grid); in between KPV levels touch the solid
  const bool wall = j == g.jws || (j == g.jwn && !g.cv.north_fold);   // (the fold line is no wall)
  const int kpv = wall ? g.Nz : (int)((C >> 16) & 255);
  int ov = iv(g, i, j, 0);
  for (int k = 0; k < min(kpv, g.Nz); k++, ov += g.pl_v) v[ov] = real(0.);
  if (wall || g.im.Hcf[o2] == real(0.)) V[o2] = real(0.);
  if (j == g.Ny) return;
  const int kc = A & 255, kpu = (C >> 8) & 255;
  int o = ic(g, i, j, 0);
  for (int k = 0; k < min(max(kc, kpu), g.Nz); k++, o += g.pl_c) {
    if (k < kpu) u[o] = real(0.);
    if (k < kc) T[o] = S[o] = real(0.);
  }
  if (g.im.Hfc[o2] == real(0.)) U[o2] = real(0.);
}

// x-slab halo exchange: pack the columns next to a slab edge / unpack into the halo.  ONE launch moves every field of
// an exchange group on both sides (blockIdx.y = piece): a group used to be up to ten 5-10 us launches in a row on the
// critical path of the staged step.  buffer layout per piece: [row][q], q in 0..ncols-1, rows = all parent rows.
struct ColumnPieces {
  real* arr[16];      // canonical (or wide) array of the piece
  real* buf[16];      // its segment of the contiguous exchange buffer of that side
  int sx[16], i0[16]; // row pitch of arr and first column (parent index) of the packed / unpacked strip
  long rows[16];
  int n, ncols;
};
template <bool PACK>
__global__ void k_move_columns(ColumnPieces P) {
  const int f = blockIdx.y;
  if (f >= P.n) return;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= P.rows[f] * P.ncols) return;
  const int q = (int)(t % P.ncols);
  const long row = t / P.ncols;
  real* a = P.arr[f] + row * P.sx[f] + P.i0[f] + q;
  if (PACK) P.buf[f][t] = *a;
  else *a = P.buf[f][t];
}

// y halo exchange of a 2-D decomposition: pack the rows next to a northern / southern edge, unpack into the halo rows.  Whole
// rows move (every parent column: the x halo columns the western / eastern neighbours sent are the corner cells of the
// diagonal neighbours).  ONE launch moves every field of a group on one side (blockIdx.z = piece).
// buffer layout per piece: [plane][row][column].
struct RowPieces {
  real* arr[16];
  real* buf[16];
  int sx[16], r0[16];     // row pitch = columns moved; first array row of the strip
  long pl[16];            // plane stride (elements)
  int k0[16], nz[16];     // first plane and number of planes (1: a 2-D array)
  int n, nrows;
};
template <bool PACK>
__global__ void k_move_rows(RowPieces P) {
  const int f = blockIdx.z, kz = blockIdx.y;
  if (f >= P.n || kz >= P.nz[f]) return;
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long n = (long)P.sx[f] * P.nrows;
  if (t >= n) return;
  real* a = P.arr[f] + (long)(P.k0[f] + kz) * P.pl[f] + (long)P.r0[f] * P.sx[f] + t;   // (the rows of a strip are contiguous)
  if (PACK) P.buf[f][(long)kz * n + t] = *a;
  else *a = P.buf[f][(long)kz * n + t];
}

}  // namespace gb25

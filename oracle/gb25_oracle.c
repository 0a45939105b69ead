/*
 * gb25_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the Oceananigans HydrostaticFreeSurfaceModel time step
 * that GB-25's baroclinic_instability_model drives.  Only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may load this library, and only as
 * the checker.  The product (gb-25_amd/, libgb25hip.so) never links or calls it.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in un-vendored Julia packages
 * (Oceananigans.jl =0.96.26, SeawaterPolynomials.jl 0.3.9, ClimaOcean.jl 0.5.10 --
 * /root/reference/Project.toml:32,37,42); Julia is absent from this image and the
 * reference commits no golden vectors.  Everything below
 * restates the published algorithms of those packages as recalled, anchored on the
 * reference's call sites:
 *   configuration   /root/reference/src/baroclinic_instability_model.jl:17-85
 *   grid / IC       /root/reference/src/model_utils.jl:56-65,83-110
 *   phase order     /root/reference/src/precompile.jl:31-42
 *   entry points    /root/reference/src/timestepping_utils.jl:21-45
 *   compared set    /root/reference/src/correctness.jl:28-90
 *
 * Conventions: logical indices are 1-based like the Julia sources (face i is the
 * west face of cell i; face j the south face; face k the bottom face).  Parent
 * arrays are column-major (i fastest) with halo H on every side, exactly
 * Oceananigans' `parent(field)` shapes (v has Ny+1 faces, w has Nz+1 faces).
 *
 * Compile twice: -DREAL=double -DSFX=_f64 and -DREAL=float -DSFX=_f32.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifndef REAL
#define REAL double
#define SFX _f64
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(CAT(gb25o_, name), SFX)

#define MAX_SUBSTEPS 512
#define PAD 2 /* extra metric padding beyond the halo */

typedef struct {
  int Nx, Ny, Nz, H;
  int substeps;
  double dt, chi;
  double lat_south, lat_north, lon_west, lon_east;
  double depth, zexp_h;
  double g, Omega, radius, rho0;
  int grid_type; /* 0: flat bottom; 1: GridFittedBottom(gaussian_islands) on this grid */
} gb25o_config;

typedef struct {
  REAL *p;
  int sx, sy, sz; /* parent dims */
} fld;

enum {
  F_U = 0, F_V, F_W, F_T, F_S, F_P,
  F_GNU, F_GNV, F_GNT, F_GNS,
  F_GMU, F_GMV, F_GMT, F_GMS,
  F_ETA, F_BU, F_BV, F_ETAB, F_UB, F_VB, F_GBU, F_GBV,
  /* closure = CATKEVerticalDiffusivity(): the TKE tracer e with its tendencies, the diffusivity fields the reference
   * compares (kappa_u, kappa_c, kappa_e at (c,c,f); L^e at (c,c,c); J^b 2-D: /root/reference/src/correctness.jl:60-67) */
  F_E, F_GNE, F_GME, F_KU, F_KC, F_KE, F_LE, F_JB,
  F_UM, F_VM, /* diffusivity_fields.previous_velocities: u, v at the previous compute_diffusivities! (CATKE's shear production) */
  F_COUNT
};

typedef struct {
  int Nx, Ny, Nz, H;
  REAL dt, chi, g, Omega, R, rho0, Lz;
  int Ns;               /* effective substeps */
  REAL dtau_frac;       /* barotropic step as a fraction of dt */
  REAL wts[MAX_SUBSTEPS];
  /* latitude metrics, index j-1+H+PAD */
  REAL *phif, *phic, *dxc, *dxf, *azc, *azf, *fcor;
  REAL dy;
  /* vertical, index k-1+H+PAD */
  REAL *zf, *zc, *dzc, *dzf;
  fld f[F_COUNT];
  double time;
  long iter;
  /* ImmersedBoundaryGrid(grid, GridFittedBottom(bottom_height)) -- /root/reference/src/model_utils.jl:134-146.
   * All 2-D, laid out like the parent of a (c,c) field (sx x sy): kbot = number of immersed cells of the column
   * (0-based index of the first active level; Nz = land), static column depths at (c,c), (f,c), (c,f). */
  int *kbot;
  REAL *Hcc, *Hfc, *Hcf;
  int immersed; /* any immersed cell at all */
  /* FluxBoundaryCondition at the top of u, v, T, S (NULL: the default no-flux): 2-D, laid out like the parent of a 2-D
   * field of the same horizontal location; J > 0 is a flux OUT of the domain through the surface (Oceananigans'
   * convention: top flux positive upward) */
  REAL *top_flux[4];
  /* quadratic bottom drag (ClimaOcean's ocean_simulation: bottom_drag_coefficient = 0.003): the bottom flux boundary condition
   * of u and v, J = -Cd |u| u at the first free level of the face's column, recomputed before every tendency evaluation */
  REAL bottom_drag;
  REAL *bottom_flux[2];
  int tracer_order; /* tracer_advection = WENO(order = 5) (0 or 5: baroclinic_instability_model) | 7 (ClimaOcean's ocean_simulation) */
  /* PrescribedAtmosphere at the ocean's cell centres (data-free forcing, /root/reference/src/data_free_ocean_climate_model.jl):
   * u_a, v_a [m/s], T_a [K], q_a [kg/kg], p_a [Pa], downwelling shortwave and longwave [W/m2]; 2-D with the parent layout of a
   * (c,c) field, halo cells included (the host evaluates the analytic fields there too).  All seven set: coupled. */
  double *atm[7];
  /* orthogonal curvilinear grid: 2-D metrics (see the macros above), cell-centre latitude for the initial condition,
   * and the topology of the northern edge: 0 = wall (Bounded), 1 = zipper fold (the tripolar grid) */
  /* closure = VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), kappa, nu)
   * (/root/reference/src/baroclinic_instability_model.jl:31); both zero: closure = nothing */
  REAL nu, kappa;
  int catke; /* closure = CATKEVerticalDiffusivity() */
  void *catke_params; /* its parameters when they are not the defaults (catke_par, below) */
  double catke_prev_time;   /* diffusivity_fields.previous_compute_time */
  int substep_order;        /* 0: eta with the old U, V, then U, V with the new eta (default); 1: U, V first, then eta with the new U, V */
  int fold_pivot_slaved;    /* 1: the fold fill also overwrites the eastern half of the pivot row with the image of its western half */
  int catke_stale_e_halos;  /* 1: the halos of e are NOT refilled after the e step inside compute_diffusivities! (upstream as recalled) */
  int curv, north_fold;
  REAL *dxfc2, *dxcc2, *dxcf2, *dxff2, *dyfc2, *dycc2, *dycf2, *dyff2, *azcc2, *azfc2, *azcf2, *azff2, *fff2, *phicc2;
  double *lamcc_d, *phicc_d; /* cell-centre coordinates in degrees (double), interior Nx x Ny, for analytic bottoms */
  double *zb_last;           /* the bottom height the level tables were last made from (Nx x Ny): re-materialised when z changes */
} model;
/* number of prognostic rows of the y-face fields: Ny whatever the northern edge is.  The zipper fold pivots on the ROW OF
 * CELL CENTRES Ny (Oceananigans' TripolarGrid: the centres run from the southernmost latitude to 90 degrees, "the north pole
 * is a Center point"; topology (Periodic, RightConnected, Bounded), so a y-face field has Ny rows): the y faces Ny+1 beyond
 * that row are halo cells, the images of the faces Ny. */
#define NYV (m->Ny)

/* ---------------------------------------------------------------- accessors */
#define HH (m->H)
#define IDX3(F, i, j, k) \
  (((long)(i)-1 + HH) + (long)(F).sx * (((long)(j)-1 + HH) + (long)(F).sy * ((long)(k)-1 + HH)))
#define IDX2(F, i, j) (((long)(i)-1 + HH) + (long)(F).sx * ((long)(j)-1 + HH))
#define A3(id, i, j, k) (m->f[id].p[IDX3(m->f[id], i, j, k)])
#define A2(id, i, j) (m->f[id].p[IDX2(m->f[id], i, j)])
#define MJ(arr, j) (m->arr[(j)-1 + HH + PAD])
#define MK(arr, k) (m->arr[(k)-1 + HH + PAD])
/* Horizontal metrics by location (Oceananigans names: Δxᶠᶜᵃ -> DXFC, Azᶜᶠᵃ -> AZCF, ...).  On the
 * LatitudeLongitudeGrid they depend on the row only: dx^fc = dx^cc at centre latitudes, dx^cf = dx^ff at face
 * latitudes, dy constant, Az^fc = Az^cc, Az^cf = Az^ff, f at face latitudes.  On an orthogonal curvilinear grid
 * (m->curv: the tripolar grid) each is a 2-D array laid out like the parent of a (c,f) field. */
#define M2(arr, i, j) (m->arr[((long)(i)-1 + HH) + (long)(m->Nx + 2 * HH) * ((long)(j)-1 + HH)])
#define DXFC(i, j) (m->curv ? M2(dxfc2, i, j) : MJ(dxc, j))
#define DXCC(i, j) (m->curv ? M2(dxcc2, i, j) : MJ(dxc, j))
#define DXCF(i, j) (m->curv ? M2(dxcf2, i, j) : MJ(dxf, j))
#define DXFF(i, j) (m->curv ? M2(dxff2, i, j) : MJ(dxf, j))
#define DYFC(i, j) (m->curv ? M2(dyfc2, i, j) : m->dy)
#define DYCC(i, j) (m->curv ? M2(dycc2, i, j) : m->dy)
#define DYCF(i, j) (m->curv ? M2(dycf2, i, j) : m->dy)
#define DYFF(i, j) (m->curv ? M2(dyff2, i, j) : m->dy)
#define AZCC(i, j) (m->curv ? M2(azcc2, i, j) : MJ(azc, j))
#define AZFC(i, j) (m->curv ? M2(azfc2, i, j) : MJ(azc, j))
#define AZCF(i, j) (m->curv ? M2(azcf2, i, j) : MJ(azf, j))
#define AZFF(i, j) (m->curv ? M2(azff2, i, j) : MJ(azf, j))
#define FFF(i, j) (m->curv ? M2(fff2, i, j) : MJ(fcor, j))
#define DZC(k) MK(dzc, k)
#define DZF(k) MK(dzf, k)

static void alloc_field(model *m, int id, int extra_y, int extra_z, int twod) {
  fld *F = &m->f[id];
  F->sx = m->Nx + 2 * m->H;
  F->sy = m->Ny + 2 * m->H + extra_y;
  F->sz = twod ? 1 : m->Nz + 2 * m->H + extra_z;
  F->p = (REAL *)calloc((size_t)F->sx * F->sy * F->sz, sizeof(REAL));
}

/* ---------------------------------------------------------------- immersed boundary
 * Oceananigans.ImmersedBoundaries, restated [UPSTREAM-UNVERIFIED]:
 *   immersed_cell(i,j,k)  = z_center(k) <= bottom_height(i,j)         (GridFittedBottom, CenterImmersedCondition)
 *   inactive_cell         = immersed_cell | outside a Bounded direction (y, z here; x is periodic)
 *   inactive_node at a face = BOTH adjacent cells inactive;  peripheral_node = EITHER adjacent cell inactive
 *   immersed_peripheral_node = peripheral on the immersed grid but not on the underlying grid
 * The materialised bottom height is the top face of the highest immersed cell, so immersed_cell(k) <=> k <= kbot
 * (1-based k, kbot = number of immersed cells). */
#define KB(i, j) (m->kbot[((long)(i)-1 + HH) + (long)(m->Nx + 2 * HH) * ((long)(j)-1 + HH)])
#define H2(arr, i, j) (m->arr[((long)(i)-1 + HH) + (long)(m->Nx + 2 * HH) * ((long)(j)-1 + HH)])
static inline int inactive_cell(const model *m, int i, int j, int k) {
  if (m->north_fold && j > m->Ny) {   /* beyond the fold: the cell it is the image of */
    i = m->Nx - i + 1;
    j = 2 * m->Ny - j;
  }
  if (j < 1 || j > m->Ny || k < 1 || k > m->Nz) return 1;
  if (i < 1 - m->H) i = 1 - m->H;              /* (beyond the x halo: never reached by an interior stencil) */
  if (i > m->Nx + m->H) i = m->Nx + m->H;
  return k <= KB(i, j);
}
static inline int peripheral_u(const model *m, int i, int j, int k) { return inactive_cell(m, i - 1, j, k) || inactive_cell(m, i, j, k); }
static inline int peripheral_v(const model *m, int i, int j, int k) { return inactive_cell(m, i, j - 1, k) || inactive_cell(m, i, j, k); }
/* (the underlying lat-lon grid has peripheral nodes of its own only on the v faces j = 1 and j = Ny+1) */
static inline int immersed_peripheral_u(const model *m, int i, int j, int k) { return peripheral_u(m, i, j, k); }
static inline int immersed_peripheral_v(const model *m, int i, int j, int k) {
  return j > 1 && j <= NYV && peripheral_v(m, i, j, k);
}
/* bottom heights at the cell centres of the interior columns -> kbot and the static column depths */
static void set_bottom(model *m, const double *zb /* Nx*Ny, i fastest */) {
  int Nx = m->Nx, Ny = m->Ny, Nz = m->Nz, H = m->H, sx = Nx + 2 * H;
  if (zb != m->zb_last) {
    free(m->zb_last);
    m->zb_last = (double *)malloc(sizeof(double) * (size_t)Nx * Ny);
    memcpy(m->zb_last, zb, sizeof(double) * (size_t)Nx * Ny);
  }
  long n2 = (long)sx * (Ny + 2 * H + 1);
  for (long q = 0; q < n2; q++) { m->kbot[q] = 0; m->Hcc[q] = m->Hfc[q] = m->Hcf[q] = 0; }
  m->immersed = 0;
  for (int j = 1; j <= Ny; j++)
    for (int i = 1; i <= Nx; i++) {
      int kb = 0;
      for (int k = 1; k <= Nz; k++)
        if ((double)MK(zc, k) <= zb[(i - 1) + (long)Nx * (j - 1)]) kb = k;   /* z_center <= bottom: immersed */
      KB(i, j) = kb;
      if (kb > 0) m->immersed = 1;
    }
  /* fill_halo_regions!(bottom_height): periodic x; rows beyond the walls lie outside the domain anyway (the static
   * depth there mirrors the wall row: zero-gradient) */
  for (int j = 1; j <= Ny; j++)
    for (int q = 0; q < H; q++) {
      KB(1 - H + q, j) = KB(Nx - H + 1 + q, j);
      KB(Nx + 1 + q, j) = KB(1 + q, j);
    }
  for (int i = 1 - H; i <= Nx + H; i++) {
    int im = Nx - i + 1;
    if (im < 1 - H) im = 1 - H;
    if (im > Nx + H) im = Nx + H;
    KB(i, 0) = KB(i, 1);
    KB(i, Ny + 1) = m->north_fold ? KB(im, Ny - 1) : KB(i, Ny);
  }
  for (int j = 0; j <= Ny + 1; j++)
    for (int i = 1 - H; i <= Nx + H; i++) {
      /* static_column_depth = z of the top face - materialised bottom height (the bottom face of the first active cell) */
      H2(Hcc, i, j) = (REAL)((double)MK(zf, Nz + 1) - (double)MK(zf, KB(i, j) + 1));
    }
  for (int j = 1; j <= Ny + 1; j++)
    for (int i = 2 - H; i <= Nx + H; i++) {
      REAL a = H2(Hcc, i, j), w = H2(Hcc, i - 1, j), sth = H2(Hcc, i, j - 1);
      H2(Hfc, i, j) = a < w ? a : w;      /* static_column_depth at (f,c) = min of the two columns */
      H2(Hcf, i, j) = a < sth ? a : sth;
    }
}
static double mtn(double lam, double phi, double lam1, double phi1) {
  const double dphi = 5;
  return exp(-((lam - lam1) * (lam - lam1) + (phi - phi1) * (phi - phi1)) / (2 * dphi * dphi));
}
/* gaussian_islands(lambda, phi) = zb + h (mtn1 + mtn2), zb = z[1], h = -zb + 100
 * (/root/reference/src/model_utils.jl:67-80,138-140) at the cell centres */
static void gaussian_islands(model *m, const gb25o_config *c, double *zb) {
  double z1 = -c->depth, h = -z1 + 100.0;
  double dlam = (c->lon_east - c->lon_west) / m->Nx;
  for (int j = 1; j <= m->Ny; j++)
    for (int i = 1; i <= m->Nx; i++) {
      double lam = c->lon_west + (i - 0.5) * dlam, phi = (double)MJ(phic, j);
      if (m->curv) {
        /* physical coordinates of the cell centre; the longitude brought next to each mountain (the tripolar grid
         * starts AT the first mountain's longitude: without this only its eastern half would exist) */
        lam = m->lamcc_d[(i - 1) + (size_t)m->Nx * (j - 1)];
        phi = m->phicc_d[(i - 1) + (size_t)m->Nx * (j - 1)];
        double l1 = lam - 360.0 * floor((lam - 70.0 + 180.0) / 360.0), l2 = lam - 360.0 * floor((lam - 250.0 + 180.0) / 360.0);
        zb[(i - 1) + (long)m->Nx * (j - 1)] = z1 + h * (mtn(l1, phi, 70, 55) + mtn(l2, phi, 70 + 180, 55));
        continue;
      }
      zb[(i - 1) + (long)m->Nx * (j - 1)] = z1 + h * (mtn(lam, phi, 70, 55) + mtn(lam, phi, 70 + 180, 55));
    }
}

/* the vertical grid from its Nz + 1 faces, bottom to top (the host's grid.z: gb25_set_vertical_faces of the library) */
static void set_vertical(model *m, const double *zint_in) {
  int H = m->H, Nz = m->Nz, nk = Nz + 2 * H + 2 * PAD + 2;
  /* the faces are numbers of the model's float type (a Float32 host holds Float32 faces); centres and spacings derive from them */
  double zint[512];
  for (int k = 0; k <= Nz; k++) zint[k] = (double)(REAL)zint_in[k];
  double *zf = (double *)calloc(nk + 1, sizeof(double));
  double *zc = (double *)calloc(nk, sizeof(double));
  /* faces incl. halos: constant extension with the first/last interior spacing */
  int off = H + PAD; /* array index of logical k=1 */
  double dlo = zint[1] - zint[0], dhi = zint[Nz] - zint[Nz - 1];
  for (int a = 0; a <= nk; a++) {
    int k = a + 1 - off; /* logical face index */
    if (k < 1) zf[a] = zint[0] + (k - 1) * dlo;
    else if (k > Nz + 1) zf[a] = zint[Nz] + (k - Nz - 1) * dhi;
    else zf[a] = zint[k - 1];
  }
  for (int a = 0; a < nk; a++) zc[a] = 0.5 * (zf[a] + zf[a + 1]);
  free(m->zf); free(m->zc); free(m->dzc); free(m->dzf);
  m->zf = (REAL *)calloc(nk + 1, sizeof(REAL));
  m->zc = (REAL *)calloc(nk, sizeof(REAL));
  m->dzc = (REAL *)calloc(nk, sizeof(REAL));
  m->dzf = (REAL *)calloc(nk, sizeof(REAL));
  for (int a = 0; a <= nk; a++) m->zf[a] = (REAL)zf[a];
  for (int a = 0; a < nk; a++) {
    m->zc[a] = (REAL)zc[a];
    m->dzc[a] = (REAL)(zf[a + 1] - zf[a]);
    m->dzf[a] = (REAL)(a > 0 ? zc[a] - zc[a - 1] : zc[1] - zc[0]);
  }
  m->Lz = (REAL)(zint[Nz] - zint[0]);
  free(zf);
  free(zc);
}
/* ---------------------------------------------------------------- grid
 * simple_latitude_longitude_grid: /root/reference/src/model_utils.jl:56-65.
 * exponential_z_faces (ClimaOcean 0.5.10, restated): k = 1..Nz+1,
 *   z_k = exp(k/h) affinely mapped so z_1 = 0, z_{Nz+1} = -depth, then reversed.
 * LatitudeLongitudeGrid metrics (Oceananigans, restated).
 */
static void build_grid(model *m, const gb25o_config *c) {
  int H = m->H, Ny = m->Ny, Nz = m->Nz;
  int nj = Ny + 2 * H + 2 * PAD + 2;
  double *phif = (double *)calloc(nj, sizeof(double));
  double *phic = (double *)calloc(nj, sizeof(double));
  m->phif = (REAL *)calloc(nj, sizeof(REAL));
  m->phic = (REAL *)calloc(nj, sizeof(REAL));
  m->dxc = (REAL *)calloc(nj, sizeof(REAL));
  m->dxf = (REAL *)calloc(nj, sizeof(REAL));
  m->azc = (REAL *)calloc(nj, sizeof(REAL));
  m->azf = (REAL *)calloc(nj, sizeof(REAL));
  m->fcor = (REAL *)calloc(nj, sizeof(REAL));
  const double d2r = M_PI / 180.0;
  double dlam = (c->lon_east - c->lon_west) / m->Nx;
  double dphi = (c->lat_north - c->lat_south) / Ny;
  double R = c->radius;
  for (int a = 0; a < nj; a++) {
    int j = a + 1 - H - PAD; /* logical index */
    phif[a] = c->lat_south + (j - 1) * dphi;
    phic[a] = c->lat_south + (j - 0.5) * dphi;
  }
  for (int a = 0; a < nj; a++) {
    m->phif[a] = (REAL)phif[a];
    m->phic[a] = (REAL)phic[a];
    m->dxc[a] = (REAL)(R * cos(phic[a] * d2r) * dlam * d2r);
    m->dxf[a] = (REAL)(R * cos(phif[a] * d2r) * dlam * d2r);
    m->fcor[a] = (REAL)(2.0 * c->Omega * sin(phif[a] * d2r));
    if (a + 1 < nj)
      m->azc[a] = (REAL)(R * R * dlam * d2r * (sin(phif[a + 1] * d2r) - sin(phif[a] * d2r)));
    if (a > 0)
      m->azf[a] = (REAL)(R * R * dlam * d2r * (sin(phic[a] * d2r) - sin(phic[a - 1] * d2r)));
  }
  m->dy = (REAL)(R * dphi * d2r);
  free(phif);
  free(phic);

  /* vertical: exponential_z_faces(Nz, depth, h) */
  double *zint = (double *)calloc(Nz + 1, sizeof(double));
  double h = c->zexp_h;
  double e1 = exp(1.0 / h), eN = exp((Nz + 1.0) / h);
  for (int k = 1; k <= Nz + 1; k++) {
    double zk = -c->depth * (exp(k / h) - e1) / (eN - e1); /* 0 at k=1, -depth at k=Nz+1 */
    zint[Nz + 1 - k] = zk;                                 /* reversed: zint[0] = -depth */
  }
  zint[Nz] = 0.0;
  set_vertical(m, zint);
  free(zint);
}

/* ---------------------------------------------------------------- orthogonal curvilinear grids
 * TripolarGrid (/root/reference/src/model_utils.jl:134-137: TripolarGrid(arch; size, halo, z)), restated as an analytic
 * bipolar cap after Murray (1996) [UPSTREAM-UNVERIFIED: Oceananigans builds its coordinates numerically; the topology,
 * pole positions (first_pole_longitude = 70, north_poles_latitude = 55), southernmost latitude (-80) and the fold are
 * the same, the interior coordinate lines of the cap need not be].
 * Computational coordinates: lambda~ uniform from the first pole's longitude eastward, phi~ uniform from the southern
 * edge to 90 degrees.  South of the poles' latitude phi_P the grid IS the lat-lon grid.  North of it the cap |z| <= r_P
 * of the polar stereographic plane (z = tan(pi/4 - phi/2) e^{i lambda}, r_P = tan(pi/4 - phi_P/2)) carries bipolar
 * coordinates with foci at the two poles: w = z / (r_P e^{i lambda_P}) = (sinh t + i sin s) / (cosh t - cos s), with
 * t = atanh(cos(lambda~ - lambda_P)) (so that the rim keeps its longitudes) and cot(s/2) = tan(pi/4 - phi~/2) / r_P (so
 * that the meridian halfway between the poles keeps its latitudes).  Lines of constant t and s are orthogonal circles;
 * phi~ = 90 is the segment between the poles: the fold line.  Metrics are great-circle distances between neighbouring
 * nodes and spherical areas of the quadrilaterals they span (as for Oceananigans' OrthogonalSphericalShellGrid). */
#define TRIPOLAR_POLE_LAT 55.0
#define TRIPOLAR_POLE_LON 70.0
typedef struct { double x, y, z, lam, phi; } gnode;
static gnode sphere_node(double lam, double phi) {
  const double d2r = M_PI / 180.0;
  if (phi < -89.999) phi = -89.999; /* (halo rows of coarse grids beyond the south pole: never used) */
  gnode n = {cos(phi * d2r) * cos(lam * d2r), cos(phi * d2r) * sin(lam * d2r), sin(phi * d2r), lam, phi};
  return n;
}
static gnode tripolar_node(double lamt, double phit) {
  const double d2r = M_PI / 180.0, lamP = TRIPOLAR_POLE_LON, phiP = TRIPOLAR_POLE_LAT;
  if (phit > 90.0) { /* beyond the fold: the image point */
    phit = 180.0 - phit;
    lamt = 2 * lamP - lamt;
  }
  if (phit <= phiP) return sphere_node(lamt, phit);
  const double rP = tan((90.0 - phiP) / 2 * d2r), rt = tan((90.0 - phit) / 2 * d2r) / rP;
  const double th = (lamt - lamP) * d2r;
  double ct = cos(th), sth = sin(th), st = fabs(sth);
  const double sg = 2 * atan2(1.0, rt); /* cot(s/2) = rt */
  const double D = 1.0 - cos(sg) * st;
  const double xw = ct / D, yw = (sth < 0 ? -1.0 : 1.0) * sin(sg) * st / D;
  const double rz = rP * sqrt(xw * xw + yw * yw);
  return sphere_node(lamP + atan2(yw, xw) / d2r, 90.0 - 2 * atan(rz) / d2r);
}
static double gc_dist(gnode a, gnode b, double R) {
  double cx = a.y * b.z - a.z * b.y, cy = a.z * b.x - a.x * b.z, cz = a.x * b.y - a.y * b.x;
  double d = R * atan2(sqrt(cx * cx + cy * cy + cz * cz), a.x * b.x + a.y * b.y + a.z * b.z);
  return d > 100.0 ? d : 100.0; /* (the coordinate lines meet at the poles: keep the metrics finite there) */
}
static double tri_area(gnode a, gnode b, gnode c) {
  double t = a.x * (b.y * c.z - b.z * c.y) + a.y * (b.z * c.x - b.x * c.z) + a.z * (b.x * c.y - b.y * c.x);
  double d = 1.0 + (a.x * b.x + a.y * b.y + a.z * b.z) + (b.x * c.x + b.y * c.y + b.z * c.z) + (c.x * a.x + c.y * a.y + c.z * a.z);
  return 2 * atan2(fabs(t), d);
}
static double quad_area(gnode a, gnode b, gnode c, gnode d, double R) {
  double A = R * R * (tri_area(a, b, c) + tri_area(a, c, d));
  return A > 1e4 ? A : 1e4;
}
/* rows beyond the pivot row of a folded grid: the metric of a location there is the metric of its image (all positive
 * scalars; copies of interior numbers, whatever generated those) */
static void mirror_metric_rows(model *m) {
  int Nx = m->Nx, Ny = m->Ny, H = m->H;
  for (int j = Ny + 1; j <= Ny + H + 1; j++)
    for (int i = 1 - H; i <= Nx + H; i++) {
      const int iw = (((i - 1) % Nx) + Nx) % Nx + 1;    /* the interior column this column is (the periodic image of) */
      const int ic = Nx - iw + 1, ifx = (Nx - iw + 2 > Nx) ? Nx - iw + 2 - Nx : Nx - iw + 2;
      const int jc = 2 * Ny - j, jf = 2 * Ny + 1 - j;   /* rows of cell centres / of y faces mirror about the centres of row Ny */
      M2(dxcc2, i, j) = M2(dxcc2, ic, jc); M2(dycc2, i, j) = M2(dycc2, ic, jc); M2(azcc2, i, j) = M2(azcc2, ic, jc);
      M2(phicc2, i, j) = M2(phicc2, ic, jc);
      M2(dxfc2, i, j) = M2(dxfc2, ifx, jc); M2(dyfc2, i, j) = M2(dyfc2, ifx, jc); M2(azfc2, i, j) = M2(azfc2, ifx, jc);
      M2(dxcf2, i, j) = M2(dxcf2, ic, jf); M2(dycf2, i, j) = M2(dycf2, ic, jf); M2(azcf2, i, j) = M2(azcf2, ic, jf);
      M2(dxff2, i, j) = M2(dxff2, ifx, jf); M2(dyff2, i, j) = M2(dyff2, ifx, jf); M2(azff2, i, j) = M2(azff2, ifx, jf);
      M2(fff2, i, j) = M2(fff2, ifx, jf);
    }
}
/* grid_type 2: the lat-lon metrics copied into the 2-D arrays (the curvilinear code path must then reproduce the plain
 * one bit for bit); 3, 4: the tripolar grid */
static void build_curv_grid(model *m, const gb25o_config *c) {
  int Nx = m->Nx, Ny = m->Ny, H = m->H, sx = Nx + 2 * H, sy = Ny + 2 * H + 1;
  size_t n2 = (size_t)sx * sy;
  REAL **arr[] = {&m->dxfc2, &m->dxcc2, &m->dxcf2, &m->dxff2, &m->dyfc2, &m->dycc2, &m->dycf2, &m->dyff2,
                  &m->azcc2, &m->azfc2, &m->azcf2, &m->azff2, &m->fff2, &m->phicc2};
  for (int q = 0; q < 14; q++) *arr[q] = (REAL *)calloc(n2, sizeof(REAL));
  m->lamcc_d = (double *)calloc((size_t)Nx * Ny, sizeof(double));
  m->phicc_d = (double *)calloc((size_t)Nx * Ny, sizeof(double));
  m->curv = 1;
  const double d2r = M_PI / 180.0, R = c->radius;
  const int tri = c->grid_type >= 3;
  const double lam0 = tri ? TRIPOLAR_POLE_LON : c->lon_west;
  const double dlam = (tri ? 360.0 : (c->lon_east - c->lon_west)) / Nx;
  /* tripolar: the Ny rows of cell centres run from the southern edge to 90 degrees (Oceananigans' TripolarGrid:
   * range(southernmost_latitude, 90, length = Ny)), the faces half a spacing south of them; the rows beyond row Ny are
   * its images and are filled from the interior below */
  const double phiN = tri ? 90.0 : c->lat_north, dphi = (phiN - c->lat_south) / (tri ? Ny - 1 : Ny);
  for (int j = 1 - H; j <= (tri ? Ny : Ny + H + 1); j++)
    for (int i = 1 - H; i <= Nx + H; i++) {
      if (!tri) {
        M2(dxfc2, i, j) = M2(dxcc2, i, j) = MJ(dxc, j);
        M2(dxcf2, i, j) = M2(dxff2, i, j) = MJ(dxf, j);
        M2(dyfc2, i, j) = M2(dycc2, i, j) = M2(dycf2, i, j) = M2(dyff2, i, j) = m->dy;
        M2(azcc2, i, j) = M2(azfc2, i, j) = MJ(azc, j);
        M2(azcf2, i, j) = M2(azff2, i, j) = MJ(azf, j);
        M2(fff2, i, j) = MJ(fcor, j);
        M2(phicc2, i, j) = MJ(phic, j);
        continue;
      }
      /* computational coordinates of the four node families around (i, j) */
      const double lf = lam0 + (i - 1) * dlam, lc = lam0 + (i - 0.5) * dlam;
      const double pc = c->lat_south + (j - 1) * dphi, pf = pc - 0.5 * dphi;
#define NODE(l, p) tripolar_node(l, p)
      gnode cc = NODE(lc, pc), fc = NODE(lf, pc), cf = NODE(lc, pf), ff = NODE(lf, pf);
      gnode fc_e = NODE(lf + dlam, pc), ff_e = NODE(lf + dlam, pf), cc_w = NODE(lc - dlam, pc), cf_w = NODE(lc - dlam, pf);
      gnode cf_n = NODE(lc, pf + dphi), ff_n = NODE(lf, pf + dphi), ff_ne = NODE(lf + dlam, pf + dphi);
      gnode cc_s = NODE(lc, pc - dphi), fc_s = NODE(lf, pc - dphi), cc_sw = NODE(lc - dlam, pc - dphi);
      gnode cf_nw = NODE(lc - dlam, pf + dphi), fc_se = NODE(lf + dlam, pc - dphi);
      M2(dxcc2, i, j) = (REAL)gc_dist(fc, fc_e, R);
      M2(dxfc2, i, j) = (REAL)gc_dist(cc_w, cc, R);
      M2(dxcf2, i, j) = (REAL)gc_dist(ff, ff_e, R);
      M2(dxff2, i, j) = (REAL)gc_dist(cf_w, cf, R);
      M2(dycc2, i, j) = (REAL)gc_dist(cf, cf_n, R);
      M2(dyfc2, i, j) = (REAL)gc_dist(ff, ff_n, R);
      M2(dycf2, i, j) = (REAL)gc_dist(cc_s, cc, R);
      M2(dyff2, i, j) = (REAL)gc_dist(fc_s, fc, R);
      M2(azcc2, i, j) = (REAL)quad_area(ff, ff_e, ff_ne, ff_n, R);
      M2(azfc2, i, j) = (REAL)quad_area(cf_w, cf, cf_n, cf_nw, R);
      M2(azcf2, i, j) = (REAL)quad_area(fc_s, fc_se, fc_e, fc, R);
      M2(azff2, i, j) = (REAL)quad_area(cc_sw, cc_s, cc, cc_w, R);
      M2(fff2, i, j) = (REAL)(2.0 * c->Omega * sin(ff.phi * d2r));
      M2(phicc2, i, j) = (REAL)cc.phi;
      if (i >= 1 && i <= Nx && j >= 1 && j <= Ny) {
        m->lamcc_d[(i - 1) + (size_t)Nx * (j - 1)] = cc.lam;
        m->phicc_d[(i - 1) + (size_t)Nx * (j - 1)] = cc.phi;
      }
#undef NODE
    }
  if (tri) mirror_metric_rows(m);
  if (!tri)
    for (int j = 1; j <= Ny; j++)
      for (int i = 1; i <= Nx; i++) {
        m->lamcc_d[(i - 1) + (size_t)Nx * (j - 1)] = c->lon_west + (i - 0.5) * dlam;
        m->phicc_d[(i - 1) + (size_t)Nx * (j - 1)] = (double)MJ(phic, j);
      }
  m->north_fold = tri;
}

/* Split-explicit averaging weights (Oceananigans FixedSubstepNumber, restated;
 * SURVEY.md appendix A.7): shape function with p=2, q=4, r=0.18927 sampled at
 * tau = 2m/Ns, m=1..Ns; searchsortedlast(weights, 0, rev=true) truncation; normalised. */
static double shape_fn(double tau) {
  const double p = 2, q = 4, r = 0.18927;
  double tau0 = (p + 2) * (p + q + 2) / (p + 1) / (p + q + 1);
  double x = tau / tau0;
  return pow(x, p) * (1 - pow(x, q)) - r * x;
}
static void build_substeps(model *m, int substeps) {
  double w[MAX_SUBSTEPS + 1];
  for (int k = 1; k <= substeps; k++) w[k] = shape_fn(2.0 * k / substeps);
  /* binary search exactly as Julia's searchsortedlast with Reverse ordering */
  int lo = 0, hi = substeps + 1;
  while (lo < hi - 1) {
    int mid = lo + ((hi - lo) >> 1);
    if (w[mid] < 0.0) hi = mid; else lo = mid;
  }
  int idx = lo;
  double s = 0;
  for (int k = 1; k <= idx; k++) s += w[k];
  m->Ns = idx;
  m->dtau_frac = (REAL)(2.0 / substeps);
  for (int k = 1; k <= idx; k++) m->wts[k - 1] = (REAL)(w[k] / s);
}

/* ---------------------------------------------------------------- lifecycle */
void *FN(create)(const gb25o_config *c) {
  if (c->substeps > MAX_SUBSTEPS || c->Nx < 8 || c->Ny < 8 || c->Nz < 4 || c->Nz > 500 || c->H < 4) return NULL;
  model *m = (model *)calloc(1, sizeof(model));
  m->Nx = c->Nx; m->Ny = c->Ny; m->Nz = c->Nz; m->H = c->H;
  m->dt = (REAL)c->dt; m->chi = (REAL)c->chi; m->g = (REAL)c->g;
  m->Omega = (REAL)c->Omega; m->R = (REAL)c->radius; m->rho0 = (REAL)c->rho0;
  build_grid(m, c);
  if (c->grid_type >= 2) build_curv_grid(m, c);
  build_substeps(m, c->substeps);
  if (m->north_fold && m->Ny - 2 < m->Ns + 1) {   /* the sub-cycle's image rows beyond the pivot row (step_free_surface_fold) need Ns + 1 rows south of it */
    free(m);
    return NULL;
  }
  {
    long n2 = (long)(m->Nx + 2 * m->H) * (m->Ny + 2 * m->H + 1);
    m->kbot = (int *)calloc(n2, sizeof(int));
    m->Hcc = (REAL *)calloc(n2, sizeof(REAL));
    m->Hfc = (REAL *)calloc(n2, sizeof(REAL));
    m->Hcf = (REAL *)calloc(n2, sizeof(REAL));
    double *zb = (double *)malloc(sizeof(double) * (size_t)m->Nx * m->Ny);
    for (long q = 0; q < (long)m->Nx * m->Ny; q++) zb[q] = -1e30;   /* flat: nothing immersed */
    if (c->grid_type == 1 || c->grid_type == 4) gaussian_islands(m, c, zb);
    set_bottom(m, zb);
    free(zb);
  }
  for (int id = 0; id < F_COUNT; id++) {
    int isv = (id == F_V || id == F_GNV || id == F_GMV || id == F_BV || id == F_VB || id == F_GBV || id == F_VM);
    int isw = (id == F_W || id == F_KU || id == F_KC || id == F_KE);
    int twod = (id >= F_ETA && id <= F_GBV) || id == F_JB;
    alloc_field(m, id, isv, isw, twod);
  }
  return m;
}
void FN(destroy)(void *h) {
  model *m = (model *)h;
  if (!m) return;
  for (int id = 0; id < F_COUNT; id++) free(m->f[id].p);
  free(m->phif); free(m->phic); free(m->dxc); free(m->dxf); free(m->azc); free(m->azf);
  free(m->fcor); free(m->zf); free(m->zc); free(m->dzc); free(m->dzf);
  free(m->kbot); free(m->Hcc); free(m->Hfc); free(m->Hcf);
  free(m->dxfc2); free(m->dxcc2); free(m->dxcf2); free(m->dxff2); free(m->dyfc2); free(m->dycc2); free(m->dycf2);
  free(m->dyff2); free(m->azcc2); free(m->azfc2); free(m->azcf2); free(m->azff2); free(m->fff2); free(m->phicc2);
  free(m->lamcc_d); free(m->phicc_d);
  for (int q = 0; q < 4; q++) free(m->top_flux[q]);
  for (int q = 0; q < 2; q++) free(m->bottom_flux[q]);
  for (int q = 0; q < 7; q++) free(m->atm[q]);
  free(m->catke_params);
  free(m);
}
REAL *FN(field_ptr)(void *h, int id) { return ((model *)h)->f[id].p; }
/* d[0..2]: the parent dims; d[3]: rows per plane in memory.  A y-face field of a folded grid has Ny rows -- the faces beyond
 * the last row of cells are halo cells -- so its parent has Ny + 2H rows; the arrays keep the row a Bounded grid needs. */
void FN(field_dims)(void *h, int id, int *d) {
  model *m = (model *)h;
  const int vshaped = m->f[id].sy == m->Ny + 2 * m->H + 1;
  d[0] = m->f[id].sx; d[1] = m->f[id].sy - ((m->north_fold && vshaped) ? 1 : 0); d[2] = m->f[id].sz;
  d[3] = m->f[id].sy;
}
/* metric id: 0 phif 1 phic 2 dxc 3 dxf 4 azc 5 azf 6 fcor 7 zf 8 zc 9 dzc 10 dzf; value at logical index */
double FN(metric)(void *h, int id, int idx) {
  model *m = (model *)h;
  REAL *a[] = {m->phif, m->phic, m->dxc, m->dxf, m->azc, m->azf, m->fcor, m->zf, m->zc, m->dzc, m->dzf};
  return (double)a[id][idx - 1 + m->H + PAD];
}
double FN(dy)(void *h) { return (double)((model *)h)->dy; }
/* 2-D metric at logical (i, j): 0 dxfc 1 dxcc 2 dxcf 3 dxff 4 dyfc 5 dycc 6 dycf 7 dyff 8 azcc 9 azfc 10 azcf 11 azff
 * 12 f(f,f) 13 phi(c,c) */
double FN(metric2)(void *h, int id, int i, int j) {
  model *m = (model *)h;
  switch (id) {
    case 0: return DXFC(i, j); case 1: return DXCC(i, j); case 2: return DXCF(i, j); case 3: return DXFF(i, j);
    case 4: return DYFC(i, j); case 5: return DYCC(i, j); case 6: return DYCF(i, j); case 7: return DYFF(i, j);
    case 8: return AZCC(i, j); case 9: return AZFC(i, j); case 10: return AZCF(i, j); case 11: return AZFF(i, j);
    case 12: return FFF(i, j);
    default: return m->curv ? (double)M2(phicc2, i, j) : (double)MJ(phic, j);
  }
}
int FN(substep_info)(void *h, double *dtau_frac, double *w) {
  model *m = (model *)h;
  *dtau_frac = m->dtau_frac;
  for (int k = 0; k < m->Ns; k++) w[k] = m->wts[k];
  return m->Ns;
}
/* bottom height at the interior cell centres (Nx*Ny doubles, i fastest); GridFittedBottom(zb) */
void FN(set_bottom_height)(void *h, const double *zb) { set_bottom((model *)h, zb); }
/* the host's grid (gb25_set_curvilinear_grid / gb25_set_vertical_faces of the library): the 14 horizontal metrics in the
 * library's gb25_metric2 order, each the parent array (Nx + 2H) x ny doubles, ny = Ny + 2H or Ny + 2H + 1; on a folded grid
 * the rows beyond the pivot row are taken from the interior by the fold's rule */
int FN(set_curvilinear_grid)(void *h, const double *const *metrics, int ny) {
  model *m = (model *)h;
  if (!m->curv) return 1;
  int Nx = m->Nx, Ny = m->Ny, H = m->H, sx = Nx + 2 * H, sy = Ny + 2 * H + 1;
  REAL *arr[] = {m->dxfc2, m->dxcc2, m->dxcf2, m->dxff2, m->dyfc2, m->dycc2, m->dycf2, m->dyff2,
                 m->azcc2, m->azfc2, m->azcf2, m->azff2, m->fff2, m->phicc2};
  if (ny != sy && ny != sy - 1) return 2;
  for (int q = 0; q < 14; q++) {
    for (long o = 0; o < (long)sx * ny; o++) arr[q][o] = (REAL)metrics[q][o];
    if (ny < sy)
      for (int i = 0; i < sx; i++) arr[q][(long)sx * ny + i] = arr[q][(long)sx * (ny - 1) + i];
  }
  if (m->north_fold) mirror_metric_rows(m);
  for (int j = 1; j <= Ny; j++)
    for (int i = 1; i <= Nx; i++) m->phicc_d[(i - 1) + (size_t)Nx * (j - 1)] = metrics[13][((long)i - 1 + H) + (long)sx * ((long)j - 1 + H)];
  return 0;
}
int FN(set_vertical_faces)(void *h, const double *zint, int n) {
  model *m = (model *)h;
  if (n != m->Nz + 1) return 1;
  set_vertical(m, zint);
  if (m->zb_last) set_bottom(m, m->zb_last);   /* the bottom is materialised on the new levels */
  return 0;
}
/* which: 0 kbot, 1 Hcc, 2 Hfc, 3 Hcf at logical (i, j) */
double FN(bottom_info)(void *h, int which, int i, int j) {
  model *m = (model *)h;
  return which == 0 ? (double)KB(i, j) : which == 1 ? (double)H2(Hcc, i, j) : which == 2 ? (double)H2(Hfc, i, j) : (double)H2(Hcf, i, j);
}
int FN(is_immersed)(void *h) { return ((model *)h)->immersed; }
void FN(set_dt)(void *h, double dt) { ((model *)h)->dt = (REAL)dt; }
double FN(get_time)(void *h) { return ((model *)h)->time; }
long FN(get_iteration)(void *h) { return ((model *)h)->iter; }

/* ---------------------------------------------------------------- TEOS-10
 * SeawaterPolynomials.TEOS10EquationOfState (Roquet et al. 2015, 55-term polynomial),
 * restated from the published coefficient table.  rho = r0(zeta) + r'(tau, s, zeta),
 * tau = Theta/40, s = sqrt((S_A + 32) * 0.875/35.16504), zeta = -Z/1e4;
 * rho' = rho - reference_density (1020 kg/m3). */
#ifndef PREAL
#define PREAL REAL /* precision of the equation of state and of the hydrostatic integral */
#endif
static const PREAL R000 = 8.0189615746e+02, R100 = 8.6672408165e+02, R200 = -1.7864682637e+03,
             R300 = 2.0375295546e+03, R400 = -1.2849161071e+03, R500 = 4.3227585684e+02,
             R600 = -6.0579916612e+01, R010 = 2.6010145068e+01, R110 = -6.5281885265e+01,
             R210 = 8.1770425108e+01, R310 = -5.6888046321e+01, R410 = 1.7681814114e+01,
             R510 = -1.9193502195e+00, R020 = -3.7074170417e+01, R120 = 6.1548258127e+01,
             R220 = -6.0362551501e+01, R320 = 2.9130021253e+01, R420 = -5.4723692739e+00,
             R030 = 2.1661789529e+01, R130 = -3.3449108469e+01, R230 = 1.9717078466e+01,
             R330 = -3.1742946532e+00, R040 = -8.3627885467e+00, R140 = 1.1311538584e+01,
             R240 = -5.3563304045e+00, R050 = 5.4048723791e-01, R150 = 4.8169980163e-01,
             R060 = -1.9083568888e-01, R001 = 1.9681925209e+01, R101 = -4.2549998214e+01,
             R201 = 5.0774768218e+01, R301 = -3.0938076334e+01, R401 = 6.6051753097e+00,
             R011 = -1.3336301113e+01, R111 = -4.4870114575e+00, R211 = 5.0042598061e+00,
             R311 = -6.5399043664e-01, R021 = 6.7080479603e+00, R121 = 3.5063081279e+00,
             R221 = -1.8795372996e+00, R031 = -2.4649669534e+00, R131 = -5.5077101279e-01,
             R041 = 5.5927935970e-01, R002 = 2.0660924175e+00, R102 = -4.9527603989e+00,
             R202 = 2.5019633244e+00, R012 = 2.0564311499e+00, R112 = -2.1311365518e-01,
             R022 = -1.2419983026e+00, R003 = -2.3342758797e-02, R103 = -1.8507636718e-02,
             R013 = 3.7969820455e-01;
static const PREAL R00 = 4.6494977072e+01, R01 = -5.2099962525e+00, R02 = 2.2601900708e-01,
             R03 = 6.4326772569e-02, R04 = 1.5616995503e-02, R05 = -1.7243708991e-03;
static inline PREAL teos10_rho(PREAL Theta, PREAL Sa, PREAL Z) {
  const PREAL t = Theta * (PREAL)0.025;
  const PREAL s = (PREAL)sqrt((double)((Sa + (PREAL)32.0) * (PREAL)(0.875 / 35.16504)));
  const PREAL z = -Z * (PREAL)1e-4;
  PREAL r3 = R013 * t + R103 * s + R003;
  PREAL r2 = (R022 * t + R112 * s + R012) * t + (R202 * s + R102) * s + R002;
  PREAL r1 = (((R041 * t + R131 * s + R031) * t + (R221 * s + R121) * s + R021) * t +
             ((R311 * s + R211) * s + R111) * s + R011) * t +
            (((R401 * s + R301) * s + R201) * s + R101) * s + R001;
  PREAL r0 = (((((R060 * t + R150 * s + R050) * t + (R240 * s + R140) * s + R040) * t +
               ((R330 * s + R230) * s + R130) * s + R030) * t +
              (((R420 * s + R320) * s + R220) * s + R120) * s + R020) * t +
             ((((R510 * s + R410) * s + R310) * s + R210) * s + R110) * s + R010) * t +
            (((((R600 * s + R500) * s + R400) * s + R300) * s + R200) * s + R100) * s + R000;
  PREAL rp = ((r3 * z + r2) * z + r1) * z + r0;
  PREAL rz = (((((R05 * z + R04) * z + R03) * z + R02) * z + R01) * z + R00) * z;
  return rz + rp;
}
double FN(teos10_rho)(double T, double S, double Z) { return (double)teos10_rho((PREAL)(REAL)T, (PREAL)(REAL)S, (PREAL)(REAL)Z); }
/* thermal_sensitivity = -d rho / d Theta and haline_sensitivity = d rho / d S_A of the same polynomial, differentiated term
 * by term (SeawaterPolynomials tabulates these derivatives as polynomials of their own: the same numbers up to the rounding of
 * its tables).  thermal_expansion = a / reference_density, haline_contraction = b / reference_density. */
static inline void teos10_sensitivities(PREAL Theta, PREAL Sa, PREAL Z, PREAL *a, PREAL *b) {
  const PREAL t = Theta * (PREAL)0.025;
  const PREAL s = (PREAL)sqrt((double)((Sa + (PREAL)32.0) * (PREAL)(0.875 / 35.16504)));
  const PREAL z = -Z * (PREAL)1e-4;
  /* the coefficient of t^j of r_k as a polynomial in s, and its derivative */
  const PREAL A00 = (((((R600 * s + R500) * s + R400) * s + R300) * s + R200) * s + R100), dA00 = ((((6 * R600 * s + 5 * R500) * s + 4 * R400) * s + 3 * R300) * s + 2 * R200) * s + R100;
  (void)A00;
  const PREAL A10 = ((((R510 * s + R410) * s + R310) * s + R210) * s + R110) * s + R010, dA10 = (((5 * R510 * s + 4 * R410) * s + 3 * R310) * s + 2 * R210) * s + R110;
  const PREAL A20 = (((R420 * s + R320) * s + R220) * s + R120) * s + R020, dA20 = ((4 * R420 * s + 3 * R320) * s + 2 * R220) * s + R120;
  const PREAL A30 = ((R330 * s + R230) * s + R130) * s + R030, dA30 = (3 * R330 * s + 2 * R230) * s + R130;
  const PREAL A40 = (R240 * s + R140) * s + R040, dA40 = 2 * R240 * s + R140;
  const PREAL A50 = R150 * s + R050, dA50 = R150;
  const PREAL A60 = R060;
  const PREAL dA01 = ((4 * R401 * s + 3 * R301) * s + 2 * R201) * s + R101;
  const PREAL A11 = ((R311 * s + R211) * s + R111) * s + R011, dA11 = (3 * R311 * s + 2 * R211) * s + R111;
  const PREAL A21 = (R221 * s + R121) * s + R021, dA21 = 2 * R221 * s + R121;
  const PREAL A31 = R131 * s + R031, dA31 = R131;
  const PREAL A41 = R041;
  const PREAL dA02 = 2 * R202 * s + R102;
  const PREAL A12 = R112 * s + R012, dA12 = R112;
  const PREAL A22 = R022;
  const PREAL dA03 = R103, A13 = R013;
  /* d r_k / d t and d r_k / d s */
  const PREAL r0t = ((((6 * A60 * t + 5 * A50) * t + 4 * A40) * t + 3 * A30) * t + 2 * A20) * t + A10;
  const PREAL r1t = ((4 * A41 * t + 3 * A31) * t + 2 * A21) * t + A11;
  const PREAL r2t = 2 * A22 * t + A12;
  const PREAL r3t = A13;
  const PREAL r0s = ((((dA50 * t + dA40) * t + dA30) * t + dA20) * t + dA10) * t + dA00;
  const PREAL r1s = ((dA31 * t + dA21) * t + dA11) * t + dA01;
  const PREAL r2s = dA12 * t + dA02;
  const PREAL r3s = dA03;
  const PREAL drdt = ((r3t * z + r2t) * z + r1t) * z + r0t, drds = ((r3s * z + r2s) * z + r1s) * z + r0s;
  *a = -(drdt * (PREAL)0.025);
  *b = drds * ((PREAL)(0.875 / 35.16504) / ((PREAL)2 * s));
}
/* (tests) out[0] = -d rho / d Theta, out[1] = d rho / d S_A */
void FN(teos10_sensitivities)(double T, double S, double Z, double *out) {
  PREAL a, b;
  teos10_sensitivities((PREAL)(REAL)T, (PREAL)(REAL)S, (PREAL)(REAL)Z, &a, &b);
  out[0] = (double)a; out[1] = (double)b;
}

/* geopotential height of a cell centre, mirrored through the boundary outside 1..Nz
 * (Oceananigans Z^ccc, restated). */
static inline REAL Zccc(const model *m, int k) {
  if (k < 1) return MK(zc, 1) + (REAL)(1 - k) * DZF(1);
  if (k > m->Nz) return MK(zc, m->Nz) - (REAL)(k - m->Nz) * DZF(m->Nz);
  return MK(zc, k);
}
/* buoyancy perturbation b = -g rho'/rho0 at (i,j,k) (SeawaterBuoyancy, restated) */
static inline PREAL buoyancy(const model *m, int i, int j, int k) {
  PREAL rho = teos10_rho((PREAL)A3(F_T, i, j, k), (PREAL)A3(F_S, i, j, k), (PREAL)Zccc(m, k));
  return -((PREAL)m->g * (rho - (PREAL)m->rho0)) / (PREAL)m->rho0;
}

/* ---------------------------------------------------------------- WENO
 * Oceananigans WENO{N} with uniform coefficients and Z-weights (appendix A.9).
 * Inputs are ordered from most-upwind (a) to most-downwind (e); left/right bias only
 * changes which grid points are gathered.  Smoothness uses the expanded quadratic
 * form with integer coefficients; eps = 1e-8. */
#define WENO_EPS ((REAL)1e-8)
static inline REAL beta5_0(REAL c, REAL d, REAL e) { /* (10,-31,11,25,-19,4) */
  return c * ((REAL)10 * c - (REAL)31 * d + (REAL)11 * e) + d * ((REAL)25 * d - (REAL)19 * e) + (REAL)4 * e * e;
}
static inline REAL beta5_1(REAL b, REAL c, REAL d) { /* (4,-13,5,13,-13,4) */
  return b * ((REAL)4 * b - (REAL)13 * c + (REAL)5 * d) + c * ((REAL)13 * c - (REAL)13 * d) + (REAL)4 * d * d;
}
static inline REAL beta5_2(REAL a, REAL b, REAL c) { /* (4,-19,11,25,-31,10) */
  return a * ((REAL)4 * a - (REAL)19 * b + (REAL)11 * c) + b * ((REAL)25 * b - (REAL)31 * c) + (REAL)10 * c * c;
}
/* Z-weights alpha_s = C_s (1 + (tau/(beta_s+eps))^2), evaluated as C_s (1 + (q rho_s)^2) with
 * q = tau/b_min and rho_s = b_min/b_s <= 1 so that q can be capped: in fp32 tau/b reaches 1e18 when one
 * indicator cancels to zero next to area-weighted divergences of O(1e8), and its square overflows.
 * The cap never binds in fp64 (exact same weights); in fp32 it changes weights below round-off. */
#define ZCAP ((REAL)(sizeof(REAL) == 4 ? 1e9 : 1e100))
static inline REAL zq(REAL tau, REAL bmin) {
  REAL q = tau / bmin;
  return q < ZCAP ? q : ZCAP;
}
/* v[0..4] = a..e values to reconstruct; b0,b1,b2 smoothness indicators */
static inline REAL weno5_combine(const REAL *v, REAL b0, REAL b1, REAL b2) {
  REAL a = v[0], b = v[1], c = v[2], d = v[3], e = v[4];
  REAL p0 = ((REAL)2 * c + (REAL)5 * d - e) / (REAL)6;
  REAL p1 = (-b + (REAL)5 * c + (REAL)2 * d) / (REAL)6;
  REAL p2 = ((REAL)2 * a - (REAL)7 * b + (REAL)11 * c) / (REAL)6;
  REAL tau = (REAL)fabs((double)(b0 - b2));
  /* the expanded quadratic forms can round to small negative numbers: clamp (a no-op in exact arithmetic) */
  b0 = (b0 > 0 ? b0 : 0) + WENO_EPS; b1 = (b1 > 0 ? b1 : 0) + WENO_EPS; b2 = (b2 > 0 ? b2 : 0) + WENO_EPS;
  REAL bmin = b0 < b1 ? (b0 < b2 ? b0 : b2) : (b1 < b2 ? b1 : b2);
  REAL q = zq(tau, bmin);
  REAL r0 = q * (bmin / b0), r1 = q * (bmin / b1), r2 = q * (bmin / b2);
  REAL a0 = (REAL)0.3 * ((REAL)1 + r0 * r0), a1 = (REAL)0.6 * ((REAL)1 + r1 * r1), a2 = (REAL)0.1 * ((REAL)1 + r2 * r2);
  return (a0 * p0 + a1 * p1 + a2 * p2) / (a0 + a1 + a2);
}
static inline REAL weno3_combine(const REAL *v, REAL b0, REAL b1) {
  REAL b = v[1], c = v[2], d = v[3];
  REAL p0 = (c + d) / (REAL)2;
  REAL p1 = (-b + (REAL)3 * c) / (REAL)2;
  REAL tau = (REAL)fabs((double)(b0 - b1));
  b0 = (b0 > 0 ? b0 : 0) + WENO_EPS; b1 = (b1 > 0 ? b1 : 0) + WENO_EPS;
  REAL bmin = b0 < b1 ? b0 : b1;
  REAL q = zq(tau, bmin);
  REAL r0 = q * (bmin / b0), r1 = q * (bmin / b1);
  REAL a0 = (REAL)(2.0 / 3.0) * ((REAL)1 + r0 * r0), a1 = (REAL)(1.0 / 3.0) * ((REAL)1 + r1 * r1);
  return (a0 * p0 + a1 * p1) / (a0 + a1);
}
static inline REAL beta3(REAL x, REAL y) { return x * (x - (REAL)2 * y) + y * y; } /* (1,-2,1) */

double FN(weno5)(double a, double b, double c, double d, double e) {
  REAL v[5] = {(REAL)a, (REAL)b, (REAL)c, (REAL)d, (REAL)e};
  return (double)weno5_combine(v, beta5_0(v[2], v[3], v[4]), beta5_1(v[1], v[2], v[3]), beta5_2(v[0], v[1], v[2]));
}
double FN(weno3)(double b, double c, double d) {
  REAL v[5] = {0, (REAL)b, (REAL)c, (REAL)d, 0};
  return (double)weno3_combine(v, beta3(v[2], v[3]), beta3(v[1], v[2]));
}

/* WENO(order = 7) (ClimaOcean's ocean_simulation: tracer_advection = WENO(order = 7)) [UPSTREAM-UNVERIFIED: the
 * reconstruction polynomials, linear weights and smoothness indicators are those of Balsara & Shu (2000) for uniform
 * spacing, with Z-weights tau_7 = |beta_0 + 3 beta_1 - 3 beta_2 - beta_3| as Oceananigans' global smoothness indicator of
 * buffer 4].  v[0..6]: upwind-most .. downwind-most, the face between v[3] and v[4]; stencil r uses v[3-r .. 6-r]. */
static inline REAL weno7_combine(const REAL *v) {
  const REAL a = v[0], b = v[1], c = v[2], d = v[3], e = v[4], f = v[5], g = v[6];
  REAL p0 = ((REAL)3 * d + (REAL)13 * e - (REAL)5 * f + g) / (REAL)12;
  REAL p1 = (-c + (REAL)7 * d + (REAL)7 * e - f) / (REAL)12;
  REAL p2 = (b - (REAL)5 * c + (REAL)13 * d + (REAL)3 * e) / (REAL)12;
  REAL p3 = ((REAL)-3 * a + (REAL)13 * b - (REAL)23 * c + (REAL)25 * d) / (REAL)12;
  REAL b0 = d * ((REAL)2107 * d - (REAL)9402 * e + (REAL)7042 * f - (REAL)1854 * g) + e * ((REAL)11003 * e - (REAL)17246 * f + (REAL)4642 * g) +
            f * ((REAL)7043 * f - (REAL)3882 * g) + (REAL)547 * g * g;
  REAL b1 = c * ((REAL)547 * c - (REAL)2522 * d + (REAL)1922 * e - (REAL)494 * f) + d * ((REAL)3443 * d - (REAL)5966 * e + (REAL)1602 * f) +
            e * ((REAL)2843 * e - (REAL)1642 * f) + (REAL)267 * f * f;
  REAL b2 = b * ((REAL)267 * b - (REAL)1642 * c + (REAL)1602 * d - (REAL)494 * e) + c * ((REAL)2843 * c - (REAL)5966 * d + (REAL)1922 * e) +
            d * ((REAL)3443 * d - (REAL)2522 * e) + (REAL)547 * e * e;
  REAL b3 = a * ((REAL)547 * a - (REAL)3882 * b + (REAL)4642 * c - (REAL)1854 * d) + b * ((REAL)7043 * b - (REAL)17246 * c + (REAL)7042 * d) +
            c * ((REAL)11003 * c - (REAL)9402 * d) + (REAL)2107 * d * d;
  REAL tau = (REAL)fabs((double)(b0 + (REAL)3 * b1 - (REAL)3 * b2 - b3));
  b0 = (b0 > 0 ? b0 : 0) + WENO_EPS; b1 = (b1 > 0 ? b1 : 0) + WENO_EPS; b2 = (b2 > 0 ? b2 : 0) + WENO_EPS; b3 = (b3 > 0 ? b3 : 0) + WENO_EPS;
  REAL bmin = b0 < b1 ? b0 : b1;
  bmin = bmin < b2 ? bmin : b2;
  bmin = bmin < b3 ? bmin : b3;
  REAL q = zq(tau, bmin);
  REAL r0 = q * (bmin / b0), r1 = q * (bmin / b1), r2 = q * (bmin / b2), r3 = q * (bmin / b3);
  REAL a0 = (REAL)(4.0 / 35.0) * ((REAL)1 + r0 * r0), a1 = (REAL)(18.0 / 35.0) * ((REAL)1 + r1 * r1),
       a2 = (REAL)(12.0 / 35.0) * ((REAL)1 + r2 * r2), a3 = (REAL)(1.0 / 35.0) * ((REAL)1 + r3 * r3);
  return (a0 * p0 + a1 * p1 + a2 * p2 + a3 * p3) / (a0 + a1 + a2 + a3);
}
double FN(weno7)(const double *u) {
  REAL v[7];
  for (int q = 0; q < 7; q++) v[q] = (REAL)u[q];
  return (double)weno7_combine(v);
}

/* ---------------------------------------------------------------- stencil functions */
typedef REAL (*fn3)(const model *, int, int, int);
static REAL f_u(const model *m, int i, int j, int k) { return A3(F_U, i, j, k); }
static REAL f_v(const model *m, int i, int j, int k) { return A3(F_V, i, j, k); }
static REAL f_T(const model *m, int i, int j, int k) { return A3(F_T, i, j, k); }
static REAL f_S(const model *m, int i, int j, int k) { return A3(F_S, i, j, k); }
static REAL f_E(const model *m, int i, int j, int k) { return A3(F_E, i, j, k); }
static REAL f_Azw(const model *m, int i, int j, int k) { return AZCC(i, j) * A3(F_W, i, j, k); }
/* vertical vorticity zeta at (f,f,c) */
static REAL f_zeta(const model *m, int i, int j, int k) {
  REAL circ = (DYCF(i, j) * A3(F_V, i, j, k) - DYCF(i - 1, j) * A3(F_V, i - 1, j, k)) -
              (DXFC(i, j) * A3(F_U, i, j, k) - DXFC(i, j - 1) * A3(F_U, i, j - 1, k));
  return circ / AZFF(i, j);
}
/* VelocityStencil smoothness inputs at (f,f,c) */
static REAL f_uy(const model *m, int i, int j, int k) { return (A3(F_U, i, j - 1, k) + A3(F_U, i, j, k)) / (REAL)2; }
static REAL f_vx(const model *m, int i, int j, int k) { return (A3(F_V, i - 1, j, k) + A3(F_V, i, j, k)) / (REAL)2; }
/* delta_x(Ax u), delta_y(Ay v) at (c,c,c) */
static REAL f_dxU(const model *m, int i, int j, int k) {
  return DYFC(i + 1, j) * DZC(k) * A3(F_U, i + 1, j, k) - DYFC(i, j) * DZC(k) * A3(F_U, i, j, k);
}
static REAL f_dyV(const model *m, int i, int j, int k) {
  return DXCF(i, j + 1) * DZC(k) * A3(F_V, i, j + 1, k) - DXCF(i, j) * DZC(k) * A3(F_V, i, j, k);
}
static REAL f_div(const model *m, int i, int j, int k) { return f_dxU(m, i, j, k) + f_dyV(m, i, j, k); }
static inline REAL half_sq(REAL x) { return x * x / (REAL)2; }
static REAL f_dxu2(const model *m, int i, int j, int k) { return half_sq(A3(F_U, i + 1, j, k)) - half_sq(A3(F_U, i, j, k)); }
static REAL f_dyv2(const model *m, int i, int j, int k) { return half_sq(A3(F_V, i, j + 1, k)) - half_sq(A3(F_V, i, j, k)); }
static REAL f_dxv2(const model *m, int i, int j, int k) { return half_sq(A3(F_V, i, j, k)) - half_sq(A3(F_V, i - 1, j, k)); }
static REAL f_dyu2(const model *m, int i, int j, int k) { return half_sq(A3(F_U, i, j, k)) - half_sq(A3(F_U, i, j - 1, k)); }
static REAL f_usm(const model *m, int i, int j, int k) { return (A3(F_U, i, j, k) + A3(F_U, i + 1, j, k)) / (REAL)2; }
static REAL f_vsm(const model *m, int i, int j, int k) { return (A3(F_V, i, j, k) + A3(F_V, i, j + 1, k)) / (REAL)2; }

enum { DX = 0, DY = 1, DZ = 2 };
enum { TO_FACE = 0, TO_CENTER = 1 };

static inline REAL at_dir(const model *m, fn3 f, int dir, int i, int j, int k, int p) {
  return dir == DX ? f(m, p, j, k) : dir == DY ? f(m, i, p, k) : f(m, i, j, p);
}

/* Upwind-biased reconstruction of psi along `dir` to a face / centre with index idx
 * (the component of (i,j,k) along dir).  left != 0: LeftBias (advecting velocity > 0).
 * s1/s2: optional smoothness functions (FunctionStencil: s1; VelocityStencil: s1,s2).
 * Bounded directions (y, z) drop to WENO3 and first-order upwind next to the walls
 * (topologically conditional interpolation, restated). */
/* Is every node of the 2*buffer-point stencil around the target active?  Face target idx: the cells idx-buffer ..
 * idx+buffer-1 (nodes at the centre location along dir); centre target idx: the faces idx-buffer+1 .. idx+buffer,
 * a face node being inactive when BOTH cells it separates are inactive.  Along the other two directions the node sits
 * at the centre location with the target's own indices (Oceananigans' near_*_immersed_boundary_* functions, restated).
 * With nothing immersed this reduces to the topological rules for Bounded directions (outside_*_halo). */
static int stencil_active(const model *m, int dir, int target, int i, int j, int k, int buffer) {
  int idx = dir == DX ? i : dir == DY ? j : k;
  /* a horizontal stencil of a quantity that lives on the z faces (the advecting Az w of the vertical momentum flux)
   * sees the activity of the cell row above the face; the top face Nz+1 that of level Nz.  [restatement choice: keeps
   * the flat-bottom immersed grid identical to the plain grid, where the rule looks at the horizontal index only] */
  if (dir != DZ && k > m->Nz) k = m->Nz;
  int p0 = target == TO_FACE ? idx - buffer : idx - buffer + 1, p1 = p0 + 2 * buffer - 1;
  for (int p = p0; p <= p1; p++) {
    int a = dir == DX ? inactive_cell(m, p, j, k) : dir == DY ? inactive_cell(m, i, p, k) : inactive_cell(m, i, j, p);
    if (target == TO_FACE) {
      if (a) return 0;
    } else {
      int b = dir == DX ? inactive_cell(m, p - 1, j, k) : dir == DY ? inactive_cell(m, i, p - 1, k) : inactive_cell(m, i, j, p - 1);
      if (a && b) return 0;
    }
  }
  return 1;
}
static int stencil_active(const model *m, int dir, int target, int i, int j, int k, int buffer);
/* WENO(order = 7) to a face, self-smoothness (the tracer fluxes of an ocean_simulation); falls back to the order-5 path
 * below where the eight-point stencil meets a wall or the immersed boundary */
static int weno7_applies(const model *m, int dir, int i, int j, int k) {
  return m->tracer_order == 7 && ((dir == DX && !m->immersed) || stencil_active(m, dir, TO_FACE, i, j, k, 4));
}
static REAL biased_interp(const model *m, int dir, int target, int i, int j, int k, int left, fn3 psi, fn3 s1, fn3 s2);
static REAL tracer_interp(const model *m, int dir, int i, int j, int k, int left, fn3 c) {
  if (!weno7_applies(m, dir, i, j, k)) return biased_interp(m, dir, TO_FACE, i, j, k, left, c, NULL, NULL);
  int idx = dir == DX ? i : dir == DY ? j : k, c0 = left ? idx - 1 : idx, sg = left ? 1 : -1;
  REAL v[7];
  for (int q = -3; q <= 3; q++) v[q + 3] = at_dir(m, c, dir, i, j, k, c0 + sg * q);
  return weno7_combine(v);
}
static REAL biased_interp(const model *m, int dir, int target, int i, int j, int k, int left, fn3 psi, fn3 s1, fn3 s2) {
  int idx = dir == DX ? i : dir == DY ? j : k;
  /* WENO5 -> WENO3 -> first-order upwind as the stencil meets a wall or the immersed boundary */
  int order = (dir == DX && !m->immersed) ? 5
            : stencil_active(m, dir, target, i, j, k, 3) ? 5 : stencil_active(m, dir, target, i, j, k, 2) ? 3 : 1;
  int c0 = target == TO_FACE ? (left ? idx - 1 : idx) : (left ? idx : idx + 1);
  int sg = left ? 1 : -1;
  if (order == 1) return at_dir(m, psi, dir, i, j, k, c0);
  REAL v[5] = {0, 0, 0, 0, 0}, x[5], y[5];
  int q0 = order == 5 ? -2 : -1, q1 = order == 5 ? 2 : 1;
  for (int q = q0; q <= q1; q++) v[q + 2] = at_dir(m, psi, dir, i, j, k, c0 + sg * q);
  const REAL *bs = v;
  if (s1) {
    for (int q = q0; q <= q1; q++) x[q + 2] = at_dir(m, s1, dir, i, j, k, c0 + sg * q);
    bs = x;
  }
  if (s2)
    for (int q = q0; q <= q1; q++) y[q + 2] = at_dir(m, s2, dir, i, j, k, c0 + sg * q);
  if (order == 5) {
    REAL b0 = beta5_0(bs[2], bs[3], bs[4]), b1 = beta5_1(bs[1], bs[2], bs[3]), b2 = beta5_2(bs[0], bs[1], bs[2]);
    if (s2) {
      b0 = (b0 + beta5_0(y[2], y[3], y[4])) / (REAL)2;
      b1 = (b1 + beta5_1(y[1], y[2], y[3])) / (REAL)2;
      b2 = (b2 + beta5_2(y[0], y[1], y[2])) / (REAL)2;
    }
    return weno5_combine(v, b0, b1, b2);
  } else {
    REAL b0 = beta3(bs[2], bs[3]), b1 = beta3(bs[1], bs[2]);
    if (s2) {
      b0 = (b0 + beta3(y[2], y[3])) / (REAL)2;
      b1 = (b1 + beta3(y[1], y[2])) / (REAL)2;
    }
    return weno3_combine(v, b0, b1);
  }
}

/* Symmetric (centred) interpolation used for advecting velocities / cross terms:
 * the WENO5 scheme's advecting_velocity_scheme is Centered(order=4); next to walls in
 * bounded directions it drops to second order. */
static REAL sym_interp(const model *m, int dir, int target, int i, int j, int k, fn3 psi) {
  int idx = dir == DX ? i : dir == DY ? j : k;
  int order = (dir == DX && !m->immersed) ? 4 : stencil_active(m, dir, target, i, j, k, 3) ? 4 : 2;
  int b = target == TO_FACE ? idx - 1 : idx; /* the lower of the two central points */
  if (order == 2) return (at_dir(m, psi, dir, i, j, k, b) + at_dir(m, psi, dir, i, j, k, b + 1)) / (REAL)2;
  return (-at_dir(m, psi, dir, i, j, k, b - 1) + (REAL)7 * at_dir(m, psi, dir, i, j, k, b) +
          (REAL)7 * at_dir(m, psi, dir, i, j, k, b + 1) - at_dir(m, psi, dir, i, j, k, b + 2)) / (REAL)12;
}

/* ---------------------------------------------------------------- halos
 * tupled_fill_halo_regions!(prognostic_fields(model), ...) -- /root/reference/src/precompile.jl:35,40,44-46.
 * Default boundary conditions (appendix A.3): bounded y / z fill ONE halo layer
 * (zero-gradient for centre-located axes, zero wall-normal velocity); periodic x is
 * filled last over the whole parent extent so corners are consistent. */
static void fill_periodic_x(const model *m, fld *F) {
  int H = m->H, Nx = m->Nx;
  for (long r = 0; r < (long)F->sy * F->sz; r++) {
    REAL *row = F->p + r * F->sx;
    for (int q = 0; q < H; q++) {
      row[q] = row[Nx + q];         /* west halo  <- east interior */
      row[H + Nx + q] = row[H + q]; /* east halo  <- west interior */
    }
  }
}
/* Zipper fold at the northern edge of the tripolar grid: Oceananigans' fold_north_{center,face}_{center,face}! of its
 * zipper boundary condition, restated from memory of v0.96 [UPSTREAM-UNVERIFIED].  The fold pivots on the CENTRES of row
 * Ny ("the Ny line is duplicated"), between the two north poles, which sit on the x faces i = 1 and i = Nx/2 + 1:
 *   (c,c):  c[i, Ny+j] = s c[Nx-i+1, Ny-j]            (f,c):  c[i, Ny+j] = s' c[i', Ny-j],   i' = Nx-i+2
 *   (c,f):  c[i, Ny+j] = s c[Nx-i+1, Ny-j+1]          where i' > Nx wraps to i' - Nx and takes s' = |s| ("for periodic
 *                                                      elements we change the sign"), s' = s otherwise,
 * j = 1..H, s = -1 for vector components.  Row Ny itself is held twice -- cell (i, Ny) IS cell (Nx-i+1, Ny) -- and both
 * copies are stepped; neither is overwritten with the other (as the fill functions of v0.96 are recalled; a later upstream
 * fix that slaves one copy to the other is not restated). */
static inline int fold_i(const model *m, int i, int xface) {
  int ip = xface ? m->Nx - i + 2 : m->Nx - i + 1;
  if (ip > m->Nx) ip -= m->Nx;
  return ip;
}
static inline REAL fold_sign(const model *m, int i, int xface, REAL sgn) {
  return (xface && m->Nx - i + 2 > m->Nx && sgn < 0) ? -sgn : sgn;
}
static void fold_rows_levels(model *m, int id, int twod, int is_v, int xface, REAL sgn, int nlev) {
  int Nx = m->Nx, Ny = m->Ny, H = m->H, k0 = 1, k1 = twod ? 1 : nlev;
  for (int k = k0; k <= k1; k++) {
    REAL *base = twod ? m->f[id].p : m->f[id].p + (long)m->f[id].sx * m->f[id].sy * (k - 1 + H);
#define AF(i, j) base[((long)(i)-1 + H) + (long)m->f[id].sx * ((long)(j)-1 + H)]
    if (is_v) {
      for (int q = 1; q <= H; q++)
        for (int i = 1; i <= Nx; i++) AF(i, Ny + q) = sgn * AF(fold_i(m, i, 0), Ny + 1 - q);
    } else {
      for (int q = 1; q <= H; q++)
        for (int i = 1; i <= Nx; i++) AF(i, Ny + q) = fold_sign(m, i, xface, sgn) * AF(fold_i(m, i, xface), Ny - q);
      /* option (the later upstream fix, as recalled): the pivot row is held twice -- its eastern half becomes the image of
       * its western half (cells i > Nx/2; x faces i > Nx/2 + 1: the face Nx/2 + 1 is a pole and its own image) */
      if (m->fold_pivot_slaved)
        for (int i = Nx / 2 + 1 + (xface ? 1 : 0); i <= Nx; i++) AF(i, Ny) = fold_sign(m, i, xface, sgn) * AF(fold_i(m, i, xface), Ny);
    }
#undef AF
  }
}
/* xface: located on x faces (u, U, G.U); sgn: -1 for vector components */
static void fold_rows(model *m, int id, int twod, int is_v, int xface, REAL sgn) {
  fold_rows_levels(m, id, twod, is_v, xface, sgn, m->Nz);
}
static void fill_halo_3d(model *m, int id, int is_v, int xface, REAL sgn) {
  int Nx = m->Nx, Ny = m->Ny, Nz = m->Nz;
  for (int k = 1; k <= Nz; k++)
    for (int i = 1; i <= Nx; i++) {
      if (is_v) {
        A3(id, i, 1, k) = 0;
        if (!m->north_fold) A3(id, i, Ny + 1, k) = 0;
      } else {
        A3(id, i, 0, k) = A3(id, i, 1, k);
        if (!m->north_fold) A3(id, i, Ny + 1, k) = A3(id, i, Ny, k);
      }
    }
  if (m->north_fold) fold_rows(m, id, 0, is_v, xface, sgn);
  /* bottom / top layer; with the fold also of the rows beyond it (the pressure of those rows enters the pressure
   * gradient on the fold line: its vertical integral starts in the top halo level) */
  for (int j = 1; j <= Ny + (m->north_fold ? m->H : 0); j++)
    for (int i = 1; i <= Nx; i++) {
      A3(id, i, j, 0) = A3(id, i, j, 1);
      A3(id, i, j, Nz + 1) = A3(id, i, j, Nz);
    }
  fill_periodic_x(m, &m->f[id]);
}
static void fill_halo_2d(model *m, int id, int is_v, int xface, REAL sgn) {
  int Nx = m->Nx, Ny = m->Ny;
  for (int i = 1; i <= Nx; i++) {
    if (is_v) {
      A2(id, i, 1) = 0;
      if (!m->north_fold) A2(id, i, Ny + 1) = 0;
    } else {
      A2(id, i, 0) = A2(id, i, 1);
      if (!m->north_fold) A2(id, i, Ny + 1) = A2(id, i, Ny);
    }
  }
  if (m->north_fold) fold_rows(m, id, 1, is_v, xface, sgn);
  fill_periodic_x(m, &m->f[id]);
}
void FN(fill_halos)(void *h) {
  model *m = (model *)h;
  fill_halo_3d(m, F_U, 0, 1, -1);
  fill_halo_3d(m, F_V, 1, 0, -1);
  fill_halo_3d(m, F_T, 0, 0, 1);
  fill_halo_3d(m, F_S, 0, 0, 1);
  fill_halo_2d(m, F_ETA, 0, 0, 1);
  fill_halo_2d(m, F_BU, 0, 1, -1);
  fill_halo_2d(m, F_BV, 1, 0, -1);
  if (m->catke) fill_halo_3d(m, F_E, 0, 0, 1);
}

/* ---------------------------------------------------------------- auxiliaries
 * compute_auxiliaries!(model) -- /root/reference/src/precompile.jl:36,113-115.
 * Both kernels run on the extended range -H+2 : N+H-1 in x and y (appendix A.5). */
void FN(compute_w)(void *h) {
  model *m = (model *)h;
  int H = m->H;
#pragma omp parallel for schedule(static)
  for (int j = -H + 2; j <= m->Ny + H - 1; j++)
    for (int i = -H + 2; i <= m->Nx + H - 1; i++) {
      A3(F_W, i, j, 1) = 0;
      for (int k = 2; k <= m->Nz + 1; k++) {
        REAL dh = f_div(m, i, j, k - 1) / AZCC(i, j);
        A3(F_W, i, j, k) = A3(F_W, i, j, k - 1) - dh;
      }
    }
}
void FN(compute_p)(void *h) {
  model *m = (model *)h;
  int H = m->H, Nz = m->Nz;
#pragma omp parallel for schedule(static)
  for (int j = -H + 2; j <= m->Ny + H - 1; j++)
    for (int i = -H + 2; i <= m->Nx + H - 1; i++) {
      PREAL bup = buoyancy(m, i, j, Nz + 1);
      PREAL bk = buoyancy(m, i, j, Nz);
      PREAL pk = -((bk + bup) / (PREAL)2) * (PREAL)DZF(Nz + 1);
      A3(F_P, i, j, Nz) = (REAL)pk;
      for (int k = Nz - 1; k >= 1; k--) {
        bup = bk;
        bk = buoyancy(m, i, j, k);
        pk = pk - ((bk + bup) / (PREAL)2) * (PREAL)DZF(k + 1);
        A3(F_P, i, j, k) = (REAL)pk;
      }
    }
}
void FN(compute_auxiliaries)(void *h) {
  FN(compute_w)(h);
  FN(compute_p)(h);
}

/* ---------------------------------------------------------------- tendencies
 * compute_tendencies!(model, callbacks) -- /root/reference/src/precompile.jl:38,48-50,63-111.
 * Momentum: WENOVectorInvariant(order=5) with VelocityStencil vorticity smoothness and
 * OnlySelfUpwinding(cross_scheme = WENO5) (appendix A.11), HydrostaticSphericalCoriolis
 * (enstrophy conserving), hydrostatic pressure gradient.  Tracers: WENO(order=5) flux form. */
static REAL Gu_at(const model *m, int i, int j, int k) {
  /* advecting v at (f,c,c) */
  REAL vhat = ((DXCF(i - 1, j) * A3(F_V, i - 1, j, k) + DXCF(i - 1, j + 1) * A3(F_V, i - 1, j + 1, k)) / (REAL)2 +
               (DXCF(i, j) * A3(F_V, i, j, k) + DXCF(i, j + 1) * A3(F_V, i, j + 1, k)) / (REAL)2) / (REAL)2 / DXFC(i, j);
  REAL zetaR = biased_interp(m, DY, TO_CENTER, i, j, k, vhat > 0, f_zeta, f_uy, f_vx);
  REAL hadv = -vhat * zetaR;
  /* vertical advection: upwinded divergence flux + vertical flux divergence */
  REAL uhat = A3(F_U, i, j, k);
  REAL dvs = sym_interp(m, DX, TO_FACE, i, j, k, f_dyV);
  REAL duR = biased_interp(m, DX, TO_FACE, i, j, k, uhat > 0, f_dxU, f_div, NULL);
  REAL phi = uhat * (dvs + duR);
  REAL fz[2];
  for (int t = 0; t < 2; t++) {
    int kk = k + t;
    REAL wt = sym_interp(m, DX, TO_FACE, i, j, kk, f_Azw);
    REAL uR = biased_interp(m, DZ, TO_FACE, i, j, kk, wt > 0, f_u, NULL, NULL);
    fz[t] = wt * uR;
  }
  REAL vadv = (phi + (fz[1] - fz[0])) / (AZFC(i, j) * DZC(k));
  /* Bernoulli head */
  REAL dKu = biased_interp(m, DX, TO_FACE, i, j, k, uhat > 0, f_dxu2, f_usm, NULL);
  REAL dKv = sym_interp(m, DY, TO_CENTER, i, j, k, f_dxv2);
  REAL bern = (dKu + dKv) / DXFC(i, j);
  /* Coriolis: x_f_cross_U = -Iy(f) * vhat */
  REAL fbar = (FFF(i, j) + FFF(i, j + 1)) / (REAL)2;
  REAL cor = -fbar * vhat;
  REAL dpdx = (A3(F_P, i, j, k) - A3(F_P, i - 1, j, k)) / DXFC(i, j);
  return -(hadv + vadv + bern) - cor - dpdx;
}
static REAL Gv_at(const model *m, int i, int j, int k) {
  /* advecting u at (c,f,c) */
  REAL uhat = ((DYFC(i, j - 1) * A3(F_U, i, j - 1, k) + DYFC(i + 1, j - 1) * A3(F_U, i + 1, j - 1, k)) / (REAL)2 +
               (DYFC(i, j) * A3(F_U, i, j, k) + DYFC(i + 1, j) * A3(F_U, i + 1, j, k)) / (REAL)2) / (REAL)2 / DYCF(i, j);
  REAL zetaR = biased_interp(m, DX, TO_CENTER, i, j, k, uhat > 0, f_zeta, f_uy, f_vx);
  REAL hadv = uhat * zetaR;
  REAL vhat = A3(F_V, i, j, k);
  REAL dus = sym_interp(m, DY, TO_FACE, i, j, k, f_dxU);
  REAL dvR = biased_interp(m, DY, TO_FACE, i, j, k, vhat > 0, f_dyV, f_div, NULL);
  REAL phi = vhat * (dus + dvR);
  REAL fz[2];
  for (int t = 0; t < 2; t++) {
    int kk = k + t;
    REAL wt = sym_interp(m, DY, TO_FACE, i, j, kk, f_Azw);
    REAL vR = biased_interp(m, DZ, TO_FACE, i, j, kk, wt > 0, f_v, NULL, NULL);
    fz[t] = wt * vR;
  }
  REAL vadv = (phi + (fz[1] - fz[0])) / (AZCF(i, j) * DZC(k));
  REAL dKv = biased_interp(m, DY, TO_FACE, i, j, k, vhat > 0, f_dyv2, f_vsm, NULL);
  REAL dKu = sym_interp(m, DX, TO_CENTER, i, j, k, f_dyu2);
  REAL bern = (dKv + dKu) / DYCF(i, j);
  REAL cor = ((FFF(i, j) + FFF(i + 1, j)) / (REAL)2) * uhat;
  REAL dpdy = (A3(F_P, i, j, k) - A3(F_P, i, j - 1, k)) / DYCF(i, j);
  return -(hadv + vadv + bern) - cor - dpdy;
}
void FN(compute_momentum_tendencies)(void *h) {
  model *m = (model *)h;
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 1; k <= m->Nz; k++)
    for (int j = 1; j <= m->Ny; j++)
      for (int i = 1; i <= m->Nx; i++) {
        /* zero at immersed peripheral nodes (faces that touch the solid): the velocity there is masked to zero and
         * stays zero.  [restatement choice: upstream evaluates the kernel there too and discards the result through
         * mask_immersed_field!; its integrated tendencies use exactly this mask] */
        A3(F_GNU, i, j, k) = immersed_peripheral_u(m, i, j, k) ? (REAL)0 : Gu_at(m, i, j, k);
        A3(F_GNV, i, j, k) = immersed_peripheral_v(m, i, j, k) ? (REAL)0 : Gv_at(m, i, j, k);
      }
}
static REAL tracer_flux(const model *m, int dir, int i, int j, int k, fn3 c) {
  if (dir == DX) {
    REAL u = A3(F_U, i, j, k);
    return DYFC(i, j) * DZC(k) * u * tracer_interp(m, DX, i, j, k, u > 0, c);
  } else if (dir == DY) {
    REAL v = A3(F_V, i, j, k);
    return DXCF(i, j) * DZC(k) * v * tracer_interp(m, DY, i, j, k, v > 0, c);
  } else {
    REAL w = A3(F_W, i, j, k);
    return AZCC(i, j) * w * tracer_interp(m, DZ, i, j, k, w > 0, c);
  }
}
static void tracer_tendency(model *m, int gid, fn3 c) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 1; k <= m->Nz; k++)
    for (int j = 1; j <= m->Ny; j++)
      for (int i = 1; i <= m->Nx; i++) {
        REAL div = (tracer_flux(m, DX, i + 1, j, k, c) - tracer_flux(m, DX, i, j, k, c)) +
                   (tracer_flux(m, DY, i, j + 1, k, c) - tracer_flux(m, DY, i, j, k, c)) +
                   (tracer_flux(m, DZ, i, j, k + 1, c) - tracer_flux(m, DZ, i, j, k, c));
        A3(gid, i, j, k) = -(div / (AZCC(i, j) * DZC(k)));
      }
}
void FN(compute_tracer_tendencies)(void *h) {
  model *m = (model *)h;
  tracer_tendency(m, F_GNT, f_T);
  tracer_tendency(m, F_GNS, f_S);
}
/* compute_hydrostatic_boundary_tendency_contributions!(Gn, arch, velocities, tracers, clock, fields, closure, buoyancy)
 * -- /root/reference/src/precompile.jl:25,52-61: flux boundary conditions enter the tendencies of the cells next to
 * the boundary; here the top ones (apply_z_top_bc!, restated): G[i,j,Nz] -= J Az / V. */
static int first_free_level(const model *m, int which, int i, int j);
/* Quadratic bottom drag, the bottom boundary condition ClimaOcean's ocean_simulation gives u and v (and their immersed
 * bottoms): FluxBoundaryCondition(-Cd u sqrt(u^2 + Ixy(v)^2)) at (f,c,c), likewise for v at (c,f,c) [UPSTREAM-UNVERIFIED].
 * The flux is positive upward, i.e. INTO the column through its bottom face: G[kbottom] += J / dz (apply_z_bottom_bc!). */
static void compute_bottom_drag_fluxes(model *m) {
  const REAL Cd = m->bottom_drag;
  for (int q = 0; q < 2; q++)
    if (!m->bottom_flux[q]) m->bottom_flux[q] = (REAL *)calloc((size_t)m->f[q ? F_GNV : F_GNU].sx * m->f[q ? F_GNV : F_GNU].sy, sizeof(REAL));
  long sxu = m->f[F_GNU].sx, sxv = m->f[F_GNV].sx;
#pragma omp parallel for schedule(static)
  for (int j = 1; j <= NYV; j++)
    for (int i = 1; i <= m->Nx; i++) {
      if (j <= m->Ny) {
        int k = first_free_level(m, 0, i, j);
        REAL J = 0;
        if (k <= m->Nz) {
          REAL u = A3(F_U, i, j, k);
          REAL vb = (A3(F_V, i - 1, j, k) + A3(F_V, i, j, k) + A3(F_V, i - 1, j + 1, k) + A3(F_V, i, j + 1, k)) / (REAL)4;
          J = -Cd * u * (REAL)sqrt((double)(u * u + vb * vb));
        }
        m->bottom_flux[0][((long)i - 1 + HH) + sxu * ((long)j - 1 + HH)] = J;
      }
      {
        int k = j >= 2 ? first_free_level(m, 1, i, j) : m->Nz + 1;
        REAL J = 0;
        if (k <= m->Nz) {
          REAL v = A3(F_V, i, j, k);
          REAL ub = (A3(F_U, i, j - 1, k) + A3(F_U, i + 1, j - 1, k) + A3(F_U, i, j, k) + A3(F_U, i + 1, j, k)) / (REAL)4;
          J = -Cd * v * (REAL)sqrt((double)(v * v + ub * ub));
        }
        m->bottom_flux[1][((long)i - 1 + HH) + sxv * ((long)j - 1 + HH)] = J;
      }
    }
}
void FN(set_bottom_drag)(void *h, double Cd) { ((model *)h)->bottom_drag = (REAL)Cd; }
void FN(set_tracer_advection_order)(void *h, int order) { ((model *)h)->tracer_order = order; }
void FN(compute_boundary_tendencies)(void *h) {
  model *m = (model *)h;
  const int gid[4] = {F_GNU, F_GNV, F_GNT, F_GNS};
  if (m->bottom_drag != 0) {
    compute_bottom_drag_fluxes(m);
    long sxu = m->f[F_GNU].sx, sxv = m->f[F_GNV].sx;
    for (int j = 1; j <= NYV; j++)
      for (int i = 1; i <= m->Nx; i++) {
        int ku = j <= m->Ny ? first_free_level(m, 0, i, j) : m->Nz + 1, kv = j >= 2 ? first_free_level(m, 1, i, j) : m->Nz + 1;
        if (ku <= m->Nz) A3(F_GNU, i, j, ku) = A3(F_GNU, i, j, ku) + m->bottom_flux[0][((long)i - 1 + HH) + sxu * ((long)j - 1 + HH)] / DZC(ku);
        if (kv <= m->Nz) A3(F_GNV, i, j, kv) = A3(F_GNV, i, j, kv) + m->bottom_flux[1][((long)i - 1 + HH) + sxv * ((long)j - 1 + HH)] / DZC(kv);
      }
  }
  for (int q = 0; q < 4; q++) {
    if (!m->top_flux[q]) continue;
    const fld *F = &m->f[gid[q]];
    for (int j = 1; j <= (q == 1 ? NYV : m->Ny); j++)
      for (int i = 1; i <= m->Nx; i++) {
        if (q == 0 && immersed_peripheral_u(m, i, j, m->Nz)) continue;
        if (q == 1 && (j == 1 || immersed_peripheral_v(m, i, j, m->Nz))) continue;
        if (q >= 2 && inactive_cell(m, i, j, m->Nz)) continue;
        REAL J = m->top_flux[q][((long)i - 1 + HH) + (long)F->sx * ((long)j - 1 + HH)];
        A3(gid[q], i, j, m->Nz) = A3(gid[q], i, j, m->Nz) - J / DZC(m->Nz);
      }
  }
}
/* q: 0 u, 1 v, 2 T, 3 S; J: interior values (Nx x Ny (+1 for v), i fastest) or NULL for the default no-flux */
void FN(set_top_flux)(void *h, int q, const double *J) {
  model *m = (model *)h;
  const int gid[4] = {F_GNU, F_GNV, F_GNT, F_GNS};
  free(m->top_flux[q]);
  m->top_flux[q] = NULL;
  if (!J) return;
  const fld *F = &m->f[gid[q]];
  m->top_flux[q] = (REAL *)calloc((size_t)F->sx * F->sy, sizeof(REAL));
  int ny = m->Ny + ((q == 1 && !m->north_fold) ? 1 : 0);
  for (int j = 1; j <= ny; j++)
    for (int i = 1; i <= m->Nx; i++)
      m->top_flux[q][((long)i - 1 + HH) + (long)F->sx * ((long)j - 1 + HH)] = (REAL)J[(i - 1) + (long)m->Nx * (j - 1)];
}
void FN(get_top_flux)(void *h, int q, double *J) {
  model *m = (model *)h;
  const int gid[4] = {F_GNU, F_GNV, F_GNT, F_GNS};
  const fld *F = &m->f[gid[q]];
  int ny = m->Ny + ((q == 1 && !m->north_fold) ? 1 : 0);
  for (int j = 1; j <= ny; j++)
    for (int i = 1; i <= m->Nx; i++)
      J[(i - 1) + (long)m->Nx * (j - 1)] = m->top_flux[q] ? (double)m->top_flux[q][((long)i - 1 + HH) + (long)F->sx * ((long)j - 1 + HH)] : 0.0;
}
static void catke_tke_tendency(model *m);
void FN(compute_tendencies)(void *h) {
  FN(compute_momentum_tendencies)(h);
  FN(compute_tracer_tendencies)(h);
  FN(compute_boundary_tendencies)(h);
  if (((model *)h)->catke) catke_tke_tendency((model *)h);
}
/* update_state!(model; compute_tendencies=true): phases 1-5 of /root/reference/src/precompile.jl:34-38
 * (mask_immersed and diffusivity halos are no-ops for this configuration). */
/* mask_immersed_model_fields!(model, grid) -- /root/reference/src/precompile.jl:21,34: prognostic fields are set to
 * zero at peripheral nodes of their location (u, v: faces that touch an inactive cell; T, S: inactive cells); the
 * barotropic transports where the column of their face has no depth. */
void FN(mask_immersed_fields)(void *h) {
  model *m = (model *)h;
  for (int k = 1; k <= m->Nz; k++)
    for (int j = 1; j <= m->Ny + 1; j++)
      for (int i = 1; i <= m->Nx; i++) {
        if ((j <= NYV || !m->north_fold) && peripheral_v(m, i, j, k)) A3(F_V, i, j, k) = 0;
        if (j > m->Ny) continue;
        if (peripheral_u(m, i, j, k)) A3(F_U, i, j, k) = 0;
        if (inactive_cell(m, i, j, k)) A3(F_T, i, j, k) = A3(F_S, i, j, k) = 0;
      }
  for (int j = 1; j <= m->Ny + 1; j++)
    for (int i = 1; i <= m->Nx; i++) {
      if (j <= m->Ny && H2(Hfc, i, j) == 0) A2(F_BU, i, j) = 0;
      if (j == 1 || j > NYV || H2(Hcf, i, j) == 0) A2(F_BV, i, j) = 0;
    }
}
static void catke_compute_diffusivities(model *m);
void FN(update_state)(void *h) {
  FN(mask_immersed_fields)(h);
  FN(fill_halos)(h);
  FN(compute_auxiliaries)(h);
  if (((model *)h)->catke) catke_compute_diffusivities((model *)h);   /* compute_diffusivities! + their halos (a14) */
  FN(compute_tendencies)(h);
}

/* ---------------------------------------------------------------- vertically implicit diffusion
 * implicit_step!(field, implicit_solver, closure, ...) after the explicit AB2 update of each prognostic field
 * (Oceananigans TimeSteppers / TurbulenceClosures.vertically_implicit_diffusion_solver, restated [UPSTREAM-UNVERIFIED]):
 * solve (1 - dt d/dz K d/dz) phi_new = phi_star per column with the batched tridiagonal (Thomas) solver,
 *   lower_k = -dt K^f_k     / (dz^c_k dz^f_k)      (face k: between cells k-1 and k)
 *   upper_k = -dt K^f_{k+1} / (dz^c_k dz^f_{k+1})
 *   diag_k  = 1 - lower_k - upper_k,
 * no flux through the bottom face of the first free level and through the top face (flux boundary conditions enter the
 * explicit tendency, compute_boundary_tendencies).  kfirst (1-based): first level of the column that is solved; levels
 * below it (immersed cells, faces that touch the solid) are left alone. */
static void implicit_column(const model *m, REAL *col, long stride, int kfirst, REAL K, REAL dt) {
  int Nz = m->Nz;
  if (kfirst > Nz || K == 0) return;
  REAL gam[512], bet = 1, prev = 0;
  for (int k = kfirst; k <= Nz; k++) {
    REAL lo = (k == kfirst) ? 0 : -dt * K / (DZC(k) * DZF(k));
    REAL up = (k == Nz) ? 0 : -dt * K / (DZC(k) * DZF(k + 1));
    REAL dg = (REAL)1 - lo - up;
    if (k == kfirst) {
      bet = dg;
      prev = col[(k - 1) * stride] / bet;
    } else {
      REAL up_below = -dt * K / (DZC(k - 1) * DZF(k));   /* upper coefficient of the level below */
      gam[k] = up_below / bet;
      bet = dg - lo * gam[k];
      prev = (col[(k - 1) * stride] - lo * prev) / bet;
    }
    col[(k - 1) * stride] = prev;
  }
  for (int k = Nz - 1; k >= kfirst; k--) col[(k - 1) * stride] -= gam[k + 1] * col[k * stride];
}
/* first level (1-based) from which the column of field `which` (0 u, 1 v, 2 tracers) at (i, j) is free */
static int first_free_level(const model *m, int which, int i, int j) {
  int k = 1;
  while (k <= m->Nz && (which == 0 ? peripheral_u(m, i, j, k) : which == 1 ? peripheral_v(m, i, j, k) : inactive_cell(m, i, j, k))) k++;
  return k;
}
static void implicit_step_field(model *m, int id, int which, REAL K, REAL dt) {
  if (K == 0) return;
  const fld *F = &m->f[id];
  long stride = (long)F->sx * F->sy;
  int nyrows = (which == 1) ? NYV : m->Ny;
#pragma omp parallel for schedule(static)
  for (int j = (which == 1 ? 2 : 1); j <= nyrows; j++)   /* (v on the southern wall face stays zero) */
    for (int i = 1; i <= m->Nx; i++)
      implicit_column(m, &A3(id, i, j, 1), stride, first_free_level(m, which, i, j), K, dt);
}
/* the same solve with CATKE's diffusivity fields: kappa at the faces of the column -- kappa_u averaged to the u / v
 * column (which = 0 / 1), kappa_c (2), kappa_e with the implicit linear term L^e on the diagonal (3) */
static inline REAL catke_kface(const model *m, int which, int i, int j, int k) {
  if (which == 0) return (A3(F_KU, i - 1, j, k) + A3(F_KU, i, j, k)) / (REAL)2;
  if (which == 1) return (A3(F_KU, i, j - 1, k) + A3(F_KU, i, j, k)) / (REAL)2;
  return which == 2 ? A3(F_KC, i, j, k) : A3(F_KE, i, j, k);
}
static void implicit_column_catke(const model *m, REAL *col, long stride, int kfirst, int which, int i, int j, REAL dt) {
  int Nz = m->Nz;
  if (kfirst > Nz) return;
  REAL gam[512], bet = 1, prev = 0;
  for (int k = kfirst; k <= Nz; k++) {
    REAL lo = (k == kfirst) ? 0 : -dt * catke_kface(m, which, i, j, k) / (DZC(k) * DZF(k));
    REAL up = (k == Nz) ? 0 : -dt * catke_kface(m, which, i, j, k + 1) / (DZC(k) * DZF(k + 1));
    REAL dg = (REAL)1 - lo - up - (which == 3 ? dt * A3(F_LE, i, j, k) : 0);
    if (k == kfirst) {
      bet = dg;
      prev = col[(k - 1) * stride] / bet;
    } else {
      REAL up_below = -dt * catke_kface(m, which, i, j, k) / (DZC(k - 1) * DZF(k));
      gam[k] = up_below / bet;
      bet = dg - lo * gam[k];
      prev = (col[(k - 1) * stride] - lo * prev) / bet;
    }
    col[(k - 1) * stride] = prev;
  }
  for (int k = Nz - 1; k >= kfirst; k--) col[(k - 1) * stride] -= gam[k + 1] * col[k * stride];
}
static void implicit_step_field_catke(model *m, int id, int which, REAL dt) {
  const fld *F = &m->f[id];
  long stride = (long)F->sx * F->sy;
  int nyrows = (which == 1) ? NYV : m->Ny;
#pragma omp parallel for schedule(static)
  for (int j = (which == 1 ? 2 : 1); j <= nyrows; j++)
    for (int i = 1; i <= m->Nx; i++)
      implicit_column_catke(m, &A3(id, i, j, 1), stride, first_free_level(m, which > 2 ? 2 : which, i, j), which, i, j, dt);
}
void FN(set_vertical_diffusivity)(void *h, double nu, double kappa) {
  model *m = (model *)h;
  m->nu = (REAL)nu;
  m->kappa = (REAL)kappa;
}

/* ---------------------------------------------------------------- CATKE
 * closure = CATKEVerticalDiffusivity() (/root/reference/sharding/less_simple_sharding_problem.jl:84-93,
 * /root/reference/src/baroclinic_instability_model.jl:30,50-51; compared fields /root/reference/src/correctness.jl:60-67).
 * Restated [UPSTREAM-UNVERIFIED] in the STRUCTURE of Oceananigans 0.96 (TurbulenceClosures/.../TKEBasedVerticalDiffusivities:
 * catke_vertical_diffusivity.jl, catke_mixing_length.jl, catke_equation.jl, time_step_catke_equation.jl,
 * tke_top_boundary_condition.jl) as recalled -- the package is not in /root/reference -- with the calibrated constants of
 * Wagner et al. (2025), "Formulation and calibration of CATKE, a one-equation parameterization for microscale ocean mixing".
 *
 * compute_diffusivities!(diffusivities, closure::CATKE, model), called by compute_auxiliaries! inside update_state!:
 *   1. dt_c = clock.time - previous_compute_time                       (the time the surface-flux filter advances by)
 *   2. if isfinite(clock.last_dt): time_step_catke_equation!(model)    -- GB-25 sets clock.last_dt = dt when it builds the model
 *        (src/baroclinic_instability_model.jl:82), so e is stepped in EVERY update_state!, the first one included.
 *        substep_turbulent_kinetic_energy! per cell, from the state as it is (kappa_u, kappa_c, J^b: the fields of the
 *        previous compute; e, T, S, u, v: current; u-, v-: the velocities of the previous compute):
 *          kappa_e  <- kappa_e(c,c,f) of the current state
 *          wb       = Iz(-kappa_c N^2);  P = shear_production(kappa_u; u-, u+, v-, v+)
 *          L^e      = min(wb, 0)/e [e > e_min] - omega - [on the bottom] C^W_eps sqrt(max(e, 0))/dz,
 *                     omega = e < 0 ? 1/tau_neg : sqrt|e| / l_D(c,c,c)
 *          G_total  = G^n.e (the "slow" tendency: advection + the top TKE flux, left by compute_tendencies!) + P + max(wb, 0)
 *          e       += dt ((3/2 + chi) G_total - (1/2 + chi) G^-.e);  G^-.e <- G_total        (chi = 0.1 always: no Euler step here)
 *        then implicit_step!(e): (1 - dt dz kappa_e dz - dt L^e) e_new = e.
 *        ab2_step! skips e and cache_previous_tendencies! skips G^-.e.
 *   3. u-, v- <- u, v (parents)
 *   4. compute_average_surface_buoyancy_flux!: J^b* = g (alpha J^T - beta J^S) from the top flux boundary conditions,
 *        J^b+ = max(J^b_min, J^b, J^b*), t* = cbrt(l_D(i,j,Nz)^2 / J^b+), eps = dt_c / t*, J^b <- (J^b + eps J^b*)/(1 + eps)
 *   5. compute_CATKE_diffusivities!: kappa_u, kappa_c, kappa_e at the faces k = 1 .. Nz from the new e.
 * then fill_halo_regions!(diffusivity_fields; only_local_halos = true) (/root/reference/src/precompile.jl:37,117-119).
 *
 * Mixing lengths at (c,c,f), with w* = Iz sqrt(max(e_min, e)), w*^2 = Iz max(e_min, e), w*^3 = Iz max(e_min, e)^(3/2),
 * N^2 = g (alpha(Iz T, Iz S, z_f) dz T - beta dz S) (SeawaterBuoyancy's dz_b; differences across an immersed face vanish),
 * S^2 = Ix (dz u)^2 + Iy (dz v)^2, Ri = N^2/S^2 (0 where N^2 = 0):
 *   stable length     l* = min(C^s depth, C^b height above the bottom, wstar / N [N^2 > 0])
 *   stability fn      sigma_psi(Ri) = C^un_psi (Ri < 0);  C^lo_psi + (C^hi_psi - C^lo_psi) clamp((Ri - CRi0)/CRid, 0, 1) (Ri >= 0)
 *   convective length l^c = max(0, eps^sp C^c_psi w*^3/(J^b + J^b_min)) where J^b > J^b_min and N^2 < 0,
 *                     l^e = max(0, eps^sp C^e_psi J^b/(w* N^2 + J^b_min)) where J^b > J^b_min, N^2 > 0 and N^2 < 0 on the face above,
 *                     eps^sp = 1 - C^sp sqrt(S^2) w*^2/(J^b + J^b_min)
 *   l_psi = min(H, max(sigma_psi l*, l^conv)), H = the static column depth;   kappa_psi = l_psi w*
 * Dissipation length at (c,c,c): min(H, max(lstar(c,c,c) / sigma_D(Ri(c,c,c)), Iz l^conv(C^c_D, C^e_D))), with N^2 and S^2
 * averaged to the centre and w* = sqrt(max(e_min, e)) of the cell.
 *
 * Choices this restatement had to make where the recalled source leaves room (a Julia dump of the `catke_*` case of
 * tools/dump_goldens.jl settles them): the in-place update of e inside substep_turbulent_kinetic_energy! is read as
 * "every cell sees the OLD e of its neighbours" (what the traced/functional Reactant program computes; a serial CPU loop
 * over k would see the new e of the level below); the halos of e are refilled after the e step (option, default on:
 * upstream, as recalled, leaves them one e step stale until the next update_state!; a decomposition cannot reproduce that
 * at its internal boundaries); L^e = 0 and J^b = 0 in immersed cells / land columns (upstream would divide 0 by 0 there). */
typedef struct {
  REAL Cs, Cb, Csp, CRid, CRi0;
  REAL Chi[4], Clo[4], Cun[4], Cc[4], Ce[4]; /* psi = u, c, e, D */
  REAL CWu, CWw, emin, Jbmin, tau_neg, CWeps;
} catke_par;
static const catke_par CATKE_DEFAULT = {
  (REAL)1.131, (REAL)0.28, (REAL)0.505, (REAL)1.02, (REAL)0.254,
  {(REAL)0.242, (REAL)0.098, (REAL)0.548, (REAL)0.579},
  {(REAL)0.361, (REAL)0.198, (REAL)7.863, (REAL)1.604},
  {(REAL)0.370, (REAL)0.369, (REAL)1.447, (REAL)0.923},
  {(REAL)3.705, (REAL)4.793, (REAL)3.642, (REAL)3.254},
  {(REAL)0.0, (REAL)0.112, (REAL)0.0, (REAL)0.0},
  (REAL)3.179, (REAL)0.383, (REAL)1e-9, (REAL)1e-11, (REAL)60.0, (REAL)1.0};

static inline const catke_par *catke_parameters_of(const model *m) {
  return m->catke_params ? (const catke_par *)m->catke_params : &CATKE_DEFAULT;
}
#define CATKE (*catke_parameters_of(m))
/* the parameters as 31 doubles in the order of gb25_catke_parameters (include/gb25.h): Cs, Cb, Csp, CRid, CRi0, Chi[4], Clo[4],
 * Cun[4], Cc[4], Ce[4], CWu, CWw, minimum TKE, minimum convective buoyancy flux, damping time scale of negative TKE, CWeps */
void FN(set_catke_parameters)(void *h, const double *p) {
  model *m = (model *)h;
  if (!m->catke_params) m->catke_params = malloc(sizeof(catke_par));
  catke_par *c = (catke_par *)m->catke_params;
  c->Cs = (REAL)p[0]; c->Cb = (REAL)p[1]; c->Csp = (REAL)p[2]; c->CRid = (REAL)p[3]; c->CRi0 = (REAL)p[4];
  for (int q = 0; q < 4; q++) {
    c->Chi[q] = (REAL)p[5 + q]; c->Clo[q] = (REAL)p[9 + q]; c->Cun[q] = (REAL)p[13 + q]; c->Cc[q] = (REAL)p[17 + q]; c->Ce[q] = (REAL)p[21 + q];
  }
  c->CWu = (REAL)p[25]; c->CWw = (REAL)p[26]; c->emin = (REAL)p[27]; c->Jbmin = (REAL)p[28]; c->tau_neg = (REAL)p[29];
  c->CWeps = (REAL)p[30];
}
static inline REAL catke_step(REAL x, REAL c, REAL w) {
  REAL t = (x - c) / w;
  return t < 0 ? 0 : (t > 1 ? 1 : t);
}
static inline REAL catke_sigma(const model *m, int psi, REAL Ri) {
  if (Ri < 0) return CATKE.Cun[psi];
  return CATKE.Clo[psi] + (CATKE.Chi[psi] - CATKE.Clo[psi]) * catke_step(Ri, CATKE.CRi0, CATKE.CRid);
}
/* both cells of the (c,c,f) face k (between the cells k-1 and k) are active */
static inline int catke_open_face(const model *m, int i, int j, int k) {
  return k >= 2 && k <= m->Nz && !inactive_cell(m, i, j, k - 1) && !inactive_cell(m, i, j, k);
}
/* dz_b(i, j, k, grid, ::SeawaterBuoyancy, tracers) = g (alpha dzT - beta dzS) at (c,c,f), alpha and beta at the vertically
 * averaged T, S and the depth of the face; the vertical differences of T, S vanish across a face that touches the solid, and
 * on the bottom / top faces by the no-flux halo fill */
static inline REAL catke_N2(const model *m, int i, int j, int k) {
  if (!catke_open_face(m, i, j, k)) return 0;
  PREAL Tl = (PREAL)A3(F_T, i, j, k - 1), Th = (PREAL)A3(F_T, i, j, k), Sl = (PREAL)A3(F_S, i, j, k - 1), Sh = (PREAL)A3(F_S, i, j, k);
  PREAL a, b;
  teos10_sensitivities((Tl + Th) / (PREAL)2, (Sl + Sh) / (PREAL)2, (PREAL)MK(zf, k), &a, &b);
  PREAL rdz = (PREAL)1 / (PREAL)DZF(k);
  return (REAL)((PREAL)m->g * (a * ((Th - Tl) * rdz) - b * ((Sh - Sl) * rdz)) / (PREAL)m->rho0);
}
/* dz at (f,c,f) / (c,f,f) of a velocity component: zero where one of the two nodes it differences is an inactive node (both
 * cells beside it inactive), and on the bottom / top faces */
static inline REAL catke_dzu(const model *m, int id, int i, int j, int k) {
  if (k <= 1 || k > m->Nz) return 0;
  if ((inactive_cell(m, i - 1, j, k) && inactive_cell(m, i, j, k)) || (inactive_cell(m, i - 1, j, k - 1) && inactive_cell(m, i, j, k - 1))) return 0;
  return (A3(id, i, j, k) - A3(id, i, j, k - 1)) / DZF(k);
}
static inline REAL catke_dzv(const model *m, int id, int i, int j, int k) {
  if (k <= 1 || k > m->Nz) return 0;
  if ((inactive_cell(m, i, j - 1, k) && inactive_cell(m, i, j, k)) || (inactive_cell(m, i, j - 1, k - 1) && inactive_cell(m, i, j, k - 1))) return 0;
  return (A3(id, i, j, k) - A3(id, i, j, k - 1)) / DZF(k);
}
/* shear(c,c,f) = Ix (dz u)^2 + Iy (dz v)^2 of the current velocities */
static inline REAL catke_S2(const model *m, int i, int j, int k) {
  REAL uw = catke_dzu(m, F_U, i, j, k), ue = catke_dzu(m, F_U, i + 1, j, k);
  REAL vs = catke_dzv(m, F_V, i, j, k), vn = catke_dzv(m, F_V, i, j + 1, k);
  return (uw * uw + ue * ue) / (REAL)2 + (vs * vs + vn * vn) / (REAL)2;
}
static inline REAL catke_tke_floor(const model *m, int i, int j, int k) {   /* max(minimum_tke, e) */
  REAL e = A3(F_E, i, j, k);
  return e > CATKE.emin ? e : CATKE.emin;
}
static inline REAL catke_zbottom(const model *m, int i, int j) { return MK(zf, KB(i, j) + 1); }
typedef struct { REAL ku, kc, ke, convD, N2, S2; } catke_face;
static catke_face catke_at_face(const model *m, int i, int j, int k) {
  catke_face f = {0, 0, 0, 0, 0, 0};
  const int Nz = m->Nz;
  if (k <= 1 || k > Nz) return f;
  f.S2 = catke_S2(m, i, j, k);
  if (!catke_open_face(m, i, j, k)) return f;   /* (there l* = 0 -- no height above the bottom -- and N^2 = 0: every length vanishes) */
  const REAL el = catke_tke_floor(m, i, j, k - 1), eh = catke_tke_floor(m, i, j, k);
  const REAL wl = (REAL)sqrt((double)el), wh = (REAL)sqrt((double)eh);
  const REAL ws = (wl + wh) / (REAL)2, ws2 = (wl * wl + wh * wh) / (REAL)2, ws3 = (wl * wl * wl + wh * wh * wh) / (REAL)2;
  const REAL N2 = catke_N2(m, i, j, k), N2above = catke_N2(m, i, j, k + 1), S2 = f.S2;
  f.N2 = N2;
  const REAL Ri = (N2 == 0) ? 0 : N2 / S2;
  REAL d_up = CATKE.Cs * (MK(zf, Nz + 1) - MK(zf, k)), d_dn = CATKE.Cb * (MK(zf, k) - catke_zbottom(m, i, j));
  if (d_up < 0) d_up = 0;
  if (d_dn < 0) d_dn = 0;
  REAL ls = d_up < d_dn ? d_up : d_dn;
  if (N2 > 0) {
    REAL lN = ws / (REAL)sqrt((double)N2);
    if (lN < ls) ls = lN;
  }
  const REAL Hcol = H2(Hcc, i, j), Jb = A2(F_JB, i, j), Jbe = CATKE.Jbmin;
  const int convecting = (Jb > Jbe) && (N2 < 0), entraining = (Jb > Jbe) && (N2 > 0) && (N2above < 0);
  REAL lconv[4] = {0, 0, 0, 0};
  if (convecting || entraining) {
    const REAL Sp = (REAL)sqrt((double)S2) * ws2 / (Jb + Jbe), esp = (REAL)1 - CATKE.Csp * Sp;
    for (int p = 0; p < 4; p++) {
      REAL l = convecting ? CATKE.Cc[p] * ws3 / (Jb + Jbe) : CATKE.Ce[p] * Jb / (ws * N2 + Jbe);
      l *= esp;
      lconv[p] = l > 0 ? l : 0;
    }
  }
  REAL lpsi[3];
  for (int p = 0; p < 3; p++) {
    REAL l = catke_sigma(m, p, Ri) * ls;
    l = lconv[p] > l ? lconv[p] : l;
    lpsi[p] = l < Hcol ? l : Hcol;
  }
  f.ku = lpsi[0] * ws; f.kc = lpsi[1] * ws; f.ke = lpsi[2] * ws;
  f.convD = lconv[3];
  return f;
}
/* dissipation_length_scale(c,c,c) of the active cell k from its two faces */
static REAL catke_dissipation_length(const model *m, int i, int j, int k, const catke_face *lo, const catke_face *hi) {
  const REAL lh = (lo->convD + hi->convD) / (REAL)2;
  const REAL N2 = (lo->N2 + hi->N2) / (REAL)2, S2 = (lo->S2 + hi->S2) / (REAL)2;
  const REAL Ri = (N2 == 0) ? 0 : N2 / S2;
  REAL d_up = CATKE.Cs * (MK(zf, m->Nz + 1) - MK(zc, k)), d_dn = CATKE.Cb * (MK(zc, k) - catke_zbottom(m, i, j));
  if (d_up < 0) d_up = 0;
  if (d_dn < 0) d_dn = 0;
  REAL ls = d_up < d_dn ? d_up : d_dn;
  if (N2 > 0) {
    REAL lN = (REAL)sqrt((double)catke_tke_floor(m, i, j, k)) / (REAL)sqrt((double)N2);
    if (lN < ls) ls = lN;
  }
  ls = ls / catke_sigma(m, 3, Ri);
  const REAL l = lh > ls ? lh : ls, Hcol = H2(Hcc, i, j);
  return l < Hcol ? l : Hcol;
}
/* top_buoyancy_flux: J^b* = g (alpha J^T - beta J^S) of the instantaneous top flux boundary conditions of T and S (zero
 * without them), alpha and beta at the surface cell */
static REAL catke_top_buoyancy_flux(const model *m, int i, int j) {
  if (!(m->top_flux[2] || m->top_flux[3]) || inactive_cell(m, i, j, m->Nz)) return 0;
  long o = ((long)i - 1 + HH) + (long)m->f[F_T].sx * ((long)j - 1 + HH);
  PREAL a, b;
  teos10_sensitivities((PREAL)A3(F_T, i, j, m->Nz), (PREAL)A3(F_S, i, j, m->Nz), (PREAL)MK(zc, m->Nz), &a, &b);
  PREAL JT = m->top_flux[2] ? (PREAL)m->top_flux[2][o] : 0, JS = m->top_flux[3] ? (PREAL)m->top_flux[3][o] : 0;
  return (REAL)((PREAL)m->g * (a * JT - b * JS) / (PREAL)m->rho0);
}
/* shear_production(c,c,c): Ix of [Iz(nu dz u- dzf dz u+) + Iz(nu dz u+ dzf dz u+)] / (2 dzc) at the x faces, likewise in y;
 * nu = kappa_u of the previous compute averaged to the face's column */
static inline REAL catke_dz_nu_uu(const model *m, int ida, int i, int j, int k) {
  REAL nu = (A3(F_KU, i - 1, j, k) + A3(F_KU, i, j, k)) / (REAL)2;
  return nu * catke_dzu(m, ida, i, j, k) * DZF(k) * catke_dzu(m, F_U, i, j, k);
}
static inline REAL catke_dz_nu_vv(const model *m, int ida, int i, int j, int k) {
  REAL nu = (A3(F_KU, i, j - 1, k) + A3(F_KU, i, j, k)) / (REAL)2;
  return nu * catke_dzv(m, ida, i, j, k) * DZF(k) * catke_dzv(m, F_V, i, j, k);
}
static inline REAL catke_Px(const model *m, int i, int j, int k) {
  REAL n = (catke_dz_nu_uu(m, F_UM, i, j, k) + catke_dz_nu_uu(m, F_UM, i, j, k + 1)) / (REAL)2;
  REAL p = (catke_dz_nu_uu(m, F_U, i, j, k) + catke_dz_nu_uu(m, F_U, i, j, k + 1)) / (REAL)2;
  return (n + p) / ((REAL)2 * DZC(k));
}
static inline REAL catke_Py(const model *m, int i, int j, int k) {
  REAL n = (catke_dz_nu_vv(m, F_VM, i, j, k) + catke_dz_nu_vv(m, F_VM, i, j, k + 1)) / (REAL)2;
  REAL p = (catke_dz_nu_vv(m, F_V, i, j, k) + catke_dz_nu_vv(m, F_V, i, j, k + 1)) / (REAL)2;
  return (n + p) / ((REAL)2 * DZC(k));
}
/* time_step_catke_equation!(model): substep_turbulent_kinetic_energy! + implicit_step!(e) with dt = clock.last_dt */
static void catke_time_step_tke(model *m) {
  const int Nx = m->Nx, Ny = m->Ny, Nz = m->Nz;
  const REAL dt = m->dt, C1 = (REAL)1.5 + m->chi, C2 = (REAL)0.5 + m->chi;
  REAL *enew = (REAL *)malloc(sizeof(REAL) * (size_t)Nx * Ny * Nz);
#pragma omp parallel for schedule(static)
  for (int j = 1; j <= Ny; j++)
    for (int i = 1; i <= Nx; i++) {
      catke_face lo = catke_at_face(m, i, j, 1);
      A3(F_KE, i, j, 1) = 0;
      for (int k = 1; k <= Nz; k++) {
        catke_face hi = catke_at_face(m, i, j, k + 1);
        A3(F_KE, i, j, k + 1) = hi.ke;
        const REAL e = A3(F_E, i, j, k);
        REAL L = 0, en = e;
        if (!inactive_cell(m, i, j, k)) {
          const REAL wb = (-(A3(F_KC, i, j, k) * lo.N2) + -(A3(F_KC, i, j, k + 1) * hi.N2)) / (REAL)2;
          const REAL wbm = wb < 0 ? wb : 0, wbp = wb > 0 ? wb : 0;
          const REAL lD = catke_dissipation_length(m, i, j, k, &lo, &hi);
          const REAL omega = e < 0 ? (REAL)1 / CATKE.tau_neg : (REAL)sqrt(fabs((double)e)) / lD;
          const int on_bottom = inactive_cell(m, i, j, k - 1);
          const REAL ep = e > 0 ? e : 0;
          const REAL divJ = on_bottom ? -(CATKE.CWeps * (REAL)sqrt((double)ep) / DZC(k)) : 0;
          L = (e > CATKE.emin ? wbm / e : 0) - omega + divJ;
          const REAL P = (catke_Px(m, i, j, k) + catke_Px(m, i + 1, j, k)) / (REAL)2 + (catke_Py(m, i, j, k) + catke_Py(m, i, j + 1, k)) / (REAL)2;
          const REAL total = A3(F_GNE, i, j, k) + (P + wbp);
          en = e + dt * (C1 * total - C2 * A3(F_GME, i, j, k));
          A3(F_GME, i, j, k) = total;
        }
        A3(F_LE, i, j, k) = L;
        enew[(i - 1) + (size_t)Nx * ((j - 1) + (size_t)Ny * (k - 1))] = en;
        lo = hi;
      }
    }
  for (int k = 1; k <= Nz; k++)
    for (int j = 1; j <= Ny; j++)
      for (int i = 1; i <= Nx; i++) A3(F_E, i, j, k) = enew[(i - 1) + (size_t)Nx * ((j - 1) + (size_t)Ny * (k - 1))];
  free(enew);
  implicit_step_field_catke(m, F_E, 3, dt);
}
/* compute_average_surface_buoyancy_flux!: the surface buoyancy flux filtered over the convective time scale t* */
static void catke_average_surface_buoyancy_flux(model *m, double dt_since) {
  const int Nz = m->Nz;
#pragma omp parallel for schedule(static)
  for (int j = 1; j <= m->Ny; j++)
    for (int i = 1; i <= m->Nx; i++) {
      if (inactive_cell(m, i, j, Nz)) { A2(F_JB, i, j) = 0; continue; }
      const REAL Jstar = catke_top_buoyancy_flux(m, i, j), J = A2(F_JB, i, j);
      catke_face lo = catke_at_face(m, i, j, Nz), hi = catke_at_face(m, i, j, Nz + 1);
      const REAL lD = catke_dissipation_length(m, i, j, Nz, &lo, &hi);
      REAL Jp = CATKE.Jbmin;
      if (J > Jp) Jp = J;
      if (Jstar > Jp) Jp = Jstar;
      const REAL tstar = (REAL)cbrt((double)(lD * lD / Jp)), eps = (REAL)dt_since / tstar;
      A2(F_JB, i, j) = (J + eps * Jstar) / ((REAL)1 + eps);
    }
  fill_halo_2d(m, F_JB, 0, 0, 1);
}
/* compute_diffusivities!: see the header of this section; then the halos of the diffusivity fields
 * (fill_halo_regions!(model.diffusivity_fields; only_local_halos = true), /root/reference/src/precompile.jl:37,117-119) */
static void catke_compute_diffusivities(model *m) {
  int Nz = m->Nz;
  const double dt_since = m->time - m->catke_prev_time;
  m->catke_prev_time = m->time;
  catke_time_step_tke(m);   /* (clock.last_dt is finite from the model's construction on: src/baroclinic_instability_model.jl:82) */
  if (!m->catke_stale_e_halos) fill_halo_3d(m, F_E, 0, 0, 1);
  memcpy(m->f[F_UM].p, m->f[F_U].p, sizeof(REAL) * (size_t)m->f[F_U].sx * m->f[F_U].sy * m->f[F_U].sz);
  memcpy(m->f[F_VM].p, m->f[F_V].p, sizeof(REAL) * (size_t)m->f[F_V].sx * m->f[F_V].sy * m->f[F_V].sz);
  catke_average_surface_buoyancy_flux(m, dt_since);
#pragma omp parallel for schedule(static)
  for (int j = 1; j <= m->Ny; j++)
    for (int i = 1; i <= m->Nx; i++) {
      A3(F_KU, i, j, 1) = A3(F_KC, i, j, 1) = A3(F_KE, i, j, 1) = 0;
      for (int k = 2; k <= Nz + 1; k++) {
        catke_face f = catke_at_face(m, i, j, k);
        A3(F_KU, i, j, k) = f.ku; A3(F_KC, i, j, k) = f.kc; A3(F_KE, i, j, k) = f.ke;
      }
    }
  /* a14: zero-gradient y layer and periodic x of the face-located diffusivities (no z layer: the boundary faces carry
   * zero), the usual fill for L^e */
  for (int id = F_KU; id <= F_KE; id++) {
    for (int k = 1; k <= Nz + 1; k++)
      for (int i = 1; i <= m->Nx; i++) {
        A3(id, i, 0, k) = A3(id, i, 1, k);
        if (!m->north_fold) A3(id, i, m->Ny + 1, k) = A3(id, i, m->Ny, k);
      }
    if (m->north_fold) fold_rows_levels(m, id, 0, 0, 0, (REAL)1, Nz + 1);   /* the rows beyond the zipper, all Nz+1 faces */
    fill_periodic_x(m, &m->f[id]);
  }
  fill_halo_3d(m, F_LE, 0, 0, 1);
}
/* the "slow" tendency of e that compute_tendencies! leaves in G^n.e: -div(u e) and the top boundary condition of e,
 * Q^e = -C^W_u* u*^3 - C^W_wD w_D^3 (tke_top_boundary_condition.jl), w_D^3 = max(J^b*, 0) dz with the INSTANTANEOUS J^b* */
static void catke_tke_tendency(model *m) {
  int Nz = m->Nz;
  tracer_tendency(m, F_GNE, f_E);
#pragma omp parallel for schedule(static)
  for (int j = 1; j <= m->Ny; j++)
    for (int i = 1; i <= m->Nx; i++) {
      if (!inactive_cell(m, i, j, Nz)) {
        long o = ((long)i - 1 + HH) + (long)m->f[F_U].sx * ((long)j - 1 + HH), ov = ((long)i - 1 + HH) + (long)m->f[F_V].sx * ((long)j - 1 + HH);
        /* u* from the boundary-condition values at (i, j), not interpolated to the cell centre: Oceananigans'
         * friction_velocity reads getbc(velocity_bcs.u, i, j, ...) and getbc(velocity_bcs.v, i, j, ...) */
        REAL Ju = m->top_flux[0] ? m->top_flux[0][o] : 0;
        REAL Jv = m->top_flux[1] ? m->top_flux[1][ov] : 0;
        REAL us2 = (REAL)sqrt((double)(Ju * Ju + Jv * Jv)), us3 = us2 * (REAL)sqrt((double)us2);   /* u*^2, u*^3 */
        REAL Jb = catke_top_buoyancy_flux(m, i, j), wD3 = (Jb > 0 ? Jb : 0) * DZC(Nz);
        REAL Qe = -(CATKE.CWu * us3 + CATKE.CWw * wD3);
        A3(F_GNE, i, j, Nz) -= Qe / DZC(Nz);
      }
    }
}
void FN(set_catke)(void *h, int on) { ((model *)h)->catke = on != 0; }
void FN(set_catke_stale_e_halos)(void *h, int on) { ((model *)h)->catke_stale_e_halos = on != 0; }
void FN(set_substep_order)(void *h, int order) { ((model *)h)->substep_order = order != 0; }
void FN(set_fold_pivot_slaved)(void *h, int on) { ((model *)h)->fold_pivot_slaved = on != 0; }

/* ---------------------------------------------------------------- AB2 + free surface
 * ab2_step!(model, dt) -- /root/reference/src/precompile.jl:39,121-123 (appendix A.4, A.7). */
static void barotropic_mode(model *m, int idU, int idV) {
#pragma omp parallel for schedule(static)
  for (int j = 1; j <= m->Ny; j++)
    for (int i = 1; i <= m->Nx; i++) {
      REAL su = DZC(1) * A3(F_U, i, j, 1), sv = DZC(1) * A3(F_V, i, j, 1);
      for (int k = 2; k <= m->Nz; k++) {
        su += DZC(k) * A3(F_U, i, j, k);
        sv += DZC(k) * A3(F_V, i, j, k);
      }
      A2(idU, i, j) = su;
      A2(idV, i, j) = sv;
    }
}
static inline REAL ab2_G(const model *m, int gn, int gm, int i, int j, int k, REAL chi) {
  REAL C1 = (REAL)1.5 + chi, C2 = (REAL)0.5 + chi;
  REAL not_euler = (C2 != 0) ? (REAL)1 : (REAL)0;
  return C1 * A3(gn, i, j, k) - A3(gm, i, j, k) * C2 * not_euler;
}
static void free_surface_tendency(model *m, REAL chi) {
#pragma omp parallel for schedule(static)
  for (int j = 1; j <= m->Ny; j++)
    for (int i = 1; i <= m->Nx; i++) {
      REAL su = DZC(1) * ab2_G(m, F_GNU, F_GMU, i, j, 1, chi);
      REAL sv = (j == 1) ? 0 : DZC(1) * ab2_G(m, F_GNV, F_GMV, i, j, 1, chi);
      for (int k = 2; k <= m->Nz; k++) {
        su += DZC(k) * ab2_G(m, F_GNU, F_GMU, i, j, k, chi);
        sv += (j == 1) ? 0 : DZC(k) * ab2_G(m, F_GNV, F_GMV, i, j, k, chi);
      }
      A2(F_GBU, i, j) = su;
      A2(F_GBV, i, j) = sv;
    }
  fill_halo_2d(m, F_GBU, 0, 1, -1);
  fill_halo_2d(m, F_GBV, 1, 0, -1);
}
static void ab2_field(model *m, int id, int gn, int gm, REAL dt, REAL chi, int velocity) {
  REAL C1 = (REAL)1.5 + chi, C2 = (REAL)0.5 + chi;
  REAL not_euler = (chi != (REAL)-0.5) ? (REAL)1 : (REAL)0;
  const int nyrows = (id == F_V) ? NYV : m->Ny;
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 1; k <= m->Nz; k++)
    for (int j = 1; j <= nyrows; j++)
      for (int i = 1; i <= m->Nx; i++) {
        if (velocity) {
          REAL G = C1 * A3(gn, i, j, k) - C2 * A3(gm, i, j, k) * not_euler;
          A3(id, i, j, k) += dt * G;
        } else {
          REAL G = C1 * A3(gn, i, j, k) - C2 * A3(gm, i, j, k);
          A3(id, i, j, k) = A3(id, i, j, k) + dt * G;
        }
      }
}
/* The sub-cycle on a folded grid.  The pivot row's cells need V on the y faces beyond them in every substep; as
 * Oceananigans does for its TripolarGrid (the free surface's halo in y is extended to the number of substeps and filled
 * ONCE per step), the state is copied to arrays with Wy = Ns + 1 more rows beyond row Ny, those rows are filled with the
 * images of the rows below it (values, forcing, metrics, depths), and the Ns substeps run on all Ny + Wy rows with no
 * further fill: what the missing neighbour of the last row spoils moves one row per substep and never reaches row Ny. */
static void step_free_surface_fold(model *m, REAL dt) {
  const int Nx = m->Nx, Ny = m->Ny;
  const int Wy = m->Ns + 1;   /* (create refuses grids with fewer than Ns + 3 rows) */
  const int NT = Ny + Wy;
  const REAL dtau = m->dtau_frac * dt;
  const size_t n = (size_t)Nx * (NT + 2);
  REAL *buf = (REAL *)calloc(12 * n, sizeof(REAL));
  REAL *e = buf, *U = e + n, *V = U + n, *GU = V + n, *GV = GU + n, *hf = GV + n, *hc = hf + n, *dyfc = hc + n,
       *dxcf = dyfc + n, *azcc = dxcf + n, *dxfc = azcc + n, *dycf = dxfc + n;
#define TT(a, i, j) a[((long)(i)-1) + (long)Nx * ((long)(j)-1)]
  for (int j = 1; j <= NT + 1; j++)
    for (int i = 1; i <= Nx; i++) {
      int ic = i, ifx = i, jc = j, jf = j;
      REAL su = 1, sv = 1;
      if (j > Ny) {
        ic = fold_i(m, i, 0); ifx = fold_i(m, i, 1);
        jc = 2 * Ny - j; jf = 2 * Ny + 1 - j;
        su = fold_sign(m, i, 1, (REAL)-1); sv = -1;
      }
      TT(e, i, j) = A2(F_ETA, ic, jc);
      TT(U, i, j) = su * A2(F_BU, ifx, jc);  TT(GU, i, j) = su * A2(F_GBU, ifx, jc);
      TT(V, i, j) = sv * A2(F_BV, ic, jf);   TT(GV, i, j) = sv * A2(F_GBV, ic, jf);
      TT(hf, i, j) = H2(Hfc, ifx, jc);       TT(hc, i, j) = H2(Hcf, ic, jf);
      TT(dyfc, i, j) = DYFC(ifx, jc);        TT(dxfc, i, j) = DXFC(ifx, jc);
      TT(dxcf, i, j) = DXCF(ic, jf);         TT(dycf, i, j) = DYCF(ic, jf);
      TT(azcc, i, j) = AZCC(ic, jc);
    }
  for (int id = F_ETAB; id <= F_VB; id++)
    memset(m->f[id].p, 0, sizeof(REAL) * (size_t)m->f[id].sx * m->f[id].sy);
  for (int s = 0; s < m->Ns; s++) {
    const REAL wgt = m->wts[s];
    for (int half = 0; half < 2; half++) {   /* (the order of the two halves: see step_free_surface) */
      if ((half == 0) == (m->substep_order == 0)) {
#pragma omp parallel for schedule(static)
        for (int j = 1; j <= NT; j++)
          for (int i = 1; i <= Nx; i++) {
            const int ip = (i == Nx) ? 1 : i + 1;
            const REAL dxU = TT(dyfc, ip, j) * TT(U, ip, j) - TT(dyfc, i, j) * TT(U, i, j);
            const REAL dyV = (j == 1) ? TT(dxcf, i, 2) * TT(V, i, 2) : TT(dxcf, i, j + 1) * TT(V, i, j + 1) - TT(dxcf, i, j) * TT(V, i, j);
            TT(e, i, j) -= dtau * (dxU + dyV) / TT(azcc, i, j);
          }
      } else {
#pragma omp parallel for schedule(static)
        for (int j = 1; j <= NT; j++)
          for (int i = 1; i <= Nx; i++) {
            const int im = (i == 1) ? Nx : i - 1;
            const REAL dxe = (TT(e, i, j) - TT(e, im, j)) / TT(dxfc, i, j);
            const REAL dye = (j == 1) ? 0 : (TT(e, i, j) - TT(e, i, j - 1)) / TT(dycf, i, j);
            TT(U, i, j) = TT(U, i, j) + dtau * (-m->g * TT(hf, i, j) * dxe + TT(GU, i, j));
            TT(V, i, j) = TT(V, i, j) + dtau * (-m->g * TT(hc, i, j) * dye + TT(GV, i, j));
          }
      }
    }
    for (int j = 1; j <= Ny; j++)
      for (int i = 1; i <= Nx; i++) {
        A2(F_ETAB, i, j) += wgt * TT(e, i, j);
        A2(F_UB, i, j) += wgt * TT(U, i, j);
        A2(F_VB, i, j) += wgt * TT(V, i, j);
      }
  }
  for (int j = 1; j <= Ny; j++)
    for (int i = 1; i <= Nx; i++) {
      A2(F_ETA, i, j) = A2(F_ETAB, i, j);
      A2(F_BU, i, j) = A2(F_UB, i, j);
      A2(F_BV, i, j) = A2(F_VB, i, j);
    }
#undef TT
  free(buf);
}
static void step_free_surface(model *m, REAL dt) {
  if (m->north_fold) {
    step_free_surface_fold(m, dt);
    return;
  }
  int Nx = m->Nx, Ny = m->Ny;
  REAL dtau = m->dtau_frac * dt;
  for (int id = F_ETAB; id <= F_VB; id++)
    memset(m->f[id].p, 0, sizeof(REAL) * (size_t)m->f[id].sx * m->f[id].sy);
  for (int s = 0; s < m->Ns; s++) {
    REAL wgt = m->wts[s];
    /* the two halves of a forward-backward substep; which comes first is an option (the order changed between
     * upstream releases): 0 = eta from the old U, V, then U, V from the new eta; 1 = U, V from the old eta, then eta from them */
    for (int half = 0; half < 2; half++) {
      if ((half == 0) == (m->substep_order == 0)) {
#pragma omp parallel for schedule(static)
        for (int j = 1; j <= Ny; j++)
          for (int i = 1; i <= Nx; i++) {
            int ip = (i == Nx) ? 1 : i + 1;
            REAL dxU = DYFC(ip, j) * A2(F_BU, ip, j) - DYFC(i, j) * A2(F_BU, i, j);
            REAL dyV = (j == Ny) ? -(DXCF(i, j) * A2(F_BV, i, j))
                     : (j == 1)  ? DXCF(i, 2) * A2(F_BV, i, 2)
                                 : DXCF(i, j + 1) * A2(F_BV, i, j + 1) - DXCF(i, j) * A2(F_BV, i, j);
            A2(F_ETA, i, j) -= dtau * (dxU + dyV) / AZCC(i, j);
          }
      } else {
#pragma omp parallel for schedule(static)
        for (int j = 1; j <= Ny; j++)
          for (int i = 1; i <= Nx; i++) {
            int im = (i == 1) ? Nx : i - 1;
            REAL dxe = (A2(F_ETA, i, j) - A2(F_ETA, im, j)) / DXFC(i, j);
            REAL dye = (j == 1) ? 0 : (A2(F_ETA, i, j) - A2(F_ETA, i, j - 1)) / DYCF(i, j);
            /* static column depth at the face: min of the two columns (0 next to land: no pressure force, and G.U is 0) */
            A2(F_BU, i, j) = A2(F_BU, i, j) + dtau * (-m->g * H2(Hfc, i, j) * dxe + A2(F_GBU, i, j));
            A2(F_BV, i, j) = A2(F_BV, i, j) + dtau * (-m->g * H2(Hcf, i, j) * dye + A2(F_GBV, i, j));
          }
      }
    }
    for (int j = 1; j <= Ny; j++)
      for (int i = 1; i <= Nx; i++) {
        A2(F_ETAB, i, j) += wgt * A2(F_ETA, i, j);
        A2(F_UB, i, j) += wgt * A2(F_BU, i, j);
        A2(F_VB, i, j) += wgt * A2(F_BV, i, j);
      }
  }
  for (int j = 1; j <= Ny; j++)
    for (int i = 1; i <= Nx; i++) {
      A2(F_ETA, i, j) = A2(F_ETAB, i, j);
      A2(F_BU, i, j) = A2(F_UB, i, j);
      A2(F_BV, i, j) = A2(F_VB, i, j);
    }
}
void FN(ab2_step)(void *h, double dt_, int euler) {
  model *m = (model *)h;
  REAL dt = (REAL)dt_;
  REAL chi = euler ? (REAL)-0.5 : m->chi;
  free_surface_tendency(m, chi);
  ab2_field(m, F_U, F_GNU, F_GMU, dt, chi, 1);
  ab2_field(m, F_V, F_GNV, F_GMV, dt, chi, 1);
  implicit_step_field(m, F_U, 0, m->nu, dt);      /* ab2_step_velocities!: explicit update, then implicit_step! */
  implicit_step_field(m, F_V, 1, m->nu, dt);
  ab2_field(m, F_T, F_GNT, F_GMT, dt, chi, 0);
  ab2_field(m, F_S, F_GNS, F_GMS, dt, chi, 0);
  implicit_step_field(m, F_T, 2, m->kappa, dt);
  implicit_step_field(m, F_S, 2, m->kappa, dt);
  if (m->catke) {   /* (the diffusivity fields are those of the last update_state!; e is skipped: stepped inside compute_diffusivities!) */
    implicit_step_field_catke(m, F_U, 0, dt);
    implicit_step_field_catke(m, F_V, 1, dt);
    implicit_step_field_catke(m, F_T, 2, dt);
    implicit_step_field_catke(m, F_S, 2, dt);
  }
  step_free_surface(m, dt);
}
/* correct_velocities_and_cache_previous_tendencies!(model, dt) --
 * /root/reference/src/precompile.jl:41,125-127 (appendix A.8). */
void FN(correct_and_cache)(void *h) {
  model *m = (model *)h;
  barotropic_mode(m, F_UB, F_VB);
#pragma omp parallel for collapse(2) schedule(static)
  for (int k = 1; k <= m->Nz; k++)
    for (int j = 1; j <= m->Ny; j++)
      for (int i = 1; i <= m->Nx; i++) {
        /* (peripheral faces keep their zero: what upstream gets from mask_immersed_model_fields! right after) */
        if (!immersed_peripheral_u(m, i, j, k))
          A3(F_U, i, j, k) = A3(F_U, i, j, k) + (A2(F_BU, i, j) - A2(F_UB, i, j)) / H2(Hfc, i, j);
        if (!immersed_peripheral_v(m, i, j, k))
          A3(F_V, i, j, k) = A3(F_V, i, j, k) + (A2(F_BV, i, j) - A2(F_VB, i, j)) / (j == 1 ? m->Lz : H2(Hcf, i, j));
      }
  for (int q = 0; q < 4; q++)
    for (int k = 1; k <= m->Nz; k++)
      for (int j = 1; j <= (q == 1 ? NYV : m->Ny); j++)
        for (int i = 1; i <= m->Nx; i++) A3(F_GMU + q, i, j, k) = A3(F_GNU + q, i, j, k);
  /* (closure = CATKE: G^-.e is written by the e step itself -- cache_previous_tendencies! skips e) */
}
/* initialize!(model): barotropic velocities from the 3-D velocities + their halos */
void FN(initialize)(void *h) {
  model *m = (model *)h;
  barotropic_mode(m, F_BU, F_BV);
  fill_halo_2d(m, F_BU, 0, 1, -1);
  fill_halo_2d(m, F_BV, 1, 0, -1);
  fill_halo_2d(m, F_ETA, 0, 0, 1);
}
/* ---------------------------------------------------------------- data-free forcing
 * /root/reference/src/data_free_ocean_climate_model.jl:12-70: an analytic PrescribedAtmosphere, Radiation and
 * SimilarityTheoryFluxes(solver_stop_criteria = FixedIterations(5)) inside ClimaOcean's OceanSeaIceModel.  ClimaOcean is not
 * in /root/reference; restated [UPSTREAM-UNVERIFIED] as Monin-Obukhov similarity theory with the published COARE 3.5
 * ingredients (Edson et al. 2013, "On the exchange of momentum over the open ocean"; Fairall et al. 2003), in the structure
 * of ClimaOcean's solver: characteristic scales (u*, theta*, q*) iterated a fixed number of times from the differences
 * between the atmospheric state at h = 10 m and the ocean surface.
 *   surface humidity   q_s = 0.98 q_sat(T_s, p_a), Clausius-Clapeyron with constant heat capacities (CliMA Thermodynamics form)
 *   buoyancy scale     b* = g / T_v (theta* (1 + 0.608 q_a) + 0.608 T_a q*),   zeta = kappa h b* / u*^2  (clipped to +-50)
 *   gustiness          U_g = max(0.2, beta (max(-u* b*, 0) z_i)^(1/3)), beta = 1.2, z_i = 600 m;  dU = sqrt(du^2 + dv^2 + U_g^2)
 *   roughness          l_u = 0.011 u*^2 / g + 0.11 nu / u*;  l_q = l_theta = min(1.6e-4, 5.8e-5 / (l_u u* / nu)^0.72)
 *   transfer           u* = kappa dU / (ln(h / l_u) - psi_u(zeta) + psi_u(zeta l_u / h)), theta*, q* alike with psi_q, l_q
 *   stability          psi_u, psi_q of COARE 3.5 (Kansas / convective blend for zeta < 0, Beljaars-Holtslag for zeta >= 0)
 * Fluxes, positive UPWARD (Oceananigans' top flux convention), at the cell centres:
 *   tau = rho_a u*^2 (du, dv) / dU into the ocean;  Q_c = -rho_a c_p u* theta*;  Q_v = -rho_a L_v u* q*;  E = -rho_a u* q*
 *   Q = Q_c + Q_v + eps (sigma T_s^4 - Q_lw) - (1 - albedo) Q_sw,  eps = 0.97, albedo = 0.05
 *   J^T = Q / (rho0 c_p^ocean), J^S = -S E / rho_fw at (c,c);  J^u = -Ix(tau_x) / rho0 at (f,c), J^v = -Iy(tau_y) / rho0 at (c,f)
 * Everything in double, whatever the float type of the model (a 2-D computation); land columns carry no flux.
 * compute_atmosphere_ocean_fluxes runs after every time step of a coupled model (OceanSeaIceModel's time_step!: ocean step,
 * then update_state! of the coupled model), so the tendencies of a step see the fluxes of the state before it. */
typedef struct { double taux, tauy, JT, JS; } ao_flux;
static double ao_psi_u(double z) {
  if (z < 0) {
    double x = sqrt(sqrt(1 - 15 * z)), pk = 2 * log((1 + x) / 2) + log((1 + x * x) / 2) - 2 * atan(x) + 2 * atan(1.0);
    double y = cbrt(1 - 10.15 * z), pc = 1.5 * log((1 + y + y * y) / 3) - sqrt(3.0) * atan((1 + 2 * y) / sqrt(3.0)) + 4 * atan(1.0) / sqrt(3.0);
    double f = z * z / (1 + z * z);
    return (1 - f) * pk + f * pc;
  }
  double dz = 0.35 * z < 50 ? 0.35 * z : 50;
  return -(0.7 * z + 0.75 * (z - 5 / 0.35) * exp(-dz) + 0.75 * 5 / 0.35);
}
static double ao_psi_q(double z) {
  if (z < 0) {
    double x = sqrt(1 - 15 * z), pk = 2 * log((1 + x) / 2);
    double y = cbrt(1 - 34.15 * z), pc = 1.5 * log((1 + y + y * y) / 3) - sqrt(3.0) * atan((1 + 2 * y) / sqrt(3.0)) + 4 * atan(1.0) / sqrt(3.0);
    double f = z * z / (1 + z * z);
    return (1 - f) * pk + f * pc;
  }
  double dz = 0.35 * z < 50 ? 0.35 * z : 50;
  return -(pow(1 + 2.0 / 3.0 * z, 1.5) + 2.0 / 3.0 * (z - 14.28) * exp(-dz) + 8.525);
}
static ao_flux ao_similarity_fluxes(double ua, double va, double Ta, double qa, double pa, double Qsw, double Qlw, double uo,
                                    double vo, double To, double So, double grav, double rho0, int iterations) {
  const double kap = 0.4, Rd = 287.0, Rv = 461.5, cpd = 1005.0, cpv = 1859.0, cpl = 4181.0, Lv0 = 2.5008e6, T0 = 273.16,
               ptr = 611.657, h = 10.0, zi = 600.0, beta = 1.2, charnock = 0.011, nu = 1.5e-5, emis = 0.97, albedo = 0.05,
               sigma = 5.670374419e-8, cpo = 3991.86795711963, rho_fw = 1000.0;
  const double Ts = To + 273.15, eps = Rd / Rv;
  const double psat = ptr * pow(Ts / T0, (cpv - cpl) / Rv) * exp((Lv0 - (cpv - cpl) * T0) / Rv * (1 / T0 - 1 / Ts));
  const double qs = 0.98 * eps * psat / (pa - (1 - eps) * psat);
  const double rho_a = pa / ((Rd * (1 - qa) + Rv * qa) * Ta), cpm = cpd * (1 - qa) + cpv * qa, Lv = Lv0 + (cpv - cpl) * (Ts - T0);
  const double du = ua - uo, dv = va - vo, dth = Ta + grav / cpm * h - Ts, dq = qa - qs, Tv = Ta * (1 + 0.608 * qa);
  double U = sqrt(du * du + dv * dv + 0.2 * 0.2), chi0 = log(h / 1e-4);
  double us = kap * U / chi0, ths = kap * dth / chi0, qst = kap * dq / chi0;
  for (int it = 0; it < iterations; it++) {
    const double bs = grav / Tv * (ths * (1 + 0.608 * qa) + 0.608 * Ta * qst), Jb = -us * bs;
    const double Ug = fmax(0.2, beta * cbrt(fmax(Jb, 0.0) * zi));
    U = sqrt(du * du + dv * dv + Ug * Ug);
    const double lu = charnock * us * us / grav + 0.11 * nu / us, lq = fmin(1.6e-4, 5.8e-5 / pow(lu * us / nu, 0.72));
    double zeta = kap * h * bs / (us * us);
    zeta = zeta > 50 ? 50 : (zeta < -50 ? -50 : zeta);
    const double chiu = log(h / lu) - ao_psi_u(zeta) + ao_psi_u(zeta * lu / h);
    const double chiq = log(h / lq) - ao_psi_q(zeta) + ao_psi_q(zeta * lq / h);
    us = kap * U / chiu;
    ths = kap * dth / chiq;
    qst = kap * dq / chiq;
  }
  ao_flux f;
  f.taux = rho_a * us * us * du / U;
  f.tauy = rho_a * us * us * dv / U;
  const double Qc = -rho_a * cpm * us * ths, Qv = -rho_a * Lv * us * qst, E = -rho_a * us * qst;
  const double Q = Qc + Qv + emis * (sigma * Ts * Ts * Ts * Ts - Qlw) - (1 - albedo) * Qsw;
  f.JT = Q / (rho0 * cpo);
  f.JS = -So * E / rho_fw;
  return f;
}
/* the flux solve at one point (tests): in = {u_a, v_a, T_a, q_a, p_a, Q_sw, Q_lw, u_o, v_o, T_o, S_o, g, rho0}; out = {tau_x, tau_y, J^T, J^S} */
void FN(similarity_fluxes_point)(const double *in, int iterations, double *out) {
  ao_flux f = ao_similarity_fluxes(in[0], in[1], in[2], in[3], in[4], in[5], in[6], in[7], in[8], in[9], in[10], in[11], in[12], iterations);
  out[0] = f.taux; out[1] = f.tauy; out[2] = f.JT; out[3] = f.JS;
}
/* one field of the prescribed atmosphere: q = 0 u, 1 v, 2 T, 3 q, 4 p, 5 shortwave, 6 longwave; parent-shaped (sx x sy of a
 * (c,c) field), i fastest; NULL clears it (uncoupled) */
void FN(set_prescribed_atmosphere)(void *h, int q, const double *a) {
  model *m = (model *)h;
  free(m->atm[q]);
  m->atm[q] = NULL;
  if (!a) return;
  size_t n = (size_t)m->f[F_T].sx * m->f[F_T].sy;
  m->atm[q] = (double *)malloc(n * sizeof(double));
  memcpy(m->atm[q], a, n * sizeof(double));
}
static int ao_coupled(const model *m) {
  for (int q = 0; q < 7; q++)
    if (!m->atm[q]) return 0;
  return 1;
}
void FN(compute_atmosphere_ocean_fluxes)(void *h) {
  model *m = (model *)h;
  if (!ao_coupled(m)) return;
  int Nx = m->Nx, Ny = m->Ny, Nz = m->Nz, jtop = Ny;
  const fld *Fc = &m->f[F_T];
  long n2 = (long)Fc->sx * Fc->sy;
  double *tx = (double *)calloc(n2, sizeof(double)), *ty = (double *)calloc(n2, sizeof(double));
  const int gid[4] = {F_GNU, F_GNV, F_GNT, F_GNS};
  for (int q = 0; q < 4; q++)
    if (!m->top_flux[q]) m->top_flux[q] = (REAL *)calloc((size_t)m->f[gid[q]].sx * m->f[gid[q]].sy, sizeof(REAL));
#define C2(i, j) (((long)(i)-1 + HH) + (long)Fc->sx * ((long)(j)-1 + HH))
  /* centres, one halo column to the west and one row to the south (and the row beyond a zipper fold): the x / y averages
   * onto the faces of the interior read them */
#pragma omp parallel for schedule(static)
  for (int j = 0; j <= jtop; j++)
    for (int i = 0; i <= Nx; i++) {
      long o = C2(i, j);
      if (inactive_cell(m, i, j, Nz)) continue;
      double uo = ((double)A3(F_U, i, j, Nz) + (double)A3(F_U, i + 1, j, Nz)) / 2, vo = ((double)A3(F_V, i, j, Nz) + (double)A3(F_V, i, j + 1, Nz)) / 2;
      ao_flux f = ao_similarity_fluxes(m->atm[0][o], m->atm[1][o], m->atm[2][o], m->atm[3][o], m->atm[4][o], m->atm[5][o],
                                       m->atm[6][o], uo, vo, (double)A3(F_T, i, j, Nz), (double)A3(F_S, i, j, Nz), (double)m->g,
                                       (double)m->rho0, 5);
      tx[o] = f.taux;
      ty[o] = f.tauy;
      if (i >= 1 && j >= 1 && j <= Ny) {
        m->top_flux[2][o] = (REAL)f.JT;
        m->top_flux[3][o] = (REAL)f.JS;
      }
    }
  long sxv = m->f[F_V].sx;
  for (int j = 1; j <= jtop; j++)
    for (int i = 1; i <= Nx; i++) {
      if (j <= Ny) m->top_flux[0][C2(i, j)] = (REAL)(-(tx[C2(i - 1, j)] + tx[C2(i, j)]) / 2 / (double)m->rho0);
      m->top_flux[1][((long)i - 1 + HH) + sxv * ((long)j - 1 + HH)] = (REAL)(-(ty[C2(i, j - 1)] + ty[C2(i, j)]) / 2 / (double)m->rho0);
    }
#undef C2
  free(tx);
  free(ty);
}

/* time_step!(model, dt; euler) -- /root/reference/src/timestepping_utils.jl:29-35 */
void FN(time_step_euler)(void *h, int euler) {
  model *m = (model *)h;
  FN(ab2_step)(h, (double)m->dt, euler);
  m->time += (double)m->dt;
  m->iter += 1;
  FN(fill_halos)(h);
  FN(correct_and_cache)(h);
  FN(update_state)(h);
  FN(compute_atmosphere_ocean_fluxes)(h);   /* (a coupled model only) */
}
void FN(time_step)(void *h) { FN(time_step_euler)(h, 0); }
/* first_time_step!(model) -- /root/reference/src/timestepping_utils.jl:21-27 */
void FN(first_time_step)(void *h) {
  FN(initialize)(h);
  FN(update_state)(h);
  if (ao_coupled((model *)h)) {   /* a coupled model updates its state at iteration 0 ("be paranoid"): the fluxes of the
                                   * initial state, then the ocean's update_state! (diffusivities, tendencies) sees them */
    FN(compute_atmosphere_ocean_fluxes)(h);
    FN(update_state)(h);
  }
  FN(time_step_euler)(h, 1);
}
/* loop!(model, Ninner) -- /root/reference/src/timestepping_utils.jl:37-45 */
void FN(loop)(void *h, int n) {
  for (int s = 0; s < n; s++) FN(time_step_euler)(h, 0);
}

/* set_baroclinic_instability!(model) -- /root/reference/src/model_utils.jl:83-87,99-127 */
void FN(set_baroclinic_instability)(void *h) {
  model *m = (model *)h;
  for (int k = 1; k <= m->Nz; k++)
    for (int j = 1; j <= m->Ny; j++)
      for (int i = 1; i <= m->Nx; i++) {
        double phi = m->curv ? (double)M2(phicc2, i, j) : (double)MJ(phic, j), z = (double)MK(zc, k);
        double step = (1.0 - tanh((fabs(phi) - 40.0) / 5.0)) / 2.0;
        A3(F_T, i, j, k) = (REAL)((30.0 + 1e-3 * z) * step);
        A3(F_S, i, j, k) = (REAL)(-5e-3 * z);
      }
  /* set!(model, ...) on an immersed grid masks what it has set */
  if (m->immersed) FN(mask_immersed_fields)(h);
}

/* diagnostics for tests: the individual terms of G_u at (i,j,k) (1-based) */
void FN(debug_gu_terms)(void *h, int i, int j, int k, double *out) {
  const model *m = (const model *)h;
  REAL vhat = ((DXCF(i - 1, j) * A3(F_V, i - 1, j, k) + DXCF(i - 1, j + 1) * A3(F_V, i - 1, j + 1, k)) / (REAL)2 +
               (DXCF(i, j) * A3(F_V, i, j, k) + DXCF(i, j + 1) * A3(F_V, i, j + 1, k)) / (REAL)2) / (REAL)2 / DXFC(i, j);
  REAL uhat = A3(F_U, i, j, k);
  out[0] = vhat;
  out[1] = biased_interp(m, DY, TO_CENTER, i, j, k, vhat > 0, f_zeta, f_uy, f_vx);
  out[2] = sym_interp(m, DX, TO_FACE, i, j, k, f_dyV);
  out[3] = biased_interp(m, DX, TO_FACE, i, j, k, uhat > 0, f_dxU, f_div, NULL);
  for (int t = 0; t < 2; t++) {
    REAL wt = sym_interp(m, DX, TO_FACE, i, j, k + t, f_Azw);
    out[4 + t] = wt * biased_interp(m, DZ, TO_FACE, i, j, k + t, wt > 0, f_u, NULL, NULL);
  }
  out[6] = biased_interp(m, DX, TO_FACE, i, j, k, uhat > 0, f_dxu2, f_usm, NULL);
  out[7] = sym_interp(m, DY, TO_CENTER, i, j, k, f_dxv2);
  out[8] = (A3(F_P, i, j, k) - A3(F_P, i - 1, j, k)) / DXFC(i, j);
}

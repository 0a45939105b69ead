/*
 * gb25.h -- C ABI of libgb25hip.so: the MI355X (gfx950) implementation of the
 * Oceananigans HydrostaticFreeSurfaceModel time-step loop that GB-25's
 * baroclinic_instability_model drives.
 *
 * This is the drop-in boundary.  Each entry point names the reference interface it
 * replaces (paths relative to the GB-25 tree).  A Julia host binds these with
 * `ccall`; this repository binds them with ctypes
 * (gb-25_amd/binding.py).  Plain pointers and sizes only; no exceptions cross the ABI:
 * every call returns a gb25_status and the message is kept per handle.
 *
 * Layout contract (src/correctness.jl:4-15 compares `parent(field)` arrays):
 * every field is stored exactly like `parent(field)` of the Oceananigans Field --
 * column-major, i fastest, halo H on every side:
 *     centre/centre/centre (T, S, pHY', u(*), G.u, G.T, G.S): (Nx+2H, Ny+2H,   Nz+2H)
 *     v, G.v  (face in bounded y):                             (Nx+2H, Ny+2H+1, Nz+2H)
 *     w       (face in bounded z):                             (Nx+2H, Ny+2H,   Nz+2H+1)
 *     eta, U, etabar, Ubar, G.U:                               (Nx+2H, Ny+2H,   1)
 *     V, Vbar, G.V:                                            (Nx+2H, Ny+2H+1, 1)
 * (*) x is periodic, so u has Nx faces.  With an x-slab decomposition Nx is the
 * LOCAL slab width (Nx_global / nranks) and the x halos hold the neighbours' columns.
 * Element type: the library is built once per Oceananigans float type -- libgb25hip.so holds Float32
 * (simulations/baroclinic_instability_simulation_run.jl:13, the headline runs) and libgb25hip_f64.so Float64
 * (the default of --float-type, src/arg_parsing.jl:12-16, used by the correctness scripts).  Both export the
 * same symbols; gb25_real_bytes() tells which one is loaded, and every `void *` data pointer below addresses
 * elements of that size.
 */
#ifndef GB25_H
#define GB25_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gb25_model gb25_model; /* opaque; owns all device memory */

typedef enum {
  GB25_OK = 0,
  GB25_ERR_INVALID_ARGUMENT = 1,
  GB25_ERR_HIP = 2,          /* a HIP runtime call failed; see gb25_last_error_string */
  GB25_ERR_OUT_OF_MEMORY = 3,
  GB25_ERR_NO_DEVICE = 4,    /* no gfx950 device visible: there is NO CPU fallback */
  GB25_ERR_STATE = 5,
  GB25_ERR_COMM = 6          /* RCCL could not be loaded / a communicator or exchange call failed */
} gb25_status;

/* Field identifiers: the set compared by compare_states (src/correctness.jl:28-90)
 * plus the diagnostic pressure and the barotropic work fields. */
typedef enum {
  GB25_U = 0, GB25_V, GB25_W, GB25_T, GB25_S, GB25_PHY,
  GB25_GN_U, GB25_GN_V, GB25_GN_T, GB25_GN_S, /* timestepper.G^n */
  GB25_GM_U, GB25_GM_V, GB25_GM_T, GB25_GM_S, /* timestepper.G^- */
  GB25_ETA, GB25_BT_U, GB25_BT_V,             /* free_surface.eta, barotropic_velocities */
  GB25_ETA_BAR, GB25_U_BAR, GB25_V_BAR,       /* free_surface.filtered_state */
  GB25_GN_BT_U, GB25_GN_BT_V,                 /* timestepper.G^n.U, .V */
  /* closure = CATKEVerticalDiffusivity() only (gb25_set_closure_catke): the TKE tracer, its tendencies, and
   * model.diffusivity_fields as src/correctness.jl:60-67 compares them: kappa_u, kappa_c, kappa_e at (Center, Center,
   * Face) [Nz + 1 levels], L^e at cell centres, the surface buoyancy flux J^b (2-D) */
  GB25_E, GB25_GN_E, GB25_GM_E, GB25_KAPPA_U, GB25_KAPPA_C, GB25_KAPPA_E, GB25_LE, GB25_JB,
  /* ... and diffusivity_fields.previous_velocities (u, v at the previous compute_diffusivities!: CATKE's shear production
   * is centred between them and the current ones); not in the compared set, exposed for state transfer and tests.
   * (A single domain keeps them, between calls, in whichever buffer already holds them -- u, v themselves right after
   * compute_diffusivities!, the look-ahead's partner buffers after the AB2 step that followed -- and brings them into
   * these fields when the host reads or writes any field or asks for a device pointer: gb25_api.hip, prev_uv_src.) */
  GB25_PREV_U, GB25_PREV_V,
  GB25_FIELD_COUNT
} gb25_field;

/* Model configuration = the keyword arguments of
 * baroclinic_instability_model(arch, Nx, Ny, Nz; dt, halo, free_surface, ...)
 * (src/baroclinic_instability_model.jl:17-40) and of simple_latitude_longitude_grid
 * (src/model_utils.jl:56-65).  Fill with gb25_default_config, then override. */
typedef struct {
  int32_t Nx, Ny, Nz;   /* GLOBAL interior size */
  int32_t halo;         /* H; 8 in every GB-25 script */
  int32_t substeps;     /* SplitExplicitFreeSurface(substeps=30) */
  int32_t rank, nranks; /* x-slab decomposition: this process owns columns
                           [rank*Nx/nranks, (rank+1)*Nx/nranks); nranks=1: periodic x is local */
  int32_t device;       /* HIP device ordinal */
  double dt;            /* model.clock.last_dt (src/baroclinic_instability_model.jl:82) */
  double chi;           /* QuasiAdamsBashforth2 chi = 0.1 */
  double lat_south, lat_north; /* (-80, 80) */
  double lon_west, lon_east;   /* (0, 360) */
  double depth, zexp_h;        /* exponential_z_faces(Nz, depth=4000, h=30) */
  double g, Omega, radius, rho0; /* 9.80665, 7.292115e-5, 6371e3, 1020 (TEOS-10 reference) */
  int32_t slab_mode;    /* 0: the x halos are a local periodic copy when nranks == 1 and come from the ring neighbours
                              when nranks > 1;  1: always by exchange (nranks == 1: the slab is its own west and east
                              neighbour -- the self-ring that runs the exchange code path on one GPU) */
  int32_t grid_type;    /* gb25_grid_type: grid_type = :simple_lat_lon | :gaussian_islands
                              (src/baroclinic_instability_model.jl:19,59-65) */
  int32_t ranks_y;      /* Partition(Rx, Ry, 1) (sharding/sharded_baroclinic_instability_simulation_run.jl:65-72): Ry, the
                              ranks along y; 0 or 1 = x slabs only.  nranks = Rx Ry, rank = ry Rx + rx; the rank owns the
                              columns [rx Nx/Rx, (rx+1) Nx/Rx) and the rows [ry Ny/Ry, (ry+1) Ny/Ry) */
} gb25_config;

typedef enum {
  GB25_GRID_LAT_LON = 0,                  /* simple_latitude_longitude_grid (src/model_utils.jl:56-65), flat bottom */
  GB25_GRID_LAT_LON_GAUSSIAN_ISLANDS = 1, /* ImmersedBoundaryGrid(that grid, GridFittedBottom(gaussian_islands);
                                             active_cells_map = false): the two mountains of src/model_utils.jl:67-80 */
  GB25_GRID_LAT_LON_AS_CURVILINEAR = 2,   /* grid 0 stepped by the orthogonal-curvilinear kernels (2-D metrics): a test
                                             vehicle, results equal those of grid 0 to round-off */
  GB25_GRID_TRIPOLAR = 3,                 /* TripolarGrid(arch; size, halo, z) (src/model_utils.jl:134-137), flat bottom:
                                             poles at (70 E, 55 N) and (250 E, 55 N), southern edge lat_south, zipper
                                             fold along the northern edge.  The poles are singular without land.
                                             Decomposed in x like the other grids: the cells beyond a slab's fold line
                                             belong to the mirrored rank nranks-1-rank, an extra point-to-point partner */
  GB25_GRID_TRIPOLAR_GAUSSIAN_ISLANDS = 4, /* grid_type = :gaussian_islands of the reference (src/model_utils.jl:129-146):
                                             the tripolar grid with the two Gaussian mountains over its poles */
  GB25_GRID_COUNT
} gb25_grid_type;

/* Per-model switches (gb25_set_option).  Defaults in brackets.  None of them changes results beyond the last bits
 * (KERNELS) or at all (the rest): they select schedules, and tests use them to prove exactly that. */
typedef enum {
  GB25_OPT_KERNELS = 0,          /* [2] 2: LDS-staged / flux-sharing tendency kernels; 1: direct-stencil kernels (cross-check) */
  GB25_OPT_AB2_LOOKAHEAD,        /* [1] the tendency kernels also write the next time level: 0 off, 1 u,v,T,S, 2 T,S only */
  GB25_OPT_SUBCYCLE_LOOKAHEAD,   /* [1 from 8 M cells and on slabs] the next step's split-explicit sub-cycle runs as soon as its
                                    G.U, G.V exist: 1 = between the momentum and the tracer kernel, 2 = beside the tracer
                                    kernel on a stream of its own (slabs: on the exchange stream), 0 = inside its own step */
  GB25_OPT_SUBCYCLE_BLOCK,       /* [5] substeps per barotropic launch: 1, 3, 5, 7 */
  GB25_OPT_FILL_FUSED,           /* [1] y, z and periodic-x halo fills in one launch */
  GB25_OPT_TWO_STREAMS,          /* [1] tracer branch (AB2, halos, pressure) on a second stream */
  GB25_OPT_STORE_PRESSURE,       /* [0] store pHY' every step (1) or only its differences, pHY' on demand (0) */
  GB25_OPT_SPLIT_TENDENCIES,     /* [1] slab: interior tile columns of the momentum tendencies before the halos arrive */
  GB25_OPT_PRESSURE_PRECISION,   /* [64] THE exception to "results unchanged": 64 = equation of state + hydrostatic integral
                                    in fp64 whatever the model's float type (default; fp32 in, fp32 out); 32 = in the float
                                    type's own arithmetic, operation for operation what an all-Float32 model computes
                                    (the stated Float32 tolerance) */
  GB25_OPT_IMMERSED_KERNELS,     /* [1 iff some cell is immersed] 1: run the immersed-boundary kernel variants anyway */
  GB25_OPT_FOLD_FILLS,           /* [1] single domain: the last writers of u,v / T,S / eta,U,V write the halo cells themselves */
  GB25_OPT_LAZY_CORRECTOR,       /* [1] single flat lat-lon domain, between the steps of one gb25_loop call: the barotropic
                                    correction of u, v is added by the kernels that read them instead of by a sweep over
                                    u and v (same bits; memory holds the corrected velocities when the call returns) */
  GB25_OPT_MOMENTUM_CHUNK_LEVELS, /* [24 on models wide enough for four rounds of blocks with it -- 1440 x 720 x 48 --, else 12] levels a
                                    block of the momentum tendency kernel marches through (>= 6); also the association of the
                                    column integrals of u, v: results change in the last bits (a bit-for-bit comparison of a wide
                                    single domain with its narrow ranks pins it on both sides) */
  GB25_OPT_TRACER_CHUNK_LEVELS,  /* [like MOMENTUM_CHUNK_LEVELS] the same for the tracer tendency kernel (bitwise neutral) */
  GB25_OPT_TRACERS_FIRST,        /* [1] single domain, composite steps: the tracer tendency kernel before the momentum kernel, so
                                    that the next step's pressure (it needs the tracer look-ahead's T, S) can run beside the next
                                    step's sub-cycle; 0: momentum first (src/precompile.jl:48-50 lists them in that order; the
                                    two evaluations are independent of each other) */
  GB25_OPT_W_ON_THE_FLY,         /* [1] with LAZY_CORRECTOR, between the steps of one gb25_loop call: the tendency kernels do not read w;
                                    they carry it up their chunks of levels from the divergence of the transports they hold, starting
                                    from 2-D chunk bases made from the column integrals of the velocity look-ahead.  No k_compute_w
                                    launch and no w traffic in those steps; the field w is recomputed from the velocities when the
                                    call returns.  Like KERNELS this changes results in the LAST BITS (another association of the
                                    vertical sum): comparisons that must be bit-exact (a decomposition against the single domain)
                                    switch it off */
  GB25_OPT_SUB_STREAM_PRIORITY,  /* [0] slab of a decomposition: the substeps of the sub-cycle look-ahead run on a HIGH-PRIORITY stream (their
                                    few blocks could be dispatched ahead of what is left of the tracer kernel's grid -- measured: no difference); 0: an ordinary stream.
                                    Read when the exchange context is built (gb25_comm_init_*) */
  GB25_OPT_SUBCYCLE_WHOLE,       /* [1] slab of a decomposition with fewer sub-cycle tiles than the device has CUs (a 180-column rank): all
                                    substeps in ONE launch, the tile and a ring as wide as the sub-cycle is long in 125 KB of LDS;
                                    0: the blocked launches (SUBCYCLE_BLOCK) */
  GB25_OPT_EARLY_STRIPS,         /* [1] x slab of a decomposition: the bundle is unpacked on the exchange stream right behind its transfer
                                    and the pressure strips next to the x halos follow it there, beside the interior momentum pass;
                                    0: unpack and strips on the main stream when it gets there */
  GB25_OPT_CATKE_STALE_E_HALOS,  /* [0] closure = CATKE, single domain: 1 = the halo cells of e are NOT refilled after e is stepped inside
                                    compute_diffusivities! (Oceananigans as recalled: the tendencies that follow see halos one e step old);
                                    0 = refilled.  A RESTATEMENT choice, not a schedule: it changes results */
  GB25_OPT_COMM_TIMEOUT_SECONDS, /* [180] gb25_comm_init_rccl: bound on ncclCommInitRank and on the first exchange with every peer; past it
                                    the call returns GB25_ERR_COMM naming the rank and the buffer set instead of hanging */
  GB25_OPT_ROCTX_RANGES,         /* [1] roctx ranges named like the reference's profiler annotations ("first_time_step", "time_step", "loop":
                                    src/timestepping_utils.jl:22,30,38) around the composites, and one per phase of src/precompile.jl:31-42
                                    around the issue of its kernels (rocprofv3 --marker-trace); free when no marker library is loaded */
  /* Two more RESTATEMENT choices (like CATKE_STALE_E_HALOS they change results; a Julia dump of tools/dump_goldens.jl decides them
   * without new kernel work; the oracle has both on every grid): */
  GB25_OPT_SUBSTEP_ORDER,        /* [0] the two halves of a split-explicit substep: 0 = eta from the old U, V, then U, V from the new eta;
                                    1 = U, V from the old eta, then eta from the new U, V (the order changed between
                                    upstream releases).  1: LatitudeLongitudeGrid only, one substep per launch */
  GB25_OPT_FOLD_PIVOT_SLAVED,    /* [0] TripolarGrid, single domain: 1 = every fold fill also overwrites the eastern half of the pivot row
                                    (cell centres of the last row, held twice) with the image of its western half, as a later upstream
                                    fix does as recalled; 0 = both copies are stepped independently */
  GB25_OPT_COUNT
} gb25_option;

/* Metric identifiers for gb25_get_metric (diagnostics / tests). */
typedef enum {
  GB25_M_PHIF = 0, GB25_M_PHIC, GB25_M_DXC, GB25_M_DXF, GB25_M_AZC, GB25_M_AZF, GB25_M_FCOR,
  GB25_M_ZF, GB25_M_ZC, GB25_M_DZC, GB25_M_DZF
} gb25_metric;
/* Horizontal metrics of an orthogonal curvilinear grid by location, for gb25_get_metric2 (Oceananigans' names:
 * GB25_M2_DXFC = dx at (Face, Center), ...; FFF = Coriolis parameter at (f,f); PHICC = latitude of the cell centres). */
typedef enum {
  GB25_M2_DXFC = 0, GB25_M2_DXCC, GB25_M2_DXCF, GB25_M2_DXFF, GB25_M2_DYFC, GB25_M2_DYCC, GB25_M2_DYCF, GB25_M2_DYFF,
  GB25_M2_AZCC, GB25_M2_AZFC, GB25_M2_AZCF, GB25_M2_AZFF, GB25_M2_FFF, GB25_M2_PHICC, GB25_M2_COUNT
} gb25_metric2;

/* Kernel identifiers for the built-in HIP-event timers (gb25_profile_*). */
typedef enum {
  GB25_K_FILL_HALOS = 0, GB25_K_COMPUTE_W, GB25_K_COMPUTE_P, GB25_K_GU, GB25_K_GV, GB25_K_TRACERS,
  GB25_K_AB2_VELOCITIES, GB25_K_AB2_TRACERS, GB25_K_BAROTROPIC, GB25_K_CORRECTOR,
  GB25_K_IMPLICIT,     /* implicit_step!: the vertical solves of a closure (all of a step's launches together)        */
  GB25_K_CLOSURE,      /* CATKE: advection of e, surface flux, diffusivities                                          */
  GB25_K_FLUXES,       /* data-free forcing: similarity-theory fluxes; the bottom drag's flux kernel                   */
  GB25_K_COUNT
} gb25_kernel;

void gb25_default_config(gb25_config *cfg, int32_t Nx, int32_t Ny, int32_t Nz);

/* ---- lifecycle: replaces baroclinic_instability_model(arch, Nx, Ny, Nz; dt, ...)
 *      (src/baroclinic_instability_model.jl:17-85): builds the grid metrics, allocates every
 *      field zeroed on the device and stores dt in the clock. */
gb25_status gb25_create(const gb25_config *cfg, gb25_model **out);
void gb25_destroy(gb25_model *m);
const char *gb25_last_error_string(const gb25_model *m); /* valid until the next call on m */
const char *gb25_version(void);
int32_t gb25_real_bytes(void); /* sizeof one field element of THIS library: 4 (Float32) or 8 (Float64) */
/* sizeof(gb25_config) and sizeof(gb25_catke_parameters) as THIS library was built: a binding in another language checks its
 * mirror of the structs against them before the first gb25_create (a field added here must not shift silently there) */
int32_t gb25_config_bytes(void);
int32_t gb25_catke_parameters_bytes(void);

/* Run all kernels of this model on the caller's HIP stream (a hipStream_t passed as void*; NULL is HIP's
 * default stream, which is what torch.cuda.current_stream() is unless the host changed it).  Lets a host
 * framework order our kernels with its own work.  gb25_use_own_stream goes back to the model's private
 * non-blocking stream (the state after gb25_create). */
gb25_status gb25_set_option(gb25_model *m, gb25_option opt, int32_t value);
gb25_status gb25_get_option(const gb25_model *m, gb25_option opt, int32_t *value);
gb25_status gb25_set_stream(gb25_model *m, void *hip_stream);
gb25_status gb25_use_own_stream(gb25_model *m);
gb25_status gb25_synchronize(gb25_model *m);

/* ---- fields: replaces parent(field)/interior(field)/set!(model, ...) and sync_states!
 *      (src/correctness.jl:92-103).  dims = parent dims (include_halos != 0) or interior dims.
 *      Host pointers are borrowed for the duration of the call. */
gb25_status gb25_field_dims(const gb25_model *m, gb25_field f, int include_halos, int32_t dims[3]);
gb25_status gb25_set_field(gb25_model *m, gb25_field f, const void *host, int include_halos);
gb25_status gb25_get_field(gb25_model *m, gb25_field f, void *host, int include_halos);
/* Device pointer of parent(field) for zero-copy wrapping (e.g. unsafe_wrap(ROCArray, ...)).
 * G^n / G^- pointers are exchanged by correct_and_cache (a pointer swap replaces the copy): ask again after a step.
 * u, v, T and S normally alternate between two buffers as well (the tendency kernels write the next time level
 * ahead of ab2_step!); asking for the pointer of a prognostic 3-D field or of a tendency pins them to the buffers
 * handed out and turns that look-ahead off for this model, because writes through the pointer cannot be seen by
 * the library.
 * GB25_PHY is a diagnostic: inside the composite steps only its horizontal differences (what the momentum tendencies
 * use) are stored, and gb25_get_field(GB25_PHY) recomputes the field from T and S on demand; asking for its device
 * pointer makes every later step store it (GB25_OPT_STORE_PRESSURE = 1 does the same from the start). */
gb25_status gb25_field_device_ptr(gb25_model *m, gb25_field f, void **dev);
gb25_status gb25_get_metric(const gb25_model *m, gb25_metric id, int32_t logical_index, double *value);
/* grid_type >= 2: one horizontal metric as fp64, the parent layout of a (Center, Face) 2-D field:
 * (Nx + 2 halo) x (Ny + 2 halo + 1) values, i fastest (the role of grid.Δxᶠᶜᵃ etc. of an OrthogonalSphericalShellGrid). */
gb25_status gb25_get_metric2(const gb25_model *m, gb25_metric2 id, double *values, int64_t count);
gb25_status gb25_get_substepping(const gb25_model *m, int32_t *n_effective, double *dtau_fraction,
                                 double *weights /* >= substeps entries */);

/* ---- the HOST's grid.  In the reference the grid is built by Oceananigans on the Julia side and handed to the model:
 *      TripolarGrid(arch; size, halo, z) wrapped in ImmersedBoundaryGrid(grid, GridFittedBottom(gaussian_islands)) with
 *      z = exponential_z_faces(Nz, depth) (src/model_utils.jl:56-62,129-146).  gb25_create builds stand-ins from cfg.grid_type
 *      (an analytic bipolar cap, not necessarily Oceananigans' coordinate lines); a host that has the real grid passes it:
 *
 *      gb25_set_curvilinear_grid: the 14 horizontal metrics in gb25_metric2 order -- grid.Δxᶠᶜᵃ, Δxᶜᶜᵃ, Δxᶜᶠᵃ, Δxᶠᶠᵃ, Δyᶠᶜᵃ, Δyᶜᶜᵃ,
 *        Δyᶜᶠᵃ, Δyᶠᶠᵃ, Azᶜᶜᵃ, Azᶠᶜᵃ, Azᶜᶠᵃ, Azᶠᶠᵃ, the Coriolis parameter at (f,f) and the latitude of the cell centres φᶜᶜᵃ [degrees]
 *        -- each the PARENT array over the GLOBAL grid as fp64, (nx, ny) = (Nx_global + 2 halo, Ny + 2 halo [+ 1]), i fastest
 *        (ny = Ny + 2 halo is what a (Periodic, RightConnected, Bounded) grid holds; a Bounded y has one more row of y faces).
 *        x halo columns are taken as the host holds them; on a folded grid the rows beyond the pivot row are taken from the
 *        interior by the fold's own rule (cell rows and y-face rows mirror about the centres of the last row of cells, x
 *        faces as i -> Nx - i + 2), whatever the host's halo rows hold.  The model must have been created with a curvilinear
 *        grid_type (2, 3, 4).  A slab of a decomposition takes its columns -- halo, widened and fold-partner columns
 *        included -- from the same global arrays.
 *      gb25_set_vertical_faces: the Nz + 1 faces of grid.z, bottom to top [m]; spacings, the TEOS-10 level tables, the bottom's
 *        level tables and a closure's elimination tables are rebuilt.
 *      gb25_set_bottom_height: GridFittedBottom(bottom_height): Nx_global x Ny doubles at the GLOBAL cell centres, i fastest
 *        [m, negative down]; replaces the bottom of grid_type, masks the fields.
 *      All three are collective on a decomposed model, void every look-ahead and may be called in any order before (or
 *      between) steps.  gb25_get_bottom_info (diagnostic; 0-based local i, j): which = 0 the number of immersed cells of the
 *      column, 1 / 2 the static column depth at its U / V face. */
gb25_status gb25_set_curvilinear_grid(gb25_model *m, const double *const *metrics /* [GB25_M2_COUNT] */, int32_t nx, int32_t ny);
gb25_status gb25_set_vertical_faces(gb25_model *m, const double *z_faces, int32_t n /* Nz + 1 */);
gb25_status gb25_set_bottom_height(gb25_model *m, const double *bottom_height);
gb25_status gb25_get_bottom_info(const gb25_model *m, int32_t which, int32_t i, int32_t j, double *value);

/* ---- flux boundary conditions: what compute_hydrostatic_boundary_tendency_contributions! (src/precompile.jl:25,52-61)
 *      adds to the tendencies.  f = GB25_U | GB25_V | GB25_T | GB25_S; `flux` holds J at the interior points of the field's
 *      horizontal location (gb25_field_dims(f, 0): dims[0] x dims[1] values, i fastest), positive upward; NULL restores
 *      the default no-flux condition.  The top cell's tendency gets -J/dz inside the tendency kernels. */
gb25_status gb25_set_top_flux(gb25_model *m, gb25_field f, const void *flux);
/* the flux as the device holds it (same shape; what the coupled model last computed, or what the host set) */
gb25_status gb25_get_top_flux(gb25_model *m, gb25_field f, void *flux);

/* ---- quadratic bottom drag: what ClimaOcean's ocean_simulation (src/data_free_ocean_climate_model.jl:26) puts at the bottom
 *      of u and v -- also the immersed bottom -- with bottom_drag_coefficient = 0.003: the flux boundary condition
 *      J = -Cd u sqrt(u^2 + Ixy(v)^2) (likewise for v), evaluated before every tendency evaluation and added to the tendency of
 *      the face's first free level.  Cd = 0 (the default; baroclinic_instability_model has none): no drag.  Collective. */
gb25_status gb25_set_bottom_drag(gb25_model *m, double Cd);
/*      tracer_advection of the same ocean_simulation: WENO(order = 7) (baroclinic_instability_model: WENO(order = 5), the
 *      default here).  T, S and CATKE's e; the order-5 path where the eight-point stencil meets a wall or the immersed
 *      boundary.  Collective. */
gb25_status gb25_set_tracer_advection_order(gb25_model *m, int32_t order);
gb25_status gb25_get_tracer_advection_order(const gb25_model *m, int32_t *order);
gb25_status gb25_get_bottom_drag(const gb25_model *m, double *Cd);

/* ---- data-free forcing (src/data_free_ocean_climate_model.jl:12-70): a PrescribedAtmosphere + Radiation +
 *      SimilarityTheoryFluxes(solver_stop_criteria = FixedIterations(5)) coupled to the ocean as ClimaOcean's OceanSeaIceModel
 *      does.  The host evaluates the atmosphere at the ocean's cell centres (the reference: analytic fields on a 360 x 180
 *      grid, interpolated) and hands each field over as DOUBLES in the parent layout of a 2-D (c,c) field, halo cells
 *      included ((Nx_local + 2 halo) x (Ny + 2 halo), i fastest): the fluxes of the first halo column / row are computed
 *      from them, not exchanged.  Once all seven fields are set the model is coupled: gb25_first_time_step computes the
 *      fluxes of the initial state, every step ends with the flux computation (similarity theory per surface cell, fp64)
 *      and the results are the top flux boundary conditions of u, v, T, S.  NULL clears a field (uncoupled).
 *      Collective on a decomposed model. */
typedef enum {
  GB25_ATM_U = 0,      /* zonal wind at 10 m [m/s]                      (zonal_wind, :1)        */
  GB25_ATM_V,          /* meridional wind [m/s]                                                 */
  GB25_ATM_T,          /* air temperature [K]                           (Tatm + 273.15, :3,6)   */
  GB25_ATM_Q,          /* specific humidity [kg/kg]                     (0, :57)                */
  GB25_ATM_P,          /* surface pressure [Pa]                                                 */
  GB25_ATM_SHORTWAVE,  /* downwelling shortwave radiation [W/m2]        (sunlight, :2)          */
  GB25_ATM_LONGWAVE,   /* downwelling longwave radiation [W/m2]                                 */
  GB25_ATM_COUNT
} gb25_atmosphere_field;
gb25_status gb25_set_prescribed_atmosphere(gb25_model *m, gb25_atmosphere_field f, const double *values);
/* compute_atmosphere_ocean_fluxes! + the net fluxes, on demand (the composites call it themselves) */
gb25_status gb25_compute_atmosphere_ocean_fluxes(gb25_model *m);

/* ---- closure: `closure = nothing` (the default, src/baroclinic_instability_model.jl:29) or
 *      VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), κ = kappa, ν = nu) (:31): after the explicit AB2
 *      update of u, v (nu) and T, S (kappa), ab2_step! solves (1 - Δt ∂z K ∂z) φ = φ* per column (implicit_step!, batched
 *      tridiagonal solver).  nu = kappa = 0 restores closure = nothing.  [m²/s] */
gb25_status gb25_set_vertical_diffusivity(gb25_model *m, double nu, double kappa);
/*      closure = Oceananigans.TurbulenceClosures.CATKEVerticalDiffusivity() (src/baroclinic_instability_model.jl:30,
 *      sharding/less_simple_sharding_problem.jl:84-93): tracers become (T, S, e); update_state! computes the diffusivity
 *      fields (and fills their halos, src/precompile.jl:37); ab2_step! mixes u, v, T, S, e implicitly with them.  Every grid
 *      type, single domain and slabs (collective there).  on = 0: back to closure = nothing. */
gb25_status gb25_set_closure_catke(gb25_model *m, int32_t on);
/*      The closure's parameters (psi = u, c, e, D in the arrays): the defaults are those of CATKEVerticalDiffusivity();
 *      ClimaOcean's ocean_simulation (src/data_free_ocean_climate_model.jl:26) uses its default_ocean_closure(), which differs
 *      in C^b.  Collective on a decomposed model. */
typedef struct {
  double Cs, Cb, Csp;              /* mixing length: distance to the surface / to the bottom, shear-plume parameter */
  double CRid, CRi0;               /* stability function: width and centre of the step in Ri */
  double Chi[4], Clo[4], Cun[4];   /* stability function: strongly stable, weakly stable, unstable */
  double Cc[4], Ce[4];             /* convective and entrainment lengths */
  double CWu, CWw;                 /* surface TKE flux: friction velocity and convective velocity terms */
  double minimum_tke, minimum_convective_buoyancy_flux, negative_tke_damping_time_scale;
  double CWeps;                    /* bottom TKE flux J^e = -CWeps e^(3/2), implicit in the bottom cell (CATKEEquation's C^W_epsilon = 1) */
} gb25_catke_parameters;
void gb25_default_catke_parameters(gb25_catke_parameters *p);
gb25_status gb25_set_catke_parameters(gb25_model *m, const gb25_catke_parameters *p);
gb25_status gb25_get_catke_parameters(const gb25_model *m, gb25_catke_parameters *p);
gb25_status gb25_get_vertical_diffusivity(const gb25_model *m, double *nu, double *kappa);

/* ---- initial conditions: set_baroclinic_instability!(model) (src/model_utils.jl:99-127) */
gb25_status gb25_set_baroclinic_instability(gb25_model *m);

/* ---- clock: model.clock fields (src/model_utils.jl:150-155) */
gb25_status gb25_get_clock(const gb25_model *m, double *time, int64_t *iteration, double *last_dt);
gb25_status gb25_set_dt(gb25_model *m, double dt);

/* ---- the phases of one time step, in the reference's order (src/precompile.jl:31-42).
 *      Each replaces the *_workload! wrapper cited. */
gb25_status gb25_initialize(gb25_model *m);              /* Oceananigans.initialize!(model) (correctness/..._run.jl:50-51) */
gb25_status gb25_mask_immersed_fields(gb25_model *m);    /* src/precompile.jl:21,34 mask_immersed_model_fields! (nothing to do on a flat bottom) */
gb25_status gb25_fill_halo_regions(gb25_model *m);       /* src/precompile.jl:35,40,44-46 tupled_fill_halo_regions_workload! */
gb25_status gb25_compute_auxiliaries(gb25_model *m);     /* src/precompile.jl:36,113-115  compute_auxiliaries_workload! */
gb25_status gb25_fill_diffusivity_halos(gb25_model *m);  /* src/precompile.jl:37,117-119  (closure=nothing: no-op) */
gb25_status gb25_compute_momentum_tendencies(gb25_model *m); /* src/precompile.jl:63-73  */
gb25_status gb25_compute_tracer_tendencies(gb25_model *m);   /* src/precompile.jl:75-111 */
gb25_status gb25_compute_boundary_tendencies(gb25_model *m); /* src/precompile.jl:52-61: applied inside the tendency kernels (gb25_set_top_flux) */
gb25_status gb25_compute_tendencies(gb25_model *m);      /* src/precompile.jl:38,48-50 compute_tendencies_workload! */
gb25_status gb25_ab2_step(gb25_model *m, double dt, int euler); /* src/precompile.jl:39,121-123 ab2_step_workload! */
gb25_status gb25_correct_velocities_and_cache_previous_tendencies(gb25_model *m, double dt); /* src/precompile.jl:41,125-127 */
gb25_status gb25_update_state(gb25_model *m);            /* Oceananigans.TimeSteppers.update_state! (correctness/..._run.jl:53-54) */
/* (the phase entry points are for single-domain models; on a slab of a decomposition only the composites are valid) */

/* ---- composites: GordonBell25.first_time_step!/time_step!/loop! (src/timestepping_utils.jl:21-45).
 *      dt is read from the clock, the model is mutated in place, nothing is returned. */
gb25_status gb25_first_time_step(gb25_model *m);
gb25_status gb25_time_step(gb25_model *m);
gb25_status gb25_loop(gb25_model *m, int32_t n_inner);

/* ---- x-slab decomposition.  Replaces Oceananigans.Distributed(arch; partition=Partition(Rx,
 *      Ry, 1)) + XLA's collective-permutes (sharding/sharded_baroclinic_instability_simulation_run.jl:65-72): each rank
 *      creates ONE slab (cfg.rank, cfg.nranks; Nx is the global size) and gives it an exchange context; after that
 *      gb25_first_time_step / gb25_time_step / gb25_loop are valid on the slab and `loop!(model, Ninner)` stays ONE
 *      call, as in the reference (:147,162).  Inside a step: packed halo columns travel by ncclSend/ncclRecv (RCCL over
 *      xGMI) on a second HIP stream, overlapped with the own-column work and the interior momentum tendencies; the 21
 *      barotropic substeps need no exchange (wide halos, filled once).  No collective, no host synchronisation.
 *
 *      Setters of a decomposed model (gb25_set_field, gb25_set_dt, gb25_set_option, gb25_set_baroclinic_instability,
 *      gb25_field_device_ptr) are COLLECTIVE: every rank makes the same call in the same order, like set!(model, ...)
 *      on a Distributed grid.  With the RCCL transport they shake hands with both neighbours and return
 *      GB25_ERR_STATE on a mismatch.
 *
 *      2-D decomposition (cfg.ranks_y = Ry > 1; Partition(Rx, Ry, 1), config 4 of the reference: 4 x 2): rank = ry Rx + rx
 *      owns a window of columns AND rows; Nx, Ny stay the global sizes, every field is the window (a y-face field has one
 *      row more only on the ranks below a wall).  Every exchange in x is followed by one of whole rows with the southern /
 *      northern neighbour (rank -/+ Rx; none beyond the walls and the fold); the fold partner of a top-row rank is the
 *      mirrored rank of that row.  Same calls, same transports. */
#define GB25_UNIQUE_ID_BYTES 128
/* rank 0 calls this and hands the 128 bytes to every rank by whatever means the host has (MPI_Bcast, a torch.distributed
 * store, a file): ncclGetUniqueId */
gb25_status gb25_comm_unique_id(void *id_out);
/* every rank, with the same id: ncclCommInitRank(nranks = cfg.nranks, rank = cfg.rank) on cfg.device.  nranks == 1 with
 * slab_mode == 1 is the self-ring.  With GB25_REHEARSE_ALONE=1 in the environment a rank of ANY decomposition gets a communicator
 * of size one and is its own neighbour on every side (a timing proxy of one rank with a GPU to itself -- tools/slab_selfring.py
 * --mesh; what crosses the seams is not a simulation's data). */
gb25_status gb25_comm_init_rccl(gb25_model *m, const void *unique_id);
/* all `n` slabs of one decomposition live in THIS process on one device (tests of decomposition invariance; also a
 * single-process multi-slab run): slabs[r] must have cfg.rank == r, cfg.nranks == n (and the same ranks_y).  The composites called on ANY of
 * them step all of them in lock-step; the exchange is a ring of device-to-device copies. */
gb25_status gb25_comm_init_local(gb25_model *const *slabs, int32_t n);
/* the host moves the buffers: fn is called once per exchange with device pointers of this slab's two packed sends and
 * two receive buffers (nbytes each); it must return 0 after recv_west holds the west neighbour's send_east and
 * recv_east the east neighbour's send_west.  The library synchronises the issuing stream before the call (no overlap):
 * a rehearsal transport for setups where RCCL cannot run (two ranks on one device).  buffer_set 3 and 4 (tripolar grid
 * only) are exchanges with the FOLD PARTNER, rank nranks-1-rank (2-D decomposition: the mirrored rank of the same row):
 * send_west goes to it, recv_west must hold what it sent, the east pointers are NULL.  buffer_set 5, 6, 7 (2-D decomposition
 * only) are the y halos: the "west" pointers belong to the SOUTHERN neighbour (rank - Rx), the "east" pointers to the
 * NORTHERN one (rank + Rx); a side without a neighbour has NULL pointers.  buffer_set 8, 9, 10 (closure = CATKE only: the TKE
 * tracer and J^b after their step inside compute_diffusivities!) go to the ring neighbours, to the southern / northern neighbours
 * and to the fold partner respectively, with the pointer conventions above. */
typedef int32_t (*gb25_exchange_fn)(void *user, int32_t buffer_set, const void *send_west, const void *send_east,
                                    void *recv_west, void *recv_east, int64_t nbytes);
gb25_status gb25_comm_init_callback(gb25_model *m, gb25_exchange_fn fn, void *user);
gb25_status gb25_comm_finalize(gb25_model *m);
/* transport: 0 none, 1 RCCL, 2 local ring, 3 host callback; comm_ranks: the communicator's size as RCCL reports it
 * (ncclCommCount; 0 without RCCL).  Either pointer may be NULL. */
gb25_status gb25_comm_info(const gb25_model *m, int32_t *transport, int32_t *comm_ranks);
/* (tests) the sends / receives one rank of an Rx x Ry decomposition posts for an exchange group, in posting order, as text;
 * returns the bytes needed incl. the terminator.  No GPU is touched. */
int64_t gb25_debug_exchange_plan(int32_t Rx, int32_t Ry, int32_t rank, int32_t folded_grid, int32_t group, char *out, int64_t cap);
/* velocities_ready: the momentum look-ahead of the next step exists (its sub-cycle can run beside the tracer kernel);
 * subcycle_adopted: the last step adopted the sub-cycle look-ahead instead of sub-cycling inside the step. */
gb25_status gb25_lookahead_state(const gb25_model *m, int32_t *velocities_ready, int32_t *subcycle_adopted);
/* The order of operations of one time step (bit 0 of `first`: of first_time_step!; bit 1: on a folded grid; bit 2: of a
 * coupled model; bit 3: with the previous step's look-ahead chain still in flight) of `nslabs` slabs as text, without
 * touching a GPU; bit 4: of a 2-D decomposition; bit 5: of a step that keeps the corrector inside its consumers; bit 6: with the
 * bundle unpacked on the exchange stream (tests of the sequencing on CPU-only machines).  Returns the bytes needed, incl. the terminator. */
int64_t gb25_debug_sequence(int32_t nslabs, int32_t first, int32_t adopted, int32_t ready, char *out, int64_t cap);

/* ---- state dump: save_model_state(dir, model, arch; label) (src/sharded_io.jl:70-96,122-138; called after each loop
 *      of the benchmark script, sharding/sharded_..._run.jl:151-155,167-171).  Every rank writes only its own slab --
 *      no communication -- to <directory>/<label>/fields_rank<rank>.npz (uncompressed NumPy .npz): per field of
 *      Oceananigans.fields(model) the local interior (<name>.data), its slice of the global array (<name>.slice: i0, i1,
 *      j0, j1, k0, k1) and the global shape, plus iteration, time, rank, nranks.  Offline gather (load_all_fields,
 *      src/sharded_io.jl:198-213): gb-25_amd/sharded_io.py. */
gb25_status gb25_save_state(gb25_model *m, const char *directory, const char *label);

/* ---- built-in per-kernel HIP-event timing (bench.py's roofline numbers) */
gb25_status gb25_profile_enable(gb25_model *m, int on); /* 0: off, 1: every kernel, 2 + k: kernel k alone */
gb25_status gb25_profile_reset(gb25_model *m);
gb25_status gb25_profile_get(gb25_model *m, gb25_kernel k, int64_t *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* GB25_H */

// gb25_api.hip -- host side of libgb25hip.so: the C ABI declared in include/gb25.h.
//
// Owns the device state of one model (one x-slab on one GPU), builds the grid metrics,
// sequences the kernels of kernels.hpp in the reference's phase order
// (GB-25 src/precompile.jl:31-42) and exposes the per-phase entry points.
// There is no CPU fallback: without a HIP device gb25_create fails with GB25_ERR_NO_DEVICE.
#include "../../include/gb25.h"
#include <dlfcn.h>
#include "kernels.hpp"
#include "tendency_kernels.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <atomic>
#include <vector>

using namespace gb25;

namespace {

constexpr int PAD = 2;  // metric tables extend 2 entries beyond the halo

struct Field {
  real* d = nullptr;
  int nx = 0, ny = 0, nz = 0;  // parent dims
  size_t elems() const { return (size_t)nx * ny * nz; }
};

struct EventPair {
  hipEvent_t a, b;
};

}  // namespace

struct gb25_model {
  gb25_config cfg;
  int Nx = 0;  // local slab width
  // 2-D (x, y) decomposition, Partition(Rx, Ry, 1) of the reference (sharding/sharded_baroclinic_instability_simulation_run.jl:
  // 65-72): rank = ry Rx + rx owns the columns [rx Nx, (rx + 1) Nx) and the rows [j0, j0 + Ny) of the global grid.  Ry = 1: x slabs.
  int Ny = 0;  // local rows (cfg.Ny / Ry)
  int Rx = 1, Ry = 1, rx = 0, ry = 0, j0 = 0;
  bool ys_open = false, yn_open = false;   // a southern / northern neighbour rank exists (no wall on that side)
  Grid g;
  Field f[GB25_FIELD_COUNT];
  Field pp[3];                   // ping-pong partners of eta, U, V
  Field dpx, dpy;                // p'(i)-p'(i-1), p'(j)-p'(j-1), differenced in fp64 by k_compute_p, stored fp32
  Field colsum[2];               // column integrals of u, v after the AB2 update (consumed by the corrector)
  bool colsum_valid = false;
  // AB2 look-ahead (kernels_v2.hpp, Ab2Ahead): the tracer tendency kernel also writes T, S of the next time level
  // into `ahead`; ab2_step! adopts them by pointer exchange when dt, chi and every input are still the same.
  Field ahead[2];
  bool ahead_valid = false, ptr_exposed = false;
  real ahead_dt = 0, ahead_chi = 0;
  // ... and the momentum kernel does the same for u, v (UvAhead): partner buffers of u, v, of G.U, G.V and of the
  // corrector's column integrals, plus the per-chunk partial sums the kernel leaves for k_ab2_velocities_finish
  Field ahead_uv[2], ahead_G[2], ahead_colsum[2];
  real* uv_partials = nullptr;
  bool ahead_uv_valid = false;
  real ahead_uv_dt = 0, ahead_uv_chi = 0;
  // ... and once G.U, G.V of the next step exist (momentum look-ahead), so does everything its split-explicit
  // sub-cycle needs: it runs on the side stream beside the tracer tendency kernel (latency-bound next to
  // issue-bound) into partner buffers of eta, U, V and of the filtered state, adopted like the others.
  Field pp2[3];                      // second scratch set: the sub-cycle never writes the arrays it starts from
  Field ahead_eta[3], ahead_bar[3];  // partners of eta, U, V and of eta_bar, U_bar, V_bar
  real* bars_ahead = nullptr;        // (the three partners of the averages are one allocation, like `bars`)
  bool ahead_baro_valid = false;
  // pHY' is a diagnostic: inside a composite step only its differences are stored (4 of the kernel's 20 B/cell
  // saved) and the field is recomputed when the host asks for it; pinned to "always stored" once its pointer is out
  bool phy_stale = false, phy_pinned = false;
  bool baro_inflight = false;        // a look-ahead sub-cycle is on the side stream and nobody has waited for it yet
  bool baro_adopted = false;         // staged path: stage 0 of this step adopted the sub-cycle look-ahead
  int baro_ahead = 1;                // option SUBCYCLE_LOOKAHEAD = 0: sub-cycle inside the step, on the critical path
  hipEvent_t ev_baro = nullptr, ev_mom = nullptr;
  int fill_fused = 1;                // y, z and periodic-x fills of a single slab in one launch (option FILL_FUSED = 0: two)
  int ab2_ahead = 1;                 // option AB2_LOOKAHEAD: 0 = stand-alone AB2 kernels, 1 = both look-aheads, 2 = tracers only
  real* bars = nullptr;         // contiguous etabar | Ubar | Vbar
  std::vector<real*> dev_tables;
  std::vector<double> h_metric[11];
  // orthogonal curvilinear grid (grid_type >= 2): the 14 horizontal metrics by location, fp64, parent layout of a (c,f)
  // field (gb25_get_metric2); cell-centre coordinates in degrees for analytic bottoms
  std::vector<double> h_curv[GB25_M2_COUNT];
  // The HOST's grid (gb25_set_curvilinear_grid, gb25_set_vertical_faces, gb25_set_bottom_height): the 14 horizontal metrics
  // over the GLOBAL parent extent ((Nx_global + 2H) x (Ny + 2H + 1) doubles each, i fastest), the Nz + 1 vertical faces, the
  // bottom height at the global cell centres.  Empty: the built-in generators of cfg.grid_type.
  std::vector<double> host_curv[GB25_M2_COUNT];
  std::vector<double> host_zf, host_bottom;
  int metric_off_j = 0, metric_off_k = 0;
  // substepping
  int Ns = 0;
  double dtau_frac = 0;
  std::vector<double> weights;
  // work arrays of the sub-cycle: widened by W columns either side on a slab (wide halos, filled once per step), tall by Wy
  // rows beyond the pivot row on a folded grid (images of the rows south of it, filled once per step); a single folded
  // domain has W = 0
  int W = 0, Wy = 0, Wys = 0;        // (Wys: rows below row 0 -- the southern neighbour's, 2-D decomposition)
  real* tall_buf = nullptr;          // single folded domain: the buffer its image rows pass through (k_tall_rows)
  Field wide[2][3];  // [pingpong][eta,U,V]
  Field wideG[2];    // GU, GV
  Field wideBar[3];  // running averages on the wide domain
  // clock
  double time = 0, last_dt = 0;
  int64_t iteration = 0;
  // streams / timing
  hipStream_t own_stream = nullptr, stream = nullptr, side_stream = nullptr;
  hipStream_t baro_stream = nullptr;   // the look-ahead sub-cycle of a single domain: a stream of its own, so that the
                                       // pressure of the next step (side stream) need not queue behind its tail
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  // Order of the two tendency kernels inside a composite step of a single domain (option TRACERS_FIRST, default 1): with the
  // tracer kernel first, T and S of the next time level (its look-ahead) exist before the momentum kernel runs, so the next
  // step's pressure -- forked off `ev_tend`, recorded right behind the momentum kernel's finish launch -- runs beside the
  // next step's sub-cycle (a chain of short LDS phases that leaves HBM idle) instead of beside w.  ev_ts: T, S of this step
  // and their halo cells are complete (side stream); the tracer kernel waits for it, the momentum kernel for ev_join.
  hipEvent_t ev_ts = nullptr, ev_tend = nullptr;
  hipEvent_t ev_strips = nullptr;    // slab: the pressure strips next to the x halos are done (exchange stream, stage 33)
  bool strips_issued = false;
  int early_strips = 1;              // option EARLY_STRIPS
  int tracers_first = 1;
  bool tend_forkable = false;       // the last tendency evaluation was a composite step's, tracers first: ev_tend covers both kernels
  bool two_streams = true;          // option TWO_STREAMS = 0: strictly sequential phases on one stream
  bool profile = false;
  int profile_only = -1;             // >= 0: time this kernel id alone (keeps the event records out of the other launches)
  std::vector<EventPair> pending[GB25_K_COUNT];
  std::vector<EventPair> free_events;
  int64_t prof_count[GB25_K_COUNT] = {0};
  int64_t prof_seen[GB25_K_COUNT] = {0};
  double prof_ms[GB25_K_COUNT] = {0};
  std::string err;
  int baro_block = 5;                // substeps per barotropic launch (option SUBCYCLE_BLOCK = 1: one launch per substep; 5: 64 x 17 tiles,
                                     // four blocks per CU, every block of a 1440 x 720 launch resident at once: 0.16 ms for 21 substeps against 0.21 with 7)
  int kernel_gen = 2;                // 2: LDS / flux-sharing tendency kernels (tendency_kernels.hpp); 1: direct-stencil kernels
  int pressure_bits = 64;            // option PRESSURE_PRECISION: 64 = fp64 EOS + integral (default); 32 = the float type's own
  // single periodic domain: the last writers of u, v (corrector), T, S (tracer look-ahead) and eta, U, V (last barotropic
  // launch) also write the halo cells the fills derive from them, and the fill launches leave the step.  Halo cells
  // deeper than one layer in y / z are static while stepping, so their x images only need the complete fills of the two
  // steps (one per buffer of each alternating pair) that follow a host write.
  int n_cu = 256;                    // compute units of the device
  int fold_fills = 1;
  bool composite = false;            // inside gb25_time_step / gb25_loop (the phase entry points leave halos alone, as the
                                     // reference's phases do: there the producers do not fold)
  int complete_fills_needed = 2;
  bool ahead_ts_folded = false, ahead_eta_folded = false, last_baro_folded = false;
  int split_tendencies = 1;          // slab of a decomposition: interior tile columns before the x-halo bundle has arrived
  int baro_whole = 1;                // narrow slab: the whole sub-cycle in one launch (option SUBCYCLE_WHOLE)
  int sub_priority = 0;              // slab: the stream of the look-ahead's substeps is a high-priority one (read when the exchange context is built)
  // immersed boundary (GridFittedBottom): first active level per column on the columns [-kb_E, Nx + kb_E) x [0, Ny)
  // (host), the folded tables of device_common.hpp (device), the depths of the wide barotropic arrays of a slab
  bool immersed = false;             // some cell is immersed: the IMM kernel variants run
  int kb_E = 0, kb_Ey = 0;           // (kbot: pitch Nx + 2 kb_E, row of local row j: j + kb_Ey)
  std::vector<int> kbot;
  unsigned* d_ord[4] = {nullptr, nullptr, nullptr, nullptr};
  real* d_H[4] = {nullptr, nullptr, nullptr, nullptr};   // Hfc, Hcf, rHfc, rHcf (parent layout of a (c,f) field)
  real* d_wideH[2] = {nullptr, nullptr};                 // Hfc, Hcf on the wide barotropic layout of a slab
  // curvilinear slab: dyfc, dxcf, 1/azcc, 1/dxfc, 1/dycf on the wide barotropic layout, and (zipper fold) the metrics of
  // the cells row Ny-1 mirrors onto, by own wide column (CurvBaro::mir, kernels.hpp)
  real* d_wideM[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  real* d_top_flux[4] = {nullptr, nullptr, nullptr, nullptr};   // FluxBoundaryCondition at the top of u, v, T, S
  // PrescribedAtmosphere at the cell centres (data-free forcing); all seven set: the composites compute the
  // atmosphere-ocean fluxes after every step.  d_tau: the stress components at the centres.
  double* d_atm[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double* d_tau[2] = {nullptr, nullptr};
  bool coupled = false;
  int tracer_order = 5;              // tracer_advection = WENO(order = 5 | 7)
  double bottom_drag = 0.0;          // quadratic bottom drag coefficient (0: none); the two flux arrays behind Grid.bottom_flux
  real* d_bottom_flux[2] = {nullptr, nullptr};
  bool halo_colsum_valid = false;    // slab: the x-halo columns of colsum hold their owner's integrals (packed with group 0)
  // the corrector applied inside its consumers (k_corrector_2d): du, dv of the current step; while uv_lazy is set, u and
  // v in memory lack them (only between the steps of one composite call: gb25_loop applies them before it returns)
  Field corr[2];
  bool uv_lazy = false;
  bool lazy_head_done = false;       // ... and its du, dv, chunk bases of w are made (stage 20)
  bool step_lazy = false;            // slab: this step keeps the corrector inside its consumers (decided in stage 0)
  int lazy_corrector = 1;            // option LAZY_CORRECTOR
  // ... and with it w ON THE FLY (option W_ON_THE_FLY): in those steps the tendency kernels carry w up their chunks of levels
  // from the divergence of the transports they hold; no k_compute_w launch, no w traffic.  wbase: w at the first level of every
  // chunk (k_w_bases).  While w_stale is set the field w in memory is the one of an earlier step (recomputed with the
  // velocities: materialize_uv).  Results agree with the stand-alone w to round-off, not to the last bit.
  real* wbase = nullptr;
  int w_fly = 1;
  bool w_stale = false, w_fly_now = false;
  bool corr_out = false;             // corrector_impl: the sweep also writes du, dv of the own columns into corr[0], corr[1]
  // The corrector through the tracer kernel (round 4; grids with a bottom / curvilinear / folded, single domain): du, dv as 2-D
  // fields; the tracer kernel -- first in the step, and every u, v is some cell's west / south face -- adds them as it loads and
  // writes the corrected velocities into this second pair of arrays, which then BECOME u, v (pointer exchange) and get their halo
  // cells from the ordinary fill; the momentum kernel reads corrected velocities like any other.  No sweep over u and v.
  Field uvc[2];
  bool uv_corr_pending = false;      // this step's tracer kernel writes the corrected velocities
  // levels a block of the momentum / tracer tendency kernel marches through (options MOMENTUM_CHUNK_LEVELS,
  // TRACER_CHUNK_LEVELS): fewer, longer chunks amortise the start-up of the vertical windows, more chunks fill the chip.
  // The momentum chunking is also the association of every column integral of u, v (all their producers share it).
  int mom_chunk_levels = 12, trc_chunk_levels = 12;
  // closure = VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), kappa, nu); both zero: closure = nothing
  double nu = 0, kappa = 0;
  bool catke = false;                // closure = CATKEVerticalDiffusivity(): the fields GB25_E .. GB25_JB exist
  Field catke_b, catke_scratch;      // N^2 on the faces; the unused half of the two-wide tracer kernel's output
  gb25_catke_parameters catke_par;   // (gb25_default_catke_parameters at creation)
  Field catke_src;                   // 2-D: the top boundary condition of e (surface TKE flux / dz of the top cell)
  real* catke_e_star = nullptr;   // where the last k_catke_tke_step left e* (e, or catke_scratch)
  bool implicit_lds_raised[2] = {false, false};   // k_implicit_vertical's dynamic-LDS attribute (Nz > 128), likewise
  bool whole_attr_set[2][2] = {{false, false}, {false, false}};   // k_barotropic_whole's dynamic-LDS attribute, per instance, on THIS model's device
  int comm_timeout_s = 180;          // option COMM_TIMEOUT_SECONDS
  bool roctx_ranges = true;          // option ROCTX_RANGES
  int substep_order = 0;             // option SUBSTEP_ORDER
  double catke_prev_time = 0;        // diffusivity_fields.previous_compute_time
  // where diffusivity_fields.previous_velocities live (single domain: no copies in the steady state).  0: in their fields
  // (GB25_PREV_U/V).  1: they ARE the current u, v (compute_diffusivities! just ended: u- <- u is pending).  2: in the
  // look-ahead's partner buffers -- the AB2 step that followed adopted the look-ahead by exchanging pointers, and the buffers
  // it left hold u, v as they were at the last compute_diffusivities!, halos included.  materialize_prev_uv brings them home.
  int prev_uv_src = 0;
  bool catke_stale_e_halos = false;  // option CATKE_STALE_E_HALOS
  bool n2_fresh = false;             // (unused since N^2 = g (alpha dzT - beta dzS) has a kernel of its own)
  Field catke_gam[2];                // Nz > 64: the elimination factors of the streamed implicit solve
  real* d_implicit[2] = {nullptr, nullptr};   // elimination tables of the implicit solve for (u, v) and (T, S): lo | 1/beta | gamma
  double implicit_key[2][2] = {{0, 0}, {0, 0}};   // the (dt, K) they were built for
  bool slab = false;                 // x halos come from a neighbour (nranks > 1, or the self-ring of slab_mode = 1)
  struct SlabGroup* group = nullptr; // exchange context (transport, buffers, comm stream) once gb25_comm_init_* was called
  int group_index = 0;               // this slab's position in group->slabs
  int fold_flip = 0;                 // folded slab: which of the two widened state sets the next substep reads
};

namespace {

gb25_status fail(gb25_model* m, gb25_status s, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (m) m->err = buf;
  return s;
}

#define HIPCHK(call)                                                                                  \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess)                                                                             \
      return fail(m, GB25_ERR_HIP, "%s:%d: %s failed: %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
  } while (0)
#define CHECK_MODEL(m) \
  if (!(m)) return GB25_ERR_INVALID_ARGUMENT
#define LAUNCHCHK() HIPCHK(hipGetLastError())

bool is_v_shaped(int id) {
  return id == GB25_V || id == GB25_GN_V || id == GB25_GM_V || id == GB25_BT_V || id == GB25_V_BAR ||
         id == GB25_GN_BT_V || id == GB25_PREV_V;
}
bool is_2d(int id) { return (id >= GB25_ETA && id <= GB25_GN_BT_V) || id == GB25_JB; }
bool is_catke_field(int id) { return id >= GB25_E && id <= GB25_PREV_V; }

// --- profiling helpers -----------------------------------------------------------------------
// Named ranges for a profiler's timeline, as the reference wraps its entry points in Reactant.Profiler.annotate("first_time_step" /
// "time_step" / "loop") (GB-25 src/timestepping_utils.jl:22,30,38): roctx ranges around the composites and around the issue of
// every phase of src/precompile.jl:31-42, so that a `rocprofv3 --kernel-trace --marker-trace` of gb25_loop shows which phase a
// kernel belongs to.  The marker library is resolved at run time (rocprofv3 preloads librocprofiler-sdk-roctx; the legacy
// libroctx64 serves the older tools); without one the ranges cost a null-pointer test.  Option ROCTX_RANGES = 0 turns them off.
struct RoctxApi {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  RoctxApi() {
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!lib) continue;
      push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
      pop = (int (*)())dlsym(lib, "roctxRangePop");
      if (push && pop) return;
      push = nullptr; pop = nullptr;
    }
  }
};
inline RoctxApi& roctx() {
  static RoctxApi api;
  return api;
}
struct Range {
  bool on;
  Range(const gb25_model* m, const char* name) : on(m->roctx_ranges && roctx().push != nullptr) {
    if (on) roctx().push(name);
  }
  ~Range() {
    if (on) roctx().pop();
  }
};
// the phase a kernel timer belongs to (src/precompile.jl:31-42; ":" + what inside the phase)
inline const char* phase_name(int k) {
  switch (k) {
    case GB25_K_FILL_HALOS: return "tupled_fill_halo_regions";
    case GB25_K_COMPUTE_W: return "compute_auxiliaries:w";
    case GB25_K_COMPUTE_P: return "compute_auxiliaries:pHY";
    case GB25_K_CLOSURE: return "compute_auxiliaries:diffusivities";
    case GB25_K_GU: case GB25_K_GV: return "compute_tendencies:momentum";
    case GB25_K_TRACERS: return "compute_tendencies:tracers";
    case GB25_K_AB2_VELOCITIES: return "ab2_step:velocities";
    case GB25_K_AB2_TRACERS: return "ab2_step:tracers";
    case GB25_K_BAROTROPIC: return "ab2_step:free_surface";
    case GB25_K_IMPLICIT: return "ab2_step:implicit_step";
    case GB25_K_CORRECTOR: return "correct_velocities_and_cache_previous_tendencies";
    case GB25_K_FLUXES: return "compute_atmosphere_ocean_fluxes";
    default: return "gb25";
  }
}
struct Timed {
  gb25_model* m;
  int k;
  EventPair ev;
  bool on;
  Range range;
  Timed(gb25_model* m_, int k_) : m(m_), k(k_), on(m_->profile && (m_->profile_only < 0 || m_->profile_only == k_)), range(m_, phase_name(k_)) {
    // one kernel alone: every fourth launch is timed (the event records cost the step ~1.6 % when every launch carries them)
    if (on && m->profile_only == k_ && (m->prof_seen[k_]++ & 3) != 0) on = false;
    if (!on) return;
    if (!m->free_events.empty()) {
      ev = m->free_events.back();
      m->free_events.pop_back();
    } else {
      hipEventCreate(&ev.a);
      hipEventCreate(&ev.b);
    }
    hipEventRecord(ev.a, m->stream);
  }
  void stop() {
    if (!on) return;
    hipEventRecord(ev.b, m->stream);
    m->pending[k].push_back(ev);
    on = false;
  }
  ~Timed() { stop(); }
};

void resolve_profile(gb25_model* m) {
  for (int k = 0; k < GB25_K_COUNT; k++) {
    for (auto& ev : m->pending[k]) {
      hipEventSynchronize(ev.b);
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess) {
        m->prof_ms[k] += ms;
        m->prof_count[k] += 1;
      }
      m->free_events.push_back(ev);
    }
    m->pending[k].clear();
  }
}

// --- grid ------------------------------------------------------------------------------------
// simple_latitude_longitude_grid (GB-25 src/model_utils.jl:56-65): regular lat-lon spacing,
// exponential_z_faces(Nz, depth, h) vertical faces, spherical-shell metrics.
gb25_status upload_table(gb25_model* m, const std::vector<double>& h, int off, const real** out) {
  std::vector<real> f(h.size());
  for (size_t a = 0; a < h.size(); a++) f[a] = (real)h[a];
  real* d = nullptr;
  HIPCHK(hipMalloc(&d, f.size() * sizeof(real)));
  HIPCHK(hipMemcpy(d, f.data(), f.size() * sizeof(real), hipMemcpyHostToDevice));
  m->dev_tables.push_back(d);
  *out = d + off;
  return GB25_OK;
}

gb25_status build_grid(gb25_model* m) {
  const gb25_config& c = m->cfg;
  const int H = c.halo, Ny = m->Ny, Nz = c.Nz;
  // (the row tables of a rank of a 2-D decomposition also cover the rows its sub-cycle is widened by)
  const int padj = PAD + (m->Ry > 1 ? c.substeps + 2 + H + 8 : 0);
  const int nj = Ny + 2 * H + 2 * padj + 2, nk = Nz + 2 * H + 2 * PAD + 2;
  const int offj = H + padj, offk = H + PAD;  // table index of 0-based logical index 0
  m->metric_off_j = offj;
  m->metric_off_k = offk;
  const double d2r = M_PI / 180.0;
  const double dlam = (c.lon_east - c.lon_west) / c.Nx, dphi = (c.lat_north - c.lat_south) / c.Ny, R = c.radius;
  std::vector<double>&phif = m->h_metric[GB25_M_PHIF], &phic = m->h_metric[GB25_M_PHIC],
  &dxc = m->h_metric[GB25_M_DXC], &dxf = m->h_metric[GB25_M_DXF], &azc = m->h_metric[GB25_M_AZC],
  &azf = m->h_metric[GB25_M_AZF], &fcor = m->h_metric[GB25_M_FCOR];
  phif.assign(nj, 0); phic.assign(nj, 0); dxc.assign(nj, 0); dxf.assign(nj, 0);
  azc.assign(nj, 0); azf.assign(nj, 0); fcor.assign(nj, 0);
  for (int a = 0; a < nj; a++) {
    int j = a - offj + m->j0;  // 0-based GLOBAL face / centre index
    phif[a] = c.lat_south + j * dphi;
    phic[a] = c.lat_south + (j + 0.5) * dphi;
  }
  for (int a = 0; a < nj; a++) {
    dxc[a] = R * std::cos(phic[a] * d2r) * dlam * d2r;
    dxf[a] = R * std::cos(phif[a] * d2r) * dlam * d2r;
    fcor[a] = 2.0 * c.Omega * std::sin(phif[a] * d2r);
    if (a + 1 < nj) azc[a] = R * R * dlam * d2r * (std::sin(phif[a + 1] * d2r) - std::sin(phif[a] * d2r));
    if (a > 0) azf[a] = R * R * dlam * d2r * (std::sin(phic[a] * d2r) - std::sin(phic[a - 1] * d2r));
  }
  // vertical faces: z_k ~ exp(k/h), k = 1..Nz+1, mapped to [0, -depth] and reversed (exponential_z_faces, src/model_utils.jl:
  // 56-62) -- or the host's own faces (gb25_set_vertical_faces)
  std::vector<double> zint(Nz + 1);
  const double h = c.zexp_h, e1 = std::exp(1.0 / h), eN = std::exp((Nz + 1.0) / h);
  for (int k = 1; k <= Nz + 1; k++) zint[Nz + 1 - k] = -c.depth * (std::exp(k / h) - e1) / (eN - e1);
  zint[Nz] = 0.0;
  if (!m->host_zf.empty()) zint = m->host_zf;
  // the faces are numbers of the model's float type (a Float32 host holds Float32 faces); centres and spacings derive from them
  for (double& z : zint) z = (double)(real)z;
  std::vector<double>&zf = m->h_metric[GB25_M_ZF], &zc = m->h_metric[GB25_M_ZC], &dzc = m->h_metric[GB25_M_DZC],
  &dzf = m->h_metric[GB25_M_DZF];
  zf.assign(nk + 1, 0); zc.assign(nk, 0); dzc.assign(nk, 0); dzf.assign(nk, 0);
  const double dlo = zint[1] - zint[0], dhi = zint[Nz] - zint[Nz - 1];
  for (int a = 0; a <= nk; a++) {
    int k = a - offk;  // 0-based face index
    zf[a] = k < 0 ? zint[0] + k * dlo : (k > Nz ? zint[Nz] + (k - Nz) * dhi : zint[k]);
  }
  for (int a = 0; a < nk; a++) zc[a] = 0.5 * (zf[a] + zf[a + 1]);
  for (int a = 0; a < nk; a++) {
    dzc[a] = zf[a + 1] - zf[a];
    dzf[a] = a > 0 ? zc[a] - zc[a - 1] : zc[1] - zc[0];
  }
  Grid& g = m->g;
  g.Nx = m->Nx; g.Ny = Ny; g.Nz = Nz; g.H = H;
  g.sx = m->Nx + 2 * H;
  g.sy_c = Ny + 2 * H; g.sy_v = Ny + 2 * H + 1;
  g.pl_c = g.sx * g.sy_c; g.pl_v = g.sx * g.sy_v;
  g.x_periodic = !m->slab;
  g.jws = -m->j0;
  g.jwn = c.grid_type >= GB25_GRID_TRIPOLAR ? (1 << 20) : c.Ny - m->j0;
  g.dy = (real)(R * dphi * d2r);
  g.g = (real)c.g; g.rho0 = (real)c.rho0; g.Lz = (real)(zint[Nz] - zint[0]);
  gb25_status s;
  if ((s = upload_table(m, dxc, offj, &g.dxc))) return s;
  if ((s = upload_table(m, dxf, offj, &g.dxf))) return s;
  if ((s = upload_table(m, azc, offj, &g.azc))) return s;
  if ((s = upload_table(m, azf, offj, &g.azf))) return s;
  if ((s = upload_table(m, fcor, offj, &g.fcor))) return s;
  if ((s = upload_table(m, phic, offj, &g.phic))) return s;
  {
    auto recip = [](const std::vector<double>& a) {
      std::vector<double> r(a.size());
      for (size_t q = 0; q < a.size(); q++) r[q] = a[q] != 0.0 ? 1.0 / a[q] : 0.0;
      return r;
    };
    if ((s = upload_table(m, recip(dxc), offj, &g.rdxc))) return s;
    if ((s = upload_table(m, recip(azc), offj, &g.razc))) return s;
    if ((s = upload_table(m, recip(azf), offj, &g.razf))) return s;
    if ((s = upload_table(m, recip(dzc), offk, &g.rdzc))) return s;
    if ((s = upload_table(m, recip(dzf), offk, &g.rdzf))) return s;
    g.rdy = (real)(1.0 / (R * dphi * d2r));
    g.rLz = (real)(1.0 / (zint[Nz] - zint[0]));
  }
  if ((s = upload_table(m, zc, offk, &g.zc))) return s;
  if ((s = upload_table(m, dzc, offk, &g.dzc))) return s;
  if ((s = upload_table(m, dzf, offk, &g.dzf))) return s;
  return GB25_OK;
}

// --- orthogonal curvilinear grids ---------------------------------------------------------------------------------
// TripolarGrid(arch; size, halo, z) (GB-25 src/model_utils.jl:134-137), restated as an analytic bipolar cap after Murray
// (1996): first pole at 70 E, both at 55 N, southern edge as configured, northern edge the fold line between the poles
// [UPSTREAM-UNVERIFIED: Oceananigans builds its cap numerically; topology, poles and fold agree, the interior coordinate
// lines of the cap need not].  South of 55 N the grid is the lat-lon grid.  In the polar stereographic plane the cap
// carries bipolar coordinates with the foci at the poles, chosen so that the rim keeps its longitudes and the meridian
// halfway between the poles its latitudes.  Metrics: great-circle distances between neighbouring nodes, spherical areas
// of the quadrilaterals they span.  oracle/gb25_oracle.c states the same construction; tests compare the two.
struct GNode { double x, y, z, lam, phi; };
GNode sphere_node(double lam, double phi) {
  const double d2r = M_PI / 180.0;
  phi = std::max(phi, -89.999);   // (halo rows of coarse grids beyond the south pole: never used)
  return {std::cos(phi * d2r) * std::cos(lam * d2r), std::cos(phi * d2r) * std::sin(lam * d2r), std::sin(phi * d2r), lam, phi};
}
// f(j) for every row j of [j_lo, j_hi) on the host's cores (rows in blocks of eight from a shared counter).  The grid and bottom
// tables of a curvilinear model are elliptic-function work per column -- 20 s on one core for config 5's 4320 x 2160 columns.
template <class F>
void parallel_rows(int j_lo, int j_hi, F f) {
  const unsigned hw = std::thread::hardware_concurrency();
  const int nt = (int)std::min<unsigned>(hw ? hw : 1u, 32u);
  if (j_hi - j_lo < 64 || nt <= 1) {
    for (int j = j_lo; j < j_hi; j++) f(j);
    return;
  }
  std::atomic<int> next{j_lo};
  std::vector<std::thread> th;
  for (int t = 0; t < nt; t++)
    th.emplace_back([&]() {
      for (;;) {
        const int j = next.fetch_add(8);
        if (j >= j_hi) break;
        for (int q = j; q < std::min(j + 8, j_hi); q++) f(q);
      }
    });
  for (auto& t : th) t.join();
}
GNode tripolar_node(double lamt, double phit) {
  const double d2r = M_PI / 180.0, lamP = 70.0, phiP = 55.0;
  if (phit > 90.0) {   // beyond the fold: the image point
    phit = 180.0 - phit;
    lamt = 2 * lamP - lamt;
  }
  if (phit <= phiP) return sphere_node(lamt, phit);
  const double rP = std::tan((90.0 - phiP) / 2 * d2r), rt = std::tan((90.0 - phit) / 2 * d2r) / rP;
  const double th = (lamt - lamP) * d2r, ct = std::cos(th), sth = std::sin(th), st = std::fabs(sth);
  const double sg = 2 * std::atan2(1.0, rt), D = 1.0 - std::cos(sg) * st;
  const double xw = ct / D, yw = (sth < 0 ? -1.0 : 1.0) * std::sin(sg) * st / D;
  const double rz = rP * std::sqrt(xw * xw + yw * yw);
  return sphere_node(lamP + std::atan2(yw, xw) / d2r, 90.0 - 2 * std::atan(rz) / d2r);
}
double gc_dist(const GNode& a, const GNode& b, double R) {
  const double cx = a.y * b.z - a.z * b.y, cy = a.z * b.x - a.x * b.z, cz = a.x * b.y - a.y * b.x;
  const double d = R * std::atan2(std::sqrt(cx * cx + cy * cy + cz * cz), a.x * b.x + a.y * b.y + a.z * b.z);
  return std::max(d, 100.0);   // the coordinate lines meet at the poles: keep the metrics finite there
}
double tri_area(const GNode& a, const GNode& b, const GNode& c) {
  const double t = a.x * (b.y * c.z - b.z * c.y) + a.y * (b.z * c.x - b.x * c.z) + a.z * (b.x * c.y - b.y * c.x);
  const double d = 1.0 + (a.x * b.x + a.y * b.y + a.z * b.z) + (b.x * c.x + b.y * c.y + b.z * c.z) + (c.x * a.x + c.y * a.y + c.z * a.z);
  return 2 * std::atan2(std::fabs(t), d);
}
double quad_area(const GNode& a, const GNode& b, const GNode& c, const GNode& d, double R) {
  return std::max(R * R * (tri_area(a, b, c) + tri_area(a, c, d)), 1e4);
}

// The 14 horizontal metrics (gb25_metric2 order) and the physical coordinates of the cell centre at GLOBAL column ig,
// row j (any integers: halo rows and columns are generated like interior ones).  grid_type 2: the lat-lon metrics; 3, 4: the
// tripolar grid, whose Ny rows of cell centres run from the southern edge to 90 degrees (Oceananigans' TripolarGrid:
// range(southernmost_latitude, 90, length = Ny); "the north pole is a Center point"), the y faces half a spacing south of them.
// `own` rows only (j <= Ny - 1 on the tripolar grid): curv_metrics_at maps the rows beyond the pivot row onto their images.
void curv_metrics_own(const gb25_model* m, int ig, int j, double out[GB25_M2_COUNT], double* lam_c, double* phi_c) {
  const gb25_config& c = m->cfg;
  const bool tri = c.grid_type >= GB25_GRID_TRIPOLAR;
  const double d2r = M_PI / 180.0, R = c.radius;
  const double lam0 = tri ? 70.0 : c.lon_west, dlam = (tri ? 360.0 : (c.lon_east - c.lon_west)) / c.Nx;
  const double phiN = tri ? 90.0 : c.lat_north, dphi = (phiN - c.lat_south) / (tri ? c.Ny - 1 : c.Ny);
  if (!m->host_curv[0].empty()) {
    // the host's arrays: halo columns as the host holds them (its periodic images), columns beyond them wrapped
    const int H = c.halo, gsx = c.Nx + 2 * H;
    const int col = (ig >= -H && ig < c.Nx + H) ? ig + H : ((ig % c.Nx) + c.Nx) % c.Nx + H;
    const size_t o = (size_t)col + (size_t)gsx * (size_t)(std::min(std::max(j, -H), c.Ny + H) + H);
    for (int q = 0; q < GB25_M2_COUNT; q++) out[q] = m->host_curv[q][o];
    if (lam_c) *lam_c = 0.0;   // (not among the metrics: an analytic bottom cannot be placed on a host grid -- gb25_set_bottom_height)
    if (phi_c) *phi_c = out[GB25_M2_PHICC];
    return;
  }
  if (!tri) {
    const int a = m->metric_off_j + j - m->j0;   // (the row tables are indexed by LOCAL row)
    out[GB25_M2_DXFC] = out[GB25_M2_DXCC] = m->h_metric[GB25_M_DXC][a];
    out[GB25_M2_DXCF] = out[GB25_M2_DXFF] = m->h_metric[GB25_M_DXF][a];
    out[GB25_M2_DYFC] = out[GB25_M2_DYCC] = out[GB25_M2_DYCF] = out[GB25_M2_DYFF] = R * dphi * d2r;
    out[GB25_M2_AZCC] = out[GB25_M2_AZFC] = m->h_metric[GB25_M_AZC][a];
    out[GB25_M2_AZCF] = out[GB25_M2_AZFF] = m->h_metric[GB25_M_AZF][a];
    out[GB25_M2_FFF] = m->h_metric[GB25_M_FCOR][a];
    out[GB25_M2_PHICC] = m->h_metric[GB25_M_PHIC][a];
    if (lam_c) *lam_c = c.lon_west + (ig + 0.5) * dlam;
    // (the row tables were rounded to the float type when they were uploaded: the same value here)
    if (phi_c) *phi_c = (double)(real)m->h_metric[GB25_M_PHIC][a];
    return;
  }
  // computational coordinates of the four node families around the 0-based (ig, j)
  const double lf = lam0 + ig * dlam, lc = lam0 + (ig + 0.5) * dlam;
  const double pc = c.lat_south + j * dphi, pf = pc - 0.5 * dphi;
  auto N = [](double l, double p) { return tripolar_node(l, p); };
  const GNode cc = N(lc, pc), fc = N(lf, pc), cf = N(lc, pf), ff = N(lf, pf);
  const GNode fc_e = N(lf + dlam, pc), ff_e = N(lf + dlam, pf), cc_w = N(lc - dlam, pc), cf_w = N(lc - dlam, pf);
  const GNode cf_n = N(lc, pf + dphi), ff_n = N(lf, pf + dphi), ff_ne = N(lf + dlam, pf + dphi);
  const GNode cc_s = N(lc, pc - dphi), fc_s = N(lf, pc - dphi), cc_sw = N(lc - dlam, pc - dphi);
  const GNode cf_nw = N(lc - dlam, pf + dphi), fc_se = N(lf + dlam, pc - dphi);
  out[GB25_M2_DXCC] = gc_dist(fc, fc_e, R);
  out[GB25_M2_DXFC] = gc_dist(cc_w, cc, R);
  out[GB25_M2_DXCF] = gc_dist(ff, ff_e, R);
  out[GB25_M2_DXFF] = gc_dist(cf_w, cf, R);
  out[GB25_M2_DYCC] = gc_dist(cf, cf_n, R);
  out[GB25_M2_DYFC] = gc_dist(ff, ff_n, R);
  out[GB25_M2_DYCF] = gc_dist(cc_s, cc, R);
  out[GB25_M2_DYFF] = gc_dist(fc_s, fc, R);
  out[GB25_M2_AZCC] = quad_area(ff, ff_e, ff_ne, ff_n, R);
  out[GB25_M2_AZFC] = quad_area(cf_w, cf, cf_n, cf_nw, R);
  out[GB25_M2_AZCF] = quad_area(fc_s, fc_se, fc_e, fc, R);
  out[GB25_M2_AZFF] = quad_area(cc_sw, cc_s, cc, cc_w, R);
  out[GB25_M2_FFF] = 2.0 * c.Omega * std::sin(ff.phi * d2r);
  out[GB25_M2_PHICC] = cc.phi;
  if (lam_c) *lam_c = cc.lam;
  if (phi_c) *phi_c = cc.phi;
}
// ... at any row: beyond the pivot row of a folded grid (j >= Ny) the metric of a location is the metric of its image --
// cell rows and y-face rows mirror about the centres of row Ny-1, x faces as ig -> (Nx - ig) mod Nx (kernels.hpp, k_fill_fold).
// The images are COPIES of interior numbers (a host-supplied grid is treated the same way), so a metric and its image agree
// to the last bit whatever generated them.
void curv_metrics_at(const gb25_model* m, int ig, int j, double out[GB25_M2_COUNT], double* lam_c, double* phi_c) {
  const gb25_config& c = m->cfg;
  if (c.grid_type < GB25_GRID_TRIPOLAR || j < c.Ny) {
    curv_metrics_own(m, ig, j, out, lam_c, phi_c);
    return;
  }
  const int iw = ((ig % c.Nx) + c.Nx) % c.Nx, icc = c.Nx - 1 - iw, ifc = iw == 0 ? 0 : c.Nx - iw;
  const int jc = 2 * (c.Ny - 1) - j, jf = 2 * c.Ny - 1 - j;
  double A[GB25_M2_COUNT], B[GB25_M2_COUNT], C[GB25_M2_COUNT], D[GB25_M2_COUNT];
  curv_metrics_own(m, icc, jc, A, lam_c, phi_c);   // (c,c)
  curv_metrics_own(m, ifc, jc, B, nullptr, nullptr);   // (f,c)
  curv_metrics_own(m, icc, jf, C, nullptr, nullptr);   // (c,f)
  curv_metrics_own(m, ifc, jf, D, nullptr, nullptr);   // (f,f)
  out[GB25_M2_DXCC] = A[GB25_M2_DXCC]; out[GB25_M2_DYCC] = A[GB25_M2_DYCC]; out[GB25_M2_AZCC] = A[GB25_M2_AZCC]; out[GB25_M2_PHICC] = A[GB25_M2_PHICC];
  out[GB25_M2_DXFC] = B[GB25_M2_DXFC]; out[GB25_M2_DYFC] = B[GB25_M2_DYFC]; out[GB25_M2_AZFC] = B[GB25_M2_AZFC];
  out[GB25_M2_DXCF] = C[GB25_M2_DXCF]; out[GB25_M2_DYCF] = C[GB25_M2_DYCF]; out[GB25_M2_AZCF] = C[GB25_M2_AZCF];
  out[GB25_M2_DXFF] = D[GB25_M2_DXFF]; out[GB25_M2_DYFF] = D[GB25_M2_DYFF]; out[GB25_M2_AZFF] = D[GB25_M2_AZFF]; out[GB25_M2_FFF] = D[GB25_M2_FFF];
}

// Fills m->h_curv (the local slab's columns, halo columns by their own global index) and uploads what the kernels read
// (device_common.hpp, Curv).
gb25_status build_curv_grid(gb25_model* m) {
  const gb25_config& c = m->cfg;
  const int Nx = m->Nx, Ny = m->Ny, H = c.halo, sx = Nx + 2 * H, sy = Ny + 2 * H + 1;
  const size_t n2 = (size_t)sx * sy;
  // (the fold is the northern edge of the top row of ranks only)
  const bool tri = c.grid_type >= GB25_GRID_TRIPOLAR && !m->yn_open;
  for (auto& a : m->h_curv) a.assign(n2, 0.0);
  auto at = [&](int id) -> std::vector<double>& { return m->h_curv[id]; };
  parallel_rows(-H, Ny + H + 1, [&](int j) {
    for (int i = -H; i < Nx + H; i++) {
      const size_t o = (size_t)(i + H) + (size_t)sx * (j + H);
      double v[GB25_M2_COUNT];
      curv_metrics_at(m, i + m->rx * Nx, j + m->j0, v, nullptr, nullptr);
      // (numbers of the model's float type, as a host's grid holds them: reciprocals below are those of the rounded values, so
      // that what gb25_get_metric2 hands out, fed back through gb25_set_curvilinear_grid, is the same grid to the last bit)
      for (int q = 0; q < GB25_M2_COUNT; q++) at(q)[o] = (double)(real)v[q];
    }
  });
  Curv& cv = m->g.cv;
  gb25_status s;
  auto up = [&](const std::vector<double>& h, const real** out) { return upload_table(m, h, 0, out); };
  auto recip = [&](int id) {
    std::vector<double> r(n2);
    for (size_t q = 0; q < n2; q++) r[q] = at(id)[q] != 0.0 ? 1.0 / at(id)[q] : 0.0;
    return r;
  };
  if ((s = up(at(GB25_M2_DXFC), &cv.dxfc)) || (s = up(at(GB25_M2_DXCF), &cv.dxcf)) || (s = up(at(GB25_M2_DYFC), &cv.dyfc)) ||
      (s = up(at(GB25_M2_DYCF), &cv.dycf)) || (s = up(at(GB25_M2_AZCC), &cv.azcc)) || (s = up(at(GB25_M2_PHICC), &cv.phicc)) ||
      (s = up(recip(GB25_M2_DXFC), &cv.rdxfc)) || (s = up(recip(GB25_M2_DYCF), &cv.rdycf)) ||
      (s = up(recip(GB25_M2_AZCC), &cv.razcc)) || (s = up(recip(GB25_M2_AZFC), &cv.razfc)) ||
      (s = up(recip(GB25_M2_AZCF), &cv.razcf)) || (s = up(recip(GB25_M2_AZFF), &cv.razff)))
    return s;
  {
    // Coriolis parameter at the u and v points: the mean of the two (f,f) values either side, formed in the float type
    // from the rounded values exactly as the plain kernels form it from their row table
    std::vector<double> fu(n2, 0.0), fv(n2, 0.0);
    const std::vector<double>& F = at(GB25_M2_FFF);
    for (int j = -H; j < Ny + H; j++)
      for (int i = -H; i < Nx + H - 1; i++) {
        const size_t o = (size_t)(i + H) + (size_t)sx * (j + H);
        fu[o] = (double)(real(0.5) * ((real)F[o] + (real)F[o + sx]));
        fv[o] = (double)(real(0.5) * ((real)F[o] + (real)F[o + 1]));
      }
    if ((s = up(fu, &cv.fbar_u)) || (s = up(fv, &cv.fbar_v))) return s;
  }
  cv.on = 1;
  cv.north_fold = tri ? 1 : 0;
  return GB25_OK;
}
// The metrics of the split-explicit sub-cycle on the work arrays of a slab (WIDENED: columns [-W, Nx + W), pitch Nx + 2W) and
// of a folded grid (TALL: Wy more rows beyond the pivot row, whose metrics are those of their images; a single folded domain
// has W = 0).  Global columns are wrapped into [0, Nx_global): the sub-cycle of a single domain wraps its indices the same
// way, so that a decomposition reads the very numbers the single domain reads.
gb25_status build_curv_wide(gb25_model* m) {
  const gb25_config& c = m->cfg;
  const int Nx = m->Nx, Ny = m->Ny, H = c.halo, W = m->W, wsx = Nx + 2 * W, sy = Ny + 2 * H + 1 + m->Wy + m->Wys;
  auto wrap = [&](int ig) { return ((ig % c.Nx) + c.Nx) % c.Nx; };
  std::vector<real> t[5];
  for (auto& a : t) a.assign((size_t)wsx * sy, real(0.));
  parallel_rows(-H - m->Wys, Ny + H + m->Wy + 1, [&](int j) {
    for (int i = -W; i < Nx + W; i++) {
      const size_t o = (size_t)(i + W) + (size_t)wsx * (j + H + m->Wys);
      double v[GB25_M2_COUNT];
      curv_metrics_at(m, wrap(i + m->rx * Nx), j + m->j0, v, nullptr, nullptr);
      t[0][o] = (real)v[GB25_M2_DYFC];
      t[1][o] = (real)v[GB25_M2_DXCF];
      t[2][o] = (real)(1.0 / (double)(real)v[GB25_M2_AZCC]);
      t[3][o] = (real)(1.0 / (double)(real)v[GB25_M2_DXFC]);
      t[4][o] = (real)(1.0 / (double)(real)v[GB25_M2_DYCF]);
    }
  });
  for (int q = 0; q < 5; q++) {
    if (!m->d_wideM[q]) HIPCHK(hipMalloc(&m->d_wideM[q], t[q].size() * sizeof(real)));
    HIPCHK(hipMemcpy(m->d_wideM[q], t[q].data(), t[q].size() * sizeof(real), hipMemcpyHostToDevice));
  }
  return GB25_OK;
}

// TEOS-10 (Roquet et al. 2015) coefficient table R[i][j][k] of s^i t^j zeta^k and the reference profile r0(zeta),
// folded per model level: rho - rho0 = sum_{i+j<=6} C_ij(k) s^i t^j with
// C_ij(k) = sum_m R_ijm zeta_k^m  (+ r0(zeta_k) - rho0 on the constant term).
struct EosTerm { int i, j, k; double v; };
const EosTerm kEos[] = {
    {0,0,0, 8.0189615746e+02}, {1,0,0, 8.6672408165e+02}, {2,0,0,-1.7864682637e+03}, {3,0,0, 2.0375295546e+03},
    {4,0,0,-1.2849161071e+03}, {5,0,0, 4.3227585684e+02}, {6,0,0,-6.0579916612e+01}, {0,1,0, 2.6010145068e+01},
    {1,1,0,-6.5281885265e+01}, {2,1,0, 8.1770425108e+01}, {3,1,0,-5.6888046321e+01}, {4,1,0, 1.7681814114e+01},
    {5,1,0,-1.9193502195e+00}, {0,2,0,-3.7074170417e+01}, {1,2,0, 6.1548258127e+01}, {2,2,0,-6.0362551501e+01},
    {3,2,0, 2.9130021253e+01}, {4,2,0,-5.4723692739e+00}, {0,3,0, 2.1661789529e+01}, {1,3,0,-3.3449108469e+01},
    {2,3,0, 1.9717078466e+01}, {3,3,0,-3.1742946532e+00}, {0,4,0,-8.3627885467e+00}, {1,4,0, 1.1311538584e+01},
    {2,4,0,-5.3563304045e+00}, {0,5,0, 5.4048723791e-01}, {1,5,0, 4.8169980163e-01}, {0,6,0,-1.9083568888e-01},
    {0,0,1, 1.9681925209e+01}, {1,0,1,-4.2549998214e+01}, {2,0,1, 5.0774768218e+01}, {3,0,1,-3.0938076334e+01},
    {4,0,1, 6.6051753097e+00}, {0,1,1,-1.3336301113e+01}, {1,1,1,-4.4870114575e+00}, {2,1,1, 5.0042598061e+00},
    {3,1,1,-6.5399043664e-01}, {0,2,1, 6.7080479603e+00}, {1,2,1, 3.5063081279e+00}, {2,2,1,-1.8795372996e+00},
    {0,3,1,-2.4649669534e+00}, {1,3,1,-5.5077101279e-01}, {0,4,1, 5.5927935970e-01}, {0,0,2, 2.0660924175e+00},
    {1,0,2,-4.9527603989e+00}, {2,0,2, 2.5019633244e+00}, {0,1,2, 2.0564311499e+00}, {1,1,2,-2.1311365518e-01},
    {0,2,2,-1.2419983026e+00}, {0,0,3,-2.3342758797e-02}, {1,0,3,-1.8507636718e-02}, {0,1,3, 3.7969820455e-01}};
const double kEosR0[6] = {4.6494977072e+01, -5.2099962525e+00, 2.2601900708e-01,
                          6.4326772569e-02, 1.5616995503e-02, -1.7243708991e-03};

gb25_status build_eos_tables(gb25_model* m) {
  const int Nz = m->cfg.Nz, offk = m->metric_off_k;
  const std::vector<double>&zc = m->h_metric[GB25_M_ZC], &dzf = m->h_metric[GB25_M_DZF], &zf = m->h_metric[GB25_M_ZF];
  // [28 (Nz+1): levels][28 (Nz+1): faces][Nz+1: dzf]
  std::vector<double> tab((size_t)56 * (Nz + 1), 0.0), dz(Nz + 1);
  auto fold = [&](double Z, double* c) {
    const double zeta = -Z * 1e-4;
    for (const EosTerm& t : kEos) {
      int base = 0;
      for (int j = 0; j < t.j; j++) base += 7 - j;
      c[base + t.i] += t.v * std::pow(zeta, t.k);
    }
    double r0 = 0;
    for (int q = 5; q >= 0; q--) r0 = (r0 + kEosR0[q]) * zeta;
    c[0] += r0 - m->cfg.rho0;
  };
  for (int k = 0; k <= Nz; k++) {
    // geopotential height of level k; the halo level above the surface is mirrored (Oceananigans Z^ccc)
    fold((k < Nz) ? zc[offk + k] : zc[offk + Nz - 1] - dzf[offk + Nz - 1], &tab[(size_t)28 * k]);
    fold((double)(real)zf[offk + k], &tab[(size_t)28 * (Nz + 1 + k)]);   // (Z^ccf: the face itself, a number of the model's float type)
    dz[k] = dzf[offk + k];
  }
  double* d = nullptr;
  HIPCHK(hipMalloc(&d, (tab.size() + dz.size()) * sizeof(double)));
  HIPCHK(hipMemcpy(d, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + tab.size(), dz.data(), dz.size() * sizeof(double), hipMemcpyHostToDevice));
  m->dev_tables.push_back(reinterpret_cast<real*>(d));
  m->g.eos = d;
  m->g.eosf = d + (size_t)28 * (Nz + 1);
  m->g.dzf_d = d + tab.size();
  return GB25_OK;
}

// Split-explicit averaging weights (Oceananigans FixedSubstepNumber, restated): shape function
// (p=2, q=4, r=0.18927) sampled at tau = 2m/Ns, truncated like searchsortedlast(w, 0, rev=true).
void build_substeps(gb25_model* m) {
  const int N = m->cfg.substeps;
  std::vector<double> w(N + 1, 0.0);
  const double p = 2, q = 4, r = 0.18927, tau0 = (p + 2) * (p + q + 2) / (p + 1) / (p + q + 1);
  for (int k = 1; k <= N; k++) {
    double x = (2.0 * k / N) / tau0;
    w[k] = std::pow(x, p) * (1 - std::pow(x, q)) - r * x;
  }
  int lo = 0, hi = N + 1;
  while (lo < hi - 1) {
    int mid = lo + ((hi - lo) >> 1);
    if (w[mid] < 0.0) hi = mid; else lo = mid;
  }
  double s = 0;
  for (int k = 1; k <= lo; k++) s += w[k];
  m->Ns = lo;
  m->dtau_frac = 2.0 / N;
  m->weights.resize(lo);
  for (int k = 1; k <= lo; k++) m->weights[k - 1] = w[k] / s;
}

// GridFittedBottom(bottom_height) -> first active level per column -> the folded tables (device_common.hpp, Immersed).
// zb(i, j): bottom height at the centre of LOCAL column i (any i in [-E, Nx + E)), GLOBAL row j in [0, Ny_global).
// Restated from Oceananigans.ImmersedBoundaries [UPSTREAM-UNVERIFIED]; oracle/gb25_oracle.c states the same rules cell
// by cell (inactive_cell / stencil_active) and tests/test_gpu_immersed.py compares the two.
template <class ZB>
gb25_status build_bottom(gb25_model* m, ZB zb) {
  const gb25_config& c = m->cfg;
  const int Nx = m->Nx, Ny = m->Ny, Nz = c.Nz, H = c.halo, offk = m->metric_off_k, j0 = m->j0;
  const int E = std::max(H, m->W) + 4, ksx = Nx + 2 * E;
  // (rows: the own ones and, on a rank of a 2-D decomposition, the neighbours' rows its stencils and its widened sub-cycle reach)
  const int Ey = m->Ry > 1 ? H + std::max(m->Wy, m->Wys) + 6 : 0;
  const std::vector<double>&zc = m->h_metric[GB25_M_ZC], &zf = m->h_metric[GB25_M_ZF];
  m->kb_E = E;
  m->kb_Ey = Ey;
  m->kbot.assign((size_t)ksx * (Ny + 2 * Ey), 255);
  std::atomic<bool> any{false};
  auto level = [&](double b) {   // number of immersed cells of a column whose bottom is at height b
    int kb = 0;
    for (int k = 0; k < Nz; k++)
      if ((double)(real)zc[offk + k] <= b) kb = k + 1;   // z_center <= bottom: immersed (CenterImmersedCondition)
    return kb;
  };
  parallel_rows(-Ey, Ny + Ey, [&](int j) {
    if (j + j0 < 0 || j + j0 >= c.Ny) return;   // (beyond the walls / the fold: below)
    bool row_any = false;
    for (int i = -E; i < Nx + E; i++) {
      const int kb = level(zb(i, j + j0));
      m->kbot[(size_t)(i + E) + (size_t)ksx * (j + Ey)] = kb;
      row_any = row_any || (kb > 0 && j >= 0 && j < Ny);
    }
    if (row_any) any = true;
  });
  m->immersed = any.load();
  // level from which cell (i, j) is active; rows beyond the walls never are; rows beyond the zipper fold are the images
  // of the cells they mirror
  const bool nfold = m->g.cv.north_fold != 0;
  auto thr = [&](int i, int j) -> int {
    if (nfold && j >= Ny) {
      // the mirrored cell: GLOBAL column Nx_global - 1 - ig, usually another rank's -- the bottom is a function of the
      // global position, evaluated here (a slab sees its partner's bottom without any exchange)
      const int il = c.Nx - 1 - (i + m->rx * Nx) - m->rx * Nx, jm = 2 * (c.Ny - 1) - (j + j0);   // (the fold pivots on the centres of row Ny-1)
      return jm < 0 ? 255 : level(zb(il, jm));
    }
    if (j + j0 < 0 || j + j0 >= c.Ny) return 255;
    return m->kbot[(size_t)(std::min(std::max(i, -E), Nx + E - 1) + E) + (size_t)ksx * (j + Ey)];
  };
  auto node_x = [&](int q, int j) { return std::min(thr(q - 1, j), thr(q, j)); };   // face node: inactive when BOTH cells are
  auto node_y = [&](int i, int q) { return std::min(thr(i, q - 1), thr(i, q)); };
  auto depth = [&](int i, int j) -> double {   // static column depth: top face - materialised bottom
    const int jj = (nfold && j >= Ny) ? j : std::min(std::max(j + j0, 0), c.Ny - 1) - j0;
    const int kb = thr(i, jj);
    return (double)(real)zf[offk + Nz] - (double)(real)zf[offk + kb];
  };
  const int sx = Nx + 2 * H, sy = Ny + 2 * H + 1;
  std::vector<unsigned> A((size_t)sx * sy, 0), B(A.size(), 0), C(A.size(), 0), D(A.size(), 0);
  std::vector<real> Hf(A.size(), 0), Hc(A.size(), 0), rHf(A.size(), 0), rHc(A.size(), 0);
  // (rows: the own ones; with a neighbour rank on a side, that side's halo rows too -- the corrector runs there as well)
  parallel_rows(m->ys_open ? -H : 0, (m->yn_open ? Ny + H : Ny) + 1, [&](int j) {
    for (int i = -H; i < Nx + H; i++) {
      const size_t o = (size_t)(i + H) + (size_t)sx * (j + H);
      int KX5 = 0, KX3 = 0, KY5 = 0, KY3 = 0, KXC5 = 0, KXC3 = 0, KYC5 = 0, KYC3 = 0;
      for (int q = -3; q <= 2; q++) { KX5 = std::max(KX5, thr(i + q, j)); KY5 = std::max(KY5, thr(i, j + q)); }
      for (int q = -2; q <= 1; q++) { KX3 = std::max(KX3, thr(i + q, j)); KY3 = std::max(KY3, thr(i, j + q)); }
      for (int q = -2; q <= 3; q++) { KXC5 = std::max(KXC5, node_x(i + q, j)); KYC5 = std::max(KYC5, node_y(i, j + q)); }
      for (int q = -1; q <= 2; q++) { KXC3 = std::max(KXC3, node_x(i + q, j)); KYC3 = std::max(KYC3, node_y(i, j + q)); }
      const int kc = std::min(thr(i, j), Nz);   // (the extra face row j = Ny has no cells: its kc is never used)
      const int KPU = std::max(thr(i - 1, j), thr(i, j));
      // wall faces: the plain grid's business (the fold line is no wall)
      const int KPV = (j + j0 == 0 || (j + j0 >= c.Ny && !nfold)) ? 0 : std::min(std::max(thr(i, j - 1), thr(i, j)), 255);
      A[o] = (unsigned)kc | (unsigned)KX5 << 8 | (unsigned)KX3 << 16 | (unsigned)KY5 << 24;
      B[o] = (unsigned)KY3 | (unsigned)KXC5 << 8 | (unsigned)KXC3 << 16 | (unsigned)KYC5 << 24;
      C[o] = (unsigned)KYC3 | (unsigned)std::min(KPU, 255) << 8 | (unsigned)KPV << 16;
      int KX7 = 0, KY7 = 0;   // WENO(order = 7) tracer advection: the eight cells around the x face / the y face
      for (int q = -4; q <= 3; q++) { KX7 = std::max(KX7, thr(i + q, j)); KY7 = std::max(KY7, thr(i, j + q)); }
      D[o] = (unsigned)std::min(KX7, 255) | (unsigned)std::min(KY7, 255) << 8 |
             (unsigned)std::min(node_x(i, j), 255) << 16 | (unsigned)std::min(node_y(i, j), 255) << 24;
      const double hf = std::min(depth(i - 1, j), depth(i, j)), hc = std::min(depth(i, j - 1), depth(i, j));
      Hf[o] = (real)hf;
      Hc[o] = (real)hc;
      rHf[o] = hf > 0 ? (real)(1.0 / hf) : real(0.);
      // (the v face on the southern wall never moves; its correction divides by the full depth as on the plain grid)
      rHc[o] = j + j0 == 0 ? (real)(1.0 / ((double)(real)zf[offk + Nz] - (double)(real)zf[offk])) : (hc > 0 ? (real)(1.0 / hc) : real(0.));
    }
  });
  auto upload = [&](const void* h, size_t bytes, void** d) -> gb25_status {
    if (!*d) HIPCHK(hipMalloc(d, bytes));
    HIPCHK(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return GB25_OK;
  };
  gb25_status s;
  if ((s = upload(A.data(), A.size() * sizeof(unsigned), (void**)&m->d_ord[0]))) return s;
  if ((s = upload(B.data(), B.size() * sizeof(unsigned), (void**)&m->d_ord[1]))) return s;
  if ((s = upload(C.data(), C.size() * sizeof(unsigned), (void**)&m->d_ord[2]))) return s;
  if ((s = upload(D.data(), D.size() * sizeof(unsigned), (void**)&m->d_ord[3]))) return s;
  if ((s = upload(Hf.data(), Hf.size() * sizeof(real), (void**)&m->d_H[0]))) return s;
  if ((s = upload(Hc.data(), Hc.size() * sizeof(real), (void**)&m->d_H[1]))) return s;
  if ((s = upload(rHf.data(), rHf.size() * sizeof(real), (void**)&m->d_H[2]))) return s;
  if ((s = upload(rHc.data(), rHc.size() * sizeof(real), (void**)&m->d_H[3]))) return s;
  m->g.im.ordA = m->d_ord[0]; m->g.im.ordB = m->d_ord[1]; m->g.im.ordC = m->d_ord[2]; m->g.im.ordD = m->d_ord[3];
  m->g.im.Hfc = m->d_H[0]; m->g.im.Hcf = m->d_H[1]; m->g.im.rHfc = m->d_H[2]; m->g.im.rHcf = m->d_H[3];
  if (m->slab || nfold) {   // the same depths on the work arrays of the sub-cycle: widened slab / tall folded grid
    const int W = m->W, wsx = Nx + 2 * W;
    std::vector<real> wf((size_t)wsx * (sy + m->Wy + m->Wys), 0), wc(wf.size(), 0);
    for (int j = -m->Wys; j <= Ny + m->Wy; j++)
      for (int i = -W; i < Nx + W; i++) {
        const size_t o = (size_t)(i + W) + (size_t)wsx * (j + H + m->Wys);
        wf[o] = (real)std::min(depth(i - 1, j), depth(i, j));
        wc[o] = (real)std::min(depth(i, j - 1), depth(i, j));
      }
    if ((s = upload(wf.data(), wf.size() * sizeof(real), (void**)&m->d_wideH[0]))) return s;
    if ((s = upload(wc.data(), wc.size() * sizeof(real), (void**)&m->d_wideH[1]))) return s;
  }
  m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
  return GB25_OK;
}
// gaussian_islands(lambda, phi) = zb + h (mtn1 + mtn2), zb = z[1], h = -zb + 100; mountains 5 degrees wide at (70 E,
// 55 N) and (250 E, 55 N) (GB-25 src/model_utils.jl:67-80,138-140).  A function of the GLOBAL longitude: the halo
// columns of a slab see the neighbour's bottom without any exchange.
double gaussian_islands_bottom(const gb25_model* m, int i_local, int j) {
  const gb25_config& c = m->cfg;
  const double dlam = (c.lon_east - c.lon_west) / c.Nx, dphi = (c.lat_north - c.lat_south) / c.Ny;
  int ig = (i_local + m->rx * m->Nx) % c.Nx;
  if (ig < 0) ig += c.Nx;
  auto mtn = [](double l, double p, double l1, double p1) {
    const double d = 5;
    return std::exp(-((l - l1) * (l - l1) + (p - p1) * (p - p1)) / (2 * d * d));
  };
  const double z1 = -c.depth, h = -z1 + 100.0;
  if (m->g.cv.on) {
    // physical coordinates of the cell centre, the longitude brought next to each mountain (the tripolar grid starts AT
    // the first mountain's longitude: without this only its eastern half would exist)
    double lam, phi;
    if (c.grid_type >= GB25_GRID_TRIPOLAR) {   // (the centre node alone: this runs for every column of the bottom tables)
      const GNode cc = tripolar_node(70.0 + (ig + 0.5) * (360.0 / c.Nx), c.lat_south + j * ((90.0 - c.lat_south) / (c.Ny - 1)));
      lam = cc.lam;
      phi = cc.phi;
    } else {
      double v[GB25_M2_COUNT];
      curv_metrics_at(m, ig, j, v, &lam, &phi);
    }
    const double l1 = lam - 360.0 * std::floor((lam - 70.0 + 180.0) / 360.0), l2 = lam - 360.0 * std::floor((lam - 250.0 + 180.0) / 360.0);
    return z1 + h * (mtn(l1, phi, 70, 55) + mtn(l2, phi, 70 + 180, 55));
  }
  const double lam = c.lon_west + (ig + 0.5) * dlam;
  // (the centre latitude as the model's float type holds it: phi_c is a metric of the grid)
  const double phi = (double)(real)(c.lat_south + (j + 0.5) * dphi);
  return z1 + h * (mtn(lam, phi, 70, 55) + mtn(lam, phi, 70 + 180, 55));
}

gb25_status alloc_field(gb25_model* m, Field& F, int nx, int ny, int nz) {
  F.nx = nx; F.ny = ny; F.nz = nz;
  hipError_t e = hipMalloc(&F.d, F.elems() * sizeof(real));
  if (e != hipSuccess)
    return fail(m, GB25_ERR_OUT_OF_MEMORY, "hipMalloc of %zu bytes failed: %s", F.elems() * sizeof(real),
                hipGetErrorString(e));
  HIPCHK(hipMemset(F.d, 0, F.elems() * sizeof(real)));
  return GB25_OK;
}

inline int mom_kchunks(const gb25_model* m) { return std::max(1, m->g.Nz / m->mom_chunk_levels); }
inline dim3 grid2(int nx, int ny, dim3 b) { return dim3((nx + b.x - 1) / b.x, (ny + b.y - 1) / b.y); }

// sel: 3 = u, v, T, S;  1 = u, v;  2 = T, S
Halo3 halo3(gb25_model* m, int sel = 3) {
  Halo3 h{};
  int n = 0;
  if (sel & 1) {   // (xf, neg: how the field crosses the zipper fold -- on x faces, changing sign)
    h.p[n] = m->f[GB25_U].d; h.xf[n] = 1; h.neg[n] = 1; h.is_v[n++] = 0;
    h.p[n] = m->f[GB25_V].d; h.neg[n] = 1; h.is_v[n++] = 1;
  }
  if (sel & 2) {
    h.p[n] = m->f[GB25_T].d; h.is_v[n++] = 0;
    h.p[n] = m->f[GB25_S].d; h.is_v[n++] = 0;
  }
  if (sel & 4) h.p[n] = m->f[GB25_E].d, h.is_v[n++] = 0;   // (the TKE tracer of CATKE: a launch of its own, sel = 4)
  h.n = n;
  return h;
}
Halo2 halo2_prognostic(gb25_model* m) {
  Halo2 h{};
  h.p[0] = m->f[GB25_ETA].d; h.is_v[0] = 0;
  h.p[1] = m->f[GB25_BT_U].d; h.is_v[1] = 0; h.xf[1] = 1; h.neg[1] = 1;
  h.p[2] = m->f[GB25_BT_V].d; h.is_v[2] = 1; h.neg[2] = 1;
  h.n = 3;
  return h;
}
Halo2 halo2_G(gb25_model* m) {   // the barotropic forcing G.U, G.V
  Halo2 h{};
  h.p[0] = m->f[GB25_GN_BT_U].d; h.is_v[0] = 0; h.xf[0] = 1; h.neg[0] = 1;
  h.p[1] = m->f[GB25_GN_BT_V].d; h.is_v[1] = 1; h.neg[1] = 1;
  h.n = 2;
  return h;
}
// the single-domain producers write the halo cells derived from their output themselves (option FOLD_FILLS) -- not across
// the zipper fold, whose images live in other threads' columns, and not with a closure (the implicit solve comes after
// the look-aheads' writes)
inline bool producers_fold(const gb25_model* m) {
  return !m->slab && m->fold_fills && !m->g.cv.north_fold && m->nu == 0 && m->kappa == 0 && !m->catke && m->tracer_order == 5;
}
// rows of y faces a launch covers (face 0 is the southern wall; the faces beyond the last row of cells are a wall or, on a
// folded grid, halo cells)
inline int v_rows(const Grid& g) { return g.Ny; }

// y/z boundary layers (always local) and, for a single slab, the periodic x copy.
// extended: also treat the x-halo columns (slab mode, after the neighbours' columns were unpacked).
// which: 3 = 3-D and 2-D fields, 1 = the 3-D bundle only, 2 = the 2-D fields only (slab pipeline).
// sel3: which 3-D fields (halo3); st: stream (nullptr = the model's stream).
gb25_status fill_halos_2d(gb25_model* m, Halo2 h2);
gb25_status fill_halos_impl(gb25_model* m, bool with_x, bool extended = false, int which = 3, int sel3 = 3,
                            hipStream_t st = nullptr, bool use_default_stream = true, const Halo2* h2_other = nullptr) {
  const Grid& g = m->g;
  if (use_default_stream) st = m->stream;
  Timed t(m, GB25_K_FILL_HALOS);
  Halo3 h3 = halo3(m, sel3);
  Halo2 h2 = h2_other ? *h2_other : halo2_prognostic(m);
  dim3 b(256);
  const int i0 = extended ? -g.H : 0, ni = extended ? g.Nx + 2 * g.H : g.Nx;
  if (which == 2 && with_x && g.x_periodic && !extended && !h2_other) return fill_halos_2d(m, halo2_prognostic(m));
  if (g.cv.north_fold && !m->slab) {   // y / z layers, the rows beyond the fold, then the periodic x copy over all of them
    if (which == 2) return fill_halos_2d(m, h2);
    if (which == 1) h2.n = 0;
    hipLaunchKernelGGL(k_fill_yz, dim3((g.Nx + 255) / 256, g.Nz + 1 + g.Ny), b, 0, st, g, h3, h2, 0, g.Nx, 0);
    hipLaunchKernelGGL(k_fill_fold, dim3((g.Nx + 255) / 256, g.H + g.cv.pivot_slaved, g.Nz + 2 + (h2.n ? 1 : 0)), b, 0, st, g, h3, h2);
    const int rows_c = g.sy_c * (g.Nz + 2 * g.H), rows_v = g.sy_v * (g.Nz + 2 * g.H);
    hipLaunchKernelGGL(k_fill_x, dim3((unsigned)(((long)rows_v * 2 * g.H + 255) / 256), 4 + h2.n), b, 0, st, g, h3, h2,
                       rows_c, rows_v);
    LAUNCHCHK();
    return GB25_OK;
  }
  if (which == 2) {
    Grid g2 = g;
    g2.Nz = 0;   // k_fill_y then runs its 2-D branch only
    hipLaunchKernelGGL(k_fill_y, dim3((ni + 255) / 256, 1), b, 0, st, g2, h3, h2, i0, ni);
    LAUNCHCHK();
    return GB25_OK;
  }
  if (which == 1) h2.n = 0;
  if (m->fill_fused && with_x && g.x_periodic && !extended) {   // single slab: y, z and periodic x in one launch
    const int nbx = (g.Nx + 255) / 256, nb_yz = nbx * (g.Nz + 1 + g.Ny);
    const int rows_c = g.sy_c * (g.Nz + 2 * g.H), rows_v = g.sy_v * (g.Nz + 2 * g.H);
    const int nbr = (int)(((long)rows_v * 2 * g.H + 255) / 256);
    hipLaunchKernelGGL(k_fill_fused, dim3(nb_yz + nbr * (4 + h2.n)), b, 0, st, g, h3, h2, nbx, nb_yz, nbr, rows_c,
                       rows_v);
    LAUNCHCHK();
    return GB25_OK;
  }
  // (z layers: the own rows; `extended` on a rank of a 2-D decomposition: the halo rows of its open sides too -- they arrived
  // with the neighbours' rows, interior levels only)
  const int jlo = (extended && m->ys_open) ? -g.H : 0, jhi = (extended && m->yn_open) ? g.Ny + g.H : g.Ny;
  hipLaunchKernelGGL(k_fill_yz, dim3((ni + 255) / 256, g.Nz + 1 + (jhi - jlo)), b, 0, st, g, h3, h2, i0, ni, jlo);
  if (with_x && g.x_periodic) {
    int rows_c = g.sy_c * (g.Nz + 2 * g.H), rows_v = g.sy_v * (g.Nz + 2 * g.H);
    long threads = (long)rows_v * 2 * g.H;
    hipLaunchKernelGGL(k_fill_x, dim3((unsigned)((threads + 255) / 256), 4 + h2.n), b, 0, st, g, h3, h2, rows_c,
                       rows_v);
  }
  LAUNCHCHK();
  return GB25_OK;
}

gb25_status fill_halos_2d(gb25_model* m, Halo2 h2) {
  const Grid& g = m->g;
  Halo3 none{};
  dim3 b(256);
  // Nz = 0 makes k_fill_y take its 2-D branch for blockIdx.y == 0; k_fill_x's 3-D slices
  // (blockIdx.y < 4) see zero rows and fall through.
  Grid g2 = g;
  g2.Nz = 0;
  hipLaunchKernelGGL(k_fill_y, dim3((g.Nx + 255) / 256, 1), b, 0, m->stream, g2, none, h2, 0, g.Nx);
  if (g.cv.north_fold && !m->slab)   // (a slab's rows beyond the fold come from its partner rank: slab_step.hpp)
    hipLaunchKernelGGL(k_fill_fold, dim3((g.Nx + 255) / 256, g.H + g.cv.pivot_slaved, 1), b, 0, m->stream, g, none, h2);
  if (g.x_periodic) {
    long threads = (long)g.sy_v * 2 * g.H;
    hipLaunchKernelGGL(k_fill_x, dim3((unsigned)((threads + 255) / 256), 4 + h2.n), b, 0, m->stream, g2, none, h2, 0,
                       0);
  }
  LAUNCHCHK();
  return GB25_OK;
}

// part: 0 = the whole extended range -H+1 .. Nx+H-2; 1 = the slab's own columns 0 .. Nx-2 (they read own columns
// only: runs while the x-halo bundle travels); 2 = the two edge strips that part 1 left out.
gb25_status compute_w_impl(gb25_model* m, int part = 0) {
  const Grid& g = m->g;
  Timed t(m, GB25_K_COMPUTE_W);
  int ia = -g.H + 1, na = g.Nx + 2 * g.H - 2, ib = 0, nbcols = 0;
  if (part == 1) {
    ia = 0; na = g.Nx - 1;
  } else if (part == 2) {
    ia = -g.H + 1; na = g.H - 1; ib = g.Nx - 1; nbcols = g.H;
  }
  const int ey = g.Ny + 2 * g.H - 2;
  // narrow strips: 16 columns x 16 rows per block instead of 64 x 4 (a 64-wide block would be three-quarters empty)
  dim3 b = (na + nbcols) >= 64 ? dim3(64, 4) : dim3(16, 16);
  const LazyCorr lz{m->corr[0].d, m->corr[1].d, m->wbase, m->g.sx * m->g.sy_v};
  auto kw = k_compute_w<false, false>;
  if (m->uv_lazy) kw = k_compute_w<false, true>;
  else if (g.cv.on) kw = k_compute_w<true, false>;
  hipLaunchKernelGGL(kw, grid2(na + nbcols, ey, b), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d, m->f[GB25_W].d, ia,
                     na, ib, nbcols, lz);
  LAUNCHCHK();
  return GB25_OK;
}
// Hydrostatic pressure on columns [i_first, i_last] (default: the whole extended range -H+1 .. Nx+H-2; column
// i_first - 1 is read as the west neighbour of the first x difference), optionally on a second range as well.
gb25_status compute_p_impl(gb25_model* m, int i_first = INT_MIN, int i_last = INT_MIN, int i_first_b = 0,
                           int i_last_b = -1, bool may_skip_p = false, int j_first = INT_MIN, int j_last = INT_MIN) {
  const Grid& g = m->g;
  const real *Tsrc = m->f[GB25_T].d, *Ssrc = m->f[GB25_S].d;
  real *dpx_out = m->dpx.d, *dpy_out = m->dpy.d;
  if (m->pressure_bits == 32) {
    // the model float type's own arithmetic, whole extended range, pHY' stored, differences from the stored values
    Timed t(m, GB25_K_COMPUTE_P);
    dim3 b(64, 4);
    const int nx = g.Nx + 2 * g.H - 2, ny = g.Ny + 2 * g.H - 2;
    hipLaunchKernelGGL(k_compute_p_literal, grid2(nx, ny, b), b, 0, m->stream, g, m->f[GB25_T].d, m->f[GB25_S].d,
                       m->f[GB25_PHY].d, -g.H + 1, nx, 0, 0);
    const long n = (long)m->f[GB25_PHY].elems();
    hipLaunchKernelGGL(k_pressure_differences, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream, g,
                       m->f[GB25_PHY].d, m->dpx.d, m->dpy.d, n);
    m->phy_stale = false;
    m->n2_fresh = false;
    LAUNCHCHK();
    return GB25_OK;
  }
  real* n2 = nullptr;   // (CATKE's N^2 is SeawaterBuoyancy's dz_b -- alpha dzT - beta dzS -- not a difference of buoyancies: k_catke_n2)
  m->n2_fresh = false;
  const bool write_p = !may_skip_p || m->phy_pinned;
  m->phy_stale = !write_p;
  if (i_first == INT_MIN) {
    i_first = -g.H + 1;
    i_last = g.Nx + g.H - 2;
  }
  Timed t(m, GB25_K_COMPUTE_P);
  dim3 b(64, 4);
  const int ncol = i_last - i_first + 2;   // written columns + the helper column
  const int ncol_b = i_last_b >= i_first_b ? i_last_b - i_first_b + 2 : 0;
  if (j_first == INT_MIN) {                // rows -H+1 .. Ny+H-2
    j_first = -g.H + 1;
    j_last = g.Ny + g.H - 2;
  }
  const int nrow = j_last - j_first + 1;
  const int tiles_a = (ncol + 62) / 63, tiles_b = (ncol_b + 62) / 63;
  // strips and narrow slabs: with four rows per thread a 180-column slab is 138 blocks on 256 CUs and the fp64 chains run at
  // their latency (0.14 ms, against 0.31 ms for 1440 columns): one row per thread (4x the waves, 2.03 evaluations per cell).
  // The narrowest of them on the lat-lon grid take the tile form: a wave = 16 columns x 4 rows, one evaluation per thread and
  // level, 15 x 3 written cells (a block of four waves: 12 rows) -- 5.6x the waves of the four-row form, 1.42 evaluations per
  // cell, rows of 64 B.  Same box, one rank alone: 180 columns 0.497 -> 0.484 ms per step, 360: 0.808 -> 0.794, the 4 x 2 mesh
  // rank 0.518 -> 0.510; but 720 columns 1.358 -> 1.401 and the folded grid's slabs 0.614 -> 0.617 (180) and 0.978 -> 1.018 (360)
  // beside their heavier curvilinear neighbours: those keep the one-row form.
  const bool narrow = (tiles_a + tiles_b) * ((nrow + PR * 4 - 1) / (PR * 4)) < 1024;
  if (narrow && !g.cv.on && ncol + ncol_b <= 400) {
    const int ta = (i_last - i_first + 1 + 14) / 15, tb = ncol_b ? (i_last_b - i_first_b + 1 + 14) / 15 : 0;
    dim3 gr(ta + tb, (nrow + 11) / 12);
    auto kern = write_p ? k_compute_p_tile<true> : k_compute_p_tile<false>;
    hipLaunchKernelGGL(kern, gr, b, 0, m->stream, g, Tsrc, Ssrc, m->f[GB25_PHY].d, dpx_out, dpy_out, i_first, i_last,
                       i_first_b, i_last_b, ta, n2, j_first, j_last);
  } else if (narrow) {
    dim3 gr(tiles_a + tiles_b, (nrow + 3) / 4);
    auto kern = write_p ? k_compute_p<1, true> : k_compute_p<1, false>;
    hipLaunchKernelGGL(kern, gr, b, 0, m->stream, g, Tsrc, Ssrc, m->f[GB25_PHY].d, dpx_out, dpy_out, i_first, i_last,
                       i_first_b, i_last_b, tiles_a, n2, j_first, j_last);
  } else {
    dim3 gr(tiles_a + tiles_b, (nrow + PR * 4 - 1) / (PR * 4));
    auto kern = write_p ? k_compute_p<PR, true> : k_compute_p<PR, false>;
    hipLaunchKernelGGL(kern, gr, b, 0, m->stream, g, Tsrc, Ssrc, m->f[GB25_PHY].d, dpx_out, dpy_out, i_first, i_last,
                       i_first_b, i_last_b, tiles_a, n2, j_first, j_last);
  }
  LAUNCHCHK();
  return GB25_OK;
}

void tile_grid(const Grid& g, int* nbx, int* nb) {
  *nbx = (g.Nx + TX - 1) / TX;
  int nby = (g.Ny + TY - 1) / TY;
  *nb = *nbx * nby * g.Nz;
}

// Tile columns of the momentum kernel that read the slab's own columns only (u, v up to 3 columns, w up to 2 columns
// away): [1, bx_hi).  Empty when the slab is too narrow.
inline int interior_tile_columns_end(const Grid& g) {
  const int nbx = (g.Nx + V2_TX - 1) / V2_TX;
  return std::max(1, std::min(nbx - 1, (g.Nx - 3) / V2_TX));
}
inline bool tendencies_split(const gb25_model* m) {
  return m->slab && m->split_tendencies && m->two_streams && m->pressure_bits == 64 && m->kernel_gen >= 2 &&
         interior_tile_columns_end(m->g) > 1 && !m->g.cv.north_fold && m->Ry == 1;   // (the rows beyond a fold / of a neighbour in y arrive last)
}

// part: 0 = every tile column; 1 = the interior tile columns (a12: launched before the x-halo bundle has arrived);
//       2 = the edge tile columns, then whatever follows the complete evaluation (the look-ahead's finish kernel).
gb25_status materialize_prev_uv(gb25_model* m);
gb25_status momentum_impl(gb25_model* m, int part = 0) {
  const Grid& g = m->g;
  int nbx, nb;
  m->tend_forkable = false;
  if (part != 2) m->ahead_uv_valid = m->ahead_baro_valid = false;   // look-aheads made from the previous tendencies are void
  if (m->kernel_gen >= 2) {
    Timed t(m, GB25_K_GU);   // the fused G_u + G_v kernel is accounted under the "gu" timer
    nbx = (g.Nx + V2_TX - 1) / V2_TX;
    constexpr int TYm = 4;   // rows (= waves) per block: 4 blocks per CU cover each other's barriers
    const int nby = (v_rows(g) + TYm - 1) / TYm;   // (zipper fold: the y faces on the fold line have a tendency too)
    const int kchunks = mom_kchunks(m);
    TileCols tc{nbx, nbx, 0, 0};
    if (part) {
      const int hi = interior_tile_columns_end(g);
      if (part == 1) tc = TileCols{hi - 1, hi - 1, 1, 0};
      else tc = TileCols{1 + nbx - hi, 1, 0, hi};
    }
    tc.cr = ChunkRange{0, kchunks, 0};
    nb = tc.n * nby * kchunks;
    // waves/SIMD the register allocator is held to: 4 (128 VGPRs) in fp32; fp64 operands are register pairs,
    // so the Float64 build asks for 2 (256 VGPRs) instead of spilling
    constexpr int MW = sizeof(real) == 8 ? 2 : 4;
    const bool ahead = m->ab2_ahead == 1 && !m->ptr_exposed;
    UvAhead nx{};
    const real dt = (real)m->last_dt, chi = (real)m->cfg.chi;
    if (ahead && m->prev_uv_src == 2)   // (the partner buffers still hold CATKE's previous velocities: a caller of the single phases)
      if (gb25_status s_ = materialize_prev_uv(m)) return s_;
    if (ahead) {   // predicted parameters of the next ab2_step!: the clock's dt and the model's chi
      nx.GmU = m->f[GB25_GM_U].d; nx.GmV = m->f[GB25_GM_V].d;
      nx.un = m->ahead_uv[0].d; nx.vn = m->ahead_uv[1].d;
      nx.P = m->uv_partials;
      nx.dt = dt; nx.C1 = real(1.5) + chi; nx.C2 = real(0.5) + chi;
      nx.plane2 = g.sx * g.sy_v;
      nx.fold = (producers_fold(m) && !m->immersed) ? 1 : 0;   // (with a bottom the corrector's own halo writes do it)
    }
    if (m->bottom_drag != 0) {
      // the bottom flux boundary condition of u, v from the (corrected) velocities this evaluation sees.  A slab's
      // interior pass runs before the x halos arrive: face 0, the one that reads v of the west halo column, waits for part 2
      const int i_first = part == 1 ? 1 : 0, n = part == 2 ? 1 : g.Nx - i_first;
      dim3 b(64, 4);
      hipLaunchKernelGGL(m->immersed ? k_bottom_drag_flux<true> : k_bottom_drag_flux<false>, grid2(n, v_rows(g), b), b, 0,
                         m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d, m->d_bottom_flux[0], m->d_bottom_flux[1],
                         (real)m->bottom_drag, i_first, n);
    }
    const LazyCorr lz{m->corr[0].d, m->corr[1].d, m->wbase, m->g.sx * m->g.sy_v};
    // (single domain: the kernel also writes the halo images of the next u, v; a slab's come with the next bundle)
    if (m->uv_lazy && !(ahead && (m->slab || (nx.fold && part == 0))))
      return fail(m, GB25_ERR_STATE, "internal: uncorrected velocities in a step whose momentum kernel cannot correct them");
    const bool drag = m->bottom_drag != 0;
    const bool fly_sweep = ahead && m->w_fly_now && part == 0 && !m->uv_lazy;   // (w on the fly beside the corrector's sweep)
    auto k5 = (drag && fly_sweep) ? (g.cv.on ? k_momentum_tendencies_v5<MW, TYm, true, true, true, false, true, true>
                                     : m->immersed ? k_momentum_tendencies_v5<MW, TYm, true, true, false, false, true, true>
                                                   : k_momentum_tendencies_v5<MW, TYm, true, false, false, false, true, true>)
              : drag ? (g.cv.on ? (ahead ? k_momentum_tendencies_v5<MW, TYm, true, true, true, false, true> : k_momentum_tendencies_v5<MW, TYm, false, true, true, false, true>)
                      : m->immersed ? (ahead ? k_momentum_tendencies_v5<MW, TYm, true, true, false, false, true> : k_momentum_tendencies_v5<MW, TYm, false, true, false, false, true>)
                                    : (ahead ? k_momentum_tendencies_v5<MW, TYm, true, false, false, false, true> : k_momentum_tendencies_v5<MW, TYm, false, false, false, false, true>))
              : (m->uv_lazy && ahead && m->w_fly_now) ? k_momentum_tendencies_v5<MW, TYm, true, false, false, true, false, true>
              : (m->uv_lazy && ahead) ? k_momentum_tendencies_v5<MW, TYm, true, false, false, true>
              // (w on the fly beside the corrector's sweep: memory holds the corrected velocities, only w is not read)
              : (ahead && m->w_fly_now && part == 0) ? (g.cv.on ? k_momentum_tendencies_v5<MW, TYm, true, true, true, false, false, true>
                                                        : m->immersed ? k_momentum_tendencies_v5<MW, TYm, true, true, false, false, false, true>
                                                                      : k_momentum_tendencies_v5<MW, TYm, true, false, false, false, false, true>)
              : g.cv.on ? (ahead ? k_momentum_tendencies_v5<MW, TYm, true, true, true> : k_momentum_tendencies_v5<MW, TYm, false, true, true>)
              : m->immersed ? (ahead ? k_momentum_tendencies_v5<MW, TYm, true, true> : k_momentum_tendencies_v5<MW, TYm, false, true>)
                            : (ahead ? k_momentum_tendencies_v5<MW, TYm, true, false> : k_momentum_tendencies_v5<MW, TYm, false, false>);
    if (nb > 0) {
      // The kernel's per-lane offsets are 32-bit BYTE offsets from the array pointers.  An array beyond that reach (4.4 GB per
      // field for config 5 as one domain) takes several launches, each a run of chunks of levels whose planes -- stencil and
      // halo layers included -- lie within 2 GB of pointers rebased by `kofs` planes.  Normally: one launch, kofs = 0.
      const int klen = (g.Nz + kchunks - 1) / kchunks;
      const double plane_bytes = (double)g.pl_v * sizeof(real), reach = 2147483648.0;
      const bool small = (double)m->f[GB25_W].elems() * sizeof(real) < reach;
      for (int c0 = 0; c0 < kchunks;) {
        // the lowest plane a run starting with chunk c0 touches: three levels below its first one, or a bottom halo layer
        const int kofs = small ? 0 : std::max(0, c0 * klen + g.H - 4);
        int c1 = c0 + 1;   // (a run takes every following chunk whose highest plane -- window, w, top halo layer -- is within reach)
        while (c1 < kchunks && (std::min(g.Nz, (c1 + 1) * klen) + g.H + 5 - kofs) * plane_bytes < reach) c1++;
        tc.cr = ChunkRange{c0, c1 - c0, kofs};
        const long oc_ = (long)g.pl_c * kofs, ov_ = (long)g.pl_v * kofs;
        UvAhead nr = nx;
        if (ahead) { nr.GmU += oc_; nr.GmV += ov_; nr.un += oc_; nr.vn += ov_; }
        const int nbr = tc.n * nby * (c1 - c0);
        hipLaunchKernelGGL(k5, dim3(nbr), dim3(V2_TX, TYm), 0, m->stream, g, m->f[GB25_U].d + oc_, m->f[GB25_V].d + ov_,
                           m->f[GB25_W].d + oc_, m->dpx.d + oc_, m->dpy.d + oc_, m->f[GB25_GN_U].d + oc_, m->f[GB25_GN_V].d + ov_,
                           tc, kchunks, nbr, nr, lz);
        c0 = c1;
      }
    }
    t.stop();   // the timer covers the tendency kernel alone
    if (ahead && part != 1) {
      dim3 b(64, 4);
      hipLaunchKernelGGL(k_ab2_velocities_finish, grid2(g.Nx, v_rows(g), b), b, 0, m->stream, g, m->uv_partials, kchunks,
                         nx.plane2, m->ahead_G[0].d, m->ahead_G[1].d, m->ahead_colsum[0].d, m->ahead_colsum[1].d);
      m->ahead_uv_valid = true;
      m->ahead_uv_dt = dt;
      m->ahead_uv_chi = chi;
    }
    LAUNCHCHK();
    return GB25_OK;
  }
  if (part == 1) return GB25_OK;   // the direct-stencil kernels are not split: everything after the halos arrived
  if (m->immersed || g.cv.on) return fail(m, GB25_ERR_STATE, "the direct-stencil kernels (GB25_OPT_KERNELS = 1) know neither immersed boundaries nor curvilinear grids");
  if (g.top_flux[0] || g.top_flux[1] || g.bottom_flux[0]) return fail(m, GB25_ERR_STATE, "the direct-stencil kernels (GB25_OPT_KERNELS = 1) know no flux boundary conditions");
  tile_grid(g, &nbx, &nb);
  dim3 b(TX, TY);
  {
    Timed t(m, GB25_K_GU);
    hipLaunchKernelGGL(k_gu, dim3(nb), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d, m->f[GB25_W].d,
                       m->dpx.d, m->f[GB25_GN_U].d, nbx, nb);
  }
  {
    Timed t(m, GB25_K_GV);
    hipLaunchKernelGGL(k_gv, dim3(nb), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d, m->f[GB25_W].d,
                       m->dpy.d, m->f[GB25_GN_V].d, nbx, nb);
  }
  LAUNCHCHK();
  return GB25_OK;
}
gb25_status tracers_impl(gb25_model* m) {
  const Grid& g = m->g;
  int nbx, nb;
  m->tend_forkable = false;
  if (m->kernel_gen >= 2) {
    Timed t(m, GB25_K_TRACERS);
    nbx = (g.Nx + V3_OUT - 1) / V3_OUT;
    const int nby = (g.Ny + 3) / 4;
    const int kchunks = std::max(1, g.Nz / m->trc_chunk_levels);   // >= 12 levels per block: the z-carry start-up stays ~3 %
    nb = nbx * nby * kchunks;
    constexpr int TW = sizeof(real) == 8 ? 3 : 6;   // see MW in momentum_impl (Float32: held to 80 VGPRs, six waves per SIMD)
    const bool ahead = m->ab2_ahead && !m->ptr_exposed;
    Ab2Ahead nx{};
    if (ahead) {   // predicted parameters of the next ab2_step!: the clock's dt and the model's chi
      nx.GmT = m->f[GB25_GM_T].d; nx.GmS = m->f[GB25_GM_S].d;
      nx.Tn = m->ahead[0].d; nx.Sn = m->ahead[1].d;
      nx.dt = (real)m->last_dt;
      nx.C1 = real(1.5) + (real)m->cfg.chi; nx.C2 = real(0.5) + (real)m->cfg.chi;
    }
    const bool fold = ahead && producers_fold(m);
    const LazyCorr lz{m->corr[0].d, m->corr[1].d, m->wbase, m->g.sx * m->g.sy_v};
    if (m->uv_lazy && !(ahead && (fold || m->slab)))
      return fail(m, GB25_ERR_STATE, "internal: uncorrected velocities in a step whose tracer kernel cannot correct them");
    constexpr int TWL = sizeof(real) == 8 ? 3 : 6;   // (the headline instance: held to 80 VGPRs, six waves per SIMD)
    constexpr int TW7 = sizeof(real) == 8 ? 2 : 5;   // (the order-7 windows: 9 register pairs per direction; 90-96 VGPRs without a spill)
    auto kern = (m->tracer_order == 7 && ahead && m->w_fly_now)   // (w on the fly beside the corrector's sweep, WENO(order = 7))
                    ? (g.cv.on ? k_tracer_tendencies_v5<TW7, true, true, false, true, false, 7, true>
                       : m->immersed ? k_tracer_tendencies_v5<TW7, true, true, false, false, false, 7, true>
                                     : k_tracer_tendencies_v5<TW7, true, false, false, false, false, 7, true>)
                : m->tracer_order == 7
                    ? (g.cv.on ? (ahead ? k_tracer_tendencies_v5<TW7, true, true, false, true, false, 7> : k_tracer_tendencies_v5<TW7, false, true, false, true, false, 7>)
                       : m->immersed ? (ahead ? k_tracer_tendencies_v5<TW7, true, true, false, false, false, 7> : k_tracer_tendencies_v5<TW7, false, true, false, false, false, 7>)
                                     : (ahead ? k_tracer_tendencies_v5<TW7, true, false, false, false, false, 7> : k_tracer_tendencies_v5<TW7, false, false, false, false, false, 7>))
                : (m->uv_lazy && ahead && fold && m->w_fly_now) ? k_tracer_tendencies_v5<TWL, true, false, true, false, true, 5, true>
                : (m->uv_lazy && ahead && fold) ? k_tracer_tendencies_v5<TWL, true, false, true, false, true>
                : (m->uv_lazy && ahead && m->w_fly_now) ? k_tracer_tendencies_v5<TWL, true, false, false, false, true, 5, true>   // (slab)
                : (m->uv_lazy && ahead) ? k_tracer_tendencies_v5<TWL, true, false, false, false, true>
                // (the corrector through the tracer kernel: adds du, dv as it loads, writes the corrected u, v; w on the fly)
                : (ahead && m->w_fly_now && m->uv_corr_pending) ? (g.cv.on ? k_tracer_tendencies_v5<TW, true, true, false, true, true, 5, true, true>
                                                                            : k_tracer_tendencies_v5<TW, true, true, false, false, true, 5, true, true>)
                // (w on the fly beside the corrector's sweep)
                : (ahead && m->w_fly_now) ? (g.cv.on ? (fold ? k_tracer_tendencies_v5<TW, true, true, true, true, false, 5, true> : k_tracer_tendencies_v5<TW, true, true, false, true, false, 5, true>)
                                             : m->immersed ? (fold ? k_tracer_tendencies_v5<TW, true, true, true, false, false, 5, true> : k_tracer_tendencies_v5<TW, true, true, false, false, false, 5, true>)
                                                           : (fold ? k_tracer_tendencies_v5<TW, true, false, true, false, false, 5, true> : k_tracer_tendencies_v5<TW, true, false, false, false, false, 5, true>))
                : g.cv.on ? (ahead ? (fold ? k_tracer_tendencies_v5<TW, true, true, true, true> : k_tracer_tendencies_v5<TW, true, true, false, true>)
                                 : k_tracer_tendencies_v5<TW, false, true, false, true>)
                : m->immersed ? (ahead ? (fold ? k_tracer_tendencies_v5<TW, true, true, true> : k_tracer_tendencies_v5<TW, true, true, false>)
                                     : k_tracer_tendencies_v5<TW, false, true, false>)
                            : (ahead ? (fold ? k_tracer_tendencies_v5<TW, true, false, true> : k_tracer_tendencies_v5<TW, true, false, false>)
                                     : k_tracer_tendencies_v5<TW, false, false, false>);
    if (m->uv_corr_pending) {
      if (!(ahead && m->w_fly_now && m->tracer_order == 5 && m->uvc[0].d)) return fail(m, GB25_ERR_STATE, "internal: no tracer kernel instance writes the corrected velocities here");
      nx.uc = m->uvc[0].d;
      nx.vc = m->uvc[1].d;
    }
    m->ahead_ts_folded = fold && !m->uv_corr_pending;
    hipLaunchKernelGGL(kern, dim3(nb), dim3(64, 4), 0, m->stream, g, m->f[GB25_U].d,
                       m->f[GB25_V].d, m->f[GB25_W].d, m->f[GB25_T].d, m->f[GB25_S].d, m->f[GB25_GN_T].d,
                       m->f[GB25_GN_S].d, nbx, kchunks, nb, nx, lz);
    LAUNCHCHK();
    m->ahead_valid = ahead;
    m->ahead_dt = nx.dt;
    m->ahead_chi = (real)m->cfg.chi;
    if (m->uv_corr_pending) {
      // the arrays the kernel wrote ARE u, v from here on (the uncorrected ones become the scratch pair of the next step), and
      // their halo cells come from the ordinary fill -- y / z layers, the rows beyond a zipper fold, the periodic x copy
      m->uv_corr_pending = false;
      for (int q = 0; q < 2; q++) std::swap(m->f[GB25_U + q].d, m->uvc[q].d);
      if (gb25_status s = fill_halos_impl(m, true, false, 1, 1)) return s;
    }
    return GB25_OK;
  }
  m->ahead_valid = false;   // only the packed kernel looks ahead
  if (m->immersed || g.cv.on) return fail(m, GB25_ERR_STATE, "the direct-stencil kernels (GB25_OPT_KERNELS = 1) know neither immersed boundaries nor curvilinear grids");
  if (g.top_flux[2] || g.top_flux[3]) return fail(m, GB25_ERR_STATE, "the direct-stencil kernels (GB25_OPT_KERNELS = 1) know no flux boundary conditions");
  tile_grid(g, &nbx, &nb);
  Timed t(m, GB25_K_TRACERS);
  hipLaunchKernelGGL(k_tracer_tendencies, dim3(nb), dim3(TX, TY), 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                     m->f[GB25_W].d, m->f[GB25_T].d, m->f[GB25_S].d, m->f[GB25_GN_T].d, m->f[GB25_GN_S].d, nbx, nb);
  LAUNCHCHK();
  return GB25_OK;
}

gb25_status catke_implicit_impl(gb25_model* m, int mode, real dt, real chi = real(0.), int z0 = 0, int nz = -1);   // (with CATKE's diffusivity fields: below)
// implicit_step! of a pair of fields (kind 0: u, v with nu, the corrector's column integrals rewritten; 1: T, S with kappa)
gb25_status implicit_tables(gb25_model* m, int kind, double dt, double K) {
  if (m->d_implicit[kind] && m->implicit_key[kind][0] == dt && m->implicit_key[kind][1] == K) return GB25_OK;
  const int Nz = m->cfg.Nz, offk = m->metric_off_k;
  const std::vector<double>&dzc = m->h_metric[GB25_M_DZC], &dzf = m->h_metric[GB25_M_DZF];
  // (the spacings as the float type holds them; the chains in fp64, rounded once)
  auto Dc = [&](int k) { return (double)(real)dzc[offk + k]; };
  auto Df = [&](int k) { return (double)(real)dzf[offk + k]; };
  const double dK = (double)(real)dt * (double)(real)K;
  std::vector<real> tab((size_t)Nz + 2 * (size_t)Nz * Nz, real(0.));
  real *lo = tab.data(), *rb = lo + Nz, *gm = rb + (size_t)Nz * Nz;
  for (int k = 1; k < Nz; k++) lo[k] = (real)(-dK / (Dc(k) * Df(k)));
  for (int kf = 0; kf < Nz; kf++) {
    double bet = 1.0;
    for (int k = kf; k < Nz; k++) {
      const double l = k == kf ? 0.0 : -dK / (Dc(k) * Df(k)), u = k == Nz - 1 ? 0.0 : -dK / (Dc(k) * Df(k + 1));
      double gk = 0.0;
      if (k > kf) gk = (-dK / (Dc(k - 1) * Df(k))) / bet;
      bet = (1.0 - l - u) - l * gk;
      rb[(size_t)kf * Nz + k] = (real)(1.0 / bet);
      gm[(size_t)kf * Nz + k] = (real)gk;
    }
  }
  HIPCHK(hipStreamSynchronize(m->stream));   // (a solve with the old tables may still be running, on either stream)
  HIPCHK(hipStreamSynchronize(m->own_stream));
  HIPCHK(hipStreamSynchronize(m->side_stream));
  if (!m->d_implicit[kind]) HIPCHK(hipMalloc(&m->d_implicit[kind], tab.size() * sizeof(real)));
  HIPCHK(hipMemcpy(m->d_implicit[kind], tab.data(), tab.size() * sizeof(real), hipMemcpyHostToDevice));
  m->implicit_key[kind][0] = dt;
  m->implicit_key[kind][1] = K;
  return GB25_OK;
}
gb25_status implicit_vertical_impl(gb25_model* m, int kind, real dt) {
  const Grid& g = m->g;
  const real K = (real)(kind == 0 ? m->nu : m->kappa);
  if (K == real(0.)) return GB25_OK;
  Timed t_implicit(m, GB25_K_IMPLICIT);
  real *a = m->f[kind == 0 ? GB25_U : GB25_T].d, *b = m->f[kind == 0 ? GB25_V : GB25_S].d;
  real *sa = kind == 0 ? m->colsum[0].d : nullptr, *sb = kind == 0 ? m->colsum[1].d : nullptr;
  const int kchunks = mom_kchunks(m);
  if (g.Nz <= 128) {   // the column in registers, the elimination factors from tables
    if (gb25_status s = implicit_tables(m, kind, (double)dt, (double)K)) return s;
    ImplicitFields A{};
    const real* t = m->d_implicit[kind];
    for (int f = 0; f < 2; f++) {
      A.f[f] = f ? b : a;
      A.vshape[f] = kind == 0 && f == 1;
      A.first[f] = kind == 0 ? f : 2;
      A.lo[f] = t; A.rb[f] = t + g.Nz; A.gm[f] = t + g.Nz + (size_t)g.Nz * g.Nz;
      A.sum[f] = f ? sb : sa;
    }
    const bool imm = m->immersed;
    void (*kern)(Grid, ImplicitFields, int) =
        g.Nz <= 32 ? (imm ? k_implicit_vertical_reg<32, true> : k_implicit_vertical_reg<32, false>)
        : g.Nz <= 48 ? (imm ? k_implicit_vertical_reg<48, true> : k_implicit_vertical_reg<48, false>)
        : g.Nz <= 64 ? (imm ? k_implicit_vertical_reg<64, true> : k_implicit_vertical_reg<64, false>)
                     : (imm ? k_implicit_vertical_reg<128, true> : k_implicit_vertical_reg<128, false>);
    Timed tm(m, kind == 0 ? GB25_K_AB2_VELOCITIES : GB25_K_AB2_TRACERS);   // (accounted with the update it completes)
    dim3 b(64, 4);
    hipLaunchKernelGGL(kern, dim3((g.Nx + 63) / 64, ((kind == 0 ? v_rows(g) : g.Ny) + 3) / 4, 2), b, 0, m->stream, g, A, kchunks);
    LAUNCHCHK();
    if (kind == 0) m->colsum_valid = true;
    return GB25_OK;
  }
  // deeper columns: the column and the per-thread elimination factors in LDS
  int T = 256;
  while (T > 64 && (size_t)(2 * T + 2) * g.Nz * sizeof(real) > 64 * 1024) T /= 2;
  const size_t lds = (size_t)(2 * T + 2) * g.Nz * sizeof(real);
  if (lds > 160 * 1024) return fail(m, GB25_ERR_INVALID_ARGUMENT, "the implicit vertical solve keeps a column in LDS: Nz = %d is too deep", g.Nz);
  auto kern = m->immersed ? k_implicit_vertical<true> : k_implicit_vertical<false>;
  // (a function attribute is set per device: the flag lives in the model, not in the process)
  if (lds > 64 * 1024 && !m->implicit_lds_raised[m->immersed ? 1 : 0]) {
    HIPCHK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    m->implicit_lds_raised[m->immersed ? 1 : 0] = true;
  }
  Timed tm(m, kind == 0 ? GB25_K_AB2_VELOCITIES : GB25_K_AB2_TRACERS);
  hipLaunchKernelGGL(kern, dim3((g.Nx + T - 1) / T, kind == 0 ? v_rows(g) : g.Ny, 2), dim3(T), lds, m->stream, g, a, b, kind, K, K,
                     dt, sa, sb, kchunks);
  LAUNCHCHK();
  if (kind == 0) m->colsum_valid = true;
  return GB25_OK;
}

// previous_velocities into their own fields (before anything writes the buffer they live in, and before the host looks)
gb25_status materialize_prev_uv(gb25_model* m) {
  if (m->prev_uv_src == 0 || !m->catke) return GB25_OK;
  for (int q = 0; q < 2; q++) {
    const real* src = m->prev_uv_src == 1 ? m->f[GB25_U + q].d : m->ahead_uv[q].d;
    HIPCHK(hipMemcpyAsync(m->f[GB25_PREV_U + q].d, src, m->f[GB25_U + q].elems() * sizeof(real), hipMemcpyDeviceToDevice, m->stream));
  }
  m->prev_uv_src = 0;
  return GB25_OK;
}
gb25_status ab2_velocities_impl(gb25_model* m, real dt, real chi) {
  const Grid& g = m->g;
  if (m->ahead_uv_valid && dt == m->ahead_uv_dt && chi == m->ahead_uv_chi) {
    // the last momentum evaluation already advanced u and v with exactly these parameters: adopt its buffers
    // (closure = CATKE: the buffers given up hold the velocities of the last compute_diffusivities!: its previous_velocities)
    if (m->prev_uv_src == 2)
      if (gb25_status s_ = materialize_prev_uv(m)) return s_;
    if (m->prev_uv_src == 1) m->prev_uv_src = 2;
    for (int q = 0; q < 2; q++) {
      std::swap(m->f[GB25_U + q].d, m->ahead_uv[q].d);
      std::swap(m->f[GB25_GN_BT_U + q].d, m->ahead_G[q].d);
      std::swap(m->colsum[q].d, m->ahead_colsum[q].d);
    }
    m->colsum_valid = true;
    m->ahead_uv_valid = false;
    if (m->catke) return catke_implicit_impl(m, 0, dt);
    return implicit_vertical_impl(m, 0, dt);
  }
  m->ahead_uv_valid = false;
  if (m->prev_uv_src == 1)   // (u, v are about to be advanced in place)
    if (gb25_status s_ = materialize_prev_uv(m)) return s_;
  dim3 b(64, 4);
  Timed t(m, GB25_K_AB2_VELOCITIES);
  hipLaunchKernelGGL(k_ab2_velocities, grid2(g.Nx, v_rows(g), b), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                     m->f[GB25_GN_U].d, m->f[GB25_GM_U].d, m->f[GB25_GN_V].d, m->f[GB25_GM_V].d,
                     m->f[GB25_GN_BT_U].d, m->f[GB25_GN_BT_V].d, m->colsum[0].d, m->colsum[1].d, dt, chi,
                     mom_kchunks(m));   // the momentum kernel's chunking (momentum_impl)
  m->colsum_valid = true;
  LAUNCHCHK();
  if (m->catke) return catke_implicit_impl(m, 0, dt);
  return implicit_vertical_impl(m, 0, dt);
}
// the implicit solve of T, S (one elimination) with CATKE's kappa_c.  e is NOT stepped here: ab2_step! skips it, it is
// stepped inside compute_diffusivities! (catke_diffusivities_impl)
gb25_status catke_tracers_impl(gb25_model* m, real dt, real chi) { return catke_implicit_impl(m, 1, dt, chi); }
gb25_status ab2_tracers_impl(gb25_model* m, real dt, real chi) {
  const Grid& g = m->g;
  if (m->ahead_valid && dt == m->ahead_dt && chi == m->ahead_chi) {
    // the last tendency evaluation already advanced T and S with exactly these parameters
    std::swap(m->f[GB25_T].d, m->ahead[0].d);
    std::swap(m->f[GB25_S].d, m->ahead[1].d);
    m->ahead_valid = false;
    if (m->catke) return catke_tracers_impl(m, dt, chi);
    return implicit_vertical_impl(m, 1, dt);
  }
  m->ahead_valid = false;
  Timed t(m, GB25_K_AB2_TRACERS);
  const real C1 = real(1.5) + chi, C2 = real(0.5) + chi;
  size_t off = (size_t)g.H * g.pl_c;
  long n = (long)g.Nz * g.pl_c;
  real *T = m->f[GB25_T].d + off, *S = m->f[GB25_S].d + off;
  const real *a = m->f[GB25_GN_T].d + off, *bb = m->f[GB25_GM_T].d + off, *c = m->f[GB25_GN_S].d + off,
              *d = m->f[GB25_GM_S].d + off;
  bool aligned = (n % 4 == 0) && (((uintptr_t)T | (uintptr_t)S | (uintptr_t)a | (uintptr_t)bb | (uintptr_t)c |
                                   (uintptr_t)d) % sizeof(real4) == 0);
  if (aligned) {
    long n4 = n / 4;
    int blocks = (int)std::min<long>((n4 + 255) / 256, 256 * 16);
    // (nontemporal tendency reads: every byte of this stream is touched once per step)
    hipLaunchKernelGGL(k_ab2_tracers4<true>, dim3(blocks), dim3(256), 0, m->stream, (real4*)T, (real4*)S, (const real4*)a,
                       (const real4*)bb, (const real4*)c, (const real4*)d, n4, dt, C1, C2);
  } else {
    int blocks = (int)std::min<long>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_ab2_tracers1, dim3(blocks), dim3(256), 0, m->stream, T, S, a, bb, c, d, n, dt, C1, C2);
  }
  LAUNCHCHK();
  if (m->catke) return catke_tracers_impl(m, dt, chi);
  return implicit_vertical_impl(m, 1, dt);
}
gb25_status ab2_local_impl(gb25_model* m, real dt, real chi) {
  gb25_status s = ab2_velocities_impl(m, dt, chi);
  return s ? s : ab2_tracers_impl(m, dt, chi);
}

// step_free_surface!: Ns fused forward-backward substeps.  Single slab: canonical arrays, periodic x wrapped
// in-kernel.  Slab of a decomposition: wide-halo work arrays (halo W >= Ns filled once by the exchange of
// group 1), every substep computes on [-W+1, Nx+W-1) and the invalid rim never reaches the interior.
// ahead (single slab only): read eta, U, V where they are and G.U, G.V from the momentum look-ahead, write the new
// eta, U, V and the averages into the partner buffers.
// the image rows of the sub-cycle's tall work arrays (folded grid): pack the rows south of the pivot row / unpack into the rows
// beyond it.  A slab exchanges the buffer with its partner rank in between (slab_step.hpp, group 8); a single domain unpacks
// what it packed.
int64_t tall_buffer_elems(const gb25_model* m) { return (int64_t)5 * (m->Wy + 1) * (m->Nx + 2 * m->W); }
gb25_status tall_rows_impl(gb25_model* m, real* buf, bool pack) {
  const Grid& g = m->g;
  TallRows T{};
  for (int q = 0; q < 3; q++) T.p[q] = m->wide[0][q].d;
  T.p[3] = m->wideG[0].d;
  T.p[4] = m->wideG[1].d;
  T.sx = g.Nx + 2 * m->W; T.xo = m->W; T.yo = g.H + m->Wys; T.Wy = m->Wy; T.wrap = m->slab ? 0 : 1;
  const dim3 gr((T.sx + 255) / 256, m->Wy + 1, 5);
  if (pack) hipLaunchKernelGGL(k_tall_rows<true>, gr, dim3(256), 0, m->stream, g, T, buf, m->rx * g.Nx, m->cfg.Nx);
  else hipLaunchKernelGGL(k_tall_rows<false>, gr, dim3(256), 0, m->stream, g, T, buf, m->rx * g.Nx, m->cfg.Nx);
  LAUNCHCHK();
  return GB25_OK;
}

// own: (slab) the interior columns of eta, U, V, G.U, G.V still sit in the canonical arrays -- copied into the work arrays here,
// or read in place by the one-launch kernel; layers_done: set when the kernel also wrote the y layers of the new eta, U, V
gb25_status barotropic_impl(gb25_model* m, real dt, bool ahead = false, const InteriorCopies* own = nullptr,
                            bool* layers_done = nullptr) {
  const Grid& g = m->g;
  if (m->baro_inflight && !ahead) {
    // a look-ahead that is not being adopted (changed dt, ...) may still be running on the side stream, and it uses
    // the same scratch sets
    HIPCHK(hipStreamWaitEvent(m->stream, m->ev_baro, 0));
    m->baro_inflight = false;
  }
  Timed t(m, GB25_K_BAROTROPIC);
  m->last_baro_folded = false;
  // (option SUBSTEP_ORDER = 1 exists in the one-substep-per-launch kernel only: the temporally blocked ones keep the default order)
  const int bblock = m->substep_order ? 1 : m->baro_block;
  // work arrays: a slab's are widened in x (filled by the exchange of group 1 / 3 before this is called), a folded grid's are
  // tall (image rows beyond the pivot row: a slab's come from its partner before this is called, a single domain's below)
  const bool wide = m->slab || g.cv.north_fold;
  const real dtau = (real)m->dtau_frac * dt;
  dim3 b(64, 4);
  Baro bb;
  real *cur[3], *nxt[3], *other[3], *out[3];
  bb.jlo = -m->Wys;
  bb.jhi = g.Ny + m->Wy;
  bb.yo = g.H + m->Wys;
  bb.top_open = (g.cv.north_fold || m->yn_open) ? 1 : 0;
  if (!wide) {
    size_t nbar = m->f[GB25_ETA_BAR].elems() + m->f[GB25_U_BAR].elems() + m->f[GB25_V_BAR].elems();
    if (bblock <= 1)   // (the blocked kernels start their averages from zero themselves)
      HIPCHK(hipMemsetAsync(ahead ? m->bars_ahead : m->bars, 0, nbar * sizeof(real), m->stream));
    // the state the sub-cycle starts from is only read; the substeps alternate between two scratch sets
    for (int q = 0; q < 3; q++) {
      cur[q] = m->f[GB25_ETA + q].d; nxt[q] = m->pp[q].d; other[q] = m->pp2[q].d;
      out[q] = ahead ? m->ahead_eta[q].d : m->f[GB25_ETA + q].d;
    }
    const Field* bar = ahead ? m->ahead_bar : &m->f[GB25_ETA_BAR];
    bb.etab = bar[0].d; bb.Ub = bar[1].d; bb.Vb = bar[2].d;
    bb.GU = ahead ? m->ahead_G[0].d : m->f[GB25_GN_BT_U].d;
    bb.GV = ahead ? m->ahead_G[1].d : m->f[GB25_GN_BT_V].d;
    bb.sx = g.sx; bb.xo = g.H; bb.ilo = 0; bb.ihi = g.Nx; bb.wrap = 1;
    bb.Hfc = m->d_H[0]; bb.Hcf = m->d_H[1];
  } else {
    const int wsx = g.Nx + 2 * m->W;
    if (!m->slab) {
      // single folded domain: the state and the forcing into the work arrays, then the image rows beyond the pivot row
      InteriorCopies C{};
      int rmax = 0;
      for (int q = 0; q < 5; q++) {
        const Field& src = q < 3 ? m->f[GB25_ETA + q] : (ahead ? m->ahead_G[q - 3] : m->f[GB25_GN_BT_U + q - 3]);
        C.dst[q] = q < 3 ? m->wide[0][q].d : m->wideG[q - 3].d; C.dsx[q] = wsx; C.dxo[q] = m->W;
        C.src[q] = src.d; C.ssx[q] = g.sx; C.sxo[q] = g.H; C.rows[q] = src.ny;
        rmax = std::max(rmax, src.ny);
      }
      C.n = 5;
      hipLaunchKernelGGL(k_copy_interior_columns, dim3((g.Nx + 255) / 256, rmax, C.n), dim3(256), 0, m->stream, C, g.Nx);
      gb25_status s;
      if ((s = tall_rows_impl(m, m->tall_buf, true))) return s;
      if ((s = tall_rows_impl(m, m->tall_buf, false))) return s;
    }
    if (bblock <= 1)
      HIPCHK(hipMemsetAsync(m->wideBar[0].d, 0,
                            (m->wideBar[0].elems() + m->wideBar[1].elems() + m->wideBar[2].elems()) * sizeof(real),
                            m->stream));
    for (int q = 0; q < 3; q++) {
      cur[q] = m->wide[0][q].d; nxt[q] = m->wide[1][q].d; other[q] = m->wide[0][q].d;
      out[q] = ahead ? m->ahead_eta[q].d : m->f[GB25_ETA + q].d;
    }
    bb.etab = m->wideBar[0].d; bb.Ub = m->wideBar[1].d; bb.Vb = m->wideBar[2].d;
    bb.GU = m->wideG[0].d; bb.GV = m->wideG[1].d;
    bb.sx = wsx; bb.xo = m->W;
    if (m->slab) { bb.ilo = -m->W + 1; bb.ihi = g.Nx + m->W - 1; bb.wrap = 0; }
    else { bb.ilo = 0; bb.ihi = g.Nx; bb.wrap = 1; }
    bb.Hfc = m->d_wideH[0]; bb.Hcf = m->d_wideH[1];
  }
  const bool imm = m->immersed;
  bool finalize_after = false;
  // temporally blocked: the lat-lon kernel (row metrics) or its curvilinear sibling (per-point metrics); on canonical arrays
  // (single domain), widened slabs and the tall arrays of a folded grid alike
  const bool blocked_curv = bblock > 1 && g.cv.on;
  const bool blocked = bblock > 1;
  const CurvBaro cb = wide ? CurvBaro{m->d_wideM[0], m->d_wideM[1], m->d_wideM[2], m->d_wideM[3], m->d_wideM[4]}
                           : CurvBaro{g.cv.dyfc, g.cv.dxcf, g.cv.razcc, g.cv.rdxfc, g.cv.rdycf};
  const int rows = bb.jhi - bb.jlo + (bb.top_open ? 1 : 0);   // (no wall: the face row behind the last advanced row is carried along)
  auto fill_multi = [&](BaroMulti& bm, int s, int Sk) {
    bb.eta0 = cur[0]; bb.U0 = cur[1]; bb.V0 = cur[2];
    bb.eta1 = nxt[0]; bb.U1 = nxt[1]; bb.V1 = nxt[2];
    bm.b = bb;
    bm.ns = std::min(Sk, m->Ns - s);
    bm.first = s == 0;
    bm.last = s + Sk >= m->Ns;
    if (bm.first && bm.last && !wide && !ahead) {
      // a sub-cycle short enough for ONE launch would read and write the canonical eta, U, V in the same launch
      bm.last = 0;
      finalize_after = true;
    }
    bm.eta_out = out[0]; bm.U_out = out[1]; bm.V_out = out[2];
    bm.eb_out = bm.ub_out = bm.vb_out = nullptr;
    bm.fold = 0;
    bm.out_halo = m->slab ? g.H : 0;   // (a widened slab also writes the x halo columns of the new eta, U, V)
    bm.out_js = m->ys_open ? -g.H : 0;   // (... and a rank of a 2-D decomposition the halo rows of its open sides)
    bm.out_jn = m->yn_open ? g.Ny + g.H : g.Ny;
    for (int q = 0; q < 5; q++) bm.own_src[q] = nullptr;
    bm.layers = 0;
    if (wide) {
      const Field* fb = ahead ? m->ahead_bar : &m->f[GB25_ETA_BAR];
      bm.eb_out = fb[0].d; bm.ub_out = fb[1].d; bm.vb_out = fb[2].d;
    }
    for (int q = 0; q < BT_SMAX; q++) bm.w[q] = (s + q < m->Ns) ? (real)m->weights[s + q] : real(0.);
  };
  // a narrow slab: fewer tiles than the chip has CUs -- the whole sub-cycle in ONE launch (k_barotropic_whole)
  constexpr int NSW = 21;   // (the substeps of SplitExplicitFreeSurface(substeps = 30); other counts take the blocked launches)
  const int wcols = (bb.ihi - bb.ilo + BT_TX - 1) / BT_TX, wrows = bb.jhi - bb.jlo;
  // rows per tile: 17 (125 KB of LDS) or, when only that brings the tile count under the number of CUs, 24 (140 KB)
  const int wty = wcols * ((wrows + 16) / 17) <= m->n_cu ? 17 : 24;
  const int wtiles = wcols * ((wrows + wty - 1) / wty);
  // (Float32: the Float64 build would need twice the LDS)
  const bool whole = blocked && !g.cv.on && m->slab && m->baro_whole && m->Ns == NSW && wtiles <= m->n_cu && sizeof(real) == 4;
  // (the one-launch kernel reads the own columns in place only when it writes elsewhere -- the look-ahead's partner buffers;
  // inside its own step the results go into the very arrays other blocks are still loading their rings from)
  const bool in_place = whole && own && own->n == 5 && ahead;
  if (own && !in_place)
    hipLaunchKernelGGL(k_copy_interior_columns, dim3((g.Nx + 255) / 256, g.sy_v, own->n), dim3(256), 0, m->stream, *own, g.Nx);
  if (whole) {
    void (*kern)(Grid, BaroMulti, real) =
        wty == 17 ? (imm ? k_barotropic_whole<NSW, true, 17> : k_barotropic_whole<NSW, false, 17>)
                  : (imm ? k_barotropic_whole<NSW, true, 24> : k_barotropic_whole<NSW, false, 24>);
    const size_t lds = (size_t)5 * (BT_TX + 2 * NSW) * (wty + 2 * NSW) * sizeof(real);
    // (the attribute belongs to the DEVICE: kept per model, not per process -- a host driving several GPUs from one process, or
    // two threads, must not skip it on the second device)
    if (!m->whole_attr_set[imm ? 1 : 0][wty == 17 ? 0 : 1]) {
      HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      m->whole_attr_set[imm ? 1 : 0][wty == 17 ? 0 : 1] = true;
    }
    BaroMulti bm;
    fill_multi(bm, 0, NSW);
    if (in_place)
      for (int q = 0; q < 5; q++) bm.own_src[q] = own->src[q];
    if (own) {
      bm.layers = 1;
      if (layers_done) *layers_done = true;
    }
    dim3 gm(wcols, (wrows + wty - 1) / wty);
    hipLaunchKernelGGL(kern, gm, dim3(BW_NT), lds, m->stream, g, bm, dtau);
    LAUNCHCHK();
    return GB25_OK;
  }
  if (blocked_curv) {
    constexpr int Sk = 5, TYc = 17;
    dim3 gm((bb.ihi - bb.ilo + BT_TX - 1) / BT_TX, (rows + TYc - 1) / TYc);
    for (int s = 0; s < m->Ns; s += Sk) {
      BaroMulti bm;
      fill_multi(bm, s, Sk);
      hipLaunchKernelGGL((k_barotropic_multi_curv<Sk, TYc>), gm, dim3(BT_NT), 0, m->stream, g, bm, cb, dtau);
      for (int q = 0; q < 3; q++) { real* w_ = nxt[q]; nxt[q] = other[q]; other[q] = w_; cur[q] = w_; }
    }
  } else if (blocked) {
    // temporally blocked: S substeps per launch on (64 x TY) tiles
    const int S = std::min(bblock, (int)BT_SMAX);
    constexpr int TYb = 16, TY5 = 17;   // (5 substeps per launch: 64 x 17 tiles keep LDS under 40 KB -- four blocks per CU)
    const int tyb = (S > 3 && S <= 5) ? TY5 : TYb;
    dim3 gm((bb.ihi - bb.ilo + BT_TX - 1) / BT_TX, (bb.jhi - bb.jlo + tyb - 1) / tyb);
    void (*kern)(Grid, BaroMulti, real) =
        imm ? (S <= 3 ? k_barotropic_multi<3, TYb, true> : (S <= 5 ? k_barotropic_multi<5, TY5, true> : k_barotropic_multi<7, TYb, true>))
            : (S <= 3 ? k_barotropic_multi<3, TYb, false> : (S <= 5 ? k_barotropic_multi<5, TY5, false> : k_barotropic_multi<7, TYb, false>));
    const int Sk = S <= 3 ? 3 : (S <= 5 ? 5 : 7);
    for (int s = 0; s < m->Ns; s += Sk) {
      BaroMulti bm;
      fill_multi(bm, s, Sk);
      bm.fold = (!wide && producers_fold(m) && m->composite && bm.last) ? 1 : 0;
      if (bm.last) m->last_baro_folded = bm.fold != 0;
      hipLaunchKernelGGL(kern, gm, dim3(BT_NT), 0, m->stream, g, bm, dtau);
      for (int q = 0; q < 3; q++) { real* w_ = nxt[q]; nxt[q] = other[q]; other[q] = w_; cur[q] = w_; }
    }
  } else {
    dim3 gr = grid2(bb.ihi - bb.ilo, g.cv.on ? rows : bb.jhi - bb.jlo, b);
    for (int s = 0; s < m->Ns; s++) {
      bb.eta0 = cur[0]; bb.U0 = cur[1]; bb.V0 = cur[2];
      bb.eta1 = nxt[0]; bb.U1 = nxt[1]; bb.V1 = nxt[2];
      if (g.cv.on)
        hipLaunchKernelGGL(k_barotropic_substep_curv, gr, b, 0, m->stream, g, bb, cb, dtau, (real)m->weights[s]);
      else
        hipLaunchKernelGGL(m->substep_order ? (imm ? k_barotropic_substep<true, 1> : k_barotropic_substep<false, 1>)
                                            : (imm ? k_barotropic_substep<true> : k_barotropic_substep<false>),
                           gr, b, 0, m->stream, g, bb, dtau, (real)m->weights[s]);
      for (int q = 0; q < 3; q++) { real* w_ = nxt[q]; nxt[q] = other[q]; other[q] = w_; cur[q] = w_; }
    }
  }
  if (blocked && !finalize_after) {   // the last blocked launch wrote eta, U, V and published the averages
    LAUNCHCHK();
    return GB25_OK;
  }
  const int fjs = m->ys_open ? -g.H : 0, fjn = m->yn_open ? g.Ny + g.H : g.Ny;
  dim3 gi = grid2(g.Nx, fjn - fjs, b);
  if (m->slab) gi = grid2(g.Nx + 2 * g.H, fjn - fjs, b);   // (with the x halo columns: nothing is exchanged after the sub-cycle)
  hipLaunchKernelGGL(k_barotropic_finalize, gi, b, 0, m->stream, g, out[0], out[1], out[2], bb.etab, bb.Ub, bb.Vb,
                     bb.sx, bb.xo, wide ? bb.yo : g.H, m->slab ? g.H : 0, fjs, fjn);
  if (wide) {  // publish the averages in the canonical filtered-state arrays (compared by compare_states)
    InteriorCopies C{};
    int rmax = 0;
    for (int q = 0; q < 3; q++) {
      Field& dst = ahead ? m->ahead_bar[q] : m->f[GB25_ETA_BAR + q];
      C.dst[q] = dst.d; C.dsx[q] = dst.nx; C.dxo[q] = g.H;
      C.src[q] = m->wideBar[q].d + (size_t)m->Wys * bb.sx; C.ssx[q] = bb.sx; C.sxo[q] = bb.xo; C.rows[q] = dst.ny;
      rmax = std::max(rmax, dst.ny);
    }
    C.n = 3;
    hipLaunchKernelGGL(k_copy_interior_columns, dim3((g.Nx + 255) / 256, rmax, C.n), dim3(256), 0, m->stream, C, g.Nx);
  }
  LAUNCHCHK();
  return GB25_OK;
}

// use_colsum: only inside a composite time step, where nothing can have touched u, v since the AB2 kernel.
// part: 0 = every column this model corrects (a slab of a decomposition includes its x-halo columns);
//       1 = the slab's own columns only (needs no halo data: runs while the last exchanges are in flight);
//       2 = the x-halo columns only.  The G^n/G^- exchange happens once, with part 0 or 1.
gb25_status corrector_impl(gb25_model* m, bool use_colsum = false, int part = 0) {
  const Grid& g = m->g;
  {
    Timed t(m, GB25_K_CORRECTOR);
    dim3 b(64, 4);
    const bool ext = m->slab;
    int i0 = ext ? -g.H : 0, ni = ext ? g.Nx + 2 * g.H : g.Nx, skip_from = INT_MAX, skip = 0;
    if (part == 1) {
      i0 = 0;
      ni = g.Nx;
    } else if (part == 2) {
      if (!ext) return GB25_OK;
      i0 = -g.H; ni = 2 * g.H; skip_from = 0; skip = g.Nx;
    }
    const bool cs = use_colsum && m->colsum_valid && part != 2;
    const bool fold = producers_fold(m) && m->composite;
    // one thread per cell where the column integrals are at hand: the own columns of a slab, and its x-halo columns when
    // the integrals came with the 3-D bundle (group 0 carries the owner's; marching 16 columns x Ny threads up 48 levels
    // for them was 36 us of latency on the critical path of a 180-column rank)
    const bool cells_halo = m->slab && part == 2 && use_colsum && m->halo_colsum_valid;
    const bool cells = (m->slab && part == 1 && cs) || cells_halo;
    auto launch = [&](int i0_, int ni_, int skf, int sk, int jr0, int nj, int jskf, int jsk) {
      if (cells) {
        hipLaunchKernelGGL(m->immersed ? k_corrector_cells<true> : k_corrector_cells<false>,
                           dim3((ni_ + 63) / 64, (nj + 3) / 4, g.Nz), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                           m->f[GB25_BT_U].d, m->f[GB25_BT_V].d, m->f[GB25_U_BAR].d, m->f[GB25_V_BAR].d, m->colsum[0].d,
                           m->colsum[1].d, i0_, ni_, skf, sk, jr0, nj, jskf, jsk);
      } else {
        auto kern = m->immersed ? (fold ? k_corrector<true, true> : k_corrector<true, false>)
                                : (fold ? k_corrector<false, true> : k_corrector<false, false>);
        hipLaunchKernelGGL(kern, grid2(ni_, nj, b), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                           m->f[GB25_BT_U].d, m->f[GB25_BT_V].d, m->f[GB25_U_BAR].d, m->f[GB25_V_BAR].d,
                           cs ? m->colsum[0].d : nullptr, cs ? m->colsum[1].d : nullptr, i0_, ni_, mom_kchunks(m),
                           skf, sk, jr0, nj, jskf, jsk, m->corr_out ? m->corr[0].d : nullptr, m->corr_out ? m->corr[1].d : nullptr);
      }
    };
    // rows: the own ones (the halo rows of a rank of a 2-D decomposition arrive corrected: group 10 travels after this)
    launch(i0, ni, skip_from, skip, 0, g.Ny, INT_MAX, 0);
    LAUNCHCHK();
  }
  if (part == 2) return GB25_OK;
  m->colsum_valid = false;
  // cache_previous_tendencies!: G^- <- G^n is a pointer exchange; the next tendency evaluation
  // overwrites the (old G^-) buffers that now carry the G^n name.
  for (int q = 0; q < 4; q++) std::swap(m->f[GB25_GN_U + q].d, m->f[GB25_GM_U + q].d);
  // (closure = CATKE: G^-.e is written by the e step itself -- cache_previous_tendencies! skips e)
  m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;   // the look-aheads used the tendency pairs as they were before
  return GB25_OK;
}

// mask_immersed_model_fields! (src/precompile.jl:34): a sweep of its own only where the kernels of a composite step do
// not already guarantee the zeros (update_state!, initialize!, after host writes)
gb25_status mask_impl(gb25_model* m) {
  if (!m->immersed) return GB25_OK;
  Range r_mask(m, "mask_immersed_model_fields");   // (the wall faces of v are the halo fill's business on the plain grid)
  const Grid& g = m->g;
  dim3 b(64, 4);
  hipLaunchKernelGGL(k_mask_immersed, grid2(g.Nx, g.Ny + 1, b), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                     m->f[GB25_T].d, m->f[GB25_S].d, m->f[GB25_BT_U].d, m->f[GB25_BT_V].d);
  LAUNCHCHK();
  // the look-ahead partners of u, v, T, S hold the same zeros from the kernels' own masks; nothing else to do
  return GB25_OK;
}

// closure = CATKEVerticalDiffusivity(): what update_state! adds -- the advection of e (the two-wide tracer kernel with e in
// both halves: a first version, the second half is thrown away), the buoyancy field, J^b, then the diffusivity fields with
// their halo cells and the explicit TKE terms added to G^n.e.  After the tendencies of T, S (same stream).
CatkePar catke_parameters(const gb25_model* m) {
  const gb25_catke_parameters& p = m->catke_par;
  CatkePar c;
  c.Cs = (real)p.Cs; c.Cb = (real)p.Cb; c.Csp = (real)p.Csp; c.CRid = (real)p.CRid; c.CRi0 = (real)p.CRi0;
  for (int q = 0; q < 4; q++) {
    c.Chi[q] = (real)p.Chi[q]; c.Clo[q] = (real)p.Clo[q]; c.Cun[q] = (real)p.Cun[q]; c.Cc[q] = (real)p.Cc[q]; c.Ce[q] = (real)p.Ce[q];
  }
  c.CWu = (real)p.CWu; c.CWw = (real)p.CWw; c.emin = (real)p.minimum_tke; c.Jbmin = (real)p.minimum_convective_buoyancy_flux;
  c.tau_neg = (real)p.negative_tke_damping_time_scale;
  c.CWeps = (real)p.CWeps;
  c.rCRid = (real)(1.0 / p.CRid);
  c.rtau_neg = (real)(1.0 / p.negative_tke_damping_time_scale);
  return c;
}
// The extended range on which a rank of a decomposition COMPUTES N^2 and the diffusivities: the first halo column on either
// side, the first halo row of an open side and the row beyond a zipper fold (single domain: the interior, halos by images)
struct CatkeRange { int i_lo, i_hi, j_lo, j_hi; };
// chunks of levels of the CATKE column kernels: enough threads for ~10 waves per SIMD, at least 8 levels per chunk
inline int catke_level_chunks(const gb25_model* m, long columns) {
  int kch = 1;
  while (kch < 8 && columns * kch < 600000L && m->g.Nz / (kch + 1) >= 8) kch++;
  return kch;
}
inline CatkeRange catke_range(const gb25_model* m) {
  const Grid& g = m->g;
  CatkeRange r;
  r.i_lo = m->slab ? -1 : 0;
  r.i_hi = g.Nx + (m->slab ? 1 : 0);
  r.j_lo = m->ys_open ? -1 : 0;
  r.j_hi = g.Ny + ((m->yn_open || (m->slab && g.cv.north_fold)) ? 1 : 0);
  return r;
}
gb25_status catke_implicit_impl(gb25_model* m, int mode, real dt, real chi, int z0, int nz);
// compute_diffusivities!(diffusivities, closure::CATKE, model), first part: N^2 of the current T, S, the e step
// (time_step_catke_equation!: k_catke_tke_step + the implicit solve of e) on the own columns, J^b filtered.
// What follows needs the halos of e and J^b: catke_diffusivities_finish_impl.
gb25_status catke_tke_step_impl(gb25_model* m) {
  if (!m->catke) return GB25_OK;
  Timed t_closure(m, GB25_K_CLOSURE);
  const Grid& g = m->g;
  const CatkeRange r = catke_range(m);
  dim3 b(64, 4);
  {
    const int ni = r.i_hi - r.i_lo, nj = r.j_hi - r.j_lo;
    hipLaunchKernelGGL(m->immersed ? k_catke_n2<true> : k_catke_n2<false>, dim3((ni + 63) / 64, (nj + 3) / 4, g.Nz - 1), b, 0,
                       m->stream, g, m->f[GB25_T].d, m->f[GB25_S].d, m->catke_b.d, r.i_lo, r.j_lo, ni, nj);
  }
  // dt = clock.last_dt (finite from the model's construction on: GB-25 src/baroclinic_instability_model.jl:82), chi = 0.1 always
  const real dt = (real)m->last_dt, chi = (real)m->cfg.chi;
  const real* prev_uv[2];   // previous_velocities, wherever they are (prev_uv_src)
  for (int q = 0; q < 2; q++)
    prev_uv[q] = m->prev_uv_src == 1 ? m->f[GB25_U + q].d : m->prev_uv_src == 2 ? m->ahead_uv[q].d : m->f[GB25_PREV_U + q].d;
  // chunks of levels (catke_kernels.hpp): with fewer than ~600 k columns a thread per column leaves the chip short of waves
  const int kch = catke_level_chunks(m, (long)g.Nx * g.Ny);
  m->catke_e_star = kch > 1 ? m->catke_scratch.d : m->f[GB25_E].d;   // (where the implicit solve of e finds e*)
  dim3 gt = grid2(g.Nx, g.Ny, b);
  gt.z = kch;
  hipLaunchKernelGGL(m->immersed ? k_catke_tke_step<true> : k_catke_tke_step<false>, gt, b, 0, m->stream, g,
                     catke_parameters(m), dt, real(1.5) + chi, real(0.5) + chi, m->f[GB25_U].d, m->f[GB25_V].d,
                     prev_uv[0], prev_uv[1], m->f[GB25_E].d, m->catke_e_star, m->catke_b.d, m->f[GB25_JB].d,
                     m->f[GB25_KAPPA_U].d, m->f[GB25_KAPPA_C].d, m->f[GB25_KAPPA_E].d, m->f[GB25_LE].d, m->f[GB25_GN_E].d,
                     m->f[GB25_GM_E].d);
  LAUNCHCHK();
  if (gb25_status s = catke_implicit_impl(m, 1, dt, chi, 1, 1)) return s;   // (the e slice alone)
  // previous velocities <- velocities (parents).  Single domain with the look-ahead of u, v: nothing moves -- the next AB2 step
  // adopts the look-ahead's buffers and leaves these behind untouched, which is where the next compute_diffusivities! finds them
  // (prev_uv_src; 0.8 GB of copies per step at 1440x720x48 otherwise).  A rank of a decomposition, whose stages write the
  // partner buffers ahead of this one, and models without the look-ahead: D2D copies behind the kernel that read them.
  m->prev_uv_src = 1;
  if (m->slab || m->ab2_ahead != 1 || m->ptr_exposed || !m->two_streams)
    if (gb25_status s_ = materialize_prev_uv(m)) return s_;
  const double dt_since = m->time - m->catke_prev_time;
  m->catke_prev_time = m->time;
  hipLaunchKernelGGL(m->immersed ? k_catke_surface_flux<true> : k_catke_surface_flux<false>, grid2(g.Nx, g.Ny, b), b, 0, m->stream,
                     g, catke_parameters(m), (real)dt_since, m->f[GB25_U].d, m->f[GB25_V].d, m->f[GB25_T].d, m->f[GB25_S].d,
                     m->f[GB25_E].d, m->catke_b.d, m->f[GB25_JB].d, m->catke_src.d);
  LAUNCHCHK();
  return GB25_OK;
}
// ... second part, once the halo cells of the new e (and, on a rank of a decomposition, of J^b) are in place: kappa_u, kappa_c,
// kappa_e with the halo cells their fill derives (a14).
gb25_status catke_diffusivities_finish_impl(gb25_model* m) {
  if (!m->catke) return GB25_OK;
  Timed t_closure(m, GB25_K_CLOSURE);
  const Grid& g = m->g;
  const CatkeRange r = catke_range(m);
  dim3 b(64, 4);
  dim3 gd = grid2(r.i_hi - r.i_lo, r.j_hi - r.j_lo, b);
  gd.z = catke_level_chunks(m, (long)(r.i_hi - r.i_lo) * (r.j_hi - r.j_lo));
  hipLaunchKernelGGL(m->immersed ? k_catke_diffusivities<true> : k_catke_diffusivities<false>,
                     gd, b, 0, m->stream, g, catke_parameters(m), m->f[GB25_U].d,
                     m->f[GB25_V].d, m->f[GB25_E].d, m->catke_b.d, m->f[GB25_JB].d, m->f[GB25_KAPPA_U].d,
                     m->f[GB25_KAPPA_C].d, m->f[GB25_KAPPA_E].d, r.i_lo, r.i_hi, r.j_lo, r.j_hi);
  if (g.cv.north_fold && !m->slab)   // the rows beyond the zipper
    hipLaunchKernelGGL(k_catke_fold, dim3((g.sx + 255) / 256, g.H, g.Nz + 3), dim3(256), 0, m->stream, g,
                       m->f[GB25_KAPPA_U].d, m->f[GB25_KAPPA_C].d, m->f[GB25_KAPPA_E].d, m->f[GB25_LE].d, m->f[GB25_JB].d);
  LAUNCHCHK();
  return GB25_OK;
}
// compute_diffusivities! on a single domain: the two parts with the halo fill of the new e between them (option
// CATKE_STALE_E_HALOS: without it -- Oceananigans as recalled; the bottom / top layers of e are never read)
gb25_status catke_diffusivities_impl(gb25_model* m) {
  if (!m->catke) return GB25_OK;
  gb25_status s;
  if ((s = catke_tke_step_impl(m))) return s;
  if (!m->catke_stale_e_halos && (s = fill_halos_impl(m, true, false, 1, 4))) return s;
  return catke_diffusivities_finish_impl(m);
}
// what compute_tendencies! leaves in G^n.e: the SLOW tendency of e -- -div(u e) (one tracer, the two halves of the packed
// arithmetic on two columns of it: k_tracer_tendencies_single) and the top boundary condition of e (the surface TKE flux)
gb25_status catke_tendency_impl(gb25_model* m) {
  if (!m->catke) return GB25_OK;
  Timed t_closure(m, GB25_K_CLOSURE);
  const Grid& g = m->g;
  {
    const int nbx = (g.Nx + V3_PAIR - 1) / V3_PAIR, nby = (g.Ny + 3) / 4, kchunks = std::max(1, g.Nz / m->trc_chunk_levels);
    const int nb = nbx * nby * kchunks;
    constexpr int TW = sizeof(real) == 8 ? 2 : 4;
    const bool fly = m->w_fly_now;   // (w on the fly: the field w is stale, the kernel carries w up its chunks like the two others)
    void (*kt)(Grid, const real*, const real*, const real*, const real*, real*, int, int, int, LazyCorr) =
        (m->tracer_order == 7 && fly)
            ? (g.cv.on       ? k_tracer_tendencies_single<TW, true, true, 7, true>
               : m->immersed ? k_tracer_tendencies_single<TW, true, false, 7, true>
                             : k_tracer_tendencies_single<TW, false, false, 7, true>)
        : m->tracer_order == 7
            ? (g.cv.on       ? k_tracer_tendencies_single<TW, true, true, 7>
               : m->immersed ? k_tracer_tendencies_single<TW, true, false, 7>
                             : k_tracer_tendencies_single<TW, false, false, 7>)
        : fly ? (g.cv.on       ? k_tracer_tendencies_single<TW, true, true, 5, true>
                 : m->immersed ? k_tracer_tendencies_single<TW, true, false, 5, true>
                               : k_tracer_tendencies_single<TW, false, false, 5, true>)
        : g.cv.on       ? k_tracer_tendencies_single<TW, true, true, 5>
        : m->immersed ? k_tracer_tendencies_single<TW, true, false, 5>
                      : k_tracer_tendencies_single<TW, false, false, 5>;
    const LazyCorr lz{m->corr[0].d, m->corr[1].d, m->wbase, g.sx * g.sy_v};
    hipLaunchKernelGGL(kt, dim3(nb), dim3(64, 4), 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d, m->f[GB25_W].d, m->f[GB25_E].d,
                       m->f[GB25_GN_E].d, nbx, kchunks, nb, lz);
  }
  dim3 b(64, 4);
  hipLaunchKernelGGL(k_catke_add_top_source, grid2(g.Nx, g.Ny, b), b, 0, m->stream, g, m->catke_src.d, m->f[GB25_GN_E].d);
  LAUNCHCHK();
  return GB25_OK;
}
// single domain: compute_diffusivities! followed by the slow tendency of e (where update_state! has both at its end)
gb25_status catke_update_impl(gb25_model* m) {
  if (!m->catke) return GB25_OK;
  gb25_status s = catke_diffusivities_impl(m);
  return s ? s : catke_tendency_impl(m);
}
// implicit_step! with CATKE's diffusivity fields.  mode 0: u, v (and the corrector's column integrals); 1: slice 0 = T with S
// (ab2_step!), slice 1 = e (inside compute_diffusivities!: e holds e* already).  z0, nz: the slices of this launch
// (default: both of mode 0, slice 0 of mode 1)
gb25_status catke_implicit_impl(gb25_model* m, int mode, real dt, real chi, int z0, int nz) {
  if (nz < 0) nz = mode == 0 ? 2 : 1;
  const Grid& g = m->g;
  Timed t_implicit(m, GB25_K_IMPLICIT);
  ImplicitVarFields A{};
  A.f[0] = m->f[GB25_U].d; A.f[1] = m->f[GB25_V].d; A.f[2] = m->f[GB25_T].d; A.f[3] = m->f[GB25_S].d; A.f[4] = m->f[GB25_E].d;
  A.KU = m->f[GB25_KAPPA_U].d; A.KC = m->f[GB25_KAPPA_C].d; A.KE = m->f[GB25_KAPPA_E].d; A.Le = m->f[GB25_LE].d;
  A.dt = dt;
  A.GnE = A.GmE = nullptr;   // (e arrives with its AB2 update done: k_catke_tke_step)
  A.src_e = (mode == 1 && z0 == 1) ? m->catke_e_star : nullptr;   // (... in e itself, or in the scratch array when the step ran in chunks of levels)
  A.C1 = real(1.5) + chi; A.C2 = real(0.5) + chi;
  A.z0 = z0;
  A.sum[0] = mode == 0 ? m->colsum[0].d : nullptr;
  A.sum[1] = mode == 0 ? m->colsum[1].d : nullptr;
  A.kchunks = mom_kchunks(m);
  A.P = mode == 0 ? m->uv_partials : nullptr;   // (null without the look-ahead of u, v)
  A.plane2 = g.sx * g.sy_v;
  const bool imm = m->immersed;
  void (*kern)(Grid, ImplicitVarFields);
#define VARK(NZT) (mode == 0 ? (imm ? k_implicit_vertical_var<NZT, true, 0> : k_implicit_vertical_var<NZT, false, 0>) \
                             : (imm ? k_implicit_vertical_var<NZT, true, 1> : k_implicit_vertical_var<NZT, false, 1>))
  constexpr int REG_LEVELS = sizeof(real) == 8 ? 32 : 64;   // three per-thread arrays of that many values fit the VGPRs
  if (g.Nz <= 32) kern = VARK(32);
  else if (g.Nz <= 48 && REG_LEVELS >= 48) kern = VARK(48);
  else if (g.Nz <= 64 && REG_LEVELS >= 64) kern = VARK(64);
  else {   // deeper columns stream through HBM with the factors in two scratch arrays
    gb25_status s;
    for (int q = 0; q < 2; q++) {
      if (!m->catke_gam[q].d && (s = alloc_field(m, m->catke_gam[q], g.sx, g.sy_v, g.Nz + 2 * g.H))) return s;
      A.gam[q] = m->catke_gam[q].d;
    }
    kern = mode == 0 ? (imm ? k_implicit_vertical_var_stream<true, 0> : k_implicit_vertical_var_stream<false, 0>)
                     : (imm ? k_implicit_vertical_var_stream<true, 1> : k_implicit_vertical_var_stream<false, 1>);
  }
#undef VARK
  dim3 b(64, 4);
  const int rows = mode == 0 ? v_rows(g) : g.Ny;   // (with the zipper fold v has the fold line too)
  hipLaunchKernelGGL(kern, dim3((g.Nx + 63) / 64, (rows + 3) / 4, nz), b, 0, m->stream, g, A);
  LAUNCHCHK();
  if (mode == 0) m->colsum_valid = true;
  return GB25_OK;
}

// compute_atmosphere_ocean_fluxes! + the net ocean fluxes of ClimaOcean's OceanSeaIceModel (data-free forcing): similarity
// theory per surface cell, then the top flux boundary conditions of u, v, T, S (allocated here if the host set none)
gb25_status atmosphere_ocean_fluxes_impl(gb25_model* m) {
  if (!m->coupled) return GB25_OK;
  Timed t_fluxes(m, GB25_K_FLUXES);
  const Grid& g = m->g;
  const size_t n2 = (size_t)g.sx * g.sy_v;
  for (int q = 0; q < 4; q++) {
    if (!m->d_top_flux[q]) {
      HIPCHK(hipMalloc(&m->d_top_flux[q], n2 * sizeof(real)));
      HIPCHK(hipMemsetAsync(m->d_top_flux[q], 0, n2 * sizeof(real), m->stream));
    }
    m->g.top_flux[q] = m->d_top_flux[q];
  }
  for (int q = 0; q < 2; q++)
    if (!m->d_tau[q]) {
      HIPCHK(hipMalloc(&m->d_tau[q], n2 * sizeof(double)));
      HIPCHK(hipMemsetAsync(m->d_tau[q], 0, n2 * sizeof(double), m->stream));
    }
  Atmosphere A;
  for (int q = 0; q < 7; q++) A.a[q] = m->d_atm[q];
  const int j_hi = g.Ny;
  dim3 b(64, 4);
  hipLaunchKernelGGL(m->immersed ? k_similarity_fluxes<true> : k_similarity_fluxes<false>, grid2(g.Nx + 1, j_hi + 1, b), b, 0,
                     m->stream, m->g, A, m->f[GB25_U].d, m->f[GB25_V].d, m->f[GB25_T].d, m->f[GB25_S].d, m->d_tau[0],
                     m->d_tau[1], m->d_top_flux[2], m->d_top_flux[3], j_hi, 5);
  hipLaunchKernelGGL(k_stress_to_faces, grid2(g.Nx, j_hi, b), b, 0, m->stream, m->g, m->d_tau[0], m->d_tau[1],
                     m->d_top_flux[0], m->d_top_flux[1], j_hi);
  LAUNCHCHK();
  return GB25_OK;
}

gb25_status update_state_impl(gb25_model* m) {
  gb25_status s;
  if ((s = mask_impl(m))) return s;
  if ((s = fill_halos_impl(m, true))) return s;
  if (m->catke && (s = fill_halos_impl(m, true, false, 1, 4))) return s;   // the TKE tracer
  if ((s = compute_w_impl(m))) return s;
  if ((s = compute_p_impl(m))) return s;
  // (compute_diffusivities! ahead of the tendencies, as update_state! has them: neither reads what the other writes, and the
  // e step must have read CATKE's previous velocities before the momentum kernel's look-ahead reuses their buffers)
  if ((s = catke_update_impl(m))) return s;
  if ((s = momentum_impl(m))) return s;
  return tracers_impl(m);
}

gb25_status ab2_step_impl(gb25_model* m, double dt, int euler) {
  gb25_status s;
  m->ahead_baro_valid = false;   // this route always runs the sub-cycle itself
  const real chi = euler ? -real(0.5) : (real)m->cfg.chi;
  if ((s = ab2_local_impl(m, (real)dt, chi))) return s;
  Halo2 hG = halo2_G(m);
  if ((s = fill_halos_2d(m, hG))) return s;
  return barotropic_impl(m, (real)dt);
}

// u, v <- u + du, v + dv: ends the state in which the corrector lives inside its consumers (before a composite call returns)
// need_w = false: the step that follows carries w inside its tendency kernels as well (w on the fly beside the corrector's sweep):
// the field w stays stale
gb25_status materialize_uv(gb25_model* m, bool need_w = true) {
  if (m->uv_lazy) {
    const Grid& g = m->g;
    dim3 b(64, 4);
    hipLaunchKernelGGL(k_apply_correction, grid2(g.sx, g.sy_v, b), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                       LazyCorr{m->corr[0].d, m->corr[1].d, nullptr, 0});
    LAUNCHCHK();
    m->uv_lazy = false;
  }
  m->w_fly_now = false;
  if (need_w && m->w_stale) {   // the steps behind carried w inside their tendency kernels: the field itself, from the corrected velocities
    m->w_stale = false;
    return compute_w_impl(m);
  }
  return GB25_OK;
}
// may this step leave u, v uncorrected in memory?  Flat lat-lon single domain, both look-aheads on and able to write
// their halos, the default kernels; `more`: the step is one of a composite call (gb25_loop), which materialises u, v, w when it returns
inline bool lazy_corrector_ok(const gb25_model* m) {
  return m->lazy_corrector && !m->coupled && m->bottom_drag == 0 && m->tracer_order == 5 && m->two_streams && producers_fold(m) && !m->immersed && !m->g.cv.on && m->kernel_gen >= 2 &&
         m->ab2_ahead == 1 && !m->ptr_exposed && m->pressure_bits == 64;
}

// w on the fly beside the corrector's SWEEP (round 4): the grids the corrector-inside-its-consumers instances do not exist for --
// a GridFittedBottom, the curvilinear grids, the zipper fold -- still drop the k_compute_w launch: the sweep leaves du, dv of the
// own columns as a by-product, k_w_bases turns the look-ahead's chunk integrals + du, dv into w at the chunk boundaries, and the
// WFLY instances of the two tendency kernels carry w up their chunks.  Single domain, between the steps of one gb25_loop call.
// ... and the sweep itself through the tracer kernel (above: uvc): everything wfly_sweep_ok asks for, the tracer kernel first
inline bool lazy_through_tracers_ok(const gb25_model* m);
inline bool wfly_sweep_ok(const gb25_model* m) {
  // (like the corrector inside its consumers it rides on the sub-cycle look-ahead -- on by default from 8 M cells on: small models
  // keep the stand-alone w, bit for bit what their decompositions compute)
  // (the sweep leaves corrected velocities in memory: the similarity-theory fluxes of a coupled model, the quadratic bottom drag and
  // WENO(order = 7) tracers -- the data-free climate model -- read them like anything else; instances of the three tendency kernels exist)
  return m->w_fly && m->baro_ahead != 0 && !m->slab && m->two_streams && m->kernel_gen >= 2 &&
         m->ab2_ahead == 1 && !m->ptr_exposed && m->nu == 0 && m->kappa == 0 &&
         // (closure = CATKE: the implicit solve of u, v that follows the AB2 update rewrites the look-ahead's chunk sums with
         // those of the velocities it leaves: catke_implicit_impl, ImplicitVarFields::P; e is advected by a kernel that carries w too)
         std::max(1, m->g.Nz / m->trc_chunk_levels) == mom_kchunks(m);
}

inline bool lazy_through_tracers_ok(const gb25_model* m) {
  // (not with a closure: its kernels read u, v ahead of the tracer kernel that would write the corrected ones)
  return m->lazy_corrector && wfly_sweep_ok(m) && !m->catke && !m->coupled && m->bottom_drag == 0 && m->tracer_order == 5 &&
         (m->immersed || m->g.cv.on) && m->tracers_first != 0 && m->pressure_bits == 64;
}

// ... and a slab of an x decomposition or a rank of a 2-D one: the same kernels without the halo images (its halos come with the
// bundles, and with them the neighbours' column integrals: du, dv of the halo columns / rows are computed locally)
inline bool slab_lazy_ok(const gb25_model* m) {
  return m->slab && m->lazy_corrector && !m->coupled && m->bottom_drag == 0 && m->tracer_order == 5 && m->two_streams &&
         !m->immersed && !m->g.cv.on && m->kernel_gen >= 2 && m->ab2_ahead == 1 && !m->ptr_exposed && m->pressure_bits == 64 &&
         m->nu == 0 && m->kappa == 0 && !m->catke;
}

inline bool slab_wfly_ok(const gb25_model* m) {   // (the chunkings of the two tendency kernels must be the one the chunk sums were made with)
  return slab_lazy_ok(m) && m->w_fly && std::max(1, m->g.Nz / m->trc_chunk_levels) == mom_kchunks(m);
}

// One time step on a single slab.  Two HIP streams: the tracer branch (AB2 of T,S -> their halos -> hydrostatic
// pressure: HBM- then fp64-bound) is independent of the velocity branch (AB2 of u,v -> split-explicit sub-cycle,
// which is latency-bound -> halos -> corrector -> halos -> w) until the tendencies need both, so it runs on a
// side stream and overlaps.  The phase order within each branch is the reference's (src/precompile.jl:31-42);
// T and S are untouched between the two halo fills of the reference sequence, so they are filled once.
gb25_status time_step_impl(gb25_model* m, int euler, bool more = false) {
  gb25_status s;
  const double dt = m->last_dt;
  struct Composite {
    gb25_model* m;
    explicit Composite(gb25_model* m_) : m(m_) { m->composite = true; }
    ~Composite() { m->composite = false; }
  } composite_scope(m);
  if (!m->two_streams) {
    if ((s = materialize_uv(m))) return s;
    if ((s = ab2_step_impl(m, dt, euler))) return s;
    m->time += dt;
    m->iteration += 1;
    if ((s = fill_halos_impl(m, true))) return s;
    if ((s = corrector_impl(m, true))) return s;
    if ((s = update_state_impl(m))) return s;
    return atmosphere_ocean_fluxes_impl(m);
  }
  const real chi = euler ? -real(0.5) : (real)m->cfg.chi;
  hipStream_t main = m->stream, side = m->side_stream;
  // ---- AB2 of u, v: normally the adoption of the look-ahead (a pointer exchange on the host, no kernel)
  const bool adopted = m->ahead_uv_valid && (real)dt == m->ahead_uv_dt && chi == m->ahead_uv_chi;
  const bool baro_adopted = adopted && m->ahead_baro_valid;   // (made from that very look-ahead, same dt)
  m->ahead_baro_valid = false;
  // The corrector inside its consumers: when everything this step needs was made ahead of time (u, v, the sub-cycle) and
  // another step follows, no sweep over u and v at all -- a 2-D kernel leaves du, dv and w, the tendency kernels add them.
  const bool lazy = more && adopted && baro_adopted && m->complete_fills_needed == 0 && lazy_corrector_ok(m);
  // ... or the sweep stays and only w moves into the tendency kernels (the look-ahead's chunk integrals must be this step's)
  // ... or the tracer kernel applies it and writes the corrected velocities (grids with a bottom, curvilinear, folded) ...
  const bool lazy_t = !lazy && more && adopted && baro_adopted && m->complete_fills_needed == 0 && lazy_through_tracers_ok(m);
  const bool wfly_sweep = !lazy && !lazy_t && more && adopted && m->complete_fills_needed == 0 && wfly_sweep_ok(m);
  if (!lazy && (s = materialize_uv(m, !(wfly_sweep || lazy_t)))) return s;   // (the stand-alone kernels below expect corrected velocities)
  if ((s = ab2_velocities_impl(m, (real)dt, chi))) return s;
  Halo2 hG = halo2_G(m);
  // the sub-cycle reads G.U, G.V at interior points only (periodic wrap and walls are in the kernel): their halo
  // fill is for the state's sake and leaves the critical path when no kernel on this stream produced them
  if (!adopted && (s = fill_halos_2d(m, hG))) return s;
  const bool ts_adopted = m->ahead_valid && (real)dt == m->ahead_dt && chi == m->ahead_chi;
  // Everything the tracer branch reads was complete when the previous step's momentum kernel had finished (both
  // look-aheads adopted, the tracer kernel ran before the momentum kernel, no closure whose implicit solve sits in between):
  // the branch then starts behind THAT kernel -- beside this step's sub-cycle look-ahead, which is still in the queue of
  // the main stream -- and not behind everything the main stream holds.  (Not across the zipper fold: the fill of G.U, G.V
  // rewrites the eastern half of the fold line, which the sub-cycle reads.)
  const bool early_fork = m->tend_forkable && adopted && ts_adopted && !m->catke && m->nu == 0 && m->kappa == 0 && !m->coupled &&
                          !m->g.cv.north_fold;
  m->tend_forkable = false;
  HIPCHK(hipEventRecord(m->ev_fork, main));
  HIPCHK(hipStreamWaitEvent(side, early_fork ? m->ev_tend : m->ev_fork, 0));
  // ---- tracer branch (side stream)
  m->stream = side;
  // (complete fills for the two steps after a host write, one per buffer of each alternating pair: see fold_fills)
  const bool complete = m->complete_fills_needed > 0;
  if (complete) m->complete_fills_needed -= 1;
  s = ab2_tracers_impl(m, (real)dt, chi);
  // y/z/x halos of T, S -- unless the look-ahead that was just adopted wrote them itself
  if (!s && (complete || !(ts_adopted && m->ahead_ts_folded))) s = fill_halos_impl(m, true, false, 1, 2);
  // (closure = CATKE: e is not stepped by ab2_step!; its halos were refilled after its step inside compute_diffusivities! --
  // unless the option keeps them stale there, as Oceananigans does as recalled: then they are filled here, with the others)
  if (!s && m->catke && m->catke_stale_e_halos) s = fill_halos_impl(m, true, false, 1, 4);
  if (!s && hipEventRecord(m->ev_ts, side) != hipSuccess) s = fail(m, GB25_ERR_HIP, "hipEventRecord(ev_ts) failed");
  if (!s) s = compute_p_impl(m, INT_MIN, INT_MIN, 0, -1, true);
  if (!s && adopted) s = fill_halos_2d(m, hG);
  m->stream = main;
  if (s) return s;
  HIPCHK(hipEventRecord(m->ev_join, side));
  // ---- velocity branch (main stream)
  if (baro_adopted) {
    // the sub-cycle of this step ran in the previous one: adopt eta, U, V and the filtered state
    if (m->baro_inflight) HIPCHK(hipStreamWaitEvent(main, m->ev_baro, 0));
    m->baro_inflight = false;
    for (int q = 0; q < 3; q++) {
      std::swap(m->f[GB25_ETA + q].d, m->ahead_eta[q].d);
      std::swap(m->f[GB25_ETA_BAR + q].d, m->ahead_bar[q].d);
    }
    std::swap(m->bars, m->bars_ahead);
    m->last_baro_folded = m->ahead_eta_folded;
  } else if ((s = barotropic_impl(m, (real)dt))) {
    return s;
  }
  const bool eta_halos_fresh = m->last_baro_folded;   // the last launch of this step's sub-cycle wrote them
  m->time += dt;
  m->iteration += 1;
  // The reference fills the halos of u, v, eta, U, V here as well as after the corrector.  On a single slab the
  // corrector reads and writes its own columns only, and the fill after it rewrites exactly the same halo cells from
  // the corrected interior, so the first fill has no effect on any later value: it is left out (3 launches).
  if (lazy) {
    const Grid& g = m->g;
    dim3 b(64, 4);
    Timed t(m, GB25_K_CORRECTOR);
    hipLaunchKernelGGL(k_corrector_2d<false>, grid2(g.Nx, g.Ny + 1, b), b, 0, main, g, m->f[GB25_BT_U].d, m->f[GB25_BT_V].d,
                       m->colsum[0].d, m->colsum[1].d, m->f[GB25_U_BAR].d, m->f[GB25_V_BAR].d, m->corr[0].d, m->corr[1].d,
                       0, g.Nx, INT_MAX, 0, 0, g.Ny + 1);
    LAUNCHCHK();
    m->uv_lazy = true;
    m->colsum_valid = false;
    // w on the fly: the chunkings of the two tendency kernels must be the one the partial sums were made with
    m->w_fly_now = m->w_fly && std::max(1, g.Nz / m->trc_chunk_levels) == mom_kchunks(m);
    if (m->w_fly_now) {
      hipLaunchKernelGGL((k_w_bases<false, false>), grid2(g.Nx + 4, g.Ny + 4, b), b, 0, main, g, m->uv_partials, mom_kchunks(m), g.sx * g.sy_v,
                         LazyCorr{m->corr[0].d, m->corr[1].d, nullptr, 0}, m->wbase, -2, g.Nx + 4, INT_MAX, 0);
      LAUNCHCHK();
      m->w_stale = true;
    }
    for (int q = 0; q < 4; q++) std::swap(m->f[GB25_GN_U + q].d, m->f[GB25_GM_U + q].d);   // cache_previous_tendencies!
    m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
  } else if (lazy_t) {
    const Grid& g = m->g;
    dim3 b(64, 4);
    Timed t(m, GB25_K_CORRECTOR);
    for (int q = 0; q < 2; q++)
      if (!m->uvc[q].d && (s = alloc_field(m, m->uvc[q], m->f[GB25_U + q].nx, m->f[GB25_U + q].ny, m->f[GB25_U + q].nz))) return s;
    // du, dv on the own faces (rows of y faces: [0, Ny) on a folded grid, [0, Ny] below a wall) ...
    hipLaunchKernelGGL(m->immersed ? k_corrector_2d<true> : k_corrector_2d<false>, grid2(g.Nx, g.Ny + (g.cv.north_fold ? 0 : 1), b), b, 0, main, g,
                       m->f[GB25_BT_U].d, m->f[GB25_BT_V].d, m->colsum[0].d, m->colsum[1].d, m->f[GB25_U_BAR].d, m->f[GB25_V_BAR].d,
                       m->corr[0].d, m->corr[1].d, 0, g.Nx, INT_MAX, 0, 0, g.Ny + (g.cv.north_fold ? 0 : 1));
    LAUNCHCHK();
    // ... and their halo cells as the fills derive them (the rows beyond a zipper fold: images with the sign of a vector component)
    if (g.cv.north_fold) {
      Halo2 hc{};
      hc.p[0] = m->corr[0].d; hc.is_v[0] = 0; hc.xf[0] = 1; hc.neg[0] = 1;
      hc.p[1] = m->corr[1].d; hc.is_v[1] = 1; hc.xf[1] = 0; hc.neg[1] = 1;
      hc.n = 2;
      if ((s = fill_halos_2d(m, hc))) return s;
    }
    m->colsum_valid = false;
    m->w_fly_now = true;
    {
      const LazyCorr lc{m->corr[0].d, m->corr[1].d, nullptr, 0};
      auto kb = g.cv.on ? k_w_bases<true, true> : (m->immersed ? k_w_bases<true, false> : k_w_bases<false, false>);
      hipLaunchKernelGGL(kb, grid2(g.Nx + 4, g.Ny + 4, b), b, 0, main, g, m->uv_partials, mom_kchunks(m), g.sx * g.sy_v, lc,
                         m->wbase, -2, g.Nx + 4, INT_MAX, 0);
      LAUNCHCHK();
    }
    m->w_stale = true;
    m->uv_corr_pending = true;
    // (the two strips of halo cells of the uncorrected u, v the tracer kernel reads: k_uncorrected_edges)
    hipLaunchKernelGGL(k_uncorrected_edges, dim3((std::max(g.Nx, g.Ny) + 255) / 256, g.Nz, g.cv.north_fold ? 3 : 2), dim3(256), 0, main, g,
                       m->f[GB25_U].d, m->f[GB25_V].d);
    LAUNCHCHK();
    for (int q = 0; q < 4; q++) std::swap(m->f[GB25_GN_U + q].d, m->f[GB25_GM_U + q].d);   // cache_previous_tendencies!
    m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
  } else {
    m->corr_out = wfly_sweep;      // (the sweep leaves du, dv of the own columns behind for k_w_bases)
    s = corrector_impl(m, true);
    m->corr_out = false;
    if (s) return s;
    m->w_fly_now = wfly_sweep;
    if (wfly_sweep) {
      const Grid& g = m->g;
      dim3 b(64, 4);
      const LazyCorr lc{m->corr[0].d, m->corr[1].d, nullptr, 0};
      auto kb = g.cv.on ? k_w_bases<true, true> : (m->immersed ? k_w_bases<true, false> : k_w_bases<false, false>);
      hipLaunchKernelGGL(kb, grid2(g.Nx + 4, g.Ny + 4, b), b, 0, main, g, m->uv_partials, mom_kchunks(m), g.sx * g.sy_v, lc,
                         m->wbase, -2, g.Nx + 4, INT_MAX, 0);
      LAUNCHCHK();
      m->w_stale = true;
    }
  }
  {
    // u, v and eta, U, V -- whatever their last writers (the corrector, the sub-cycle's last launch) did not fill
    // (lazy_t: u, v get their halo cells behind the tracer kernel, which writes their corrected interior: tracers_impl)
    const bool uv_fresh = lazy || lazy_t || (producers_fold(m) && m->composite && !complete);
    const int which = (uv_fresh ? 0 : 1) | ((eta_halos_fresh && !complete) ? 0 : 2);
    if (which && (s = fill_halos_impl(m, true, false, which, 1))) return s;
  }
  if (!((lazy || wfly_sweep || lazy_t) && m->w_fly_now) && (s = compute_w_impl(m))) return s;
  if (lazy_t && !m->tracers_first) return fail(m, GB25_ERR_STATE, "internal: the corrector through the tracer kernel needs the tracer kernel first");
  // ---- join: the tendencies need w, u, v and T, S (the tracers) / the pressure differences (the momentum)
  const bool tracers_first = m->tracers_first != 0;
  if (m->catke) {   // compute_diffusivities! + the slow tendency of e, ahead of the tendency kernels (see update_state_impl)
    HIPCHK(hipStreamWaitEvent(main, m->ev_ts, 0));
    if ((s = catke_update_impl(m))) return s;
  }
  if (tracers_first) {
    HIPCHK(hipStreamWaitEvent(main, m->ev_ts, 0));
    if ((s = tracers_impl(m))) return s;
  }
  HIPCHK(hipStreamWaitEvent(main, m->ev_join, 0));
  if ((s = momentum_impl(m))) return s;
  if (tracers_first) {
    HIPCHK(hipEventRecord(m->ev_tend, main));
    m->tend_forkable = true;
  }
  if (m->baro_ahead && m->ahead_uv_valid && !m->ptr_exposed) {
    // G.U, G.V of the next step exist now, and with them everything its split-explicit sub-cycle needs: it runs here,
    // into the partner buffers, and leaves the head of the next step (where the corrector waits for it).
    if (m->baro_ahead == 2) {
      // ... on a stream of its own beside the tracer tendency kernel.  (Measured on MI355X, profiles/r02b: the first of
      // its three launches is dispatched 7 us after the tracer kernel's 16 560 blocks and does not become resident until
      // they have drained -- 800 us instead of 60 -- so the "beside" is mostly an "after", and the other two launches
      // then compete with the next step's pressure kernel.  Kept as a schedule; not the default.)
      if (!m->baro_stream) HIPCHK(hipStreamCreateWithFlags(&m->baro_stream, hipStreamNonBlocking));
      HIPCHK(hipEventRecord(m->ev_mom, main));
      HIPCHK(hipStreamWaitEvent(m->baro_stream, m->ev_mom, 0));
      m->stream = m->baro_stream;
      s = barotropic_impl(m, m->ahead_uv_dt, true);
      m->ahead_eta_folded = m->last_baro_folded;
      m->stream = main;
      if (s) return s;
      HIPCHK(hipEventRecord(m->ev_baro, m->baro_stream));
      m->baro_inflight = true;
    } else {
      // ... on the main stream, between the momentum and the tracer tendencies: three launches of 60 us with the GPU to
      // themselves, no cross-stream dependency at all.
      if ((s = barotropic_impl(m, m->ahead_uv_dt, true))) return s;
      m->ahead_eta_folded = m->last_baro_folded;
    }
    m->ahead_baro_valid = true;
  }
  if (!tracers_first && (s = tracers_impl(m))) return s;
  return atmosphere_ocean_fluxes_impl(m);   // (a coupled model: the fluxes the NEXT evaluation of the tendencies sees)
}
// a coupled model at iteration 0: the fluxes of the initial state, then the tendencies again, which now see them
gb25_status first_fluxes_impl(gb25_model* m) {
  if (!m->coupled) return GB25_OK;
  gb25_status s;
  if ((s = atmosphere_ocean_fluxes_impl(m))) return s;
  if ((s = catke_update_impl(m))) return s;
  if ((s = momentum_impl(m))) return s;
  return tracers_impl(m);
}

gb25_status initialize_impl(gb25_model* m) {
  const Grid& g = m->g;
  dim3 b(64, 4);
  hipLaunchKernelGGL(k_barotropic_mode, grid2(g.Nx, v_rows(g), b), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                     m->f[GB25_BT_U].d, m->f[GB25_BT_V].d);
  LAUNCHCHK();
  return fill_halos_2d(m, halo2_prognostic(m));
}

// The bottom tables from whatever the bottom currently is: the host's bottom height (gb25_set_bottom_height, global cell centres),
// the analytic mountains of grid_type, or nothing (flat: tables only where the kernels take everything from tables).
gb25_status rebuild_bottom(gb25_model* m) {
  const gb25_config& c = m->cfg;
  const bool islands = c.grid_type == GB25_GRID_LAT_LON_GAUSSIAN_ISLANDS || c.grid_type == GB25_GRID_TRIPOLAR_GAUSSIAN_ISLANDS;
  if (!m->host_bottom.empty() || islands || m->g.cv.on) {
    if (c.Nz > 254) return fail(m, GB25_ERR_INVALID_ARGUMENT, "an immersed boundary / a curvilinear grid needs Nz <= 254 (8-bit level tables)");
  }
  if (!m->host_bottom.empty()) {
    const int Nxg = c.Nx, i0 = m->rx * m->Nx;
    gb25_status s = build_bottom(m, [&](int i, int j) { return m->host_bottom[(size_t)((((i + i0) % Nxg) + Nxg) % Nxg) + (size_t)Nxg * j]; });
    if (s) return s;
    if (m->slab) m->immersed = true;   // (every slab of a decomposition runs the kernel variants its neighbours run)
    return GB25_OK;
  }
  if (islands) {
    if (!m->host_curv[0].empty())
      return fail(m, GB25_ERR_STATE, "grid_type = gaussian_islands places its mountains by the built-in generator's coordinates: "
                                     "with a host grid pass the bottom height too (gb25_set_bottom_height)");
    gb25_status s = build_bottom(m, [&](int i, int j) { return gaussian_islands_bottom(m, i, j); });
    m->immersed = true;   // (every slab of a decomposition runs the kernel variants the single domain runs, mountains or not)
    return s;
  }
  // the curvilinear kernels take every reconstruction order and mask from the tables (that is also where the fold's
  // "north is not a wall" lives): a flat bottom is a bottom nothing touches
  if (m->g.cv.on) return build_bottom(m, [&](int, int) { return -1e30; });
  return GB25_OK;
}
// everything in flight finishes and every look-ahead is void: the grid under the model is about to change
gb25_status quiesce_for_grid_change(gb25_model* m) {
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(hipStreamSynchronize(m->own_stream));
  HIPCHK(hipStreamSynchronize(m->side_stream));
  if (m->baro_stream) HIPCHK(hipStreamSynchronize(m->baro_stream));
  m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
  m->colsum_valid = m->halo_colsum_valid = false;
  m->tend_forkable = false;
  m->complete_fills_needed = 2;
  m->implicit_key[0][0] = m->implicit_key[1][0] = -1.0;   // (the elimination tables of a closure hold the old spacings)
  return GB25_OK;
}

}  // namespace

#include "slab_step.hpp"
#include "state_io.hpp"

// =============================================================================================
extern "C" {

const char* gb25_version(void) { return sizeof(real) == 8 ? "gb25hip 0.2 (gfx950, Float64)" : "gb25hip 0.2 (gfx950, Float32)"; }
int32_t gb25_real_bytes(void) { return (int32_t)sizeof(real); }
int32_t gb25_config_bytes(void) { return (int32_t)sizeof(gb25_config); }
int32_t gb25_catke_parameters_bytes(void) { return (int32_t)sizeof(gb25_catke_parameters); }

void gb25_default_config(gb25_config* c, int32_t Nx, int32_t Ny, int32_t Nz) {
  memset(c, 0, sizeof *c);
  c->Nx = Nx; c->Ny = Ny; c->Nz = Nz;
  c->halo = 8; c->substeps = 30; c->rank = 0; c->nranks = 1; c->device = 0;
  c->dt = 60.0; c->chi = 0.1;
  c->lat_south = -80; c->lat_north = 80; c->lon_west = 0; c->lon_east = 360;
  c->depth = 4000; c->zexp_h = 30;
  c->g = 9.80665; c->Omega = 7.292115e-5; c->radius = 6371e3; c->rho0 = 1020.0;
  c->slab_mode = 0; c->grid_type = GB25_GRID_LAT_LON; c->ranks_y = 1;
}

gb25_status gb25_create(const gb25_config* cfg, gb25_model** out) {
  if (!cfg || !out) return GB25_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  gb25_model* m = new gb25_model();
  gb25_default_catke_parameters(&m->catke_par);
  *out = m;  // returned even on failure so the caller can read the error string, then destroy
  m->cfg = *cfg;
  m->Ry = cfg->ranks_y > 1 ? cfg->ranks_y : 1;
  if (cfg->Nx < 8 || cfg->Ny < 8 || cfg->Nz < 4 || cfg->halo < 4 || cfg->substeps < 1 || cfg->substeps > 4096 ||
      cfg->nranks < 1 || cfg->rank < 0 || cfg->rank >= cfg->nranks || cfg->nranks % m->Ry != 0 ||
      cfg->Nx % (cfg->nranks / m->Ry) != 0 || cfg->Ny % m->Ry != 0)
    return fail(m, GB25_ERR_INVALID_ARGUMENT,
                "invalid configuration: need Nx,Ny >= 8, Nz >= 4, halo >= 4, 1 <= substeps <= 4096, nranks = Rx ranks_y, "
                "Nx %% Rx == 0, Ny %% ranks_y == 0");
  if (cfg->slab_mode < 0 || cfg->slab_mode > 1)
    return fail(m, GB25_ERR_INVALID_ARGUMENT, "slab_mode must be 0 (x halos by exchange iff nranks > 1) or 1 (always)");
  m->Rx = cfg->nranks / m->Ry;
  m->rx = cfg->rank % m->Rx;
  m->ry = cfg->rank / m->Rx;
  m->Nx = cfg->Nx / m->Rx;
  m->Ny = cfg->Ny / m->Ry;
  m->j0 = m->ry * m->Ny;
  m->ys_open = m->ry > 0;
  m->yn_open = m->ry < m->Ry - 1;
  m->slab = cfg->nranks > 1 || cfg->slab_mode == 1;
  {
    // Levels per chunk of the two tendency kernels (options MOMENTUM_ / TRACER_CHUNK_LEVELS): 24 where that still leaves four
    // rounds of blocks on the chip's 1024 block slots -- 1440 x 720 x 48: two chunks per column instead of four, +1.3 % steps/s
    // (415.5 -> 420.9 on one box; 1440 x 720 x 60 with the islands 280 -> 285.5), the start-up of a chunk's march paid half as
    // often --, 12 on narrower models: a 180-column rank of an 8-way decomposition would drop to one round and 0.47 -> 0.53 ms
    // per step (tools/slab_selfring.py).  The chunking is the association of the column integrals of u, v: results differ in the
    // last bits between the two, so a bit-for-bit comparison of a wide single domain with its narrow ranks pins the options.
    const int nbx = (m->Nx + 63) / 64, nby = (m->Ny + 3) / 4;
    if ((long)nbx * nby * std::max(1, cfg->Nz / 24) >= 4096) m->mom_chunk_levels = m->trc_chunk_levels = 24;
  }
  if (m->Nx < cfg->halo) return fail(m, GB25_ERR_INVALID_ARGUMENT, "slab narrower than the halo");
  if (m->Ry > 1) {
    // a rank of a 2-D decomposition steps on the staged sequence without the interior / edge split of the tendencies and
    // without the early pressure of the own columns
    m->split_tendencies = 0;
    if (m->Ny < cfg->halo + 2) return fail(m, GB25_ERR_INVALID_ARGUMENT, "a rank's band of rows is narrower than the halo");
  }
  {
    // The kernels address a parent array with 32-bit ELEMENT indices; the tendency kernels with 32-bit BYTE offsets from the
    // first plane their block touches (tendency_kernels.hpp): a chunk of levels plus its stencil planes must stay below 2 GB.
    const double plane = (double)(m->Nx + 2 * cfg->halo) * (m->Ny + 2 * cfg->halo + 1);
    const double elems = plane * (cfg->Nz + 2 * cfg->halo + 1);
    const int kchunks = std::max(1, cfg->Nz / 12), klen = (cfg->Nz + kchunks - 1) / kchunks;
    if (elems >= 2147483648.0 || plane * (klen + 10) * sizeof(real) >= 2147483648.0)
      return fail(m, GB25_ERR_INVALID_ARGUMENT,
                  "a %dx%dx%d slab has %.2g elements per 3-D array (%.1f GB) and %.2g per plane; the kernels index at most 2^31 "
                  "elements per array and %d planes of 2 GB together: decompose in x (nranks) so that the local slab is narrower",
                  m->Nx, m->Ny, cfg->Nz, elems, elems * sizeof(real) / 1e9, plane, klen + 10);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(m, GB25_ERR_NO_DEVICE, "no HIP device visible; libgb25hip has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(m, GB25_ERR_INVALID_ARGUMENT, "device ordinal %d out of range (%d devices)", cfg->device, ndev);
  HIPCHK(hipSetDevice(cfg->device));
  {
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    m->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  HIPCHK(hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking));
  // (few streams on purpose: HIP multiplexes streams onto a handful of hardware queues, and two streams that share one
  // run in order -- a rocprof trace of eight slabs in one process, 25 streams, shows the exchange stream and the main
  // stream taking turns.  A slab has three: own, side, and the exchange stream of its context; the stream of
  // SUBCYCLE_LOOKAHEAD = 2 is created on demand.)
#ifdef GB25_SIDE_PRIO
  {   // (tools/build_variant.sh experiment: the side stream -- the pressure branch -- at the LOWEST / main at the highest priority)
    int least = 0, greatest = 0;
    HIPCHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    HIPCHK(hipStreamCreateWithPriority(&m->side_stream, hipStreamNonBlocking, GB25_SIDE_PRIO > 0 ? least : greatest));
  }
#else
  HIPCHK(hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking));
#endif
  HIPCHK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&m->ev_baro, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&m->ev_mom, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&m->ev_ts, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&m->ev_tend, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&m->ev_strips, hipEventDisableTiming));

  m->stream = m->own_stream;
  m->last_dt = cfg->dt;
  // On launch-latency-bound grids the extra cross-stream hops of the sub-cycle look-ahead cost more than the
  // sub-cycle they hide (0.252 vs 0.262 ms/step at 360x180x24, 0.173 vs 0.185 at 128x64x8; +3.5 % at 1440x720x48).
  // A slab of a decomposition always uses it: there it also takes two exchanges off the critical path.
  // (gb25_set_option changes any of these defaults; nothing is read from the environment.)
  m->baro_ahead = (m->slab || (long)cfg->Nx * cfg->Ny * cfg->Nz >= 8000000L) ? 1 : 0;
  gb25_status s;
  if ((s = build_grid(m))) return s;
  if ((s = build_eos_tables(m))) return s;
  build_substeps(m);
  if (cfg->grid_type < 0 || cfg->grid_type >= GB25_GRID_COUNT)
    return fail(m, GB25_ERR_INVALID_ARGUMENT, "grid_type %d: 0 lat-lon, 1 lat-lon with the Gaussian islands, 2 lat-lon "
                "through the curvilinear kernels, 3 tripolar, 4 tripolar with the Gaussian islands", cfg->grid_type);
  if (cfg->grid_type >= GB25_GRID_LAT_LON_AS_CURVILINEAR) {
    if (cfg->grid_type >= GB25_GRID_TRIPOLAR && (cfg->Nx % 2 || m->Ny < 2 * cfg->halo))
      return fail(m, GB25_ERR_INVALID_ARGUMENT, "the tripolar grid needs an even Nx and Ny >= 2 halo (the fold maps columns onto columns)");
    if ((s = build_curv_grid(m))) return s;
  }
  const int H = cfg->halo, sx = m->Nx + 2 * H;
  for (int id = 0; id < GB25_FIELD_COUNT; id++) {
    if (id >= GB25_ETA_BAR && id <= GB25_V_BAR) continue;  // allocated contiguously below
    if (is_catke_field(id)) continue;   // (allocated when the closure is switched on)
    int ny = m->Ny + 2 * H + (is_v_shaped(id) ? 1 : 0);
    int nz = is_2d(id) ? 1 : cfg->Nz + 2 * H + (id == GB25_W ? 1 : 0);
    if ((s = alloc_field(m, m->f[id], sx, ny, nz))) return s;
  }
  {
    size_t nc = (size_t)sx * (m->Ny + 2 * H), nv = (size_t)sx * (m->Ny + 2 * H + 1);
    HIPCHK(hipMalloc(&m->bars, (2 * nc + nv) * sizeof(real)));
    HIPCHK(hipMemset(m->bars, 0, (2 * nc + nv) * sizeof(real)));
    Field& e = m->f[GB25_ETA_BAR]; e.d = m->bars; e.nx = sx; e.ny = m->Ny + 2 * H; e.nz = 1;
    Field& u = m->f[GB25_U_BAR]; u.d = m->bars + nc; u.nx = sx; u.ny = m->Ny + 2 * H; u.nz = 1;
    Field& v = m->f[GB25_V_BAR]; v.d = m->bars + 2 * nc; v.nx = sx; v.ny = m->Ny + 2 * H + 1; v.nz = 1;
  }
  for (int q = 0; q < 3; q++) {
    if ((s = alloc_field(m, m->pp[q], sx, m->f[GB25_ETA + q].ny, 1))) return s;
    if ((s = alloc_field(m, m->pp2[q], sx, m->f[GB25_ETA + q].ny, 1))) return s;
    if ((s = alloc_field(m, m->ahead_eta[q], sx, m->f[GB25_ETA + q].ny, 1))) return s;
  }
  {   // partners of the filtered state, laid out like `bars`
    size_t off = 0, tot = 0;
    for (int q = 0; q < 3; q++) tot += m->f[GB25_ETA_BAR + q].elems();
    HIPCHK(hipMalloc(&m->bars_ahead, tot * sizeof(real)));
    HIPCHK(hipMemset(m->bars_ahead, 0, tot * sizeof(real)));
    for (int q = 0; q < 3; q++) {
      m->ahead_bar[q] = m->f[GB25_ETA_BAR + q];
      m->ahead_bar[q].d = m->bars_ahead + off;
      off += m->f[GB25_ETA_BAR + q].elems();
    }
  }
  if ((s = alloc_field(m, m->dpx, m->f[GB25_PHY].nx, m->f[GB25_PHY].ny, m->f[GB25_PHY].nz))) return s;
  if ((s = alloc_field(m, m->dpy, m->f[GB25_PHY].nx, m->f[GB25_PHY].ny, m->f[GB25_PHY].nz))) return s;

  for (int q = 0; q < 2; q++) {
    if ((s = alloc_field(m, m->ahead[q], m->f[GB25_T].nx, m->f[GB25_T].ny, m->f[GB25_T].nz))) return s;
    const Field &V3 = m->f[GB25_U + q], &V2 = m->f[GB25_GN_BT_U + q], &C2 = m->f[GB25_BT_U + q];
    if ((s = alloc_field(m, m->ahead_uv[q], V3.nx, V3.ny, V3.nz))) return s;
    if ((s = alloc_field(m, m->ahead_G[q], V2.nx, V2.ny, 1))) return s;
    if ((s = alloc_field(m, m->ahead_colsum[q], C2.nx, C2.ny, 1))) return s;
  }
  {
    const size_t np = (size_t)4 * std::max(1, m->g.Nz / 6) * m->g.sx * m->g.sy_v;   // (room for chunks of 6 levels)
    HIPCHK(hipMalloc(&m->uv_partials, np * sizeof(real)));
    HIPCHK(hipMemset(m->uv_partials, 0, np * sizeof(real)));
    HIPCHK(hipMalloc(&m->wbase, (np / 4) * sizeof(real)));
    HIPCHK(hipMemset(m->wbase, 0, (np / 4) * sizeof(real)));
  }
  if ((s = alloc_field(m, m->corr[0], sx, m->f[GB25_BT_V].ny, 1))) return s;
  if ((s = alloc_field(m, m->corr[1], sx, m->f[GB25_BT_V].ny, 1))) return s;
  if ((s = alloc_field(m, m->colsum[0], sx, m->f[GB25_BT_U].ny, 1))) return s;
  if ((s = alloc_field(m, m->colsum[1], sx, m->f[GB25_BT_V].ny, 1))) return s;
  if (m->slab || m->g.cv.north_fold) {
    if (m->slab) {
      // wide enough that after the Ns substeps the valid region still covers the slab's x HALO columns of eta, U, V: they are
      // computed here like the neighbour computes them (same inputs, same arithmetic) and no exchange follows the sub-cycle
      m->W = m->Ns + 1 + m->cfg.halo;
      if (m->Nx < m->W)
        return fail(m, GB25_ERR_INVALID_ARGUMENT, "slab width %d is narrower than the barotropic halo %d", m->Nx, m->W);
    }
    // folded grid: image rows beyond the pivot row, enough that what the last row's missing neighbour spoils (one row per
    // substep) never reaches the pivot row
    // (Oceananigans errors when the extended halo of its free surface exceeds the grid; so does this: with fewer rows the spoiled
    // rows would reach the pivot row and eta, U, V next to the fold would be wrong without a word)
    if (m->g.cv.north_fold) {
      if (m->Ny - 2 < m->Ns + 1)
        return fail(m, GB25_ERR_INVALID_ARGUMENT, "a folded grid needs Ny >= %d rows per rank for its %d effective substeps (the sub-cycle's "
                    "image rows beyond the pivot row: Ns + 1 of them, made from the rows south of it); this rank has %d", m->Ns + 3, m->Ns, m->Ny);
      m->Wy = m->Ns + 1;
    }
    // 2-D decomposition: wide halos in y as in x on the sides where a neighbour rank exists
    if (m->yn_open) m->Wy = m->W;
    if (m->ys_open) m->Wys = m->W;
    if ((m->yn_open || m->ys_open) && m->Ny < m->W)
      return fail(m, GB25_ERR_INVALID_ARGUMENT, "a rank's band of %d rows is narrower than the barotropic halo %d", m->Ny, m->W);
    const int wsx = m->Nx + 2 * m->W, wy = m->Wy + m->Wys;
    for (int a = 0; a < 2; a++)
      for (int q = 0; q < 3; q++)
        if ((s = alloc_field(m, m->wide[a][q], wsx, m->f[GB25_ETA + q].ny + wy, 1))) return s;
    {   // the three running averages are one allocation, zeroed by one memset per step
      size_t tot = 0;
      for (int q = 0; q < 3; q++) {
        m->wideBar[q].nx = wsx; m->wideBar[q].ny = m->f[GB25_ETA + q].ny + wy; m->wideBar[q].nz = 1;
        tot += m->wideBar[q].elems();
      }
      real* base = nullptr;
      HIPCHK(hipMalloc(&base, tot * sizeof(real)));
      HIPCHK(hipMemset(base, 0, tot * sizeof(real)));
      for (int q = 0; q < 3; q++) {
        m->wideBar[q].d = base;
        base += m->wideBar[q].elems();
      }
    }
    if ((s = alloc_field(m, m->wideG[0], wsx, m->f[GB25_GN_BT_U].ny + wy, 1))) return s;
    if ((s = alloc_field(m, m->wideG[1], wsx, m->f[GB25_GN_BT_V].ny + wy, 1))) return s;
    if (m->g.cv.on && (s = build_curv_wide(m))) return s;
    if (m->Wy && !m->slab) HIPCHK(hipMalloc(&m->tall_buf, (size_t)5 * (m->Wy + 1) * wsx * sizeof(real)));
  }
  if ((s = rebuild_bottom(m))) return s;
  HIPCHK(hipDeviceSynchronize());
  return GB25_OK;
}

void gb25_destroy(gb25_model* m) {
  if (!m) return;
  if (m->group) group_destroy(m->group);   // (the exchange context of every slab it holds)
  if (m->own_stream) hipStreamSynchronize(m->own_stream);
  for (int id = 0; id < GB25_FIELD_COUNT; id++)
    if (!(id >= GB25_ETA_BAR && id <= GB25_V_BAR) && m->f[id].d) hipFree(m->f[id].d);
  if (m->bars) hipFree(m->bars);
  if (m->dpx.d) hipFree(m->dpx.d);
  if (m->dpy.d) hipFree(m->dpy.d);

  for (int q = 0; q < 3; q++)
    for (Field* p : {&m->pp[q], &m->pp2[q], &m->ahead_eta[q]})
      if (p->d) hipFree(p->d);
  if (m->bars_ahead) hipFree(m->bars_ahead);
  if (m->tall_buf) hipFree(m->tall_buf);
  for (auto& p : m->colsum)
    if (p.d) hipFree(p.d);
  for (auto& p : m->corr)
    if (p.d) hipFree(p.d);
  for (int q = 0; q < 2; q++)
    for (Field* p : {&m->ahead[q], &m->ahead_uv[q], &m->ahead_G[q], &m->ahead_colsum[q]})
      if (p->d) hipFree(p->d);
  if (m->uv_partials) hipFree(m->uv_partials);
  if (m->wbase) hipFree(m->wbase);
  for (auto p : m->d_ord)
    if (p) hipFree(p);
  for (auto p : m->d_H)
    if (p) hipFree(p);
  for (auto p : m->d_wideH)
    if (p) hipFree(p);
  for (auto p : m->d_wideM)
    if (p) hipFree(p);
  for (auto p : m->d_bottom_flux)
    if (p) hipFree(p);
  for (auto p : m->d_atm)
    if (p) hipFree(p);
  for (auto p : m->d_tau)
    if (p) hipFree(p);
  for (auto p : m->d_top_flux)
    if (p) hipFree(p);
  for (auto p : m->d_implicit)
    if (p) hipFree(p);
  if (m->catke_b.d) hipFree(m->catke_b.d);
  if (m->catke_scratch.d) hipFree(m->catke_scratch.d);
  if (m->catke_src.d) hipFree(m->catke_src.d);
  for (auto& F : m->uvc) if (F.d) hipFree(F.d);
  for (auto& F : m->catke_gam) if (F.d) hipFree(F.d);
  for (int a = 0; a < 2; a++) {
    for (auto& w : m->wide[a])
      if (w.d) hipFree(w.d);
    if (m->wideG[a].d) hipFree(m->wideG[a].d);
  }
  if (m->wideBar[0].d) hipFree(m->wideBar[0].d);   // one allocation for all three
  for (real* t : m->dev_tables) hipFree(t);
  resolve_profile(m);
  for (auto& ev : m->free_events) {
    hipEventDestroy(ev.a);
    hipEventDestroy(ev.b);
  }
  if (m->side_stream) {
    hipStreamSynchronize(m->side_stream);
    hipStreamDestroy(m->side_stream);
  }
  if (m->baro_stream) {
    hipStreamSynchronize(m->baro_stream);
    hipStreamDestroy(m->baro_stream);
  }
  if (m->ev_fork) hipEventDestroy(m->ev_fork);
  if (m->ev_join) hipEventDestroy(m->ev_join);
  if (m->ev_baro) hipEventDestroy(m->ev_baro);
  if (m->ev_mom) hipEventDestroy(m->ev_mom);
  if (m->ev_ts) hipEventDestroy(m->ev_ts);
  if (m->ev_tend) hipEventDestroy(m->ev_tend);
  if (m->ev_strips) hipEventDestroy(m->ev_strips);
  if (m->own_stream) hipStreamDestroy(m->own_stream);
  delete m;
}

const char* gb25_last_error_string(const gb25_model* m) { return m ? m->err.c_str() : "null model"; }

gb25_status gb25_set_stream(gb25_model* m, void* s) {
  CHECK_MODEL(m);
  if (m->group) return fail(m, GB25_ERR_STATE, "the slabs of an exchange context run on the context's own two streams");
  m->stream = (hipStream_t)s;  // NULL = HIP's default stream; ordering between streams is the caller's business
  return GB25_OK;
}
gb25_status gb25_use_own_stream(gb25_model* m) {
  CHECK_MODEL(m);
  if (m->group) return GB25_OK;
  HIPCHK(hipStreamSynchronize(m->stream));
  m->stream = m->own_stream;
  return GB25_OK;
}
gb25_status gb25_synchronize(gb25_model* m) {
  CHECK_MODEL(m);
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(hipStreamSynchronize(m->side_stream));
  if (m->baro_stream) HIPCHK(hipStreamSynchronize(m->baro_stream));   // the sub-cycle look-ahead may still be running there
  if (m->group) HIPCHK(m->group->sync_side());   // (a slab's runs on the second stream of its context)
  return GB25_OK;
}

gb25_status gb25_field_dims(const gb25_model* m, gb25_field id, int include_halos, int32_t d[3]) {
  if (!m || id < 0 || id >= GB25_FIELD_COUNT || !d) return GB25_ERR_INVALID_ARGUMENT;
  const Field& F = m->f[id];
  if (!F.d) return GB25_ERR_INVALID_ARGUMENT;   // (a field of CATKE on a model whose closure is not CATKE)
  const int H = m->cfg.halo;
  // A y-face field of a folded grid has Ny rows ((Periodic, RightConnected, Bounded): the faces beyond the last row of cells are
  // halo cells), so its parent has Ny + 2H rows like a cell-centred one; the device arrays keep the row a Bounded grid needs.
  const int ny = F.ny - (((m->g.cv.north_fold || m->yn_open) && is_v_shaped(id)) ? 1 : 0);   // (likewise below a northern neighbour rank)
  if (include_halos) {
    d[0] = F.nx; d[1] = ny; d[2] = F.nz;
  } else {
    d[0] = F.nx - 2 * H; d[1] = ny - 2 * H; d[2] = is_2d(id) ? 1 : F.nz - 2 * H;
  }
  return GB25_OK;
}

static gb25_status copy_field(gb25_model* m, gb25_field id, real* host, int include_halos, bool to_device) {
  if (!m || id < 0 || id >= GB25_FIELD_COUNT || !host) return GB25_ERR_INVALID_ARGUMENT;
  if (!m->f[id].d) return fail(m, GB25_ERR_INVALID_ARGUMENT, "this model has no such field (closure = CATKEVerticalDiffusivity() only)");
  if (to_device && (id == GB25_U || id == GB25_V)) m->colsum_valid = false;  // cached column integrals are stale
  if (m->uv_lazy)   // (only after a composite call that failed half-way: memory must hold the corrected velocities)
    if (gb25_status s = materialize_uv(m)) return s;
  if (gb25_status s = materialize_prev_uv(m)) return s;   // (CATKE's previous velocities into their fields before the host reads or writes)
  Field& F = m->f[id];
  HIPCHK(hipStreamSynchronize(m->stream));
  if (to_device) {   // a look-ahead may still be reading the old values
    HIPCHK(hipStreamSynchronize(m->side_stream));
    if (m->baro_stream) HIPCHK(hipStreamSynchronize(m->baro_stream));
    if (m->group) HIPCHK(m->group->sync_side());
  }
  int32_t d[3];
  gb25_field_dims(m, id, include_halos, d);
  if (include_halos && d[1] == F.ny) {
    if (to_device) HIPCHK(hipMemcpy(F.d, host, F.elems() * sizeof(real), hipMemcpyHostToDevice));
    else HIPCHK(hipMemcpy(host, F.d, F.elems() * sizeof(real), hipMemcpyDeviceToHost));
    return GB25_OK;
  }
  const int H = include_halos ? 0 : m->cfg.halo;
  hipMemcpy3DParms p = {};
  hipPitchedPtr dev = make_hipPitchedPtr(F.d, (size_t)F.nx * sizeof(real), F.nx, F.ny);
  hipPitchedPtr hst = make_hipPitchedPtr(host, (size_t)d[0] * sizeof(real), d[0], d[1]);
  hipPos dpos = make_hipPos((size_t)H * sizeof(real), H, is_2d(id) ? 0 : H), zero = make_hipPos(0, 0, 0);
  p.extent = make_hipExtent((size_t)d[0] * sizeof(real), d[1], d[2]);
  if (to_device) {
    p.srcPtr = hst; p.srcPos = zero; p.dstPtr = dev; p.dstPos = dpos; p.kind = hipMemcpyHostToDevice;
  } else {
    p.srcPtr = dev; p.srcPos = dpos; p.dstPtr = hst; p.dstPos = zero; p.kind = hipMemcpyDeviceToHost;
  }
  HIPCHK(hipMemcpy3D(&p));
  return GB25_OK;
}
static gb25_status widen_phy(gb25_model* m) {   // the host uploaded pHY': rebuild the differences from it
  long n = (long)m->f[GB25_PHY].elems();
  hipLaunchKernelGGL(k_pressure_differences, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, m->stream, m->g,
                     m->f[GB25_PHY].d, m->dpx.d, m->dpy.d, n);
  LAUNCHCHK();
  return GB25_OK;
}
// T and S alternate between two buffers (AB2 look-ahead).  Whatever the host writes into one, halos included, goes
// into the partner too, so that the halo layers no kernel ever rewrites are the same in both.
static gb25_status mirror_tracers(gb25_model* m) {
  m->ahead_valid = false;
  m->complete_fills_needed = 2;
  for (int q = 0; q < 2; q++)
    HIPCHK(hipMemcpyAsync(m->ahead[q].d, m->f[GB25_T + q].d, m->f[GB25_T + q].elems() * sizeof(real),
                          hipMemcpyDeviceToDevice, m->stream));
  return GB25_OK;
}
static gb25_status mirror_velocities(gb25_model* m) {   // u and v alternate between two buffers likewise
  m->ahead_uv_valid = false;
  m->complete_fills_needed = 2;
  for (int q = 0; q < 2; q++)
    HIPCHK(hipMemcpyAsync(m->ahead_uv[q].d, m->f[GB25_U + q].d, m->f[GB25_U + q].elems() * sizeof(real),
                          hipMemcpyDeviceToDevice, m->stream));
  return GB25_OK;
}
gb25_status gb25_set_field(gb25_model* m, gb25_field f, const void* host, int include_halos) {
  CHECK_MODEL(m);
  gb25_status s = collective_guard(m, 1, (unsigned)f, 0.0);
  if (s) return s;
  s = copy_field(m, f, static_cast<real*>(const_cast<void*>(host)), include_halos, true);
  if (s == GB25_OK && f == GB25_PHY) {
    m->phy_stale = false;
    s = widen_phy(m);
  }
  if (s == GB25_OK && m->immersed && (f == GB25_U || f == GB25_V || f == GB25_T || f == GB25_S || f == GB25_BT_U || f == GB25_BT_V))
    s = mask_impl(m);   // set!(model, ...) masks what it has set (as Oceananigans' set! does on an immersed grid)
  if (s == GB25_OK) {
    m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;   // any input of the look-aheads may have changed
    m->complete_fills_needed = 2;
    if (f == GB25_T || f == GB25_S) {
      s = mirror_tracers(m);
      m->n2_fresh = false;
    }
    if (f == GB25_U || f == GB25_V) s = mirror_velocities(m);
    if (s == GB25_OK && f >= GB25_ETA && f <= GB25_V_BAR) {   // eta, U, V and the filtered state alternate likewise
      Field& P = (f <= GB25_BT_V) ? m->ahead_eta[f - GB25_ETA] : m->ahead_bar[f - GB25_ETA_BAR];
      HIPCHK(hipMemcpyAsync(P.d, m->f[f].d, m->f[f].elems() * sizeof(real), hipMemcpyDeviceToDevice, m->stream));
    }
  }
  return s;
}
gb25_status gb25_get_field(gb25_model* m, gb25_field f, void* host_, int include_halos) {
  real* host = static_cast<real*>(host_);
  if (m && f == GB25_PHY && m->phy_stale) {   // the step stored only the differences: T, S are those it was made from
    gb25_status s = compute_p_impl(m);
    if (s) return s;
  }
  return copy_field(m, f, host, include_halos, false);
}
gb25_status gb25_field_device_ptr(gb25_model* m, gb25_field id, void** dev) {
  if (!m || id < 0 || id >= GB25_FIELD_COUNT || !dev) return GB25_ERR_INVALID_ARGUMENT;
  if (!m->f[id].d) return fail(m, GB25_ERR_INVALID_ARGUMENT, "this model has no such field (closure = CATKEVerticalDiffusivity() only)");
  if (gb25_status s = collective_guard(m, 2, (unsigned)id, 0.0)) return s;
  if (m->uv_lazy)
    if (gb25_status s = materialize_uv(m)) return s;
  if (gb25_status s = materialize_prev_uv(m)) return s;
  if (id == GB25_U || id == GB25_V || id == GB25_T || id == GB25_S || (id >= GB25_GN_U && id <= GB25_GM_S) ||
      (id >= GB25_ETA && id <= GB25_GN_BT_V)) {
    // the host can now write prognostic fields or their tendencies behind our back: no more look-ahead for this
    // model, u, v, T, S stay in the buffers whose addresses are handed out
    m->ptr_exposed = true;
    m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
  }
  if (id == GB25_PHY) {
    m->phy_pinned = true;
    if (m->phy_stale) {
      gb25_status s = compute_p_impl(m);
      if (s) return s;
    }
  }
  *dev = m->f[id].d;
  return GB25_OK;
}
gb25_status gb25_get_metric(const gb25_model* m, gb25_metric id, int32_t logical_index, double* v) {
  if (!m || id < 0 || id > GB25_M_DZF || !v) return GB25_ERR_INVALID_ARGUMENT;
  int off = (id <= GB25_M_FCOR) ? m->metric_off_j : m->metric_off_k;
  long a = (long)logical_index - 1 + off;  // logical_index is 1-based like the Julia sources
  if (a < 0 || a >= (long)m->h_metric[id].size()) return GB25_ERR_INVALID_ARGUMENT;
  *v = (double)(real)m->h_metric[id][a];
  return GB25_OK;
}
gb25_status gb25_get_metric2(const gb25_model* m, gb25_metric2 id, double* v, int64_t count) {
  if (!m || id < 0 || id >= GB25_M2_COUNT || !v) return GB25_ERR_INVALID_ARGUMENT;
  const std::vector<double>& a = m->h_curv[id];
  if (a.empty() || count != (int64_t)a.size()) return GB25_ERR_INVALID_ARGUMENT;   // (not a curvilinear grid, or a wrong size)
  for (size_t q = 0; q < a.size(); q++) v[q] = (double)(real)a[q];
  return GB25_OK;
}
gb25_status gb25_get_substepping(const gb25_model* m, int32_t* n, double* frac, double* w) {
  if (!m) return GB25_ERR_INVALID_ARGUMENT;
  if (n) *n = m->Ns;
  if (frac) *frac = m->dtau_frac;
  if (w) for (int k = 0; k < m->Ns; k++) w[k] = (double)(real)m->weights[k];
  return GB25_OK;
}

gb25_status gb25_set_baroclinic_instability(gb25_model* m) {
  CHECK_MODEL(m);
  if (gb25_status s = collective_guard(m, 3, 0, 0.0)) return s;
  const Grid& g = m->g;
  hipLaunchKernelGGL(k_set_baroclinic_instability, dim3((g.Nx + 255) / 256, g.Ny, g.Nz), dim3(256), 0, m->stream, g,
                     m->f[GB25_T].d, m->f[GB25_S].d);
  LAUNCHCHK();
  if (gb25_status s = mask_impl(m)) return s;
  return mirror_tracers(m);
}

gb25_status gb25_get_clock(const gb25_model* m, double* time, int64_t* it, double* last_dt) {
  if (!m) return GB25_ERR_INVALID_ARGUMENT;
  if (time) *time = m->time;
  if (it) *it = m->iteration;
  if (last_dt) *last_dt = m->last_dt;
  return GB25_OK;
}
gb25_status gb25_set_dt(gb25_model* m, double dt) {
  CHECK_MODEL(m);
  if (gb25_status s = collective_guard(m, 4, 0, dt)) return s;
  m->last_dt = dt;
  return GB25_OK;
}

// closure = VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), kappa, nu) (src/baroclinic_instability_model.jl:
// 31); nu = kappa = 0 is closure = nothing.  Collective on a decomposed model.
gb25_status gb25_set_vertical_diffusivity(gb25_model* m, double nu, double kappa) {
  CHECK_MODEL(m);
  if (!(nu >= 0) || !(kappa >= 0)) return fail(m, GB25_ERR_INVALID_ARGUMENT, "viscosity and diffusivity must be >= 0");
  if (gb25_status s = collective_guard(m, 7, 0, nu)) return s;
  if (gb25_status s = collective_guard(m, 7, 1, kappa)) return s;
  m->nu = nu;
  m->kappa = kappa;
  m->ahead_valid = false;   // (a look-ahead of T, S written with its halos predates the solve)
  if (!m->slab) m->complete_fills_needed = 2;
  return GB25_OK;
}
gb25_status gb25_set_closure_catke(gb25_model* m, int32_t on) {
  CHECK_MODEL(m);
  if (gb25_status s = collective_guard(m, 8, (unsigned)(on != 0), 0.0)) return s;
  if (on && (m->nu != 0 || m->kappa != 0)) return fail(m, GB25_ERR_STATE, "one closure at a time: the vertical diffusivity is set");
  if (on && m->cfg.Nz < 2) return fail(m, GB25_ERR_INVALID_ARGUMENT, "CATKE needs at least two levels (it lives on the faces between them)");
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(hipStreamSynchronize(m->side_stream));
  if (on && !m->f[GB25_E].d) {
    const int H = m->cfg.halo, sx = m->Nx + 2 * H, sy = m->Ny + 2 * H, nz = m->cfg.Nz + 2 * H;
    gb25_status s;
    for (int id = GB25_E; id <= GB25_PREV_V; id++) {
      const bool faces = id >= GB25_KAPPA_U && id <= GB25_KAPPA_E;
      if ((s = alloc_field(m, m->f[id], sx, sy + (is_v_shaped(id) ? 1 : 0), id == GB25_JB ? 1 : nz + (faces ? 1 : 0)))) return s;
    }
    if ((s = alloc_field(m, m->catke_b, sx, sy, nz))) return s;
    if ((s = alloc_field(m, m->catke_src, sx, sy, 1))) return s;
    if ((s = alloc_field(m, m->catke_scratch, sx, sy, nz))) return s;
  }
  if (gb25_status s_ = materialize_prev_uv(m)) return s_;   // (while the closure still says where they are)
  m->catke = on != 0;
  m->n2_fresh = false;
  m->ahead_valid = false;
  m->complete_fills_needed = 2;
  return GB25_OK;
}
void gb25_default_catke_parameters(gb25_catke_parameters* p) {
  if (!p) return;
  const double hi[4] = {0.242, 0.098, 0.548, 0.579}, lo[4] = {0.361, 0.198, 7.863, 1.604}, un[4] = {0.370, 0.369, 1.447, 0.923};
  const double cc[4] = {3.705, 4.793, 3.642, 3.254}, ce[4] = {0.0, 0.112, 0.0, 0.0};
  p->Cs = 1.131; p->Cb = 0.28; p->Csp = 0.505; p->CRid = 1.02; p->CRi0 = 0.254;
  for (int q = 0; q < 4; q++) { p->Chi[q] = hi[q]; p->Clo[q] = lo[q]; p->Cun[q] = un[q]; p->Cc[q] = cc[q]; p->Ce[q] = ce[q]; }
  p->CWu = 3.179; p->CWw = 0.383;
  p->minimum_tke = 1e-9; p->minimum_convective_buoyancy_flux = 1e-11; p->negative_tke_damping_time_scale = 60.0;
  p->CWeps = 1.0;
}
gb25_status gb25_set_catke_parameters(gb25_model* m, const gb25_catke_parameters* p) {
  CHECK_MODEL(m);
  if (!p) return GB25_ERR_INVALID_ARGUMENT;
  if (!(p->CRid > 0) || !(p->negative_tke_damping_time_scale > 0) || !(p->minimum_convective_buoyancy_flux > 0) || !(p->minimum_tke > 0))
    return fail(m, GB25_ERR_INVALID_ARGUMENT, "CATKE parameters: CRid, the damping time scale, the minimum convective buoyancy flux and the minimum TKE must be positive");
  if (gb25_status s = collective_guard(m, 10, 0, p->Cb)) return s;
  m->catke_par = *p;
  m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
  return GB25_OK;
}
gb25_status gb25_get_catke_parameters(const gb25_model* m, gb25_catke_parameters* p) {
  if (!m || !p) return GB25_ERR_INVALID_ARGUMENT;
  *p = m->catke_par;
  return GB25_OK;
}
gb25_status gb25_get_vertical_diffusivity(const gb25_model* m, double* nu, double* kappa) {
  if (!m || !nu || !kappa) return GB25_ERR_INVALID_ARGUMENT;
  *nu = m->nu;
  *kappa = m->kappa;
  return GB25_OK;
}

gb25_status gb25_initialize(gb25_model* m) { CHECK_MODEL(m); return initialize_impl(m); }
gb25_status gb25_mask_immersed_fields(gb25_model* m) {
  CHECK_MODEL(m);
  gb25_status s = mask_impl(m);
  if (s == GB25_OK && m->immersed) {   // the partner buffers of the look-aheads follow
    m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
    if ((s = mirror_tracers(m))) return s;
    s = mirror_velocities(m);
  }
  return s;
}
// GridFittedBottom(bottom_height): bottom height (metres, negative down) at the centres of the interior columns,
// Nx x Ny doubles, i fastest.  Single-domain models only (a slab would need its neighbours' columns: decomposed
// models take an analytic grid_type).  Fields are masked at once.
gb25_status gb25_set_bottom_height(gb25_model* m, const double* zb) {
  CHECK_MODEL(m);
  if (!zb) return GB25_ERR_INVALID_ARGUMENT;
  if (gb25_status s = collective_guard(m, 7, 0, zb[0])) return s;   // (synchronises the exchange stream of a decomposed model)
  if (gb25_status s = quiesce_for_grid_change(m)) return s;
  if (m->group) HIPCHK(m->group->sync_side());
  m->host_bottom.assign(zb, zb + (size_t)m->cfg.Nx * m->cfg.Ny);   // GLOBAL cell centres (a slab picks its columns, its
  gb25_status s = rebuild_bottom(m);                                 // neighbours' and its fold partner's)
  if (s) return s;
  if (m->kernel_gen == 1 && m->immersed) m->kernel_gen = 2;
  return m->slab ? mask_impl(m) : gb25_mask_immersed_fields(m);
}
// ---- the host's horizontal grid
gb25_status gb25_set_curvilinear_grid(gb25_model* m, const double* const* metrics, int32_t nx, int32_t ny) {
  CHECK_MODEL(m);
  const gb25_config& c = m->cfg;
  const int H = c.halo, gsx = c.Nx + 2 * H, gsy = c.Ny + 2 * H + 1;
  if (!m->g.cv.on)
    return fail(m, GB25_ERR_STATE, "the model steps a LatitudeLongitudeGrid with row tables: create it with a curvilinear grid_type "
                                   "(lat_lon_as_curvilinear, tripolar, gaussian_islands) to hand it 2-D metrics");
  if (!metrics || nx != gsx || (ny != gsy && ny != gsy - 1))
    return fail(m, GB25_ERR_INVALID_ARGUMENT, "gb25_set_curvilinear_grid: %d arrays of (Nx_global + 2 halo) x (Ny + 2 halo [+ 1]) = "
                                              "%d x %d [%d] doubles, got %d x %d", (int)GB25_M2_COUNT, gsx, gsy - 1, gsy, (int)nx, (int)ny);
  for (int q = 0; q < GB25_M2_COUNT; q++)
    if (!metrics[q]) return fail(m, GB25_ERR_INVALID_ARGUMENT, "gb25_set_curvilinear_grid: metric %d is null", q);
  if (gb25_status s = collective_guard(m, 8, (unsigned)ny, metrics[0][(size_t)H + (size_t)gsx * H])) return s;
  if (gb25_status s = quiesce_for_grid_change(m)) return s;
  if (m->group) HIPCHK(m->group->sync_side());
  for (int q = 0; q < GB25_M2_COUNT; q++) {
    std::vector<double>& a = m->host_curv[q];
    a.assign((size_t)gsx * gsy, 0.0);
    std::memcpy(a.data(), metrics[q], (size_t)gsx * ny * sizeof(double));
    if (ny < gsy) std::memcpy(a.data() + (size_t)gsx * ny, a.data() + (size_t)gsx * (ny - 1), (size_t)gsx * sizeof(double));   // (never read)
    // lengths and areas must be positive on the rows the kernels divide by
    if (q != GB25_M2_FFF && q != GB25_M2_PHICC)
      for (int j = 0; j < c.Ny; j++)
        for (int i = 0; i < c.Nx; i++)
          if (!(a[(size_t)(i + H) + (size_t)gsx * (j + H)] > 0.0)) {
            for (auto& h : m->host_curv) h.clear();
            return fail(m, GB25_ERR_INVALID_ARGUMENT, "gb25_set_curvilinear_grid: metric %d is not positive at interior point (%d, %d)", q, i, j);
          }
  }
  gb25_status s;
  if ((s = build_curv_grid(m))) return s;
  if ((m->slab || m->g.cv.north_fold) && (s = build_curv_wide(m))) return s;
  return GB25_OK;
}
// ---- the host's vertical grid: grid.z faces, bottom to top (exponential_z_faces(Nz, depth, h) in the reference: src/model_utils.jl:56-62)
gb25_status gb25_set_vertical_faces(gb25_model* m, const double* zf, int32_t n) {
  CHECK_MODEL(m);
  if (!zf || n != m->cfg.Nz + 1) return fail(m, GB25_ERR_INVALID_ARGUMENT, "gb25_set_vertical_faces: Nz + 1 = %d faces, bottom to top", m->cfg.Nz + 1);
  for (int k = 0; k < m->cfg.Nz; k++)
    if (!(zf[k + 1] > zf[k])) return fail(m, GB25_ERR_INVALID_ARGUMENT, "gb25_set_vertical_faces: faces must increase (face %d)", k + 1);
  if (gb25_status s = collective_guard(m, 9, (unsigned)n, zf[0])) return s;
  if (gb25_status s = quiesce_for_grid_change(m)) return s;
  if (m->group) HIPCHK(m->group->sync_side());
  m->host_zf.assign(zf, zf + n);
  gb25_status s;
  if ((s = build_grid(m))) return s;          // (row tables again too: small)
  if ((s = build_eos_tables(m))) return s;
  if ((s = rebuild_bottom(m))) return s;
  return mask_impl(m);
}
// which: 0 = number of immersed cells of column (i, j) (0-based local indices), 1 = static depth at the U face,
// 2 = at the V face
gb25_status gb25_get_bottom_info(const gb25_model* m, int32_t which, int32_t i, int32_t j, double* value) {
  if (!m || !value || j < 0 || j >= m->Ny || i < 0 || i >= m->Nx || which < 0 || which > 2) return GB25_ERR_INVALID_ARGUMENT;
  if (m->kbot.empty()) { *value = which == 0 ? 0.0 : (double)m->g.Lz; return GB25_OK; }
  const int E = m->kb_E, ksx = m->Nx + 2 * E, offk = m->metric_off_k, Nz = m->cfg.Nz;
  auto kb = [&](int ii, int jj) {
    const int jl = std::min(std::max(jj + m->j0, 0), m->cfg.Ny - 1) - m->j0;   // (clamped at the walls; a neighbour's row otherwise)
    return m->kbot[(size_t)(ii + E) + (size_t)ksx * (std::min(std::max(jl, -m->kb_Ey), m->Ny + m->kb_Ey - 1) + m->kb_Ey)];
  };
  auto depth = [&](int ii, int jj) { return (double)(real)m->h_metric[GB25_M_ZF][offk + Nz] - (double)(real)m->h_metric[GB25_M_ZF][offk + kb(ii, jj)]; };
  *value = which == 0 ? (double)kb(i, j) : which == 1 ? std::min(depth(i - 1, j), depth(i, j)) : std::min(depth(i, j - 1), depth(i, j));
  return GB25_OK;
}
gb25_status gb25_fill_halo_regions(gb25_model* m) {
  CHECK_MODEL(m);
  if (m->slab) return fail(m, GB25_ERR_STATE, "phase-by-phase driving is for single-domain models: a slab's x halos come from the exchange inside gb25_first_time_step / gb25_time_step / gb25_loop");
  return fill_halos_impl(m, true);
}
gb25_status gb25_compute_auxiliaries(gb25_model* m) {
  CHECK_MODEL(m);
  gb25_status s = compute_w_impl(m);
  if (!s) s = compute_p_impl(m);
  // (closure = CATKE: compute_diffusivities! is the third auxiliary -- it steps e and makes kappa_u, kappa_c, kappa_e, L^e, J^b)
  if (!s && m->catke && m->slab) return fail(m, GB25_ERR_STATE, "gb25_compute_auxiliaries with CATKE: phase-by-phase driving is for single-domain models");
  return s ? s : catke_diffusivities_impl(m);
}
gb25_status gb25_fill_diffusivity_halos(gb25_model* m) { CHECK_MODEL(m); return GB25_OK; }
gb25_status gb25_compute_momentum_tendencies(gb25_model* m) { CHECK_MODEL(m); return momentum_impl(m); }
gb25_status gb25_compute_tracer_tendencies(gb25_model* m) { CHECK_MODEL(m); return tracers_impl(m); }
// compute_hydrostatic_boundary_tendency_contributions! (src/precompile.jl:25,52-61): the flux boundary conditions are
// applied INSIDE the tendency kernels (the top cell's tendency gets -J/dz where it is computed, so that the AB2
// look-aheads see the complete tendency); as a phase of its own there is nothing left to do.
gb25_status gb25_compute_boundary_tendencies(gb25_model* m) { CHECK_MODEL(m); return GB25_OK; }
// FluxBoundaryCondition at the top of u, v, T or S (f = GB25_U, GB25_V, GB25_T, GB25_S): J at the interior points of the
// field's horizontal location (dims = gb25_field_dims(f, 0)[0..1], i fastest), positive upward (out of the ocean), in
// the units of the field times m/s.  NULL restores the default no-flux condition.  With ClimaOcean's ocean_simulation
// these are the arrays the coupled model fills every step (wind stress, heat and fresh-water flux).
gb25_status gb25_set_top_flux(gb25_model* m, gb25_field f, const void* host) {
  CHECK_MODEL(m);
  const int q = f == GB25_U ? 0 : f == GB25_V ? 1 : f == GB25_T ? 2 : f == GB25_S ? 3 : -1;
  if (q < 0) return fail(m, GB25_ERR_INVALID_ARGUMENT, "top flux boundary conditions exist for u, v, T, S");
  if (gb25_status s = collective_guard(m, 6, (unsigned)f, host ? 1.0 : 0.0)) return s;
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(hipStreamSynchronize(m->side_stream));
  m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
  const Field& F = m->f[f];
  const size_t n2 = (size_t)F.nx * F.ny;
  if (!host) {
    m->g.top_flux[q] = nullptr;   // (the array stays allocated for the next use)
    return GB25_OK;
  }
  if (!m->d_top_flux[q]) {
    HIPCHK(hipMalloc(&m->d_top_flux[q], n2 * sizeof(real)));
    HIPCHK(hipMemset(m->d_top_flux[q], 0, n2 * sizeof(real)));
  }
  int32_t di[3];
  gb25_field_dims(m, f, 0, di);   // (the interior of the field's horizontal location: Ny rows of y faces on a folded grid)
  const int H = m->cfg.halo, nxi = di[0], nyi = di[1];
  HIPCHK(hipMemcpy2D(m->d_top_flux[q] + (size_t)H * F.nx + H, (size_t)F.nx * sizeof(real), host, (size_t)nxi * sizeof(real),
                     (size_t)nxi * sizeof(real), nyi, hipMemcpyHostToDevice));
  m->g.top_flux[q] = m->d_top_flux[q];
  return GB25_OK;
}
gb25_status gb25_set_bottom_drag(gb25_model* m, double Cd) {
  CHECK_MODEL(m);
  if (!(Cd >= 0)) return fail(m, GB25_ERR_INVALID_ARGUMENT, "the bottom drag coefficient must be >= 0");
  if (gb25_status s = collective_guard(m, 11, 0, Cd)) return s;
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(hipStreamSynchronize(m->side_stream));
  const size_t n2 = (size_t)m->g.sx * m->g.sy_v;
  for (int q = 0; q < 2; q++) {
    if (Cd != 0 && !m->d_bottom_flux[q]) {
      HIPCHK(hipMalloc(&m->d_bottom_flux[q], n2 * sizeof(real)));
      HIPCHK(hipMemset(m->d_bottom_flux[q], 0, n2 * sizeof(real)));
    }
    m->g.bottom_flux[q] = Cd != 0 ? m->d_bottom_flux[q] : nullptr;
  }
  m->bottom_drag = Cd;
  m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
  return GB25_OK;
}
gb25_status gb25_set_tracer_advection_order(gb25_model* m, int32_t order) {
  CHECK_MODEL(m);
  if (order != 5 && order != 7) return fail(m, GB25_ERR_INVALID_ARGUMENT, "tracer_advection = WENO(order = 5) or WENO(order = 7), not %d", order);
  if (order == 7 && m->cfg.halo < 4) return fail(m, GB25_ERR_INVALID_ARGUMENT, "WENO(order = 7) needs a halo of at least 4");
  if (order == 7 && m->kernel_gen < 2) return fail(m, GB25_ERR_STATE, "the direct-stencil kernels (GB25_OPT_KERNELS = 1) know WENO(order = 5) only");
  if (gb25_status s = collective_guard(m, 12, (unsigned)order, 0.0)) return s;
  m->tracer_order = order;
  m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
  m->complete_fills_needed = 2;
  return GB25_OK;
}
gb25_status gb25_get_tracer_advection_order(const gb25_model* m, int32_t* order) {
  if (!m || !order) return GB25_ERR_INVALID_ARGUMENT;
  *order = m->tracer_order;
  return GB25_OK;
}
gb25_status gb25_get_bottom_drag(const gb25_model* m, double* Cd) {
  if (!m || !Cd) return GB25_ERR_INVALID_ARGUMENT;
  *Cd = m->bottom_drag;
  return GB25_OK;
}
gb25_status gb25_get_top_flux(gb25_model* m, gb25_field f, void* host) {
  CHECK_MODEL(m);
  const int q = f == GB25_U ? 0 : f == GB25_V ? 1 : f == GB25_T ? 2 : f == GB25_S ? 3 : -1;
  if (q < 0 || !host) return fail(m, GB25_ERR_INVALID_ARGUMENT, "top flux boundary conditions exist for u, v, T, S");
  if (!m->g.top_flux[q]) return fail(m, GB25_ERR_STATE, "no top flux boundary condition is set for this field");
  HIPCHK(hipStreamSynchronize(m->stream));
  const Field& F = m->f[f];
  int32_t di[3];
  gb25_field_dims(m, f, 0, di);
  const int H = m->cfg.halo, nxi = di[0], nyi = di[1];
  HIPCHK(hipMemcpy2D(host, (size_t)nxi * sizeof(real), m->g.top_flux[q] + (size_t)H * F.nx + H, (size_t)F.nx * sizeof(real),
                     (size_t)nxi * sizeof(real), nyi, hipMemcpyDeviceToHost));
  return GB25_OK;
}
gb25_status gb25_set_prescribed_atmosphere(gb25_model* m, gb25_atmosphere_field f, const double* host) {
  CHECK_MODEL(m);
  if ((int)f < 0 || (int)f >= GB25_ATM_COUNT) return fail(m, GB25_ERR_INVALID_ARGUMENT, "atmosphere field %d", (int)f);
  if (gb25_status s = collective_guard(m, 9, (unsigned)f, host ? 1.0 : 0.0)) return s;
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(hipStreamSynchronize(m->side_stream));
  const size_t n2 = (size_t)m->g.sx * m->g.sy_c;
  if (!host) {
    if (m->d_atm[f]) hipFree(m->d_atm[f]);
    m->d_atm[f] = nullptr;
  } else {
    if (!m->d_atm[f]) HIPCHK(hipMalloc(&m->d_atm[f], n2 * sizeof(double)));
    HIPCHK(hipMemcpy(m->d_atm[f], host, n2 * sizeof(double), hipMemcpyHostToDevice));
  }
  m->coupled = true;
  for (auto p : m->d_atm) m->coupled = m->coupled && p != nullptr;
  m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
  return GB25_OK;
}
gb25_status gb25_compute_atmosphere_ocean_fluxes(gb25_model* m) {
  CHECK_MODEL(m);
  if (!m->coupled) return fail(m, GB25_ERR_STATE, "no prescribed atmosphere: set all of its fields first (gb25_set_prescribed_atmosphere)");
  if (gb25_status s = materialize_uv(m)) return s;
  m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;   // (made from tendencies with the old fluxes)
  return atmosphere_ocean_fluxes_impl(m);
}
gb25_status gb25_compute_tendencies(gb25_model* m) {
  CHECK_MODEL(m);
  gb25_status s = momentum_impl(m);
  if (!s) s = tracers_impl(m);
  return s ? s : catke_tendency_impl(m);   // (closure = CATKE: the slow tendency of e)
}
gb25_status gb25_ab2_step(gb25_model* m, double dt, int euler) {
  CHECK_MODEL(m);
  if (m->slab) return fail(m, GB25_ERR_STATE, "gb25_ab2_step: phase-by-phase driving is for single-domain models");
  // ab2_step! leaves every halo as it is; look-ahead buffers written with their halos (fold_fills) are therefore not
  // adopted here: the stand-alone kernels give the same interior bits
  if (m->fold_fills) m->ahead_valid = false;
  return ab2_step_impl(m, dt, euler);
}
gb25_status gb25_correct_velocities_and_cache_previous_tendencies(gb25_model* m, double) {
  CHECK_MODEL(m);
  return corrector_impl(m);
}
gb25_status gb25_update_state(gb25_model* m) {
  CHECK_MODEL(m);
  if (m->slab) return fail(m, GB25_ERR_STATE, "gb25_update_state: phase-by-phase driving is for single-domain models");
  return update_state_impl(m);
}
// ---- options: every switch of the library is a per-model option; nothing is read from the environment
gb25_status gb25_set_option(gb25_model* m, gb25_option opt, int32_t v) {
  CHECK_MODEL(m);
  if (gb25_status s = collective_guard(m, 5, (unsigned)opt, (double)v)) return s;
  // a switch may change which buffers carry the next time level: whatever is in flight finishes, every look-ahead is void
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(hipStreamSynchronize(m->side_stream));
  if (m->baro_stream) HIPCHK(hipStreamSynchronize(m->baro_stream));
  if (m->group) HIPCHK(m->group->sync_side());
  m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
  m->tend_forkable = false;
  switch (opt) {
    case GB25_OPT_KERNELS:
      if (v != 1 && v != 2) return fail(m, GB25_ERR_INVALID_ARGUMENT, "GB25_OPT_KERNELS: 1 (direct stencil) or 2 (default)");
      if (v == 1 && m->immersed) return fail(m, GB25_ERR_STATE, "the direct-stencil kernels know no immersed boundary");
      m->kernel_gen = v;
      return GB25_OK;
    case GB25_OPT_AB2_LOOKAHEAD:
      if (v < 0 || v > 2) return fail(m, GB25_ERR_INVALID_ARGUMENT, "GB25_OPT_AB2_LOOKAHEAD: 0, 1 or 2 (tracers only)");
      m->ab2_ahead = v;
      return GB25_OK;
    case GB25_OPT_SUBCYCLE_LOOKAHEAD:
      if (v < 0 || v > 2) return fail(m, GB25_ERR_INVALID_ARGUMENT, "GB25_OPT_SUBCYCLE_LOOKAHEAD: 0, 1 (main stream) or 2 (own stream)");
      m->baro_ahead = v;
      return GB25_OK;
    case GB25_OPT_SUBCYCLE_BLOCK:
      if (v != 1 && v != 3 && v != 5 && v != 7)
        return fail(m, GB25_ERR_INVALID_ARGUMENT, "GB25_OPT_SUBCYCLE_BLOCK: 1, 3, 5 or 7 substeps per launch");
      m->baro_block = v;
      return GB25_OK;
    case GB25_OPT_FILL_FUSED: m->fill_fused = v != 0; return GB25_OK;
    case GB25_OPT_TWO_STREAMS: m->two_streams = v != 0; return GB25_OK;
    case GB25_OPT_STORE_PRESSURE: m->phy_pinned = v != 0; return GB25_OK;
    case GB25_OPT_SPLIT_TENDENCIES: m->split_tendencies = v != 0; return GB25_OK;
    case GB25_OPT_PRESSURE_PRECISION:
      if (v != 32 && v != 64) return fail(m, GB25_ERR_INVALID_ARGUMENT, "GB25_OPT_PRESSURE_PRECISION: 64 or 32");
      m->pressure_bits = (sizeof(real) == 8) ? 64 : v;   // (a Float64 model's own arithmetic IS fp64)
      return GB25_OK;
    case GB25_OPT_FOLD_FILLS:
      m->fold_fills = v != 0;
      m->complete_fills_needed = 2;
      return GB25_OK;
    case GB25_OPT_LAZY_CORRECTOR: m->lazy_corrector = v != 0; return GB25_OK;
    case GB25_OPT_TRACERS_FIRST: m->tracers_first = v != 0; return GB25_OK;
    case GB25_OPT_W_ON_THE_FLY: m->w_fly = v != 0; return GB25_OK;
    case GB25_OPT_SUB_STREAM_PRIORITY: m->sub_priority = v != 0; return GB25_OK;
    case GB25_OPT_SUBCYCLE_WHOLE: m->baro_whole = v != 0; return GB25_OK;
    case GB25_OPT_EARLY_STRIPS: m->early_strips = v != 0; return GB25_OK;
    case GB25_OPT_ROCTX_RANGES: m->roctx_ranges = v != 0; return GB25_OK;
    case GB25_OPT_SUBSTEP_ORDER:
      if (v != 0 && v != 1) return fail(m, GB25_ERR_INVALID_ARGUMENT, "substep_order: 0 (eta, then U, V) or 1 (U, V, then eta)");
      if (v == 1 && m->g.cv.on) return fail(m, GB25_ERR_STATE, "substep_order = 1 is built for the LatitudeLongitudeGrid kernels only (the oracle has it on every grid)");
      m->substep_order = v;
      return GB25_OK;
    case GB25_OPT_FOLD_PIVOT_SLAVED:
      if (v != 0 && !m->g.cv.north_fold) return fail(m, GB25_ERR_STATE, "fold_pivot_slaved: this grid has no zipper fold");
      if (v != 0 && m->slab) return fail(m, GB25_ERR_STATE, "fold_pivot_slaved is built for the single domain only (a slab's eastern half of the pivot row belongs to its partner rank)");
      m->g.cv.pivot_slaved = v != 0 ? 1 : 0;
      return GB25_OK;
    case GB25_OPT_COMM_TIMEOUT_SECONDS:
      if (v < 1) return fail(m, GB25_ERR_INVALID_ARGUMENT, "comm_timeout_seconds: at least 1");
      m->comm_timeout_s = v;
      return GB25_OK;
    case GB25_OPT_CATKE_STALE_E_HALOS:
      if (v != 0 && m->slab) return fail(m, GB25_ERR_STATE, "catke_stale_e_halos: a decomposition cannot leave the halos of e stale at its internal boundaries");
      m->catke_stale_e_halos = v != 0;
      return GB25_OK;
    case GB25_OPT_MOMENTUM_CHUNK_LEVELS:
    case GB25_OPT_TRACER_CHUNK_LEVELS:
      if (v < 6 || v > 4096) return fail(m, GB25_ERR_INVALID_ARGUMENT, "chunk levels: 6 or more");
      {
        // the tendency kernels reach one chunk of levels plus its stencil planes with 32-bit byte offsets (gb25_create checks the
        // default chunking): the chosen one must stay within that reach too
        const double plane = (double)(m->Nx + 2 * m->cfg.halo) * (m->Ny + 2 * m->cfg.halo + 1);
        const int kchunks = std::max(1, m->cfg.Nz / v), klen = (m->cfg.Nz + kchunks - 1) / kchunks;
        if (plane * (klen + 10) * sizeof(real) >= 2147483648.0)
          return fail(m, GB25_ERR_INVALID_ARGUMENT, "chunks of %d levels (+ 10 stencil planes of %.3g elements) exceed the 2 GB the tendency "
                      "kernels address from one base: choose fewer levels per chunk", klen, plane);
      }
      (opt == GB25_OPT_MOMENTUM_CHUNK_LEVELS ? m->mom_chunk_levels : m->trc_chunk_levels) = v;
      m->colsum_valid = false;
      return GB25_OK;
    case GB25_OPT_IMMERSED_KERNELS:
      // 1: run the immersed-boundary kernel variants even where nothing is immersed (they must then give the bits of
      // the plain ones: tests); 0: back to the choice the bottom makes
      if (v != 0 && m->kernel_gen == 1) return fail(m, GB25_ERR_STATE, "the direct-stencil kernels know no immersed boundary");
      if (v != 0 && !m->d_ord[0]) {
        if (m->cfg.Nz > 254) return fail(m, GB25_ERR_INVALID_ARGUMENT, "an immersed boundary needs Nz <= 254");
        if (gb25_status s = build_bottom(m, [](int, int) { return -1e30; })) return s;
      }
      {
        bool any = false;
        for (int kb : m->kbot) any = any || (kb > 0 && kb < 255);
        m->immersed = any || v != 0;
      }
      return GB25_OK;
    default: return fail(m, GB25_ERR_INVALID_ARGUMENT, "unknown option %d", (int)opt);
  }
}
gb25_status gb25_get_option(const gb25_model* m, gb25_option opt, int32_t* v) {
  if (!m || !v) return GB25_ERR_INVALID_ARGUMENT;
  switch (opt) {
    case GB25_OPT_KERNELS: *v = m->kernel_gen; break;
    case GB25_OPT_AB2_LOOKAHEAD: *v = m->ab2_ahead; break;
    case GB25_OPT_SUBCYCLE_LOOKAHEAD: *v = m->baro_ahead; break;
    case GB25_OPT_SUBCYCLE_BLOCK: *v = m->baro_block; break;
    case GB25_OPT_FILL_FUSED: *v = m->fill_fused; break;
    case GB25_OPT_TWO_STREAMS: *v = m->two_streams; break;
    case GB25_OPT_STORE_PRESSURE: *v = m->phy_pinned; break;
    case GB25_OPT_SPLIT_TENDENCIES: *v = m->split_tendencies; break;
    case GB25_OPT_PRESSURE_PRECISION: *v = m->pressure_bits; break;
    case GB25_OPT_IMMERSED_KERNELS: *v = m->immersed; break;
    case GB25_OPT_FOLD_FILLS: *v = m->fold_fills; break;
    case GB25_OPT_LAZY_CORRECTOR: *v = m->lazy_corrector; break;
    case GB25_OPT_MOMENTUM_CHUNK_LEVELS: *v = m->mom_chunk_levels; break;
    case GB25_OPT_TRACER_CHUNK_LEVELS: *v = m->trc_chunk_levels; break;
    case GB25_OPT_TRACERS_FIRST: *v = m->tracers_first; break;
    case GB25_OPT_W_ON_THE_FLY: *v = m->w_fly; break;
    case GB25_OPT_SUB_STREAM_PRIORITY: *v = m->sub_priority; break;
    case GB25_OPT_SUBCYCLE_WHOLE: *v = m->baro_whole; break;
    case GB25_OPT_EARLY_STRIPS: *v = m->early_strips; break;
    case GB25_OPT_CATKE_STALE_E_HALOS: *v = m->catke_stale_e_halos; break;
    case GB25_OPT_COMM_TIMEOUT_SECONDS: *v = m->comm_timeout_s; break;
    case GB25_OPT_ROCTX_RANGES: *v = m->roctx_ranges; break;
    case GB25_OPT_SUBSTEP_ORDER: *v = m->substep_order; break;
    case GB25_OPT_FOLD_PIVOT_SLAVED: *v = m->g.cv.pivot_slaved; break;
    default: return GB25_ERR_INVALID_ARGUMENT;
  }
  return GB25_OK;
}

// ---- exchange context of a decomposed model ----------------------------------------------------------------------
gb25_status gb25_comm_unique_id(void* id_out) {
  if (!id_out) return GB25_ERR_INVALID_ARGUMENT;
  static_assert(sizeof(ncclUniqueId) == GB25_UNIQUE_ID_BYTES, "gb25.h: GB25_UNIQUE_ID_BYTES");
  if (!rccl().load()) return GB25_ERR_COMM;
  ncclUniqueId id;
  if (rccl().GetUniqueId(&id) != ncclSuccess) return GB25_ERR_COMM;
  memcpy(id_out, &id, sizeof id);
  return GB25_OK;
}
gb25_status gb25_comm_init_rccl(gb25_model* m, const void* unique_id) {
  CHECK_MODEL(m);
  if (!unique_id) return GB25_ERR_INVALID_ARGUMENT;
  if (!m->slab) return fail(m, GB25_ERR_STATE, "a single periodic domain (nranks = 1, slab_mode = 0) has nothing to exchange");
  if (!rccl().load()) return fail(m, GB25_ERR_COMM, "%s", rccl().error.c_str());
  HIPCHK(hipSetDevice(m->cfg.device));
  RcclTransport* tr = new RcclTransport();
  tr->rank = m->cfg.rank;
  tr->nranks = m->cfg.nranks;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof id);
  (void)hipGetLastError();   // (RCCL reports a stale, already-handled HIP error of this process as "unhandled cuda error")
  // GB25_REHEARSE_ALONE=1: this rank of a decomposition runs alone, its neighbours are itself (a timing proxy, not a simulation)
  const char* solo = getenv("GB25_REHEARSE_ALONE");
  tr->alone = solo && solo[0] == '1' && tr->nranks > 1;
  // ncclCommInitRank blocks until every rank of the communicator has called it: a rank that never arrives (a crashed process, a
  // launcher that started fewer ranks than --gpus says) would leave the others waiting for ever.  It runs on a helper thread and
  // this one waits for it with a bound (option COMM_TIMEOUT_SECONDS); past it the call fails with the rank -- the helper thread
  // is abandoned, the process is expected to exit.
  ncclResult_t r = ncclInvalidArgument;
  if (!tr->alone) {
    struct InitState { std::mutex mu; std::condition_variable cv; bool done = false; ncclResult_t r = ncclSuccess; ncclComm_t comm = nullptr; };
    auto st = std::make_shared<InitState>();
    const int dev = m->cfg.device, nr = tr->nranks, rk = tr->rank;
    std::thread([st, id, dev, nr, rk]() {
      (void)hipSetDevice(dev);
      ncclComm_t c = nullptr;
      const ncclResult_t rr = rccl().CommInitRank(&c, nr, id, rk);
      std::lock_guard<std::mutex> lk(st->mu);
      st->r = rr; st->comm = c; st->done = true;
      st->cv.notify_all();
    }).detach();
    std::unique_lock<std::mutex> lk(st->mu);
    if (!st->cv.wait_for(lk, std::chrono::seconds(m->comm_timeout_s), [&] { return st->done; })) {
      delete tr;
      return fail(m, GB25_ERR_COMM, "rank %d of %d: ncclCommInitRank did not return within %d s -- not every rank of the communicator "
                  "called it (were all %d processes started, with the same unique id?)", rk, nr, m->comm_timeout_s, nr);
    }
    r = st->r;
    tr->comm = st->comm;
  }
  if (r != ncclSuccess && (tr->nranks == 1 || tr->alone)) {
    // a one-rank communicator (the self-ring) needs nobody's agreement: once more with a fresh id
    (void)hipGetLastError();
    if (rccl().GetUniqueId(&id) == ncclSuccess) r = rccl().CommInitRank(&tr->comm, 1, id, 0);
  }
  if (r != ncclSuccess) {
    tr->comm = nullptr;
    delete tr;
    return fail(m, GB25_ERR_COMM, "ncclCommInitRank(rank %d of %d) failed: %s", m->cfg.rank, m->cfg.nranks,
                rccl().GetErrorString(r));
  }
  gb25_model* one[1] = {m};
  if (gb25_status s = group_create(one, 1, tr)) return s;
  // The first exchange with every peer this rank will ever talk to -- the ring neighbours, the rows' neighbours of a 2-D
  // decomposition, the fold partner -- a few bytes each, bounded by the same timeout: a peer that built another decomposition (or
  // none) shows here, with the rank and the buffer set, instead of as a hang in the first time step.
  SlabGroup* G = m->group;
  for (int b : {0, 5, 3}) {
    if ((b == 5 && m->Ry < 2) || (b == 3 && !m->g.cv.north_fold)) continue;
    const size_t nbytes = std::min<size_t>(16, G->elems[b] * sizeof(real));
    if (gb25_status s = tr->exchange(*G, b, nbytes, G->main)) return s;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t q;
    while ((q = hipStreamQuery(G->main)) == hipErrorNotReady) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(m->comm_timeout_s))
        return fail(m, GB25_ERR_COMM, "rank %d of %d (%d x %d): the first exchange of buffer set %d (%s) did not complete within %d s",
                    m->cfg.rank, m->cfg.nranks, m->Rx, m->Ry, b, b == 0 ? "west / east ring neighbours" : b == 5 ? "southern / northern neighbours" : "fold partner",
                    m->comm_timeout_s);
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    if (q != hipSuccess) return fail(m, GB25_ERR_HIP, "the first exchange of buffer set %d failed: %s", b, hipGetErrorString(q));
  }
  return GB25_OK;
}
// What the exchange context of this model is: transport 0 none, 1 RCCL, 2 local ring (all slabs in this process), 3 host callback;
// comm_ranks: the size of the communicator AS RCCL REPORTS IT (ncclCommCount; 0 without RCCL) -- bench.py prints it so that a
// scaling line says how many ranks really talked to each other
gb25_status gb25_comm_info(const gb25_model* m, int32_t* transport, int32_t* comm_ranks) {
  if (!m) return GB25_ERR_INVALID_ARGUMENT;
  int t = 0, n = 0;
  if (m->group && m->group->transport) {
    const std::string name = m->group->transport->name();
    t = name == "rccl" ? 1 : name == "local" ? 2 : 3;
    if (t == 1) {
      RcclTransport* R = static_cast<RcclTransport*>(m->group->transport);
      if (R->comm && rccl().CommCount(R->comm, &n) != ncclSuccess) n = -1;
    }
  }
  if (transport) *transport = t;
  if (comm_ranks) *comm_ranks = n;
  return GB25_OK;
}
// The sends and receives ONE rank of an Rx x Ry decomposition posts for exchange group `group`, in posting order, as text:
// "send <peer> <side>" / "recv <peer> <side>" per line ("copy" when the rank is its own fold partner).  No GPU is touched: this is
// RcclTransport::exchange's own plan (exchange_plan).  folded_grid: the GLOBAL grid has a zipper fold (only the top row of ranks folds).
int64_t gb25_debug_exchange_plan(int32_t Rx, int32_t Ry, int32_t rank, int32_t folded_grid, int32_t group, char* out, int64_t cap) {
  if (Rx < 1 || Ry < 1 || rank < 0 || rank >= Rx * Ry) return -1;
  const MeshPos q(Rx, Ry, rank);
  const bool fold = folded_grid != 0 && q.ry == Ry - 1;
  const int b = buffer_set(group);
  std::string log;
  if (set_kind(b) == 1 && fold && q.partner() == rank) log = "copy\n";
  else if (!(set_kind(b) == 1 && !fold) && !(set_kind(b) == 2 && Ry < 2))
    for (const PlanOp& op : exchange_plan(q, fold, b)) {
      char line[48];
      snprintf(line, sizeof line, "%s %d %d\n", op.send ? "send" : "recv", op.peer, op.side);
      log += line;
    }
  const int64_t need = (int64_t)log.size() + 1;
  if (out && cap > 0) {
    const int64_t ncopy = std::min<int64_t>(cap - 1, (int64_t)log.size());
    memcpy(out, log.data(), (size_t)ncopy);
    out[ncopy] = 0;
  }
  return need;
}
gb25_status gb25_comm_init_callback(gb25_model* m, gb25_exchange_fn fn, void* user) {
  CHECK_MODEL(m);
  if (!fn) return GB25_ERR_INVALID_ARGUMENT;
  CallbackTransport* tr = new CallbackTransport();
  tr->fn = fn;
  tr->user = user;
  gb25_model* one[1] = {m};
  return group_create(one, 1, tr);
}
gb25_status gb25_comm_init_local(gb25_model* const* slabs, int32_t n) {
  if (!slabs || n < 1 || !slabs[0]) return GB25_ERR_INVALID_ARGUMENT;
  gb25_model* m = slabs[0];
  for (int s = 0; s < n; s++) {
    if (!slabs[s]) return GB25_ERR_INVALID_ARGUMENT;
    const gb25_config &a = slabs[s]->cfg, &b = m->cfg;
    if (a.nranks != n || a.rank != s || a.device != b.device || a.Nx != b.Nx || a.Ny != b.Ny || a.Nz != b.Nz || a.ranks_y != b.ranks_y ||
        a.halo != b.halo || a.substeps != b.substeps)
      return fail(m, GB25_ERR_INVALID_ARGUMENT,
                  "gb25_comm_init_local wants the %d slabs of ONE decomposition in rank order on one device (slab %d does "
                  "not fit)", n, s);
  }
  return group_create(slabs, n, new LocalRingTransport());
}
gb25_status gb25_comm_finalize(gb25_model* m) {
  CHECK_MODEL(m);
  if (m->group) group_destroy(m->group);
  return GB25_OK;
}

static gb25_status need_group(gb25_model* m, const char* what) {
  if (m->group) return group_refresh(m->group);   // (the bundles grow when a closure adds fields)
  return fail(m, GB25_ERR_STATE,
              "%s on a slab of a decomposition needs an exchange context first: gb25_comm_init_rccl (one process per GPU), "
              "gb25_comm_init_local (all slabs in this process) or gb25_comm_init_callback", what);
}

// ---- composites: GordonBell25.first_time_step!/time_step!/loop! (src/timestepping_utils.jl:21-45) ----------------------
// On a slab these step every slab of the exchange context in lock-step (one slab per process with RCCL).
gb25_status gb25_first_time_step(gb25_model* m) {
  CHECK_MODEL(m);
  Range r_first(m, "first_time_step");
  gb25_status s;
  if (m->slab) {
    if ((s = need_group(m, "gb25_first_time_step"))) return s;
    GroupOps ops(*m->group);
    return sequence_first_time_step(ops, m->group->lookahead_in_flight);
  }
  if ((s = initialize_impl(m))) return s;
  if ((s = update_state_impl(m))) return s;
  if ((s = first_fluxes_impl(m))) return s;
  return time_step_impl(m, 1);
}
static gb25_status loop_impl(gb25_model* m, int32_t n);
gb25_status gb25_time_step(gb25_model* m) {
  CHECK_MODEL(m);
  Range r_step(m, "time_step");
  return loop_impl(m, 1);
}
gb25_status gb25_loop(gb25_model* m, int32_t n) {
  CHECK_MODEL(m);
  Range r_loop(m, "loop");
  return loop_impl(m, n);
}
static gb25_status loop_impl(gb25_model* m, int32_t n) {
  if (m->slab) {
    if (gb25_status s = need_group(m, "gb25_loop")) return s;
    GroupOps ops(*m->group);
    for (int it = 0; it < n; it++)
      if (gb25_status s = sequence_time_step(ops, 0, m->group->lookahead_in_flight)) return s;
    // (steps that kept the corrector inside its consumers: memory holds the corrected velocities when the call returns)
    for (gb25_model* q : m->group->slabs)
      if (gb25_status s = materialize_uv(q)) return fail(m, s, "%s", q->err.c_str());
    return GB25_OK;
  }
  // (every step of the call may keep the corrector inside its consumers, the last one too: one sweep u += du, v += dv and one w
  // kernel when the call returns -- a last step with the stand-alone corrector would first materialise the step before it, then
  // run its own corrector and w: 0.75 ms more than a steady step at 1440x720x48, against 0.37)
  for (int it = 0; it < n; it++)
    if (gb25_status s = time_step_impl(m, 0, true)) return s;
  return materialize_uv(m);   // (no-op unless the last step kept the corrector inside its consumers)
}

gb25_status gb25_lookahead_state(const gb25_model* m, int32_t* velocities_ready, int32_t* subcycle_adopted) {
  if (!m) return GB25_ERR_INVALID_ARGUMENT;
  if (velocities_ready) *velocities_ready = (m->ahead_uv_valid && m->baro_ahead && !m->ptr_exposed) ? 1 : 0;
  if (subcycle_adopted) *subcycle_adopted = m->baro_adopted ? 1 : 0;
  return GB25_OK;
}

// The order of operations of one (first) time step of `nslabs` slabs as text, one operation per line, without touching
// a GPU: "stage <n> slab <s> euler <e> <stream>", "pack|unpack <group> slab <s> <stream>", "exchange <group> <stream>",
// "record|wait <event> <stream>".  adopted / ready: what the slabs would report (sub-cycle look-ahead adopted by stage 0;
// velocity look-ahead available after stage 3).  Returns the number of bytes needed (incl. the terminator).
int64_t gb25_debug_sequence(int32_t nslabs, int32_t first, int32_t adopted, int32_t ready, char* out, int64_t cap) {
  if (nslabs < 1) return -1;
  TraceOps ops(nslabs, adopted != 0, ready != 0);
  if (first & 2) ops.fold = true;   // (bit 1 of `first`: a folded grid)
  if (first & 4) ops.is_coupled = true;   // (bit 2: a coupled model -- data-free forcing)
  if (first & 64) ops.early = true;       // (bit 6: plain x slabs whose bundle is unpacked on the exchange stream)
  if (first & 32) ops.is_lazy = true;     // (bit 5: a step that keeps the corrector inside its consumers)
  if (first & 16) ops.mesh = true;        // (bit 4: a 2-D decomposition -- y halos from the southern / northern neighbour)
  if (first & 128) ops.is_catke = true;   // (bit 7: closure = CATKE -- the halos of e and J^b travel inside update_state!)
  bool in_flight = (first & 8) != 0;      // (bit 3: the previous step left the look-ahead chain in flight)
  first &= 1;
  if (first) sequence_first_time_step(ops, in_flight);
  else sequence_time_step(ops, 0, in_flight);
  ops.add("lookahead_in_flight %d", in_flight ? 1 : 0);
  const int64_t need = (int64_t)ops.log.size() + 1;
  if (out && cap > 0) {
    const int64_t ncopy = std::min<int64_t>(cap - 1, (int64_t)ops.log.size());
    memcpy(out, ops.log.data(), (size_t)ncopy);
    out[ncopy] = 0;
  }
  return need;
}

// ---- state dump: save_model_state(dir, model, arch; label) (src/sharded_io.jl:122-138) -- see state_io.hpp
gb25_status gb25_save_state(gb25_model* m, const char* directory, const char* label) {
  CHECK_MODEL(m);
  if (!directory || !*directory) return GB25_ERR_INVALID_ARGUMENT;
  const std::string dir = std::string(directory) + "/" + (label && *label ? label : "checkpoint");
  for (size_t q = 1; q <= dir.size(); q++)   // mkpath
    if (q == dir.size() || dir[q] == '/') mkdir(dir.substr(0, q).c_str(), 0777);
  const std::string path = dir + "/fields_rank" + std::to_string(m->cfg.rank) + ".npz";
  NpzWriter z;
  if (!z.open(path)) return fail(m, GB25_ERR_STATE, "cannot write %s", path.c_str());
  const int64_t meta[4] = {m->iteration, (int64_t)m->cfg.rank, (int64_t)m->cfg.nranks, (int64_t)sizeof(real)};
  z.add_array("iteration", "<i8", {}, &meta[0]);
  z.add_array("time", "<f8", {}, &m->time);
  z.add_array("rank", "<i8", {}, &meta[1]);
  z.add_array("nranks", "<i8", {}, &meta[2]);
  // Oceananigans.fields(model): velocities, free surface, tracers -- (T, S), and e with closure = CATKEVerticalDiffusivity()
  struct Member { const char* name; gb25_field id; };
  std::vector<Member> fs = {{"u", GB25_U}, {"v", GB25_V}, {"w", GB25_W}, {"eta", GB25_ETA}, {"T", GB25_T}, {"S", GB25_S}};
  if (m->catke) fs.push_back({"e", GB25_E});
  std::string names;
  std::vector<real> host;
  for (auto& fld : fs) {
    int32_t d[3];
    gb25_field_dims(m, fld.id, 0, d);
    host.resize((size_t)d[0] * d[1] * d[2]);
    if (gb25_status s = gb25_get_field(m, fld.id, host.data(), 0)) {
      z.close();
      return s;
    }
    const int64_t i0 = (int64_t)m->rx * m->Nx, j0 = m->j0;
    // (global rows: Ny, + 1 for a y-face field of a grid with a northern wall)
    const int64_t gny = (int64_t)m->cfg.Ny + ((is_v_shaped(fld.id) && m->cfg.grid_type < GB25_GRID_TRIPOLAR) ? 1 : 0);
    const int64_t slice[6] = {i0, i0 + d[0], j0, j0 + d[1], 0, d[2]}, gshape[3] = {m->cfg.Nx, m->Ry > 1 ? gny : d[1], d[2]};
    z.add_array(std::string(fld.name) + ".data", sizeof(real) == 8 ? "<f8" : "<f4", {d[0], d[1], d[2]}, host.data());
    z.add_array(std::string(fld.name) + ".slice", "<i8", {6}, slice);
    z.add_array(std::string(fld.name) + ".global_shape", "<i8", {3}, gshape);
    names += std::string(fld.name) + "\n";
  }
  z.add("field_names", names, nullptr, 0);
  if (!z.close()) return fail(m, GB25_ERR_STATE, "writing %s failed (disk full, or a member beyond the 4 GB of zip32)", path.c_str());
  return GB25_OK;
}

// ---- profiling ------------------------------------------------------------------------------
gb25_status gb25_profile_enable(gb25_model* m, int on) {
  CHECK_MODEL(m);
  m->profile = on != 0;
  m->profile_only = on >= 2 ? on - 2 : -1;   // on = 2 + k: only kernel k
  if (m->profile_only >= GB25_K_COUNT) return fail(m, GB25_ERR_INVALID_ARGUMENT, "no such kernel id %d", on - 2);
  return GB25_OK;
}
gb25_status gb25_profile_reset(gb25_model* m) {
  CHECK_MODEL(m);
  resolve_profile(m);
  for (int k = 0; k < GB25_K_COUNT; k++) {
    m->prof_count[k] = 0;
    m->prof_ms[k] = 0;
  }
  return GB25_OK;
}
gb25_status gb25_profile_get(gb25_model* m, gb25_kernel k, int64_t* launches, double* total_ms) {
  if (!m || k < 0 || k >= GB25_K_COUNT) return GB25_ERR_INVALID_ARGUMENT;
  resolve_profile(m);
  if (launches) *launches = m->prof_count[k];
  if (total_ms) *total_ms = m->prof_ms[k];
  return GB25_OK;
}

}  // extern "C"

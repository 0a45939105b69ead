// gb25_api.hip -- host side of libgb25hip.so: the C ABI declared in include/gb25.h.
//
// Owns the device state of one model (one x-slab on one GPU), builds the grid metrics,
// sequences the kernels of kernels.hpp in the reference's phase order
// (GB-25 src/precompile.jl:31-42) and exposes the per-phase entry points.
// There is no CPU fallback: without a HIP device gb25_create fails with GB25_ERR_NO_DEVICE.
#include "../../include/gb25.h"
#include "kernels.hpp"
#include "kernels_v2.hpp"

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
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

// Everything a time step changes on the HOST side (the device work is in the kernels): which buffer carries which
// name after the pointer exchanges, and the validity flags of the cached by-products.
struct HostState {
  // G^n/G^- of u,v,T,S (8); T, S and partners (4); column integrals and partners (4); u, v and partners (4);
  // G.U, G.V and partners (4)
  real* ptr[24];
  bool ahead_valid, ahead_uv_valid, colsum_valid;
  real ahead_dt, ahead_chi, ahead_uv_dt, ahead_uv_chi;
  bool operator==(const HostState& o) const {
    for (int q = 0; q < 24; q++)
      if (ptr[q] != o.ptr[q]) return false;
    return ahead_valid == o.ahead_valid && ahead_uv_valid == o.ahead_uv_valid && colsum_valid == o.colsum_valid &&
           (!ahead_valid || (ahead_dt == o.ahead_dt && ahead_chi == o.ahead_chi)) &&
           (!ahead_uv_valid || (ahead_uv_dt == o.ahead_uv_dt && ahead_uv_chi == o.ahead_uv_chi));
  }
};
// One captured time step: valid when the model is in state `pre` with the same dt and stream; leaves it in `post`.
struct StepGraph {
  HostState pre, post;
  double dt;
  hipStream_t stream;
  hipGraphExec_t exec;
  hipGraph_t graph;
};

}  // namespace

struct gb25_model {
  gb25_config cfg;
  int Nx = 0;  // local slab width
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
  int baro_ahead = 1;                // GB25_BARO_AHEAD=0: sub-cycle inside the step, on the critical path
  hipEvent_t ev_baro = nullptr, ev_mom = nullptr;
  int use_graphs = 0;                // GB25_GRAPH=1: replay a captured HIP graph of the step (see step_with_graph)
  std::vector<StepGraph> graphs;
  std::vector<HostState> seen;       // states met once: a state is captured when it comes round again
  int fill_fused = 1;                // y, z and periodic-x fills of a single slab in one launch (GB25_FILL_FUSED=0: two)
  int ab2_ahead = 1;                 // GB25_AB2_AHEAD=0: always run the stand-alone tracer AXPY kernel
  real* bars = nullptr;         // contiguous etabar | Ubar | Vbar
  std::vector<real*> dev_tables;
  std::vector<double> h_metric[11];
  int metric_off_j = 0, metric_off_k = 0;
  // substepping
  int Ns = 0;
  double dtau_frac = 0;
  std::vector<double> weights;
  // wide-halo barotropic work arrays (slab mode)
  int W = 0;
  Field wide[2][3];  // [pingpong][eta,U,V]
  Field wideG[2];    // GU, GV
  Field wideBar[3];  // running averages on the wide domain
  // clock
  double time = 0, last_dt = 0;
  int64_t iteration = 0;
  // streams / timing
  hipStream_t own_stream = nullptr, stream = nullptr, side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool two_streams = true;          // GB25_TWO_STREAMS=0: strictly sequential phases on one stream
  bool profile = false;
  int profile_only = -1;             // >= 0: time this kernel id alone (keeps the event records out of the other launches)
  std::vector<EventPair> pending[GB25_K_COUNT];
  std::vector<EventPair> free_events;
  int64_t prof_count[GB25_K_COUNT] = {0};
  double prof_ms[GB25_K_COUNT] = {0};
  std::string err;
  int baro_rows = 16;                // tile height of the blocked barotropic kernel: 16 or 32 (GB25_BARO_ROWS)
  int baro_block = 7;             // substeps per barotropic launch (GB25_BARO_BLOCK=1: one launch per substep)
  int momentum_v5 = 1;               // packed (G_u term, G_v term) reconstructions; GB25_MOMENTUM_V5=0: scalar v2 kernel
  int momentum_v4 = 0;               // GB25_MOMENTUM_V4=1: single-barrier pipelined momentum kernel
  int tracer_v3 = 5;                 // 5: packed (T,S) wave-autonomous kernel, buffer addressing; 1: scalar v3; 0: LDS v2 (GB25_TRACER_V3)
  int tile_rows = 4;                 // rows (= waves) per block of the LDS tendency kernels: 4 or 8 (GB25_TILE_ROWS); 4 wins with the packed kernel: more blocks per CU to cover the barriers
  int variant_c = 1;                 // nontemporal tendency reads in the tracer AB2 stream (GB25_VARIANT_C)
  int variant_a = 1, variant_b = 1;  // tuning switches (GB25_VARIANT_A / _B), see momentum_impl / tracers_impl
  int kernel_gen = 2;  // 2: LDS flux-sharing tendency kernels (kernels_v2.hpp); 1: direct-stencil kernels (GB25_KERNELS=v1)
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
         id == GB25_GN_BT_V;
}
bool is_2d(int id) { return id >= GB25_ETA; }

// --- profiling helpers -----------------------------------------------------------------------
struct Timed {
  gb25_model* m;
  int k;
  EventPair ev;
  bool on;
  Timed(gb25_model* m_, int k_) : m(m_), k(k_), on(m_->profile && (m_->profile_only < 0 || m_->profile_only == k_)) {
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
  const int H = c.halo, Ny = c.Ny, Nz = c.Nz;
  const int nj = Ny + 2 * H + 2 * PAD + 2, nk = Nz + 2 * H + 2 * PAD + 2;
  const int offj = H + PAD, offk = H + PAD;  // table index of 0-based logical index 0
  m->metric_off_j = offj;
  m->metric_off_k = offk;
  const double d2r = M_PI / 180.0;
  const double dlam = (c.lon_east - c.lon_west) / c.Nx, dphi = (c.lat_north - c.lat_south) / Ny, R = c.radius;
  std::vector<double>&phif = m->h_metric[GB25_M_PHIF], &phic = m->h_metric[GB25_M_PHIC],
  &dxc = m->h_metric[GB25_M_DXC], &dxf = m->h_metric[GB25_M_DXF], &azc = m->h_metric[GB25_M_AZC],
  &azf = m->h_metric[GB25_M_AZF], &fcor = m->h_metric[GB25_M_FCOR];
  phif.assign(nj, 0); phic.assign(nj, 0); dxc.assign(nj, 0); dxf.assign(nj, 0);
  azc.assign(nj, 0); azf.assign(nj, 0); fcor.assign(nj, 0);
  for (int a = 0; a < nj; a++) {
    int j = a - offj;  // 0-based face / centre index
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
  // vertical faces: z_k ~ exp(k/h), k = 1..Nz+1, mapped to [0, -depth] and reversed
  std::vector<double> zint(Nz + 1);
  const double h = c.zexp_h, e1 = std::exp(1.0 / h), eN = std::exp((Nz + 1.0) / h);
  for (int k = 1; k <= Nz + 1; k++) zint[Nz + 1 - k] = -c.depth * (std::exp(k / h) - e1) / (eN - e1);
  zint[Nz] = 0.0;
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
  g.x_periodic = (c.nranks == 1);
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
    g.rdy = (real)(1.0 / (R * dphi * d2r));
    g.rLz = (real)(1.0 / (zint[Nz] - zint[0]));
  }
  if ((s = upload_table(m, zc, offk, &g.zc))) return s;
  if ((s = upload_table(m, dzc, offk, &g.dzc))) return s;
  if ((s = upload_table(m, dzf, offk, &g.dzf))) return s;
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
  const std::vector<double>&zc = m->h_metric[GB25_M_ZC], &dzf = m->h_metric[GB25_M_DZF];
  std::vector<double> tab((size_t)28 * (Nz + 1), 0.0), dz(Nz + 1);
  for (int k = 0; k <= Nz; k++) {
    // geopotential height of level k; the halo level above the surface is mirrored (Oceananigans Z^ccc)
    double Z = (k < Nz) ? zc[offk + k] : zc[offk + Nz - 1] - dzf[offk + Nz - 1];
    double zeta = -Z * 1e-4;
    double* c = &tab[(size_t)28 * k];
    for (const EosTerm& t : kEos) {
      int base = 0;
      for (int j = 0; j < t.j; j++) base += 7 - j;
      c[base + t.i] += t.v * std::pow(zeta, t.k);
    }
    double r0 = 0;
    for (int q = 5; q >= 0; q--) r0 = (r0 + kEosR0[q]) * zeta;
    c[0] += r0 - m->cfg.rho0;
    dz[k] = dzf[offk + k];
  }
  double* d = nullptr;
  HIPCHK(hipMalloc(&d, (tab.size() + dz.size()) * sizeof(double)));
  HIPCHK(hipMemcpy(d, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(d + tab.size(), dz.data(), dz.size() * sizeof(double), hipMemcpyHostToDevice));
  m->dev_tables.push_back(reinterpret_cast<real*>(d));
  m->g.eos = d;
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

gb25_status alloc_field(gb25_model* m, Field& F, int nx, int ny, int nz) {
  F.nx = nx; F.ny = ny; F.nz = nz;
  hipError_t e = hipMalloc(&F.d, F.elems() * sizeof(real));
  if (e != hipSuccess)
    return fail(m, GB25_ERR_OUT_OF_MEMORY, "hipMalloc of %zu bytes failed: %s", F.elems() * sizeof(real),
                hipGetErrorString(e));
  HIPCHK(hipMemset(F.d, 0, F.elems() * sizeof(real)));
  return GB25_OK;
}

inline dim3 grid2(int nx, int ny, dim3 b) { return dim3((nx + b.x - 1) / b.x, (ny + b.y - 1) / b.y); }

// sel: 3 = u, v, T, S;  1 = u, v;  2 = T, S
Halo3 halo3(gb25_model* m, int sel = 3) {
  Halo3 h{};
  int n = 0;
  if (sel & 1) {
    h.p[n] = m->f[GB25_U].d; h.is_v[n++] = 0;
    h.p[n] = m->f[GB25_V].d; h.is_v[n++] = 1;
  }
  if (sel & 2) {
    h.p[n] = m->f[GB25_T].d; h.is_v[n++] = 0;
    h.p[n] = m->f[GB25_S].d; h.is_v[n++] = 0;
  }
  h.n = n;
  return h;
}
Halo2 halo2_prognostic(gb25_model* m) {
  Halo2 h;
  h.p[0] = m->f[GB25_ETA].d; h.is_v[0] = 0;
  h.p[1] = m->f[GB25_BT_U].d; h.is_v[1] = 0;
  h.p[2] = m->f[GB25_BT_V].d; h.is_v[2] = 1;
  h.n = 3;
  return h;
}

// y/z boundary layers (always local) and, for a single slab, the periodic x copy.
// extended: also treat the x-halo columns (slab mode, after the neighbours' columns were unpacked).
// which: 3 = 3-D and 2-D fields, 1 = the 3-D bundle only, 2 = the 2-D fields only (slab pipeline).
// sel3: which 3-D fields (halo3); st: stream (nullptr = the model's stream).
gb25_status fill_halos_impl(gb25_model* m, bool with_x, bool extended = false, int which = 3, int sel3 = 3,
                            hipStream_t st = nullptr, bool use_default_stream = true, const Halo2* h2_other = nullptr) {
  const Grid& g = m->g;
  if (use_default_stream) st = m->stream;
  Timed t(m, GB25_K_FILL_HALOS);
  Halo3 h3 = halo3(m, sel3);
  Halo2 h2 = h2_other ? *h2_other : halo2_prognostic(m);
  dim3 b(256);
  const int i0 = extended ? -g.H : 0, ni = extended ? g.Nx + 2 * g.H : g.Nx;
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
  hipLaunchKernelGGL(k_fill_yz, dim3((ni + 255) / 256, g.Nz + 1 + g.Ny), b, 0, st, g, h3, h2, i0, ni);
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
  if (g.x_periodic) {
    long threads = (long)g.sy_v * 2 * g.H;
    hipLaunchKernelGGL(k_fill_x, dim3((unsigned)((threads + 255) / 256), 4 + h2.n), b, 0, m->stream, g2, none, h2, 0,
                       0);
  }
  LAUNCHCHK();
  return GB25_OK;
}

gb25_status compute_w_impl(gb25_model* m) {
  const Grid& g = m->g;
  Timed t(m, GB25_K_COMPUTE_W);
  dim3 b(64, 4);
  int ex = g.Nx + 2 * g.H - 2, ey = g.Ny + 2 * g.H - 2;
  hipLaunchKernelGGL(k_compute_w, grid2(ex, ey, b), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                     m->f[GB25_W].d);
  LAUNCHCHK();
  return GB25_OK;
}
// Hydrostatic pressure on columns [i_first, i_last] (default: the whole extended range -H+1 .. Nx+H-2; column
// i_first - 1 is read as the west neighbour of the first x difference), optionally on a second range as well.
gb25_status compute_p_impl(gb25_model* m, int i_first = INT_MIN, int i_last = INT_MIN, int i_first_b = 0,
                           int i_last_b = -1, bool may_skip_p = false) {
  const Grid& g = m->g;
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
  const int nrow = g.Ny + 2 * g.H - 2;     // rows -H+1 .. Ny+H-2
  const int tiles_a = (ncol + 62) / 63, tiles_b = (ncol_b + 62) / 63;
  if (ncol <= 63) {   // strips: one row per thread (4x the waves, short chains)
    dim3 gr(tiles_a + tiles_b, (nrow + 3) / 4);
    auto kern = write_p ? k_compute_p<1, true> : k_compute_p<1, false>;
    hipLaunchKernelGGL(kern, gr, b, 0, m->stream, g, m->f[GB25_T].d, m->f[GB25_S].d, m->f[GB25_PHY].d, m->dpx.d,
                       m->dpy.d, i_first, i_last, i_first_b, i_last_b, tiles_a);
  } else {
    dim3 gr(tiles_a + tiles_b, (nrow + PR * 4 - 1) / (PR * 4));
    auto kern = write_p ? k_compute_p<PR, true> : k_compute_p<PR, false>;
    hipLaunchKernelGGL(kern, gr, b, 0, m->stream, g, m->f[GB25_T].d, m->f[GB25_S].d, m->f[GB25_PHY].d, m->dpx.d,
                       m->dpy.d, i_first, i_last, i_first_b, i_last_b, tiles_a);
  }
  LAUNCHCHK();
  return GB25_OK;
}

void tile_grid(const Grid& g, int* nbx, int* nb) {
  *nbx = (g.Nx + TX - 1) / TX;
  int nby = (g.Ny + TY - 1) / TY;
  *nb = *nbx * nby * g.Nz;
}

gb25_status momentum_impl(gb25_model* m) {
  const Grid& g = m->g;
  int nbx, nb;
  m->ahead_uv_valid = m->ahead_baro_valid = false;   // look-aheads made from the previous tendencies are void
  if (m->kernel_gen >= 2) {
    Timed t(m, GB25_K_GU);   // the fused G_u + G_v kernel is accounted under the "gu" timer
    nbx = (g.Nx + V2_TX - 1) / V2_TX;
    const int TY = m->tile_rows;
    const int nby = (g.Ny + TY - 1) / TY;
    const int kchunks = std::max(1, g.Nz / 12);
    nb = nbx * nby * kchunks;
    // waves/SIMD the register allocator is held to: 4 (128 VGPRs) in fp32; fp64 operands are register pairs,
    // so the Float64 build asks for 2 (256 VGPRs) instead of spilling
    constexpr int MW = sizeof(real) == 8 ? 2 : 4;
    auto kern = TY == 4 ? (m->variant_b ? k_momentum_tendencies_v2<MW, 4> : k_momentum_tendencies_v2<2, 4>)
                        : (m->variant_b ? k_momentum_tendencies_v2<MW, 8> : k_momentum_tendencies_v2<2, 8>);
    if (m->momentum_v4) kern = k_momentum_tendencies_v4<MW, 8>;
    if (m->momentum_v5) {
      const bool ahead = m->ab2_ahead && !m->ptr_exposed && m->ab2_ahead != 2;   // GB25_AB2_AHEAD=2: tracers only
      UvAhead nx{};
      const real dt = (real)m->last_dt, chi = (real)m->cfg.chi;
      if (ahead) {   // predicted parameters of the next ab2_step!: the clock's dt and the model's chi
        nx.GmU = m->f[GB25_GM_U].d; nx.GmV = m->f[GB25_GM_V].d;
        nx.un = m->ahead_uv[0].d; nx.vn = m->ahead_uv[1].d;
        nx.P = m->uv_partials;
        nx.dt = dt; nx.C1 = real(1.5) + chi; nx.C2 = real(0.5) + chi;
        nx.plane2 = g.sx * g.sy_v;
      }
      auto k5 = TY == 4 ? (ahead ? k_momentum_tendencies_v5<MW, 4, true> : k_momentum_tendencies_v5<MW, 4, false>)
                        : (ahead ? k_momentum_tendencies_v5<MW, 8, true> : k_momentum_tendencies_v5<MW, 8, false>);
      hipLaunchKernelGGL(k5, dim3(nb), dim3(V2_TX, TY), 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                         m->f[GB25_W].d, m->dpx.d, m->dpy.d, m->f[GB25_GN_U].d, m->f[GB25_GN_V].d, nbx, kchunks, nb,
                         nx);
      t.stop();   // the timer covers the tendency kernel alone
      if (ahead) {
        dim3 b(64, 4);
        hipLaunchKernelGGL(k_ab2_velocities_finish, grid2(g.Nx, g.Ny, b), b, 0, m->stream, g, m->uv_partials, kchunks,
                           nx.plane2, m->ahead_G[0].d, m->ahead_G[1].d, m->ahead_colsum[0].d, m->ahead_colsum[1].d);
        m->ahead_uv_valid = true;
        m->ahead_uv_dt = dt;
        m->ahead_uv_chi = chi;
      }
      LAUNCHCHK();
      return GB25_OK;
    }
    hipLaunchKernelGGL(kern, dim3(nb), dim3(V2_TX, TY), 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                       m->f[GB25_W].d, m->dpx.d, m->dpy.d, m->f[GB25_GN_U].d, m->f[GB25_GN_V].d, nbx, kchunks, nb);
    LAUNCHCHK();
    return GB25_OK;
  }
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
  if (m->kernel_gen >= 2 && m->tracer_v3) {
    Timed t(m, GB25_K_TRACERS);
    nbx = (g.Nx + V3_OUT - 1) / V3_OUT;
    const int nby = (g.Ny + 3) / 4;
    const int kchunks = std::max(1, g.Nz / 12);
    nb = nbx * nby * kchunks;
    constexpr int TW = sizeof(real) == 8 ? 3 : 5;   // see MW in momentum_impl
    const bool ahead = m->ab2_ahead && !m->ptr_exposed;
    Ab2Ahead nx{};
    if (ahead) {   // predicted parameters of the next ab2_step!: the clock's dt and the model's chi
      nx.GmT = m->f[GB25_GM_T].d; nx.GmS = m->f[GB25_GM_S].d;
      nx.Tn = m->ahead[0].d; nx.Sn = m->ahead[1].d;
      nx.dt = (real)m->last_dt;
      nx.C1 = real(1.5) + (real)m->cfg.chi; nx.C2 = real(0.5) + (real)m->cfg.chi;
    }
    auto kern = m->variant_a == 6 ? k_tracer_tendencies_v3<6, false>
                                  : (m->variant_a == 7 ? k_tracer_tendencies_v3<7, false>
                                                       : (ahead ? k_tracer_tendencies_v3<TW, true>
                                                                : k_tracer_tendencies_v3<TW, false>));
    if (m->tracer_v3 == 5) {   // packed (T, S) pairs
      kern = ahead ? k_tracer_tendencies_v5<TW, true> : k_tracer_tendencies_v5<TW, false>;
      if (m->variant_a == 4) kern = ahead ? k_tracer_tendencies_v5<4, true> : k_tracer_tendencies_v5<4, false>;
    }
    const bool ahead_run = ahead && ((m->variant_a != 6 && m->variant_a != 7) || m->tracer_v3 == 5);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(64, 4), 0, m->stream, g, m->f[GB25_U].d,
                       m->f[GB25_V].d, m->f[GB25_W].d, m->f[GB25_T].d, m->f[GB25_S].d, m->f[GB25_GN_T].d,
                       m->f[GB25_GN_S].d, nbx, kchunks, nb, nx);
    LAUNCHCHK();
    m->ahead_valid = ahead_run;
    m->ahead_dt = nx.dt;
    m->ahead_chi = (real)m->cfg.chi;
    return GB25_OK;
  }
  m->ahead_valid = false;   // only the v3 kernel looks ahead
  if (m->kernel_gen >= 2) {
    Timed t(m, GB25_K_TRACERS);
    nbx = (g.Nx + V2_TX - 1) / V2_TX;
    const int TY = m->tile_rows;
    const int nby = (g.Ny + TY - 1) / TY;
    const int kchunks = std::max(1, g.Nz / 12);   // >= 12 levels per block: the z-carry start-up stays ~3 %
    nb = nbx * nby * kchunks;
    auto kern = TY == 4 ? (m->variant_a ? k_tracer_tendencies_v2<true, 4> : k_tracer_tendencies_v2<false, 4>)
                        : (m->variant_a ? k_tracer_tendencies_v2<true, 8> : k_tracer_tendencies_v2<false, 8>);
    hipLaunchKernelGGL(kern, dim3(nb), dim3(V2_TX, TY), 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                       m->f[GB25_W].d, m->f[GB25_T].d, m->f[GB25_S].d, m->f[GB25_GN_T].d, m->f[GB25_GN_S].d, nbx,
                       kchunks, nb);
    LAUNCHCHK();
    return GB25_OK;
  }
  tile_grid(g, &nbx, &nb);
  Timed t(m, GB25_K_TRACERS);
  hipLaunchKernelGGL(k_tracer_tendencies, dim3(nb), dim3(TX, TY), 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                     m->f[GB25_W].d, m->f[GB25_T].d, m->f[GB25_S].d, m->f[GB25_GN_T].d, m->f[GB25_GN_S].d, nbx, nb);
  LAUNCHCHK();
  return GB25_OK;
}

gb25_status ab2_velocities_impl(gb25_model* m, real dt, real chi) {
  const Grid& g = m->g;
  if (m->ahead_uv_valid && dt == m->ahead_uv_dt && chi == m->ahead_uv_chi) {
    // the last momentum evaluation already advanced u and v with exactly these parameters: adopt its buffers
    for (int q = 0; q < 2; q++) {
      std::swap(m->f[GB25_U + q].d, m->ahead_uv[q].d);
      std::swap(m->f[GB25_GN_BT_U + q].d, m->ahead_G[q].d);
      std::swap(m->colsum[q].d, m->ahead_colsum[q].d);
    }
    m->colsum_valid = true;
    m->ahead_uv_valid = false;
    return GB25_OK;
  }
  m->ahead_uv_valid = false;
  dim3 b(64, 4);
  Timed t(m, GB25_K_AB2_VELOCITIES);
  hipLaunchKernelGGL(k_ab2_velocities, grid2(g.Nx, g.Ny, b), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                     m->f[GB25_GN_U].d, m->f[GB25_GM_U].d, m->f[GB25_GN_V].d, m->f[GB25_GM_V].d,
                     m->f[GB25_GN_BT_U].d, m->f[GB25_GN_BT_V].d, m->colsum[0].d, m->colsum[1].d, dt, chi,
                     std::max(1, g.Nz / 12));   // the momentum kernel's chunking (momentum_impl)
  m->colsum_valid = true;
  LAUNCHCHK();
  return GB25_OK;
}
gb25_status ab2_tracers_impl(gb25_model* m, real dt, real chi) {
  const Grid& g = m->g;
  if (m->ahead_valid && dt == m->ahead_dt && chi == m->ahead_chi) {
    // the last tendency evaluation already advanced T and S with exactly these parameters
    std::swap(m->f[GB25_T].d, m->ahead[0].d);
    std::swap(m->f[GB25_S].d, m->ahead[1].d);
    m->ahead_valid = false;
    return GB25_OK;
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
    auto kern = m->variant_c ? k_ab2_tracers4<true> : k_ab2_tracers4<false>;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, m->stream, (real4*)T, (real4*)S, (const real4*)a,
                       (const real4*)bb, (const real4*)c, (const real4*)d, n4, dt, C1, C2);
  } else {
    int blocks = (int)std::min<long>((n + 255) / 256, 256 * 16);
    hipLaunchKernelGGL(k_ab2_tracers1, dim3(blocks), dim3(256), 0, m->stream, T, S, a, bb, c, d, n, dt, C1, C2);
  }
  LAUNCHCHK();
  return GB25_OK;
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
gb25_status barotropic_impl(gb25_model* m, real dt, bool ahead = false) {
  const Grid& g = m->g;
  if (m->baro_inflight && !ahead) {
    // a look-ahead that is not being adopted (changed dt, ...) may still be running on the side stream, and it uses
    // the same scratch sets
    HIPCHK(hipStreamWaitEvent(m->stream, m->ev_baro, 0));
    m->baro_inflight = false;
  }
  Timed t(m, GB25_K_BAROTROPIC);
  const bool wide = m->cfg.nranks > 1;
  const real dtau = (real)m->dtau_frac * dt;
  dim3 b(64, 4);
  Baro bb;
  real *cur[3], *nxt[3], *other[3], *out[3];
  if (!wide) {
    size_t nbar = m->f[GB25_ETA_BAR].elems() + m->f[GB25_U_BAR].elems() + m->f[GB25_V_BAR].elems();
    if (m->baro_block <= 1)   // (the blocked kernel starts its averages from zero itself)
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
  } else {
    if (m->baro_block <= 1)
      HIPCHK(hipMemsetAsync(m->wideBar[0].d, 0,
                            (m->wideBar[0].elems() + m->wideBar[1].elems() + m->wideBar[2].elems()) * sizeof(real),
                            m->stream));
    for (int q = 0; q < 3; q++) {
      cur[q] = m->wide[0][q].d; nxt[q] = m->wide[1][q].d; other[q] = m->wide[0][q].d;
      out[q] = ahead ? m->ahead_eta[q].d : m->f[GB25_ETA + q].d;
    }
    bb.etab = m->wideBar[0].d; bb.Ub = m->wideBar[1].d; bb.Vb = m->wideBar[2].d;
    bb.GU = m->wideG[0].d; bb.GV = m->wideG[1].d;
    bb.sx = g.Nx + 2 * m->W; bb.xo = m->W; bb.ilo = -m->W + 1; bb.ihi = g.Nx + m->W - 1; bb.wrap = 0;
  }
  bool finalize_after = false;
  if (m->baro_block > 1) {
    // temporally blocked: S substeps per launch on (64 x TY) tiles
    const int S = std::min(m->baro_block, (int)BT_SMAX), TYb = m->baro_rows;
    dim3 gm((bb.ihi - bb.ilo + BT_TX - 1) / BT_TX, (g.Ny + TYb - 1) / TYb);
    void (*kern)(Grid, BaroMulti, real) = nullptr;
    if (TYb == 16) kern = S <= 3 ? k_barotropic_multi<3, 16> : (S <= 5 ? k_barotropic_multi<5, 16> : k_barotropic_multi<7, 16>);
    else kern = S <= 3 ? k_barotropic_multi<3, 32> : (S <= 5 ? k_barotropic_multi<5, 32> : k_barotropic_multi<7, 32>);
    const int Sk = S <= 3 ? 3 : (S <= 5 ? 5 : 7);
    for (int s = 0; s < m->Ns; s += Sk) {
      BaroMulti bm;
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
      if (wide) {
        const Field* fb = ahead ? m->ahead_bar : &m->f[GB25_ETA_BAR];
        bm.eb_out = fb[0].d; bm.ub_out = fb[1].d; bm.vb_out = fb[2].d;
      }
      for (int q = 0; q < BT_SMAX; q++) bm.w[q] = (s + q < m->Ns) ? (real)m->weights[s + q] : real(0.);
      hipLaunchKernelGGL(kern, gm, dim3(BT_NT), 0, m->stream, g, bm, dtau);
      for (int q = 0; q < 3; q++) { real* w_ = nxt[q]; nxt[q] = other[q]; other[q] = w_; cur[q] = w_; }
    }
  } else {
    dim3 gr = grid2(bb.ihi - bb.ilo, g.Ny, b);
    for (int s = 0; s < m->Ns; s++) {
      bb.eta0 = cur[0]; bb.U0 = cur[1]; bb.V0 = cur[2];
      bb.eta1 = nxt[0]; bb.U1 = nxt[1]; bb.V1 = nxt[2];
      hipLaunchKernelGGL(k_barotropic_substep, gr, b, 0, m->stream, g, bb, dtau, (real)m->weights[s]);
      for (int q = 0; q < 3; q++) { real* w_ = nxt[q]; nxt[q] = other[q]; other[q] = w_; cur[q] = w_; }
    }
  }
  if (m->baro_block > 1 && !finalize_after) {   // the last blocked launch wrote eta, U, V and published the averages
    LAUNCHCHK();
    return GB25_OK;
  }
  dim3 gi = grid2(g.Nx, g.Ny, b);
  hipLaunchKernelGGL(k_barotropic_finalize, gi, b, 0, m->stream, g, out[0], out[1], out[2], bb.etab, bb.Ub, bb.Vb,
                     bb.sx, bb.xo);
  if (wide) {  // publish the averages in the canonical filtered-state arrays (compared by compare_states)
    InteriorCopies C{};
    int rmax = 0;
    for (int q = 0; q < 3; q++) {
      Field& dst = ahead ? m->ahead_bar[q] : m->f[GB25_ETA_BAR + q];
      C.dst[q] = dst.d; C.dsx[q] = dst.nx; C.dxo[q] = g.H;
      C.src[q] = m->wideBar[q].d; C.ssx[q] = bb.sx; C.sxo[q] = bb.xo; C.rows[q] = dst.ny;
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
    const bool ext = m->cfg.nranks > 1;
    int i0 = ext ? -g.H : 0, ni = ext ? g.Nx + 2 * g.H : g.Nx, skip_from = INT_MAX, skip = 0;
    if (part == 1) {
      i0 = 0;
      ni = g.Nx;
    } else if (part == 2) {
      if (!ext) return GB25_OK;
      i0 = -g.H; ni = 2 * g.H; skip_from = 0; skip = g.Nx;
    }
    const bool cs = use_colsum && m->colsum_valid && part != 2;
    hipLaunchKernelGGL(k_corrector, grid2(ni, g.Ny, b), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                       m->f[GB25_BT_U].d, m->f[GB25_BT_V].d, m->f[GB25_U_BAR].d, m->f[GB25_V_BAR].d,
                       cs ? m->colsum[0].d : nullptr, cs ? m->colsum[1].d : nullptr, i0, ni, std::max(1, g.Nz / 12),
                       skip_from, skip);
    LAUNCHCHK();
  }
  if (part == 2) return GB25_OK;
  m->colsum_valid = false;
  // cache_previous_tendencies!: G^- <- G^n is a pointer exchange; the next tendency evaluation
  // overwrites the (old G^-) buffers that now carry the G^n name.
  for (int q = 0; q < 4; q++) std::swap(m->f[GB25_GN_U + q].d, m->f[GB25_GM_U + q].d);
  m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;   // the look-aheads used the tendency pairs as they were before
  return GB25_OK;
}

gb25_status update_state_impl(gb25_model* m) {
  gb25_status s;
  if ((s = fill_halos_impl(m, true))) return s;
  if ((s = compute_w_impl(m))) return s;
  if ((s = compute_p_impl(m))) return s;
  if ((s = momentum_impl(m))) return s;
  return tracers_impl(m);
}

gb25_status ab2_step_impl(gb25_model* m, double dt, int euler) {
  gb25_status s;
  m->ahead_baro_valid = false;   // this route always runs the sub-cycle itself
  const real chi = euler ? -real(0.5) : (real)m->cfg.chi;
  if ((s = ab2_local_impl(m, (real)dt, chi))) return s;
  Halo2 hG;
  hG.p[0] = m->f[GB25_GN_BT_U].d; hG.is_v[0] = 0;
  hG.p[1] = m->f[GB25_GN_BT_V].d; hG.is_v[1] = 1;
  hG.p[2] = nullptr; hG.is_v[2] = 0;
  hG.n = 2;
  if ((s = fill_halos_2d(m, hG))) return s;
  return barotropic_impl(m, (real)dt);
}

// One time step on a single slab.  Two HIP streams: the tracer branch (AB2 of T,S -> their halos -> hydrostatic
// pressure: HBM- then fp64-bound) is independent of the velocity branch (AB2 of u,v -> split-explicit sub-cycle,
// which is latency-bound -> halos -> corrector -> halos -> w) until the tendencies need both, so it runs on a
// side stream and overlaps.  The phase order within each branch is the reference's (src/precompile.jl:31-42);
// T and S are untouched between the two halo fills of the reference sequence, so they are filled once.
gb25_status time_step_impl(gb25_model* m, int euler) {
  if (m->cfg.nranks != 1)
    return fail(m, GB25_ERR_STATE, "gb25_time_step on a slab of a %d-rank decomposition: drive gb25_time_step_stage",
                m->cfg.nranks);
  gb25_status s;
  const double dt = m->last_dt;
  if (!m->two_streams) {
    if ((s = ab2_step_impl(m, dt, euler))) return s;
    m->time += dt;
    m->iteration += 1;
    if ((s = fill_halos_impl(m, true))) return s;
    if ((s = corrector_impl(m, true))) return s;
    return update_state_impl(m);
  }
  const real chi = euler ? -real(0.5) : (real)m->cfg.chi;
  hipStream_t main = m->stream, side = m->side_stream;
  // ---- AB2 of u, v: normally the adoption of the look-ahead (a pointer exchange on the host, no kernel)
  const bool adopted = m->ahead_uv_valid && (real)dt == m->ahead_uv_dt && chi == m->ahead_uv_chi;
  const bool baro_adopted = adopted && m->ahead_baro_valid;   // (made from that very look-ahead, same dt)
  m->ahead_baro_valid = false;
  if ((s = ab2_velocities_impl(m, (real)dt, chi))) return s;
  Halo2 hG;
  hG.p[0] = m->f[GB25_GN_BT_U].d; hG.is_v[0] = 0;
  hG.p[1] = m->f[GB25_GN_BT_V].d; hG.is_v[1] = 1;
  hG.p[2] = nullptr; hG.is_v[2] = 0;
  hG.n = 2;
  // the sub-cycle reads G.U, G.V at interior points only (periodic wrap and walls are in the kernel): their halo
  // fill is for the state's sake and leaves the critical path when no kernel on this stream produced them
  if (!adopted && (s = fill_halos_2d(m, hG))) return s;
  HIPCHK(hipEventRecord(m->ev_fork, main));
  HIPCHK(hipStreamWaitEvent(side, m->ev_fork, 0));
  // ---- tracer branch (side stream)
  m->stream = side;
  s = ab2_tracers_impl(m, (real)dt, chi);
  if (!s) s = fill_halos_impl(m, true, false, 1, 2);      // y/z/x halos of T, S
  if (!s) s = compute_p_impl(m, INT_MIN, INT_MIN, 0, -1, true);
  if (!s && adopted) s = fill_halos_2d(m, hG);
  m->stream = main;
  if (s) return s;
  HIPCHK(hipEventRecord(m->ev_join, side));
  // ---- velocity branch (main stream)
  if (baro_adopted) {
    // the sub-cycle of this step ran beside the last tracer kernel: adopt eta, U, V and the filtered state
    HIPCHK(hipStreamWaitEvent(main, m->ev_baro, 0));
    m->baro_inflight = false;
    for (int q = 0; q < 3; q++) {
      std::swap(m->f[GB25_ETA + q].d, m->ahead_eta[q].d);
      std::swap(m->f[GB25_ETA_BAR + q].d, m->ahead_bar[q].d);
    }
    std::swap(m->bars, m->bars_ahead);
  } else if ((s = barotropic_impl(m, (real)dt))) {
    return s;
  }
  m->time += dt;
  m->iteration += 1;
  // The reference fills the halos of u, v, eta, U, V here as well as after the corrector.  On a single slab the
  // corrector reads and writes its own columns only, and the fill after it rewrites exactly the same halo cells from
  // the corrected interior, so the first fill has no effect on any later value: it is left out (3 launches).
  if ((s = corrector_impl(m, true))) return s;
  if ((s = fill_halos_impl(m, true, false, 3, 1))) return s;   // u, v and eta, U, V
  if ((s = compute_w_impl(m))) return s;
  // ---- join: the tendencies need w, u, v and the pressure differences, T, S
  HIPCHK(hipStreamWaitEvent(main, m->ev_join, 0));
  if ((s = momentum_impl(m))) return s;
  if (m->baro_ahead && m->ahead_uv_valid && !m->use_graphs && !m->ptr_exposed) {
    // G.U, G.V of the next step exist now: its sub-cycle (latency-bound) runs on the side stream beside the tracer
    // tendency kernel (issue-bound), into the partner buffers
    HIPCHK(hipEventRecord(m->ev_mom, main));
    HIPCHK(hipStreamWaitEvent(side, m->ev_mom, 0));
    m->stream = side;
    s = barotropic_impl(m, m->ahead_uv_dt, true);
    m->stream = main;
    if (s) return s;
    HIPCHK(hipEventRecord(m->ev_baro, side));
    m->ahead_baro_valid = true;
    m->baro_inflight = true;
  }
  return tracers_impl(m);
}

gb25_status initialize_impl(gb25_model* m) {
  const Grid& g = m->g;
  dim3 b(64, 4);
  hipLaunchKernelGGL(k_barotropic_mode, grid2(g.Nx, g.Ny, b), b, 0, m->stream, g, m->f[GB25_U].d, m->f[GB25_V].d,
                     m->f[GB25_BT_U].d, m->f[GB25_BT_V].d);
  LAUNCHCHK();
  return fill_halos_2d(m, halo2_prognostic(m));
}

}  // namespace

// =============================================================================================
extern "C" {

const char* gb25_version(void) { return sizeof(real) == 8 ? "gb25hip 0.1 (gfx950, Float64)" : "gb25hip 0.1 (gfx950, Float32)"; }
int32_t gb25_real_bytes(void) { return (int32_t)sizeof(real); }

void gb25_default_config(gb25_config* c, int32_t Nx, int32_t Ny, int32_t Nz) {
  memset(c, 0, sizeof *c);
  c->Nx = Nx; c->Ny = Ny; c->Nz = Nz;
  c->halo = 8; c->substeps = 30; c->rank = 0; c->nranks = 1; c->device = 0;
  c->dt = 60.0; c->chi = 0.1;
  c->lat_south = -80; c->lat_north = 80; c->lon_west = 0; c->lon_east = 360;
  c->depth = 4000; c->zexp_h = 30;
  c->g = 9.80665; c->Omega = 7.292115e-5; c->radius = 6371e3; c->rho0 = 1020.0;
}

gb25_status gb25_create(const gb25_config* cfg, gb25_model** out) {
  if (!cfg || !out) return GB25_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  gb25_model* m = new gb25_model();
  *out = m;  // returned even on failure so the caller can read the error string, then destroy
  m->cfg = *cfg;
  if (cfg->Nx < 8 || cfg->Ny < 8 || cfg->Nz < 4 || cfg->halo < 4 || cfg->substeps < 1 || cfg->substeps > 4096 ||
      cfg->nranks < 1 || cfg->rank < 0 || cfg->rank >= cfg->nranks || cfg->Nx % cfg->nranks != 0)
    return fail(m, GB25_ERR_INVALID_ARGUMENT,
                "invalid configuration: need Nx,Ny >= 8, Nz >= 4, halo >= 4, 1 <= substeps <= 4096, Nx %% nranks == 0");
  m->Nx = cfg->Nx / cfg->nranks;
  if (m->Nx < cfg->halo) return fail(m, GB25_ERR_INVALID_ARGUMENT, "slab narrower than the halo");
  {
    // the kernels address a parent array with 32-bit element indices and 32-bit byte offsets (buffer addressing)
    const double bytes = (double)(m->Nx + 2 * cfg->halo) * (cfg->Ny + 2 * cfg->halo + 1) * (cfg->Nz + 2 * cfg->halo + 1) *
                         sizeof(real);
    if (bytes >= 2147483648.0)
      return fail(m, GB25_ERR_INVALID_ARGUMENT,
                  "a %dx%dx%d slab needs %.1f GB per 3-D array; the kernels address at most 2 GB per array: decompose "
                  "in x (nranks) so that the local slab is narrower", m->Nx, cfg->Ny, cfg->Nz, bytes / 1e9);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(m, GB25_ERR_NO_DEVICE, "no HIP device visible; libgb25hip has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev)
    return fail(m, GB25_ERR_INVALID_ARGUMENT, "device ordinal %d out of range (%d devices)", cfg->device, ndev);
  HIPCHK(hipSetDevice(cfg->device));
  HIPCHK(hipStreamCreateWithFlags(&m->own_stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&m->ev_baro, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&m->ev_mom, hipEventDisableTiming));
  if (const char* e = getenv("GB25_TWO_STREAMS")) m->two_streams = atoi(e) != 0;
  m->stream = m->own_stream;
  m->last_dt = cfg->dt;
  if (const char* e = getenv("GB25_KERNELS")) m->kernel_gen = (strcmp(e, "v1") == 0) ? 1 : 2;
  if (const char* e = getenv("GB25_TRACER_V3")) m->tracer_v3 = atoi(e);
  if (const char* e = getenv("GB25_MOMENTUM_V4")) m->momentum_v4 = atoi(e);
  if (const char* e = getenv("GB25_MOMENTUM_V5")) m->momentum_v5 = atoi(e);
  if (const char* e = getenv("GB25_BARO_BLOCK")) m->baro_block = atoi(e);
  if (const char* e = getenv("GB25_BARO_ROWS")) m->baro_rows = (atoi(e) == 32) ? 32 : 16;
  if (const char* e = getenv("GB25_TILE_ROWS")) m->tile_rows = (atoi(e) == 4) ? 4 : 8;
  if (const char* e = getenv("GB25_VARIANT_A")) m->variant_a = atoi(e);
  if (const char* e = getenv("GB25_VARIANT_B")) m->variant_b = atoi(e);
  if (const char* e = getenv("GB25_VARIANT_C")) m->variant_c = atoi(e);
  if (const char* e = getenv("GB25_AB2_AHEAD")) m->ab2_ahead = atoi(e);
  // On launch-latency-bound grids the extra cross-stream hops of the sub-cycle look-ahead cost more than the
  // sub-cycle they hide (0.252 vs 0.262 ms/step at 360x180x24, 0.173 vs 0.185 at 128x64x8; +3.5 % at 1440x720x48).
  // A slab of a decomposition always uses it: there it also takes two exchanges off the critical path.
  m->baro_ahead = (cfg->nranks > 1 || (long)cfg->Nx * cfg->Ny * cfg->Nz >= 8000000L) ? 1 : 0;
  if (const char* e = getenv("GB25_BARO_AHEAD")) m->baro_ahead = atoi(e);
  if (const char* e = getenv("GB25_FILL_FUSED")) m->fill_fused = atoi(e);
  if (const char* e = getenv("GB25_LAZY_PHY")) m->phy_pinned = atoi(e) == 0;   // 0: store pHY' every step
  if (const char* e = getenv("GB25_GRAPH")) m->use_graphs = atoi(e);
  gb25_status s;
  if ((s = build_grid(m))) return s;
  if ((s = build_eos_tables(m))) return s;
  build_substeps(m);
  const int H = cfg->halo, sx = m->Nx + 2 * H;
  for (int id = 0; id < GB25_FIELD_COUNT; id++) {
    if (id >= GB25_ETA_BAR && id <= GB25_V_BAR) continue;  // allocated contiguously below
    int ny = cfg->Ny + 2 * H + (is_v_shaped(id) ? 1 : 0);
    int nz = is_2d(id) ? 1 : cfg->Nz + 2 * H + (id == GB25_W ? 1 : 0);
    if ((s = alloc_field(m, m->f[id], sx, ny, nz))) return s;
  }
  {
    size_t nc = (size_t)sx * (cfg->Ny + 2 * H), nv = (size_t)sx * (cfg->Ny + 2 * H + 1);
    HIPCHK(hipMalloc(&m->bars, (2 * nc + nv) * sizeof(real)));
    HIPCHK(hipMemset(m->bars, 0, (2 * nc + nv) * sizeof(real)));
    Field& e = m->f[GB25_ETA_BAR]; e.d = m->bars; e.nx = sx; e.ny = cfg->Ny + 2 * H; e.nz = 1;
    Field& u = m->f[GB25_U_BAR]; u.d = m->bars + nc; u.nx = sx; u.ny = cfg->Ny + 2 * H; u.nz = 1;
    Field& v = m->f[GB25_V_BAR]; v.d = m->bars + 2 * nc; v.nx = sx; v.ny = cfg->Ny + 2 * H + 1; v.nz = 1;
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
    const size_t np = (size_t)4 * std::max(1, m->g.Nz / 12) * m->g.sx * m->g.sy_v;
    HIPCHK(hipMalloc(&m->uv_partials, np * sizeof(real)));
    HIPCHK(hipMemset(m->uv_partials, 0, np * sizeof(real)));
  }
  if ((s = alloc_field(m, m->colsum[0], sx, m->f[GB25_BT_U].ny, 1))) return s;
  if ((s = alloc_field(m, m->colsum[1], sx, m->f[GB25_BT_V].ny, 1))) return s;
  if (cfg->nranks > 1) {
    m->W = m->Ns + 1;
    if (m->Nx < m->W)
      return fail(m, GB25_ERR_INVALID_ARGUMENT, "slab width %d is narrower than the barotropic halo %d", m->Nx, m->W);
    const int wsx = m->Nx + 2 * m->W;
    for (int a = 0; a < 2; a++)
      for (int q = 0; q < 3; q++)
        if ((s = alloc_field(m, m->wide[a][q], wsx, m->f[GB25_ETA + q].ny, 1))) return s;
    {   // the three running averages are one allocation, zeroed by one memset per step
      size_t tot = 0;
      for (int q = 0; q < 3; q++) {
        m->wideBar[q].nx = wsx; m->wideBar[q].ny = m->f[GB25_ETA + q].ny; m->wideBar[q].nz = 1;
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
    if ((s = alloc_field(m, m->wideG[0], wsx, m->f[GB25_GN_BT_U].ny, 1))) return s;
    if ((s = alloc_field(m, m->wideG[1], wsx, m->f[GB25_GN_BT_V].ny, 1))) return s;
  }
  HIPCHK(hipDeviceSynchronize());
  return GB25_OK;
}

void gb25_destroy(gb25_model* m) {
  if (!m) return;
  if (m->own_stream) hipStreamSynchronize(m->own_stream);
  for (auto& e : m->graphs) {
    hipGraphExecDestroy(e.exec);
    hipGraphDestroy(e.graph);
  }
  for (int id = 0; id < GB25_FIELD_COUNT; id++)
    if (!(id >= GB25_ETA_BAR && id <= GB25_V_BAR) && m->f[id].d) hipFree(m->f[id].d);
  if (m->bars) hipFree(m->bars);
  if (m->dpx.d) hipFree(m->dpx.d);
  if (m->dpy.d) hipFree(m->dpy.d);
  for (int q = 0; q < 3; q++)
    for (Field* p : {&m->pp[q], &m->pp2[q], &m->ahead_eta[q]})
      if (p->d) hipFree(p->d);
  if (m->bars_ahead) hipFree(m->bars_ahead);
  for (auto& p : m->colsum)
    if (p.d) hipFree(p.d);
  for (int q = 0; q < 2; q++)
    for (Field* p : {&m->ahead[q], &m->ahead_uv[q], &m->ahead_G[q], &m->ahead_colsum[q]})
      if (p->d) hipFree(p->d);
  if (m->uv_partials) hipFree(m->uv_partials);
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
  if (m->ev_fork) hipEventDestroy(m->ev_fork);
  if (m->ev_join) hipEventDestroy(m->ev_join);
  if (m->ev_baro) hipEventDestroy(m->ev_baro);
  if (m->ev_mom) hipEventDestroy(m->ev_mom);
  if (m->own_stream) hipStreamDestroy(m->own_stream);
  delete m;
}

const char* gb25_last_error_string(const gb25_model* m) { return m ? m->err.c_str() : "null model"; }

gb25_status gb25_set_stream(gb25_model* m, void* s) {
  CHECK_MODEL(m);
  m->stream = (hipStream_t)s;  // NULL = HIP's default stream; ordering between streams is the caller's business
  return GB25_OK;
}
gb25_status gb25_use_own_stream(gb25_model* m) {
  CHECK_MODEL(m);
  HIPCHK(hipStreamSynchronize(m->stream));
  m->stream = m->own_stream;
  return GB25_OK;
}
gb25_status gb25_synchronize(gb25_model* m) {
  CHECK_MODEL(m);
  HIPCHK(hipStreamSynchronize(m->stream));
  HIPCHK(hipStreamSynchronize(m->side_stream));   // the sub-cycle look-ahead may still be running there
  return GB25_OK;
}

gb25_status gb25_field_dims(const gb25_model* m, gb25_field id, int include_halos, int32_t d[3]) {
  if (!m || id < 0 || id >= GB25_FIELD_COUNT || !d) return GB25_ERR_INVALID_ARGUMENT;
  const Field& F = m->f[id];
  const int H = m->cfg.halo;
  if (include_halos) {
    d[0] = F.nx; d[1] = F.ny; d[2] = F.nz;
  } else {
    d[0] = F.nx - 2 * H; d[1] = F.ny - 2 * H; d[2] = is_2d(id) ? 1 : F.nz - 2 * H;
  }
  return GB25_OK;
}

static gb25_status copy_field(gb25_model* m, gb25_field id, real* host, int include_halos, bool to_device) {
  if (!m || id < 0 || id >= GB25_FIELD_COUNT || !host) return GB25_ERR_INVALID_ARGUMENT;
  if (to_device && (id == GB25_U || id == GB25_V)) m->colsum_valid = false;  // cached column integrals are stale
  Field& F = m->f[id];
  HIPCHK(hipStreamSynchronize(m->stream));
  if (to_device) HIPCHK(hipStreamSynchronize(m->side_stream));   // a look-ahead may still be reading the old values
  if (include_halos) {
    if (to_device) HIPCHK(hipMemcpy(F.d, host, F.elems() * sizeof(real), hipMemcpyHostToDevice));
    else HIPCHK(hipMemcpy(host, F.d, F.elems() * sizeof(real), hipMemcpyDeviceToHost));
    return GB25_OK;
  }
  const int H = m->cfg.halo;
  int32_t d[3];
  gb25_field_dims(m, id, 0, d);
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
  for (int q = 0; q < 2; q++)
    HIPCHK(hipMemcpyAsync(m->ahead[q].d, m->f[GB25_T + q].d, m->f[GB25_T + q].elems() * sizeof(real),
                          hipMemcpyDeviceToDevice, m->stream));
  return GB25_OK;
}
static gb25_status mirror_velocities(gb25_model* m) {   // u and v alternate between two buffers likewise
  m->ahead_uv_valid = false;
  for (int q = 0; q < 2; q++)
    HIPCHK(hipMemcpyAsync(m->ahead_uv[q].d, m->f[GB25_U + q].d, m->f[GB25_U + q].elems() * sizeof(real),
                          hipMemcpyDeviceToDevice, m->stream));
  return GB25_OK;
}
gb25_status gb25_set_field(gb25_model* m, gb25_field f, const void* host, int include_halos) {
  gb25_status s = copy_field(m, f, static_cast<real*>(const_cast<void*>(host)), include_halos, true);
  if (s == GB25_OK && f == GB25_PHY) {
    m->phy_stale = false;
    s = widen_phy(m);
  }
  if (s == GB25_OK) {
    m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;   // any input of the look-aheads may have changed
    if (f == GB25_T || f == GB25_S) s = mirror_tracers(m);
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
gb25_status gb25_get_substepping(const gb25_model* m, int32_t* n, double* frac, double* w) {
  if (!m) return GB25_ERR_INVALID_ARGUMENT;
  if (n) *n = m->Ns;
  if (frac) *frac = m->dtau_frac;
  if (w) for (int k = 0; k < m->Ns; k++) w[k] = (double)(real)m->weights[k];
  return GB25_OK;
}

gb25_status gb25_set_baroclinic_instability(gb25_model* m) {
  CHECK_MODEL(m);
  const Grid& g = m->g;
  hipLaunchKernelGGL(k_set_baroclinic_instability, dim3((g.Nx + 255) / 256, g.Ny, g.Nz), dim3(256), 0, m->stream, g,
                     m->f[GB25_T].d, m->f[GB25_S].d);
  LAUNCHCHK();
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
  m->last_dt = dt;
  return GB25_OK;
}

gb25_status gb25_initialize(gb25_model* m) { CHECK_MODEL(m); return initialize_impl(m); }
gb25_status gb25_mask_immersed_fields(gb25_model* m) { CHECK_MODEL(m); return GB25_OK; }
gb25_status gb25_fill_halo_regions(gb25_model* m) { CHECK_MODEL(m); return fill_halos_impl(m, true); }
gb25_status gb25_fill_halo_regions_local(gb25_model* m) { CHECK_MODEL(m); return fill_halos_impl(m, false); }
gb25_status gb25_compute_auxiliaries(gb25_model* m) {
  CHECK_MODEL(m);
  gb25_status s = compute_w_impl(m);
  return s ? s : compute_p_impl(m);
}
gb25_status gb25_fill_diffusivity_halos(gb25_model* m) { CHECK_MODEL(m); return GB25_OK; }
gb25_status gb25_compute_momentum_tendencies(gb25_model* m) { CHECK_MODEL(m); return momentum_impl(m); }
gb25_status gb25_compute_tracer_tendencies(gb25_model* m) { CHECK_MODEL(m); return tracers_impl(m); }
gb25_status gb25_compute_boundary_tendencies(gb25_model* m) { CHECK_MODEL(m); return GB25_OK; }
gb25_status gb25_compute_tendencies(gb25_model* m) {
  CHECK_MODEL(m);
  gb25_status s = momentum_impl(m);
  return s ? s : tracers_impl(m);
}
gb25_status gb25_ab2_step(gb25_model* m, double dt, int euler) {
  CHECK_MODEL(m);
  if (m->cfg.nranks != 1) return fail(m, GB25_ERR_STATE, "gb25_ab2_step needs the staged path on a multi-rank slab");
  return ab2_step_impl(m, dt, euler);
}
gb25_status gb25_correct_velocities_and_cache_previous_tendencies(gb25_model* m, double) {
  CHECK_MODEL(m);
  return corrector_impl(m);
}
gb25_status gb25_update_state(gb25_model* m) {
  CHECK_MODEL(m);
  if (m->cfg.nranks != 1) return fail(m, GB25_ERR_STATE, "gb25_update_state needs gb25_update_state_local + exchange");
  return update_state_impl(m);
}
gb25_status gb25_update_state_local(gb25_model* m) {
  CHECK_MODEL(m);
  gb25_status s;
  if ((s = fill_halos_impl(m, false, m->cfg.nranks > 1))) return s;
  if ((s = compute_w_impl(m))) return s;
  if ((s = compute_p_impl(m))) return s;
  if ((s = momentum_impl(m))) return s;
  return tracers_impl(m);
}

// ---- HIP-graph replay of the AB2 time step (opt-in: GB25_GRAPH=1) --------------------------------
// A step is ~25 dependent launches; on the small configurations (128x64x8, 360x180x24) their dispatch latency, not
// the kernels, sets the step time.  With GB25_GRAPH=1 the step is captured once per host state (the pointer
// exchanges make the state alternate with period 2) and replayed with one hipGraphLaunch; the host-side transition
// (pointer names, flags, clock) is re-applied from the recorded `post` state.  Bitwise identical to eager launches
// (tests/test_gpu_parity.py).  It is OFF by default because it does not pay on this stack: measured on MI355X /
// ROCm 7.2 (profiles/r01_tuning_log.md) 0.220 vs 0.201 ms/step at 128x64x8 with the two-stream step, 0.190 vs 0.191
// single-stream, 0.320 vs 0.307 at 360x180x24 -- the ~8 us per dependent kernel are spent on the device side of the
// dispatch, which a graph does not remove; only fewer kernels would.
static HostState host_state(const gb25_model* m) {
  HostState h;
  for (int q = 0; q < 8; q++) h.ptr[q] = m->f[GB25_GN_U + q].d;
  h.ptr[8] = m->f[GB25_T].d; h.ptr[9] = m->f[GB25_S].d;
  h.ptr[10] = m->ahead[0].d; h.ptr[11] = m->ahead[1].d;
  for (int q = 0; q < 2; q++) {
    h.ptr[12 + q] = m->colsum[q].d;    h.ptr[14 + q] = m->ahead_colsum[q].d;
    h.ptr[16 + q] = m->f[GB25_U + q].d; h.ptr[18 + q] = m->ahead_uv[q].d;
    h.ptr[20 + q] = m->f[GB25_GN_BT_U + q].d; h.ptr[22 + q] = m->ahead_G[q].d;
  }
  h.ahead_valid = m->ahead_valid; h.colsum_valid = m->colsum_valid; h.ahead_uv_valid = m->ahead_uv_valid;
  h.ahead_dt = m->ahead_dt; h.ahead_chi = m->ahead_chi;
  h.ahead_uv_dt = m->ahead_uv_dt; h.ahead_uv_chi = m->ahead_uv_chi;
  return h;
}
static void set_host_state(gb25_model* m, const HostState& h) {
  for (int q = 0; q < 8; q++) m->f[GB25_GN_U + q].d = h.ptr[q];
  m->f[GB25_T].d = h.ptr[8]; m->f[GB25_S].d = h.ptr[9];
  m->ahead[0].d = h.ptr[10]; m->ahead[1].d = h.ptr[11];
  for (int q = 0; q < 2; q++) {
    m->colsum[q].d = h.ptr[12 + q];    m->ahead_colsum[q].d = h.ptr[14 + q];
    m->f[GB25_U + q].d = h.ptr[16 + q]; m->ahead_uv[q].d = h.ptr[18 + q];
    m->f[GB25_GN_BT_U + q].d = h.ptr[20 + q]; m->ahead_G[q].d = h.ptr[22 + q];
  }
  m->ahead_valid = h.ahead_valid; m->colsum_valid = h.colsum_valid; m->ahead_uv_valid = h.ahead_uv_valid;
  m->ahead_dt = h.ahead_dt; m->ahead_chi = h.ahead_chi;
  m->ahead_uv_dt = h.ahead_uv_dt; m->ahead_uv_chi = h.ahead_uv_chi;
}
static void drop_graphs(gb25_model* m) {
  for (auto& e : m->graphs) {
    hipGraphExecDestroy(e.exec);
    hipGraphDestroy(e.graph);
  }
  m->graphs.clear();
}
static gb25_status step_with_graph(gb25_model* m) {
  if (!m->use_graphs || m->profile || m->cfg.nranks != 1 || m->stream == nullptr)
    return time_step_impl(m, 0);
  const HostState pre = host_state(m);
  const double dt = m->last_dt;
  StepGraph* hit = nullptr;
  for (auto& e : m->graphs)
    if (e.dt == dt && e.stream == m->stream && e.pre == pre) hit = &e;
  if (!hit) {
    bool again = false;
    for (auto& h : m->seen) again |= (h == pre);
    if (!again) {   // one-off states (after a host write, a changed dt, ...) are not worth a capture
      if (m->seen.size() >= 8) m->seen.clear();
      m->seen.push_back(pre);
      return time_step_impl(m, 0);
    }
    const double time0 = m->time;
    const int64_t it0 = m->iteration;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    bool ok = hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    gb25_status st = ok ? time_step_impl(m, 0) : GB25_ERR_HIP;
    if (ok) ok = (hipStreamEndCapture(m->stream, &graph) == hipSuccess) && st == GB25_OK && graph != nullptr;
    const HostState post = host_state(m);
    set_host_state(m, pre);   // nothing has run yet: back to the state the graph starts from
    m->time = time0;
    m->iteration = it0;
    if (ok) ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    if (!ok) {   // capture is a convenience, never a requirement
      if (graph) hipGraphDestroy(graph);
      (void)hipGetLastError();
      m->use_graphs = 0;
      return time_step_impl(m, 0);
    }
    if (m->graphs.size() >= 8) drop_graphs(m);   // states cycle with period 2; more than a few means churn
    m->graphs.push_back({pre, post, dt, m->stream, exec, graph});
    hit = &m->graphs.back();
  }
  HIPCHK(hipGraphLaunch(hit->exec, m->stream));
  set_host_state(m, hit->post);
  m->time += dt;
  m->iteration += 1;
  return GB25_OK;
}

gb25_status gb25_first_time_step(gb25_model* m) {
  CHECK_MODEL(m);
  gb25_status s;
  if ((s = initialize_impl(m))) return s;
  if ((s = update_state_impl(m))) return s;
  return time_step_impl(m, 1);
}
gb25_status gb25_time_step(gb25_model* m) { CHECK_MODEL(m); return step_with_graph(m); }
gb25_status gb25_loop(gb25_model* m, int32_t n) {
  CHECK_MODEL(m);
  for (int s = 0; s < n; s++) {
    gb25_status st = step_with_graph(m);
    if (st) return st;
  }
  return GB25_OK;
}

// ---- x-slab exchange -------------------------------------------------------------------------
// group 0: H columns of u, v, T, S (all parent rows) and of eta, U, V -> the neighbour's x halo.
// group 1: W columns of eta, U, V, G.U, G.V -> the neighbour's wide barotropic halo.
struct Piece {
  real* src;      // array that is packed from (canonical layout)
  real* dst;      // array that is unpacked into
  int src_sx, src_xo, dst_sx, dst_xo;
  long rows;
};
// groups 3 and 4 are groups 1 and 2 of the sub-cycle LOOK-AHEAD: G.U, G.V come from the momentum look-ahead's partner
// buffers, the new eta, U, V live in theirs
static void group_pieces(gb25_model* m, int group, std::vector<Piece>& out, int* ncols) {
  const int H = m->cfg.halo, sx = m->Nx + 2 * H;
  if (group == 0 || group == 2 || group == 4) {
    *ncols = H;
    if (group == 0) {
      for (int id : {GB25_U, GB25_V, GB25_T, GB25_S}) {
        Field& F = m->f[id];
        out.push_back({F.d, F.d, sx, H, sx, H, (long)F.ny * F.nz});
      }
    } else {
      for (int q = 0; q < 3; q++) {
        Field& F = group == 2 ? m->f[GB25_ETA + q] : m->ahead_eta[q];
        out.push_back({F.d, F.d, sx, H, sx, H, (long)F.ny * F.nz});
      }
    }
  } else {
    *ncols = m->W;
    const int wsx = m->Nx + 2 * m->W;
    for (int q = 0; q < 3; q++) {
      Field& F = m->f[GB25_ETA + q];
      out.push_back({F.d, m->wide[0][q].d, sx, H, wsx, m->W, (long)F.ny});
    }
    for (int q = 0; q < 2; q++) {
      Field& F = group == 1 ? m->f[GB25_GN_BT_U + q] : m->ahead_G[q];
      out.push_back({F.d, m->wideG[q].d, sx, H, wsx, m->W, (long)F.ny});
    }
  }
}
gb25_status gb25_halo_buffer_elems(const gb25_model* m, int group, int64_t* n) {
  if (!m || !n || group < 0 || group > 4) return GB25_ERR_INVALID_ARGUMENT;
  if (m->cfg.nranks == 1) { *n = 0; return GB25_OK; }
  std::vector<Piece> ps;
  int nc = 0;
  group_pieces(const_cast<gb25_model*>(m), group, ps, &nc);
  int64_t t = 0;
  for (auto& p : ps) t += p.rows * nc;
  *n = t;
  return GB25_OK;
}
// side_mask: bit 0 = west, bit 1 = east; buf[side] = that side's contiguous device buffer
static gb25_status pack_unpack(gb25_model* m, int group, int side_mask, real* const buf[2], bool pack) {
  if (!m || group < 0 || group > 4 || !(side_mask & 3)) return GB25_ERR_INVALID_ARGUMENT;
  if (m->cfg.nranks == 1) return fail(m, GB25_ERR_STATE, "halo pack/unpack on a single-slab model");
  std::vector<Piece> ps;
  int nc = 0;
  group_pieces(m, group, ps, &nc);
  ColumnPieces P{};
  P.ncols = nc;
  long max_n = 0;
  for (int side = 0; side < 2; side++) {
    if (!(side_mask & (1 << side))) continue;
    if (!buf[side]) return GB25_ERR_INVALID_ARGUMENT;
    size_t off = 0;
    for (auto& p : ps) {
      const int f = P.n++;
      P.rows[f] = p.rows;
      P.buf[f] = buf[side] + off;
      if (pack) {   // west side: interior columns [0, nc); east side: [Nx-nc, Nx)
        P.arr[f] = p.src; P.sx[f] = p.src_sx; P.i0[f] = p.src_xo + (side == 0 ? 0 : m->Nx - nc);
      } else {      // west halo: columns [-nc, 0); east halo: [Nx, Nx+nc)
        P.arr[f] = p.dst; P.sx[f] = p.dst_sx; P.i0[f] = p.dst_xo + (side == 0 ? -nc : m->Nx);
      }
      off += (size_t)p.rows * nc;
      max_n = std::max(max_n, p.rows * nc);
    }
  }
  dim3 gr((unsigned)((max_n + 255) / 256), (unsigned)P.n);
  if (pack) hipLaunchKernelGGL(k_move_columns<true>, gr, dim3(256), 0, m->stream, P);
  else hipLaunchKernelGGL(k_move_columns<false>, gr, dim3(256), 0, m->stream, P);
  LAUNCHCHK();
  return GB25_OK;
}
gb25_status gb25_halo_pack(gb25_model* m, int group, int side, void* buf) {
  if (side < 0 || side > 1) return GB25_ERR_INVALID_ARGUMENT;
  real* b[2] = {nullptr, nullptr};
  b[side] = static_cast<real*>(buf);
  return pack_unpack(m, group, 1 << side, b, true);
}
gb25_status gb25_halo_unpack(gb25_model* m, int group, int side, const void* buf) {
  if (side < 0 || side > 1) return GB25_ERR_INVALID_ARGUMENT;
  real* b[2] = {nullptr, nullptr};
  b[side] = static_cast<real*>(const_cast<void*>(buf));
  return pack_unpack(m, group, 1 << side, b, false);
}
gb25_status gb25_halo_pack_both(gb25_model* m, int group, void* west_buf, void* east_buf) {
  real* b[2] = {static_cast<real*>(west_buf), static_cast<real*>(east_buf)};
  return pack_unpack(m, group, 3, b, true);
}
gb25_status gb25_halo_unpack_both(gb25_model* m, int group, const void* west_buf, const void* east_buf) {
  real* b[2] = {static_cast<real*>(const_cast<void*>(west_buf)), static_cast<real*>(const_cast<void*>(east_buf))};
  return pack_unpack(m, group, 3, b, false);
}

// The time step of one slab, cut at its two exchange points (see include/gb25.h).
gb25_status gb25_time_step_stage(gb25_model* m, int stage, int euler) {
  CHECK_MODEL(m);
  if (m->cfg.nranks == 1) return fail(m, GB25_ERR_STATE, "gb25_time_step_stage on a single-slab model");
  const Grid& g = m->g;
  gb25_status s;
  const double dt = m->last_dt;
  const real chi = euler ? -real(0.5) : (real)m->cfg.chi;
  if (stage == 0) {
    // AB2 update of u,v,T,S + barotropic forcing, then the y/z boundary layers of the 3-D bundle so that its packed
    // x columns (group 0) can travel WHILE the sub-cycle runs; the host also exchanges group 1 now
    const bool uv_adopted = m->ahead_uv_valid && (real)dt == m->ahead_uv_dt && chi == m->ahead_uv_chi;
    m->baro_adopted = uv_adopted && m->ahead_baro_valid;
    m->ahead_baro_valid = false;
    if ((s = ab2_local_impl(m, (real)dt, chi))) return s;
    if (m->baro_adopted) {
      // the sub-cycle of this step, its wide-halo exchange and the exchange of the new eta, U, V columns all ran
      // beside the last tracer kernel (stage 5): adopt the results, stages 1 and groups 1, 2 are skipped
      for (int q = 0; q < 3; q++) {
        std::swap(m->f[GB25_ETA + q].d, m->ahead_eta[q].d);
        std::swap(m->f[GB25_ETA_BAR + q].d, m->ahead_bar[q].d);
      }
      std::swap(m->bars, m->bars_ahead);
      m->time += dt;
      m->iteration += 1;
    }
    if ((s = fill_halos_impl(m, false, false, 1))) return s;
    if (m->two_streams) {
      // T, S of the slab's own columns are final from here on: their pressure (fp64-bound) runs on the side stream
      // beside the exchanges and the sub-cycle; the strips next to the x halos follow in stage 2.  The first x
      // difference of this pass reads a stale halo column and is redone by the west strip.
      HIPCHK(hipEventRecord(m->ev_fork, m->stream));
      HIPCHK(hipStreamWaitEvent(m->side_stream, m->ev_fork, 0));
      hipStream_t main = m->stream;
      m->stream = m->side_stream;
      s = compute_p_impl(m, 0, g.Nx - 1, 0, -1, true);
      m->stream = main;
      if (s) return s;
      HIPCHK(hipEventRecord(m->ev_join, m->side_stream));
    }
    return GB25_OK;
  } else if (stage == 1 || stage == 5) {
    // stage 1: group 1 has been unpacked into the wide halos: copy the interiors, sub-cycle, publish.
    // stage 5: the same for the NEXT step (look-ahead): group 3 has been unpacked, G.U, G.V come from the momentum
    //          look-ahead, the results go to the partner buffers of eta, U, V and of the filtered state.
    const bool ahead = stage == 5;
    if (ahead && !m->ahead_uv_valid) return fail(m, GB25_ERR_STATE, "stage 5 without a velocity look-ahead");
    if (!ahead && m->baro_adopted) return fail(m, GB25_ERR_STATE, "stage 1 after stage 0 adopted the sub-cycle");
    std::vector<Piece> ps;
    int nc = 0;
    group_pieces(m, ahead ? 3 : 1, ps, &nc);
    {
      InteriorCopies C{};
      int rmax = 0;
      for (auto& p : ps) {
        const int q = C.n++;
        C.dst[q] = p.dst; C.dsx[q] = p.dst_sx; C.dxo[q] = p.dst_xo;
        C.src[q] = p.src; C.ssx[q] = p.src_sx; C.sxo[q] = p.src_xo; C.rows[q] = (int)p.rows;
        rmax = std::max(rmax, (int)p.rows);
      }
      hipLaunchKernelGGL(k_copy_interior_columns, dim3((g.Nx + 255) / 256, rmax, C.n), dim3(256), 0, m->stream, C,
                         g.Nx);
    }
    LAUNCHCHK();
    if (ahead) {
      if ((s = barotropic_impl(m, m->ahead_uv_dt, true))) return s;
      Halo2 h2;
      for (int q = 0; q < 3; q++) { h2.p[q] = m->ahead_eta[q].d; h2.is_v[q] = q == 2; }
      h2.n = 3;
      if ((s = fill_halos_impl(m, false, false, 2, 3, nullptr, true, &h2))) return s;   // their x columns: group 4
      m->ahead_baro_valid = true;
      return GB25_OK;
    }
    if ((s = barotropic_impl(m, (real)dt))) return s;
    m->time += dt;
    m->iteration += 1;
    // y layer of the new eta, U, V; their x columns are group 2
    return fill_halos_impl(m, false, false, 2);
  } else if (stage == 2) {
    // the barotropic corrector on the slab's own columns: needs nothing from the neighbours, so it runs while
    // groups 2 and 0 are still travelling
    return corrector_impl(m, true, 1);
  } else if (stage == 3) {
    // groups 2 and 0 have been unpacked: corrector on the x-halo columns, then update_state without any
    // further exchange (y/z layers re-filled over the extended x range; w and p recomputed in the halos)
    if ((s = corrector_impl(m, true, 2))) return s;
    // the interior pressure pass of stage 0 must be over before the fills touch T, S and before the west strip
    // rewrites column 0 (it has been for a while: the exchanges and the sub-cycle ran in between)
    if (m->two_streams) HIPCHK(hipStreamWaitEvent(m->stream, m->ev_join, 0));
    if ((s = fill_halos_impl(m, false, true))) return s;
    if (m->two_streams) {   // the two pressure strips run beside w (side stream)
      hipStream_t main = m->stream;
      HIPCHK(hipEventRecord(m->ev_fork, main));
      HIPCHK(hipStreamWaitEvent(m->side_stream, m->ev_fork, 0));
      m->stream = m->side_stream;
      s = compute_p_impl(m, -g.H + 1, 0, g.Nx, g.Nx + g.H - 2, true);   // west strip (redoes column 0) + east strip
      m->stream = main;
      if (s) return s;
      HIPCHK(hipEventRecord(m->ev_join, m->side_stream));
    }
    if ((s = compute_w_impl(m))) return s;
    if (m->two_streams) {
      HIPCHK(hipStreamWaitEvent(m->stream, m->ev_join, 0));
    } else {
      if ((s = compute_p_impl(m))) return s;
    }
    return momentum_impl(m);
  } else if (stage == 4) {
    // the tracer tendencies; the host runs the look-ahead of the next sub-cycle (groups 3, 4 and stage 5) beside them
    return tracers_impl(m);
  }
  return fail(m, GB25_ERR_INVALID_ARGUMENT, "stage must be 0 .. 5");
}

gb25_status gb25_lookahead_state(const gb25_model* m, int32_t* velocities_ready, int32_t* subcycle_adopted) {
  if (!m) return GB25_ERR_INVALID_ARGUMENT;
  if (velocities_ready) *velocities_ready = (m->ahead_uv_valid && m->baro_ahead && !m->ptr_exposed) ? 1 : 0;
  if (subcycle_adopted) *subcycle_adopted = m->baro_adopted ? 1 : 0;
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

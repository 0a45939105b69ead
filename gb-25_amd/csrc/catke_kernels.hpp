// catke_kernels.hpp -- closure = CATKEVerticalDiffusivity(): buoyancy / surface flux / diffusivity kernels and the
// variable-coefficient implicit vertical solves (register-resident and streamed).  Included by kernels.hpp, inside
// namespace gb25, after the grid / halo helpers it uses (store_x_images, teos10_level, ab2_advance).
#pragma once
// =============================================================================================
// closure = CATKEVerticalDiffusivity() (GB-25 src/baroclinic_instability_model.jl:30,50-51; sharding/
// less_simple_sharding_problem.jl:84-93; compared fields src/correctness.jl:60-67).  The formulas are those of
// oracle/gb25_oracle.c (catke_at_face: Oceananigans' TKE-based closure restated [UPSTREAM-UNVERIFIED] after Wagner et al.
// 2025); single domain, lat-lon grid (flat or with a GridFittedBottom).
//   k_catke_buoyancy        b = -g rho'(T, S, z) / rho0 per cell (the equation of state in fp64, as the pressure kernel)
//   k_catke_surface_flux    J^b = g (alpha J^T - beta J^S) from the top flux boundary conditions (zero without them)
//   k_catke_diffusivities   one thread per column, marching up the faces: kappa_u, kappa_c, kappa_e, L^e with the halo cells
//                           their fill derives (a14), and the explicit TKE terms added to G^n.e (shear production, positive
//                           buoyancy flux, surface TKE flux)
//   k_implicit_vertical_var the tridiagonal solve with these diffusivity fields (u, v, T, S, e in one launch)
// =============================================================================================
struct CatkePar {
  real Cs, Cb, Csp, CRid, CRi0;
  real Chi[4], Clo[4], Cun[4], Cc[4], Ce[4];   // psi = u, c, e, D
  real CWu, CWw, emin, Jbmin, tau_neg;
};
// N^2 at face k (between cells k-1 and k; k = blockIdx.z + 1), stored at the index of cell k.  Both buoyancies and their
// difference in fp64 (as the pressure kernel differences its pressure): in a mixed layer the difference of two Float32
// buoyancies is a handful of ulps, and the stratification-limited mixing length goes with N^-1.
__global__ void k_catke_buoyancy(Grid g, const real* __restrict__ T, const real* __restrict__ S, real* __restrict__ n2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z + 1;
  if (i >= g.Nx || j >= g.Ny) return;
  const int o = ic(g, i, j, k), ob = o - g.pl_c;
  const double gr = -(double)g.g / (double)g.rho0, sc = 0.875 / 35.16504;
  const double bk = gr * teos10_level(g.eos + 28 * k, sqrt_pos(((double)S[o] + 32.0) * sc), (double)T[o] * 0.025);
  const double bb = gr * teos10_level(g.eos + 28 * (k - 1), sqrt_pos(((double)S[ob] + 32.0) * sc), (double)T[ob] * 0.025);
  n2[o] = (real)((bk - bb) / g.dzf_d[k]);
}
__global__ void k_catke_surface_flux(Grid g, const real* __restrict__ T, const real* __restrict__ S, real* __restrict__ Jb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  const int o2 = i2(g, i, j);
  double J = 0.0;
  if (g.top_flux[2] || g.top_flux[3]) {
    const int o = ic(g, i, j, g.Nz - 1);
    const double sc = 0.875 / 35.16504, d = 1e-2, Tc = (double)T[o], Sc = (double)S[o];
    const double* c = g.eos + 28 * (g.Nz - 1);
    auto rho = [&](double t, double s) { return teos10_level(c, sqrt_pos((s + 32.0) * sc), t * 0.025); };
    const double drdT = (rho(Tc + d, Sc) - rho(Tc - d, Sc)) / (2 * d), drdS = (rho(Tc, Sc + d) - rho(Tc, Sc - d)) / (2 * d);
    const double JT = g.top_flux[2] ? (double)g.top_flux[2][o2] : 0.0, JS = g.top_flux[3] ? (double)g.top_flux[3][o2] : 0.0;
    J = (double)g.g * (-drdT * JT - drdS * JS) / (double)g.rho0;
  }
  // (a slab of a decomposition: the x halo columns and the rows beyond a zipper fold arrive with the 3-D bundle)
  const bool xw = g.x_periodic && i < g.H, xe = g.x_periodic && i >= g.Nx - g.H;
  store_x_images(g, Jb, o2, (real)J, xw, xe);
  // (the y layers exist next to walls only: a rank of a 2-D decomposition gets the rows of its open sides from the neighbour)
  if (j == 0 && g.jws == 0) store_x_images(g, Jb, o2 - g.sx, (real)J, xw, xe);
  if (j == g.Ny - 1 && (g.jwn == g.Ny || g.cv.north_fold) && !(g.cv.north_fold && !g.x_periodic)) store_x_images(g, Jb, o2 + g.sx, (real)J, xw, xe);
}
struct CatkeFace {
  real ku, kc, ke, lD, P, wb;
};
template <bool IMM>
__global__ __launch_bounds__(256) void k_catke_diffusivities(Grid g, CatkePar c, const real* __restrict__ u,
                                                             const real* __restrict__ v, const real* __restrict__ e,
                                                             const real* __restrict__ b, const real* __restrict__ Jb,
                                                             real* __restrict__ KU, real* __restrict__ KC,
                                                             real* __restrict__ KE, real* __restrict__ Le,
                                                             real* __restrict__ Ge, int i_lo, int j_hi, int j_lo) {
  // Columns i_lo .. Nx-1, rows 0 .. j_hi-1.  Single domain: the interior (0, Ny), the halo cells written as images.  A slab
  // of a decomposition: i_lo = -1 and, with the zipper fold, j_hi = Ny + 1 -- the one halo column / row whose kappa_u the
  // implicit solves of u (averaged in x) and of v (in y, on the fold line) read is COMPUTED here from the halo columns of
  // e, u, v, N^2 and J^b (all of them exchanged already), bit for bit what its owner computes, instead of exchanged.
  // (2-D decomposition: j_lo = -1 below a southern neighbour -- kappa_u of that row is averaged into the v faces of row 0)
  const int i = i_lo + (int)(blockIdx.x * blockDim.x + threadIdx.x), j = j_lo + (int)(blockIdx.y * blockDim.y + threadIdx.y);
  if (i >= g.Nx || j >= j_hi) return;
  const bool own = i >= 0 && j >= 0 && j < g.Ny;
  const int Nz = g.Nz, o2 = i2(g, i, j), pc = g.pl_c, pv = g.pl_v;
  const int kc0 = IMM ? min((int)(g.im.ordA[o2] & 255), Nz) : 0;   // first active level of the column
  const bool xw = g.x_periodic && i < g.H, xe = g.x_periodic && i >= g.Nx - g.H;
  // (the y layers exist next to walls only)
  const bool ys = own && j == 0 && g.jws == 0, yn = own && j == g.Ny - 1 && j_hi == g.Ny && (g.jwn == g.Ny || g.cv.north_fold);
  auto put = [&](real* a, int o, real x) {   // the cell and the halo cells its fill derives from it (a14)
    store_x_images(g, a, o, x, xw, xe);
    if (ys) store_x_images(g, a, o - g.sx, x, xw, xe);
    if (yn) store_x_images(g, a, o + g.sx, x, xw, xe);
  };
  const real zt = g.zc[Nz - 1] + real(0.5) * g.dzc[Nz - 1];          // surface
  const real jb = Jb[o2], jbp = jb > c.Jbmin ? jb : c.Jbmin, rjbp = real(1.) / jbp, rCRid = real(1.) / c.CRid;
  const bool cooled = jb > c.Jbmin;
  // A window of three levels travels up the column: what a cell level contributes -- e, u on its two x faces, v on its two
  // y faces, N^2 on the face below it (zero on the boundary faces and next to the solid) -- is loaded two levels AHEAD of
  // its use, so the loads of a level are in flight during the arithmetic of the level below (a first version loaded and
  // used level by level and ran at the latency of 48 dependent round trips).
  struct Level { real e, uw, ue, vs, vn, n2; };
  auto load_level = [&](int k, int oc_, int ov_) -> Level {
    Level L = {real(0.), real(0.), real(0.), real(0.), real(0.), real(0.)};
    if (k < Nz) {
      L.e = e[oc_]; L.uw = u[oc_]; L.ue = u[oc_ + 1]; L.vs = v[ov_]; L.vn = v[ov_ + g.sx];
      if (k > kc0) L.n2 = b[oc_];
    }
    return L;
  };
  int o = ic(g, i, j, 0), ov = iv(g, i, j, 0);
  CatkeFace lo = {real(0.), real(0.), real(0.), real(0.), real(0.), real(0.)};
  put(KU, o, real(0.)); put(KC, o, real(0.)); put(KE, o, real(0.));
  real zf_k = g.zc[0] - real(0.5) * g.dzc[0], zbot = zf_k;            // z of face 0; bottom of the column
  for (int q = 0; q < kc0; q++) zbot += g.dzc[q];
  Level cur = load_level(0, o, ov), nxt = load_level(1, o + pc, ov + pv);
  for (int k = 0; k < Nz; k++) {
    const Level pre = load_level(k + 2, o + 2 * pc, ov + 2 * pv);
    const real ge = own ? Ge[o] : real(0.);
    // ---- face k+1 (top of cell k)
    zf_k += g.dzc[k];
    CatkeFace hi = {real(0.), real(0.), real(0.), real(0.), real(0.), real(0.)};
    const int kf = k + 1;
    if (kf > kc0 && kf < Nz) {
      const real rdz = g.rdzf[kf];
      const real ef = (cur.e + nxt.e) / real(2.), ep = ef > real(0.) ? ef : real(0.), ws = sqrt(ep);
      const real uw = (nxt.uw - cur.uw) * rdz, ue = (nxt.ue - cur.ue) * rdz;
      const real vs = (nxt.vs - cur.vs) * rdz, vn = (nxt.vn - cur.vn) * rdz;
      const real S2 = (uw * uw + ue * ue) / real(2.) + (vs * vs + vn * vn) / real(2.);
      const real N2 = nxt.n2, N2above = pre.n2;
      const real Ri = (N2 == real(0.)) ? real(0.) : N2 / S2;
      const real dup = c.Cs * (zt - zf_k), ddn = c.Cb * (zf_k - zbot);
      real ls = dup < ddn ? dup : ddn;
      if (N2 > real(0.)) {
        const real lN = ws / sqrt(N2);
        ls = lN < ls ? lN : ls;
      }
      // convective lengths: l^h_psi = C^c_psi w*^3 / J^b+ * max(0, 1 - C^sp sqrt(S^2) w*^2 / J^b+) where the column loses
      // buoyancy and N^2 < 0; l^e_psi = C^e_psi J^b+ / (w* N^2 + J^b_min) in the stable level just below such a layer.
      // The factor common to the four psi is formed once (and only in columns with J^b > J^b_min at all).
      real conv_scale = real(0.);
      bool entraining = false;
      if (cooled) {
        if (N2 < real(0.)) {
          const real esp = real(1.) - c.Csp * sqrt(S2) * ws * ws * rjbp;
          conv_scale = ws * ws * ws * rjbp * (esp > real(0.) ? esp : real(0.));
        } else if (N2above < real(0.)) {
          entraining = true;
          conv_scale = jbp / (ws * N2 + c.Jbmin);
        }
      }
      // stability functions: one step function of Ri for the four psi
      real tstep = (Ri - c.CRi0) * rCRid;
      tstep = tstep < real(0.) ? real(0.) : (tstep > real(1.) ? real(1.) : tstep);
      real lpsi[4];
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const real lconv = (entraining ? c.Ce[p] : c.Cc[p]) * conv_scale;
        const real sg = Ri < real(0.) ? c.Cun[p] : c.Clo[p] + (c.Chi[p] - c.Clo[p]) * tstep;
        const real lst = p < 3 ? sg * ls : ls / sg;
        lpsi[p] = lconv > lst ? lconv : lst;
      }
      hi.ku = lpsi[0] * ws; hi.kc = lpsi[1] * ws; hi.ke = lpsi[2] * ws; hi.lD = lpsi[3];
      hi.P = hi.ku * S2;
      hi.wb = -hi.kc * N2;
    }
    put(KU, o + pc, hi.ku); put(KC, o + pc, hi.kc); put(KE, o + pc, hi.ke);
    // ---- cell k (own columns: the halo ones are there for their kappa only)
    real L = real(0.);
    if (own && k >= kc0) {
      const real ek = cur.e, lD = (lo.lD + hi.lD) / real(2.), wb = (lo.wb + hi.wb) / real(2.);
      const real omega = lD > real(0.) ? sqrt(rabs(ek)) / lD : real(0.);
      const real wbm = wb < real(0.) ? wb : real(0.);
      L = -omega + (ek > c.emin ? wbm / ek : real(0.)) - (ek < real(0.) ? real(1.) / c.tau_neg : real(0.));
      real src = (lo.P + hi.P) / real(2.) + (wb > real(0.) ? wb : real(0.));
      if (k == Nz - 1) {   // the surface TKE flux: -(C^W_u* u*^3 + C^W_wD w_D^3), into the top cell
        // (friction velocity from the boundary-condition values AT (i, j), as Oceananigans' friction_velocity takes them)
        const real Ju = g.top_flux[0] ? g.top_flux[0][o2] : real(0.);
        const real Jv = g.top_flux[1] ? g.top_flux[1][o2] : real(0.);
        const real us2 = sqrt(Ju * Ju + Jv * Jv), us3 = us2 * sqrt(us2);
        const real wD3 = (jb > real(0.) ? jb : real(0.)) * g.dzc[k];
        src += (c.CWu * us3 + c.CWw * wD3) / g.dzc[k];
      }
      Ge[o] = ge + src;
    }
    if (own) {
      put(Le, o, L);
      if (k == 0) store_x_images(g, Le, o - pc, L, xw, xe);        // bottom / top layer (interior rows only, like the fill)
      if (k == Nz - 1) store_x_images(g, Le, o + pc, L, xw, xe);
    }
    lo = hi;
    cur = nxt;
    nxt = pre;
    o += pc;
    ov += pv;
  }
}
// TripolarGrid: the rows beyond the zipper fold of the diffusivity fields and of J^b (cell-centred in x and y, no sign
// change): (i, Ny-1+q) <- (Nx-1-i, Ny-1-q), q = 1 .. H, over
// every parent column (source column wrapped periodically), which
// replaces the zero-gradient northern layer k_catke_diffusivities wrote.  blockIdx.z: face levels 0 .. Nz of the three
// kappa, levels -1 .. Nz of L^e (its bottom / top layer on those rows), then J^b.
__global__ void k_catke_fold(Grid g, real* __restrict__ KU, real* __restrict__ KC, real* __restrict__ KE,
                             real* __restrict__ Le, real* __restrict__ Jb) {
  const int ip = blockIdx.x * blockDim.x + threadIdx.x;
  if (ip >= g.sx) return;
  const int i = ip - g.H, q = blockIdx.y + 1, z = blockIdx.z, Nz = g.Nz;
  const int iw = ((i % g.Nx) + g.Nx) % g.Nx, isrc = g.Nx - 1 - iw;
  const int jd = g.Ny - 1 + q, js = g.Ny - 1 - q;
  if (z <= Nz) {
    const int od = ic(g, i, jd, z), os = ic(g, isrc, js, z);
    KU[od] = KU[os];
    KC[od] = KC[os];
    KE[od] = KE[os];
  }
  if (z <= Nz + 1) Le[ic(g, i, jd, z - 1)] = Le[ic(g, isrc, js, min(max(z - 1, 0), Nz - 1))];
  if (z == Nz + 2) Jb[i2(g, i, jd)] = Jb[i2(g, isrc, js)];
}
// The tridiagonal solve with diffusivity FIELDS (implicit_step! with CATKE's kappa_u, kappa_c, kappa_e, L^e).  The
// elimination factors depend on the column, so each thread eliminates its own.  Two launches per step:
//   MODE 0 (after the AB2 update of u, v): blockIdx.z = 0 u (kappa_u averaged in x), 1 v (averaged in y; with the zipper
//          fold also the fold line); the column integrals of the new u, v are rewritten for the barotropic corrector;
//   MODE 1 (after that of T, S):           blockIdx.z = 0 T AND S in one thread -- they share kappa_c, hence the factors:
//          one elimination, two right-hand sides; 1 e with kappa_e and -dt L^e on the diagonal, its AB2 update
//          e* = e + dt (C1 G^n.e - C2 G^-.e) formed as the column is loaded (no separate sweep over e).
// Register kernel (Nz <= NZT <= 64): three per-thread arrays -- a column, a second column (S, or dt L^e), and the
// diffusivities dt kappa(face k+1), ALL loaded before the elimination starts (the loads of a wave are in flight together;
// a first version fetched kappa inside the dependent chain and ran at a quarter of the bandwidth) and overwritten by the
// factors gamma_k as the chain passes.  Per level one reciprocal of the pivot (the 1-ulp v_rcp_f32 in Float32, as the
// constant-coefficient kernels) and a dozen multiply-adds: with true divisions the kernel was bound by VALU issue.
struct ImplicitVarFields {
  real* f[5];                  // u, v, T, S, e
  const real *KU, *KC, *KE, *Le;
  const real *GnE, *GmE;       // null: e already holds e*
  real dt, C1, C2;
  real* sum[2];                // column integrals of the new u, v (the look-ahead's predate the solve)
  int kchunks;
  real* gam[2];                // streaming kernel only: the factors of the two blockIdx.z slices ((c,f,c)-shaped scratch)
};
template <bool IMM, int MODE>
__device__ __forceinline__ int implicit_var_first_level(const Grid& g, int z, int o2, int j) {
  int kf = 0;
  if (IMM) {
    const unsigned w = MODE == 1 ? g.im.ordA[o2] : g.im.ordC[o2] >> (z == 0 ? 8 : 16);
    kf = min((int)(w & 255), g.Nz);
  }
  if (MODE == 0 && z == 1 && j == g.jws) kf = g.Nz;   // (the wall face: nothing to solve, its column integral is zero)
  return kf;
}
// column integral of u / v with the chunked association every other producer of these sums uses
template <class Get>
__device__ __forceinline__ real implicit_var_colsum(const Grid& g, int kchunks, Get x) {
  const int Nz = g.Nz, klen = (Nz + kchunks - 1) / kchunks;
  real tot = real(0.), q = real(0.);
  int kk = 0;
  for (int k = 0; k < Nz; k++) {
    q = (kk == 0) ? g.dzc[k] * x(k) : rfma(g.dzc[k], x(k), q);
    if (++kk == klen || k == Nz - 1) {
      tot = (k < klen) ? q : tot + q;
      kk = 0;
    }
  }
  return tot;
}
template <int NZT, bool IMM, int MODE>
__global__ __launch_bounds__(256) void k_implicit_vertical_var(Grid g, ImplicitVarFields A) {
  const int z = blockIdx.z, Nz = g.Nz;
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  const bool vsh = MODE == 0 && z == 1, pair = MODE == 1 && z == 0, tke = MODE == 1 && z == 1;
  if (i >= g.Nx || j >= (vsh ? g.Ny : g.Ny)) return;
  const int o2 = i2(g, i, j), kf = implicit_var_first_level<IMM, MODE>(g, z, o2, j);
  real* Fa = MODE == 0 ? A.f[z] : (pair ? A.f[2] : A.f[4]);
  real* Fb = A.f[3];
  const real* K = MODE == 0 ? A.KU : (pair ? A.KC : A.KE);
  const int pl = vsh ? g.pl_v : g.pl_c, o0 = vsh ? iv(g, i, j, 0) : ic(g, i, j, 0), oc = ic(g, i, j, 0), pc = g.pl_c;
  const int nb = MODE == 0 ? (z == 0 ? -1 : -g.sx) : 0;   // the second column kappa_u is averaged with
  real a[NZT], b[NZT], gm[NZT];
  const bool ab2 = tke && A.GnE != nullptr;
  // ---- loads only, no arithmetic between them: every load of the column is in flight before the first wait
  if (MODE == 1 && ab2) {   // e* = e + dt (C1 G^n.e - C2 G^-.e) first (three arrays in, one out), then L^e and kappa_e
#pragma unroll
    for (int k = 0; k < NZT; k++)
      if (k < Nz) {
        a[k] = Fa[o0 + k * pl];
        b[k] = A.GnE[o0 + k * pl];
        gm[k] = A.GmE[o0 + k * pl];
      }
#pragma unroll
    for (int k = 0; k < NZT; k++)
      if (k < Nz) a[k] = ab2_advance(a[k], b[k], gm[k], A.dt, A.C1, A.C2);
  }
#pragma unroll
  for (int k = 0; k < NZT; k++) {
    if (!(MODE == 1 && ab2)) a[k] = (k < Nz) ? Fa[o0 + k * pl] : real(0.);
    const int of = oc + (k + 1) * pc;                      // the top face of level k
    gm[k] = (k < Nz - 1) ? K[of] : real(0.);
    if (MODE == 0) b[k] = (k < Nz - 1) ? K[of + nb] : real(0.);
    else b[k] = (k < Nz) ? (pair ? Fb[o0 + k * pl] : A.Le[o0 + k * pl]) : real(0.);
  }
  real rbet = real(1.), pa = real(0.), pb = real(0.), kup = real(0.);   // rbet: reciprocal of the last pivot
#pragma unroll
  for (int k = 0; k < NZT; k++)
    if (k < Nz && k >= kf) {
      const real ktop = A.dt * (MODE == 0 ? (b[k] + gm[k]) / real(2.) : gm[k]);   // dt kappa at the top face of level k
      const real t = (k == kf) ? real(0.) : -(kup * g.rdzf[k]);    // coupling through face k, without the cell height
      const real lo = t * g.rdzc[k];
      const real up = -(ktop * g.rdzc[k]) * g.rdzf[k + 1];         // (ktop = 0 at the top level)
      real dg = real(1.) - lo - up;
      if (MODE == 1 && tke) dg -= A.dt * b[k];
      const real gk = (t * g.rdzc[k - 1]) * rbet;                   // upper coefficient of the level below / its pivot
      rbet = rcp(dg - lo * gk);                                     // (k = kf: t = lo = gk = 0)
      pa = (a[k] - lo * pa) * rbet;
      a[k] = pa;
      if (MODE == 1 && pair) {
        pb = (b[k] - lo * pb) * rbet;
        b[k] = pb;
      }
      gm[k] = gk;
      kup = ktop;
    }
#pragma unroll
  for (int k = NZT - 2; k >= 0; k--)
    if (k < Nz - 1 && k >= kf) {
      a[k] = a[k] - gm[k + 1] * a[k + 1];
      if (MODE == 1 && pair) b[k] = b[k] - gm[k + 1] * b[k + 1];
    }
  real tot = real(0.);
  if (MODE == 0 && A.sum[z] != nullptr) {   // (before the stores: the table loads below must not wait behind them)
    const int klen = (Nz + A.kchunks - 1) / A.kchunks;
    real q = real(0.);
    int kk = 0;
#pragma unroll
    for (int k = 0; k < NZT; k++)
      if (k < Nz) {
        q = (kk == 0) ? g.dzc[k] * a[k] : rfma(g.dzc[k], a[k], q);
        if (++kk == klen || k == Nz - 1) {
          tot = (k < klen) ? q : tot + q;
          kk = 0;
        }
      }
  }
#pragma unroll
  for (int k = 0; k < NZT; k++)
    if (k < Nz && (k >= kf || ab2)) {
      Fa[o0 + k * pl] = a[k];
      if (MODE == 1 && pair) Fb[o0 + k * pl] = b[k];
    }
  if (MODE == 0 && A.sum[z] != nullptr) A.sum[z][o2] = (vsh && j == g.jws) ? real(0.) : tot;
}
// Any Nz (the register kernel stops at 64 levels): the same elimination streamed through HBM.  Forward sweep: the
// eliminated right-hand side goes back into the field, the factors into a scratch array; backward sweep reads both.
// 7 instead of 3 accesses per cell and field, ~30 registers, loads independent of the chain (unrolled by 4).
template <bool IMM, int MODE>
__global__ __launch_bounds__(256) void k_implicit_vertical_var_stream(Grid g, ImplicitVarFields A) {
  const int z = blockIdx.z, Nz = g.Nz;
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  const bool vsh = MODE == 0 && z == 1, pair = MODE == 1 && z == 0, tke = MODE == 1 && z == 1;
  if (i >= g.Nx || j >= (vsh ? g.Ny : g.Ny)) return;
  const int o2 = i2(g, i, j), kf = implicit_var_first_level<IMM, MODE>(g, z, o2, j);
  real* Fa = MODE == 0 ? A.f[z] : (pair ? A.f[2] : A.f[4]);
  real* Fb = A.f[3];
  real* G = A.gam[z];
  const real* K = MODE == 0 ? A.KU : (pair ? A.KC : A.KE);
  const int pl = vsh ? g.pl_v : g.pl_c, o0 = vsh ? iv(g, i, j, 0) : ic(g, i, j, 0), oc = ic(g, i, j, 0), pc = g.pl_c;
  const int og = iv(g, i, j, 0), pg = g.pl_v;              // the scratch array has the larger of the two shapes
  const int nb = MODE == 0 ? (z == 0 ? -1 : -g.sx) : 0;
  const bool ab2 = tke && A.GnE != nullptr;
  if (ab2)
    for (int k = 0; k < kf; k++)
      Fa[o0 + k * pl] = ab2_advance(Fa[o0 + k * pl], A.GnE[o0 + k * pl], A.GmE[o0 + k * pl], A.dt, A.C1, A.C2);
  real rbet = real(1.), pa = real(0.), pb = real(0.), kup = real(0.);
#pragma unroll 4
  for (int k = kf; k < Nz; k++) {
    real xa = Fa[o0 + k * pl], xb = real(0.), le = real(0.);
    if (MODE == 1 && pair) xb = Fb[o0 + k * pl];
    if (MODE == 1 && tke) le = A.dt * A.Le[o0 + k * pl];
    if (MODE == 1 && ab2) xa = ab2_advance(xa, A.GnE[o0 + k * pl], A.GmE[o0 + k * pl], A.dt, A.C1, A.C2);
    real ktop = real(0.);
    if (k < Nz - 1) {
      const int of = oc + (k + 1) * pc;
      ktop = A.dt * (MODE == 0 ? (K[of + nb] + K[of]) / real(2.) : K[of]);
    }
    const real t = (k == kf) ? real(0.) : -(kup * g.rdzf[k]);
    const real lo = t * g.rdzc[k];
    const real up = -(ktop * g.rdzc[k]) * g.rdzf[k + 1];
    const real dg = real(1.) - lo - up - le;
    const real gk = (t * g.rdzc[k - 1]) * rbet;
    rbet = rcp(dg - lo * gk);
    pa = (xa - lo * pa) * rbet;
    Fa[o0 + k * pl] = pa;
    if (MODE == 1 && pair) {
      pb = (xb - lo * pb) * rbet;
      Fb[o0 + k * pl] = pb;
    }
    G[og + k * pg] = gk;
    kup = ktop;
  }
#pragma unroll 4
  for (int k = Nz - 2; k >= kf; k--) {
    const real gk = G[og + (k + 1) * pg];
    pa = Fa[o0 + k * pl] - gk * pa;
    Fa[o0 + k * pl] = pa;
    if (MODE == 1 && pair) {
      pb = Fb[o0 + k * pl] - gk * pb;
      Fb[o0 + k * pl] = pb;
    }
  }
  if (MODE == 0 && A.sum[z] != nullptr) {
    const real tot = implicit_var_colsum(g, A.kchunks, [&](int k) { return Fa[o0 + k * pl]; });
    A.sum[z][o2] = (vsh && j == g.jws) ? real(0.) : tot;
  }
}


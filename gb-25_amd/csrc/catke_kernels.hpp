// catke_kernels.hpp -- closure = CATKEVerticalDiffusivity(): buoyancy / surface flux / diffusivity kernels and the
// variable-coefficient implicit vertical solves (register-resident and streamed).  Included by kernels.hpp, inside
// namespace gb25, after the grid / halo helpers it uses (store_x_images, teos10_level, ab2_advance).
#pragma once
// =============================================================================================
// closure = CATKEVerticalDiffusivity() (GB-25 src/baroclinic_instability_model.jl:30,50-51; sharding/
// less_simple_sharding_problem.jl:84-93; compared fields src/correctness.jl:60-67).  The formulas and their order are those of
// oracle/gb25_oracle.c (the CATKE section's header: Oceananigans 0.96's structure as recalled, [UPSTREAM-UNVERIFIED]).
// What compute_diffusivities! does inside update_state!, kernel by kernel:
//   k_catke_n2              N^2 = g (alpha dzT - beta dzS) on the faces, alpha and beta of TEOS-10 at the face (fp64)
//   k_catke_tke_step        time_step_catke_equation!, first half: per column the new kappa_e, L^e, the fast TKE terms (shear
//                           production between the previous and the current velocities with the OLD kappa_u, buoyancy flux with
//                           the OLD kappa_c) and the AB2 update e* = e + dt (C1 (G^n.e + fast) - C2 G^-.e), G^-.e <- the total
//   k_implicit_vertical_var ... second half: (1 - dt dz kappa_e dz - dt L^e) e = e*           (MODE 1, the e slice alone)
//   [halos of e; previous velocities <- velocities]
//   k_catke_surface_flux    compute_average_surface_buoyancy_flux!: J^b filtered over t* = cbrt(l_D^2 / J^b+); also the top
//                           boundary condition of e (the surface TKE flux) as a 2-D source for G^n.e
//   k_catke_diffusivities   compute_CATKE_diffusivities!: kappa_u, kappa_c, kappa_e from the new e, with the halo cells their
//                           fill derives (a14)
// and compute_tendencies! adds: k_tracer_tendencies_single (-div(u e) into G^n.e) + k_catke_add_top_source.
// =============================================================================================
struct CatkePar {
  real Cs, Cb, Csp, CRid, CRi0;
  real Chi[4], Clo[4], Cun[4], Cc[4], Ce[4];   // psi = u, c, e, D
  real CWu, CWw, emin, Jbmin, tau_neg, CWeps;
  real rCRid, rtau_neg;      // reciprocals (host)
};
// -d rho / d Theta and d rho / d S_A from a folded level table (teos10_level's layout), differentiated term by term
__device__ __forceinline__ void teos10_level_sens(const double* __restrict__ c, double s, double t, double& a, double& b) {
  const double p1 = c[7] + s * (c[8] + s * (c[9] + s * (c[10] + s * (c[11] + s * c[12]))));
  const double p2 = c[13] + s * (c[14] + s * (c[15] + s * (c[16] + s * c[17])));
  const double p3 = c[18] + s * (c[19] + s * (c[20] + s * c[21]));
  const double p4 = c[22] + s * (c[23] + s * c[24]);
  const double p5 = c[25] + s * c[26];
  const double p6 = c[27];
  const double d0 = c[1] + s * (2.0 * c[2] + s * (3.0 * c[3] + s * (4.0 * c[4] + s * (5.0 * c[5] + s * (6.0 * c[6])))));
  const double d1 = c[8] + s * (2.0 * c[9] + s * (3.0 * c[10] + s * (4.0 * c[11] + s * (5.0 * c[12]))));
  const double d2 = c[14] + s * (2.0 * c[15] + s * (3.0 * c[16] + s * (4.0 * c[17])));
  const double d3 = c[19] + s * (2.0 * c[20] + s * (3.0 * c[21]));
  const double d4 = c[23] + s * (2.0 * c[24]);
  const double d5 = c[26];
  const double rt = p1 + t * (2.0 * p2 + t * (3.0 * p3 + t * (4.0 * p4 + t * (5.0 * p5 + t * (6.0 * p6)))));
  const double rs = d0 + t * (d1 + t * (d2 + t * (d3 + t * (d4 + t * d5))));
  a = -(rt * 0.025);
  b = rs * ((0.875 / 35.16504) / (2.0 * s));
}
// N^2 at face k (between cells k-1 and k; k = blockIdx.z + 1 in 1 .. Nz-1), stored at the index of cell k; zero where the face
// touches the solid.  Columns [i_lo, i_lo + ni), rows [j_lo, j_lo + nj): a rank of a decomposition computes the first halo
// column / row too (kappa there is COMPUTED, not exchanged).  alpha, beta and the differences in fp64: in a mixed layer the
// difference of two Float32 temperatures is a handful of ulps, and the stratification-limited length goes with N^-1.
template <bool IMM>
__global__ void k_catke_n2(Grid g, const real* __restrict__ T, const real* __restrict__ S, real* __restrict__ n2, int i_lo,
                           int j_lo, int ni, int nj) {
  const int a = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y * blockDim.y + threadIdx.y, k = blockIdx.z + 1;
  if (a >= ni || b >= nj) return;
  const int i = i_lo + a, j = j_lo + b;
  const int o = ic(g, i, j, k), ob = o - g.pl_c;
  const int kc0 = IMM ? min((int)(g.im.ordA[i2(g, i, j)] & 255), g.Nz) : 0;
  real r = real(0.);
  if (k > kc0) {
    const double Tl = (double)T[ob], Th = (double)T[o], Sl = (double)S[ob], Sh = (double)S[o], sc = 0.875 / 35.16504;
    double al, be;
    teos10_level_sens(g.eosf + 28 * k, sqrt_pos(((Sl + Sh) * 0.5 + 32.0) * sc), (Tl + Th) * 0.5 * 0.025, al, be);
    const double rdz = 1.0 / g.dzf_d[k];
    r = (real)((double)g.g * (al * ((Th - Tl) * rdz) - be * ((Sh - Sl) * rdz)) / (double)g.rho0);
  }
  n2[o] = r;
}
// per-column context of the CATKE column kernels
struct CatkeColumn {
  int kc0;                   // first active level
  int NUw, NUe, NVs, NVn;    // first level from which the u faces i, i+1 / the v faces j, j+1 are active nodes
  real zt, zbot, Hcol, jb;
  real rjb;                  // 1 / (J^b + J^b_min): the convective lengths divide by it four times per face
};
template <bool IMM>
__device__ __forceinline__ CatkeColumn catke_column(const Grid& g, int i, int j, int o2, real jb) {
  CatkeColumn q;
  q.kc0 = IMM ? min((int)(g.im.ordA[o2] & 255), g.Nz) : 0;
  q.NUw = q.NUe = q.NVs = q.NVn = 0;
  if (IMM) {
    q.NUw = (int)((g.im.ordD[o2] >> 16) & 255); q.NUe = (int)((g.im.ordD[o2 + 1] >> 16) & 255);
    q.NVs = (int)((g.im.ordD[o2] >> 24) & 255); q.NVn = (int)((g.im.ordD[o2 + g.sx] >> 24) & 255);
  }
  q.zt = uniform_at(g.zc, g.Nz - 1) + real(0.5) * uniform_at(g.dzc, g.Nz - 1);
  real zb = uniform_at(g.zc, 0) - real(0.5) * uniform_at(g.dzc, 0);
  for (int l = 0; l < q.kc0; l++) zb += uniform_at(g.dzc, l);
  q.zbot = zb;
  q.Hcol = q.zt - zb;
  q.jb = jb;
  q.rjb = real(0.);          // (the caller sets it: it needs the closure's parameters)
  return q;
}
struct CatkeLengths { real ku, kc, ke, convD; };
// Square roots and quotients of the column kernels.  Float32: the hardware's 1-ulp v_sqrt_f32 / v_rcp_f32 (as the WENO weights
// take their reciprocals, device_common.hpp) -- the correctly rounded sequences clang emits for sqrtf and '/' are 10-14
// instructions each, and a level of k_catke_tke_step has eight of the one and seven of the other (a third of its 500 instructions);
// operands are e >= e_min, N^2, S^2, lengths: normal numbers, or zeros whose quotients the callers guard.  Float64: the true ones.
#ifndef GB25_CATKE_FAST
#define GB25_CATKE_FAST 0
#endif
__device__ __forceinline__ float csqrt(float x) { return GB25_CATKE_FAST ? __builtin_amdgcn_sqrtf(x) : sqrt(x); }
__device__ __forceinline__ double csqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float cdiv(float a, float b) { return GB25_CATKE_FAST ? a * __builtin_amdgcn_rcpf(b) : a / b; }
__device__ __forceinline__ double cdiv(double a, double b) { return a / b; }
// the mixing lengths of an OPEN (c,c,f) face (both cells active): wl, wh = sqrt(max(e_min, e)) of the cells below and above
__device__ __forceinline__ CatkeLengths catke_face_eval(const CatkePar& c, const CatkeColumn& q, real wl, real wh, real N2,
                                                        real N2above, real S2, real zf) {
  const real ws = (wl + wh) / real(2.), ws2 = (wl * wl + wh * wh) / real(2.), ws3 = (wl * wl * wl + wh * wh * wh) / real(2.);
  const real Ri = (N2 == real(0.)) ? real(0.) : cdiv(N2, S2);
  real dup = c.Cs * (q.zt - zf), ddn = c.Cb * (zf - q.zbot);
  dup = dup < real(0.) ? real(0.) : dup;
  ddn = ddn < real(0.) ? real(0.) : ddn;
  real ls = dup < ddn ? dup : ddn;
  if (N2 > real(0.)) {
    const real lN = cdiv(ws, csqrt(N2));
    ls = lN < ls ? lN : ls;
  }
  const real jb = q.jb, jbe = c.Jbmin;
  const bool convecting = jb > jbe && N2 < real(0.), entraining = jb > jbe && N2 > real(0.) && N2above < real(0.);
  real lconv[4] = {real(0.), real(0.), real(0.), real(0.)};
  if (convecting || entraining) {
    // (one reciprocal per column and one per face instead of five divisions: the f32 division is a ten-instruction sequence and
    // this function was half of the column kernels' 490 instructions per level)
    const real Sp = csqrt(S2) * ws2 * q.rjb, esp = real(1.) - c.Csp * Sp;
    const real scale = convecting ? ws3 * q.rjb : cdiv(jb, ws * N2 + jbe);
#pragma unroll
    for (int p = 0; p < 4; p++) {
      // (both coefficients as kernel arguments, THEN the select: a select of the two arrays is a per-lane pointer and a vector load)
      const real cc_ = c.Cc[p], ce_ = c.Ce[p];
      real l = (convecting ? cc_ : ce_) * scale;
      l *= esp;
      lconv[p] = l > real(0.) ? l : real(0.);
    }
  }
  real tstep = (Ri - c.CRi0) * c.rCRid;
  tstep = tstep < real(0.) ? real(0.) : (tstep > real(1.) ? real(1.) : tstep);
  real lpsi[3];
#pragma unroll
  for (int p = 0; p < 3; p++) {
    const real sg = Ri < real(0.) ? c.Cun[p] : c.Clo[p] + (c.Chi[p] - c.Clo[p]) * tstep;
    real l = sg * ls;
    l = lconv[p] > l ? lconv[p] : l;
    lpsi[p] = l < q.Hcol ? l : q.Hcol;
  }
  CatkeLengths r;
  r.ku = lpsi[0] * ws; r.kc = lpsi[1] * ws; r.ke = lpsi[2] * ws;
  r.convD = lconv[3];
  return r;
}
// dissipation_length_scale(c,c,c) of an active cell from the N^2, S^2 and convective dissipation lengths of its two faces
// (wc = sqrt(max(e_min, e)) of the cell: the face evaluations hold it already)
__device__ __forceinline__ real catke_dissipation_length(const CatkePar& c, const CatkeColumn& q, real wc, real zc, real N2lo,
                                                         real N2hi, real S2lo, real S2hi, real cDlo, real cDhi) {
  const real lh = (cDlo + cDhi) / real(2.), N2 = (N2lo + N2hi) / real(2.), S2 = (S2lo + S2hi) / real(2.);
  const real Ri = (N2 == real(0.)) ? real(0.) : cdiv(N2, S2);
  real dup = c.Cs * (q.zt - zc), ddn = c.Cb * (zc - q.zbot);
  dup = dup < real(0.) ? real(0.) : dup;
  ddn = ddn < real(0.) ? real(0.) : ddn;
  real ls = dup < ddn ? dup : ddn;
  if (N2 > real(0.)) {
    const real lN = cdiv(wc, csqrt(N2));
    ls = lN < ls ? lN : ls;
  }
  real tstep = (Ri - c.CRi0) * c.rCRid;
  tstep = tstep < real(0.) ? real(0.) : (tstep > real(1.) ? real(1.) : tstep);
  const real sg = Ri < real(0.) ? c.Cun[3] : c.Clo[3] + (c.Chi[3] - c.Clo[3]) * tstep;
  ls = cdiv(ls, sg);
  const real l = lh > ls ? lh : ls;
  return l < q.Hcol ? l : q.Hcol;
}
// the vertical derivatives of u at the x faces i, i+1 and of v at the y faces j, j+1 of column (i, j) on face kf (1 .. Nz-1):
// zero where one of the two nodes is an inactive node (both cells beside it inactive)
struct CatkeShear { real uw, ue, vs, vn; };
// the four face velocities of column (i, j) on one level: oc, ov = offsets of cell (i, j, k) in a cell-shaped / v-shaped array.
// The column kernels march up and keep the level below the face in registers: four loads per face and field instead of eight.
__device__ __forceinline__ CatkeShear catke_face_velocities(const Grid& g, const real* __restrict__ u, const real* __restrict__ v,
                                                            int oc, int ov) {
  CatkeShear r;
  r.uw = u[oc]; r.ue = u[oc + 1]; r.vs = v[ov]; r.vn = v[ov + g.sx];
  return r;
}
__device__ __forceinline__ CatkeShear catke_dz_velocities(const Grid& g, const CatkeColumn& q, const CatkeShear& lo,
                                                          const CatkeShear& hi, int kf) {
  const real rdz = uniform_at(g.rdzf, kf);
  // (every load unconditional -- the masks are per lane, and a load inside a divergent branch is issued behind it -- then selects)
  const real a = (hi.uw - lo.uw) * rdz, b = (hi.ue - lo.ue) * rdz;
  const real c = (hi.vs - lo.vs) * rdz, e = (hi.vn - lo.vn) * rdz;
  CatkeShear d;
  d.uw = kf > q.NUw ? a : real(0.);
  d.ue = kf > q.NUe ? b : real(0.);
  d.vs = kf > q.NVs ? c : real(0.);
  d.vn = kf > q.NVn ? e : real(0.);
  return d;
}
__device__ __forceinline__ CatkeShear catke_dz_velocities(const Grid& g, const CatkeColumn& q, const real* __restrict__ u,
                                                          const real* __restrict__ v, int oc, int ov, int kf) {
  return catke_dz_velocities(g, q, catke_face_velocities(g, u, v, oc - g.pl_c, ov - g.pl_v), catke_face_velocities(g, u, v, oc, ov), kf);
}
// time_step_catke_equation!, first half (see the header).  One thread per own column and CHUNK of levels (blockIdx.z), marching
// up.  Nothing here is a recurrence -- what a level takes from the face below it is that face's own quantities -- so a chunk
// starts one level early, evaluates the face under its first level exactly as the chunk below does, and goes on: the same
// bits whatever the chunking.  A rank of a decomposition (360 x 360 columns: two waves per SIMD with one thread per column)
// takes four chunks; a single domain of a million columns one.  e* goes to `e_out`: with one chunk that is e itself (a column
// reads nobody else's e, and the old e of a cell is last needed by the face above it, evaluated before the cell); with
// several, a scratch array -- the chunk above reads the OLD e of the level under its first -- from which the implicit solve
// takes it (ImplicitVarFields::src_e).
#ifndef GB25_CATKE_MINW
#define GB25_CATKE_MINW 4
#endif
template <bool IMM>
__global__ __launch_bounds__(256, GB25_CATKE_MINW) void k_catke_tke_step(Grid g, CatkePar c, real dt, real C1, real C2,
                                                        const real* __restrict__ u, const real* __restrict__ v,
                                                        const real* __restrict__ um, const real* __restrict__ vm,
                                                        const real* e, real* e_out, const real* __restrict__ n2,
                                                        const real* __restrict__ Jb, const real* __restrict__ KU,
                                                        const real* __restrict__ KC, real* __restrict__ KE,
                                                        real* __restrict__ Le, const real* __restrict__ Gn,
                                                        real* __restrict__ Gm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  const int Nz = g.Nz, o2 = i2(g, i, j), pc = g.pl_c, pv = g.pl_v;
  const int klen = (Nz + (int)gridDim.z - 1) / (int)gridDim.z, k0 = (int)blockIdx.z * klen, k1 = min(Nz, k0 + klen);
  if (k0 >= k1) return;
  const int ks = k0 > 0 ? k0 - 1 : 0;   // (the level whose top face is the face under the chunk's first level)
  CatkeColumn q = catke_column<IMM>(g, i, j, o2, Jb[o2]);
  q.rjb = real(1.) / (q.jb + c.Jbmin);
  const bool xw = g.x_periodic && i < g.H, xe = g.x_periodic && i >= g.Nx - g.H;
  const bool ys = j == 0 && g.jws == 0, yn = j == g.Ny - 1 && (g.jwn == g.Ny || g.cv.north_fold);
  auto put = [&](real* a, int o, real x) {   // the cell and the halo cells its fill derives from it (a14)
    store_x_images(g, a, o, x, xw, xe);
    if (ys) store_x_images(g, a, o - g.sx, x, xw, xe);
    if (yn) store_x_images(g, a, o + g.sx, x, xw, xe);
  };
  int o = ic(g, i, j, ks), ov = iv(g, i, j, ks);
  if (k0 == 0) KE[o] = real(0.);
  real zf = uniform_at(g.zc, 0) - real(0.5) * uniform_at(g.dzc, 0);
  for (int l = 0; l < ks; l++) zf += uniform_at(g.dzc, l);   // (the face heights by the same running sum whatever the chunk)
  // the face below the current cell: N^2, S^2, the convective dissipation length, -kappa_c N^2, the shear-production sum
  real N2lo = real(0.), S2lo = real(0.), cDlo = real(0.), wblo = real(0.), PFlo = real(0.);
  real ecur = e[o];
  // carried up the column: the face velocities of the level below the face, N^2 of the face (loaded as the face above of the
  // one before), sqrt(max(e_min, e)) of the cell below
  CatkeShear ulo = catke_face_velocities(g, u, v, o, ov), mlo = catke_face_velocities(g, um, vm, o, ov);
  real n2f = Nz > 1 ? n2[o + pc] : real(0.);
  real wcur = csqrt(ecur > c.emin ? ecur : c.emin);
  for (int k = ks; k < k1; k++) {
    const bool own = k >= k0;           // (false in the one level a chunk starts early: its top face only)
    const int kf = k + 1, of = o + pc, ovf = ov + pv;
    zf += uniform_at(g.dzc, k);
    real N2hi = real(0.), S2hi = real(0.), cDhi = real(0.), wbhi = real(0.), PFhi = real(0.), kehi = real(0.);
    real enext = real(0.), wnext = real(0.);
    const real gn = Gn[o], gm = Gm[o];   // (with the level's other loads: behind the face's arithmetic they were a sleep of their own)
    if (kf < Nz) {
      enext = e[of];
      const CatkeShear uhi = catke_face_velocities(g, u, v, of, ovf), mhi = catke_face_velocities(g, um, vm, of, ovf);
      const CatkeShear d = catke_dz_velocities(g, q, ulo, uhi, kf), dm = catke_dz_velocities(g, q, mlo, mhi, kf);
      ulo = uhi;
      mlo = mhi;
      S2hi = (d.uw * d.uw + d.ue * d.ue) / real(2.) + (d.vs * d.vs + d.vn * d.vn) / real(2.);
      const real kuc = KU[of];
      const real nw = (KU[of - 1] + kuc) / real(2.), ne = (kuc + KU[of + 1]) / real(2.);
      const real ns = (KU[of - g.sx] + kuc) / real(2.), nn = (kuc + KU[of + g.sx]) / real(2.);
      const real dzf = uniform_at(g.dzf, kf);
      // (nu dz u- dzf dz u+) + (nu dz u+ dzf dz u+), averaged over the two x faces, plus the same over the two y faces
      const real fw = (nw * dm.uw * dzf * d.uw) + (nw * d.uw * dzf * d.uw), fe = (ne * dm.ue * dzf * d.ue) + (ne * d.ue * dzf * d.ue);
      const real fs = (ns * dm.vs * dzf * d.vs) + (ns * d.vs * dzf * d.vs), fn = (nn * dm.vn * dzf * d.vn) + (nn * d.vn * dzf * d.vn);
      PFhi = (fw + fe) / real(2.) + (fs + fn) / real(2.);
      // (loads and the evaluation for every lane, the mask of the solid applied to the results: N^2 is stored as zero on the
      // faces that touch the solid, so the loaded values are harmless there)
      const real n2a = n2[of + pc], kcf = KC[of];
      wnext = csqrt(enext > c.emin ? enext : c.emin);
      const CatkeLengths L = catke_face_eval(c, q, wcur, wnext, n2f, n2a, S2hi, zf);
      const bool open = kf > q.kc0;
      N2hi = open ? n2f : real(0.);
      kehi = open ? L.ke : real(0.);
      cDhi = open ? L.convD : real(0.);
      wbhi = open ? -(kcf * n2f) : real(0.);
      n2f = n2a;
    }
    if (own) KE[of] = kehi;
    real Lk = real(0.);
    if (own && k >= q.kc0) {
      const real ek = ecur, wb = (wblo + wbhi) / real(2.);
      const real wbm = wb < real(0.) ? wb : real(0.), wbp = wb > real(0.) ? wb : real(0.);
      const real lD = catke_dissipation_length(c, q, wcur, uniform_at(g.zc, k), N2lo, N2hi, S2lo, S2hi, cDlo, cDhi);
      // (sqrt|e| and sqrt(max(e, 0)): the carried root where e >= e_min, which is nearly everywhere)
      const real ep = ek > real(0.) ? ek : real(0.);
      const bool floored = !(ek >= c.emin);
      const real wabs = floored ? csqrt(rabs(ek)) : wcur, wpos = floored ? csqrt(ep) : wcur;
      const real omega = ek < real(0.) ? c.rtau_neg : cdiv(wabs, lD);
      const real divJ = k == q.kc0 ? -(c.CWeps * wpos * uniform_at(g.rdzc, k)) : real(0.);      // (the bottom cell of the column)
      Lk = (ek > c.emin ? cdiv(wbm, ek) : real(0.)) - omega + divJ;
      const real P = ((PFlo + PFhi) / real(2.)) * (real(0.5) * uniform_at(g.rdzc, k));
      const real total = gn + (P + wbp);
      e_out[o] = ek + dt * (C1 * total - C2 * gm);
      Gm[o] = total;
    }
    if (own) {
      put(Le, o, Lk);
      if (k == 0) store_x_images(g, Le, o - pc, Lk, xw, xe);        // bottom / top layer (interior rows only, like the fill)
      if (k == Nz - 1) store_x_images(g, Le, o + pc, Lk, xw, xe);
    }
    N2lo = N2hi; S2lo = S2hi; cDlo = cDhi; wblo = wbhi; PFlo = PFhi;
    ecur = enext;
    wcur = wnext;
    o = of;
    ov = ovf;
  }
}
// compute_average_surface_buoyancy_flux! on the own columns (+ the halo cells the fill derives): J^b* from the top flux
// boundary conditions of T, S and the surface cell's alpha, beta; the dissipation length of the top cell from the NEW e and
// the OLD J^b; J^b <- (J^b + eps J^b*) / (1 + eps), eps = dt_since / cbrt(l_D^2 / max(J^b_min, J^b, J^b*)).  Land: zero.
// Also leaves the top boundary condition of e, (C^W_u* u*^3 + C^W_wD max(J^b*, 0) dz) / dz, for k_catke_add_top_source.
template <bool IMM>
__global__ void k_catke_surface_flux(Grid g, CatkePar c, real dt_since, const real* __restrict__ u, const real* __restrict__ v,
                                     const real* __restrict__ T, const real* __restrict__ S, const real* __restrict__ e,
                                     const real* __restrict__ n2, real* __restrict__ Jb, real* __restrict__ src) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  const int Nz = g.Nz, o2 = i2(g, i, j), k = Nz - 1, o = ic(g, i, j, k), ov = iv(g, i, j, k);
  const real J = Jb[o2];
  CatkeColumn q = catke_column<IMM>(g, i, j, o2, J);
  q.rjb = real(1.) / (q.jb + c.Jbmin);
  real Jnew = real(0.), source = real(0.);
  if (q.kc0 < Nz) {
    double Js = 0.0;
    if (g.top_flux[2] || g.top_flux[3]) {
      double al, be;
      teos10_level_sens(g.eos + 28 * k, sqrt_pos(((double)S[o] + 32.0) * (0.875 / 35.16504)), (double)T[o] * 0.025, al, be);
      const double JT = g.top_flux[2] ? (double)g.top_flux[2][o2] : 0.0, JS = g.top_flux[3] ? (double)g.top_flux[3][o2] : 0.0;
      Js = (double)g.g * (al * JT - be * JS) / (double)g.rho0;
    }
    const real Jstar = (real)Js;
    // the face below the top cell (kf = Nz-1) and the surface face (nothing there)
    real N2lo = real(0.), S2lo = real(0.), cDlo = real(0.);
    const real ek = e[o];
    if (k >= 1) {
      const CatkeShear d = catke_dz_velocities(g, q, u, v, o, ov, k);
      S2lo = (d.uw * d.uw + d.ue * d.ue) / real(2.) + (d.vs * d.vs + d.vn * d.vn) / real(2.);
      if (k > q.kc0) {
        N2lo = n2[o];
        const real eb = e[o - g.pl_c];
        const real el = eb > c.emin ? eb : c.emin, eh = ek > c.emin ? ek : c.emin;
        cDlo = catke_face_eval(c, q, csqrt(el), csqrt(eh), N2lo, real(0.), S2lo, uniform_at(g.zc, k) - real(0.5) * uniform_at(g.dzc, k)).convD;
      }
    }
    const real lD = catke_dissipation_length(c, q, csqrt(ek > c.emin ? ek : c.emin), uniform_at(g.zc, k), N2lo, real(0.), S2lo, real(0.), cDlo, real(0.));
    real Jp = c.Jbmin;
    Jp = J > Jp ? J : Jp;
    Jp = Jstar > Jp ? Jstar : Jp;
    const real tstar = cbrt(lD * lD / Jp), eps = dt_since / tstar;
    Jnew = (J + eps * Jstar) / (real(1.) + eps);
    // u* from the boundary-condition values AT (i, j), as Oceananigans' friction_velocity takes them
    const real Ju = g.top_flux[0] ? g.top_flux[0][o2] : real(0.);
    const real Jv = g.top_flux[1] ? g.top_flux[1][o2] : real(0.);
    const real us2 = sqrt(Ju * Ju + Jv * Jv), us3 = us2 * sqrt(us2);
    const real wD3 = (Jstar > real(0.) ? Jstar : real(0.)) * uniform_at(g.dzc, k);
    source = (c.CWu * us3 + c.CWw * wD3) / uniform_at(g.dzc, k);
  }
  src[o2] = source;
  // (a rank of a decomposition: the x halo columns / rows of its open sides / the rows beyond a zipper fold arrive by exchange)
  const bool xw = g.x_periodic && i < g.H, xe = g.x_periodic && i >= g.Nx - g.H;
  store_x_images(g, Jb, o2, Jnew, xw, xe);
  // (the y layers exist next to walls only)
  if (j == 0 && g.jws == 0) store_x_images(g, Jb, o2 - g.sx, Jnew, xw, xe);
  if (j == g.Ny - 1 && (g.jwn == g.Ny || g.cv.north_fold) && !(g.cv.north_fold && !g.x_periodic)) store_x_images(g, Jb, o2 + g.sx, Jnew, xw, xe);
}
// G^n.e of the top cell += the surface TKE flux / dz (the top boundary condition of e: compute_boundary_tendencies)
__global__ void k_catke_add_top_source(Grid g, const real* __restrict__ src, real* __restrict__ Ge) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= g.Ny) return;
  const int o = ic(g, i, j, g.Nz - 1);
  Ge[o] = Ge[o] + src[i2(g, i, j)];
}
// compute_CATKE_diffusivities!: kappa_u, kappa_c, kappa_e on the faces 1 .. Nz-1 (zero on the bottom and top faces and where
// the face touches the solid).
template <bool IMM>
__global__ __launch_bounds__(256, GB25_CATKE_MINW) void k_catke_diffusivities(Grid g, CatkePar c, const real* __restrict__ u,
                                                             const real* __restrict__ v, const real* __restrict__ e,
                                                             const real* __restrict__ n2, const real* __restrict__ Jb,
                                                             real* __restrict__ KU, real* __restrict__ KC,
                                                             real* __restrict__ KE, int i_lo, int i_hi, int j_lo, int j_hi) {
  // Columns i_lo .. i_hi-1, rows j_lo .. j_hi-1.  Single domain: the interior, the halo cells written as images.  A rank of a
  // decomposition: the first halo column on either side (and the first halo row of an open side / beyond the zipper fold) as
  // well -- kappa_u there is averaged into the implicit solves of u, v and into the shear production of the edge columns, and
  // is COMPUTED from the exchanged halos of e, u, v, T, S and J^b, bit for bit what its owner computes, instead of exchanged.
  const int i = i_lo + (int)(blockIdx.x * blockDim.x + threadIdx.x), j = j_lo + (int)(blockIdx.y * blockDim.y + threadIdx.y);
  if (i >= i_hi || j >= j_hi) return;
  const bool own = i >= 0 && i < g.Nx && j >= 0 && j < g.Ny;
  const int Nz = g.Nz, o2 = i2(g, i, j), pc = g.pl_c, pv = g.pl_v;
  CatkeColumn q = catke_column<IMM>(g, i, j, o2, Jb[o2]);
  q.rjb = real(1.) / (q.jb + c.Jbmin);
  const bool xw = own && g.x_periodic && i < g.H, xe = own && g.x_periodic && i >= g.Nx - g.H;
  // (the y layers exist next to walls only)
  const bool ys = own && j == 0 && g.jws == 0, yn = own && j == g.Ny - 1 && j_hi == g.Ny && (g.jwn == g.Ny || g.cv.north_fold);
  auto put = [&](real* a, int o, real x) {   // the cell and the halo cells its fill derives from it (a14)
    store_x_images(g, a, o, x, xw, xe);
    if (ys) store_x_images(g, a, o - g.sx, x, xw, xe);
    if (yn) store_x_images(g, a, o + g.sx, x, xw, xe);
  };
  // (chunks of levels in blockIdx.z, as in k_catke_tke_step: level k makes the face above it from loads alone)
  const int klen = (Nz + (int)gridDim.z - 1) / (int)gridDim.z, k0 = (int)blockIdx.z * klen, k1 = min(Nz, k0 + klen);
  if (k0 >= k1) return;
  int o = ic(g, i, j, k0), ov = iv(g, i, j, k0);
  if (k0 == 0) { put(KU, o, real(0.)); put(KC, o, real(0.)); put(KE, o, real(0.)); }
  real zf = uniform_at(g.zc, 0) - real(0.5) * uniform_at(g.dzc, 0);
  for (int l = 0; l < k0; l++) zf += uniform_at(g.dzc, l);
  real ecur = e[o];
  CatkeShear ulo = catke_face_velocities(g, u, v, o, ov);      // (carried up the column like k_catke_tke_step's)
  real n2f = Nz > 1 ? n2[o + pc] : real(0.);
  real wcur = csqrt(ecur > c.emin ? ecur : c.emin);
  // (e of the level above comes one level early: its square root is the head of the face's arithmetic, and loaded with the level's
  // other values the compiler waited for it alone, ahead of them -- a second sleep per level; the top halo layer exists in the parent)
  real e_above = e[o + pc];
  for (int k = k0; k < k1; k++) {
    const int kf = k + 1, of = o + pc, ovf = ov + pv;
    zf += uniform_at(g.dzc, k);
    CatkeLengths L = {real(0.), real(0.), real(0.), real(0.)};
    real wnext = real(0.);
    const real enext = e_above;
    e_above = e[of + pc];
    if (kf < Nz) {
      const CatkeShear uhi = catke_face_velocities(g, u, v, of, ovf);
      const CatkeShear d = catke_dz_velocities(g, q, ulo, uhi, kf);
      ulo = uhi;
      const real S2 = (d.uw * d.uw + d.ue * d.ue) / real(2.) + (d.vs * d.vs + d.vn * d.vn) / real(2.);
      const real n2a = n2[of + pc];
      wnext = csqrt(enext > c.emin ? enext : c.emin);
      const CatkeLengths Lf = catke_face_eval(c, q, wcur, wnext, n2f, n2a, S2, zf);   // (for every lane; the solid's mask below)
      n2f = n2a;
      if (kf > q.kc0) L = Lf;
    }
    put(KU, of, L.ku); put(KC, of, L.kc); put(KE, of, L.ke);
    wcur = wnext;
    o = of;
    ov = ovf;
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
  const real* src_e;           // null, or where k_catke_tke_step left e* (MODE 1, the e slice: the solve reads it there, writes e)
  real* P;                     // null, or the look-ahead's chunk sums (UvAhead::P): the sums of the NEW u dz, v dz per chunk of levels
  int plane2;                  // replace the look-ahead's there (w on the fly: k_w_bases takes w at the chunk boundaries from them)
  int z0;                      // first slice of the launch (MODE 1: 0 = T with S [+ e], 1 = e alone)
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
__device__ __forceinline__ real implicit_var_colsum(const Grid& g, int kchunks, Get x, real* part = nullptr, size_t pstride = 0, bool zero = false) {
  const int Nz = g.Nz, klen = (Nz + kchunks - 1) / kchunks;
  real tot = real(0.), q = real(0.);
  int kk = 0, ch = 0;
  for (int k = 0; k < Nz; k++) {
    q = (kk == 0) ? uniform_at(g.dzc, k) * x(k) : rfma(uniform_at(g.dzc, k), x(k), q);
    if (++kk == klen || k == Nz - 1) {
      tot = (k < klen) ? q : tot + q;
      if (part) part[(size_t)ch * pstride] = zero ? real(0.) : q;   // (the chunk's own sum: see ImplicitVarFields::P)
      ch++;
      kk = 0;
    }
  }
  return tot;
}
template <int NZT, bool IMM, int MODE>
__global__ __launch_bounds__(256) void k_implicit_vertical_var(Grid g, ImplicitVarFields A) {
  const int z = blockIdx.z + A.z0, Nz = g.Nz;
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
  const real* Fsrc = (tke && A.src_e != nullptr) ? A.src_e : Fa;
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
    if (!(MODE == 1 && ab2)) a[k] = (k < Nz) ? Fsrc[o0 + k * pl] : real(0.);
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
      const real t = (k == kf) ? real(0.) : -(kup * uniform_at(g.rdzf, k));    // coupling through face k, without the cell height
      const real lo = t * uniform_at(g.rdzc, k);
      const real up = -(ktop * uniform_at(g.rdzc, k)) * uniform_at(g.rdzf, k + 1);         // (ktop = 0 at the top level)
      real dg = real(1.) - lo - up;
      if (MODE == 1 && tke) dg -= A.dt * b[k];
      const real gk = (t * uniform_at(g.rdzc, k - 1)) * rbet;                   // upper coefficient of the level below / its pivot
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
    int kk = 0, ch = 0;
    const bool wall = vsh && j == g.jws;
#pragma unroll
    for (int k = 0; k < NZT; k++)
      if (k < Nz) {
        q = (kk == 0) ? uniform_at(g.dzc, k) * a[k] : rfma(uniform_at(g.dzc, k), a[k], q);
        if (++kk == klen || k == Nz - 1) {
          tot = (k < klen) ? q : tot + q;
          if (A.P) A.P[((size_t)(2 + z) * A.kchunks + ch) * A.plane2 + o2] = wall ? real(0.) : q;
          ch++;
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
  const int z = blockIdx.z + A.z0, Nz = g.Nz;
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  const bool vsh = MODE == 0 && z == 1, pair = MODE == 1 && z == 0, tke = MODE == 1 && z == 1;
  if (i >= g.Nx || j >= (vsh ? g.Ny : g.Ny)) return;
  const int o2 = i2(g, i, j), kf = implicit_var_first_level<IMM, MODE>(g, z, o2, j);
  real* Fa = MODE == 0 ? A.f[z] : (pair ? A.f[2] : A.f[4]);
  real* Fb = A.f[3];
  real* G = A.gam[blockIdx.z];
  const real* K = MODE == 0 ? A.KU : (pair ? A.KC : A.KE);
  const int pl = vsh ? g.pl_v : g.pl_c, o0 = vsh ? iv(g, i, j, 0) : ic(g, i, j, 0), oc = ic(g, i, j, 0), pc = g.pl_c;
  const int og = iv(g, i, j, 0), pg = g.pl_v;              // the scratch array has the larger of the two shapes
  const int nb = MODE == 0 ? (z == 0 ? -1 : -g.sx) : 0;
  const bool ab2 = tke && A.GnE != nullptr;
  if (ab2)
    for (int k = 0; k < kf; k++)
      Fa[o0 + k * pl] = ab2_advance(Fa[o0 + k * pl], A.GnE[o0 + k * pl], A.GmE[o0 + k * pl], A.dt, A.C1, A.C2);
  real rbet = real(1.), pa = real(0.), pb = real(0.), kup = real(0.);
  const real* Fsrc = (tke && A.src_e != nullptr) ? A.src_e : Fa;
#pragma unroll 4
  for (int k = kf; k < Nz; k++) {
    real xa = Fsrc[o0 + k * pl], xb = real(0.), le = real(0.);
    if (MODE == 1 && pair) xb = Fb[o0 + k * pl];
    if (MODE == 1 && tke) le = A.dt * A.Le[o0 + k * pl];
    if (MODE == 1 && ab2) xa = ab2_advance(xa, A.GnE[o0 + k * pl], A.GmE[o0 + k * pl], A.dt, A.C1, A.C2);
    real ktop = real(0.);
    if (k < Nz - 1) {
      const int of = oc + (k + 1) * pc;
      ktop = A.dt * (MODE == 0 ? (K[of + nb] + K[of]) / real(2.) : K[of]);
    }
    const real t = (k == kf) ? real(0.) : -(kup * uniform_at(g.rdzf, k));
    const real lo = t * uniform_at(g.rdzc, k);
    const real up = -(ktop * uniform_at(g.rdzc, k)) * uniform_at(g.rdzf, k + 1);
    const real dg = real(1.) - lo - up - le;
    const real gk = (t * uniform_at(g.rdzc, k - 1)) * rbet;
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
    const bool wall = vsh && j == g.jws;
    const real tot = implicit_var_colsum(g, A.kchunks, [&](int k) { return Fa[o0 + k * pl]; },
                                         A.P ? A.P + (size_t)(2 + z) * A.kchunks * A.plane2 + o2 : nullptr, (size_t)A.plane2, wall);
    A.sum[z][o2] = wall ? real(0.) : tot;
  }
}


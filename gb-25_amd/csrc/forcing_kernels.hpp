// forcing_kernels.hpp -- the data-free forcing: similarity-theory atmosphere-ocean fluxes per surface cell and their
// interpolation onto the velocity faces.  Included by kernels.hpp, inside namespace gb25.
#pragma once
// =============================================================================================
// Data-free forcing (GB-25 src/data_free_ocean_climate_model.jl:12-70): the atmosphere-ocean fluxes of
// ClimaOcean's OceanSeaIceModel -- SimilarityTheoryFluxes(FixedIterations(5)) + Radiation -- from a PrescribedAtmosphere held
// at the ocean's cell centres, written into the top flux boundary conditions of u, v, T, S (a13).  The algorithm is the
// oracle's (oracle/gb25_oracle.c, "data-free forcing": Monin-Obukhov similarity theory with the COARE 3.5 roughness lengths
// and stability functions, restated [UPSTREAM-UNVERIFIED]); a 2-D computation, all of it in fp64.
//   k_similarity_fluxes   one thread per surface cell, one halo column to the west and one row to the south (and the row
//                         beyond a zipper fold) included -- computed there from the halo cells of u, v, T, S, never exchanged:
//                         J^T, J^S at the centres, the stress components into two scratch arrays
//   k_stress_to_faces     J^u = -Ix(tau_x) / rho0 on the x faces, J^v = -Iy(tau_y) / rho0 on the y faces
// =============================================================================================
struct Atmosphere {
  const double* a[7];   // u_a, v_a, T_a [K], q_a, p_a, shortwave, longwave: parent layout of a 2-D (c,c) field
};
template <class F>
__device__ __forceinline__ F ao_psi_c(F y) {
  const F r3 = F(1.7320508075688772);
  return F(1.5) * log((1 + y + y * y) / 3) - r3 * atan((1 + 2 * y) / r3) + F(3.141592653589793) / r3;
}
template <class F>
__device__ __forceinline__ F ao_psi_u(F z) {
  if (z < 0) {
    const F x = sqrt(sqrt(1 - 15 * z)), pk = 2 * log((1 + x) / 2) + log((1 + x * x) / 2) - 2 * atan(x) + F(1.5707963267948966);
    const F f = z * z / (1 + z * z);
    return (1 - f) * pk + f * ao_psi_c<F>(cbrt(1 - F(10.15) * z));
  }
  const F dz = F(0.35) * z < 50 ? F(0.35) * z : F(50);
  return -(F(0.7) * z + F(0.75) * (z - F(5 / 0.35)) * exp(-dz) + F(0.75 * 5 / 0.35));
}
template <class F>
__device__ __forceinline__ F ao_psi_q(F z) {
  if (z < 0) {
    const F x = sqrt(1 - 15 * z), pk = 2 * log((1 + x) / 2);
    const F f = z * z / (1 + z * z);
    return (1 - f) * pk + f * ao_psi_c<F>(cbrt(1 - F(34.15) * z));
  }
  const F dz = F(0.35) * z < 50 ? F(0.35) * z : F(50);
  const F t = 1 + F(2.0 / 3.0) * z;
  return -(t * sqrt(t) + F(2.0 / 3.0) * (z - F(14.28)) * exp(-dz) + F(8.525));
}
template <bool IMM>
__global__ void k_similarity_fluxes(Grid g, Atmosphere A, const real* __restrict__ u, const real* __restrict__ v,
                                    const real* __restrict__ T, const real* __restrict__ S, double* __restrict__ tx,
                                    double* __restrict__ ty, real* __restrict__ JT, real* __restrict__ JS, int j_hi,
                                    int iterations) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x) - 1, j = (int)(blockIdx.y * blockDim.y + threadIdx.y) - 1;
  if (i >= g.Nx || j >= j_hi) return;
  const int o2 = i2(g, i, j), k = g.Nz - 1, o = ic(g, i, j, k), ov = iv(g, i, j, k);
  const bool own = i >= 0 && j >= 0 && j < g.Ny;
  if (IMM && (int)(g.im.ordA[o2] & 255) >= g.Nz) {   // land
    tx[o2] = 0.0;
    ty[o2] = 0.0;
    if (own) { JT[o2] = real(0.); JS[o2] = real(0.); }
    return;
  }
  const double kap = 0.4, Rd = 287.0, Rv = 461.5, cpd = 1005.0, cpv = 1859.0, cpl = 4181.0, Lv0 = 2.5008e6, T0 = 273.16,
               ptr = 611.657, h = 10.0, zi = 600.0, beta = 1.2, charnock = 0.011, nu = 1.5e-5, emis = 0.97, albedo = 0.05,
               sigma = 5.670374419e-8, cpo = 3991.86795711963, rho_fw = 1000.0;
  const double grav = (double)g.g, rho0 = (double)g.rho0;
  const double ua = A.a[0][o2], va = A.a[1][o2], Ta = A.a[2][o2], qa = A.a[3][o2], pa = A.a[4][o2], Qsw = A.a[5][o2], Qlw = A.a[6][o2];
  const double uo = ((double)u[o] + (double)u[o + 1]) / 2, vo = ((double)v[ov] + (double)v[ov + g.sx]) / 2;
  const double To = (double)T[o], So = (double)S[o];
  const double Ts = To + 273.15, eps = Rd / Rv;
  const double psat = ptr * pow(Ts / T0, (cpv - cpl) / Rv) * exp((Lv0 - (cpv - cpl) * T0) / Rv * (1 / T0 - 1 / Ts));
  const double qs = 0.98 * eps * psat / (pa - (1 - eps) * psat);
  const double rho_a = pa / ((Rd * (1 - qa) + Rv * qa) * Ta), cpm = cpd * (1 - qa) + cpv * qa, Lv = Lv0 + (cpv - cpl) * (Ts - T0);
  const double du = ua - uo, dv = va - vo, dth = Ta + grav / cpm * h - Ts, dq = qa - qs, Tv = Ta * (1 + 0.608 * qa);
  double U = sqrt(du * du + dv * dv + 0.2 * 0.2);
  const double chi0 = log(h / 1e-4);
  double us = kap * U / chi0, ths = kap * dth / chi0, qst = kap * dq / chi0;
  // The five iterations in the float type of the model (ClimaOcean iterates in it too): in Float32 the transcendental
  // functions of the loop are single-precision ones, several times cheaper than their fp64 siblings (the kernel was
  // 0.35 ms at 1440 x 720, all of it fp64 log / pow / atan / cbrt); thermodynamics and the fluxes themselves stay in fp64.
  using F = real;
  {
    const F kapF = (F)kap, hF = (F)h, gravF = (F)grav, rTv = (F)(1 / Tv), qaF = (F)qa, TaF = (F)Ta;
    const F duF = (F)du, dvF = (F)dv, dthF = (F)dth, dqF = (F)dq, nuF = (F)nu;
    F usF = (F)us, thsF = (F)ths, qstF = (F)qst, UF = (F)U;
    for (int it = 0; it < iterations; it++) {
      const F bs = gravF * rTv * (thsF * (1 + F(0.608) * qaF) + F(0.608) * TaF * qstF), Jb = -usF * bs;
      const F Ug = fmax(F(0.2), (F)beta * cbrt(fmax(Jb, F(0.)) * (F)zi));
      UF = sqrt(duF * duF + dvF * dvF + Ug * Ug);
      const F lu = (F)charnock * usF * usF / gravF + F(0.11) * nuF / usF;
      const F lq = fmin(F(1.6e-4), F(5.8e-5) / pow(lu * usF / nuF, F(0.72)));
      F zeta = kapF * hF * bs / (usF * usF);
      zeta = zeta > 50 ? F(50) : (zeta < -50 ? F(-50) : zeta);
      const F chiu = log(hF / lu) - ao_psi_u<F>(zeta) + ao_psi_u<F>(zeta * lu / hF);
      const F chiq = log(hF / lq) - ao_psi_q<F>(zeta) + ao_psi_q<F>(zeta * lq / hF);
      usF = kapF * UF / chiu;
      thsF = kapF * dthF / chiq;
      qstF = kapF * dqF / chiq;
    }
    us = (double)usF; ths = (double)thsF; qst = (double)qstF; U = (double)UF;
  }
  tx[o2] = rho_a * us * us * du / U;
  ty[o2] = rho_a * us * us * dv / U;
  if (own) {
    const double Qc = -rho_a * cpm * us * ths, Qv = -rho_a * Lv * us * qst, E = -rho_a * us * qst;
    const double Q = Qc + Qv + emis * (sigma * Ts * Ts * Ts * Ts - Qlw) - (1 - albedo) * Qsw;
    JT[o2] = (real)(Q / (rho0 * cpo));
    JS[o2] = (real)(-So * E / rho_fw);
  }
}
__global__ void k_stress_to_faces(Grid g, const double* __restrict__ tx, const double* __restrict__ ty, real* __restrict__ Ju,
                                  real* __restrict__ Jv, int j_hi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= g.Nx || j >= j_hi) return;
  const int o2 = i2(g, i, j);
  if (j < g.Ny) Ju[o2] = (real)(-(tx[o2 - 1] + tx[o2]) / 2 / (double)g.rho0);
  Jv[o2] = (real)(-(ty[o2 - g.sx] + ty[o2]) / 2 / (double)g.rho0);
}


// Quadratic bottom drag: the bottom boundary condition ClimaOcean's ocean_simulation gives u and v (and their immersed bottoms),
// FluxBoundaryCondition(-Cd u sqrt(u^2 + Ixy(v)^2)) at the first free level of the face's column (oracle:
// compute_bottom_drag_fluxes).  One thread per column; x faces [i_first, i_first + n) (a slab computes its face 0 -- the one
// that reads a halo column of v -- after the halos arrived).  The tendency kernels add J / dz at that level.
template <bool IMM>
__global__ void k_bottom_drag_flux(Grid g, const real* __restrict__ u, const real* __restrict__ v, real* __restrict__ Ju,
                                   real* __restrict__ Jv, real Cd, int i_first, int n) {
  const int i = i_first + (int)(blockIdx.x * blockDim.x + threadIdx.x), j = blockIdx.y * blockDim.y + threadIdx.y;
  if (i >= i_first + n || j >= g.Ny) return;
  const int o2 = i2(g, i, j);
  int ku = 0, kv = 0;
  if (IMM) {
    const unsigned C = g.im.ordC[o2];
    ku = (C >> 8) & 255;
    kv = (C >> 16) & 255;
  }
  if (j < g.Ny) {
    real J = real(0.);
    if (ku < g.Nz) {
      const int o = ic(g, i, j, ku), ov = iv(g, i, j, ku);
      const real uu = u[o], vb = (v[ov - 1] + v[ov] + v[ov - 1 + g.sx] + v[ov + g.sx]) / real(4.);
      J = -Cd * uu * sqrt(uu * uu + vb * vb);
    }
    Ju[o2] = J;
  }
  {
    real J = real(0.);
    if (j != g.jws && kv < g.Nz) {   // (the face on the southern wall never moves)
      const int o = ic(g, i, j, kv), ov = iv(g, i, j, kv);
      const real vv = v[ov], ub = (u[o - g.sx] + u[o - g.sx + 1] + u[o] + u[o + 1]) / real(4.);
      J = -Cd * vv * sqrt(vv * vv + ub * ub);
    }
    Jv[o2] = J;
  }
}

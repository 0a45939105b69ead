"""An independent numpy statement of the CATKE restatement (flat bottom, any horizontal grid): what one
compute_diffusivities! does to a model state -- the e step (substep_turbulent_kinetic_energy! + implicit solve), the
filtered surface buoyancy flux, the diffusivities.  Written from the formulas in the header comment of the CATKE section of
oracle/gb25_oracle.c, array-wise, sharing no code with the C loops; test infrastructure only.
[UPSTREAM-UNVERIFIED like the oracle itself: it pins the C to the stated formulas, not the formulas to Oceananigans.]"""
import numpy as np

DEFAULTS = dict(Cs=1.131, Cb=0.28, Csp=0.505, CRid=1.02, CRi0=0.254,
                Chi=(0.242, 0.098, 0.548, 0.579), Clo=(0.361, 0.198, 7.863, 1.604), Cun=(0.370, 0.369, 1.447, 0.923),
                Cc=(3.705, 4.793, 3.642, 3.254), Ce=(0.0, 0.112, 0.0, 0.0), CWu=3.179, CWw=0.383,
                emin=1e-9, Jbmin=1e-11, tau_neg=60.0, CWeps=1.0)
G, RHO0 = 9.80665, 1020.0


def sigma(P, p, Ri):
    st = np.clip((Ri - P["CRi0"]) / P["CRid"], 0.0, 1.0)
    return np.where(Ri < 0, P["Cun"][p], P["Clo"][p] + (P["Chi"][p] - P["Clo"][p]) * st)


class State:
    """Parent arrays (halos included) of the oracle model `m`, as float64 copies, and its vertical grid."""

    def __init__(self, m):
        b = m.backend
        self.H = H = b.H
        self.Nx, self.Ny, self.Nz = m.grid.size
        for name, key in (("u", "u"), ("v", "v"), ("T", "T"), ("S", "S"), ("e", "e"), ("ku", "kappa_u"), ("kc", "kappa_c"),
                          ("Jb", "Jb"), ("um", "previous_u"), ("vm", "previous_v"), ("Gn", "Gn.e"), ("Gm", "Gm.e")):
            setattr(self, name, np.array(b.get_field(key, True), dtype=np.float64))
        Nz = self.Nz
        self.zf = np.array([b.metric("zf", k) for k in range(1, Nz + 2)])
        self.zc = np.array([b.metric("zc", k) for k in range(1, Nz + 1)])
        self.dzc = np.array([b.metric("dzc", k) for k in range(1, Nz + 1)])
        self.dzf = np.array([b.metric("dzf", k) for k in range(1, Nz + 2)])     # dzf[f]: between the centres f-1 and f
        self.backend = b

    def cells(self, a, di=0, dj=0):
        """interior cells of the parent `a`, shifted by (di, dj) columns / rows: shape (Nx, Ny, Nz)"""
        H = self.H
        return a[H + di:H + di + self.Nx, H + dj:H + dj + self.Ny, H:H + self.Nz]

    def faces(self, a, di=0, dj=0):
        H = self.H
        return a[H + di:H + di + self.Nx, H + dj:H + dj + self.Ny, H:H + self.Nz + 1]


def sensitivities(st, T, S, Z):
    """(alpha, beta) by centred differences of the oracle's own rho (pinned by the published TEOS-10 check value)."""
    rho = np.vectorize(st.backend.teos10_rho)
    dT, dS = 1e-3, 1e-3
    a = -(rho(T + dT, S, Z) - rho(T - dT, S, Z)) / (2 * dT) / RHO0
    b = (rho(T, S + dS, Z) - rho(T, S - dS, Z)) / (2 * dS) / RHO0
    return a, b


def dz_faces(st, a, di=0, dj=0):
    """vertical derivative at the faces 0 .. Nz of the columns (i + di, j + dj); zero on the bottom and top faces"""
    c = st.cells(a, di, dj)
    out = np.zeros(c.shape[:2] + (st.Nz + 1,))
    out[:, :, 1:st.Nz] = (c[:, :, 1:] - c[:, :, :-1]) / st.dzf[1:st.Nz]
    return out


def face_quantities(st, P=DEFAULTS, e=None, Jb=None):
    """N2, S2, w* (three means), and the four convective lengths + three kappas at the faces 0 .. Nz of every interior column"""
    Nz = st.Nz
    e = st.cells(st.e) if e is None else e
    Jb = st.Jb[st.H:st.H + st.Nx, st.H:st.H + st.Ny, 0] if Jb is None else Jb
    T, S = st.cells(st.T), st.cells(st.S)
    N2 = np.zeros(T.shape[:2] + (Nz + 1,))
    Tf, Sf = 0.5 * (T[:, :, 1:] + T[:, :, :-1]), 0.5 * (S[:, :, 1:] + S[:, :, :-1])
    al, be = sensitivities(st, Tf, Sf, st.zf[1:Nz][None, None, :])
    N2[:, :, 1:Nz] = G * (al * (T[:, :, 1:] - T[:, :, :-1]) - be * (S[:, :, 1:] - S[:, :, :-1])) / st.dzf[1:Nz]
    S2 = 0.5 * (dz_faces(st, st.u) ** 2 + dz_faces(st, st.u, 1, 0) ** 2) + 0.5 * (dz_faces(st, st.v) ** 2 + dz_faces(st, st.v, 0, 1) ** 2)
    ef = np.maximum(e, P["emin"])
    mean = lambda x: np.concatenate([np.zeros(x.shape[:2] + (1,)), 0.5 * (x[:, :, 1:] + x[:, :, :-1]), np.zeros(x.shape[:2] + (1,))], axis=2)
    ws, ws2, ws3 = mean(np.sqrt(ef)), mean(ef), mean(ef ** 1.5)
    with np.errstate(divide="ignore", invalid="ignore"):
        Ri = np.where(N2 == 0, 0.0, N2 / S2)
        d = np.minimum(P["Cs"] * (st.zf[Nz] - st.zf), P["Cb"] * (st.zf - st.zf[0]))[None, None, :]
        ls = np.where(N2 > 0, np.minimum(d, ws / np.sqrt(np.where(N2 > 0, N2, 1.0))), d) * np.ones_like(N2)
        Jb3 = Jb[:, :, None]
        N2above = np.concatenate([N2[:, :, 1:], np.zeros(N2.shape[:2] + (1,))], axis=2)
        conv = (Jb3 > P["Jbmin"]) & (N2 < 0)
        entr = (Jb3 > P["Jbmin"]) & (N2 > 0) & (N2above < 0)
        esp = 1 - P["Csp"] * np.sqrt(S2) * ws2 / (Jb3 + P["Jbmin"])
        lconv = []
        for p in range(4):
            lc = np.maximum(esp * P["Cc"][p] * ws3 / (Jb3 + P["Jbmin"]), 0.0)
            le = np.maximum(esp * P["Ce"][p] * Jb3 / (ws * N2 + P["Jbmin"]), 0.0)
            lconv.append(np.where(conv, lc, np.where(entr, le, 0.0)))
    Hcol = st.zf[Nz] - st.zf[0]
    kap = []
    interior = np.zeros(Nz + 1, bool)
    interior[1:Nz] = True
    for p in range(3):
        l = np.minimum(Hcol, np.maximum(sigma(P, p, Ri) * ls, lconv[p]))
        kap.append(np.where(interior[None, None, :], l * ws, 0.0))
    convD = np.where(interior[None, None, :], lconv[3], 0.0)
    return dict(N2=N2, S2=S2, ku=kap[0], kc=kap[1], ke=kap[2], convD=convD)


def dissipation_length(st, F, e, P=DEFAULTS):
    Nz = st.Nz
    c = lambda x: 0.5 * (x[:, :, :-1] + x[:, :, 1:])
    N2, S2, lh = c(F["N2"]), c(F["S2"]), c(F["convD"])
    with np.errstate(divide="ignore", invalid="ignore"):
        Ri = np.where(N2 == 0, 0.0, N2 / S2)
        d = np.minimum(P["Cs"] * (st.zf[Nz] - st.zc), P["Cb"] * (st.zc - st.zf[0]))[None, None, :]
        wc = np.sqrt(np.maximum(e, P["emin"]))
        ls = np.where(N2 > 0, np.minimum(d, wc / np.sqrt(np.where(N2 > 0, N2, 1.0))), d) / sigma(P, 3, Ri)
    return np.minimum(st.zf[Nz] - st.zf[0], np.maximum(ls, lh))


def shear_production(st):
    """Ix [Iz(nu dz u- dzf dz u+) + Iz(nu dz u+ dzf dz u+)] / (2 dzc) + the same in y, nu = kappa_u (old) at the face's column"""
    def px(di):
        nu = 0.5 * (st.faces(st.ku, di - 1, 0) + st.faces(st.ku, di, 0))
        dm, dp = dz_faces(st, st.um, di, 0), dz_faces(st, st.u, di, 0)
        a, b = nu * dm * st.dzf * dp, nu * dp * st.dzf * dp
        return (0.5 * (a[:, :, :-1] + a[:, :, 1:]) + 0.5 * (b[:, :, :-1] + b[:, :, 1:])) / (2 * st.dzc)

    def py(dj):
        nu = 0.5 * (st.faces(st.ku, 0, dj - 1) + st.faces(st.ku, 0, dj))
        dm, dp = dz_faces(st, st.vm, 0, dj), dz_faces(st, st.v, 0, dj)
        a, b = nu * dm * st.dzf * dp, nu * dp * st.dzf * dp
        return (0.5 * (a[:, :, :-1] + a[:, :, 1:]) + 0.5 * (b[:, :, :-1] + b[:, :, 1:])) / (2 * st.dzc)

    return 0.5 * (px(0) + px(1)) + 0.5 * (py(0) + py(1))


def tke_step(st, dt, chi=0.1, P=DEFAULTS):
    """e after time_step_catke_equation!, with G^-.e, L^e and the kappa_e it solved with"""
    Nz = st.Nz
    e = st.cells(st.e)
    F = face_quantities(st, P)
    wbf = -st.faces(st.kc) * F["N2"]
    wb = 0.5 * (wbf[:, :, :-1] + wbf[:, :, 1:])
    lD = dissipation_length(st, F, e, P)
    with np.errstate(divide="ignore", invalid="ignore"):
        omega = np.where(e < 0, 1.0 / P["tau_neg"], np.sqrt(np.abs(e)) / lD)
        L = np.where(e > P["emin"], np.minimum(wb, 0.0) / e, 0.0) - omega
    L[:, :, 0] -= P["CWeps"] * np.sqrt(np.maximum(e[:, :, 0], 0.0)) / st.dzc[0]
    total = st.cells(st.Gn) + shear_production(st) + np.maximum(wb, 0.0)
    es = e + dt * ((1.5 + chi) * total - (0.5 + chi) * st.cells(st.Gm))
    ke = F["ke"]
    out = np.empty_like(es)
    for i in range(es.shape[0]):
        for j in range(es.shape[1]):
            A = np.zeros((Nz, Nz))
            for k in range(Nz):
                lo = -dt * ke[i, j, k] / (st.dzc[k] * st.dzf[k]) if k > 0 else 0.0
                up = -dt * ke[i, j, k + 1] / (st.dzc[k] * st.dzf[k + 1]) if k < Nz - 1 else 0.0
                A[k, k] = 1 - lo - up - dt * L[i, j, k]
                if k > 0:
                    A[k, k - 1] = lo
                if k < Nz - 1:
                    A[k, k + 1] = up
            out[i, j] = np.linalg.solve(A, es[i, j])
    return dict(e=out, Gm=total, Le=L, ke=ke)


def filtered_surface_flux(st, e_new, Jstar, dt_since, P=DEFAULTS):
    """J^b after compute_average_surface_buoyancy_flux! (the dissipation length of the top cell from the NEW e and the OLD J^b)"""
    F = face_quantities(st, P, e=e_new)
    lD = dissipation_length(st, F, e_new, P)[:, :, -1]
    J = st.Jb[st.H:st.H + st.Nx, st.H:st.H + st.Ny, 0]
    Jp = np.maximum(np.maximum(P["Jbmin"], J), Jstar)
    eps = dt_since / np.cbrt(lD ** 2 / Jp)
    return (J + eps * Jstar) / (1 + eps)

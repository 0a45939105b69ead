"""compare_states / sync_states! of the reference (GB-25 src/correctness.jl:4-103).

`isapprox(a, b; rtol, atol)` on Julia arrays is a NORM test:
    norm(a - b) <= max(atol, rtol * max(norm(a), norm(b)))
and that is what `approx_equal` implements.  The report line has the same content as the
reference's @printf (name, verdict, max|psi1|, max|psi2|, max|delta| and its 1-based index).
"""
import math

import numpy as np


def norm_error(a, b):
    """(||a-b||_2, max(||a||_2, ||b||_2)): the two sides of Julia's isapprox for arrays."""
    a64, b64 = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm((a64 - b64).ravel())), float(max(np.linalg.norm(a64.ravel()),
                                                                   np.linalg.norm(b64.ravel())))


def approx_equal(a, b, rtol, atol):
    d, n = norm_error(a, b)
    if not np.isfinite(d):
        return False
    return bool(d <= max(atol, rtol * n))


def _compare(name, psi1, psi2, rtol, atol, out):
    """compare_parent / compare_interior (src/correctness.jl:4-26): psi1 may be smaller than psi2."""
    nx, ny, nz = psi1.shape
    psi2 = psi2[:nx, :ny, :nz]
    delta = np.abs(psi1.astype(np.float64) - psi2.astype(np.float64))
    idx = np.unravel_index(np.argmax(delta), delta.shape) if delta.size else (0, 0, 0)
    ok = approx_equal(psi1, psi2, rtol, atol)
    dn, nn = norm_error(psi1, psi2)
    rec = dict(name=name, ok=ok, rel=(dn / nn if nn > 0 else (0.0 if dn == 0 else float("inf"))), max1=float(np.max(np.abs(psi1))), max2=float(np.max(np.abs(psi2))),
               maxdelta=float(delta.max()) if delta.size else 0.0, index=tuple(int(i) + 1 for i in idx))
    out.append(rec)
    return ok


def compare_states(m1, m2, *, rtol=None, atol=0.0, include_halos=False, throw_error=False, verbose=True):
    """compare_states(m1, m2; rtol=sqrt(eps(eltype(grid))), atol=0, include_halos, throw_error)
    -- src/correctness.jl:28-90.  Walks fields(model) = (u, v, w, eta, T, S), G^n and G^- of every name
    but w and eta, and the split-explicit filtered state (U, V, eta).  Returns (ok, report)."""
    if rtol is None:   # sqrt(eps(eltype(grid)))
        rtol = math.sqrt(np.finfo(getattr(m1.backend, "dtype", np.float32)).eps)
    get = (lambda f: f.parent) if include_halos else (lambda f: f.interior)
    report, ok = [], True
    f1, f2 = m1.fields(), m2.fields()
    for name in f1:
        ok &= _compare(name, get(f1[name]), get(f2[name]), rtol, atol, report)
        if name not in ("w", "eta"):
            ok &= _compare(f"Gn.{name}", get(getattr(m1.timestepper.Gn, name)), get(getattr(m2.timestepper.Gn, name)),
                           rtol, atol, report)
            ok &= _compare(f"Gm.{name}", get(getattr(m1.timestepper.Gm, name)), get(getattr(m2.timestepper.Gm, name)),
                           rtol, atol, report)
    for name in ("U", "V", "eta"):
        ok &= _compare(f"filtered.{name}", get(getattr(m1.free_surface.filtered_state, name)),
                       get(getattr(m2.free_surface.filtered_state, name)), rtol, atol, report)
    # if m1.closure isa CATKEVerticalDiffusivity: the diffusivity fields (src/correctness.jl:60-67)
    if getattr(m1, "diffusivity_fields", None) is not None and getattr(m2, "diffusivity_fields", None) is not None:
        for name in ("kappa_u", "kappa_c", "kappa_e", "Le", "Jb"):
            ok &= _compare(name, get(getattr(m1.diffusivity_fields, name)), get(getattr(m2.diffusivity_fields, name)),
                           rtol, atol, report)
    if verbose:
        for r in report:
            print("(%10s) psi1 ~ psi2: %-5s, max|psi1|, max|psi2|: %.9e, %.9e, max|d|: %.9e at %d %d %d"
                  % (r["name"], r["ok"], r["max1"], r["max2"], r["maxdelta"], *r["index"]))
    if not ok and throw_error:
        bad = [r["name"] for r in report if not r["ok"]]
        raise AssertionError(f"There is a discrepancy between the models: {bad} (rtol={rtol}, atol={atol})")
    return ok, report


def sync_states(m1, m2):
    """sync_states!(m1, m2): copy parent(field) of every field of m2 into m1 -- src/correctness.jl:92-103."""
    f1, f2 = m1.fields(), m2.fields()
    for name in f1:
        p2 = f2[name].parent
        nx, ny, nz = m1.backend.field_dims(f1[name].name, True)
        f1[name].set_parent(p2[:nx, :ny, :nz])

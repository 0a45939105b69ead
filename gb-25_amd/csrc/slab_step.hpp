// slab_step.hpp -- the time step of an x-slab decomposition, inside the library.
//
// Replaces what the reference gets from `Oceananigans.Distributed(arch; partition=Partition(Rx, Ry, 1))` + XLA's SPMD
// partitioner (GB-25 sharding/sharded_baroclinic_instability_simulation_run.jl:65-72,145-164): there `loop!(model,
// Ninner)` is ONE executable whose halo copies XLA turned into collective-permutes (ncclSend/Recv kernels); here
// gb25_loop on a slab is one call that sequences the stages of the step, the pack / unpack kernels and the
// point-to-point exchanges on two HIP streams.  Three transports move the packed columns:
//   * RCCL  (product, one process per GPU): ncclGroupStart; ncclSend/ncclRecv to west and east; ncclGroupEnd on the
//            stream the exchange belongs to.  librccl is loaded at run time (dlopen), so the library itself has no
//            link-time dependency and loads on a machine without RCCL (where only single-domain models are built).
//   * local (several slabs of one decomposition in ONE process, same device): device-to-device copies.  This is what the
//            decomposition-invariance tests run at full size on a one-GPU box.
//   * host callback (rehearsal of the multi-process path where RCCL cannot run, e.g. two ranks on one device): the host
//            moves the buffers; synchronous.
// The sequencing itself (`sequence_time_step`) is written against an abstract StepOps so that a dry run can record the
// order of operations without a GPU (gb25_debug_sequence; tests/test_distributed_cpu.py).
//
//   stage 0   AB2 update of u,v,T,S (adoption of the look-aheads), y/z layers of the 3-D bundle; pressure of the own
//             columns starts on the slab's side stream                                            (main stream)
//   group 0   H columns of u,v,T,S          -> x halos        packed + sent on the COMM stream, in flight during stage 2
//   stage 2   barotropic corrector on the own columns, their y/z layers, w on the own columns, momentum tendencies of
//             the INTERIOR tile columns                                              (main stream)
//   stage 3   [wait for group 0] corrector in the halo columns, w and p' strips, momentum tendencies of the edge tile
//             columns                                                                                (main stream)
//   stage 4   tracer tendencies                                                                     (main stream)
//     beside stage 4, on the COMM stream, the sub-cycle of the NEXT step (its G.U, G.V exist since stage 3):
//   group 3   W = Ns+1+H columns of eta,U,V and of the next G.U,G.V -> wide barotropic halos
//   stage 5   Ns split-explicit substeps on the widened slab into the partner buffers of eta,U,V, filtered state; the slab
//             is wide enough that the x HALO columns of the new eta,U,V come out valid too: nothing is exchanged after it
//   The next stage 0 adopts them.  When a look-ahead is not valid (first step, changed dt, host writes) the same work
//   runs inside the step instead: group 1 (= 3), stage 1 (= 5), on the critical path.  (Group 2 -- H columns of eta,U,V --
//   remains for the initial state.)
//   Folded (tripolar) grid: the work arrays of the sub-cycle are also TALL -- Wy rows beyond the pivot row, the images of the
//   partner rank's rows south of it -- so between stage 1 / 5 (which then only copies the interiors) and the substeps
//   (stage 16 / 56) one more exchange, group 8, carries those rows to the partner: ONE partner exchange per step for the
//   sub-cycle, none inside it.  Group 6 is the partner exchange of the 3-D bundle's rows (and of eta, U, V).
//   2-D decomposition (Partition(Rx, Ry, 1), cfg.ranks_y > 1): every exchange in x is followed by the exchange of whole rows with
//   the southern / northern neighbour, which carry the x halo columns just received (the corners): group 11 / 13 after group
//   1 / 3 and the interior copy (the work arrays are widened in y as in x), group 10 after group 0 and the corrector of the own
//   rows' x halo columns (stage 32: the rows arrive corrected), group 12 after group 2.  The fold partner is the mirrored rank of
//   the top row.  No interior / edge split of the tendencies and no early pressure on such a rank.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <array>
#include <string>
#include <vector>

// (included by gb25_api.hip after the model and its phase implementations)

namespace {

// ---- x-slab exchange pieces --------------------------------------------------------------------------------------
// group 0: H columns of u, v, T, S (all parent rows) -> the neighbour's x halo.
// group 1: W columns of eta, U, V, G.U, G.V -> the neighbour's wide barotropic halo.
// group 2: H columns of eta, U, V -> x halos.
// groups 3 and 4 are groups 1 and 2 of the sub-cycle LOOK-AHEAD: G.U, G.V come from the momentum look-ahead's partner
// buffers, the new eta, U, V live in theirs.
struct Piece {
  real* src;      // array that is packed from (canonical layout)
  real* dst;      // array that is unpacked into
  int src_sx, src_xo, dst_sx, dst_xo;
  long rows;
};
void group_pieces(gb25_model* m, int group, std::vector<Piece>& out, int* ncols) {
  const int H = m->cfg.halo, sx = m->Nx + 2 * H;
  if (group == 20) {
    // closure = CATKE: the TKE tracer after its step inside compute_diffusivities!, and the filtered J^b (kappa of the first
    // halo column is COMPUTED from them)
    *ncols = H;
    if (!m->catke) return;
    for (int id : {GB25_E, GB25_JB}) {
      Field& F = m->f[id];
      out.push_back({F.d, F.d, sx, H, sx, H, (long)F.ny * F.nz});
    }
    return;
  }
  if (group == 0 || group == 2 || group == 4) {
    *ncols = H;
    if (group == 0) {
      for (int id : {GB25_U, GB25_V, GB25_T, GB25_S}) {
        Field& F = m->f[id];
        out.push_back({F.d, F.d, sx, H, sx, H, (long)F.ny * F.nz});
      }
      // the column integrals of u, v (the corrector's): the receiver corrects its halo columns cell by cell with them
      for (int q = 0; q < 2; q++) out.push_back({m->colsum[q].d, m->colsum[q].d, sx, H, sx, H, (long)m->colsum[q].ny});
      // ... and their sums over the momentum kernel's chunks of levels (w on the fly: the chunk bases of w next to the x halos)
      if (slab_wfly_ok(m)) {
        const int kch = mom_kchunks(m);
        const long plane2 = (long)m->g.sx * m->g.sy_v;
        for (int q = 2; q < 4; q++) {
          real* P = m->uv_partials + (long)q * kch * plane2;
          out.push_back({P, P, sx, H, sx, H, (long)kch * m->g.sy_v});
        }
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
    const size_t up = (size_t)m->Wys * wsx;   // (2-D decomposition: the work arrays start Wys rows below the canonical ones)
    for (int q = 0; q < 3; q++) {
      Field& F = m->f[GB25_ETA + q];
      out.push_back({F.d, m->wide[0][q].d + up, sx, H, wsx, m->W, (long)F.ny});
    }
    for (int q = 0; q < 2; q++) {
      Field& F = group == 1 ? m->f[GB25_GN_BT_U + q] : m->ahead_G[q];
      out.push_back({F.d, m->wideG[q].d + up, sx, H, wsx, m->W, (long)F.ny});
    }
  }
}
int64_t halo_buffer_elems(gb25_model* m, int group) {
  std::vector<Piece> ps;
  int nc = 0;
  group_pieces(m, group, ps, &nc);
  int64_t t = 0;
  for (auto& p : ps) t += p.rows * nc;
  return t > 0 ? t : 1;
}
// Both sides of a group in ONE launch (a group is up to ten small strips); buf[side] = that side's contiguous buffer.
gb25_status pack_unpack(gb25_model* m, int group, real* const buf[2], bool pack) {
  std::vector<Piece> ps;
  int nc = 0;
  group_pieces(m, group, ps, &nc);
  ColumnPieces P{};
  P.ncols = nc;
  long max_n = 0;
  for (int side = 0; side < 2; side++) {
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

// ---- y halos of a 2-D (x, y) decomposition: whole rows (every parent column) to the southern / northern neighbour ------------
// group 10: H rows of u, v, T, S (interior levels) -- after group 0 and the barotropic corrector of the own rows (x halo columns
//           included: the corners), so that the rows arrive corrected, as the rows beyond a fold do (group 6);
// group 11: W rows of the sub-cycle's work arrays eta, U, V, G.U, G.V (all widened columns) -- after group 1 and the interior copy;
// group 12: H rows of eta, U, V (initial state) -- after group 2.      13, 14: the same for the sub-cycle look-ahead.
// side 0: southern edge / halo, side 1: northern.  Rows [0, n) / [Ny - n, Ny) are packed, [-n, 0) / [Ny, Ny + n) unpacked.
inline bool y_neighbour(const gb25_model* m, int side) { return side == 0 ? m->ys_open : m->yn_open; }
void row_pieces(gb25_model* m, int group, int side, bool pack, real* buf, RowPieces& P) {
  const Grid& g = m->g;
  const int H = g.H;
  P = RowPieces{};
  size_t off = 0;
  auto add = [&](real* arr, int sx, int yoff, long pl, int k0, int nz, int nrows) {
    const int f = P.n++;
    const int j = pack ? (side == 0 ? 0 : g.Ny - nrows) : (side == 0 ? -nrows : g.Ny);
    P.arr[f] = arr; P.buf[f] = buf + off; P.sx[f] = sx; P.r0[f] = j + yoff; P.pl[f] = pl; P.k0[f] = k0; P.nz[f] = nz;
    P.nrows = nrows;
    off += (size_t)nz * nrows * sx;
  };
  if (group == 21) {   // (CATKE: the TKE tracer after its step, and J^b for kappa in the first halo row)
    if (m->catke) {
      Field& F = m->f[GB25_E];
      add(F.d, g.sx, H, (long)g.sx * F.ny, H, g.Nz, H);
      add(m->f[GB25_JB].d, g.sx, H, 0, 0, 1, H);
    }
  } else if (group == 10) {
    for (int id : {GB25_U, GB25_V, GB25_T, GB25_S}) {
      Field& F = m->f[id];
      add(F.d, g.sx, H, (long)g.sx * F.ny, H, g.Nz, H);
    }
    if (slab_lazy_ok(m)) {
      // the corrector inside its consumers: the column integrals of u, v of the rows (the receiver's du, dv there) and, for w on
      // the fly, their sums over the chunks of levels
      for (int q = 0; q < 2; q++) add(m->colsum[q].d, g.sx, H, 0, 0, 1, H);
      if (slab_wfly_ok(m)) {
        const int kch = mom_kchunks(m);
        const long plane2 = (long)g.sx * g.sy_v;
        for (int q = 2; q < 4; q++) add(m->uv_partials + (long)q * kch * plane2, g.sx, H, plane2, 0, kch, H);
      }
    }
  } else if (group == 11 || group == 13) {
    const int wsx = g.Nx + 2 * m->W;
    for (int q = 0; q < 3; q++) add(m->wide[0][q].d, wsx, H + m->Wys, 0, 0, 1, m->W);
    for (int q = 0; q < 2; q++) add(m->wideG[q].d, wsx, H + m->Wys, 0, 0, 1, m->W);
  } else {
    for (int q = 0; q < 3; q++) add((group == 12 ? m->f[GB25_ETA + q] : m->ahead_eta[q]).d, g.sx, H, 0, 0, 1, H);
  }
}
int64_t row_buffer_elems(gb25_model* m, int group) {
  if (m->Ry < 2) return 1;
  RowPieces P;
  row_pieces(m, group, 0, true, nullptr, P);
  int64_t t = 0;
  for (int f = 0; f < P.n; f++) t += (int64_t)P.nz[f] * P.nrows * P.sx[f];
  return t > 0 ? t : 1;
}
gb25_status move_rows(gb25_model* m, int group, real* const buf[2], bool pack) {
  for (int side = 0; side < 2; side++) {
    if (!y_neighbour(m, side)) continue;
    RowPieces P;
    row_pieces(m, group, side, pack, buf[side], P);
    if (P.n == 0) continue;
    int nzmax = 1, sxmax = 0;
    for (int f = 0; f < P.n; f++) { nzmax = std::max(nzmax, P.nz[f]); sxmax = std::max(sxmax, P.sx[f]); }
    const dim3 gr((unsigned)(((long)sxmax * P.nrows + 255) / 256), nzmax, P.n);
    if (pack) hipLaunchKernelGGL(k_move_rows<true>, gr, dim3(256), 0, m->stream, P);
    else hipLaunchKernelGGL(k_move_rows<false>, gr, dim3(256), 0, m->stream, P);
  }
  LAUNCHCHK();
  return GB25_OK;
}

// ---- zipper fold of a decomposed tripolar grid: the partner rank P-1-r holds the cells beyond this slab's fold line -----
// buffer set 3: the H rows south of the pivot row of u, v, T, S (CATKE: e, J^b too) and eta, U, V (all parent
// columns, interior levels);
// buffer set 4: the Wy (+1) rows south of the pivot row of the sub-cycle's work arrays eta, U, V, G.U, G.V (TallRows, kernels.hpp)
FoldFields fold_fields(gb25_model* m, bool closure = false) {   // closure: group 22 -- e and J^b after the e step
  FoldFields F{};
  const Grid& g = m->g;
  long off = 0;
  auto add = [&](real* p, int is_v, int xf, int neg, int nz) {
    const int f = F.n++;
    F.p[f] = p; F.is_v[f] = is_v; F.xf[f] = xf; F.neg[f] = neg; F.nz[f] = nz; F.off[f] = off;
    off += (long)nz * g.H * g.sx;
  };
  if (closure) {
    if (m->catke) {
      add(m->f[GB25_E].d, 0, 0, 0, g.Nz);
      add(m->f[GB25_JB].d, 0, 0, 0, 1);
    }
    return F;
  }
  add(m->f[GB25_U].d, 0, 1, 1, g.Nz);
  add(m->f[GB25_V].d, 1, 0, 1, g.Nz);
  add(m->f[GB25_T].d, 0, 0, 0, g.Nz);
  add(m->f[GB25_S].d, 0, 0, 0, g.Nz);
  add(m->f[GB25_ETA].d, 0, 0, 0, 1);
  add(m->f[GB25_BT_U].d, 0, 1, 1, 1);
  add(m->f[GB25_BT_V].d, 1, 0, 1, 1);
  return F;
}
int fold_levels(const FoldFields& F, bool with_layers) {   // blockIdx.z extent of k_fold_pack / k_fold_unpack
  int t = 0;
  for (int f = 0; f < F.n; f++) t += F.nz[f] == 1 ? 1 : F.nz[f] + (with_layers ? 2 : 0);
  return t;
}
int64_t fold_buffer_elems(gb25_model* m, int b) {
  const Grid& g = m->g;
  if (!g.cv.north_fold) return 1;
  if (b == 10) return std::max<int64_t>(1, (int64_t)g.H * g.sx * fold_levels(fold_fields(m, true), false));
  return b == 3 ? (int64_t)g.H * g.sx * fold_levels(fold_fields(m), false) : tall_buffer_elems(m);
}
gb25_status fold_pack(gb25_model* m, real* buf, bool closure = false) {
  const Grid& g = m->g;
  FoldFields F = fold_fields(m, closure);
  if (F.n == 0) return GB25_OK;
  hipLaunchKernelGGL(k_fold_pack, dim3((g.sx + 255) / 256, g.H, fold_levels(F, false)), dim3(256), 0, m->stream, g, F, buf);
  LAUNCHCHK();
  return GB25_OK;
}
gb25_status fold_unpack(gb25_model* m, const real* buf, bool closure = false) {
  const Grid& g = m->g;
  FoldFields F = fold_fields(m, closure);
  if (F.n == 0) return GB25_OK;
  hipLaunchKernelGGL(k_fold_unpack, dim3((g.sx + 255) / 256, g.H, fold_levels(F, true)), dim3(256), 0, m->stream, g, F, buf,
                     m->rx * m->Nx, m->cfg.Nx);
  LAUNCHCHK();
  return GB25_OK;
}

// the pressure strips next to the x halos on the exchange stream, right behind the unpacked bundle (plain x slabs, no closure
// whose fields and fills sit in between)
inline bool strips_on_comm(const gb25_model* m) {
  return m->early_strips && m->two_streams && m->pressure_bits == 64 && m->Ry == 1 && !m->catke;
}

// closure = CATKE on a rank of a decomposition.  compute_diffusivities! steps e on the own columns (time_step_catke_equation!)
// and filters J^b there; the diffusivities of the first halo column / row and the advection of e then need the NEW e and J^b of
// the neighbours: one more exchange per update_state! (groups 20: x columns, 21: rows of a 2-D decomposition, 22: the rows
// beyond a zipper fold), between these two halves.
gb25_status catke_step_local(gb25_model* m) {
  gb25_status s;
  if ((s = catke_tke_step_impl(m))) return s;
  return fill_halos_impl(m, false, false, 1, 4);   // y/z layers of the new e on the own columns: the packed columns travel complete
}
gb25_status catke_finish_local(gb25_model* m) {
  gb25_status s;
  if ((s = fill_halos_impl(m, false, true, 1, 4))) return s;   // y/z layers of e over the extended range (the halo rows' bottom / top layers)
  if ((s = catke_diffusivities_finish_impl(m))) return s;
  return catke_tendency_impl(m);
}

// ---- the stages of one slab's time step (see the header of this file) -------------------------------------------------
gb25_status slab_stage(gb25_model* m, int stage, int euler) {
  const Grid& g = m->g;
  gb25_status s;
  const double dt = m->last_dt;
  const real chi = euler ? -real(0.5) : (real)m->cfg.chi;
  const bool split = tendencies_split(m);
  // own columns' pressure early, on the side stream.  (A folded slab too: the pressure of a cell and its differences to the west
  // and south never look north -- what arrives last, the rows beyond the fold, is no input of theirs; the pressure of halo cells
  // is not stored inside a composite step.)
  // (a rank of a 2-D decomposition computes its pressure in one pass once all halos are in: the early pass plus a one-row strip
  // for the y difference of row 0 measured slower on the one-rank proxy, 0.566 against 0.537 ms -- profiles/r03_tuning_log.md)
  const bool p_early = m->two_streams && m->pressure_bits == 64 && m->Ry == 1;
  if (stage == 0) {
    // AB2 update of u,v,T,S + barotropic forcing, then the y/z boundary layers of the 3-D bundle so that its packed
    // x columns (group 0) can travel WHILE the own columns are corrected
    const bool uv_adopted = m->ahead_uv_valid && (real)dt == m->ahead_uv_dt && chi == m->ahead_uv_chi;
    m->baro_adopted = uv_adopted && m->ahead_baro_valid;
    m->ahead_baro_valid = false;
    // The corrector inside its consumers (as on a single domain, time_step_impl): when everything this step needs was made
    // ahead of time, no sweep over u and v -- 2-D kernels leave du, dv (own columns in stage 2, halo columns in stage 3 from the
    // column integrals the bundle carries) and the kernels that read u, v add them.  Memory holds the uncorrected velocities
    // until the composite call returns (gb25_loop).
    m->step_lazy = uv_adopted && m->baro_adopted && slab_lazy_ok(m);
    m->lazy_head_done = false;
    // ... and with it w on the fly: no k_compute_w launch, the tendency kernels carry w up their chunks of levels from 2-D bases
    m->w_fly_now = m->step_lazy && slab_wfly_ok(m);
    if (!m->step_lazy && (s = materialize_uv(m))) return s;   // (the sweeps below expect corrected velocities)
    if ((s = ab2_local_impl(m, (real)dt, chi))) return s;
    if (m->baro_adopted) {
      // the sub-cycle of this step, its wide-halo exchange and the exchange of the new eta, U, V columns all ran
      // beside the last tracer kernel (stage 5): adopt the results, stage 1 and groups 1, 2 are skipped
      for (int q = 0; q < 3; q++) {
        std::swap(m->f[GB25_ETA + q].d, m->ahead_eta[q].d);
        std::swap(m->f[GB25_ETA_BAR + q].d, m->ahead_bar[q].d);
      }
      std::swap(m->bars, m->bars_ahead);
      m->time += dt;
      m->iteration += 1;
    }
    if ((s = fill_halos_impl(m, false, false, 1))) return s;
    if (p_early) {
      // T, S of the slab's own columns are final from here on: their pressure (fp64-bound) runs on the side stream
      // beside the exchanges and the sub-cycle; the strips next to the x halos follow in stage 3.  The first x
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
  } else if (stage == 1 || stage == 5 || stage == 16 || stage == 56) {
    // stage 1: group 1 has been unpacked into the wide halos: copy the interiors, sub-cycle, publish.
    // stage 5: the same for the NEXT step (look-ahead): group 3 has been unpacked, G.U, G.V come from the momentum
    //          look-ahead, the results go to the partner buffers of eta, U, V and of the filtered state.
    // Folded grid: stage 1 / 5 end after the interior copy (the image rows beyond the pivot row travel next: group 8),
    // stage 16 / 56 do the rest.
    const bool ahead = stage == 5 || stage == 56, second_half = stage == 16 || stage == 56;
    if (ahead && !m->ahead_uv_valid) return fail(m, GB25_ERR_STATE, "stage 5 without a velocity look-ahead");
    if (!ahead && m->baro_adopted) return fail(m, GB25_ERR_STATE, "stage 1 after stage 0 adopted the sub-cycle");
    std::vector<Piece> ps;
    int nc = 0;
    group_pieces(m, ahead ? 3 : 1, ps, &nc);
    const bool two_halves = g.cv.north_fold || m->Ry > 1;   // (more rows of the work arrays travel between the copy and the substeps)
    InteriorCopies C{};
    if (!second_half) {
      int rmax = 0;
      for (auto& p : ps) {
        const int q = C.n++;
        C.dst[q] = p.dst; C.dsx[q] = p.dst_sx; C.dxo[q] = p.dst_xo;
        C.src[q] = p.src; C.ssx[q] = p.src_sx; C.sxo[q] = p.src_xo; C.rows[q] = (int)p.rows;
        rmax = std::max(rmax, (int)p.rows);
      }
      // (a plain slab hands the copy to the sub-cycle: its one-launch kernel reads the own columns where they are)
      if (two_halves)
        hipLaunchKernelGGL(k_copy_interior_columns, dim3((g.Nx + 255) / 256, rmax, C.n), dim3(256), 0, m->stream, C, g.Nx);
    }
    LAUNCHCHK();
    if (two_halves && !second_half) return GB25_OK;
    const InteriorCopies* own = two_halves ? nullptr : &C;
    bool layers_done = false;
    if (ahead) {
      if ((s = barotropic_impl(m, m->ahead_uv_dt, true, own, &layers_done))) return s;
      Halo2 h2{};
      for (int q = 0; q < 3; q++) { h2.p[q] = m->ahead_eta[q].d; h2.is_v[q] = q == 2; }
      h2.n = 3;
      // y layer, x halo columns included: the widened sub-cycle computed those like the neighbour did (no group 4)
      if (!layers_done && (s = fill_halos_impl(m, false, true, 2, 3, nullptr, true, &h2))) return s;
      m->ahead_baro_valid = true;
      return GB25_OK;
    }
    if ((s = barotropic_impl(m, (real)dt, false, own, &layers_done))) return s;
    m->time += dt;
    m->iteration += 1;
    // y layer of the new eta, U, V, x halo columns included (computed by the widened sub-cycle: no group 2)
    return layers_done ? GB25_OK : fill_halos_impl(m, false, true, 2);
  } else if (stage == 2 || stage == 20) {
    // Everything that needs nothing from the neighbours runs while the exchanges are in flight: the barotropic
    // corrector on the slab's own columns and, when the tendency kernels are split (a12), the y/z layers and w of the
    // own columns and the momentum tendencies of the interior tile columns.
    // (stage 20: the head of a lazy step -- du, dv and the chunk bases of w -- ahead of the wait for the packed bundle)
    if (stage == 2 && m->lazy_head_done) {
      // (stage 20 did the corrector's part)
    } else if (m->step_lazy) {
      m->lazy_head_done = stage == 20;
      if (!m->colsum_valid) return fail(m, GB25_ERR_STATE, "internal: a lazy step without the column integrals of u, v");
      dim3 b(64, 4);
      Timed t(m, GB25_K_CORRECTOR);
      // (a rank of a 2-D decomposition has no interior pass: du, dv and the chunk bases of w over its whole extended range at
      // once, in stage 3, when every halo is in)
      if (m->Ry == 1)
        hipLaunchKernelGGL(k_corrector_2d<false>, grid2(g.Nx, g.Ny + 1, b), b, 0, m->stream, g, m->f[GB25_BT_U].d, m->f[GB25_BT_V].d,
                           m->colsum[0].d, m->colsum[1].d, m->f[GB25_U_BAR].d, m->f[GB25_V_BAR].d, m->corr[0].d, m->corr[1].d,
                           0, g.Nx, INT_MAX, 0, 0, g.Ny + 1);
      LAUNCHCHK();
      m->uv_lazy = true;
      if (m->Ry == 1) m->colsum_valid = false;
      if (m->w_fly_now && m->Ry == 1) {
        // chunk bases of w on the columns [0, Nx - 2] (their u faces are own columns), before the interior momentum pass
        // overwrites the chunk sums they are made from
        hipLaunchKernelGGL((k_w_bases<false, false>), grid2(g.Nx - 1, g.Ny + 4, b), b, 0, m->stream, g, m->uv_partials, mom_kchunks(m), g.sx * g.sy_v,
                           LazyCorr{m->corr[0].d, m->corr[1].d, nullptr, 0}, m->wbase, 0, g.Nx - 1, INT_MAX, 0);
        LAUNCHCHK();
        m->w_stale = true;
      }
      for (int q = 0; q < 4; q++) std::swap(m->f[GB25_GN_U + q].d, m->f[GB25_GM_U + q].d);   // cache_previous_tendencies!
      m->ahead_valid = m->ahead_uv_valid = m->ahead_baro_valid = false;
    } else if ((s = corrector_impl(m, true, 1))) {
      return s;
    }
    if (stage == 20 || !split) return GB25_OK;
    // y/z layers of the corrected u, v, own columns (lazy: the layers of the uncorrected ones are in place since stage 0)
    if (!m->step_lazy && (s = fill_halos_impl(m, false, false, 1, 1))) return s;
    if (!m->w_fly_now && (s = compute_w_impl(m, 1))) return s;
    HIPCHK(hipStreamWaitEvent(m->stream, m->ev_join, 0));       // the own columns' pressure differences (side stream)
    return momentum_impl(m, 1);
  } else if (stage == 33) {
    // (issued on the exchange stream behind the unpack of group 0) the two pressure strips: T, S of the halo columns are in; the
    // interior pass of stage 0 (side stream: event ev_join) must be over -- the west strip redoes its column 0
    HIPCHK(hipStreamWaitEvent(m->stream, m->ev_join, 0));
    if ((s = compute_p_impl(m, -g.H + 1, 0, g.Nx, g.Nx + g.H - 2, true))) return s;
    HIPCHK(hipEventRecord(m->ev_strips, m->stream));
    m->strips_issued = true;
    return GB25_OK;
  } else if (stage == 32) {
    // 2-D decomposition: group 0 has been unpacked; the corrector on the x-halo columns of the own rows, so that the rows that
    // leave for the southern / northern neighbour next (group 10) are corrected over their whole width
    // (a lazy step: the rows leave uncorrected, like the columns; the receiver makes du, dv of its halo rows itself)
    return m->step_lazy ? GB25_OK : corrector_impl(m, true, 2);
  } else if (stage == 3 || stage == 30 || stage == 31) {
    // groups 2 and 0 have been unpacked: corrector on the x-halo columns, then update_state without any
    // further exchange (y/z layers re-filled over the extended x range; w and p recomputed in the halos).
    // Folded grid: stage 30 = up to the y/z layers, then the rows beyond the fold arrive from the partner, stage 31 = the rest.
    // (a closure's fields travel in the bundle as well and its fills follow: the strips keep their old place behind them)
    const bool strips_first = p_early && !m->catke;
    const bool strips_done = m->strips_issued;   // (stage 33 ran them on the exchange stream)
    if (stage != 30) m->strips_issued = false;
    auto pressure_strips = [&]() -> gb25_status {
      hipStream_t main = m->stream;
      HIPCHK(hipEventRecord(m->ev_fork, main));
      HIPCHK(hipStreamWaitEvent(m->side_stream, m->ev_fork, 0));
      m->stream = m->side_stream;
      gb25_status r = compute_p_impl(m, -g.H + 1, 0, g.Nx, g.Nx + g.H - 2, true);   // west strip (redoes column 0) + east strip
      m->stream = main;
      if (r) return r;
      HIPCHK(hipEventRecord(m->ev_join, m->side_stream));
      return GB25_OK;
    };
    if (stage != 31) {
      if (strips_first && !strips_done) {
        // the two pressure strips next to the x halos start as soon as the bundle is unpacked: T, S of the halo columns arrived
        // complete (their y/z layers were filled by their owner before it packed them), and the side stream runs them behind the
        // interior pass of stage 0 -- beside the corrector, the fills and w of the edge strips instead of after them
        if ((s = pressure_strips())) return s;
      }
      if (m->step_lazy) {
        // du, dv of the x halo columns: the neighbours' column integrals came with the bundle, the new U, V of those columns
        // from the widened sub-cycle; their y/z layers of u, v arrived filled -- nothing else to do
        if (!m->halo_colsum_valid) return fail(m, GB25_ERR_STATE, "internal: a lazy step without the neighbours' column integrals");
        if (m->Ry > 1) {
          // 2-D decomposition: everything at once -- own cells, halo columns, halo rows of the open sides (corners included)
          dim3 b(64, 4);
          // rows: from the southern halo rows (or row 0) to the last northern halo row of the cell-shaped arrays (or the wall face)
          const int hs = m->ys_open ? g.H : 0, nj = hs + g.Ny + (m->yn_open ? g.H : 1);
          hipLaunchKernelGGL(k_corrector_2d<false>, grid2(g.Nx + 2 * g.H, nj, b), b, 0, m->stream, g, m->f[GB25_BT_U].d,
                             m->f[GB25_BT_V].d, m->colsum[0].d, m->colsum[1].d, m->f[GB25_U_BAR].d, m->f[GB25_V_BAR].d, m->corr[0].d,
                             m->corr[1].d, -g.H, g.Nx + 2 * g.H, INT_MAX, 0, -hs, nj);
          m->colsum_valid = false;
          if (m->w_fly_now) {
            hipLaunchKernelGGL((k_w_bases<false, false>), grid2(g.Nx + 4, g.Ny + 4, b), b, 0, m->stream, g, m->uv_partials, mom_kchunks(m), g.sx * g.sy_v,
                               LazyCorr{m->corr[0].d, m->corr[1].d, nullptr, 0}, m->wbase, -2, g.Nx + 4, INT_MAX, 0);
            m->w_stale = true;
          }
          LAUNCHCHK();   // (the bottom / top layers of the halo rows, which arrived with their interior levels: the fill below)
        } else {
          dim3 b(16, 16);
          hipLaunchKernelGGL(k_corrector_2d<false>, grid2(2 * g.H, g.Ny + 1, b), b, 0, m->stream, g, m->f[GB25_BT_U].d, m->f[GB25_BT_V].d,
                             m->colsum[0].d, m->colsum[1].d, m->f[GB25_U_BAR].d, m->f[GB25_V_BAR].d, m->corr[0].d, m->corr[1].d,
                             -g.H, 2 * g.H, 0, g.Nx, 0, g.Ny + 1);
          if (m->w_fly_now)   // the chunk bases of w on the columns -2, -1 and Nx - 1, Nx, Nx + 1 (the w tiles reach two columns out)
            hipLaunchKernelGGL((k_w_bases<false, false>), grid2(5, g.Ny + 4, b), b, 0, m->stream, g, m->uv_partials, mom_kchunks(m), g.sx * g.sy_v,
                               LazyCorr{m->corr[0].d, m->corr[1].d, nullptr, 0}, m->wbase, -2, 5, 0, g.Nx - 1);
          LAUNCHCHK();
        }
      } else if (m->Ry == 1 && (s = corrector_impl(m, true, 2))) {   // (2-D decomposition: done in stage 32)
        return s;
      }
      // (with the early strips T and S are left alone here: their layers are in place, own columns since stage 0)
      if (p_early && !strips_first) HIPCHK(hipStreamWaitEvent(m->stream, m->ev_join, 0));   // (the interior pass reads T, S)
      if (!(m->step_lazy && strips_first) && (s = fill_halos_impl(m, false, true, 3, strips_first ? 1 : 3))) return s;
      if (stage == 30) return GB25_OK;
    }
    if (p_early && !strips_first && (s = pressure_strips())) return s;   // (beside w)
    if (!m->w_fly_now && (s = compute_w_impl(m, split ? 2 : 0))) return s;
    if (strips_done) {
      HIPCHK(hipStreamWaitEvent(m->stream, m->ev_strips, 0));
    } else if (p_early) {
      HIPCHK(hipStreamWaitEvent(m->stream, m->ev_join, 0));
    } else {
      if ((s = compute_p_impl(m))) return s;
    }
    return momentum_impl(m, split ? 2 : 0);
  } else if (stage == 4) {
    // the tracer tendencies; the look-ahead of the next sub-cycle (groups 3, 4 and stage 5) runs beside them
    if ((s = tracers_impl(m))) return s;
    if (m->catke) return catke_step_local(m);   // (the rest after the halos of the new e and J^b: stage 41)
    return atmosphere_ocean_fluxes_impl(m);
  } else if (stage == 41) {
    // closure = CATKE: groups 20 - 22 have been unpacked -- the diffusivities and the slow tendency of e, then the fluxes
    if ((s = catke_finish_local(m))) return s;
    return atmosphere_ocean_fluxes_impl(m);
  }
  return fail(m, GB25_ERR_INVALID_ARGUMENT, "unknown stage %d", stage);
}

gb25_status update_state_local_impl(gb25_model* m) {   // update_state! without the x-halo fill
  gb25_status s;
  if ((s = mask_impl(m))) return s;   // (own columns; the halo columns arrived masked by their owners)
  if ((s = fill_halos_impl(m, false, m->slab))) return s;
  if ((s = compute_w_impl(m))) return s;
  if ((s = compute_p_impl(m))) return s;
  if ((s = momentum_impl(m))) return s;
  if ((s = tracers_impl(m))) return s;
  return m->catke ? catke_step_local(m) : GB25_OK;   // (the sequencer goes on with groups 20 - 22 and catke_finish_local)
}

// ---- sequencing ------------------------------------------------------------------------------------------------------
// Everything a step of the slabs driven by this process does, in issue order.  Streams: `main` (on_comm = false) and
// `comm`; record(slot, on_comm) marks everything issued so far on a stream, wait(slot, comm_waits) makes the other (or
// the same) stream wait for that mark.  No host synchronisation anywhere.
// Streams by number: 0 = main, 1 = comm (the exchanges: packs, transfers, unpacks), 2 = sub (the substeps of the sub-cycle
// look-ahead: on a stream of their own so that the next step's bundle -- posted on the comm stream right behind stage 0 -- does
// not queue behind five sub-cycle launches it has nothing to do with; 95 us of a 180-column rank's 640).
struct StepOps {
  virtual ~StepOps() {}
  virtual int n() const = 0;
  virtual gb25_status stage(int s, int stage, int euler, int on) = 0;
  virtual gb25_status pack(int s, int group, int on) = 0;
  virtual gb25_status unpack(int s, int group, int on) = 0;
  virtual gb25_status exchange(int group, int on) = 0;   // every slab's packs -> its neighbours' receive buffers
  virtual gb25_status local(int s, int what) = 0;              // 0: initialize!, 1: y/z halo layers, 2: update_state! (local)
  virtual bool velocities_ready(int s) = 0;
  virtual bool subcycle_adopted(int s) = 0;
  virtual bool coupled() { return false; }   // a prescribed atmosphere is set (data-free forcing)
  virtual bool catke() { return false; }     // closure = CATKE: the halos of the new e, J^b travel inside update_state! (groups 20 - 22)
  virtual bool folded() = 0;          // zipper fold: exchanges with the partner rank (buffer sets 3 and 4; groups 6 and 8)
  virtual bool lazy() { return false; }     // this step keeps the corrector inside its consumers (known after stage 0)
  // the bundle is unpacked on the exchange stream right behind its transfer (halo columns only: nothing the main stream touches
  // before it waits for event 3); on plain x slabs the two pressure strips next to the x halos follow it there (stage 33) --
  // beside the interior momentum pass instead of in front of the edge pass
  virtual bool early_unpack() { return false; }
  virtual bool early_strips() { return false; }
  virtual bool mesh_y() { return false; }   // 2-D decomposition: y halos from the southern / northern neighbour (groups 10 - 14)
  virtual gb25_status record(int slot, int on) = 0;
  virtual gb25_status wait(int slot, int waiter) = 0;
};
#define SEQ(call)            \
  do {                       \
    gb25_status st_ = (call); \
    if (st_) return st_;     \
  } while (0)
#define EACH(expr)                         \
  for (int s = 0; s < n; s++) SEQ(expr)

// closure = CATKE: e was stepped and J^b filtered on the own columns; their halos, then the second half of compute_diffusivities!
// and the slow tendency of e (`finish`: a stage of the step, or local operation 7 inside first_time_step!)
gb25_status sequence_catke_halos(StepOps& o, int n) {
  EACH(o.pack(s, 20, false));
  SEQ(o.exchange(20, false));
  EACH(o.unpack(s, 20, false));
  if (o.mesh_y()) {       // whole rows, with the x halo columns just received (the corners)
    EACH(o.pack(s, 21, false));
    SEQ(o.exchange(21, false));
    EACH(o.unpack(s, 21, false));
  }
  if (o.folded()) {
    EACH(o.pack(s, 22, false));
    SEQ(o.exchange(22, false));
    EACH(o.unpack(s, 22, false));
  }
  return GB25_OK;
}
gb25_status sequence_time_step(StepOps& o, int euler, bool& lookahead_in_flight) {
  const int n = o.n();
  // The previous step may have left the look-ahead chain (group 3, stage 5) running on the second stream.  Stage 0 touches
  // nothing of it on the device (the adoption of eta, U, V is a pointer exchange on the host; its kernels update and fill
  // u, v, T, S and start the pressure), so it does not wait: on a narrow slab the chain (an exchange + the sub-cycle,
  // ~230 us) outlasts the tracer kernel it runs beside (~120 us), and stage 0 fills part of the difference.
  const bool in_flight = lookahead_in_flight;
  lookahead_in_flight = false;
  EACH(o.stage(s, 0, euler, false));
  bool adopted = true;         // the sub-cycle of this step is already done
  for (int s = 0; s < n; s++) adopted = adopted && o.subcycle_adopted(s);
  if (in_flight && !adopted)   // not adopted after all: it must be over before its buffers and work arrays are reused
    SEQ(o.wait(4, false));    // (event 4: recorded behind the chain when it was issued)
  SEQ(o.record(0, false));     // everything stage 0 wrote
  if (!adopted) {
    // The small barotropic exchange is on the critical path and is posted FIRST: the point-to-point transfers of one
    // communicator run in posting order, so the 6 MB bundle must not be queued ahead of it.
    EACH(o.pack(s, 1, false));
    SEQ(o.exchange(1, false));
  }
  SEQ(o.wait(0, true));        // the 3-D bundle leaves on the second stream ...
  EACH(o.pack(s, 0, true));
  SEQ(o.record(1, true));      // (packed)
  SEQ(o.exchange(0, true));
  const bool early = o.early_unpack();
  if (early)
    for (int s = 0; s < n; s++) {
      SEQ(o.unpack(s, 0, true));
      if (o.early_strips()) SEQ(o.stage(s, 33, euler, true));
    }
  // 2-D decomposition, a step that keeps the corrector inside its consumers: the rows leave as they are (stage 32 has nothing to
  // do), so their exchange follows the bundle on the exchange stream instead of waiting for the main stream to get there (one
  // cross-stream hop less: 0.538 -> 0.504 ms per step of a 360 x 360 rank, tools/slab_selfring.py --mesh 4 2)
  const bool rows_early = early && o.mesh_y() && o.lazy();
  if (rows_early) {
    EACH(o.pack(s, 10, true));
    SEQ(o.exchange(10, true));
    EACH(o.unpack(s, 10, true));
  }
  if (!adopted && (o.folded() || o.mesh_y())) {
    // zipper fold: the work arrays are tall as well as wide.  Once every slab has its wide halo columns, the rows south of
    // the pivot row go to the partner rank P-1-r (group 8) and become its image rows beyond the pivot row; then the substeps
    // run with no further exchange.  2-D decomposition: likewise the W rows next to an open side go to the southern /
    // northern neighbour (group 11), with the wide halo columns just received: the corners
    for (int s = 0; s < n; s++) {
      SEQ(o.unpack(s, 1, false));
      SEQ(o.stage(s, 1, euler, false));      // (the interior copy only)
      if (o.mesh_y()) SEQ(o.pack(s, 11, false));
      if (o.folded()) SEQ(o.pack(s, 8, false));
    }
    if (o.mesh_y()) SEQ(o.exchange(11, false));
    if (o.folded()) SEQ(o.exchange(8, false));
    for (int s = 0; s < n; s++) {
      if (o.mesh_y()) SEQ(o.unpack(s, 11, false));
      if (o.folded()) SEQ(o.unpack(s, 8, false));
      SEQ(o.stage(s, 16, euler, false));
    }
  } else if (!adopted) {       // ... and is in flight while the sub-cycle runs here (it leaves the x halo columns of the
    for (int s = 0; s < n; s++) {   // new eta, U, V behind as well: the slab is widened by Ns + 1 + H columns, no group 2)
      SEQ(o.unpack(s, 1, false));
      SEQ(o.stage(s, 1, euler, false));
    }
  }
  if (o.lazy()) {
    // the corrector inside its consumers writes nothing the bundle is packed from: du, dv and the chunk bases of w of the own
    // columns (stage 20) need the adopted sub-cycle only; the interior momentum pass, which overwrites chunk sums the bundle
    // carries, waits for the pack
    if (in_flight && adopted) SEQ(o.wait(4, false));
    EACH(o.stage(s, 20, euler, false));
    SEQ(o.wait(1, false));
  } else {
    SEQ(o.wait(1, false));       // the corrector rewrites the columns the bundle was packed from
    if (in_flight && adopted)    // ... and reads the adopted sub-cycle: the look-ahead chain has finished (event 4 sits
      SEQ(o.wait(4, false));     // behind the chain, ahead of this step's bundle on the same stream)
  }
  EACH(o.stage(s, 2, euler, false));   // own columns + interior tendencies, while the exchanges are in flight
  SEQ(o.record(3, true));
  SEQ(o.wait(3, false));       // the halo columns have arrived
  if (o.mesh_y() && !rows_early) {
    // 2-D decomposition: the H rows next to an open side, with the x halo columns just received (the corners)
    for (int s = 0; s < n; s++) {
      if (!early) SEQ(o.unpack(s, 0, false));
      SEQ(o.stage(s, 32, euler, false));   // (the corrector on the x halo columns: the rows leave corrected)
      SEQ(o.pack(s, 10, false));
    }
    SEQ(o.exchange(10, false));
    EACH(o.unpack(s, 10, false));
  }
  if (o.folded()) {
    // the rows beyond the fold come from the partner once every slab has its x halos and y/z layers (the partner sends
    // its halo columns too: the corners), then the rest of update_state!
    for (int s = 0; s < n; s++) {
      if (!o.mesh_y() && !early) SEQ(o.unpack(s, 0, false));
      SEQ(o.stage(s, 30, euler, false));
      SEQ(o.pack(s, 6, false));
    }
    SEQ(o.exchange(6, false));
    for (int s = 0; s < n; s++) {
      SEQ(o.unpack(s, 6, false));
      SEQ(o.stage(s, 31, euler, false));
    }
  } else {
    for (int s = 0; s < n; s++) {
      if (!o.mesh_y() && !early) SEQ(o.unpack(s, 0, false));
      SEQ(o.stage(s, 3, euler, false));
    }
  }
  // the next step's G.U, G.V exist now: its wide-halo exchange, sub-cycle and eta,U,V exchange run on the second stream
  // beside the tracer tendencies
  bool ready = true;
  for (int s = 0; s < n; s++) ready = ready && o.velocities_ready(s);
  if (ready) {
    SEQ(o.record(2, false));
    SEQ(o.wait(2, true));
    EACH(o.pack(s, 3, true));
    SEQ(o.exchange(3, true));
    if (o.folded() || o.mesh_y()) {
      for (int s = 0; s < n; s++) {
        SEQ(o.unpack(s, 3, 1));
        SEQ(o.stage(s, 5, euler, 1));    // (the interior copy only)
        if (o.mesh_y()) SEQ(o.pack(s, 13, 1));
        if (o.folded()) SEQ(o.pack(s, 8, 1));
      }
      // the neighbours' rows / the image rows beyond the pivot row, then the substeps on their own stream
      if (o.mesh_y()) SEQ(o.exchange(13, 1));
      if (o.folded()) SEQ(o.exchange(8, 1));
      for (int s = 0; s < n; s++) {
        if (o.mesh_y()) SEQ(o.unpack(s, 13, 1));
        if (o.folded()) SEQ(o.unpack(s, 8, 1));
      }
      SEQ(o.record(5, 1));
      SEQ(o.wait(5, 2));
      EACH(o.stage(s, 56, euler, 2));
    } else {
      EACH(o.unpack(s, 3, 1));
      SEQ(o.record(5, 1));
      SEQ(o.wait(5, 2));
      EACH(o.stage(s, 5, euler, 2));     // (x halo columns of the new eta, U, V included: nothing to exchange after it)
    }
    SEQ(o.record(4, 2));
    lookahead_in_flight = true;
  }
  EACH(o.stage(s, 4, euler, false));
  if (o.catke()) {
    SEQ(sequence_catke_halos(o, n));
    EACH(o.stage(s, 41, euler, false));
  }
  return GB25_OK;
}
// first_time_step!: initialize!, update_state!, then an Euler step (GB-25 src/timestepping_utils.jl:21-27)
gb25_status sequence_first_time_step(StepOps& o, bool& lookahead_in_flight) {
  const int n = o.n();
  if (lookahead_in_flight) {
    SEQ(o.wait(4, 0));             // (the chain's last launches, on the sub stream)
    SEQ(o.record(3, true));
    SEQ(o.wait(3, false));
    lookahead_in_flight = false;
  }
  for (int s = 0; s < n; s++) {
    SEQ(o.local(s, 0));
    SEQ(o.local(s, 1));
    SEQ(o.pack(s, 0, false));
    SEQ(o.pack(s, 2, false));
  }
  SEQ(o.exchange(0, false));
  SEQ(o.exchange(2, false));
  if (o.mesh_y()) {
    for (int s = 0; s < n; s++) {
      SEQ(o.unpack(s, 0, false));
      SEQ(o.unpack(s, 2, false));
      SEQ(o.pack(s, 10, false));
      SEQ(o.pack(s, 12, false));
    }
    SEQ(o.exchange(10, false));
    SEQ(o.exchange(12, false));
    for (int s = 0; s < n; s++) {
      SEQ(o.unpack(s, 10, false));
      SEQ(o.unpack(s, 12, false));
    }
  }
  if (o.folded()) {
    for (int s = 0; s < n; s++) {
      if (!o.mesh_y()) {
        SEQ(o.unpack(s, 0, false));
        SEQ(o.unpack(s, 2, false));
      }
      SEQ(o.local(s, 3));          // mask, y/z layers over the extended columns
      SEQ(o.pack(s, 6, false));
    }
    SEQ(o.exchange(6, false));
    for (int s = 0; s < n; s++) {
      SEQ(o.unpack(s, 6, false));
      SEQ(o.local(s, 4));          // w, pressure, tendencies
    }
  } else {
    for (int s = 0; s < n; s++) {
      if (!o.mesh_y()) {
        SEQ(o.unpack(s, 0, false));
        SEQ(o.unpack(s, 2, false));
      }
      SEQ(o.local(s, 2));
    }
  }
  if (o.catke()) {
    SEQ(sequence_catke_halos(o, n));
    EACH(o.local(s, 7));
  }
  if (o.coupled()) {
    // a coupled model (data-free forcing) updates its state at iteration 0: the atmosphere-ocean fluxes of the initial state,
    // then the tendencies (and, with CATKE, another compute_diffusivities!) that see them
    EACH(o.local(s, 5));
    EACH(o.local(s, 6));
    if (o.catke()) {
      SEQ(sequence_catke_halos(o, n));
      EACH(o.local(s, 7));
    }
  }
  return sequence_time_step(o, 1, lookahead_in_flight);
}
#undef EACH
#undef SEQ

// Dry run: records the operations instead of launching anything (no HIP call) -- how the CPU-only tests see the order.
struct TraceOps : StepOps {
  int nslabs;
  bool adopted, ready;
  std::string log;
  bool fold = false, is_coupled = false, mesh = false, is_lazy = false, early = false, is_catke = false;
  TraceOps(int n_, bool a, bool r) : nslabs(n_), adopted(a), ready(r) {}
  bool folded() override { return fold; }
  bool mesh_y() override { return mesh; }
  bool lazy() override { return is_lazy; }
  bool early_unpack() override { return early; }
  bool early_strips() override { return early && !fold && !mesh; }
  bool coupled() override { return is_coupled; }
  bool catke() override { return is_catke; }
  void add(const char* fmt, ...) {
    char buf[96];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    log += buf;
    log += '\n';
  }
  static const char* st(int on) { return on == 2 ? "sub" : on == 1 ? "comm" : "main"; }
  int n() const override { return nslabs; }
  gb25_status stage(int s, int stage, int euler, int c) override { add("stage %d slab %d euler %d %s", stage, s, euler, st(c)); return GB25_OK; }
  gb25_status pack(int s, int group, int c) override { add("pack %d slab %d %s", group, s, st(c)); return GB25_OK; }
  gb25_status unpack(int s, int group, int c) override { add("unpack %d slab %d %s", group, s, st(c)); return GB25_OK; }
  gb25_status exchange(int group, int c) override { add("exchange %d %s", group, st(c)); return GB25_OK; }
  gb25_status local(int s, int what) override {
    static const char* names[] = {"initialize", "fill_local", "update_state_local", "mask_fill_local", "auxiliaries_tendencies_local",
                                  "first_fluxes_local", "tendencies_local", "catke_finish_local"};
    add("%s slab %d main", names[what], s);
    return GB25_OK;
  }
  bool velocities_ready(int) override { return ready; }
  bool subcycle_adopted(int) override { return adopted; }
  gb25_status record(int slot, int c) override { add("record %d %s", slot, st(c)); return GB25_OK; }
  gb25_status wait(int slot, int waiter) override { add("wait %d %s", slot, st(waiter)); return GB25_OK; }
};

// ---- transports ------------------------------------------------------------------------------------------------------
struct Transport {
  virtual ~Transport() {}
  // buffer set b (0: group 0; 1: groups 1, 3; 2: groups 2, 4), nbytes per side, on stream st
  virtual gb25_status exchange(SlabGroup& G, int b, size_t nbytes, hipStream_t st) = 0;
  virtual const char* name() const = 0;
};

}  // namespace

// The slabs this process drives, their exchange buffers, the second stream and the transport.
struct SlabGroup {
  std::vector<gb25_model*> slabs;
  hipStream_t main = nullptr, comm = nullptr, sub = nullptr;   // (sub: the substeps of the sub-cycle look-ahead)
  hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  // everything the second and third stream hold is over (host writes, grid changes, collective setters)
  hipError_t sync_side() {
    hipError_t e = comm ? hipStreamSynchronize(comm) : hipSuccess;
    if (e == hipSuccess && sub) e = hipStreamSynchronize(sub);
    return e;
  }
  Transport* transport = nullptr;
  // [slab][buffer set][side: 0 west, 1 east]; sets 3, 4 go to the fold partner (side 0 only); sets 5, 6, 7 to the southern
  // (side 0) and northern (side 1) neighbour of a 2-D decomposition
  // sets 8, 9, 10: closure = CATKE -- e and J^b after the e step: x columns (west / east), rows (south / north), fold partner
  static constexpr int NSETS = 11;
  std::vector<std::array<std::array<real*, 2>, NSETS>> send, recv;
  size_t elems[NSETS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};      // elements per side of buffer set b in an exchange
  size_t capacity[NSETS] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // ... and allocated
  bool lookahead_in_flight = false;
  // neighbour handshake of the collective mutators (see collective_guard)
  unsigned long long *tok_dev = nullptr, *tok_host = nullptr;
};

namespace {

// (groups 6, 8: the partner exchanges of a folded grid -- the rows next to the pivot row; the image rows of the sub-cycle)
inline int buffer_set(int group) {
  if (group >= 20) return group - 12;   // (closure = CATKE: groups 20, 21, 22 -> sets 8, 9, 10)
  if (group >= 10) return group == 10 ? 5 : ((group == 11 || group == 13) ? 6 : 7);   // (the y halos of a 2-D decomposition)
  return group == 6 ? 3 : group == 8 ? 4 : group == 0 ? 0 : ((group == 1 || group == 3) ? 1 : 2);
}
// whom a buffer set travels to: 0 the west / east ring neighbours, 1 the fold partner, 2 the southern / northern neighbour
inline int set_kind(int b) { return (b == 3 || b == 4 || b == 10) ? 1 : ((b >= 5 && b <= 7) || b == 9) ? 2 : 0; }
inline int set_sides(int b) { return set_kind(b) == 1 ? 1 : 2; }
// rank = ry Rx + rx: the ring neighbours within the row, the neighbours in the column, the fold partner within the top row
struct MeshPos {
  int Rx, Ry, rx, ry;
  explicit MeshPos(const gb25_model* m) : Rx(m->Rx), Ry(m->Ry), rx(m->rx), ry(m->ry) {}
  MeshPos(int Rx_, int Ry_, int rank) : Rx(Rx_), Ry(Ry_), rx(rank % Rx_), ry(rank / Rx_) {}
  int west() const { return ry * Rx + (rx + Rx - 1) % Rx; }
  int east() const { return ry * Rx + (rx + 1) % Rx; }
  int south() const { return ry > 0 ? (ry - 1) * Rx + rx : -1; }
  int north() const { return ry < Ry - 1 ? (ry + 1) * Rx + rx : -1; }
  int partner() const { return ry * Rx + (Rx - 1 - rx); }
};

// several slabs of one decomposition in this process, all on one device: a ring of device-to-device copies
struct LocalRingTransport : Transport {
  const char* name() const override { return "local"; }
  gb25_status exchange(SlabGroup& G, int b, size_t nbytes, hipStream_t st) override {
    const int P = (int)G.slabs.size();
    for (int r = 0; r < P; r++) {
      gb25_model* m = G.slabs[r];
      const MeshPos q(m);
      if (set_kind(b) == 1) {   // zipper fold: slab rx <-> slab Rx-1-rx of the top row (the middle slab of an odd count is its own partner)
        if (m->g.cv.north_fold) HIPCHK(hipMemcpyAsync(G.recv[q.partner()][b][0], G.send[r][b][0], nbytes, hipMemcpyDeviceToDevice, st));
      } else if (set_kind(b) == 2) {      // my southern pack -> the southern neighbour's northern halo; my northern pack -> ... southern halo
        if (q.south() >= 0) HIPCHK(hipMemcpyAsync(G.recv[q.south()][b][1], G.send[r][b][0], nbytes, hipMemcpyDeviceToDevice, st));
        if (q.north() >= 0) HIPCHK(hipMemcpyAsync(G.recv[q.north()][b][0], G.send[r][b][1], nbytes, hipMemcpyDeviceToDevice, st));
      } else {                  // my west pack -> west neighbour's east halo; my east pack -> east neighbour's west halo
        HIPCHK(hipMemcpyAsync(G.recv[q.west()][b][1], G.send[r][b][0], nbytes, hipMemcpyDeviceToDevice, st));
        HIPCHK(hipMemcpyAsync(G.recv[q.east()][b][0], G.send[r][b][1], nbytes, hipMemcpyDeviceToDevice, st));
      }
    }
    return GB25_OK;
  }
};

// librccl, resolved at run time.  When the host process has RCCL loaded already (PyTorch-ROCm ships one with the same
// soname) dlopen hands back that copy, so there is exactly one RCCL in the process.
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
  bool load() {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) {
      error = std::string("librccl.so.1 could not be loaded: ") + dlerror();
      return false;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(lib, n);
      if (!p) error = std::string("librccl lacks ") + n;
      return p;
    };
    GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
    CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
    CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    CommCount = (decltype(CommCount))sym("ncclCommCount");
    Send = (decltype(Send))sym("ncclSend");
    Recv = (decltype(Recv))sym("ncclRecv");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
    GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    if (!GetUniqueId || !CommInitRank || !CommDestroy || !CommCount || !Send || !Recv || !GroupStart || !GroupEnd || !GetErrorString) {
      lib = nullptr;
      return false;
    }
    return true;
  }
};
RcclApi& rccl() {
  static RcclApi api;
  return api;
}
#define NCCLCHK(call)                                                                                             \
  do {                                                                                                            \
    ncclResult_t r_ = (call);                                                                                     \
    if (r_ != ncclSuccess)                                                                                        \
      return fail(m, GB25_ERR_COMM, "%s:%d: %s failed: %s", __FILE__, __LINE__, #call, rccl().GetErrorString(r_)); \
  } while (0)

// What one rank posts inside ONE ncclGroup for buffer set b, in posting order: (send?, peer rank, side of the buffer set).  The
// single place that knows the protocol -- RcclTransport::exchange posts exactly this, gb25_debug_exchange_plan prints it, and the
// CPU tests prove from it that every rank's ordered sends to a peer mirror that peer's ordered receives (tests/test_distributed_cpu.py).
struct PlanOp { bool send; int peer; int side; };
inline std::vector<PlanOp> exchange_plan(const MeshPos& q, bool north_fold, int b) {
  std::vector<PlanOp> p;
  const int kind = set_kind(b);
  if (kind == 2) {          // 2-D decomposition: the southern and the northern neighbour, where they exist
    if (q.south() >= 0) p.push_back({true, q.south(), 0});
    if (q.north() >= 0) p.push_back({true, q.north(), 1});
    if (q.north() >= 0) p.push_back({false, q.north(), 1});
    if (q.south() >= 0) p.push_back({false, q.south(), 0});
  } else if (kind == 1) {   // zipper fold: the partner is the mirrored rank of the (top) row; a rank that is its own partner copies
    const int self = q.ry * q.Rx + q.rx;
    if (north_fold && q.partner() != self) {
      p.push_back({true, q.partner(), 0});
      p.push_back({false, q.partner(), 0});
    }
  } else {                  // the ring: sends [west pack, east pack], receives [east halo, west halo]
    p.push_back({true, q.west(), 0});
    p.push_back({true, q.east(), 1});
    p.push_back({false, q.east(), 1});
    p.push_back({false, q.west(), 0});
  }
  return p;
}

// one slab per process, one process per GPU: the ring neighbours are ranks rank-1 and rank+1 of the communicator.
// Posting order is part of the protocol: sends [west pack, east pack], receives [east halo, west halo].  With two
// ranks both neighbours are the same peer and the messages of one pair match in posting order, so the peer's FIRST send
// (its west pack) must meet our FIRST receive (our east halo); with one rank (the self-ring) likewise.
struct RcclTransport : Transport {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  // rehearsal of ONE rank of a decomposition alone on one device (tools/slab_selfring.py --mesh): a communicator of size one, every
  // neighbour is the rank itself -- the messages of a group match in posting order, so a southern pack lands in the northern halo
  bool alone = false;
  int peer(int r) const { return alone ? 0 : r; }
  const char* name() const override { return "rccl"; }
  ~RcclTransport() override {
    if (comm) rccl().CommDestroy(comm);
  }
  gb25_status send_recv(gb25_model* m, const void* sw, const void* se, void* rw, void* re, size_t nbytes, hipStream_t st) {
    const MeshPos q(m);
    const int west = peer(q.west()), east = peer(q.east());
    RcclApi& R = rccl();
    NCCLCHK(R.GroupStart());
    NCCLCHK(R.Send(sw, nbytes, ncclInt8, west, comm, st));
    NCCLCHK(R.Send(se, nbytes, ncclInt8, east, comm, st));
    NCCLCHK(R.Recv(re, nbytes, ncclInt8, east, comm, st));
    NCCLCHK(R.Recv(rw, nbytes, ncclInt8, west, comm, st));
    NCCLCHK(R.GroupEnd());
    return GB25_OK;
  }
  gb25_status exchange(SlabGroup& G, int b, size_t nbytes, hipStream_t st) override {
    gb25_model* m = G.slabs[0];
    const MeshPos q(m);
    if (set_kind(b) == 1) {
      if (!m->g.cv.north_fold) return GB25_OK;
      if (q.partner() == rank || alone) {   // (its own partner: the middle slab of an odd count, the self-ring)
        HIPCHK(hipMemcpyAsync(G.recv[0][b][0], G.send[0][b][0], nbytes, hipMemcpyDeviceToDevice, st));
        return GB25_OK;
      }
    }
    const std::vector<PlanOp> plan = exchange_plan(q, m->g.cv.north_fold != 0, b);
    if (plan.empty()) return GB25_OK;
    RcclApi& R = rccl();
    NCCLCHK(R.GroupStart());
    for (const PlanOp& op : plan) {
      if (op.send) NCCLCHK(R.Send(G.send[0][b][op.side], nbytes, ncclInt8, peer(op.peer), comm, st));
      else NCCLCHK(R.Recv(G.recv[0][b][op.side], nbytes, ncclInt8, peer(op.peer), comm, st));
    }
    NCCLCHK(R.GroupEnd());
    return GB25_OK;
  }
};

// the host moves the buffers (synchronous): rehearsal of the multi-process path where RCCL cannot run
struct CallbackTransport : Transport {
  gb25_exchange_fn fn = nullptr;
  void* user = nullptr;
  const char* name() const override { return "callback"; }
  gb25_status exchange(SlabGroup& G, int b, size_t nbytes, hipStream_t st) override {
    gb25_model* m = G.slabs[0];
    HIPCHK(hipStreamSynchronize(st));   // the packs are complete
    // (buffer sets 3, 4: to and from the fold partner rank P-1-r; the east pointers are null)
    // (buffer sets 5 - 7: to and from the southern [west pointers] and northern [east pointers] neighbour of a 2-D
    // decomposition; null where there is none)
    const int kind = set_kind(b);
    if (kind == 1 && !m->g.cv.north_fold) return GB25_OK;
    int rc;
    if (kind == 1) rc = fn(user, b, G.send[0][b][0], nullptr, G.recv[0][b][0], nullptr, (int64_t)nbytes);
    else if (kind == 2) rc = fn(user, b, m->ys_open ? G.send[0][b][0] : nullptr, m->yn_open ? G.send[0][b][1] : nullptr,
                             m->ys_open ? G.recv[0][b][0] : nullptr, m->yn_open ? G.recv[0][b][1] : nullptr, (int64_t)nbytes);
    else rc = fn(user, b, G.send[0][b][0], G.send[0][b][1], G.recv[0][b][0], G.recv[0][b][1], (int64_t)nbytes);
    if (rc != 0) return fail(m, GB25_ERR_COMM, "the host's exchange callback failed with code %d (buffer set %d)", rc, b);
    return GB25_OK;
  }
};

// the real StepOps: the slabs of a SlabGroup
struct GroupOps : StepOps {
  SlabGroup& G;
  explicit GroupOps(SlabGroup& g_) : G(g_) {}
  int n() const override { return (int)G.slabs.size(); }
  struct OnStream {   // run model calls with the model's kernels on the comm stream
    gb25_model* m;
    hipStream_t saved;
    OnStream(gb25_model* m_, hipStream_t st) : m(m_), saved(m_->stream) { m->stream = st; }
    ~OnStream() { m->stream = saved; }
  };
  hipStream_t st(int on) const { return on == 2 ? G.sub : on == 1 ? G.comm : G.main; }
  gb25_status stage(int s, int stage, int euler, int c) override {
    OnStream on(G.slabs[s], st(c));
    return slab_stage(G.slabs[s], stage, euler);
  }
  bool folded() override {   // (2-D decomposition: the top row of ranks only; the others skip the partner's groups)
    for (gb25_model* m : G.slabs)
      if (m->g.cv.north_fold) return true;
    return false;
  }
  bool mesh_y() override { return G.slabs[0]->Ry > 1; }
  bool lazy() override {
    for (gb25_model* m : G.slabs)
      if (!m->step_lazy) return false;
    return true;
  }
  bool early_unpack() override {
    for (gb25_model* m : G.slabs)
      if (!m->early_strips || !m->two_streams) return false;
    return true;
  }
  bool early_strips() override {
    for (gb25_model* m : G.slabs)
      if (!strips_on_comm(m)) return false;
    return true;
  }
  bool coupled() override { return G.slabs[0]->coupled; }
  bool catke() override { return G.slabs[0]->catke; }
  gb25_status pack(int s, int group, int c) override {
    OnStream on(G.slabs[s], st(c));
    const int b = buffer_set(group);
    if ((group == 6 || group == 8 || group == 22) && !G.slabs[s]->g.cv.north_fold) return GB25_OK;
    if (group == 6 || group == 22) return fold_pack(G.slabs[s], G.send[s][b][0], group == 22);
    if (group == 8) return tall_rows_impl(G.slabs[s], G.send[s][b][0], true);
    if ((group >= 10 && group < 20) || group == 21) {
      real* rb[2] = {G.send[s][b][0], G.send[s][b][1]};
      return move_rows(G.slabs[s], group, rb, true);
    }
    if (group == 0) G.slabs[s]->halo_colsum_valid = G.slabs[s]->colsum_valid;   // (every slab alike: same calls, same state)
    real* buf[2] = {G.send[s][b][0], G.send[s][b][1]};
    return pack_unpack(G.slabs[s], group, buf, true);
  }
  gb25_status unpack(int s, int group, int c) override {
    OnStream on(G.slabs[s], st(c));
    const int b = buffer_set(group);
    if ((group == 6 || group == 8 || group == 22) && !G.slabs[s]->g.cv.north_fold) return GB25_OK;
    if (group == 6 || group == 22) return fold_unpack(G.slabs[s], G.recv[s][b][0], group == 22);
    if (group == 8) return tall_rows_impl(G.slabs[s], G.recv[s][b][0], false);
    if ((group >= 10 && group < 20) || group == 21) {
      real* rb[2] = {G.recv[s][b][0], G.recv[s][b][1]};
      return move_rows(G.slabs[s], group, rb, false);
    }
    real* buf[2] = {G.recv[s][b][0], G.recv[s][b][1]};
    return pack_unpack(G.slabs[s], group, buf, false);
  }
  gb25_status exchange(int group, int c) override {
    const int b = buffer_set(group);
    return G.transport->exchange(G, b, G.elems[b] * sizeof(real), st(c));
  }
  gb25_status local(int s, int what) override {
    gb25_model* m = G.slabs[s];
    if (what == 0) return initialize_impl(m);
    if (what == 1) {
      return fill_halos_impl(m, false);
    }
    if (what == 3) {
      gb25_status s_;
      if ((s_ = mask_impl(m))) return s_;
      return fill_halos_impl(m, false, true);
    }
    if (what == 4) {
      gb25_status s_;
      if ((s_ = compute_w_impl(m))) return s_;
      if ((s_ = compute_p_impl(m))) return s_;
      if ((s_ = momentum_impl(m))) return s_;
      if ((s_ = tracers_impl(m))) return s_;
      return m->catke ? catke_step_local(m) : GB25_OK;
    }
    if (what == 5) return atmosphere_ocean_fluxes_impl(m);   // coupled model, iteration 0: fluxes of the initial state
    if (what == 6) {
      gb25_status s_;
      if ((s_ = momentum_impl(m))) return s_;
      if ((s_ = tracers_impl(m))) return s_;
      return m->catke ? catke_step_local(m) : GB25_OK;
    }
    if (what == 7) return catke_finish_local(m);
    return update_state_local_impl(m);
  }
  bool velocities_ready(int s) override {
    gb25_model* m = G.slabs[s];
    return m->ahead_uv_valid && m->baro_ahead && !m->ptr_exposed;
  }
  bool subcycle_adopted(int s) override { return G.slabs[s]->baro_adopted; }
  gb25_status record(int slot, int c) override {
    gb25_model* m = G.slabs[0];
    HIPCHK(hipEventRecord(G.ev[slot], st(c)));
    return GB25_OK;
  }
  gb25_status wait(int slot, int waiter) override {
    gb25_model* m = G.slabs[0];
    HIPCHK(hipStreamWaitEvent(st(waiter), G.ev[slot], 0));
    return GB25_OK;
  }
};

void group_destroy(SlabGroup* G) {
  if (!G) return;
  G->sync_side();
  if (G->main) hipStreamSynchronize(G->main);
  delete G->transport;
  for (auto& s : G->send)
    for (auto& b : s)
      for (real* p : b)
        if (p) hipFree(p);
  for (auto& s : G->recv)
    for (auto& b : s)
      for (real* p : b)
        if (p) hipFree(p);
  for (hipEvent_t e : G->ev)
    if (e) hipEventDestroy(e);
  if (G->comm) hipStreamDestroy(G->comm);
  if (G->sub) hipStreamDestroy(G->sub);
  if (G->tok_dev) hipFree(G->tok_dev);
  if (G->tok_host) hipHostFree(G->tok_host);
  for (gb25_model* m : G->slabs) {
    m->group = nullptr;
    m->stream = m->own_stream;
  }
  delete G;
}

// The bundle of group 0 carries more once an option adds the chunk sums of u, v to it, and a closure brings exchanges of its
// own (CATKE: e, J^b -- sets 8 - 10): sizes again, larger buffers if needed.  Called at the head of the composites; every rank made the same (collective) setter calls.
gb25_status group_refresh(SlabGroup* G) {
  gb25_model* m = G->slabs[0];
  const int n = (int)G->slabs.size();
  for (int b : {0, 3, 5, 8, 9, 10}) {   // (8 - 10: closure = CATKE switched on after the context was built -- its sets grow from one element)
    size_t need = 0;
    for (gb25_model* q : G->slabs)
      need = std::max(need, (size_t)(b == 0 ? halo_buffer_elems(q, 0) : b == 3 ? fold_buffer_elems(q, 3) : b == 5 ? row_buffer_elems(q, 10)
                                     : b == 8 ? halo_buffer_elems(q, 20) : b == 10 ? fold_buffer_elems(q, 10) : row_buffer_elems(q, 21)));
    if (need == G->elems[b]) continue;
    HIPCHK(G->sync_side());
    HIPCHK(hipStreamSynchronize(G->main));
    if (need > G->capacity[b]) {
      for (int s = 0; s < n; s++)
        for (int side = 0; side < set_sides(b); side++) {
          if (G->send[s][b][side]) hipFree(G->send[s][b][side]);
          if (G->recv[s][b][side]) hipFree(G->recv[s][b][side]);
          G->send[s][b][side] = G->recv[s][b][side] = nullptr;
          if (hipMalloc(&G->send[s][b][side], need * sizeof(real)) != hipSuccess ||
              hipMalloc(&G->recv[s][b][side], need * sizeof(real)) != hipSuccess)
            return fail(m, GB25_ERR_OUT_OF_MEMORY, "exchange buffers of %zu elements", need);
        }
      G->capacity[b] = need;
    }
    G->elems[b] = need;
  }
  return GB25_OK;
}

// (The comm stream is an ordinary stream.  A high-priority one -- tried so that it would not share a hardware queue with
// the slab's side stream -- made the 180-column step 2.3x slower beside RCCL's kernel.  What helps is more hardware queues
// for the process: GPU_MAX_HW_QUEUES=8, which bench.py and tools/slab_selfring.py set before the runtime starts (one model per
// process; the Python package itself leaves the runtime's default alone, see bench.py).)
// Builds the exchange context of `n` slabs (n > 1 only with the local transport).  Takes ownership of `tr`.
gb25_status group_create(gb25_model* const* slabs, int n, Transport* tr) {
  gb25_model* m = slabs[0];
  for (int s = 0; s < n; s++) {
    if (!slabs[s] || !slabs[s]->slab) {
      delete tr;
      return fail(m, GB25_ERR_STATE, "slab %d is not a slab of an x decomposition (nranks > 1 or slab_mode = 1)", s);
    }
    if (slabs[s]->group) group_destroy(slabs[s]->group);
  }
  SlabGroup* G = new SlabGroup();
  G->transport = tr;
  G->slabs.assign(slabs, slabs + n);
  G->main = m->own_stream;
  for (int s = 0; s < n; s++) {
    slabs[s]->group = G;
    slabs[s]->group_index = s;
    slabs[s]->stream = G->main;
  }
  gb25_status st = GB25_OK;
  do {
    if (hipSetDevice(m->cfg.device) != hipSuccess ||
        hipStreamCreateWithFlags(&G->comm, hipStreamNonBlocking) != hipSuccess ||
        [&]() {
          // the substeps' few blocks should not queue behind what is left of the tracer kernel's grid: highest priority
          // (option SUB_STREAM_PRIORITY = 0: an ordinary stream)
          int least = 0, greatest = 0;
          if (m->sub_priority && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
            return hipStreamCreateWithPriority(&G->sub, hipStreamNonBlocking, greatest);
          return hipStreamCreateWithFlags(&G->sub, hipStreamNonBlocking);
        }() != hipSuccess) { st = GB25_ERR_HIP; break; }
    for (auto& e : G->ev)
      if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) st = GB25_ERR_HIP;
    if (st) break;
    for (int b = 0; b < 3; b++) G->elems[b] = (size_t)halo_buffer_elems(m, b);
    // (2-D decomposition: the top row of ranks folds, the others do not; every slab gets buffers of the size a folded one needs)
    for (int b = 3; b < 5; b++)
      for (int s = 0; s < n; s++) G->elems[b] = std::max(G->elems[b], (size_t)fold_buffer_elems(slabs[s], b));
    for (int b = 5; b < 8; b++) G->elems[b] = (size_t)row_buffer_elems(m, b == 5 ? 10 : (b == 6 ? 11 : 12));
    G->elems[8] = (size_t)halo_buffer_elems(m, 20);
    G->elems[9] = (size_t)row_buffer_elems(m, 21);
    for (int s = 0; s < n; s++) G->elems[10] = std::max(G->elems[10], (size_t)fold_buffer_elems(slabs[s], 10));
    for (int b = 0; b < SlabGroup::NSETS; b++) G->capacity[b] = G->elems[b];
    G->send.resize(n);
    G->recv.resize(n);
    for (int s = 0; s < n && !st; s++)
      for (int b = 0; b < SlabGroup::NSETS && !st; b++)
        for (int side = 0; side < 2; side++) G->send[s][b][side] = G->recv[s][b][side] = nullptr;
    for (int s = 0; s < n && !st; s++)
      for (int b = 0; b < SlabGroup::NSETS && !st; b++)
        for (int side = 0; side < set_sides(b) && !st; side++) {
          G->send[s][b][side] = G->recv[s][b][side] = nullptr;
          if (hipMalloc(&G->send[s][b][side], G->elems[b] * sizeof(real)) != hipSuccess ||
              hipMalloc(&G->recv[s][b][side], G->elems[b] * sizeof(real)) != hipSuccess)
            st = GB25_ERR_OUT_OF_MEMORY;
        }
    if (st) break;
    if (hipMalloc(&G->tok_dev, 8 * sizeof(unsigned long long)) != hipSuccess ||
        hipHostMalloc(&G->tok_host, 8 * sizeof(unsigned long long)) != hipSuccess) st = GB25_ERR_OUT_OF_MEMORY;
  } while (0);
  if (st) {
    group_destroy(G);
    return fail(m, st, "could not build the exchange context (%s)", hipGetErrorString(hipGetLastError()));
  }
  return GB25_OK;
}

// Calls that change what the look-aheads were made from (host writes, a new dt, an option, a handed-out pointer) void
// the look-aheads of the calling slab -- and, through the halo columns the neighbours hold, theirs.  On a decomposed
// model they are therefore COLLECTIVE: every rank must make the same call (as with set!(model, ...) on an Oceananigans
// Distributed grid).  With the RCCL transport the call shakes hands with both ring neighbours (16 bytes each way): a
// rank that made a different call gets GB25_ERR_STATE here instead of a mismatched exchange later, a rank that made no
// call leaves the others waiting AT the call.  The branch every rank takes in the next step (adopt the look-aheads or
// not) then depends on agreed values only.
gb25_status collective_guard(gb25_model* m, unsigned op, unsigned arg, double payload) {
  SlabGroup* G = m->group;
  if (!G) return GB25_OK;
  RcclTransport* R = dynamic_cast<RcclTransport*>(G->transport);
  if (!R) return GB25_OK;   // local slabs share one host thread; the callback transport is a rehearsal
  unsigned long long bits;
  memcpy(&bits, &payload, sizeof bits);
  unsigned long long* h = G->tok_host;
  h[0] = h[2] = ((unsigned long long)op << 32) | arg;
  h[1] = h[3] = bits;
  HIPCHK(G->sync_side());   // a look-ahead still in flight belongs to the state this call is about to void
  HIPCHK(hipMemcpyAsync(G->tok_dev, h, 4 * sizeof *h, hipMemcpyHostToDevice, G->main));
  gb25_status s = R->send_recv(m, G->tok_dev, G->tok_dev + 2, G->tok_dev + 4, G->tok_dev + 6, 2 * sizeof *h, G->main);
  if (s) return s;
  HIPCHK(hipMemcpyAsync(h + 4, G->tok_dev + 4, 4 * sizeof *h, hipMemcpyDeviceToHost, G->main));
  HIPCHK(hipStreamSynchronize(G->main));
  for (int q = 0; q < 2; q++)
    if (h[4 + 2 * q] != h[0] || h[5 + 2 * q] != h[1])
      return fail(m, GB25_ERR_STATE,
                  "collective call mismatch on rank %d: this rank made call %u(%u, %g) but its %s neighbour made call "
                  "%llu(%llu): setters of a decomposed model must be called alike on every rank",
                  R->rank, op, arg, payload, q == 0 ? "west" : "east", h[4 + 2 * q] >> 32, h[4 + 2 * q] & 0xffffffffull);
  return GB25_OK;
}

}  // namespace

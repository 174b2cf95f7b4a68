// rollout_body.h -- one env.step (PD control -> rigid-body physics step -> reward) executed by ONE
// wavefront on LDS-resident state, written as lane-parallel phases (see wave.h).
//
// This is the product restatement of the reference's hot loop
//     rollout_us -> env.step -> pipeline_step                dial_mpc/core/dial_core.py:36-42
//     UnitreeGo2Env.step / SeqJumpEnv.step / UnitreeH1WalkEnv.step
//                                                           dial_mpc/envs/unitree_go2_env.py:126-261,403-521
//                                                           dial_mpc/envs/unitree_h1_env.py:181-321
//     BaseEnv.act2joint / act2tau                            dial_mpc/envs/base_env.py:38-66
// with the third-party physics (brax -> mujoco.mjx.step, not in the reference tree) re-designed for a
// 64-lane wavefront: every stage below names the MJX function whose result it reproduces.
//
// The file compiles for gfx950 (hipcc) and, with -DDIAL_EMU, for the host wave emulator used by the
// tests (tests/wave_emu).  All per-sample state lives in LDS.  The constants come as `const M* m`: a
// dimension-specialised CModel<D> staged in LDS (Go2, H1, H1 loco: compile-time sizes, square matrix layout,
// register-resident solver of solver_reg.h) or the generic CModel<DimsMax> read from global memory (any model
// within the ABI capacities: run-time sizes, packed triangles, LDS solver below).
#pragma once
#include "derived.h"
#include "dmath.h"

#define MJ_MINVAL 1e-15f
#define MJ_MINIMP 0.0001f
#define MJ_MAXIMP 0.9999f
#define DIAL_PI 3.14159265358979323846f

namespace dial {

// Does the instantiation keep a Cholesky factor in LDS (LDS solver path)?  The dimension-specialised
// instantiations factor in registers instead.
template <class D>
inline constexpr bool kNeedL = D::gen;

// ---------------------------------------------------------------- constraint rows (implicit J)
// Row r < nlim is a joint-limit row (J = lsign * e_dof); the other rows are pyramid edges of contact
// c = (r - nlim) / 4: J = Jn + f * Jt with f = +-friction (constraint._instantiate_contact).
//
// Generic instantiation: the constraint section works on the COMPACTED list of contacts that touch (dist < margin):
// compact index c' -> model contact s.clist[c'] (forward(): after the narrow phase).  A contact that does not touch
// contributes rows with D = 0, aref = 0 -- exact zeros in every sum of the dense formulation -- so dropping it changes
// nothing but the order of the additions; the crate scene carries 52 candidates of which 4-8 touch.
struct RowRef { int is_lim, dof, c, tan; float f; };
template <class M>
DIAL_DEV int con_of(const M*, const Ws& s, int c) {   // model contact of compact contact c
  if constexpr (!M::D::gen) return c;
  else return (int)s.clist[c];
}
template <class M>
DIAL_DEV RowRef row_ref(const M* m, const Ws& s, int r) {
  RowRef rr;
  const int nl = dim_nl(m), nlf = nl + dim_nf(m);   // rows: limits | dry friction | 4 pyramid edges per contact
  rr.is_lim = r < nlf;
  if (rr.is_lim) {
    if constexpr (!M::D::gen) rr.dof = m->jnt_dofadr[m->lim_jnt[r]];
    else rr.dof = r < nl ? m->jnt_dofadr[m->lim_jnt[r]] : m->fri_dof[r - nl];   // (a friction row is J = +e_dof: lsign = 1)
    rr.c = 0; rr.tan = 0; rr.f = 0.f;
  } else {
    int e = (r - nlf) & 3;
    rr.c = (r - nlf) >> 2;
    rr.tan = 1 + (e >> 1);
    float mu = m->con_friction[con_of(m, s, rr.c)][rr.tan - 1];
    rr.f = (e & 1) ? -mu : mu;
    rr.dof = 0;
  }
  return rr;
}
template <class M>
DIAL_DEV float row_dot(const M* m, const Ws& s, int r, const float* v) {
  RowRef rr = row_ref(m, s, r);
  if (rr.is_lim) return s.lsign[r] * v[rr.dof];
  const int nv = dim_nv(m);
  const float* jn = s.Jc + (rr.c * 3) * nv;
  const float* jt = s.Jc + (rr.c * 3 + rr.tan) * nv;
  float acc = 0.f;
  for (int i = 0; i < nv; i++) acc += (jn[i] + jt[i] * rr.f) * v[i];
  return acc;
}
// (J^T f)_i over the first nca (compact) contacts
template <class M>
DIAL_DEV float jt_dot(const M* m, const Ws& s, int i, const float* f, int nca) {
  const int nv = dim_nv(m), nl = dim_nl(m) + dim_nf(m);
  float acc = 0.f;
  int lr = m->dof_limrow[i];
  if (lr >= 0) acc += s.lsign[lr] * f[lr];
  if constexpr (M::D::NFRI != 0) { const int fr = m->dof_frirow[i]; if (fr >= 0) acc += f[fr]; }
  for (int c = 0; c < nca; c++) {
    float jn = s.Jc[(c * 3) * nv + i], j1 = s.Jc[(c * 3 + 1) * nv + i], j2 = s.Jc[(c * 3 + 2) * nv + i];
    const int co = con_of(m, s, c);
    float mu1 = m->con_friction[co][0], mu2 = m->con_friction[co][1];
    const float* fc = f + nl + 4 * c;
    acc += (jn + j1 * mu1) * fc[0];
    acc += (jn - j1 * mu1) * fc[1];
    acc += (jn + j2 * mu2) * fc[2];
    acc += (jn - j2 * mu2) * fc[3];
  }
  return acc;
}
DIAL_DEV float msym(const Ws& s, int i, int j) { return i >= j ? s.M[tri_idx(i, j)] : s.M[tri_idx(j, i)]; }

// ---------------------------------------------------------------- dense Cholesky solve, fused
// Left-looking Cholesky of the packed lower triangle A (row i at i(i+1)/2) into the packed Lo, one phase per column; the
// forward substitution of `rhs` is fused into the same phases (item i owns L[i][k] and rhs[i]), then n
// phases of column-oriented back substitution.  On return x = A^-1 rhs0; rhs and ysol are clobbered.
template <class W>
DIAL_DEV void chol_solve(W& w, int n, const float* A, float* Lo, float* rhs, float* y, float* x) {
  for (int k = 0; k < n; k++) {
    w.items(n - k, [&](int idx) {
      int i = k + idx;
      const int ri = tri_idx(i, 0), rk = tri_idx(k, 0);
      float sik = A[ri + k], dkk = A[rk + k];
      for (int p = 0; p < k; p++) {
        float lkp = Lo[rk + p];
        sik -= Lo[ri + p] * lkp;
        dkk -= lkp * lkp;
      }
      float lkk = DM_SQRT(dkk);
      float yk = rhs[k] / lkk;
      if (i == k) {
        Lo[rk + k] = lkk;
        y[k] = yk;
      } else {
        float lik = sik / lkk;
        Lo[ri + k] = lik;
        rhs[i] -= lik * yk;
      }
    });
  }
  for (int k = n - 1; k >= 0; k--) {
    w.items(k + 1, [&](int i) {
      float xk = y[k] / Lo[tri_idx(k, k)];
      if (i == k) x[k] = xk;
      else y[i] -= Lo[tri_idx(k, i)] * xk;
    });
  }
}
// x = A^-1 rhs for the packed SPD matrix A (M or H).  rhs is clobbered (LDS path).
template <class W, class M>
DIAL_DEV void solve_spd(W& w, const M* m, const Ws& s, const float* A, float* rhs, float* x) {
  chol_solve(w, dim_nv(m), A, s.L, rhs, s.ysol, x);
}

// ---------------------------------------------------------------- constraint._kbi
// (round 6: k, b and the curve's constants come from the row's table -- CModel::kbi_tab, derived.h: kbi_row -- the
//  per-step part is what depends on the position)
DIAL_DEV void kbi(const float* t, float pos, float& k, float& b, float& imp) {
  float t0, t1, dmin, dmax, rwidth, mid, rmid, r1mm;
  load4(t, t0, t1, dmin, dmax);
  load4(t + 4, rwidth, mid, rmid, r1mm);
  const float power = t[8];
  k = t0; b = t1;
  float x = dm::absf(pos) * rwidth;
  float ia, ib;
  if (power == 2.f) {          // MuJoCo's default solimp power: x^2 / mid, no transcendental needed
    ia = rmid * (x * x);
    ib = 1.f - r1mm * ((1.f - x) * (1.f - x));
  } else {
    ia = rmid * DM_POW(x, power);
    ib = 1.f - r1mm * DM_POW(1.f - x, power);
  }
  float yv = x < mid ? ia : ib;
  float im = dmin + yv * (dmax - dmin);
  im = dm::clip(im, dmin, dmax);
  if (x > 1.f) im = dmax;
  imp = im;
}

// ---------------------------------------------------------------- MJX narrow phase helpers (collision_primitive / math)
// math.closest_segment_point: point on [a, b] closest to pt (the 1e-6 in the denominator is MJX's)
DIAL_DEV void closest_segment_point(float* o, const float* a, const float* b, const float* pt) {
  const float ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, pa[3] = {pt[0] - a[0], pt[1] - a[1], pt[2] - a[2]};
  const float t = dm::clip(dm::dot3(pa, ab) / (dm::dot3(ab, ab) + 1e-6f), 0.f, 1.f);
  for (int k = 0; k < 3; k++) o[k] = a[k] + t * ab[k];
}
// math.closest_segment_to_segment_points
DIAL_DEV void closest_segment_to_segment(float* best_a, float* best_b, const float* a0, const float* a1, const float* b0, const float* b1) {
  float da[3] = {a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2]}, db[3] = {b1[0] - b0[0], b1[1] - b0[1], b1[2] - b0[2]};
  const float len_a = DM_SQRT(dm::dot3(da, da)), len_b = DM_SQRT(dm::dot3(db, db));
  for (int k = 0; k < 3; k++) { da[k] = len_a > 0.f ? da[k] / len_a : 0.f; db[k] = len_b > 0.f ? db[k] / len_b : 0.f; }
  const float half_a = len_a * 0.5f, half_b = len_b * 0.5f;
  float a_mid[3], b_mid[3], trans[3];
  for (int k = 0; k < 3; k++) { a_mid[k] = a0[k] + da[k] * half_a; b_mid[k] = b0[k] + db[k] * half_b; trans[k] = a_mid[k] - b_mid[k]; }
  const float dadb = dm::dot3(da, db), dat = dm::dot3(da, trans), dbt = dm::dot3(db, trans);
  const float denom = 1.f - dadb * dadb;
  const float orig_ta = (-dat + dadb * dbt) / (denom + 1e-6f);
  const float orig_tb = dbt + orig_ta * dadb;
  const float ta = dm::clip(orig_ta, -half_a, half_a), tb = dm::clip(orig_tb, -half_b, half_b);
  float ba[3], bb[3], na[3], nb[3];
  for (int k = 0; k < 3; k++) { ba[k] = a_mid[k] + da[k] * ta; bb[k] = b_mid[k] + db[k] * tb; }
  closest_segment_point(na, a0, a1, bb);
  closest_segment_point(nb, b0, b1, ba);
  const float e1[3] = {na[0] - bb[0], na[1] - bb[1], na[2] - bb[2]}, e2[3] = {nb[0] - ba[0], nb[1] - ba[1], nb[2] - ba[2]};
  const float d1 = dm::dot3(e1, e1), d2 = dm::dot3(e2, e2);
  for (int k = 0; k < 3; k++) { best_a[k] = d1 < d2 ? na[k] : ba[k]; best_b[k] = d1 < d2 ? bb[k] : nb[k]; }
}
// collision_primitive make_frame(a): a normalised, b = the more orthogonal of y / z made orthonormal, c = a x b
DIAL_DEV void make_frame(float* fr, const float* a_in) {
  float a[3] = {a_in[0], a_in[1], a_in[2]}, nn = DM_SQRT(dm::dot3(a, a));
  for (int k = 0; k < 3; k++) a[k] /= nn;
  float bb[3] = {0.f, 0.f, 0.f};
  if (-0.5f < a[1] && a[1] < 0.5f) bb[1] = 1.f; else bb[2] = 1.f;
  const float ab = dm::dot3(a, bb);
  for (int k = 0; k < 3; k++) bb[k] -= a[k] * ab;
  nn = DM_SQRT(dm::dot3(bb, bb));
  for (int k = 0; k < 3; k++) bb[k] /= nn;
  float cc[3];
  dm::cross3(cc, a, bb);
  for (int k = 0; k < 3; k++) { fr[k] = a[k]; fr[3 + k] = bb[k]; fr[6 + k] = cc[k]; }
}

}  // namespace dial
#include "box_collide.h"
#include "ls_bracket.h"
#include "solver_reg.h"
#include "solver_cone.h"
#include "smooth_quad.h"
#include "smooth_rows.h"
#include "solver_reg2.h"
#include "smooth_quad2.h"
namespace dial {

// Generic instantiation: x = A^-1 rhs for the packed SPD matrix A (M or H) with the register-resident L D L^T of
// solver_reg.h, instantiated ONCE for the capacity dimension (DIAL_MAX_V, dense elimination order): A is copied into a
// square with an identity block for the dofs the model does not have.  The LDS Cholesky above (one phase per column,
// ~35 k cycles for 18 dofs) stays as the reference implementation the emulator tests compare against (-DDIAL_LDS_CHOL).
template <int NP_>
struct DimsPadV {
  static constexpr int NV = NP_;
  static constexpr bool square = true;
  using Topo = TopoDense;
};
// TopoT: the fill pattern of A (TopoDense, or the dof tree when A has exact zeros off it: see Dims::h_dense)
template <class TopoT = TopoDense, class W, class M>
DIAL_DEV void solve_spd_reg(W& w, const M* m, const Ws& s, const float* A, const float* rhs, float* x) {
  constexpr int NP = M::D::NVP, S = kCholStride<NP>;   // the capacity dimension, or the model's own (compile-time dimensions)
  const int nv = dim_nv(m);
  w.items(NP * S, [&](int e) {
    const int i = e / S, j = e - i * S;
    float v = i == j ? 1.f : 0.f;
    if (i < nv && j < nv) v = A[i >= j ? tri_idx(i, j) : tri_idx(j, i)];
    s.sq[e] = v;
  });
  const vfloat b = w.per_lane([&](int l) { return l < nv ? rhs[l] : 0.f; });
  const vfloat xv = reg_chol<DimsPadV<NP>, TopoT>(w, m, s.sq, b, s.sq);
  w.items(nv, [&](int i) { x[i] = lane_val(xv, i); });
}
// the same with the square already in s.sq (the Hessian's accumulator tile is written straight into it)
template <class TopoT = TopoDense, class W, class M>
DIAL_DEV void solve_sq_reg(W& w, const M* m, const Ws& s, const float* rhs, float* x) {
  constexpr int NP = M::D::NVP;
  const int nv = dim_nv(m);
  const vfloat b = w.per_lane([&](int l) { return l < nv ? rhs[l] : 0.f; });
  const vfloat xv = reg_chol<DimsPadV<NP>, TopoT>(w, m, s.sq, b, s.sq);
  w.items(nv, [&](int i) { x[i] = lane_val(xv, i); });
}

// ================================================================ mjx.forward, second half: contact Jacobians, constraint rows,
// qacc_smooth, Newton solver.  A function of its own so that the generic instantiation can run it on two workspaces (see the
// end of forward()).
template <class W, class M>
DIAL_DEV void forward_constraints(W& w, const M* m, const Ws& s, int nca, int nea) {
  const int nv = dim_nv(m), nl = dim_nl(m), ntri = m->ntri;
  const int nf = dim_nf(m), nlf = nl + nf;   // dry-friction rows sit between the limit rows and the contact rows
  // cost (x 2), force and curvature of ONE row at the value j = J_r qacc - aref_r.  Inequality rows (limits, contacts) are
  // active for j < 0; a dry-friction row (solver._update_constraint) is quadratic while |D j| < frictionloss, i.e. |j| < rf = R f,
  // and beyond that exerts -+f with the cost f (-0.5 rf -+ j)
  const auto row_floss = [&](int r) -> float {
    if constexpr (M::D::NFRI == 0) return 0.f;
    else return (r >= nl && r < nlf) ? m->fri_loss[r - nl] : 0.f;
  };
  const auto row_cost2 = [&](int r, float j) -> float {
    const float d = s.D[r], f = row_floss(r);
    if (f > 0.f) {
      const float rf = f / d;
      return j <= -rf ? 2.f * f * (-0.5f * rf - j) : (j >= rf ? 2.f * f * (-0.5f * rf + j) : d * j * j);
    }
    return j < 0.f ? d * j * j : 0.f;
  };
  const auto row_force = [&](int r, float j) -> float {
    const float d = s.D[r], f = row_floss(r);
    if (f > 0.f) { const float rf = f / d; return j <= -rf ? f : (j >= rf ? -f : d * -j); }
    return j < 0.f ? d * -j : 0.f;
  };
  const auto row_curv = [&](int r, float j) -> float {
    const float d = s.D[r], f = row_floss(r);
    if (f > 0.f) { const float rf = f / d; return (j > -rf && j < rf) ? d : 0.f; }
    return j < 0.f ? d : 0.f;
  };
  // ---- contact Jacobians in the contact frame: Jc[(c,a), i] = frame_a . (jacp_b2 - jacp_b1)(:, i)
  DIAL_MARK(w, 1);
  {
  // Row-layout robots (H1 walk / loco): the item -> table-address arithmetic of these two LDS phases is a set of loop invariants of
  // the T-step loop; hoisted, they were the 11 VGPRs (44 B of scratch) the H1's kernels spilled at the loop entry and re-loaded --
  // scratch_load + s_waitcnt vmcnt -- in every step right here (ISA of the shipped library, round 6).  An opaque copy of the lane id
  // for the two phases keeps them local: a dozen integer instructions per step instead of scratch traffic.
#ifndef DIAL_ROWS_PHASE_SCOPE
#define DIAL_ROWS_PHASE_SCOPE (kRowsDims<typename M::D> && !M::D::gen && !M::D::ell)   // (the Allegro's kernels never spilled here: measured +0.7 % with it)
#endif
  DIAL_LANE_SCOPE_IF(DIAL_ROWS_PHASE_SCOPE, w);
  if constexpr (M::D::ell) {
    // compact rows: J_c is dim x ndof over the dofs that move body1 or body2 (support.jac of both bodies at the contact
    // point, translational rows in the contact frame, then -- condim 6 -- the rotational ones); one item per (contact, dof)
    // -- over the list of contacts that are on (2-4 of the 19 with the ball in the hand: one pass instead of three)
    const int n_con = w.compact(M::D::NC, [&](int c) { return s.con_on[c] != 0.f; }, s.ulist);
    w.items(n_con * M::D::NCD, [&](int it) {
      const int idx = it / M::D::NCD, a = it - idx * M::D::NCD, c = (int)s.ulist[idx];
      const int nd = m->con_ndof[c];
      if (a >= nd) return;
      const int i = m->con_dof[c][a], b1 = m->con_body1[c], b2 = m->con_body2[c], dim = m->con_dim[c];
      float cd[6];
      for (int k = 0; k < 6; k++) cd[k] = s.cdof[6 * i + k];
      const float p[3] = {s.cpos[3 * c], s.cpos[3 * c + 1], s.cpos[3 * c + 2]};
      float dp[3] = {0.f, 0.f, 0.f}, dr[3] = {0.f, 0.f, 0.f};
      if ((m->body_ancmask[b2] >> i) & 1u) {
        const float* cm = s.com + 3 * m->body_rootid[b2];
        const float off[3] = {p[0] - cm[0], p[1] - cm[1], p[2] - cm[2]};
        float cr[3];
        dm::cross3(cr, cd, off);
        for (int k = 0; k < 3; k++) { dp[k] += cd[3 + k] + cr[k]; dr[k] += cd[k]; }
      }
      if ((m->body_ancmask[b1] >> i) & 1u) {
        const float* cm = s.com + 3 * m->body_rootid[b1];
        const float off[3] = {p[0] - cm[0], p[1] - cm[1], p[2] - cm[2]};
        float cr[3];
        dm::cross3(cr, cd, off);
        for (int k = 0; k < 3; k++) { dp[k] -= cd[3 + k] + cr[k]; dr[k] -= cd[k]; }
      }
      float* J = s.Jc + m->con_joff[c] + a;
      for (int k = 0; k < 3; k++) J[k * nd] = dm::dot3(s.cframe + 9 * c + 3 * k, dp);
      if (dim == 6) for (int k = 0; k < 3; k++) J[(3 + k) * nd] = dm::dot3(s.cframe + 9 * c + 3 * k, dr);
    });
    // row velocities J_c qvel, one item per (contact, row) -- not a dim x ndof double loop inside the contact's lane
    w.items(6 * n_con, [&](int it) {
      const int idx = it / 6, k = it - 6 * idx, c = (int)s.ulist[idx];
      if (k >= m->con_dim[c]) return;
      const int nd = m->con_ndof[c];
      const float* J = s.Jc + m->con_joff[c] + k * nd;
      float vel = 0.f;
      for (int a = 0; a < nd; a++) vel += J[a] * s.qvel[m->con_dof[c][a]];
      s.jv[m->con_adr[c] + k] = vel;
    });
  } else if constexpr (!M::D::quad_stage)   // (quadruped stage: Jacobian and rows come out of smooth_quad.h)
  w.items(nca * nv, [&](int it) {
    const int c = it / nv, i = it - c * nv;
    const int co = con_of(m, s, c);   // model contact (positions, frames and constants are indexed by it; Jc by the compact c)
    const int b1 = m->con_body1[co], b2 = m->con_body2[co];
    float cd[6];
    for (int k = 0; k < 6; k++) cd[k] = s.cdof[6 * i + k];
    float p[3] = {s.cpos[3 * co], s.cpos[3 * co + 1], s.cpos[3 * co + 2]};
    float diff[3] = {0.f, 0.f, 0.f};
    if ((m->body_ancmask[b2] >> i) & 1u) {
      const float* cm = s.com + 3 * m->body_rootid[b2];
      float off[3] = {p[0] - cm[0], p[1] - cm[1], p[2] - cm[2]}, cr[3];
      dm::cross3(cr, cd, off);
      for (int k = 0; k < 3; k++) diff[k] += cd[3 + k] + cr[k];
    }
    if ((m->body_ancmask[b1] >> i) & 1u) {
      const float* cm = s.com + 3 * m->body_rootid[b1];
      float off[3] = {p[0] - cm[0], p[1] - cm[1], p[2] - cm[2]}, cr[3];
      dm::cross3(cr, cd, off);
      for (int k = 0; k < 3; k++) diff[k] -= cd[3 + k] + cr[k];
    }
    if constexpr (M::D::square) {   // dof-major pyramid rows J^T[i][4c + e] = Jn +- mu * Jt (one 16-byte store)
      const float jn = dm::dot3(s.cframe + 9 * c, diff), t1 = dm::dot3(s.cframe + 9 * c + 3, diff) * m->con_friction[c][0];
      const float t2 = dm::dot3(s.cframe + 9 * c + 6, diff) * m->con_friction[c][1];
      float* jt = s.Jc + i * M::D::T + 4 * c;
      jt[0] = jn + t1; jt[1] = jn - t1; jt[2] = jn + t2; jt[3] = jn - t2;
    } else {
      for (int a = 0; a < 3; a++) s.Jc[(c * 3 + a) * nv + i] = dm::dot3(s.cframe + 9 * co + 3 * a, diff);
    }
  });
  // ---- constraint.make_constraint: per row D, aref (rows that are "off" get D = 0, aref = 0)
  if constexpr (M::D::ell) {
    // elliptic cones (_efc_contact_elliptic): one item per limit row and per contact; the friction rows' R follows
    // from the normal row (impratio, friction ratios) and their reference acceleration has no position term
    w.items(nl + M::D::NC, [&](int it) {
      if (it < nl) {
        const int r = it, ji = m->lim_jnt[r], qa = m->jnt_qposadr[ji], da = m->jnt_dofadr[ji];
        const float q = s.qpos[qa];
        const float dist_min = q - m->jnt_range[ji][0], dist_max = m->jnt_range[ji][1] - q;
        const float pos = dm::fminf_(dist_min, dist_max) - m->jnt_margin[ji];
        const float sgn = dist_min < dist_max ? 1.f : -1.f;
        s.lsign[r] = sgn;
        if (!(pos < 0.f)) { s.D[r] = 0.f; s.aref[r] = 0.f; return; }
        float k_, b_, imp;
        kbi(m->kbi_tab[m->jnt_kbi[ji]], pos, k_, b_, imp);
        const float R = dm::fmaxf_(m->dof_invweight0[da] * (1.f - imp) / imp, MJ_MINVAL);
        s.aref[r] = -b_ * (sgn * s.qvel[da]) - k_ * imp * pos;
        s.D[r] = 1.f / R;
        return;
      }
      const int c = it - nl, r0 = m->con_adr[c], dim = m->con_dim[c];
      if (s.con_on[c] == 0.f) {
        for (int j = 0; j < dim; j++) { s.D[r0 + j] = 0.f; s.aref[r0 + j] = 0.f; }
        return;
      }
      const float pos = s.cdist[c] - m->con_margin[c];
      const float t = m->con_invw[c][0], iw1 = m->con_invw[c][1];   // (the bodies' inverse weights, summed | / impratio: derived.h)
      const float f0 = m->con_friction[c][0];
      float k_, b_, imp;
      kbi(m->kbi_tab[m->con_kbi[c]], pos, k_, b_, imp);
      for (int j = 0; j < dim; j++) {
        float invw = j == 0 ? t : iw1;
        if (j >= 2) { const float fj = m->con_friction[c][j - 1]; invw = iw1 * (f0 * f0) / (fj * fj); }
        const float vel = s.jv[r0 + j];
        const float R = dm::fmaxf_(invw * (1.f - imp) / imp, MJ_MINVAL);
        s.aref[r0 + j] = -b_ * vel - k_ * imp * (j == 0 ? pos : 0.f);
        s.D[r0 + j] = 1.f / R;
      }
    });
  } else if constexpr (!M::D::quad_stage)
  w.items(nea, [&](int r) {
    if (r < nl) {
      const int ji = m->lim_jnt[r], qa = m->jnt_qposadr[ji], da = m->jnt_dofadr[ji];
      float q = s.qpos[qa];
      float dist_min = q - m->jnt_range[ji][0], dist_max = m->jnt_range[ji][1] - q;
      float pos = dm::fminf_(dist_min, dist_max) - m->jnt_margin[ji];
      float sgn = dist_min < dist_max ? 1.f : -1.f;
      s.lsign[r] = sgn;
      if (!(pos < 0.f)) { s.D[r] = 0.f; s.aref[r] = 0.f; return; }
      float k_, b_, imp;
      kbi(m->kbi_tab[m->jnt_kbi[ji]], pos, k_, b_, imp);
      float R = dm::fmaxf_(m->dof_invweight0[da] * (1.f - imp) / imp, MJ_MINVAL);
      float vel = sgn * s.qvel[da];
      s.aref[r] = -b_ * vel - k_ * imp * pos;
      s.D[r] = 1.f / R;
    } else if (r < nlf) {
      if constexpr (M::D::NFRI != 0) {   // constraint._instantiate_friction: J = e_dof, pos = 0, aref = -b qvel
        const int q = r - nl, da = m->fri_dof[q];
        s.lsign[r] = 1.f;
        float k_, b_, imp;
        kbi(m->kbi_tab[m->fri_kbi[q]], 0.f, k_, b_, imp);
        const float R = dm::fmaxf_(m->dof_invweight0[da] * (1.f - imp) / imp, MJ_MINVAL);
        s.aref[r] = -b_ * s.qvel[da];
        s.D[r] = 1.f / R;
      }
    } else {
      const int c = con_of(m, s, (r - nlf) >> 2);
      s.lsign[r] = 0.f;
      float pos = s.cdist[c] - m->con_margin[c];
      if (!(pos < 0.f)) { s.D[r] = 0.f; s.aref[r] = 0.f; return; }
      const float invweight = m->con_invw[c][0];   // (_efc_contact_pyramidal's row weight: a model constant, derived.h)
      float k_, b_, imp;
      kbi(m->kbi_tab[m->con_kbi[c]], pos, k_, b_, imp);
      float R = dm::fmaxf_(invweight * (1.f - imp) / imp, MJ_MINVAL);
      float vel;
      if constexpr (M::D::square) {
        vel = 0.f;
        for (int i = 0; i < M::D::NV; i++) vel += s.Jc[i * M::D::T + (r - nlf)] * s.qvel[i];
      } else {
        vel = row_dot(m, s, r, s.qvel);
      }
      s.aref[r] = -b_ * vel - k_ * imp * pos;
      s.D[r] = 1.f / R;
    }
  });
  }   // (lane scope of the Jacobian / row phases)
  // ---- smooth.factor_m + forward.fwd_acceleration: qacc_smooth = M^-1 qfrc_smooth (rhs = qfs copy)
  DIAL_MARK(w, 2);
  w.redraw_priority();   // second draw of the physics step (the first: rollout_driver.h), see wave.h
  if constexpr (!M::D::gen) {
    const vfloat vq = reg_chol<typename M::D>(w, m, s.M, w.per_lane([&](int l) { return l < M::D::NV ? s.rhs[l] : 0.f; }), s.H);
    w.items(M::D::NV, [&](int i) { s.qas[i] = lane_val(vq, i); });
  } else {
#ifdef DIAL_LDS_CHOL
    solve_spd(w, m, s, s.M, s.rhs, s.qas);
#else
    solve_spd_reg<typename M::D::Topo>(w, m, s, s.M, s.rhs, s.qas);
#endif
  }
  DIAL_MARK(w, 3);
  if (nea == 0) {
    w.items(nv, [&](int i) { s.qacc[i] = s.qas[i]; });
    return;
  }
  if constexpr (W::half2) {
    solver_reg2(w, m, s);
    return;
  } else if constexpr (M::D::ell) {
    solver_cone(w, m, s);  // elliptic cones: per-contact Newton solver (solver_cone.h)
    return;
  } else if constexpr (!M::D::gen) {
    solver_reg(w, m, s);   // register-resident Newton solver (solver_reg.h)
    return;
  }

  // ================================================================ solver.solve (Newton)
  // The products J v, M v, J^T f of this solver run over the contacts' three FRAME rows (Jn, Jt1, Jt2), not over the four pyramid
  // rows J_e = Jn +- mu Jt: J_e . v = Jn . v +- mu (Jt . v) -- 3 dot products per contact instead of 4 x 2 -- and J^T f =
  // Jn^T (f0+f1+f2+f3) + Jt1^T mu1 (f0-f1) + Jt2^T mu2 (f2-f3).  The friction coefficients are read by COMPACT index (s.cmu,
  // filled once per step below): inside a loop of run-time length a table look-up through the compaction list is two
  // dependent LDS round trips per iteration that nothing can be scheduled around (round 4 section profile: 20 k + 17 k of the
  // crate scene's 114 k cycles per step sat in these products).  `fdot`: scratch for the frame-row products, in s.quad
  // (free until the Hessian's weights / a wide line search).
  w.items(nca, [&](int c) {
    const int co = con_of(m, s, c);
    s.cmu[2 * c] = m->con_friction[co][0];
    s.cmu[2 * c + 1] = m->con_friction[co][1];
  });
  float* const fdot = s.quad;
  // value of constraint row r given the frame-row products `fd` (stride: 1 vector) and the vector itself
  const auto row_from = [&](int r, const float* fd, const float* v) -> float {
    if (r < nl) return s.lsign[r] * v[m->jnt_dofadr[m->lim_jnt[r]]];
    if constexpr (M::D::NFRI != 0) { if (r < nlf) return v[m->fri_dof[r - nl]]; }
    const int c = (r - nlf) >> 2, e = (r - nlf) & 3;
    const float mu = s.cmu[2 * c + (e >> 1)], ft = fd[3 * c + 1 + (e >> 1)];
    return fd[3 * c] + ((e & 1) ? -mu : mu) * ft;
  };
  // warm-start selection: cost at qacc_warmstart vs cost at qacc_smooth
  w.items(3 * nca + nv, [&](int it) {
    if (it < 3 * nca) {   // frame row `it` against both candidates: one pass over the row
      const float* J = s.Jc + it * nv;
      float aw0 = 0.f, aw1 = 0.f, as0 = 0.f, as1 = 0.f;
      int i = 0;
      for (; i + 1 < nv; i += 2) {
        aw0 += J[i] * s.warm[i]; aw1 += J[i + 1] * s.warm[i + 1];
        as0 += J[i] * s.qas[i]; as1 += J[i + 1] * s.qas[i + 1];
      }
      if (i < nv) { aw0 += J[i] * s.warm[i]; as0 += J[i] * s.qas[i]; }
      fdot[it] = aw0 + aw1;
      fdot[3 * nca + it] = as0 + as1;
    } else {
      const int i = it - 3 * nca;
      float aw = 0.f, as = 0.f;
      for (int j = 0; j < nv; j++) { const float mij = msym(s, i, j); aw += mij * s.warm[j]; as += mij * s.qas[j]; }
      s.MaW[i] = aw;
      s.MaS[i] = as;
    }
  });
  w.items(nea, [&](int r) {
    s.JarefW[r] = row_from(r, fdot, s.warm) - s.aref[r];
    s.JarefS[r] = row_from(r, fdot + 3 * nca, s.qas) - s.aref[r];
  });
  float cw, gw, cs, gs;
  if (nea <= 64) {   // one row per lane: the four sums as one batch of stage-interleaved reductions
    vfloat t4[4];
    t4[0] = w.per_lane([&](int l) { return l < nea ? row_cost2(l, s.JarefW[l]) : 0.f; });
    t4[1] = w.per_lane([&](int l) { return l < nv ? (s.MaW[l] - s.qfs[l]) * (s.warm[l] - s.qas[l]) : 0.f; });
    t4[2] = w.per_lane([&](int l) { return l < nea ? row_cost2(l, s.JarefS[l]) : 0.f; });
    t4[3] = w.per_lane([&](int l) { return l < nv ? (s.MaS[l] - s.qfs[l]) * (s.qas[l] - s.qas[l]) : 0.f; });
    float r4[4];
    w.vsumN(t4, r4);
    cw = r4[0]; gw = r4[1]; cs = r4[2]; gs = r4[3];
  } else {
    cw = w.sum(nea, [&](int r) { return row_cost2(r, s.JarefW[r]); });
    gw = w.sum(nv, [&](int i) { return (s.MaW[i] - s.qfs[i]) * (s.warm[i] - s.qas[i]); });
    cs = w.sum(nea, [&](int r) { return row_cost2(r, s.JarefS[r]); });
    gs = w.sum(nv, [&](int i) { return (s.MaS[i] - s.qfs[i]) * (s.qas[i] - s.qas[i]); });
  }
  const float cost_w = 0.5f * cw + 0.5f * gw, cost_s = 0.5f * cs + 0.5f * gs;
  const bool use_warm = cost_w < cost_s;
  w.items(nea + nv, [&](int it) {
    if (it < nea) s.Jaref[it] = use_warm ? s.JarefW[it] : s.JarefS[it];
    else {
      const int i = it - nea;
      s.qacc[i] = use_warm ? s.warm[i] : s.qas[i];
      s.Ma[i] = use_warm ? s.MaW[i] : s.MaS[i];
    }
  });
  float cost = use_warm ? cost_w : cost_s;
  float gauss = use_warm ? 0.5f * gw : 0.5f * gs;
  float prev_cost = INFINITY;
  const float scale = 1.f / (m->meaninertia * (float)(nv > 1 ? nv : 1));
  const bool rule_swap = m->ls_rule == DIAL_LS_SWAP;

  // _update_constraint forces + _update_gradient; returns through LDS (frc, qfc, grad)
  auto constraint_grad = [&]() {
    // per contact the three force combinations the frame rows see: (f0+f1+f2+f3, mu1 (f0-f1), mu2 (f2-f3))
    w.items(nca, [&](int c) {
      const int r0 = nlf + 4 * c;
      const float f0 = row_force(r0, s.Jaref[r0]), f1 = row_force(r0 + 1, s.Jaref[r0 + 1]);
      const float f2 = row_force(r0 + 2, s.Jaref[r0 + 2]), f3 = row_force(r0 + 3, s.Jaref[r0 + 3]);
      fdot[3 * c] = (f0 + f1) + (f2 + f3);
      fdot[3 * c + 1] = s.cmu[2 * c] * (f0 - f1);
      fdot[3 * c + 2] = s.cmu[2 * c + 1] * (f2 - f3);
    });
    w.items(nv, [&](int i) {
      float qa = 0.f, qb = 0.f;
      const int lr = m->dof_limrow[i];
      if (lr >= 0) qa += s.lsign[lr] * row_force(lr, s.Jaref[lr]);
      if constexpr (M::D::NFRI != 0) { const int fr = m->dof_frirow[i]; if (fr >= 0) qa += row_force(fr, s.Jaref[fr]); }
      const float* J = s.Jc + i;
      int c = 0;
      for (; c + 1 < nca; c += 2) {   // two contacts per trip: six independent fetches in flight
        const float* Ja = J + 3 * c * nv;
        const float* Jb = Ja + 3 * nv;
        qa += (Ja[0] * fdot[3 * c] + Ja[nv] * fdot[3 * c + 1]) + Ja[2 * nv] * fdot[3 * c + 2];
        qb += (Jb[0] * fdot[3 * c + 3] + Jb[nv] * fdot[3 * c + 4]) + Jb[2 * nv] * fdot[3 * c + 5];
      }
      if (c < nca) { const float* Ja = J + 3 * c * nv; qa += (Ja[0] * fdot[3 * c] + Ja[nv] * fdot[3 * c + 1]) + Ja[2 * nv] * fdot[3 * c + 2]; }
      float qc = qa + qb;
      s.qfc[i] = qc;
      float g = s.Ma[i] - s.qfs[i] - qc;
      s.grad[i] = g;
      s.rhs[i] = g;
    });
  };
  // H = M + J^T diag(D*active) J (lower triangle), Cholesky, search = -H^-1 grad
  auto newton_dir = [&]() {
    // The contact part of H as a DENSE GEMM on the matrix cores: with the five weights of a contact's 3 x 3 block
    //   W = [[d0+d1+d2+d3, mu1 (d0-d1), mu2 (d2-d3)], [., mu1^2 (d0+d1), 0], [., 0, mu2^2 (d2+d3)]]   (d_e = D of the active edges)
    // the four pyramid rows J_e = Jn +- mu Jt of contact c contribute Jc^T W Jc (Jc = the contact's 3 frame rows), i.e.
    //   H = M + [Jc_0 ... Jc_n]^T blockdiag(W_c) [Jc_0 ... Jc_n]   =   A (nv x 3 nca)  .  B (3 nca x nv),   B = W Jc.
    // One v_mfma_f32_32x32x2_f32 multiplies a 32 x 2 slice of A with a 2 x 32 slice of B into the 32 x 32 accumulator tile
    // (16 VGPRs): two per contact (frame rows 0-1, then row 2 and a zero row), A and B built lane-wise from THREE LDS reads
    // per contact (lane = column / dof l & 31, k-slot l >> 5).  With 8-16 touching contacts K = 24 .. 48: this IS the batched
    // dense GEMM that pays on MFMA -- the per-entry VALU loop over contacts it replaces took 12.6 k of the crate scene's 114 k
    // cycles per step (profiles/r04_sections_*crate*), where the Go2's 4-contact, work-list assembly did not (round 2:
    // profiles/r02_ubench_mfma_jtdj.txt).  Structural zeros (dofs of different branches under world-only contacts) come out
    // as exact zeros: every product of such a pair has a zero factor.
    const int nent = (nv * (nv + 1)) / 2;
    w.items(nca, [&](int c) {   // the contact's weights, laid out for the two k-slots: [W00 W01 W02 | W02 0 W22] and [W01 W11 0 | 0 0 0]
      const int r0 = nlf + 4 * c;
      const float mu1 = s.cmu[2 * c], mu2 = s.cmu[2 * c + 1];
      const float d0 = s.Jaref[r0] < 0.f ? s.D[r0] : 0.f, d1 = s.Jaref[r0 + 1] < 0.f ? s.D[r0 + 1] : 0.f;
      const float d2 = s.Jaref[r0 + 2] < 0.f ? s.D[r0 + 2] : 0.f, d3 = s.Jaref[r0 + 3] < 0.f ? s.D[r0 + 3] : 0.f;
      const float W00 = (d0 + d1) + (d2 + d3), W01 = mu1 * (d0 - d1), W11 = (mu1 * mu1) * (d0 + d1);
      const float W02 = mu2 * (d2 - d3), W22 = (mu2 * mu2) * (d2 + d3);
      float* q = s.quad + 12 * c;   // (free here: the line search fills it later, and only when there are more than 64 rows)
      q[0] = W00; q[1] = W01; q[2] = W02; q[3] = W02; q[4] = 0.f; q[5] = W22;
      q[6] = W01; q[7] = W11; q[8] = 0.f; q[9] = 0.f; q[10] = 0.f; q[11] = 0.f;
    });
    DIAL_MARK(w, 14);
    // limit / friction rows only touch the diagonal
    const auto diag_rows = [&](int i) -> float {
      float acc = 0.f;
      const int lr = m->dof_limrow[i];
      if (lr >= 0 && s.Jaref[lr] < 0.f) acc += s.D[lr];  // lsign^2 = 1
      if constexpr (M::D::NFRI != 0) { const int fr = m->dof_frirow[i]; if (fr >= 0) acc += row_curv(fr, s.Jaref[fr]); }
      return acc;
    };
#ifndef DIAL_EMU
    {
      typedef float f16v __attribute__((ext_vector_type(16)));
      f16v acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int col = w.lane & 31, half = w.lane >> 5;
      const bool live = col < nv;
      const float* jcol = s.Jc + (live ? col : 0);
      const float* qh = s.quad + 6 * half;
      for (int c = 0; c < nca; c++) {
        const float* J = jcol + 3 * c * nv;
        const float jn = live ? J[0] : 0.f, jt1 = live ? J[nv] : 0.f, jt2 = live ? J[2 * nv] : 0.f;
        const float* q = qh + 12 * c;
        const float b1 = (q[0] * jn + q[1] * jt1) + q[2] * jt2;
        const float b2 = (q[3] * jn + q[4] * jt1) + q[5] * jt2;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? jt1 : jn, b1, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(half ? 0.f : jt2, b2, acc, 0, 0, 0);
      }
      DIAL_MARK(w, 12);
      // accumulator a of lane (col, half) is entry (row, col), row = 8 (a / 4) + 4 half + a % 4 (profiles/r02_ubench_mfma_jtdj.txt).
      // The tile goes STRAIGHT into the square the register L D L^T reads (s.sq, both triangles; identity block for the dofs a
      // capacity-dimension model does not have): no packed H, no packed -> square copy.
      // (Compile-time dimensions only: in the capacity-dimension kernel this write-back runs into an LLVM address-space
      // inference bug -- "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base" --; it keeps the packed H.)
      constexpr int NP = M::D::NVP, SS = kCholStride<NP>;
      if constexpr (!M::D::is_static) {
#pragma unroll
        for (int a = 0; a < 16; a++) {
          const int row = (a >> 2) * 8 + half * 4 + (a & 3);
          if (row < nv && col <= row) { const int e = tri_idx(row, col); s.H[e] = s.M[e] + acc[a]; }
        }
        w.sync();
        w.items(nv, [&](int i) { s.H[tri_idx(i, i)] += diag_rows(i); });
      } else {
#pragma unroll
      for (int a = 0; a < 16; a++) {
        const int row = (a >> 2) * 8 + half * 4 + (a & 3);
        if (row < NP && col < NP) {
          const bool in = row < nv && col < nv;
          const int hi_ = row > col ? row : col, lo_ = row > col ? col : row;
          const float mv_ = s.M[in ? tri_idx(hi_, lo_) : 0];   // (an unconditional fetch from a clamped index: no select of addresses)
          s.sq[row * SS + col] = in ? mv_ + acc[a] : (row == col ? 1.f : 0.f);
        }
      }
      w.sync();
      // the limit / friction rows' diagonal terms in a phase of their own (inside the unrolled tile loop each was a divergent
      // branch with two dependent LDS round trips: 16 of them made the write-back cost more than the GEMM)
      w.items(nv, [&](int i) { s.sq[i * SS + i] += diag_rows(i); });
      }
    }
#else
    // host emulator: the same algebra entry by entry (the matrix cores' internal summation order is not modelled)
    w.items(nent, [&](int e) {
      int i = 0;
      while (((i + 1) * (i + 2)) / 2 <= e) i++;
      const int j = e - (i * (i + 1)) / 2;
      float acc = 0.f;
      for (int c = 0; c < nca; c++) {
        const float* J = s.Jc + 3 * c * nv;
        const float* q = s.quad + 12 * c;
        const float jn = J[j], jt1 = J[nv + j], jt2 = J[2 * nv + j];
        acc += J[i] * ((q[0] * jn + q[1] * jt1) + q[2] * jt2) + J[nv + i] * ((q[6] * jn + q[7] * jt1) + q[8] * jt2);
        acc += J[2 * nv + i] * ((q[3] * jn + q[4] * jt1) + q[5] * jt2);
      }
      s.H[e] = (s.M[e] + acc) + (i == j ? diag_rows(i) : 0.f);
    });
#endif
    DIAL_MARK(w, 5);
#ifdef DIAL_LDS_CHOL
    solve_spd(w, m, s, s.H, s.rhs, s.search);
#elif defined(DIAL_EMU)
    solve_spd_reg<std::conditional_t<M::D::h_dense, TopoDense, typename M::D::Topo>>(w, m, s, s.H, s.rhs, s.search);
#else
    if constexpr (M::D::is_static) solve_sq_reg<std::conditional_t<M::D::h_dense, TopoDense, typename M::D::Topo>>(w, m, s, s.rhs, s.search);
    else solve_spd_reg(w, m, s, s.H, s.rhs, s.search);
#endif
    w.items(nv, [&](int i) { s.search[i] = -s.search[i]; });
  };

  // Newton iterations.  Each stage appears once in the instruction stream (the kernel is instruction-cache
  // sensitive): [forces + gradient] -> convergence test -> [H, Cholesky, search] -> [line search].
  int niter = 0;
  for (;;) {
#ifndef DIAL_EMU
    asm volatile("" : "+v"(w.lane));   // (as at the top of the step loop, rollout_driver.h: nothing lane-derived is hoisted out of
    w.lane_r = w.lane;                 //  the Newton loop into registers that the Hessian tile and the line search need)
#endif
    constraint_grad();
    float gn_b = 0.f;
    const bool batched = nea <= 64;   // one row per lane: cost, Gauss term and |grad|^2 as one batch of reductions
    if (batched) {
      vfloat t3[3];
      t3[0] = w.per_lane([&](int l) { return l < nea ? row_cost2(l, s.Jaref[l]) : 0.f; });
      t3[1] = w.per_lane([&](int l) { return l < nv ? (s.Ma[l] - s.qfs[l]) * (s.qacc[l] - s.qas[l]) : 0.f; });
      t3[2] = w.per_lane([&](int l) { return l < nv ? s.grad[l] * s.grad[l] : 0.f; });
      float r3[3];
      w.vsumN(t3, r3);
      if (niter > 0) {
        gauss = 0.5f * r3[1];
        prev_cost = cost;
        cost = 0.5f * r3[0] + gauss;
      }
      gn_b = r3[2];
    } else if (niter > 0) {
      float c2 = w.sum(nea, [&](int r) { return row_cost2(r, s.Jaref[r]); });
      float g2 = w.sum(nv, [&](int i) { return (s.Ma[i] - s.qfs[i]) * (s.qacc[i] - s.qas[i]); });
      gauss = 0.5f * g2;
      prev_cost = cost;
      cost = 0.5f * c2 + gauss;
    }
    DIAL_MARK(w, 4);
    bool done;
    if (m->iterations != 1) {
      float gn = batched ? gn_b : w.sum(nv, [&](int i) { return s.grad[i] * s.grad[i]; });
      float improvement = scale * (prev_cost - cost), gradient = scale * DM_SQRT(gn);
      done = niter >= m->iterations || improvement < m->tolerance || gradient < m->tolerance;
    } else {
      done = niter >= 1;
    }
    if (done) break;
    newton_dir();
    DIAL_MARK(w, 6);
    // ---------------- solver._linesearch
    DIAL_MARK(w, 8);
    w.items(3 * nca + nv, [&](int it) {
      if (it < 3 * nca) {
        const float* J = s.Jc + it * nv;
        float a0 = 0.f, a1 = 0.f;
        int i = 0;
        for (; i + 1 < nv; i += 2) { a0 += J[i] * s.search[i]; a1 += J[i + 1] * s.search[i + 1]; }
        if (i < nv) a0 += J[i] * s.search[i];
        fdot[it] = a0 + a1;
      } else {
        const int i = it - 3 * nca;
        float acc = 0.f;
        for (int j = 0; j < nv; j++) acc += msym(s, i, j) * s.search[j];
        s.mv[i] = acc;
      }
    });
    w.items(nea, [&](int r) { s.jv[r] = row_from(r, fdot, s.search); });
    float sn2, s1, s2;
    w.sum3(nv, [&](int i, float& a, float& b, float& c) {
      float sv = s.search[i];
      a = sv * sv; b = sv * s.Ma[i] - sv * s.qfs[i]; c = sv * s.mv[i];
    }, sn2, s1, s2);
    const float smag = DM_SQRT(sn2) * m->meaninertia * (float)(nv > 1 ? nv : 1);
    const float gtol = m->tolerance * m->ls_tolerance * smag;
    const float qg0 = gauss, qg1 = s1, qg2 = 0.5f * s2;
    // the three coefficients one row contributes at the step alpha (solver._eval_pt): an inequality row its quadratic while
    // Jaref + alpha jv < 0; a dry-friction row its quadratic inside |x| < rf and the linear pieces f (-0.5 rf -+ x) outside
    const auto row_terms = [&](float f, float ja, float jv, float d, float k0, float k1, float k2, float alpha, float& a, float& b, float& c) {
      const float x = ja + jv * alpha;
      if (M::D::NFRI != 0 && f > 0.f) {
        const float rf = f / d;
        const bool neg = x <= -rf, pos = x >= rf;
        a = neg ? f * (-0.5f * rf - ja) : (pos ? f * (-0.5f * rf + ja) : k0);
        b = neg ? -f * jv : (pos ? f * jv : k1);
        c = (neg || pos) ? 0.f : k2;
        return;
      }
      const bool act = x < 0.f;
      a = act ? k0 : 0.f; b = act ? k1 : 0.f; c = act ? k2 : 0.f;
    };
    const int kg = fkey(gtol), kng = fkey(-gtol);
    float alpha;
    bool improved;
    if (nea <= 64) {
      // Up to 64 rows (after compaction that is every step of the crate scenes but the rare ones with > 13 touching contacts):
      // the THREE trial points of a bracketing iteration (lo_next, hi_next, mid) are evaluated at once by three 16-lane groups,
      // as in solver_reg.h -- lane (g, l) = (lane >> 4, lane & 15) owns rows l, l + 16, l + 32, l + 48 for point g; per row a
      // handful of selects and three adds, then three interleaved 16-lane DPP reductions, and every lane of a group finishes
      // its point (cost, slope, Newton step, integer keys), so that the scalar bracket logic fetches 4 words per point.
      // (Round 3 evaluated the points one after the other, each with full-wave reductions: 19 k of 114 k cycles per step.)
      constexpr int RPL = 4;
      const vfloat vzero = vsplat(0.f);
      vfloat lJa[RPL], ljv[RPL], lD[RPL], lF[RPL], Q0[RPL], Q1[RPL], Q2[RPL];
      constexpr bool has_fri = M::D::NFRI != 0;   // only then the three-zone rows exist (and D / frictionloss stay live)
#pragma unroll
      for (int q = 0; q < RPL; q++) {
        const auto rowi = [&](int l) { return (l & 15) + 16 * q; };
        lJa[q] = w.per_lane([&](int l) { const int r = rowi(l); return (l < 48 && r < nea) ? s.Jaref[r] : 0.f; });
        ljv[q] = w.per_lane([&](int l) { const int r = rowi(l); return (l < 48 && r < nea) ? s.jv[r] : 0.f; });
        lD[q] = w.per_lane([&](int l) { const int r = rowi(l); return (l < 48 && r < nea) ? s.D[r] : 0.f; });
        lF[q] = w.per_lane([&](int l) { const int r = rowi(l); return (has_fri && l < 48 && r < nea) ? row_floss(r) : 0.f; });
        const vfloat dja = lD[q] * lJa[q], djv = lD[q] * ljv[q];
        Q0[q] = (lJa[q] * 0.5f) * dja;
        Q1[q] = ljv[q] * dja;
        Q2[q] = (ljv[q] * 0.5f) * djv;
      }
      const vbool g0 = w.lane_lt(16), g01 = w.lane_lt(32);
      vfloat pk[4];
      auto ls_eval3 = [&](float a0, float a1, float a2) {
        const vfloat va = vsel(g0, vsplat(a0), vsel(g01, vsplat(a1), vsplat(a2)));
        vfloat t3[3] = {vzero, vzero, vzero};
#pragma unroll
        for (int q = 0; q < RPL; q++) {
          vfloat c3[3];
          w.per_lane_n(c3, [&](int l, float* o) {
            row_terms(has_fri ? lane_val(lF[q], l) : 0.f, lane_val(lJa[q], l), lane_val(ljv[q], l), lane_val(lD[q], l), lane_val(Q0[q], l),
                      lane_val(Q1[q], l), lane_val(Q2[q], l), lane_val(va, l), o[0], o[1], o[2]);
          });
          t3[0] = t3[0] + c3[0]; t3[1] = t3[1] + c3[1]; t3[2] = t3[2] + c3[2];
        }
        w.row16_sum3(t3[0], t3[1], t3[2]);
        const vfloat q0 = t3[0] + vsplat(qg0), q1 = t3[1] + vsplat(qg1), q2 = t3[2] + vsplat(qg2);
        const vfloat vcost = (va * va) * q2 + va * q1 + q0;
        // single-rounding slope 2 alpha q2 + q1, as on the reference's platform (XLA contracts it into an FMA): with two
        // roundings the slope at a Newton point evaluates to EXACTLY 0 about half of the time, `_in_bracket` rejects such a
        // candidate and the truncated search falls back to bisection -- a rounding lottery the reference does not play
        const vfloat vd0 = vfma(va * 2.f, q2, q1);
        const vfloat vd1 = q2 * 2.f + vsel(veq0(q2), vsplat(MJ_MINVAL), vzero);
        w.per_lane_n(pk, [&](int l, float* o) {
          ls_pack(lane_val(va, l), lane_val(vcost, l), lane_val(vd0, l), lane_val(vd1, l), o[0], o[1], o[2], o[3]);
        });
      };
      auto point_at = [&](int lane) {   // all four words of the point held by the group that starts at `lane`
        LsPt p;
        p.alpha = fbits(bcast(pk[0], lane)); p.nalpha = fbits(bcast(pk[1], lane)); p.cost = fbits(bcast(pk[2], lane)); p.d0 = fbits(bcast(pk[3], lane));
        return p;
      };
      ls_eval3(0.f, 0.f, 0.f);
      const LsPt p0 = point_at(0);
      ls_eval3(bitsf(p0.nalpha), bitsf(p0.nalpha), bitsf(p0.nalpha));
      LsPt lo, hi;
      ls_open(p0, point_at(0), lo, hi);
      {
        const LsGate gate = ls_gate(kg, kng);   // (one scalar compare + branch per loop condition, see solver_reg.h)
        const int max_ls = DM_UNIFORM_I(m->ls_iterations);
        int ls_iter = 0;
        while (ls_iter < max_ls) {
          DM_NOFOLD();
          if (ls_converged_lo(lo, gate)) break;
          DM_NOFOLD();
          if (ls_converged_hi(hi, gate)) break;
          ls_eval3(bitsf(lo.nalpha), bitsf(hi.nalpha), 0.5f * (bitsf(lo.alpha) + bitsf(hi.alpha)));   // groups: lo_next, hi_next, mid
          const bool swap = ls_update_lazy<true>(rule_swap, lo, hi, fbits(bcast(pk[3], 0)), fbits(bcast(pk[3], 16)), fbits(bcast(pk[3], 32)), 0, 16, 32,
                                                       [&](int word, int lane) { return fbits(bcast(pk[word], lane)); });
          ls_iter++;
          if (!swap) break;
        }
      }
      improved = ls_result(p0, lo, hi, alpha);
    } else {
      // more than 64 rows: the coefficients stay in LDS, the points are evaluated one after the other with lane-strided sums
      w.items(nea, [&](int r) {
        const float ja = s.Jaref[r], jv = s.jv[r], d = s.D[r];
        s.quad[3 * r] = (ja * 0.5f) * ja * d; s.quad[3 * r + 1] = jv * ja * d; s.quad[3 * r + 2] = (jv * 0.5f) * jv * d;
      });
      auto ls_point = [&](float alpha_) {
        float q0, q1, q2;
        w.sum3(nea, [&](int r, float& a, float& b, float& c) {
          row_terms(row_floss(r), s.Jaref[r], s.jv[r], s.D[r], s.quad[3 * r], s.quad[3 * r + 1], s.quad[3 * r + 2], alpha_, a, b, c);
        }, q0, q1, q2);
        q0 += qg0; q1 += qg1; q2 += qg2;
        const float cost_ = alpha_ * alpha_ * q2 + alpha_ * q1 + q0;
        const float d0 = DM_FMA(2.f * alpha_, q2, q1);   // single rounding (see above)
        const float d1 = 2.f * q2 + (q2 == 0.f ? MJ_MINVAL : 0.f);
        float pa, pn, pc, pd;
        ls_pack(alpha_, cost_, d0, d1, pa, pn, pc, pd);   // integer keys: ls_bracket.h
        LsPt p;
        p.alpha = fbits(pa); p.nalpha = fbits(pn); p.cost = fbits(pc); p.d0 = fbits(pd);
        return p;
      };
      const LsPt p0 = ls_point(0.f);
      LsPt lo, hi;
      ls_open(p0, ls_point(bitsf(p0.nalpha)), lo, hi);
      bool swap = true;
      int ls_iter = 0;
      for (;;) {
        const bool done = (ls_iter >= m->ls_iterations) | !swap | ls_converged(lo, hi, kg, kng);
        if (done) break;
        const LsPt lo_next = ls_point(bitsf(lo.nalpha));
        const LsPt hi_next = ls_point(bitsf(hi.nalpha));
        const LsPt mid = ls_point(0.5f * (bitsf(lo.alpha) + bitsf(hi.alpha)));
        swap = ls_update(rule_swap, lo, hi, lo_next, hi_next, mid);
        ls_iter++;
      }
      improved = ls_result(p0, lo, hi, alpha);
    }
    if (improved) {
      w.items(nv + nea, [&](int it) {
        if (it < nv) { s.qacc[it] += s.search[it] * alpha; s.Ma[it] += s.mv[it] * alpha; }
        else s.Jaref[it - nv] += s.jv[it - nv] * alpha;
      });
    }
    niter++;
    DIAL_MARK(w, 7);
  }
  w.items(nv, [&](int i) { s.warm[i] = s.qacc[i]; });
  DIAL_MARK(w, 8);
}

// Suffix sum of `src` (stride `st`, component k) up the exclusive tail of chain c into `dst` -- leaf first, as the plain loop
//   for (q = len - 1; q >= excl; q--) { acc += src[st * body_q + k]; dst[st * body_q + k] = acc; }
// does it, with a FIXED trip count for the compile-time instantiations: the chain's bodies come as two 32-bit words, every
// fetch is issued before the first addition (the table-driven loop paid two dependent LDS round trips per body).
// kFixedTrip: where it pays.  Measured on one box (profiles/r04_ab_fixed_trip.txt): Allegro -3.0 % (five roots, 22 bodies:
// the COM loop was its longest serial stretch); Go2 +2.5 %, H1 +3.4 %, H1 loco +6 % SLOWER -- those kernels hoist every
// lane-derived address out of the step loop and sit at the 168-VGPR budget: the unrolled fetches' addresses pushed 14 / 31 /
// 25 registers into scratch.  The generic-feature-set kernels re-derive addresses per step (wave.h: launder) and have room.
template <class M>
inline constexpr bool kFixedTrip = M::D::is_static && (M::D::ell || M::D::gen);
template <class M>
DIAL_DEV void chain_suffix_sum(const M* m, int c, int k, int st, const float* src, float* dst) {
  const int len = m->chain_len[c], ex = m->chain_excl[c];
  if constexpr (kFixedTrip<M>) {
    constexpr int CL = M::D::CHAINLEN;
    static_assert(CL == 8, "chain_body rows are read as two 32-bit words");
    const uint32_t* cw = reinterpret_cast<const uint32_t*>(m->chain_body[c]);
    const uint32_t w0 = cw[0], w1 = cw[1];
    int cb[CL];
    float v[CL];
#pragma unroll
    for (int q = 0; q < CL; q++) {
      const int b = (int)(((q < 4 ? w0 : w1) >> (8 * (q & 3))) & 255u);
      cb[q] = q < len ? b : 0;
    }
#pragma unroll
    for (int q = 0; q < CL; q++) v[q] = src[st * cb[q] + k];
    float acc = 0.f;
#pragma unroll
    for (int q = CL - 1; q >= 0; q--) {
      const bool on = q < len && q >= ex;
      const float t = acc + v[q];
      acc = on ? t : acc;
      if (on) dst[st * cb[q] + k] = acc;
    }
  } else {
    float acc = 0.f;
    for (int q = len - 1; q >= ex; q--) {
      const int b = m->chain_body[c][q];
      acc += src[st * b + k];
      dst[st * b + k] = acc;
    }
  }
}
// The bodies no chain tail covers (branching bodies and what lies above them), deepest first: dst[b] = own[b] + sum of dst[child]
template <class M>
DIAL_DEV void shared_subtree_sum(const M* m, int k, int st, const float* own, float* dst) {
  if constexpr (kFixedTrip<M>) {
#pragma unroll
    for (int sh = 0; sh < 4; sh++) {
      if (sh < m->nshared) {
        const int b = m->shared_body[sh], nch = m->shared_nchild[sh];
        const uint32_t cw = *reinterpret_cast<const uint32_t*>(m->shared_child[sh]);
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = dst[st * (int)(q < nch ? (cw >> (8 * q)) & 255u : 0u) + k];
        float acc = own[st * b + k];
#pragma unroll
        for (int q = 0; q < 4; q++) { const float t = acc + v[q]; acc = q < nch ? t : acc; }
        dst[st * b + k] = acc;
      }
    }
  } else {
    for (int sh = 0; sh < m->nshared; sh++) {
      const int b = m->shared_body[sh];
      float acc = own[st * b + k];
      for (int q = 0; q < m->shared_nchild[sh]; q++) acc += dst[st * m->shared_child[sh][q] + k];
      dst[st * b + k] = acc;
    }
  }
}

// collision_driver: contact c of the static list (narrow phases: plane-sphere / plane-capsule end; elliptic models also
// sphere-capsule and capsule-capsule; generic feature set also the box routines of box_collide.h)
template <class M>
DIAL_DEV void collide_contact(const M* m, const Ws& s, int c) {
  const int g1 = m->con_geom1[c], g2 = m->con_geom2[c];
  float n[3] = {s.gaxis[3 * g1], s.gaxis[3 * g1 + 1], s.gaxis[3 * g1 + 2]};
  float ctr[3] = {s.gpos[3 * g2], s.gpos[3 * g2 + 1], s.gpos[3 * g2 + 2]};
  float radius = m->geom_size[g2][0];
  float* fr = s.cframe + 9 * c;
  if constexpr (M::D::gen) {
    if (m->con_kind[c] >= DIAL_CON_PLANE_BOX) {   // box narrow phases (box_collide.h); geom2 is the box
      const auto box_of = [&](int g, BoxG& b) {
        const int bd = m->geom_bodyid[g];
        const float bq[4] = {s.xquat[4 * bd], s.xquat[4 * bd + 1], s.xquat[4 * bd + 2], s.xquat[4 * bd + 3]};
        const float gq[4] = {m->geom_quat[g][0], m->geom_quat[g][1], m->geom_quat[g][2], m->geom_quat[g][3]};
        dm::quat_mul(b.q, bq, gq);
        for (int k = 0; k < 3; k++) { b.c[k] = s.gpos[3 * g + k]; b.h[k] = m->geom_size[g][k]; }
      };
      BoxG b2;
      box_of(g2, b2);
      const float p1[3] = {s.gpos[3 * g1], s.gpos[3 * g1 + 1], s.gpos[3 * g1 + 2]};
      float dist, cp[3];
      if (m->con_kind[c] == DIAL_CON_PLANE_BOX) plane_box(n, p1, b2, m->con_sub[c], dist, cp, fr);
      else if (m->con_kind[c] == DIAL_CON_SPHERE_BOX) sphere_box(p1, m->geom_size[g1][0], b2, dist, cp, fr);
      else if (m->con_kind[c] == DIAL_CON_CAPSULE_BOX) capsule_box(p1, n, m->geom_size[g1][1], m->geom_size[g1][0], b2, m->con_sub[c], dist, cp, fr);
      else {
        BoxG b1;
        box_of(g1, b1);
        // the clipping polygons live in this candidate's slice of cdofdot | cacc | cfl (carved back to back, derived.h):
        // the velocity temporaries are dead by now -- cfrc, which this phase still reads, lies behind them (dial_create
        // checks that the slices fit)
        box_box(b1, b2, m->con_sub[c], dist, cp, fr, s.cdofdot + DIAL_BOX_POLY_WORDS * m->con_bbslot[c]);
      }
      s.cdist[c] = dist;
      for (int k = 0; k < 3; k++) s.cpos[3 * c + k] = cp[k];
      return;
    }
  }
  if constexpr (M::D::ell) {
    if (m->con_kind[c] == DIAL_CON_SPHERE_CAPSULE || m->con_kind[c] == DIAL_CON_CAPSULE_CAPSULE) {
      // MJX sphere_capsule / capsule_capsule: closest points on the capsule segment(s), then _sphere_sphere
      const float ax2[3] = {s.gaxis[3 * g2], s.gaxis[3 * g2 + 1], s.gaxis[3 * g2 + 2]}, hl2 = m->geom_size[g2][1];
      float b0[3], b1[3], p1[3], p2[3];
      for (int k = 0; k < 3; k++) { b0[k] = ctr[k] - ax2[k] * hl2; b1[k] = ctr[k] + ax2[k] * hl2; }
      if (m->con_kind[c] == DIAL_CON_SPHERE_CAPSULE) {
        for (int k = 0; k < 3; k++) p1[k] = s.gpos[3 * g1 + k];
        closest_segment_point(p2, b0, b1, p1);
      } else {
        const float hl1 = m->geom_size[g1][1];
        float a0[3], a1[3];
        for (int k = 0; k < 3; k++) { a0[k] = s.gpos[3 * g1 + k] - n[k] * hl1; a1[k] = s.gpos[3 * g1 + k] + n[k] * hl1; }
        closest_segment_to_segment(p1, p2, a0, a1, b0, b1);
      }
      const float r1 = m->geom_size[g1][0];
      float nn[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
      const float len = DM_SQRT(dm::dot3(nn, nn));
      if (len == 0.f) { nn[0] = 1.f; nn[1] = 0.f; nn[2] = 0.f; }
      else for (int k = 0; k < 3; k++) nn[k] /= len;
      const float dist = len - (r1 + radius);
      s.cdist[c] = dist;
      for (int k = 0; k < 3; k++) s.cpos[3 * c + k] = p1[k] + nn[k] * (r1 + dist * 0.5f);
      make_frame(fr, nn);
      s.con_on[c] = (dist - m->con_margin[c]) < 0.f ? 1.f : 0.f;
      return;
    }
  }
  if (m->con_kind[c] == DIAL_CON_PLANE_SPHERE) {
    make_frame(fr, n);   // collision_primitive make_frame(n)
  } else {
    float axis[3] = {s.gaxis[3 * g2], s.gaxis[3 * g2 + 1], s.gaxis[3 * g2 + 2]};
    float na = dm::dot3(n, axis), bb[3];
    for (int k = 0; k < 3; k++) bb[k] = axis[k] - n[k] * na;
    float bn = DM_SQRT(dm::dot3(bb, bb));
    if (bn < 0.5f) {
      bb[0] = 0.f; bb[1] = 0.f; bb[2] = 0.f;
      if (-0.5f < n[1] && n[1] < 0.5f) bb[1] = 1.f; else bb[2] = 1.f;
    } else {
      for (int k = 0; k < 3; k++) bb[k] /= bn;
    }
    float cc[3];
    dm::cross3(cc, n, bb);
    for (int k = 0; k < 3; k++) { fr[k] = n[k]; fr[3 + k] = bb[k]; fr[6 + k] = cc[k]; }
    float sgn = m->con_kind[c] == DIAL_CON_PLANE_CAPSULE_P ? 1.f : -1.f, hl = m->geom_size[g2][1];
    for (int k = 0; k < 3; k++) ctr[k] += sgn * axis[k] * hl;
  }
  float diff[3] = {ctr[0] - s.gpos[3 * g1], ctr[1] - s.gpos[3 * g1 + 1], ctr[2] - s.gpos[3 * g1 + 2]};
  float dist = dm::dot3(diff, n) - radius;
  s.cdist[c] = dist;
  for (int k = 0; k < 3; k++) s.cpos[3 * c + k] = ctr[k] - n[k] * (radius + 0.5f * dist);
  if constexpr (M::D::ell) s.con_on[c] = (dist - m->con_margin[c]) < 0.f ? 1.f : 0.f;
}

// The end of forward(): the generic feature set compacts the contacts that touch and, past its LDS cap, moves to the overflow
// area; then the constraint half.
template <class W, class M>
DIAL_DEV void forward_tail(W& w, const M* m, const Ws& s) {
  const int nv = dim_nv(m), nc = dim_nc(m), ne = dim_ne(m), nl = dim_nl(m);
  int nca = nc, nea = ne;   // contacts / rows the constraint section works on (generic instantiation: the touching ones)
  (void)nv; (void)nl;
  if constexpr (M::D::gen) {
    // compact the contacts that touch (see con_of above); with more than 64 candidates the list would need a second pass
    if (nc <= 64) {
      nca = w.compact(nc, [&](int c) { return s.cdist[c] - m->con_margin[c] < 0.f; }, s.clist);
    } else {
      w.items(nc, [&](int c) { s.clist[c] = (float)c; });
    }
    nea = nl + dim_nf(m) + 4 * nca;
  }
  if constexpr (M::D::gen) {
    // The LDS workspace of a rollout wavefront holds the Jacobian and the per-row arrays of at most s.con_cap touching
    // contacts (derived.h: ws_carve).  A sample that touches with more runs the SAME constraint code on its overflow
    // area in global memory (a second inlined copy: slower, bit-identical, rare) -- nothing is dropped.
    if (s.con_cap > 0 && nca > s.con_cap && s.ovf != nullptr) {   // (a capped workspace always comes with its overflow area)
      Ws sg = s;
      float* ovf = s.ovf;
#ifndef DIAL_EMU
      // (opaque to the optimiser: with the provenance of both workspaces in sight LLVM merges the two copies of the constraint
      //  code into one over pointer PHIs and trips over its own address-space inference -- "Illegal instruction detected:
      //  V_CMP_NE_U32 0, $src_shared_base" -- in the capacity-dimension kernel; the overflow copy simply uses flat accesses)
      asm volatile("" : "+v"(ovf));
#endif
      ws_overflow(sg, ovf, nv, nc, ne);
      forward_constraints(w, m, sg, nca, nea);
      return;
    }
  }
  forward_constraints(w, m, s, nca, nea);
}

// ================================================================ mjx.forward
template <class W, class M>
DIAL_DEV void forward(W& w, const M* m, const Ws& s) {
  const int nb = dim_nb(m), nv = dim_nv(m), nj = dim_nj(m), ng = dim_ng(m), nsite = dim_ns(m), nc = dim_nc(m);
  const int ne = dim_ne(m), nl = dim_nl(m), ntri = m->ntri;   // ntri: structurally non-zero entries of M / H
  int nca = nc, nea = ne;   // contacts / rows the constraint section works on (generic instantiation: the touching ones)

  if constexpr (W::half2) {   // two samples per wavefront (wave.h: WaveH): the quadruped stage on 32 lanes (smooth_quad2.h)
    static_assert(kQuadDims<typename M::D>, "the half-wave kernel exists for the Go2's own instantiation");
    forward_smooth_quad2(w, m, s);
    forward_constraints(w, m, s, nca, nea);
    return;
  } else
  if constexpr (kQuadDims<typename M::D>) {   // quadruped topology: the whole position / velocity stage in registers (smooth_quad.h)
    forward_smooth_quad(w, m, s);
    forward_constraints(w, m, s, nca, nea);
    return;
  }
  if constexpr (kQuadGenDims<typename M::D>) {   // the Go2's tree under the generic feature set (crate climb): bodies and dofs in registers
    forward_smooth_quad<false>(w, m, s);
    w.items(nc, [&](int c) { collide_contact(m, s, c); });
    forward_tail(w, m, s);
    return;
  }
  if constexpr (kRowsGenDims<typename M::D>) {   // the H1's tree under the generic feature set (push crate): the row layout + the solo crate
    forward_smooth_rows<true>(w, m, s);
    w.items(nc, [&](int c) { collide_contact(m, s, c); });
    forward_tail(w, m, s);
    return;
  }
  if constexpr (kRowsDims<typename M::D>) {   // one tree under a free root (H1): the same stage on the row layout (smooth_rows.h)
    forward_smooth_rows(w, m, s);
    {
      DIAL_LANE_SCOPE_IF(DIAL_ROWS_PHASE_SCOPE, w);   // (see forward_constraints)
      w.items(nc, [&](int c) { collide_contact(m, s, c); });
    }
    forward_constraints(w, m, s, nca, nea);
    return;
  }
  DIAL_MARK(w, 15);
  // ---- smooth.kinematics
  const bool kin_fast = m->kin_fast != 0;   // every body has at most one joint (wave-uniform)
  if (kin_fast) {
    // The transform of a body RELATIVE to its parent does not depend on the parent: with q_b = body_quat * q_joint,
    //   hinge:  l_q = q_b,        l_p = body_pos + R(body_quat) jnt_pos - R(q_b) jnt_pos
    //   slide:  l_q = body_quat,  l_p = body_pos + R(body_quat) jnt_axis * (q - q0)
    // (MJX composes the same maps body by body: anchor = pos + R jnt_pos, rotate, pos = anchor - R' jnt_pos).  Lane b
    // owns body b: it evaluates its joint's sin / cos and local transform into registers (all bodies at once), then
    // the level sweep is one fetch of the parent's pose, one quaternion product and one rotation per level -- not the
    // per-joint chain with its table look-ups at every level.
    vfloat L[9];   // l_q (4), l_p (3), depth, parent
    w.per_lane_n(L, [&](int b, float* o) {
      for (int k = 0; k < 9; k++) o[k] = 0.f;
      o[7] = -1.f;
      if (b == 0 || b >= nb) return;
      const int bflags = m->body_flags[b];
      float lq[4] = {m->body_quat[b][0], m->body_quat[b][1], m->body_quat[b][2], m->body_quat[b][3]};
      float lp[3] = {m->body_pos[b][0], m->body_pos[b][1], m->body_pos[b][2]};
      if (m->body_jntnum[b] == 1) {
        const int ji = m->body_jntadr[b], qa = m->jnt_qposadr[ji], type = m->jnt_type[ji];
        const float jp[3] = {m->jnt_pos[ji][0], m->jnt_pos[ji][1], m->jnt_pos[ji][2]};
        const float ja[3] = {m->jnt_axis[ji][0], m->jnt_axis[ji][1], m->jnt_axis[ji][2]};
        if (type == DIAL_JNT_FREE) {          // absolute pose: the sweep copies it
          for (int k = 0; k < 3; k++) lp[k] = s.qpos[qa + k];
          for (int k = 0; k < 4; k++) lq[k] = s.qpos[qa + 3 + k];
          dm::normalize4(lq);
          for (int k = 0; k < 4; k++) s.qpos[qa + 3 + k] = lq[k];
        } else if (type == DIAL_JNT_HINGE) {
          float qloc[4], qb[4], t0[3], t1[3];
          dm::axis_angle_to_quat(qloc, ja, s.qpos[qa] - m->qpos0[qa]);
          if (bflags & 1) { qb[0] = qloc[0]; qb[1] = qloc[1]; qb[2] = qloc[2]; qb[3] = qloc[3]; }
          else dm::quat_mul(qb, lq, qloc);
          if (!(bflags & 2)) {
            if (bflags & 1) { t0[0] = jp[0]; t0[1] = jp[1]; t0[2] = jp[2]; }
            else dm::rotate(t0, jp, lq);
            dm::rotate(t1, jp, qb);
            for (int k = 0; k < 3; k++) lp[k] += t0[k] - t1[k];
          }
          for (int k = 0; k < 4; k++) lq[k] = qb[k];
        } else {
          float ax[3];
          dm::rotate(ax, ja, lq);
          const float disp = s.qpos[qa] - m->qpos0[qa];
          for (int k = 0; k < 3; k++) lp[k] += ax[k] * disp;
        }
      }
      for (int k = 0; k < 4; k++) o[k] = lq[k];
      for (int k = 0; k < 3; k++) o[4 + k] = lp[k];
      o[7] = (bflags & 4) ? 0.f : (float)m->body_depth[b];   // free-joint bodies: "depth 0" = absolute, written below
      o[8] = (float)m->body_parent[b];
    });
    w.items(nb, [&](int b) {   // free-joint bodies (and nothing else) before the sweep
      if (b == 0 || lane_val(L[7], b) != 0.f) return;
      for (int k = 0; k < 3; k++) s.xpos[3 * b + k] = lane_val(L[4 + k], b);
      for (int k = 0; k < 4; k++) s.xquat[4 * b + k] = lane_val(L[k], b);
    });
    for (int d = 1; d <= m->nlevel; d++) {
      w.items(nb, [&](int b) {
        if (lane_val(L[7], b) != (float)d) return;
        const int p = (int)lane_val(L[8], b);
        const float lq[4] = {lane_val(L[0], b), lane_val(L[1], b), lane_val(L[2], b), lane_val(L[3], b)};
        const float lp[3] = {lane_val(L[4], b), lane_val(L[5], b), lane_val(L[6], b)};
        const float pq[4] = {s.xquat[4 * p], s.xquat[4 * p + 1], s.xquat[4 * p + 2], s.xquat[4 * p + 3]};
        float pos[3], quat[4];
        dm::rotate(pos, lp, pq);
        for (int k = 0; k < 3; k++) pos[k] += s.xpos[3 * p + k];
        dm::quat_mul(quat, pq, lq);
        for (int k = 0; k < 3; k++) s.xpos[3 * b + k] = pos[k];
        for (int k = 0; k < 4; k++) s.xquat[4 * b + k] = quat[k];
      });
    }
  } else
  // generic models: level-synchronous sweep over the body tree, joints applied one after the other
  for (int d = 1; d <= m->nlevel; d++) {
    const int b0 = m->lvl_start[d - 1];
    w.items(m->lvl_start[d] - b0, [&](int idx) {
      const int b = m->lvl_body[b0 + idx], p = m->body_parent[b];
      float pq[4] = {s.xquat[4 * p], s.xquat[4 * p + 1], s.xquat[4 * p + 2], s.xquat[4 * p + 3]};
      float bp[3] = {m->body_pos[b][0], m->body_pos[b][1], m->body_pos[b][2]};
      float bq[4] = {m->body_quat[b][0], m->body_quat[b][1], m->body_quat[b][2], m->body_quat[b][3]};
      float pos[3], quat[4];
      const int bflags = m->body_flags[b];
      dm::rotate(pos, bp, pq);
      for (int k = 0; k < 3; k++) pos[k] += s.xpos[3 * p + k];
      if (bflags & 1) { quat[0] = pq[0]; quat[1] = pq[1]; quat[2] = pq[2]; quat[3] = pq[3]; }   // identity body_quat
      else dm::quat_mul(quat, pq, bq);
      for (int ji = m->body_jntadr[b]; ji < m->body_jntadr[b] + m->body_jntnum[b]; ji++) {
        const int qa = m->jnt_qposadr[ji];
        float jp[3] = {m->jnt_pos[ji][0], m->jnt_pos[ji][1], m->jnt_pos[ji][2]};
        float ja[3] = {m->jnt_axis[ji][0], m->jnt_axis[ji][1], m->jnt_axis[ji][2]};
        if (m->jnt_type[ji] == DIAL_JNT_FREE) {
          for (int k = 0; k < 3; k++) { pos[k] = s.qpos[qa + k]; s.xanchor[3 * ji + k] = pos[k]; }
          s.xaxis[3 * ji] = 0.f; s.xaxis[3 * ji + 1] = 0.f; s.xaxis[3 * ji + 2] = 1.f;
          for (int k = 0; k < 4; k++) quat[k] = s.qpos[qa + 3 + k];
          dm::normalize4(quat);
          for (int k = 0; k < 4; k++) s.qpos[qa + 3 + k] = quat[k];
        } else {
          float anchor[3], axis[3];
          const bool at_origin = (bflags & 2) != 0;   // joint anchor at the body origin: anchor = pos, no offset
          if (at_origin) { anchor[0] = pos[0]; anchor[1] = pos[1]; anchor[2] = pos[2]; }
          else {
            dm::rotate(anchor, jp, quat);
            for (int k = 0; k < 3; k++) anchor[k] += pos[k];
          }
          dm::rotate(axis, ja, quat);
          for (int k = 0; k < 3; k++) { s.xanchor[3 * ji + k] = anchor[k]; s.xaxis[3 * ji + k] = axis[k]; }
          if (m->jnt_type[ji] == DIAL_JNT_HINGE) {
            float qloc[4], t3[3];
            dm::axis_angle_to_quat(qloc, ja, s.qpos[qa] - m->qpos0[qa]);
            dm::quat_mul(quat, quat, qloc);
            if (!at_origin) {
              dm::rotate(t3, jp, quat);
              for (int k = 0; k < 3; k++) pos[k] = anchor[k] - t3[k];
            }
          } else {
            float disp = s.qpos[qa] - m->qpos0[qa];
            for (int k = 0; k < 3; k++) pos[k] += axis[k] * disp;
          }
        }
      }
      for (int k = 0; k < 3; k++) s.xpos[3 * b + k] = pos[k];
      for (int k = 0; k < 4; k++) s.xquat[4 * b + k] = quat[k];
    });
  }
  // ---- local_to_global for inertial frames, geoms and sites
  DIAL_MARK(w, 0);
  w.items(nb + ng + nsite, [&](int it) {
    if (it < nb) {
      const int b = it;
      float q[4] = {s.xquat[4 * b], s.xquat[4 * b + 1], s.xquat[4 * b + 2], s.xquat[4 * b + 3]};
      float mat[9], t3[3], qi[4];   // (xmat is only needed for the free joint's cdof: formed there from xquat)
      float ip[3] = {m->body_ipos[b][0], m->body_ipos[b][1], m->body_ipos[b][2]};
      float iq[4] = {m->body_iquat[b][0], m->body_iquat[b][1], m->body_iquat[b][2], m->body_iquat[b][3]};
      dm::rotate(t3, ip, q);
      for (int k = 0; k < 3; k++) s.xipos[3 * b + k] = s.xpos[3 * b + k] + t3[k];
      dm::quat_mul(qi, q, iq);
      dm::quat_to_mat(mat, qi);
      for (int k = 0; k < 9; k++) s.ximat[9 * b + k] = mat[k];
    } else if (it < nb + ng) {
      const int g = it - nb, b = m->geom_bodyid[g];
      float q[4] = {s.xquat[4 * b], s.xquat[4 * b + 1], s.xquat[4 * b + 2], s.xquat[4 * b + 3]};
      float gp[3] = {m->geom_pos[g][0], m->geom_pos[g][1], m->geom_pos[g][2]};
      float gq[4] = {m->geom_quat[g][0], m->geom_quat[g][1], m->geom_quat[g][2], m->geom_quat[g][3]};
      float t3[3], qg[4], mat[9];
      dm::rotate(t3, gp, q);
      for (int k = 0; k < 3; k++) s.gpos[3 * g + k] = s.xpos[3 * b + k] + t3[k];
      dm::quat_mul(qg, q, gq);
      dm::quat_to_mat(mat, qg);
      s.gaxis[3 * g] = mat[2]; s.gaxis[3 * g + 1] = mat[5]; s.gaxis[3 * g + 2] = mat[8];
    } else {
      const int si = it - nb - ng, b = m->site_bodyid[si];
      float q[4] = {s.xquat[4 * b], s.xquat[4 * b + 1], s.xquat[4 * b + 2], s.xquat[4 * b + 3]};
      float sp[3] = {m->site_pos[si][0], m->site_pos[si][1], m->site_pos[si][2]}, t3[3];
      dm::rotate(t3, sp, q);
      for (int k = 0; k < 3; k++) s.spos[3 * si + k] = s.xpos[3 * b + k] + t3[k];
    }
  });
  DIAL_MARK(w, 16);
  // ---- smooth.com_pos: subtree COM of every kinematic-tree root (stored at the root's index)
  w.items(3 * nb, [&](int it) {
    const int b = it / 3, k = it - 3 * b;
    if (b == 0 || m->body_parent[b] != 0) return;
    float mp = 0.f, ms = 0.f;
    if constexpr (kFixedTrip<M>) {
      // fixed trip count, bodies outside the subtree masked: every fetch is issued up front.  (A loop whose length comes out
      // of a table pays one exposed LDS round trip per body; same additions in the same order.)
      const int e = m->body_subtree_end[b];
#pragma unroll
      for (int d = M::D::NB - 1; d >= 1; d--) {
        const bool on = d >= b && d < e;
        const float x = s.xipos[3 * d + k], mm = m->body_mass[d];
        const float t1 = mp + x * mm, t2 = ms + mm;
        mp = on ? t1 : mp;
        ms = on ? t2 : ms;
      }
    } else
    for (int d = m->body_subtree_end[b] - 1; d >= b; d--) {
      mp += s.xipos[3 * d + k] * m->body_mass[d];
      ms += m->body_mass[d];
    }
    s.com[3 * b + k] = ms < MJ_MINVAL ? s.xipos[3 * b + k] : mp / ms;
  });
  DIAL_MARK(w, 17);
  // ---- cinert (per body) and cdof (per joint)
  w.items(nb + nj, [&](int it) {
    if (it < nb) {
      const int b = it;
      float* ci = s.cinert + 10 * b;
      if (b == 0) { for (int k = 0; k < 10; k++) ci[k] = 0.f; return; }
      const float* c = s.com + 3 * m->body_rootid[b];
      const float* R = s.ximat + 9 * b;
      float off[3] = {s.xipos[3 * b] - c[0], s.xipos[3 * b + 1] - c[1], s.xipos[3 * b + 2] - c[2]};
      float mb = m->body_mass[b], oo = dm::dot3(off, off);
      float in0 = m->body_inertia[b][0], in1 = m->body_inertia[b][1], in2 = m->body_inertia[b][2];
      const int ii[6] = {0, 1, 2, 0, 0, 1}, jj[6] = {0, 1, 2, 1, 2, 2};
      for (int e = 0; e < 6; e++) {
        int i = ii[e], j = jj[e];
        float v = R[3 * i] * in0 * R[3 * j] + R[3 * i + 1] * in1 * R[3 * j + 1] + R[3 * i + 2] * in2 * R[3 * j + 2];
        float hh = (i == j ? oo : 0.f) - off[i] * off[j];
        ci[e] = v + hh * mb;
      }
      ci[6] = off[0] * mb; ci[7] = off[1] * mb; ci[8] = off[2] * mb; ci[9] = mb;
    } else {
      const int ji = it - nb, b = m->jnt_bodyid[ji], da = m->jnt_dofadr[ji];
      const float* c = s.com + 3 * m->body_rootid[b];
      float anchor[3], jaxis[3];
      if (kin_fast) {   // anchor = xpos + R(xquat) jnt_pos, axis = R(xquat) jnt_axis (a hinge leaves its own axis in place)
        const float bq[4] = {s.xquat[4 * b], s.xquat[4 * b + 1], s.xquat[4 * b + 2], s.xquat[4 * b + 3]};
        const float jp[3] = {m->jnt_pos[ji][0], m->jnt_pos[ji][1], m->jnt_pos[ji][2]};
        const float ja[3] = {m->jnt_axis[ji][0], m->jnt_axis[ji][1], m->jnt_axis[ji][2]};
        if ((m->body_flags[b] & 2) || m->jnt_type[ji] == DIAL_JNT_FREE) { anchor[0] = 0.f; anchor[1] = 0.f; anchor[2] = 0.f; }
        else dm::rotate(anchor, jp, bq);
        for (int k = 0; k < 3; k++) anchor[k] += s.xpos[3 * b + k];
        dm::rotate(jaxis, ja, bq);
      } else {
        for (int k = 0; k < 3; k++) { anchor[k] = s.xanchor[3 * ji + k]; jaxis[k] = s.xaxis[3 * ji + k]; }
      }
      float off[3] = {c[0] - anchor[0], c[1] - anchor[1], c[2] - anchor[2]};
      if (m->jnt_type[ji] == DIAL_JNT_FREE) {
        for (int i = 0; i < 3; i++)
          for (int k = 0; k < 6; k++) s.cdof[6 * (da + i) + k] = (k == 3 + i) ? 1.f : 0.f;
        float bq[4] = {s.xquat[4 * b], s.xquat[4 * b + 1], s.xquat[4 * b + 2], s.xquat[4 * b + 3]}, xm[9];
        dm::quat_to_mat(xm, bq);
        for (int i = 0; i < 3; i++) {
          float a[3] = {xm[i], xm[3 + i], xm[6 + i]}, cr[3];
          dm::cross3(cr, a, off);
          for (int k = 0; k < 3; k++) { s.cdof[6 * (da + 3 + i) + k] = a[k]; s.cdof[6 * (da + 3 + i) + 3 + k] = cr[k]; }
        }
      } else if (m->jnt_type[ji] == DIAL_JNT_HINGE) {
        float ax[3] = {jaxis[0], jaxis[1], jaxis[2]}, cr[3];
        dm::cross3(cr, ax, off);
        for (int k = 0; k < 3; k++) { s.cdof[6 * da + k] = ax[k]; s.cdof[6 * da + 3 + k] = cr[k]; }
      } else {
        for (int k = 0; k < 3; k++) { s.cdof[6 * da + k] = 0.f; s.cdof[6 * da + 3 + k] = jaxis[k]; }
      }
    }
  });
  DIAL_MARK(w, 18);
  // ---- smooth.com_vel + cdof_dot + the forward part of smooth.rne in ONE walk down every root-to-leaf chain
  if (m->nchain > 0 && kin_fast) {
    // cvel[b] = sum of cdof * qvel over the ancestor dofs, cdof_dot[i] = motion_cross(velocity accumulated BEFORE joint
    // i, cdof[i]), cacc[b] = [0, -g] + sum of cdof_dot * qvel: three dependent sweeps in MJX.  One lane per chain carries
    // the running velocity and acceleration in registers (cdof_dot is consumed on the spot and never stored); bodies
    // shared by several chains are written by each of them with the same value.
    w.items(m->nchain, [&](int c) {
      float vel[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      float acc[6] = {0.f, 0.f, 0.f, -m->gravity[0], -m->gravity[1], -m->gravity[2]};
      if (c == 0) for (int k = 0; k < 6; k++) { s.cvel[k] = vel[k]; s.cacc[k] = acc[k]; }   // world body
      const int len = m->chain_len[c];
      // the chain's tables first (two LDS round trips for the whole chain, not two per body: the stores below would
      // otherwise pin every look-up behind them)
      constexpr int CL = M::D::CHAINLEN;
      int cb[CL], cd0[CL], cnd[CL];
#pragma unroll
      for (int q = 0; q < CL; q++) cb[q] = q < len ? (int)m->chain_body[c][q] : 0;
#pragma unroll
      for (int q = 0; q < CL; q++) { cd0[q] = m->body_dofadr[cb[q]]; cnd[q] = m->body_dofnum[cb[q]]; }
#pragma unroll
      for (int q = 0; q < CL; q++) {
        if (q >= len) break;
        const int b = cb[q], d0 = cd0[q], nd = cnd[q];
        if (nd == 6) {   // free joint: the translational dofs have cdof_dot = 0, the rotational ones see the velocity after them
          float vs[6];
          for (int k = 0; k < 6; k++) vs[k] = vel[k];
          for (int j = 0; j < 3; j++) {   // cdof of translational dof j is the unit vector e_(3+j) (written as such above)
            const float qv = s.qvel[d0 + j];
            vs[3 + j] += qv;
            vel[3 + j] += qv;
          }
          for (int j = 3; j < 6; j++) {
            const float qv = s.qvel[d0 + j];
            float cd[6], cdd[6];
            for (int k = 0; k < 6; k++) cd[k] = s.cdof[6 * (d0 + j) + k];
            dm::motion_cross(cdd, vs, cd);
            for (int k = 0; k < 6; k++) { acc[k] += cdd[k] * qv; vel[k] += cd[k] * qv; }
          }
        } else {
          float vpar[6];
          for (int k = 0; k < 6; k++) vpar[k] = vel[k];
          for (int j = 0; j < nd; j++) {
            const float qv = s.qvel[d0 + j];
            float cd[6], cdd[6];
            for (int k = 0; k < 6; k++) cd[k] = s.cdof[6 * (d0 + j) + k];
            dm::motion_cross(cdd, vpar, cd);
            for (int k = 0; k < 6; k++) { acc[k] += cdd[k] * qv; vel[k] += cd[k] * qv; }
          }
        }
        for (int k = 0; k < 6; k++) { s.cvel[6 * b + k] = vel[k]; s.cacc[6 * b + k] = acc[k]; }
      }
    });
    DIAL_MARK(w, 19);
  } else {
  // cvel[b] = sum over ancestor dofs (root first), no recursion needed
  if (m->nchain > 0) {
    // prefix sums down every root-to-leaf chain: item = (chain, component); bodies shared by several chains
    // are written by each of them with the same value
    w.items(6 * m->nchain, [&](int it) {
      const int c = it / 6, k = it - 6 * c;
      float acc = 0.f;
      if (c == 0) s.cvel[k] = 0.f;   // world body
      for (int q = 0; q < m->chain_len[c]; q++) {
        const int b = m->chain_body[c][q];
        for (int i = m->body_dofadr[b]; i < m->body_dofadr[b] + m->body_dofnum[b]; i++) acc += s.cdof[6 * i + k] * s.qvel[i];
        s.cvel[6 * b + k] = acc;
      }
    });
  } else {
    w.items(6 * nb, [&](int it) {
      const int b = it / 6, k = it - 6 * b;
      float acc = 0.f;
      const int na = m->body_nanc[b];
      for (int a = 0; a < na; a++) {
        const int i = m->body_anc[b][a];
        acc += s.cdof[6 * i + k] * s.qvel[i];
      }
      s.cvel[6 * b + k] = acc;
    });
  }
  DIAL_MARK(w, 19);
  // ---- cdof_dot (per dof): motion_cross(velocity accumulated BEFORE this joint, cdof)
  w.items(nv, [&](int i) {
    const int ji = m->dof_jntid[i], b = m->dof_bodyid[i], p = m->body_parent[b], da = m->jnt_dofadr[ji];
    float* out = s.cdofdot + 6 * i;
    if (m->jnt_type[ji] == DIAL_JNT_FREE && i < da + 3) { for (int k = 0; k < 6; k++) out[k] = 0.f; return; }
    float vs[6];
    for (int k = 0; k < 6; k++) vs[k] = s.cvel[6 * p + k];
    const int jend = (m->jnt_type[ji] == DIAL_JNT_FREE) ? da + 3 : da;
    for (int j = m->body_dofadr[b]; j < jend; j++)
      for (int k = 0; k < 6; k++) vs[k] += s.cdof[6 * j + k] * s.qvel[j];
    float cd[6];
    for (int k = 0; k < 6; k++) cd[k] = s.cdof[6 * i + k];
    dm::motion_cross(out, vs, cd);
  });
  DIAL_MARK(w, 20);
  // ---- smooth.rne forward part: cacc[b] = [0,-g] + sum over ancestor dofs cdof_dot*qvel
  if (m->nchain > 0) {
    w.items(6 * m->nchain, [&](int it) {
      const int c = it / 6, k = it - 6 * c;
      float acc = k >= 3 ? -m->gravity[k - 3] : 0.f;
      if (c == 0) s.cacc[k] = acc;   // world body
      for (int q = 0; q < m->chain_len[c]; q++) {
        const int b = m->chain_body[c][q];
        for (int i = m->body_dofadr[b]; i < m->body_dofadr[b] + m->body_dofnum[b]; i++) acc += s.cdofdot[6 * i + k] * s.qvel[i];
        s.cacc[6 * b + k] = acc;
      }
    });
  } else {
    w.items(6 * nb, [&](int it) {
      const int b = it / 6, k = it - 6 * b;
      float acc = k >= 3 ? -m->gravity[k - 3] : 0.f;
      const int na = m->body_nanc[b];
      for (int a = 0; a < na; a++) {
        const int i = m->body_anc[b][a];
        acc += s.cdofdot[6 * i + k] * s.qvel[i];
      }
      s.cacc[6 * b + k] = acc;
    });
  }
  }
  DIAL_MARK(w, 21);
  // ---- smooth.crb composite inertias (subtree sums) | rne local body forces
  if (m->nshared >= 0) {
    // subtree sums as suffix sums: (chain, component) items walk the exclusive tail of their chain leaf -> root
    // (<= chain length iterations instead of one subtree-sized loop per body); the branching bodies follow below
    w.items(10 * m->nchain + nb, [&](int it) {
      if (it < 10 * m->nchain) {
        const int c = it / 10, k = it - 10 * c;
        chain_suffix_sum(m, c, k, 10, s.cinert, s.crb);
      } else {
        const int b = it - 10 * m->nchain;
        float ci[10], ca[6], cv[6], f1[6], f2[6], f3[6];
        for (int k = 0; k < 10; k++) ci[k] = s.cinert[10 * b + k];
        for (int k = 0; k < 6; k++) { ca[k] = s.cacc[6 * b + k]; cv[k] = s.cvel[6 * b + k]; }
        dm::inert_mul(f1, ci, ca);
        dm::inert_mul(f2, ci, cv);
        dm::motion_cross_force(f3, cv, f2);
        for (int k = 0; k < 6; k++) s.cfl[6 * b + k] = f1[k] + f3[k];
      }
    });
    w.items(10 + 6 * m->nchain, [&](int it) {
      if (it < 10) {          // crb of the branching bodies, deepest first (one lane per component: no cross-lane order)
        const int k = it;
        s.crb[k] = 0.f;       // world body
        shared_subtree_sum(m, k, 10, s.cinert, s.crb);
      } else {                // cfrc suffix sums up the exclusive chain tails
        const int c = (it - 10) / 6, k = (it - 10) - 6 * c;
        chain_suffix_sum(m, c, k, 6, s.cfl, s.cfrc);
      }
    });
    DIAL_MARK(w, 22);
    w.items(nv + 6, [&](int it) {
      if (it < nv) {
        const int i = it, b = m->dof_bodyid[i];
        float ci[10], cd[6], f[6];
        for (int k = 0; k < 10; k++) ci[k] = s.crb[10 * b + k];
        for (int k = 0; k < 6; k++) cd[k] = s.cdof[6 * i + k];
        dm::inert_mul(f, ci, cd);
        for (int k = 0; k < 6; k++) s.Fd[6 * i + k] = f[k];
      } else {                // cfrc of the branching bodies
        const int k = it - nv;
        shared_subtree_sum(m, k, 6, s.cfl, s.cfrc);
      }
    });
    DIAL_MARK(w, 23);
  } else {
  w.items(11 * nb, [&](int it) {
    if (it < 10 * nb) {
      const int b = it / 10, k = it - 10 * b;
      float acc = 0.f;
      if (b > 0)
        for (int d = m->body_subtree_end[b] - 1; d >= b; d--) acc += s.cinert[10 * d + k];
      s.crb[10 * b + k] = acc;
    } else {
      const int b = it - 10 * nb;
      float ci[10], ca[6], cv[6], f1[6], f2[6], f3[6];
      for (int k = 0; k < 10; k++) ci[k] = s.cinert[10 * b + k];
      for (int k = 0; k < 6; k++) { ca[k] = s.cacc[6 * b + k]; cv[k] = s.cvel[6 * b + k]; }
      dm::inert_mul(f1, ci, ca);
      dm::inert_mul(f2, ci, cv);
      dm::motion_cross_force(f3, cv, f2);
      for (int k = 0; k < 6; k++) s.cfl[6 * b + k] = f1[k] + f3[k];
    }
  });
  DIAL_MARK(w, 22);
  // ---- F_i = crb[body_i] * cdof_i | cfrc subtree sums (rne backward part)
  w.items(nv + 6 * nb, [&](int it) {
    if (it < nv) {
      const int i = it, b = m->dof_bodyid[i];
      float ci[10], cd[6], f[6];
      for (int k = 0; k < 10; k++) ci[k] = s.crb[10 * b + k];
      for (int k = 0; k < 6; k++) cd[k] = s.cdof[6 * i + k];
      dm::inert_mul(f, ci, cd);
      for (int k = 0; k < 6; k++) s.Fd[6 * i + k] = f[k];
    } else {
      const int b = (it - nv) / 6, k = (it - nv) - 6 * b;
      float acc = 0.f;
      for (int d = m->body_subtree_end[b] - 1; d >= b; d--) acc += s.cfl[6 * d + k];
      s.cfrc[6 * b + k] = acc;
    }
  });
  DIAL_MARK(w, 23);
  }
  if constexpr (M::D::gen) w.items((nv * (nv + 1)) / 2, [&](int e) { s.M[e] = 0.f; });
  // ---- M (lower triangle, support.make_m) | qfrc_smooth = passive - bias + actuator
  //      | collision_driver (static contact list)
  w.items(ntri + nv + nc, [&](int it) {
    if (it < ntri) {
      const int i = m->tri[it] >> 8, j = m->tri[it] & 0xff;
      float v = 0.f;
      if ((m->dof_ancmask[i] >> j) & 1u) {
        for (int k = 0; k < 6; k++) v += s.Fd[6 * i + k] * s.cdof[6 * j + k];
      }
      if (i == j) v += m->dof_armature[i];
      if constexpr (M::D::square) { s.M[i * M::D::S + j] = v; s.M[j * M::D::S + i] = v; }
      else s.M[tri_idx(i, j)] = v;
    } else if (it < ntri + nv) {
      const int i = it - ntri, b = m->dof_bodyid[i];
      float bias = 0.f;
      for (int k = 0; k < 6; k++) bias += s.cdof[6 * i + k] * s.cfrc[6 * b + k];
      float passive = -m->dof_damping[i] * s.qvel[i];
      float actf = 0.f;
      const int a = m->dof_act[i];
      if (a >= 0) {
        float c = s.ctrl[a];
        if (m->act_ctrllimited[a]) c = dm::clip(c, m->act_ctrlrange[a][0], m->act_ctrlrange[a][1]);
        float force = m->act_isposition[a] ? m->act_kp[a] * (c - s.qpos[m->act_qposadr[a]]) : c;
        actf = m->act_gear[a] * force;
      }
      float qf = passive - bias + actf;
      s.qfs[i] = qf;
      s.rhs[i] = qf;
    } else {
      collide_contact(m, s, it - ntri - nv);
    }
  });
  forward_tail(w, m, s);
}

// Task kinds a kernel instantiation can be asked to run (the dimension-specialised ones are per robot; dial_create checks).
template <class D>
constexpr uint32_t task_kind_mask() {
  if (std::is_same<D, DimsGo2Crate>::value) return 1u << DIAL_TASK_GO2_CRATE;   // (dispatched ahead of the reward phase)
  if (std::is_same<D, DimsH1PushCrate>::value) return 1u << DIAL_TASK_H1_PUSH_CRATE;
  if (std::is_same<typename D::Topo, TopoGo2>::value) return (1u << DIAL_TASK_GO2_WALK) | (1u << DIAL_TASK_GO2_SEQ_JUMP);
  if (std::is_same<typename D::Topo, TopoH1>::value) return 1u << DIAL_TASK_H1_WALK;
  if (std::is_same<typename D::Topo, TopoH1Loco>::value) return 1u << DIAL_TASK_H1_LOCO;
  return (1u << DIAL_TASK_GO2_WALK) | (1u << DIAL_TASK_GO2_SEQ_JUMP) | (1u << DIAL_TASK_H1_WALK) | (1u << DIAL_TASK_H1_LOCO) |
         (D::gen ? (1u << DIAL_TASK_H1_PUSH_CRATE) | (1u << DIAL_TASK_GO2_CRATE) : 0u);
}

// Velocity command of one env.step (unitree_go2_env.py:142-155, unitree_h1_env.py:199-212): component k < 3 of the linear,
// k - 3 of the angular command.  With randomize_tasks the command of a step whose (pre-increment) index is a multiple of
// 500 is the episode's entry of dial_task::cmd_table -- and ONLY of that step: upstream does not store the sampled command,
// every other step computes from the default again.  The table stays in the global dial_task (read once per 500 steps).
template <class M>
DIAL_DEV void step_cmd(const M* m, const dial_task* tg, float step, float* cmd /* [vx vy vz | wx wy wz] */) {
  for (int k = 0; k < 3; k++) { cmd[k] = m->cmd_vel[k]; cmd[3 + k] = m->cmd_ang_vel[k]; }
  if (m->randomize_tasks) {
    const int is = (int)step;
    if (is % 500 == 0) {
      const int e = (is / 500) % m->n_cmd;
      cmd[0] = tg->cmd_table[e][0]; cmd[1] = tg->cmd_table[e][1]; cmd[2] = 0.f;
      cmd[3] = 0.f; cmd[4] = 0.f; cmd[5] = tg->cmd_table[e][2];
    }
  }
}

// Push-crate task constants: they exist in the generic instantiation's constants only (cmodel.h: CModelGeneric); function
// templates, so that code naming them also compiles (and is discarded) for the dimension-specialised instantiations.
template <class M>
DIAL_DEV int pc_foot_contact(const M* m, int f, int k) {
  if constexpr (!M::D::gen) return 0;
  else return m->pc_foot_contact[f][k];
}
template <class M>
DIAL_DEV float pc_contact_reward(const M* m, const Ws& s) {
  if constexpr (!M::D::gen) return 0.f;
  else {
    // unitree_h1_env.py:525-531: +1 per hand on the crate (contact point below 1.1 m), -1 per other part touching it
    float rc = 0.f;
    for (int q = 0; q < 2; q++) {
      const int cc = m->pc_wanted[q];
      rc += (s.cdist[cc] < 1e-3f && s.cpos[3 * cc + 2] < m->pc_wanted_zmax) ? 1.f : 0.f;
    }
    for (int q = 0; q < m->pc_n_unwanted; q++) rc -= s.cdist[m->pc_unwanted[q]] < 1e-3f ? 1.f : 0.f;
    return rc;
  }
}

// ================================================================ forward.euler (eulerdamp disabled)
template <class W, class M>
DIAL_DEV void euler(W& w, const M* m, const Ws& s) {
  const float dt = m->timestep;
  if constexpr (M::D::ell) {
    if (m->eulerdamp) {
      // implicit joint damping (forward.euler): qacc <- (M + dt diag(damping))^-1 (qfrc_smooth + qfrc_constraint);
      // qacc_warmstart keeps the solver's solution.  M + dt B has M's sparsity: the tree elimination order applies.
      constexpr int NV = M::D::NV, S = M::D::S;
      // M + dt B: 16-byte copies, the damping on the chunk that holds the row's diagonal (round 6: was one ELEMENT per item, nine
      // passes of the wavefront over the 22 x 24 square in every physics sub-step; cfg 4: -1.5 % per iteration, profiles/r06_ab_allegro_issue_bound.txt)
      w.items(NV * S / 4, [&](int e) {
        const int i = (4 * e) / S, j0 = 4 * e - i * S;
        float a = s.M[4 * e], b = s.M[4 * e + 1], c2 = s.M[4 * e + 2], d2 = s.M[4 * e + 3];
        const float add = dt * m->dof_damping[i];
        const int dj = i - j0;   // position of the diagonal inside the chunk, if 0 .. 3
        a += dj == 0 ? add : 0.f; b += dj == 1 ? add : 0.f; c2 += dj == 2 ? add : 0.f; d2 += dj == 3 ? add : 0.f;
        store4(s.H + 4 * e, a, b, c2, d2);
      });
      const vfloat rhs = w.per_lane([&](int l) { return l < NV ? s.qfs[l] + s.qfc[l] : 0.f; });
      const vfloat x = reg_chol<typename M::D>(w, m, s.H, rhs, s.H);
      w.items(NV, [&](int i) { s.qvel[i] += lane_val(x, i) * dt; });
    } else {
      w.items(dim_nv(m), [&](int i) { s.qvel[i] += s.qacc[i] * dt; });
    }
  } else if constexpr (M::D::pre_ctrl) {
    // (Dims::pre_ctrl rollouts: the step's qd row is stored from here -- Wave::out_io --, one element per dof lane; the q row by the
    //  reward phase's idle lanes: the free joint's lane below is the long pole of this stage and should not issue seven stores)
    float* qdrow = nullptr;
    if (w.out_io && w.out_io->qdss) qdrow = w.out_io->qdss + (size_t)w.out_row * dim_nv(m);
    w.items(dim_nv(m), [&](int i) {
      const float v = s.qvel[i] + s.qacc[i] * dt;
      s.qvel[i] = v;
      if (qdrow) qdrow[i] = v;
    });
  } else
  w.items(dim_nv(m), [&](int i) { s.qvel[i] += s.qacc[i] * dt; });
  w.items(dim_nj(m), [&](int ji) {
    const int qa = m->jnt_qposadr[ji], da = m->jnt_dofadr[ji];
    if (m->jnt_type[ji] == DIAL_JNT_FREE) {
      for (int k = 0; k < 3; k++) s.qpos[qa + k] += dt * s.qvel[da + k];
      float v[3] = {s.qvel[da + 3], s.qvel[da + 4], s.qvel[da + 5]};
      // An angular velocity whose SQUARE is below the smallest normal fp32 (|w| < 1.1e-19 rad/s: a body that has come to rest) is
      // no rotation: under the fast-math flags v / sqrt(d) becomes v * v_rsq(d), and v_rsq flushes a denormal d to 0 -> inf -> NaN
      // (the Allegro's ball lying still on the floor, tests/test_gpu_parity.py closed loop; DESIGN.md deliberate deviations)
      const float w2 = dm::dot3(v, v);
      float nrm = w2 >= DM_FLT_MIN ? DM_SQRT(w2) : 0.f, axis[3] = {1.f, 0.f, 0.f}, qr[4], qn[4];
      if (nrm > 0.f) { axis[0] = v[0] / nrm; axis[1] = v[1] / nrm; axis[2] = v[2] / nrm; }
      dm::axis_angle_to_quat(qr, axis, dt * nrm);
      float q0[4] = {s.qpos[qa + 3], s.qpos[qa + 4], s.qpos[qa + 5], s.qpos[qa + 6]};
      dm::quat_mul(qn, q0, qr);
      dm::normalize4(qn);
      for (int k = 0; k < 4; k++) s.qpos[qa + 3 + k] = qn[k];
    } else {
      s.qpos[qa] += dt * s.qvel[da];
    }
  });
}

// ================================================================ env.step
DIAL_DEV float foot_step_height(float tt, float footphase, float duty) {
  const float two_pi = 2.f * DIAL_PI;
  float x = tt + DIAL_PI - footphase;
  float angle = x - two_pi * DM_FLOOR(x / two_pi) - DIAL_PI;
  if (duty < 1.f) angle = angle * 0.5f / (1.f - duty);
  float clipped = dm::clip(angle, -DIAL_PI / 2.f, DIAL_PI / 2.f);
  float value = duty < 1.f ? DM_COS(clipped) : 0.f;
  return dm::absf(value) >= 1e-6f ? dm::absf(value) : 0.f;
}
DIAL_DEV float quat_yaw(const float* q) {
  return DM_ATAN2(-2.f * q[1] * q[2] + 2.f * q[0] * q[3], q[1] * q[1] + q[0] * q[0] - q[3] * q[3] - q[2] * q[2]);
}

// One env.step from the action in s.act (nu values).  Returns the (wave-uniform) reward.
// The scalar reward terms are independent of each other, so they are spread over lanes (one term per
// lane) and summed by one lane in the reference's order afterwards: the critical path is the longest term
// (atan2 / sin / cos), not their sum.
// FULL_INFO = false (rollouts): the write-only info fields (done, feet_air_time, last_contact) are not
// maintained -- nothing reads them inside a rollout (done never terminates one).
// BaseEnv.act2joint (base_env.py:38-49): normalised action -> joint target, clipped to the physical range
// (joint_offset: the keyframe pose AllegroReorientEnv.act2joint adds, manipulation.py:107-109; 0 elsewhere)
template <class M>
DIAL_DEV float act2joint(const M* m, float act, int a) {
  float an = (act * m->action_scale + 1.0f) / 2.0f;
  float jt = (m->joint_range[a][0] + m->joint_offset[a]) + an * (m->joint_range[a][1] - m->joint_range[a][0]);
  return dm::clip(jt, m->phys_range[a][0], m->phys_range[a][1]);
}
// get_foot_step for foot f at the (pre-increment) step counter `step` (function_utils.py:18-43)
template <class M>
DIAL_DEV float gait_ztar(const M* m, int f, float step) {
  return m->gait_amp * foot_step_height(step * m->dt * 2.f * DIAL_PI * m->gait_cadence + DIAL_PI, 2.f * DIAL_PI * m->gait_phase[f], m->gait_duty);
}
// PRE (rollouts of the Dims::pre_ctrl instantiations): the joint targets and the gait clock of control step `st` come from the
// rollout's tables (Ws::jtab / ztab, built in its prologue: rollout_driver.h), and the step's outputs are stored by the phases that
// produce them (Wave::out_io / out_row: x.pos by the position stage, q / qd by the integrator, the reward by its lane).
template <bool FULL_INFO, bool PRE = false, class W, class M>
DIAL_DEV float env_step(W& w, const M* m, const dial_task* tg, const Ws& s, int st = 0) {
  const int nu = dim_nu(m);
  const bool walk = m->kind == DIAL_TASK_GO2_WALK || m->kind == DIAL_TASK_H1_WALK || m->kind == DIAL_TASK_H1_LOCO ||
                    m->kind == DIAL_TASK_H1_PUSH_CRATE;
  static_assert(!PRE || M::D::pre_ctrl, "control tables: carved for the Dims::pre_ctrl instantiations only");
  // act2joint / act2tau (base_env.py:38-66) | desired foot heights from the gait clock (get_foot_step)
  if constexpr (PRE) {
    w.jrow = s.jtab + st * nu;   // act2tau runs in the position stage's actuation lanes (smooth_quad.h: MO)
  } else
  w.items(nu + DIAL_MAX_FEET, [&](int it) {
    if (it < nu) {
      const int a = it;
      const float jt = act2joint(m, s.act[a], a);
      float c;
      if (m->position_control) c = jt;
      else {
        float q_err = jt - s.qpos[7 + a];
        c = dm::clip(m->kp[a] * q_err - m->kd[a] * s.qvel[6 + a], m->tau_range[a][0], m->tau_range[a][1]);
      }
      s.ctrl[a] = c;
    } else {
      const int f = it - nu;
      if (walk && f < m->nfeet) s.ztar[f] = gait_ztar(m, f, s.info[DIAL_INFO_STEP]);
    }
  });
  DIAL_MARK(w, 25);
  const int n_frames = PRE ? 1 : m->n_frames;   // (Dims::pre_ctrl: one physics step per control step, dial_create checks)
  for (int f = 0; f < n_frames; f++) {  // pipeline_step
#ifndef DIAL_EMU
    if (w.launder) { asm volatile("" : "+v"(w.lane)); w.lane_r = w.lane; }   // see rollout_driver.h: the step loop
#endif
    forward(w, m, s);
    euler(w, m, s);
    DIAL_MARK(w, 9);
  }
  if (m->kind == DIAL_TASK_ALLEGRO) {
    // manipulation.py:75-100 (torso_x = the object body): three sums of squares, one lane each
    w.items(3, [&](int it) {
      const int ob = m->torso_x + 1;
      float acc = 0.f;
      if (it == 0) {
        for (int k = 0; k < 3; k++) { const float e = s.cvel[6 * ob + k] * DIAL_PI / 180.0f - s.info[DIAL_INFO_ANG_VEL_TAR + k]; acc += e * e; }
      } else if (it == 1) {
        for (int k = 0; k < 3; k++) { const float e = s.xpos[3 * ob + k] - s.info[DIAL_INFO_POS_TAR + k]; acc += e * e; }
      } else {
        for (int a = 0; a < nu; a++) { const float e = s.qpos[7 + a] - m->joint_offset[a]; acc += e * e; }
      }
      s.rpart[it] = acc;
    });
    w.items(1, [&](int) {
      const float step = s.info[DIAL_INFO_STEP];
      const float reward = -s.rpart[0] * 1.0f + -s.rpart[1] * 5.0f + -s.rpart[2] * 0.1f;
      if (FULL_INFO) s.info[DIAL_INFO_DONE] = step >= 100.f ? 1.f : 0.f;
      s.info[DIAL_INFO_STEP] = step + 1.f;
      s.info[DIAL_INFO_REWARD] = reward;
    });
    DIAL_MARK(w, 10);
    return s.info[DIAL_INFO_REWARD];
  }
  if constexpr (M::D::gen && ((task_kind_mask<typename M::D>() >> DIAL_TASK_GO2_CRATE) & 1u)) {
    if (m->kind == DIAL_TASK_GO2_CRATE) {
      // UnitreeGo2CrateEnv.step (unitree_go2_env.py:679-795).  Of its eleven terms only four carry a non-zero weight:
      // head position (:711-719), upright (:720-723), yaw (:724-727) and the feet-on-the-crate count (:741-766); the
      // others are multiplied by 0.0 and add an exact zero.  done = 0; vel_tar / yaw_tar are not updated by the step.
      w.items(1, [&](int) {
        float* info = s.info;
        const int tb = m->torso_x + 1, ub = m->upright_x + 1;
        const float step = info[DIAL_INFO_STEP], dt = m->dt;
        const float tq[4] = {s.xquat[4 * tb], s.xquat[4 * tb + 1], s.xquat[4 * tb + 2], s.xquat[4 * tb + 3]};
        const float uq[4] = {s.xquat[4 * ub], s.xquat[4 * ub + 1], s.xquat[4 * ub + 2], s.xquat[4 * ub + 3]};
        const float hv[3] = {m->head_vec[0], m->head_vec[1], m->head_vec[2]};
        float mat[9], head[3];
        dm::quat_to_mat(mat, tq);   // head_pos = pos + R head_vec (math.quat_to_3x3, jnp.dot)
        float reward_pos = 0.f;
        for (int k = 0; k < 3; k++) {
          head[k] = s.xpos[3 * tb + k] + (mat[3 * k] * hv[0] + mat[3 * k + 1] * hv[1] + mat[3 * k + 2] * hv[2]);
          const float e = head[k] - (info[DIAL_INFO_POS_TAR + k] + info[DIAL_INFO_VEL_TAR + k] * dt * step);
          reward_pos += e * e;
        }
        const float up[3] = {0.f, 0.f, 1.f};
        float vec[3];
        dm::rotate(vec, up, uq);
        const float reward_upright = -((vec[0] - 0.f) * (vec[0] - 0.f) + (vec[1] - 0.f) * (vec[1] - 0.f) + (vec[2] - 1.f) * (vec[2] - 1.f));
        const float ey = quat_yaw(tq) - info[DIAL_INFO_YAW_TAR];
        float reward_contact = 0.f;
        for (int i = 0; i < 4; i++) {
          const float* cp = s.cpos + 3 * m->crate_contact[i];
          const bool cond = cp[0] > m->crate_region[0] && cp[0] < m->crate_region[1] && cp[1] > m->crate_region[2] &&
                            cp[1] < m->crate_region[3] && cp[2] > m->crate_region[4] && cp[2] < m->crate_region[5];
          reward_contact += cond ? 1.f : 0.f;
        }
        const float reward = -reward_pos * 1.0f + reward_upright * 0.01f + -(ey * ey) * 0.3f + reward_contact * 0.02f;
        if (FULL_INFO) info[DIAL_INFO_DONE] = 0.f;
        info[DIAL_INFO_STEP] = step + 1.f;
        info[DIAL_INFO_REWARD] = reward;
      });
      DIAL_MARK(w, 10);
      return s.info[DIAL_INFO_REWARD];
    }
  }
  // ---- reward terms (all read the PRE-integration forward quantities)
  //   0 gaits | contact   1 upright   2 yaw   3 vel (walk) | pos (jump)   4 ang_vel | penalty   5 height   6 energy   7 done
  // The terms are independent scalar chains.  One term per lane sounds parallel but is not: lanes that take different
  // branches are SERIALISED by the SIMD, each chain paying the ~10-cycle dependent-issue latency on its own.  Instead ONE
  // lane evaluates all of them in straight-line code (term index = compile-time constant): the scheduler interleaves the
  // independent chains, which then issue back to back.  For that the whole phase has to be ONE basic block: the task kind
  // is therefore a compile-time constant inside (`reward_phase(KIND)`, dispatched once, wave-uniformly, below) -- with the
  // kind tested at run time inside every term, each term was its own chain of blocks and the ~400 instructions issued one
  // dependent instruction after the other (4.7 k cycles per step, round 3 section profile).
  auto reward_phase = [&](auto KIND) {
    constexpr int kind = decltype(KIND)::value;
    constexpr bool walk = kind == DIAL_TASK_GO2_WALK || kind == DIAL_TASK_H1_WALK || kind == DIAL_TASK_H1_LOCO ||
                          kind == DIAL_TASK_H1_PUSH_CRATE;
    constexpr int NF = (kind == DIAL_TASK_GO2_WALK || kind == DIAL_TASK_GO2_SEQ_JUMP) ? 4 : 2;   // feet (dial_create checks task.nfeet)
    auto term = [&](auto IT, const float* cmd) -> float {
      constexpr int it = decltype(IT)::value;
      const float dt = m->dt;
      const int tb = m->torso_x + 1, ub = m->upright_x + 1;
      float* info = s.info;
      const float step = info[DIAL_INFO_STEP];
      // torso / upright-body state, fetched once up front by every term lane (same addresses: LDS broadcasts, one
      // round trip) instead of inside the divergent branches, where each term would pay its own dependent fetch
      const float tq[4] = {s.xquat[4 * tb], s.xquat[4 * tb + 1], s.xquat[4 * tb + 2], s.xquat[4 * tb + 3]};
      const float uq[4] = {s.xquat[4 * ub], s.xquat[4 * ub + 1], s.xquat[4 * ub + 2], s.xquat[4 * ub + 3]};
      const float tp[3] = {s.xpos[3 * tb], s.xpos[3 * tb + 1], s.xpos[3 * tb + 2]};
      const float* cmr = s.com + 3 * m->body_rootid[tb];
      const float tcom[3] = {cmr[0], cmr[1], cmr[2]};
      const float tv[6] = {s.cvel[6 * tb], s.cvel[6 * tb + 1], s.cvel[6 * tb + 2], s.cvel[6 * tb + 3], s.cvel[6 * tb + 4], s.cvel[6 * tb + 5]};
      const float yaw_tar0 = info[DIAL_INFO_YAW_TAR], pos_tar_z = info[DIAL_INFO_POS_TAR + 2];
      float out = 0.f;
      if (it == 0) {
        if (walk) {
          float reward_gaits = 0.f;
  #pragma unroll
          for (int f = 0; f < NF; f++) {
            const float z_tar = PRE ? s.ztab[st * DIAL_ZTAB_W + f] : s.ztar[f], zs = s.spos[3 * m->feet_site[f] + 2];   // (PRE: the rollout's gait-clock table)
            float fz;
            if (kind == DIAL_TASK_GO2_WALK) {
              float e = (z_tar - zs) / 0.05f;
              reward_gaits += e * e;
              fz = zs - m->foot_radius;
            } else if constexpr (kind == DIAL_TASK_H1_PUSH_CRATE) {   // the foot capsule's two floor contacts (unitree_h1_env.py:474-480)
              float zf = dm::fminf_(s.cdist[pc_foot_contact(m, f, 0)], s.cdist[pc_foot_contact(m, f, 1)]);
              reward_gaits += (z_tar - zf) * (z_tar - zf);
              fz = zs;
            } else if (kind == DIAL_TASK_H1_WALK) {
              float zf = dm::fminf_(s.cdist[2 * f], s.cdist[2 * f + 1]);
              reward_gaits += (z_tar - zf) * (z_tar - zf);
              fz = zs;
            } else {  // H1 loco: four contacts per foot (unitree_h1_env.py:746-752)
              float zf = dm::fminf_(dm::fminf_(s.cdist[4 * f], s.cdist[4 * f + 1]), dm::fminf_(s.cdist[4 * f + 2], s.cdist[4 * f + 3]));
              reward_gaits += (z_tar - zf) * (z_tar - zf);
              fz = zs;
            }
            if (FULL_INFO) {
              const bool contact = fz < 1e-3f;
              const bool filt = contact || (info[DIAL_INFO_LAST_CONTACT + f] != 0.f);
              info[DIAL_INFO_AIR_TIME + f] = (info[DIAL_INFO_AIR_TIME + f] + dt) * (filt ? 0.f : 1.f);
              info[DIAL_INFO_LAST_CONTACT + f] = contact ? 1.f : 0.f;
            }
          }
          out = -reward_gaits;
        } else {
          const int stage = (int)info[DIAL_INFO_STAGE];
          float reward_contact = 0.f, penalty_contact = 0.f;
          for (int i = 0; i < 4; i++) {
            bool pen = s.cdist[i] <= 0.001f;
            for (int j = 0; j < m->n_stage; j++) {
              float dx = s.cpos[3 * i] - tg->contact_targets[j][i][0], dy = s.cpos[3 * i + 1] - tg->contact_targets[j][i][1];
              bool cond = (dx * dx + dy * dy) <= tg->contact_radius[j][i] * tg->contact_radius[j][i];
              float val = (j == stage ? 1.f : 0.f) * dm::clip(s.cdist[i] * -1.0f + 1.0f, 0.f, 1.f);
              reward_contact += cond ? val : 0.f;
              pen = pen && !cond;
            }
            penalty_contact += pen ? 1.f : 0.f;
          }
          out = reward_contact;
          s.rpart[4] = penalty_contact;
        }
      } else if (it == 1) {
        float rot_u[4] = {uq[0], uq[1], uq[2], uq[3]};
        float up[3] = {0.f, 0.f, 1.f}, vec[3];
        dm::rotate(vec, up, rot_u);
        out = -((vec[0] - 0.f) * (vec[0] - 0.f) + (vec[1] - 0.f) * (vec[1] - 0.f) + (vec[2] - 1.f) * (vec[2] - 1.f));
      } else if (it == 2) {
        float rot_t[4] = {tq[0], tq[1], tq[2], tq[3]};
        const float yaw = quat_yaw(rot_t);
        if (walk) {
          float yaw_tar;
          if constexpr (PRE) yaw_tar = s.ztab[st * DIAL_ZTAB_W + DIAL_MAX_FEET + 3];   // (the rollout's table of ramped targets)
          else {
            const float a2 = cmd[5];
            const float avt = dm::fminf_(a2 * step * dt / m->ramp_up_time, a2);
            yaw_tar = yaw_tar0 + avt * dt * step;
          }
          const float d_yaw = yaw - yaw_tar;
          // atan2(sin d, cos d) wraps d to (-pi, pi]; d - 2 pi rint(d / 2 pi) is the same angle without trig
          const float wy = d_yaw - 6.283185307179586f * DM_RINT(d_yaw * 0.15915494309189535f);
          out = -(wy * wy);
        } else {
          const float ey = yaw - tg->yaw_targets[(int)info[DIAL_INFO_STAGE]];
          out = -(ey * ey);
        }
      } else if (it == 3 || it == 4) {
        if (walk) {
          float rot_t[4] = {tq[0], tq[1], tq[2], tq[3]};
          float off[3] = {tp[0] - tcom[0], tp[1] - tcom[1], tp[2] - tcom[2]};
          float ang[3] = {tv[0], tv[1], tv[2]};
          if (it == 3) {
            float cr[3], vel[3], vb[3];
            dm::cross3(cr, off, ang);
            for (int k = 0; k < 3; k++) vel[k] = tv[3 + k] - cr[k];
            dm::inv_rotate(vb, vel, rot_t);
            float vt[2];
            for (int k = 0; k < 2; k++) {
              if constexpr (PRE) vt[k] = s.ztab[st * DIAL_ZTAB_W + DIAL_MAX_FEET + k];
              else { const float v = cmd[k]; vt[k] = dm::fminf_(v * step * dt / m->ramp_up_time, v); }
            }
            const float e0 = vb[0] - vt[0], e1 = vb[1] - vt[1];
            out = -(e0 * e0 + e1 * e1);
          } else {
            float ab[3], angs[3] = {ang[0] * DIAL_PI / 180.0f, ang[1] * DIAL_PI / 180.0f, ang[2] * DIAL_PI / 180.0f};
            dm::inv_rotate(ab, angs, rot_t);
            if (kind == DIAL_TASK_H1_LOCO) {   // all three components (unitree_h1_env.py:797)
              float e3 = 0.f;
              for (int k = 0; k < 3; k++) {
                const float a = cmd[3 + k];
                const float e = ab[k] - dm::fminf_(a * step * dt / m->ramp_up_time, a);
                e3 += e * e;
              }
              out = -e3;
            } else {
              float avt;
              if constexpr (PRE) avt = s.ztab[st * DIAL_ZTAB_W + DIAL_MAX_FEET + 2];
              else { const float a2 = cmd[5]; avt = dm::fminf_(a2 * step * dt / m->ramp_up_time, a2); }
              const float ea = ab[2] - avt;
              out = -(ea * ea);
            }
          }
        } else if (it == 3) {
          const int stage = (int)info[DIAL_INFO_STAGE];
          float rp = 0.f;
          for (int k = 0; k < 3; k++) { float e = tp[k] - tg->pose_targets[stage][k]; rp += e * e; }
          out = -rp;
        } else {
          return s.rpart[4];  // seq-jump: the penalty count, written by term 0
        }
      } else if (it == 5) {
        const float dh = tp[2] - pos_tar_z;
        out = -(dh * dh);
      } else if (it == 6) {
        float reward_energy = 0.f;
        if (kind == DIAL_TASK_H1_WALK || kind == DIAL_TASK_H1_PUSH_CRATE)
          for (int a = 0; a < nu; a++) { float e = s.ctrl[a] / m->tau_range[a][1]; reward_energy += e * e; }
        if (kind == DIAL_TASK_H1_LOCO) {   // energy uses the post-step qvel; foot-level term shares this lane
          for (int a = 0; a < nu; a++) { float e = s.ctrl[a] / m->tau_range[a][1] * s.qvel[6 + a] / 160.0f; reward_energy += e * e; }
          float lvl = 0.f;
          for (int f = 0; f < 2; f++) {
            const int si = m->feet_site[f], sb = m->site_bodyid[si];
            float bq[4] = {s.xquat[4 * sb], s.xquat[4 * sb + 1], s.xquat[4 * sb + 2], s.xquat[4 * sb + 3]};
            float sq[4] = {m->site_quat[si][0], m->site_quat[si][1], m->site_quat[si][2], m->site_quat[si][3]};
            float q[4], mat[9];
            dm::quat_mul(q, bq, sq);
            dm::quat_to_mat(mat, q);
            lvl += (mat[2] - 0.f) * (mat[2] - 0.f) + (mat[5] - 0.f) * (mat[5] - 0.f) + (mat[8] - 1.f) * (mat[8] - 1.f);
          }
          out = -reward_energy;
          s.rpart[8] = -lvl;                                 // foot-level term (slot 8)
        } else {
          out = -reward_energy;
        }
      } else {
        float rot_t[4] = {tq[0], tq[1], tq[2], tq[3]};
        float up[3] = {0.f, 0.f, 1.f}, upv[3];
        dm::rotate(upv, up, rot_t);
        bool done = upv[2] < 0.f;
        for (int a = 0; a < nu; a++) {
          float q = s.qpos[7 + a];
          done = done || q < m->joint_range[a][0] || q > m->joint_range[a][1];
        }
        done = done || tp[2] < m->done_height;
        out = done ? 1.f : 0.f;
      }
      return out;
    };
    // ---- terms, total in the reference's summation order and info update (one lane; last_ctrl by nu lanes)
    // (PRE: lanes 1 .. nq also store this step's q row -- the integrator's results, one element per lane)
    w.items(1 + (PRE ? dim_nq(m) : nu), [&](int it) {
      float* info = s.info;
      if (it > 0) {
        if (!walk && it <= nu) info[DIAL_INFO_LAST_CTRL + it - 1] = s.ctrl[it - 1];
        if constexpr (PRE) { if (w.out_io && w.out_io->qss) w.out_io->qss[(size_t)w.out_row * dim_nq(m) + (it - 1)] = s.qpos[it - 1]; }
        return;
      }
      // this step's velocity command, once for all terms (randomize_tasks: the episode's draw -- the only branch left
      // between here and the end of the phase)
      float cmd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (walk && !PRE) step_cmd(m, tg, info[DIAL_INFO_STEP], cmd);   // (PRE: the targets come from the rollout's table)
      float r[9];
      r[0] = term(std::integral_constant<int, 0>{}, cmd);
      r[1] = term(std::integral_constant<int, 1>{}, cmd);
      r[2] = term(std::integral_constant<int, 2>{}, cmd);
      r[3] = term(std::integral_constant<int, 3>{}, cmd);
      r[4] = term(std::integral_constant<int, 4>{}, cmd);
      r[5] = term(std::integral_constant<int, 5>{}, cmd);
      r[6] = term(std::integral_constant<int, 6>{}, cmd);
      r[7] = FULL_INFO ? term(std::integral_constant<int, 7>{}, cmd) : 0.f;
      r[8] = kind == DIAL_TASK_H1_LOCO ? s.rpart[8] : 0.f;      // foot-level term, written by term 6
      if constexpr (kind == DIAL_TASK_H1_PUSH_CRATE) r[8] = pc_contact_reward(m, s);
      const float dt = m->dt, step = info[DIAL_INFO_STEP];
      float reward;
      if (kind == DIAL_TASK_GO2_WALK) {          // unitree_go2_env.py:227-239
        reward = r[0] * 0.1f + r[1] * 0.5f + r[2] * 0.3f + r[3] * 1.0f + r[4] * 1.0f + r[5] * 1.0f;
      } else if (kind == DIAL_TASK_H1_WALK) {    // unitree_h1_env.py:286-298
        reward = r[0] * 5.0f + r[1] * 0.5f + r[2] * 0.1f + r[3] * 1.0f + r[4] * 1.0f + r[5] * 0.5f + r[6] * 0.01f;
      } else if (kind == DIAL_TASK_H1_PUSH_CRATE) {   // unitree_h1_env.py:534-548 (air_time, pos and alive carry the weight 0.0)
        reward = r[0] * 5.0f + r[1] * 0.01f + r[2] * 0.1f + r[3] * 1.0f + r[4] * 1.0f + r[5] * 0.5f + r[6] * 0.01f + r[8] * 0.05f;
      } else if (kind == DIAL_TASK_H1_LOCO) {    // unitree_h1_env.py:812-827
        reward = r[0] * 10.0f + r[1] * 0.5f + r[2] * 0.5f + r[3] * 1.0f + r[4] * 1.0f + r[5] * 0.5f +
                 r[8] * 0.02f + r[6] * 0.01f;
      } else {                                      // unitree_go2_env.py:485-496
        reward = r[3] * 1.0f + r[1] * 1.0f + r[2] * 0.3f + r[0] * 0.1f - r[4] * 0.1f + 1.0f * 10.0f;
      }
      if (walk && !PRE) {   // (info.vel_tar / ang_vel_tar: written for the caller of env.step; no walking reward reads them back)
        for (int k = 0; k < 3; k++) {
          const float v = cmd[k], a = cmd[3 + k];
          info[DIAL_INFO_VEL_TAR + k] = dm::fminf_(v * step * dt / m->ramp_up_time, v);
          info[DIAL_INFO_ANG_VEL_TAR + k] = dm::fminf_(a * step * dt / m->ramp_up_time, a);
        }
      }
      if (FULL_INFO) info[DIAL_INFO_DONE] = r[7];
      info[DIAL_INFO_STEP] = step + 1.f;
      if (kind == DIAL_TASK_GO2_SEQ_JUMP) {
        float st = DM_FLOOR(info[DIAL_INFO_STEP] * dt / m->jump_dt);
        info[DIAL_INFO_STAGE] = dm::fminf_(st, (float)(m->n_stage - 1));
      }
      info[DIAL_INFO_REWARD] = reward;
      if constexpr (PRE) { if (w.out_io && w.out_io->rewss) w.out_io->rewss[w.out_row] = reward; }
    });
  };
  // wave-uniform dispatch on the task kind; a dimension-specialised instantiation only carries its robot's kinds
  {
    constexpr uint32_t kmask = task_kind_mask<typename M::D>();
    const int kind_u = DM_UNIFORM_I(m->kind);
    if ((kmask >> DIAL_TASK_GO2_WALK & 1u) && kind_u == DIAL_TASK_GO2_WALK) reward_phase(std::integral_constant<int, DIAL_TASK_GO2_WALK>{});
    else if ((kmask >> DIAL_TASK_GO2_SEQ_JUMP & 1u) && kind_u == DIAL_TASK_GO2_SEQ_JUMP) reward_phase(std::integral_constant<int, DIAL_TASK_GO2_SEQ_JUMP>{});
    else if ((kmask >> DIAL_TASK_H1_WALK & 1u) && kind_u == DIAL_TASK_H1_WALK) reward_phase(std::integral_constant<int, DIAL_TASK_H1_WALK>{});
    else if ((kmask >> DIAL_TASK_H1_LOCO & 1u) && kind_u == DIAL_TASK_H1_LOCO) reward_phase(std::integral_constant<int, DIAL_TASK_H1_LOCO>{});
    else if constexpr ((kmask >> DIAL_TASK_H1_PUSH_CRATE) & 1u) {
      if (kind_u == DIAL_TASK_H1_PUSH_CRATE) reward_phase(std::integral_constant<int, DIAL_TASK_H1_PUSH_CRATE>{});
    }
  }
  DIAL_MARK(w, 10);
  return s.info[DIAL_INFO_REWARD];
}

// Initialise the world-body entries of the kinematic arrays (done once per wavefront).
template <class W>
DIAL_DEV void init_world(W& w, const Ws& s) {
  w.items(1, [&](int) {
    for (int k = 0; k < 3; k++) { s.xpos[k] = 0.f; s.com[k] = 0.f; }
    s.xquat[0] = 1.f; s.xquat[1] = 0.f; s.xquat[2] = 0.f; s.xquat[3] = 0.f;
  });
}
// Square layout: the structurally zero entries of M are written once per kernel; the sparse writers never touch them.
template <class W, class M>
DIAL_DEV void init_square(W& w, const M* m, const Ws& s) {
  (void)m;
  if constexpr (M::D::square) w.items(M::D::NV * M::D::S, [&](int e) { s.M[e] = 0.f; });
  if constexpr (kQuadDims<typename M::D>) init_quad(w, m, s);
  if constexpr (W::half2) init_quad2(w, m, s);
  if constexpr (kQuadGenDims<typename M::D>) init_quad_gen(w, m, s);
}

}  // namespace dial

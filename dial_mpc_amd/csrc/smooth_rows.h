// smooth_rows.h -- the position / velocity stage of mjx.forward (smooth.kinematics, com_pos, com_vel, crb, rne, make_m,
// qfrc_smooth) in registers for robots that are ONE kinematic tree under a free root body: the H1 (walk, jog, loco).
// The table-driven generalisation of smooth_quad.h (which stays the Go2's: its header explains the idea and the evidence).
//
// Layout (cmodel.h: RowTab, built on the host by rows_build): every root-to-leaf chain of the body tree is a ROW of 16 lanes,
//   lane 16 r + d = the chain's body at depth d   (d = 0: the free root, every row keeps a copy)
// so that "my parent" is the lane below (DPP row_shr) and the subtree of a body is the rest of its row.  A body on several
// chains (the H1's torso: two arms) is OWNED by the first row and carried as a copy by the other: copies take part in the
// root-to-leaf scans (poses, velocities) but contribute nothing of their own to the leaf-to-root sums; the owner adds the
// copy's accumulated subtree afterwards (one v_readlane per component).  Lanes 8..13 of row 0 stand for the root's six dofs
// in the dof-indexed outputs (M, qfrc_smooth, cdof).  Bodies without a joint (the loco model's welded arms) are chain
// members with a constant local transform and no dof.
//   * root-to-leaf quantities are PREFIX SCANS along the row with shifts 1, 2, 4: poses compose (a . b = [a.p + R(a.q) b.p,
//     a.q b.q], associative), velocities and accelerations add -- three rounds for chains up to 7 deep instead of one LDS
//     phase per level;
//   * leaf-to-root sums (composite inertia, subtree forces) are suffix scans with row_shl.
// Sums therefore associate differently from forward()'s sequential loops (rounding level; the oracle parity tests cover it).
// Contacts: one LDS phase after the stage, as in forward() (collide_contact).
// Second form of the layout (RowsOf::static_root, the Allegro hand): the chains hang off a body WELDED to the world (the palm:
// constant pose, no velocity, no dofs) and the free body is on its own (the object: lane 15, its six dofs on lanes 8..13) --
// two kinematic trees, two subtree centres of mass.
#pragma once
#include "derived.h"
#include "dmath.h"

namespace dial {

template <class D>
inline constexpr bool kRowsDims = D::rows_stage;
template <class D>
inline constexpr bool kRowsGenDims = D::rows_gen;

// A body that is a kinematic tree of its own: one SLIDE joint on the world, no children (the push-crate scene's crate).  One lane,
// forward()'s formulas for that body alone: pose, subtree centre of mass (its own), cinert, cdof = [0, axis], cvel = cdof qvel,
// cacc = [0, -g] (no velocity-dependent cdof_dot: the parent is at rest), cfl = cinert cacc + cvel x* (cinert cvel), M = cdof .
// (cinert cdof) + armature, qfrc_smooth = passive - cdof . cfl + actuator.
template <class M>
DIAL_DEV void solo_slide_body(const M* m, const Ws& s, int b) {
  const int ji = m->body_jntadr[b], qa = m->jnt_qposadr[ji], da = m->jnt_dofadr[ji];
  const float lq[4] = {m->body_quat[b][0], m->body_quat[b][1], m->body_quat[b][2], m->body_quat[b][3]};
  const float ja[3] = {m->jnt_axis[ji][0], m->jnt_axis[ji][1], m->jnt_axis[ji][2]};
  float ax[3], pos[3], t3[3], qi[4], R[9], xi[3], cm[3], off[3];
  dm::rotate(ax, ja, lq);
  const float disp = s.qpos[qa] - m->qpos0[qa];
  for (int k = 0; k < 3; k++) pos[k] = m->body_pos[b][k] + ax[k] * disp;
  const float ip[3] = {m->body_ipos[b][0], m->body_ipos[b][1], m->body_ipos[b][2]};
  const float iq[4] = {m->body_iquat[b][0], m->body_iquat[b][1], m->body_iquat[b][2], m->body_iquat[b][3]};
  dm::rotate(t3, ip, lq);
  dm::quat_mul(qi, lq, iq);
  dm::quat_to_mat(R, qi);
  const float mb = m->body_mass[b];
  for (int k = 0; k < 3; k++) {
    xi[k] = pos[k] + t3[k];
    cm[k] = mb < MJ_MINVAL ? xi[k] : (xi[k] * mb) / mb;   // smooth.com_pos of a one-body tree
    off[k] = xi[k] - cm[k];
  }
  float ci[10];
  {
    const float oo = dm::dot3(off, off);
    const float in0 = m->body_inertia[b][0], in1 = m->body_inertia[b][1], in2 = m->body_inertia[b][2];
    const int ii[6] = {0, 1, 2, 0, 0, 1}, jj[6] = {0, 1, 2, 1, 2, 2};
    for (int e = 0; e < 6; e++) {
      const int i = ii[e], j = jj[e];
      const float v = R[3 * i] * in0 * R[3 * j] + R[3 * i + 1] * in1 * R[3 * j + 1] + R[3 * i + 2] * in2 * R[3 * j + 2];
      const float hh = (i == j ? oo : 0.f) - off[i] * off[j];
      ci[e] = v + hh * mb;
    }
    for (int k = 0; k < 3; k++) ci[6 + k] = off[k] * mb;
    ci[9] = mb;
  }
  const float qv = s.qvel[da];
  const float cd[6] = {0.f, 0.f, 0.f, ax[0], ax[1], ax[2]};
  float cv[6], ca[6], f1[6], f2[6], f3[6], fd[6];
  for (int k = 0; k < 6; k++) { cv[k] = cd[k] * qv; ca[k] = k >= 3 ? -m->gravity[k - 3] : 0.f; }
  dm::inert_mul(f1, ci, ca);
  dm::inert_mul(f2, ci, cv);
  dm::motion_cross_force(f3, cv, f2);
  dm::inert_mul(fd, ci, cd);
  float mdd = 0.f, bias = 0.f;
  for (int k = 0; k < 6; k++) { mdd += fd[k] * cd[k]; bias += cd[k] * (f1[k] + f3[k]); }
  const float passive = -m->dof_damping[da] * qv;
  const int a = m->dof_act[da], aa = a >= 0 ? a : 0;
  const float c0 = s.ctrl[aa];
  const float c = m->act_ctrllimited[aa] ? dm::clip(c0, m->act_ctrlrange[aa][0], m->act_ctrlrange[aa][1]) : c0;
  const float force = m->act_isposition[aa] ? m->act_kp[aa] * (c - s.qpos[m->act_qposadr[aa]]) : c;
  const float qf = passive - bias + (a >= 0 ? m->act_gear[aa] * force : 0.f);
  for (int k = 0; k < 3; k++) { s.xpos[3 * b + k] = pos[k]; s.com[3 * m->body_rootid[b] + k] = cm[k]; }
  for (int k = 0; k < 4; k++) s.xquat[4 * b + k] = lq[k];
  for (int k = 0; k < 6; k++) { s.cvel[6 * b + k] = cv[k]; s.cdof[6 * da + k] = cd[k]; }
  s.qfs[da] = qf;
  s.rhs[da] = qf;
  s.M[tri_idx(da, da)] = mdd + m->dof_armature[da];
}

// GEN (Dims::rows_gen, the generic feature set on the layout -- push crate): M as the packed lower triangle, no geoms in the lanes
// (every geom frame comes out of one LDS phase at the end, from the stored poses), the solo slide body on lane 15 of the dofs'
// output phase; collisions / Jacobians / rows stay the generic feature set's.
template <bool GEN = false, class W, class M>
DIAL_DEV void forward_smooth_rows(W& w, const M* m, const Ws& s) {
  using RT = RowsOf<typename M::D::Topo>;
  constexpr int S = M::D::S, MAXD = RT::maxd;
  static_assert(GEN != M::D::square, "square layout: the robots' own instantiations; packed M: the generic feature set");
  constexpr bool SR = RT::static_root;      // chains under a welded root, the free body on its own lane
  constexpr int FREE = SR ? 15 : 0;         // the lane whose pose / velocity are the free body's
  static_assert(MAXD >= 1 && MAXD <= 7, "chains of at most 7 bodies below the root (three scan rounds; lanes 8..13 = root dofs)");
  DIAL_MARK(w, 15);
#if !defined(DIAL_EMU) && !defined(DIAL_QUAD_HOIST)
  // opaque copy of the lane id for this stage (smooth_quad.h: role masks and table addresses re-derived per step instead of
  // ~20 VGPRs hoisted out of the T-step loop and carried through the solver)
  const int lane_keep = w.lane;
  { int lq = w.lane; asm volatile("" : "+v"(lq)); w.lane = lq; }
#endif
  // ---- smooth.kinematics: every lane's transform relative to its parent (root lanes: the absolute pose), then the scan
  vfloat P[7];   // pos(3) quat(4)
  w.per_lane_n(P, [&](int l, float* o) {
    const int fl = m->rows.flags[l], b = m->rows.body[l];
    const bool joint = (fl & ROWS_JOINT) != 0, root = SR ? (fl & ROWS_SOLO) != 0 : ((fl & ROWS_BODY) != 0 && (l & 15) == 0);   // the FREE body
    const int ji = joint ? m->body_jntadr[b] : 0, qa = joint ? m->jnt_qposadr[ji] : 7;
    float tq[4] = {s.qpos[3], s.qpos[4], s.qpos[5], s.qpos[6]};
    dm::normalize4(tq);
    const int bflags = m->body_flags[b];
    float lq[4] = {m->body_quat[b][0], m->body_quat[b][1], m->body_quat[b][2], m->body_quat[b][3]};
    float lp[3] = {m->body_pos[b][0], m->body_pos[b][1], m->body_pos[b][2]};
    const float jp[3] = {m->jnt_pos[ji][0], m->jnt_pos[ji][1], m->jnt_pos[ji][2]};
    const float ja[3] = {m->jnt_axis[ji][0], m->jnt_axis[ji][1], m->jnt_axis[ji][2]};
    float qloc[4], qb[4], t0[3], t1[3];
    dm::axis_angle_to_quat(qloc, ja, joint ? s.qpos[qa] - m->qpos0[qa] : 0.f);   // (no joint: angle 0 -> the identity)
    if (bflags & 1) { qb[0] = qloc[0]; qb[1] = qloc[1]; qb[2] = qloc[2]; qb[3] = qloc[3]; }
    else dm::quat_mul(qb, lq, qloc);
    if (joint && !(bflags & 2)) {
      if (bflags & 1) { t0[0] = jp[0]; t0[1] = jp[1]; t0[2] = jp[2]; }
      else dm::rotate(t0, jp, lq);
      dm::rotate(t1, jp, qb);
      for (int k = 0; k < 3; k++) lp[k] += t0[k] - t1[k];
    }
    const bool chain = (fl & ROWS_BODY) != 0 && !root;   // idle lanes and the static-geom lane keep the identity
    for (int k = 0; k < 3; k++) o[k] = root ? s.qpos[k] : (chain ? lp[k] : 0.f);
    for (int k = 0; k < 4; k++) o[3 + k] = root ? tq[k] : (chain ? qb[k] : (k == 0 ? 1.f : 0.f));
  });
  static_for<0, 3>([&](auto IT) {
    constexpr int sh = 1 << decltype(IT)::value;
    if constexpr (sh <= MAXD) {
      vfloat Q[7], N[7];
      DIAL_UNROLL_FULL
      for (int k = 0; k < 7; k++) Q[k] = w.template row_shr_lo<sh>(P[k]);
      w.per_lane_n(N, [&](int l, float* o) {
        const int d = l & 15;
        const bool on = d >= sh && d <= MAXD;
        const float ap[3] = {lane_val(Q[0], l), lane_val(Q[1], l), lane_val(Q[2], l)};
        const float aq[4] = {lane_val(Q[3], l), lane_val(Q[4], l), lane_val(Q[5], l), lane_val(Q[6], l)};
        const float bp[3] = {lane_val(P[0], l), lane_val(P[1], l), lane_val(P[2], l)};
        const float bq[4] = {lane_val(P[3], l), lane_val(P[4], l), lane_val(P[5], l), lane_val(P[6], l)};
        float pos[3], quat[4];
        dm::rotate(pos, bp, aq);
        for (int k = 0; k < 3; k++) pos[k] += ap[k];
        dm::quat_mul(quat, aq, bq);
        for (int k = 0; k < 3; k++) o[k] = on ? pos[k] : bp[k];
        for (int k = 0; k < 4; k++) o[3 + k] = on ? quat[k] : bq[k];
      });
      DIAL_UNROLL_FULL
      for (int k = 0; k < 7; k++) P[k] = N[k];
    }
  });
  DIAL_MARK(w, 0);
  // ---- local_to_global: inertial frames, up to two geoms and one site per lane
  vfloat F[16];   // xipos(3) ximat(9) | mass-weighted xipos(3), mass
  vfloat G[15];   // site(3) | geom 0: centre(3) axis(3) | geom 1: centre(3) axis(3)
  {
    vfloat T[31];
    w.per_lane_n(T, [&](int l, float* o) {
      const int fl = m->rows.flags[l], b = m->rows.body[l];
      const bool owner = (fl & ROWS_OWNER) != 0 && !(SR && (fl & ROWS_SOLO));   // (the solo body is a tree of its own)
      const int si = m->rows.site[l] == 255 ? 0 : m->rows.site[l];
      const float p[3] = {lane_val(P[0], l), lane_val(P[1], l), lane_val(P[2], l)};
      const float q[4] = {lane_val(P[3], l), lane_val(P[4], l), lane_val(P[5], l), lane_val(P[6], l)};
      const float ip[3] = {m->body_ipos[b][0], m->body_ipos[b][1], m->body_ipos[b][2]};
      const float iq[4] = {m->body_iquat[b][0], m->body_iquat[b][1], m->body_iquat[b][2], m->body_iquat[b][3]};
      const float sp[3] = {m->site_pos[si][0], m->site_pos[si][1], m->site_pos[si][2]};
      const float mass = m->body_mass[b];
      float t3[3], qi[4], mat[9], ts[3];
      dm::rotate(t3, ip, q);
      dm::quat_mul(qi, q, iq);
      dm::quat_to_mat(mat, qi);
      dm::rotate(ts, sp, q);
      for (int k = 0; k < 3; k++) {
        const float xi = p[k] + t3[k];
        o[k] = xi;
        o[12 + k] = owner ? xi * mass : 0.f;
        o[16 + k] = p[k] + ts[k];
      }
      for (int k = 0; k < 9; k++) o[3 + k] = mat[k];
      o[15] = owner ? mass : 0.f;
      for (int e = 0; e < 2; e++) {
        if constexpr (GEN) { for (int k = 0; k < 6; k++) o[19 + 6 * e + k] = 0.f; }
        else {
        const int g = m->rows.geom[l][e] == 255 ? 0 : m->rows.geom[l][e];
        const float gp[3] = {m->geom_pos[g][0], m->geom_pos[g][1], m->geom_pos[g][2]};
        const float gq[4] = {m->geom_quat[g][0], m->geom_quat[g][1], m->geom_quat[g][2], m->geom_quat[g][3]};
        float tg[3], qg[4], gm[9];
        dm::rotate(tg, gp, q);
        dm::quat_mul(qg, q, gq);
        dm::quat_to_mat(gm, qg);
        for (int k = 0; k < 3; k++) o[19 + 6 * e + k] = p[k] + tg[k];
        o[22 + 6 * e] = gm[2]; o[23 + 6 * e] = gm[5]; o[24 + 6 * e] = gm[8];
        }
      }
    });
    DIAL_UNROLL_FULL
    for (int k = 0; k < 16; k++) F[k] = T[k];
    DIAL_UNROLL_FULL
    for (int k = 0; k < 15; k++) G[k] = T[16 + k];
  }
  DIAL_MARK(w, 16);
  // ---- smooth.com_pos
  float com[3];
  {
    vfloat c4[4] = {F[12], F[13], F[14], F[15]};
    float r4[4];
    w.vsumN(c4, r4);
    for (int k = 0; k < 3; k++) com[k] = r4[3] < MJ_MINVAL ? bcast(F[k], 0) : r4[k] / r4[3];
  }
  // (static root: the free body's subtree is itself -- its centre of mass is its own xipos)
  const float comF[3] = {SR ? bcast(F[0], FREE) : com[0], SR ? bcast(F[1], FREE) : com[1], SR ? bcast(F[2], FREE) : com[2]};
  // the free body's pose, rotation matrix and rotational cdofs: the same value in every lane
  const float tpos[3] = {bcast(P[0], FREE), bcast(P[1], FREE), bcast(P[2], FREE)};
  const float tquat[4] = {bcast(P[3], FREE), bcast(P[4], FREE), bcast(P[5], FREE), bcast(P[6], FREE)};
  float Rt[9], cdT[3][6];
  dm::quat_to_mat(Rt, tquat);
  const float offt[3] = {comF[0] - tpos[0], comF[1] - tpos[1], comF[2] - tpos[2]};
  for (int i = 0; i < 3; i++) {
    const float a[3] = {Rt[i], Rt[3 + i], Rt[6 + i]};
    float cr[3];
    dm::cross3(cr, a, offt);
    for (int k = 0; k < 3; k++) { cdT[i][k] = a[k]; cdT[i][3 + k] = cr[k]; }
  }
  DIAL_MARK(w, 17);
  // ---- cinert (body lanes) and cdof (joint lanes, root-dof lanes)
  vfloat X[16];   // cinert(10) | local force cfl(6): the quantities summed over subtrees
  vfloat CD[6];
  {
    vfloat T[16];
    w.per_lane_n(T, [&](int l, float* o) {
      const int fl = m->rows.flags[l], b = m->rows.body[l];
      const bool body = (fl & ROWS_BODY) != 0, joint = (fl & ROWS_JOINT) != 0, tdof = (fl & ROWS_TDOF) != 0;
      const int ji = joint ? m->body_jntadr[b] : 0, kd = tdof ? m->rows.dof[l] : 3;
      const float R[9] = {lane_val(F[3], l), lane_val(F[4], l), lane_val(F[5], l), lane_val(F[6], l), lane_val(F[7], l),
                          lane_val(F[8], l), lane_val(F[9], l), lane_val(F[10], l), lane_val(F[11], l)};
      const bool solo = SR && (fl & ROWS_SOLO) != 0;
      const float off[3] = {lane_val(F[0], l) - (solo ? comF[0] : com[0]), lane_val(F[1], l) - (solo ? comF[1] : com[1]),
                            lane_val(F[2], l) - (solo ? comF[2] : com[2])};
      const float mb = m->body_mass[b], oo = dm::dot3(off, off);
      const float in0 = m->body_inertia[b][0], in1 = m->body_inertia[b][1], in2 = m->body_inertia[b][2];
      const int ii[6] = {0, 1, 2, 0, 0, 1}, jj[6] = {0, 1, 2, 1, 2, 2};
      for (int e = 0; e < 6; e++) {
        const int i = ii[e], j = jj[e];
        const float v = R[3 * i] * in0 * R[3 * j] + R[3 * i + 1] * in1 * R[3 * j + 1] + R[3 * i + 2] * in2 * R[3 * j + 2];
        const float hh = (i == j ? oo : 0.f) - off[i] * off[j];
        o[e] = body ? v + hh * mb : 0.f;
      }
      for (int k = 0; k < 3; k++) o[6 + k] = body ? off[k] * mb : 0.f;
      o[9] = body ? mb : 0.f;
      const float bq[4] = {lane_val(P[3], l), lane_val(P[4], l), lane_val(P[5], l), lane_val(P[6], l)};
      const float jp[3] = {m->jnt_pos[ji][0], m->jnt_pos[ji][1], m->jnt_pos[ji][2]};
      const float ja[3] = {m->jnt_axis[ji][0], m->jnt_axis[ji][1], m->jnt_axis[ji][2]};
      float anchor[3], jaxis[3], cr[3];
      if (m->body_flags[b] & 2) { anchor[0] = 0.f; anchor[1] = 0.f; anchor[2] = 0.f; }
      else dm::rotate(anchor, jp, bq);
      for (int k = 0; k < 3; k++) anchor[k] += lane_val(P[k], l);
      dm::rotate(jaxis, ja, bq);
      const float offj[3] = {com[0] - anchor[0], com[1] - anchor[1], com[2] - anchor[2]};
      dm::cross3(cr, jaxis, offj);
      // (0 / 1 weights, not a select chain over the array: smooth_quad.h)
      const float w0 = kd == 3 ? 1.f : 0.f, w1 = kd == 4 ? 1.f : 0.f, w2 = kd == 5 ? 1.f : 0.f;
      for (int k = 0; k < 3; k++) {
        const float ta = w0 * cdT[0][k] + w1 * cdT[1][k] + w2 * cdT[2][k];
        const float tl = (k == kd ? 1.f : 0.f) + (w0 * cdT[0][3 + k] + w1 * cdT[1][3 + k] + w2 * cdT[2][3 + k]);
        o[10 + k] = joint ? jaxis[k] : (tdof ? ta : 0.f);
        o[13 + k] = joint ? cr[k] : (tdof ? tl : 0.f);
      }
    });
    DIAL_UNROLL_FULL
    for (int k = 0; k < 10; k++) X[k] = T[k];
    DIAL_UNROLL_FULL
    for (int k = 0; k < 6; k++) CD[k] = T[10 + k];
  }
  DIAL_MARK(w, 18);
  // ---- smooth.com_vel + cdof_dot + rne forward: cvel = prefix sum of cdof qvel; cacc = prefix sum of (cvel[parent] x cdof) qvel
  const vfloat QVL = w.per_lane([&](int l) {
    const int fl = m->rows.flags[l];
    return (fl & (ROWS_JOINT | ROWS_TDOF)) ? s.qvel[m->rows.dof[l]] : 0.f;   // this lane's joint velocity (copies included)
  });
  vfloat V[6], A[6];
  float velT[6], accT[6];
  {
    const float qv[6] = {s.qvel[0], s.qvel[1], s.qvel[2], s.qvel[3], s.qvel[4], s.qvel[5]};
    const float vs[6] = {0.f, 0.f, 0.f, qv[0], qv[1], qv[2]};   // the rotational dofs see the velocity after the translational ones
    for (int k = 0; k < 3; k++) { velT[k] = 0.f; velT[3 + k] = qv[k]; accT[k] = 0.f; accT[3 + k] = -m->gravity[k]; }
    for (int j = 0; j < 3; j++) {   // (vs has no angular part: cdof_dot = (0, vs.lin x cdof.ang), spelled out -- see smooth_quad.h)
      float cl[3];
      dm::cross3(cl, vs + 3, cdT[j]);
      for (int k = 0; k < 3; k++) accT[3 + k] += cl[k] * qv[3 + j];
      for (int k = 0; k < 6; k++) velT[k] += cdT[j][k] * qv[3 + j];
    }
  }
  w.per_lane_n(V, [&](int l, float* o) {
    const int fl = m->rows.flags[l];
    const bool root = SR ? (fl & ROWS_SOLO) != 0 : ((fl & ROWS_BODY) != 0 && (l & 15) == 0), joint = (fl & ROWS_JOINT) != 0;
    for (int k = 0; k < 6; k++) o[k] = root ? velT[k] : (joint ? lane_val(CD[k], l) * lane_val(QVL, l) : 0.f);   // (a welded root: 0)
  });
  static_for<0, 3>([&](auto IT) {
    constexpr int sh = 1 << decltype(IT)::value;
    if constexpr (sh <= MAXD) {
      DIAL_UNROLL_FULL
      for (int k = 0; k < 6; k++) V[k] = V[k] + w.template row_shr_lo<sh>(V[k]);
    }
  });
  {
    vfloat VP[6];
    DIAL_UNROLL_FULL
    for (int k = 0; k < 6; k++) VP[k] = w.template row_shr<1>(V[k]);
    w.per_lane_n(A, [&](int l, float* o) {
      const int fl = m->rows.flags[l];
      const bool root = SR ? (fl & ROWS_SOLO) != 0 : ((fl & ROWS_BODY) != 0 && (l & 15) == 0), joint = (fl & ROWS_JOINT) != 0;
      const bool sroot = SR && (fl & ROWS_BODY) != 0 && (l & 15) == 0;   // the welded root: at rest, acceleration = -gravity
      const float vp[6] = {lane_val(VP[0], l), lane_val(VP[1], l), lane_val(VP[2], l), lane_val(VP[3], l), lane_val(VP[4], l), lane_val(VP[5], l)};
      const float cd[6] = {lane_val(CD[0], l), lane_val(CD[1], l), lane_val(CD[2], l), lane_val(CD[3], l), lane_val(CD[4], l), lane_val(CD[5], l)};
      float cdd[6];
      dm::motion_cross(cdd, vp, cd);
      for (int k = 0; k < 6; k++) o[k] = root ? accT[k] : (joint ? cdd[k] * lane_val(QVL, l) : (sroot && k >= 3 ? -m->gravity[k - 3] : 0.f));
    });
  }
  static_for<0, 3>([&](auto IT) {
    constexpr int sh = 1 << decltype(IT)::value;
    if constexpr (sh <= MAXD) {
      DIAL_UNROLL_FULL
      for (int k = 0; k < 6; k++) A[k] = A[k] + w.template row_shr_lo<sh>(A[k]);
    }
  });
  // the bodies' outputs are complete: stored now, not at the end (32 registers fewer to carry through the dof stage)
  w.items(64, [&](int l) {
    const int fl = m->rows.flags[l], b = m->rows.body[l];
    if (fl & ROWS_OWNER) {
      for (int k = 0; k < 3; k++) s.xpos[3 * b + k] = lane_val(P[k], l);
      store4(s.xquat + 4 * b, lane_val(P[3], l), lane_val(P[4], l), lane_val(P[5], l), lane_val(P[6], l));
      for (int k = 0; k < 3; k++) store2(s.cvel + 6 * b + 2 * k, lane_val(V[2 * k], l), lane_val(V[2 * k + 1], l));
      const int si = m->rows.site[l];
      if (si != 255) for (int k = 0; k < 3; k++) s.spos[3 * si + k] = lane_val(G[k], l);
    }
    if constexpr (GEN) {   // packed M: the structural zeros (the dof lanes below write the tree's entries only)
      for (int e = l; e < M::D::NTRI; e += 64) s.M[e] = 0.f;
    } else
    for (int e = 0; e < 2; e++) {
      const int g = m->rows.geom[l][e];
      if (g != 255) for (int k = 0; k < 3; k++) { s.gpos[3 * g + k] = lane_val(G[3 + 6 * e + k], l); s.gaxis[3 * g + k] = lane_val(G[6 + 6 * e + k], l); }
    }
    if (l == 0) {   // subtree centres of mass live at their root body's index (row 0's root lane: the chains' tree)
      for (int k = 0; k < 3; k++) s.com[3 * m->body_rootid[b] + k] = com[k];
      for (int k = 0; k < 4; k++) s.qpos[3 + k] = tquat[k];   // (kinematics normalises the free joint's quaternion in place)
    }
    if (SR && l == FREE) for (int k = 0; k < 3; k++) s.com[3 * m->body_rootid[b] + k] = comF[k];
  });
  DIAL_MARK(w, 19);
  // ---- rne: local body forces cfl = cinert cacc + cvel x* (cinert cvel); copies contribute nothing to the subtree sums
  {
    vfloat T[16];
    w.per_lane_n(T, [&](int l, float* o) {
      const bool owner = (m->rows.flags[l] & ROWS_OWNER) != 0;
      float ci[10], ca[6], cv[6], f1[6], f2[6], f3[6];
      for (int k = 0; k < 10; k++) ci[k] = lane_val(X[k], l);
      for (int k = 0; k < 6; k++) { cv[k] = lane_val(V[k], l); ca[k] = lane_val(A[k], l); }
      dm::inert_mul(f1, ci, ca);
      dm::inert_mul(f2, ci, cv);
      dm::motion_cross_force(f3, cv, f2);
      for (int k = 0; k < 10; k++) o[k] = owner ? ci[k] : 0.f;
      for (int k = 0; k < 6; k++) o[10 + k] = owner ? f1[k] + f3[k] : 0.f;
    });
    DIAL_UNROLL_FULL
    for (int k = 0; k < 16; k++) X[k] = T[k];
  }
  // ---- subtree sums (smooth.crb, rne backward): suffix scan along the rows, the shared body's owner adds its copy's row,
  // the root = the sum of the rows' lane 0
  static_for<0, 3>([&](auto IT) {
    constexpr int sh = 1 << decltype(IT)::value;
    if constexpr (sh <= MAXD) {
      DIAL_UNROLL_FULL
      for (int k = 0; k < 16; k++) X[k] = X[k] + w.template row_shl_lo<sh>(X[k]);
    }
  });
  float XT[16];
  for (int k = 0; k < 16; k++) {
    if constexpr (SR) XT[k] = bcast(X[k], FREE);   // the free body has no children
    else XT[k] = (bcast(X[k], 0) + bcast(X[k], 16)) + (bcast(X[k], 32) + bcast(X[k], 48));
  }
  if constexpr (RT::merge_src >= 0) {
    float xm[16];
    for (int k = 0; k < 16; k++) xm[k] = bcast(X[k], RT::merge_src);
    vfloat N[16];
    w.per_lane_n(N, [&](int l, float* o) {
      for (int k = 0; k < 16; k++) o[k] = l == RT::merge_dst ? lane_val(X[k], l) + xm[k] : lane_val(X[k], l);
    });
    DIAL_UNROLL_FULL
    for (int k = 0; k < 16; k++) X[k] = N[k];
  }
  DIAL_MARK(w, 22);
  // ---- F_i = crb cdof_i, M = F . cdof over the ancestors (support.make_m), qfrc_smooth = passive - bias + actuator
  vfloat MO[8];      // columns 0..5 (root dofs) | own diagonal | qfrc_smooth
  vfloat MA[MAXD];   // in-row ancestors 1 .. MAXD - 1: M[i][ancestor's dof]
  vfloat AD[MAXD];   // ... and that dof's index (as a float), -1: the lane k below carries no dof
  {
    vfloat FD[6];
    {
      vfloat T[14];
      w.per_lane_n(T, [&](int l, float* o) {
        const int fl = m->rows.flags[l];
        const bool hinge = (fl & ROWS_DOF) != 0, tdof = (fl & ROWS_TDOF) != 0;
        const int i = (hinge || tdof) ? m->rows.dof[l] : 0;
        float crb[10], cfrc[6], cd[6], f[6];
        for (int k = 0; k < 10; k++) crb[k] = hinge ? lane_val(X[k], l) : XT[k];
        for (int k = 0; k < 6; k++) { cfrc[k] = hinge ? lane_val(X[10 + k], l) : XT[10 + k]; cd[k] = lane_val(CD[k], l); }
        dm::inert_mul(f, crb, cd);
        const float arm = m->dof_armature[i];
        for (int j = 0; j < 6; j++) {
          float v = 0.f;
          if (j < 3) v = f[3 + j];
          else for (int k = 0; k < 6; k++) v += f[k] * cdT[j - 3][k];
          o[j] = (tdof && j == i) ? v + arm : v;
        }
        float own = 0.f, bias = 0.f;
        for (int k = 0; k < 6; k++) { own += f[k] * cd[k]; bias += cd[k] * cfrc[k]; }
        o[6] = own + arm;
        const float passive = -m->dof_damping[i] * lane_val(QVL, l);
        const int a = m->dof_act[i];
        const int aa = a >= 0 ? a : 0;
        const float c0 = s.ctrl[aa], lo = m->act_ctrlrange[aa][0], hi = m->act_ctrlrange[aa][1], kp = m->act_kp[aa];
        const float qp = s.qpos[m->act_qposadr[aa]], gear = m->act_gear[aa];
        const float c = m->act_ctrllimited[aa] ? dm::clip(c0, lo, hi) : c0;
        const float force = m->act_isposition[aa] ? kp * (c - qp) : c;
        const float actf = a >= 0 ? gear * force : 0.f;
        o[7] = passive - bias + actf;
        for (int k = 0; k < 6; k++) o[8 + k] = f[k];
      });
      DIAL_UNROLL_FULL
      for (int k = 0; k < 8; k++) MO[k] = T[k];
      DIAL_UNROLL_FULL
      for (int k = 0; k < 6; k++) FD[k] = T[8 + k];
    }
    const vfloat DOFI = w.per_lane([&](int l) { return (m->rows.flags[l] & ROWS_JOINT) ? (float)m->rows.dof[l] : -1.f; });
    static_for<1, MAXD>([&](auto KK) {   // one ancestor at a time: six DPP fetches, one dot product
      constexpr int k = decltype(KK)::value;
      vfloat PA[6];
      DIAL_UNROLL_FULL
      for (int c = 0; c < 6; c++) PA[c] = w.template row_shr<k>(CD[c]);
      AD[k] = w.template row_shr<k>(DOFI);
      MA[k] = w.per_lane([&](int l) {
        float p = 0.f;
        for (int c = 0; c < 6; c++) p += lane_val(FD[c], l) * lane_val(PA[c], l);
        return p;
      });
    });
  }
  DIAL_MARK(w, 23);
  // ---- the dofs' outputs
  w.items(64, [&](int l) {
    const int fl = m->rows.flags[l], d = l & 15;
    const bool hinge = (fl & ROWS_DOF) != 0, tdof = (fl & ROWS_TDOF) != 0;
    if (hinge || tdof) {
      const int i = m->rows.dof[l];
      for (int k = 0; k < 3; k++) store2(s.cdof + 6 * i + 2 * k, lane_val(CD[2 * k], l), lane_val(CD[2 * k + 1], l));
      const float qf = lane_val(MO[7], l);
      s.qfs[i] = qf;
      s.rhs[i] = qf;
      if constexpr (!GEN) {
      if (tdof || !SR) {   // (under a welded root the chains have no root-dof columns: structural zeros)
        for (int j = 0; j < 6; j++) {
          if (j <= i) { const float v = lane_val(MO[j], l); s.M[i * S + j] = v; s.M[j * S + i] = v; }
        }
      }
      if (hinge) {
        s.M[i * S + i] = lane_val(MO[6], l);
        static_for<1, MAXD>([&](auto KK) {
          constexpr int k = decltype(KK)::value;
          const float ja = lane_val(AD[k], l);
          if (d - k >= 1 && ja >= 0.f) { const int j = (int)ja; const float v = lane_val(MA[k], l); s.M[i * S + j] = v; s.M[j * S + i] = v; }
        });
      }
      } else {   // packed lower triangle (an ancestor's dof index is smaller than the own)
        (void)S;
        for (int j = 0; j < 6; j++) {
          if (j <= i) s.M[tri_idx(i, j)] = lane_val(MO[j], l);
        }
        if (hinge) {
          s.M[tri_idx(i, i)] = lane_val(MO[6], l);
          static_for<1, MAXD>([&](auto KK) {
            constexpr int k = decltype(KK)::value;
            const float ja = lane_val(AD[k], l);
            if (d - k >= 1 && ja >= 0.f) s.M[tri_idx(i, (int)ja)] = lane_val(MA[k], l);
          });
        }
      }
    }
    if constexpr (GEN) { if (l == 15) solo_slide_body(m, s, M::D::NB - 1); }
  });
  if constexpr (GEN) {
    // ---- local_to_global for the geoms (forward(): the geom items of its frames phase), from the poses stored above
    w.items(M::D::NG, [&](int g) {
      const int b = m->geom_bodyid[g];
      const float q[4] = {s.xquat[4 * b], s.xquat[4 * b + 1], s.xquat[4 * b + 2], s.xquat[4 * b + 3]};
      const float gp[3] = {m->geom_pos[g][0], m->geom_pos[g][1], m->geom_pos[g][2]};
      const float gq[4] = {m->geom_quat[g][0], m->geom_quat[g][1], m->geom_quat[g][2], m->geom_quat[g][3]};
      float t3[3], qg[4], mat[9];
      dm::rotate(t3, gp, q);
      for (int k = 0; k < 3; k++) s.gpos[3 * g + k] = s.xpos[3 * b + k] + t3[k];
      dm::quat_mul(qg, q, gq);
      dm::quat_to_mat(mat, qg);
      s.gaxis[3 * g] = mat[2]; s.gaxis[3 * g + 1] = mat[5]; s.gaxis[3 * g + 2] = mat[8];
    });
  }
#if !defined(DIAL_EMU) && !defined(DIAL_QUAD_HOIST)
  w.lane = lane_keep;
#endif
  DIAL_MARK(w, 1);
}

}  // namespace dial

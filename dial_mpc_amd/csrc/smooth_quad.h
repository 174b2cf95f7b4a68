// smooth_quad.h -- the position / velocity stage of mjx.forward (smooth.kinematics, com_pos, com_vel, crb, rne, make_m,
// qfrc_smooth, the foot contacts) for a QUADRUPED topology, held in registers: no LDS hand-over between its sub-stages.
//
// rollout_body.h: forward() runs these sub-stages as ~14 LDS phases (bodies / dofs / chains -> lanes, results handed on
// through LDS).  On a lone wavefront that is 23 k of the Go2's 59 k cycles per env.step (profiles/r04_sections_unitree_go2_trot_
// cycles.txt): every phase starts with table look-ups and dependent LDS round trips, the chain walks and subtree sums are
// serial loops over LDS, and N = 2048 gives a SIMD two wavefronts -- the launch is as long as one rollout's dependence chain.
// For the topology "free trunk + legs of three one-hinge bodies" (TopoGo2) the tree fits the DPP network instead:
//
//   lane 16 r + d  (row r = leg r):  d = 0 the trunk (every row keeps its own copy), d = 1, 2, 3 hip, thigh, calf of leg r,
//                                    d = 4 .. 9 of row 0: the trunk's six dofs (rows of M and of qfrc_smooth only)
//
// so that "my parent" is the lane below (DPP row_shr:1), "my child" the lane above (row_shl:1) and the trunk is an ordinary
// v_readlane broadcast.  Root-to-leaf sweeps (poses, velocities, accelerations) are three rounds of row_shr + compose,
// leaf-to-root sums (composite inertia, body forces) two rounds of row_shl + add plus one wave reduction over the four hips.
// Every lane loads its own constants and state once, all arithmetic is register-to-register, and what later stages consume
// (xpos, xquat, site positions, com, cvel, cdof, M, qfrc_smooth, the four foot contacts) is stored at the end; xipos, ximat,
// cinert, cacc, cfl, cfrc, crb and F are never materialised.  Same formulas as forward() (each block names its counterpart);
// sums over bodies are taken in a different order (rounding level: covered by the oracle parity tests, not bit-identical to
// the phase version, which stays the implementation of every other robot and the reference -DDIAL_NO_QUAD builds compare to).
#pragma once
#include "derived.h"
#include "dmath.h"

#ifndef DIAL_NO_SITE_SHORTCUT
#define DIAL_NO_SITE_SHORTCUT 0   /* measurement switch: rotate the foot geom's centre even when it is the foot site's */
#endif

namespace dial {

template <class D>
inline constexpr bool kQuadDims = D::quad_stage;
template <class D>
inline constexpr bool kQuadGenDims = D::quad_gen;

// FUSED (the Go2's own instantiation): the four plane-sphere foot contacts, their Jacobian and the constraint rows come out of the
// stage, M in the square layout.  !FUSED (the generic feature set on the Go2's tree, Dims::quad_gen -- crate climb): bodies and
// dofs only, M as the packed lower triangle, then the geom frames of ALL geoms as one LDS phase; collisions, Jacobians and rows
// stay the generic feature set's (forward(), forward_constraints()).
template <bool FUSED = true, class W, class M>
DIAL_DEV void forward_smooth_quad(W& w, const M* m, const Ws& s) {
  constexpr int S = M::D::S;
  static_assert(M::D::NB >= 14 && M::D::NV == 18, "quadruped layout: trunk + 4 legs of 3");
  static_assert(!FUSED || (M::D::NB == 14 && M::D::NC == 4 && M::D::square), "fused foot contacts: the Go2's own scene");
  static_assert(FUSED || !M::D::square, "generic feature set: packed M");
  DIAL_MARK(w, 15);
#if !defined(DIAL_EMU) && !defined(DIAL_QUAD_HOIST)
  // An opaque copy of the lane id for this stage: its role masks and table addresses are loop invariants of the T-step loop,
  // and hoisted they are ~20 more VGPRs that live through the solver, which sits on the 168-register budget (ISA probe: 24
  // spilled VGPRs, 100 B of scratch); re-derived per step they cost a few dozen integer instructions.
  const int lane_keep = w.lane;
  { int lq = w.lane; asm volatile("" : "+v"(lq)); w.lane = lq; }
#endif
  // lane roles (derived from the lane id only: loop invariants)
  //   body lane:  d <= 3 (trunk copies d == 0; only lane 0 stores / counts the trunk)      body b, joint b - 1
  //   dof lane:   leg lanes (dof b + 4) and lanes 4 .. 9 (trunk dof d - 4)
  // ---- smooth.kinematics: local transforms (leg lanes), absolute pose (trunk lanes); forward(): kin_fast
  vfloat PL[14];   // [0..7): world pose pos(3) quat(4); [7..14): leg lanes' transform relative to the parent l_p(3) l_q(4)
  w.per_lane_n(PL, [&](int l, float* o) {
    const int d = l & 15, r = l >> 4;
    const bool leg = d >= 1 && d <= 3;
    const int b = leg ? 3 * r + d + 1 : 1, ji = b - 1, qa = leg ? b + 5 : 7;
    float tq[4] = {s.qpos[3], s.qpos[4], s.qpos[5], s.qpos[6]};
    dm::normalize4(tq);
    const int bflags = m->body_flags[b];
    float lq[4] = {m->body_quat[b][0], m->body_quat[b][1], m->body_quat[b][2], m->body_quat[b][3]};
    float lp[3] = {m->body_pos[b][0], m->body_pos[b][1], m->body_pos[b][2]};
    const float jp[3] = {m->jnt_pos[ji][0], m->jnt_pos[ji][1], m->jnt_pos[ji][2]};
    const float ja[3] = {m->jnt_axis[ji][0], m->jnt_axis[ji][1], m->jnt_axis[ji][2]};
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
    for (int k = 0; k < 3; k++) { o[k] = leg ? 0.f : s.qpos[k]; o[7 + k] = lp[k]; }
    for (int k = 0; k < 4; k++) { o[3 + k] = leg ? (k == 0 ? 1.f : 0.f) : tq[k]; o[10 + k] = qb[k]; }
  });
  // root-to-leaf: a lane at depth d is final after round d (its parent, the lane below, after round d - 1)
  DIAL_UNROLL_FULL
  for (int it = 0; it < 3; it++) {
    vfloat Q[7], N[7];
    DIAL_UNROLL_FULL
    for (int k = 0; k < 7; k++) Q[k] = w.template row_shr<1>(PL[k]);
    w.per_lane_n(N, [&](int l, float* o) {
      const int d = l & 15;
      const bool leg = d >= 1 && d <= 3;
      const float pp[3] = {lane_val(Q[0], l), lane_val(Q[1], l), lane_val(Q[2], l)};
      const float pq[4] = {lane_val(Q[3], l), lane_val(Q[4], l), lane_val(Q[5], l), lane_val(Q[6], l)};
      const float lp[3] = {lane_val(PL[7], l), lane_val(PL[8], l), lane_val(PL[9], l)};
      const float lq[4] = {lane_val(PL[10], l), lane_val(PL[11], l), lane_val(PL[12], l), lane_val(PL[13], l)};
      float pos[3], quat[4];
      dm::rotate(pos, lp, pq);
      for (int k = 0; k < 3; k++) pos[k] += pp[k];
      dm::quat_mul(quat, pq, lq);
      for (int k = 0; k < 3; k++) o[k] = leg ? pos[k] : lane_val(PL[k], l);
      for (int k = 0; k < 4; k++) o[3 + k] = leg ? quat[k] : lane_val(PL[3 + k], l);
    });
    DIAL_UNROLL_FULL
    for (int k = 0; k < 7; k++) PL[k] = N[k];
  }
  DIAL_MARK(w, 0);
  // ---- local_to_global: inertial frames (all bodies), the foot geom and site (calf lanes), the trunk's site (trunk lanes)
  vfloat F[22];   // xipos(3) ximat(9) | mass-weighted xipos(3), mass | site(3) | foot geom centre(3)
  const bool site_is_geom = FUSED && DM_UNIFORM_I(m->quad_site_is_geom) != 0 && !DIAL_NO_SITE_SHORTCUT;
  w.per_lane_n(F, [&](int l, float* o) {
    const int d = l & 15, r = l >> 4;
    const bool leg = d >= 1 && d <= 3, counted = leg || l == 0;
    const int b = leg ? 3 * r + d + 1 : 1, gs = d == 3 ? 1 + r : 0;
    const float p[3] = {lane_val(PL[0], l), lane_val(PL[1], l), lane_val(PL[2], l)};
    const float q[4] = {lane_val(PL[3], l), lane_val(PL[4], l), lane_val(PL[5], l), lane_val(PL[6], l)};
    const float ip[3] = {m->body_ipos[b][0], m->body_ipos[b][1], m->body_ipos[b][2]};
    const float iq[4] = {m->body_iquat[b][0], m->body_iquat[b][1], m->body_iquat[b][2], m->body_iquat[b][3]};
    const float sp[3] = {m->site_pos[gs][0], m->site_pos[gs][1], m->site_pos[gs][2]};
    const float mass = m->body_mass[b];
    float t3[3], qi[4], mat[9], ts[3], tg[3] = {0.f, 0.f, 0.f};
    dm::rotate(t3, ip, q);
    dm::quat_mul(qi, q, iq);
    dm::quat_to_mat(mat, qi);
    dm::rotate(ts, sp, q);
    if constexpr (FUSED) {
      if (site_is_geom) { tg[0] = ts[0]; tg[1] = ts[1]; tg[2] = ts[2]; }   // (the foot site IS the foot geom's centre: wave-uniform, one rotation less)
      else {
        const float gp[3] = {m->geom_pos[gs][0], m->geom_pos[gs][1], m->geom_pos[gs][2]};
        dm::rotate(tg, gp, q);
      }
    }
    for (int k = 0; k < 3; k++) {
      const float xi = p[k] + t3[k];
      o[k] = xi;
      o[12 + k] = counted ? xi * mass : 0.f;
      o[16 + k] = p[k] + ts[k];
      o[19 + k] = p[k] + tg[k];
    }
    for (int k = 0; k < 9; k++) o[3 + k] = mat[k];
    o[15] = counted ? mass : 0.f;
  });
  DIAL_MARK(w, 16);
  // ---- smooth.com_pos: one wave reduction (the 13 bodies form one kinematic tree)
  float com[3];
  {
    vfloat c4[4] = {F[12], F[13], F[14], F[15]};
    float r4[4];
    w.vsumN(c4, r4);
    for (int k = 0; k < 3; k++) com[k] = r4[3] < MJ_MINVAL ? bcast(F[k], 0) : r4[k] / r4[3];
  }
  // the trunk's pose, rotation matrix and rotational cdofs: the same value in every lane
  const float tpos[3] = {bcast(PL[0], 0), bcast(PL[1], 0), bcast(PL[2], 0)};
  const float tquat[4] = {bcast(PL[3], 0), bcast(PL[4], 0), bcast(PL[5], 0), bcast(PL[6], 0)};
  float Rt[9], cdT[3][6];
  dm::quat_to_mat(Rt, tquat);
  const float offt[3] = {com[0] - tpos[0], com[1] - tpos[1], com[2] - tpos[2]};
  for (int i = 0; i < 3; i++) {
    const float a[3] = {Rt[i], Rt[3 + i], Rt[6 + i]};
    float cr[3];
    dm::cross3(cr, a, offt);
    for (int k = 0; k < 3; k++) { cdT[i][k] = a[k]; cdT[i][3 + k] = cr[k]; }
  }
  DIAL_MARK(w, 17);
  // ---- cinert (body lanes) and cdof (dof lanes)
  vfloat X[16];   // cinert(10) | local force cfl(6): the quantities summed over subtrees
  vfloat CD[6];
  {
    vfloat T[16];
    w.per_lane_n(T, [&](int l, float* o) {
      const int d = l & 15, r = l >> 4;
      const bool leg = d >= 1 && d <= 3, body = d <= 3, tdof = d >= 4 && d <= 9;   // (trunk dofs: in every row, for its contact's Jacobian)
      const int b = leg ? 3 * r + d + 1 : 1, ji = b - 1, kd = tdof ? d - 4 : 3;
      const float R[9] = {lane_val(F[3], l), lane_val(F[4], l), lane_val(F[5], l), lane_val(F[6], l), lane_val(F[7], l),
                          lane_val(F[8], l), lane_val(F[9], l), lane_val(F[10], l), lane_val(F[11], l)};
      const float off[3] = {lane_val(F[0], l) - com[0], lane_val(F[1], l) - com[1], lane_val(F[2], l) - com[2]};
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
      // cdof: hinge = [axis, axis x (com - anchor)]; trunk dof k < 3: [0, e_k], k >= 3: column k - 3 of the trunk's xmat
      const float bq[4] = {lane_val(PL[3], l), lane_val(PL[4], l), lane_val(PL[5], l), lane_val(PL[6], l)};
      const float jp[3] = {m->jnt_pos[ji][0], m->jnt_pos[ji][1], m->jnt_pos[ji][2]};
      const float ja[3] = {m->jnt_axis[ji][0], m->jnt_axis[ji][1], m->jnt_axis[ji][2]};
      float anchor[3], jaxis[3], cr[3];
      if (m->body_flags[b] & 2) { anchor[0] = 0.f; anchor[1] = 0.f; anchor[2] = 0.f; }
      else dm::rotate(anchor, jp, bq);
      for (int k = 0; k < 3; k++) anchor[k] += lane_val(PL[k], l);
      dm::rotate(jaxis, ja, bq);
      const float offj[3] = {com[0] - anchor[0], com[1] - anchor[1], com[2] - anchor[2]};
      dm::cross3(cr, jaxis, offj);
      // (the trunk's rotational cdofs are blended with 0 / 1 weights, not selected: a select chain over the elements of a local
      //  array is turned into a dynamically indexed scratch array by the optimiser -- 80 B of private memory in the ISA)
      const float w0 = kd == 3 ? 1.f : 0.f, w1 = kd == 4 ? 1.f : 0.f, w2 = kd == 5 ? 1.f : 0.f;
      for (int k = 0; k < 3; k++) {
        const float ta = w0 * cdT[0][k] + w1 * cdT[1][k] + w2 * cdT[2][k];
        const float tl = (k == kd ? 1.f : 0.f) + (w0 * cdT[0][3 + k] + w1 * cdT[1][3 + k] + w2 * cdT[2][3 + k]);
        o[10 + k] = leg ? jaxis[k] : (tdof ? ta : 0.f);
        o[13 + k] = leg ? cr[k] : (tdof ? tl : 0.f);
      }
    });
    DIAL_UNROLL_FULL
    for (int k = 0; k < 10; k++) X[k] = T[k];
    DIAL_UNROLL_FULL
    for (int k = 0; k < 6; k++) CD[k] = T[10 + k];
  }
  DIAL_MARK(w, 18);
  // ---- smooth.com_vel + cdof_dot + rne forward: the trunk's velocity / acceleration (every lane the same), then root-to-leaf
  vfloat VA[12];   // cvel(6) | cacc(6)
  {
    const float qv[6] = {s.qvel[0], s.qvel[1], s.qvel[2], s.qvel[3], s.qvel[4], s.qvel[5]};
    const float vs[6] = {0.f, 0.f, 0.f, qv[0], qv[1], qv[2]};   // the rotational dofs see the velocity after the translational ones
    float velT[6] = {0.f, 0.f, 0.f, qv[0], qv[1], qv[2]};
    float accT[6] = {0.f, 0.f, 0.f, -m->gravity[0], -m->gravity[1], -m->gravity[2]};
    for (int j = 0; j < 3; j++) {
      // cdof_dot of the trunk's rotational dof j = motion_cross(vs, cdof): vs has no angular part, so it is (0, vs.lin x cdof.ang).
      // (Spelled out: handed the explicit zeros the compiler kept their SIGNS alive -- fifteen v_bfi copysigns and a dozen
      //  subtractions of signed zeros per step, in every lane.)
      float cl[3];
      dm::cross3(cl, vs + 3, cdT[j]);
      for (int k = 0; k < 3; k++) accT[3 + k] += cl[k] * qv[3 + j];
      for (int k = 0; k < 6; k++) velT[k] += cdT[j][k] * qv[3 + j];
    }
    w.per_lane_n(VA, [&](int l, float* o) {
      for (int k = 0; k < 6; k++) { o[k] = velT[k]; o[6 + k] = accT[k]; }   // the trunk's, in EVERY lane: the sums below start from it
    });
  }
  const vfloat QVL = w.per_lane([&](int l) {
    const int d = l & 15, r = l >> 4;
    const bool leg = d >= 1 && d <= 3, tdof = r == 0 && d >= 4 && d <= 9;
    return s.qvel[leg ? 3 * r + d + 5 : (tdof ? d - 4 : 0)];   // this dof lane's velocity
  });
  // The recurrences cvel_d = cvel_{d-1} + cdof_d qvel_d, cdof_dot_d = cvel_{d-1} x cdof_d, cacc_d = cacc_{d-1} + cdof_dot_d qvel_d down a
  // leg of depth 3.  Round 6, second form: a lane at depth d needs trunk + P_{d-2} + P_{d-1} + P_d with P = cdof qvel of the lanes below
  // it in its row -- the products are formed ONCE (zero outside the leg lanes) and added through two DPP-operand additions (row_shr:2,
  // row_shr:1: both read the products themselves, no round waits for the one before) and one multiply-add, in the order the recurrence
  // adds them: 25 vector instructions per sweep where three rounds of shift + multiply-add + select were 54.  The ancestors' products
  // are rounded before they are added (they were fused into their own lanes' sums): rounding level, covered by the oracle parity tests.
  {
    const vfloat QL = w.per_lane([&](int l) { const int d = l & 15; return (d >= 1 && d <= 3) ? lane_val(QVL, l) : 0.f; });
    const auto chain_sum = [&](vfloat* acc, const vfloat* cd) {   // acc[k] (the trunk's value in every lane) -> the body's, leg lanes
      vfloat P[6], P1[6], P2[6];
      DIAL_UNROLL_FULL
      for (int k = 0; k < 6; k++) P[k] = cd[k] * QL;
      DIAL_UNROLL_FULL
      for (int k = 0; k < 6; k++) { P2[k] = w.template row_shr<2>(P[k]); P1[k] = w.template row_shr<1>(P[k]); }
      DIAL_UNROLL_FULL
      for (int k = 0; k < 6; k++) acc[k] = vfma(cd[k], QL, (acc[k] + P2[k]) + P1[k]);
    };
    chain_sum(VA, CD);
    vfloat VP[6], CDD[6];   // the parent's (final) velocity; cdof_dot
    DIAL_UNROLL_FULL
    for (int k = 0; k < 6; k++) VP[k] = w.template row_shr<1>(VA[k]);
    w.per_lane_n(CDD, [&](int l, float* o) {
      const float vp[6] = {lane_val(VP[0], l), lane_val(VP[1], l), lane_val(VP[2], l), lane_val(VP[3], l), lane_val(VP[4], l), lane_val(VP[5], l)};
      const float cd[6] = {lane_val(CD[0], l), lane_val(CD[1], l), lane_val(CD[2], l), lane_val(CD[3], l), lane_val(CD[4], l), lane_val(CD[5], l)};
      dm::motion_cross(o, vp, cd);
    });
    chain_sum(VA + 6, CDD);
  }
  // the bodies' outputs are complete: stored now (fewer registers to carry through the rest of the stage)
  // (Dims::pre_ctrl rollouts: this control step's x.pos row goes to HBM from here -- Wave::out_io -- not from a phase of its own)
  float* xrow = nullptr;
  if constexpr (M::D::pre_ctrl) { if (w.out_io && w.out_io->xss) xrow = w.out_io->xss + (size_t)w.out_row * ((M::D::NB - 1) * 3); }
  w.items(64, [&](int l) {
    const int d = l & 15, r = l >> 4;
    const bool leg = d >= 1 && d <= 3;
    const int b = leg ? 3 * r + d + 1 : 1;
    if (leg || l == 0) {
      for (int k = 0; k < 3; k++) s.xpos[3 * b + k] = lane_val(PL[k], l);
      if constexpr (M::D::pre_ctrl) { if (xrow) for (int k = 0; k < 3; k++) xrow[3 * (b - 1) + k] = lane_val(PL[k], l); }
      store4(s.xquat + 4 * b, lane_val(PL[3], l), lane_val(PL[4], l), lane_val(PL[5], l), lane_val(PL[6], l));
      for (int k = 0; k < 3; k++) store2(s.cvel + 6 * b + 2 * k, lane_val(VA[2 * k], l), lane_val(VA[2 * k + 1], l));
    }
    if (l == 0) {
      for (int k = 0; k < 3; k++) { s.com[3 * m->body_rootid[1] + k] = com[k]; s.spos[k] = lane_val(F[16 + k], l); }
      for (int k = 0; k < 4; k++) s.qpos[3 + k] = tquat[k];   // (kinematics normalises the free joint's quaternion in place)
    }
    if (d == 3) for (int k = 0; k < 3; k++) s.spos[3 * (1 + r) + k] = lane_val(F[16 + k], l);
    if constexpr (!FUSED) {   // packed M: the structural zeros (the dof lanes below write the tree's entries only)
      for (int e = l; e < M::D::NTRI; e += 64) s.M[e] = 0.f;
    }
  });
  // ---- collision_driver (the four plane-sphere foot contacts), the contact Jacobian and constraint.make_constraint, fused:
  // forward_constraints()'s first two LDS phases (72 (contact, dof) items; 28 rows with an 18-term J qvel each) become
  //   * calf lanes: distance, contact point, the four pyramid rows' D and aref -- their velocity is the contact POINT's
  //     velocity (J qvel = cvel.lin + cvel.ang x (p - com), the body velocity this lane holds) projected on the frame;
  //   * dof lanes: the contact of their own row only (a leg dof moves no other foot; lanes 4..9 of EVERY row stand for the
  //     trunk's dofs against that row's contact): one 16-byte store of J^T[i][4c..4c+3]; the other entries of a leg dof's row
  //     are structural zeros written once per kernel (init_quad);
  //   * leg lanes: their joint's limit row.
  if constexpr (FUSED) {
    const float pn[3] = {m->geom0_normal[0], m->geom0_normal[1], m->geom0_normal[2]};   // (the floor's normal: a model constant, derived.h)
    const float fr[9] = {s.cframe[0], s.cframe[1], s.cframe[2], s.cframe[3], s.cframe[4], s.cframe[5], s.cframe[6], s.cframe[7], s.cframe[8]};
    vfloat CP[4];   // contact point (3), distance
    w.per_lane_n(CP, [&](int l, float* o) {
      const int r = l >> 4, g2 = 1 + r;
      const float ctr[3] = {lane_val(F[19], l), lane_val(F[20], l), lane_val(F[21], l)};
      const float radius = m->geom_size[g2][0];
      const float diff[3] = {ctr[0] - m->geom_pos[0][0], ctr[1] - m->geom_pos[0][1], ctr[2] - m->geom_pos[0][2]};
      const float dist = dm::dot3(diff, pn) - radius;
      for (int k = 0; k < 3; k++) o[k] = ctr[k] - pn[k] * (radius + 0.5f * dist);
      o[3] = dist;
    });
    vfloat PC[3];
    DIAL_UNROLL_FULL
    for (int k = 0; k < 3; k++) PC[k] = w.template row_bcast<3>(CP[k]);   // the calf lane's contact point, to its whole row
    // (three compute-and-store phases, each with a short live range: one fused phase cost the 128-register large-batch build 15 spills)
    w.items(64, [&](int l) {   // -- Jacobian of contact r with respect to this lane's dof (world vs the calf: jacp_b2 only)
      const int d = l & 15, r = l >> 4;
      const bool leg = d >= 1 && d <= 3, tk = d >= 4 && d <= 9;
      if (!(leg || tk)) return;
      const int i = leg ? 3 * r + d + 5 : d - 4;
      const float mu1 = m->con_friction[r][0], mu2 = m->con_friction[r][1];
      const float cd[6] = {lane_val(CD[0], l), lane_val(CD[1], l), lane_val(CD[2], l), lane_val(CD[3], l), lane_val(CD[4], l), lane_val(CD[5], l)};
      const float off[3] = {lane_val(PC[0], l) - com[0], lane_val(PC[1], l) - com[1], lane_val(PC[2], l) - com[2]};
      float cr[3];
      dm::cross3(cr, cd, off);
      const float diff[3] = {cd[3] + cr[0], cd[4] + cr[1], cd[5] + cr[2]};
      const float jn = dm::dot3(fr, diff), t1 = dm::dot3(fr + 3, diff) * mu1, t2 = dm::dot3(fr + 6, diff) * mu2;
      store4(s.Jc + i * M::D::T + 4 * r, jn + t1, jn - t1, jn + t2, jn - t2);
    });
    w.items(64, [&](int l) {   // -- the contact's four pyramid rows (calf lanes)
      const int d = l & 15, r = l >> 4;
      if (d != 3) return;
      const float mu1 = m->con_friction[r][0], mu2 = m->con_friction[r][1];
      const float pos = lane_val(CP[3], l) - m->con_margin[r];
      const float invweight = m->con_invw[r][0];   // (_efc_contact_pyramidal's row weight: a model constant, derived.h)
      float k_, b_, imp;
      kbi(m->kbi_tab[m->con_kbi[r]], pos, k_, b_, imp);
      const float Rr = dm::fmaxf_(invweight * (1.f - imp) / imp, MJ_MINVAL);
      const float va[3] = {lane_val(VA[0], l), lane_val(VA[1], l), lane_val(VA[2], l)};
      const float off[3] = {lane_val(CP[0], l) - com[0], lane_val(CP[1], l) - com[1], lane_val(CP[2], l) - com[2]};
      float cr[3];
      dm::cross3(cr, va, off);
      const float vp[3] = {lane_val(VA[3], l) + cr[0], lane_val(VA[4], l) + cr[1], lane_val(VA[5], l) + cr[2]};
      const float vn = dm::dot3(fr, vp), v1 = dm::dot3(fr + 3, vp) * mu1, v2 = dm::dot3(fr + 6, vp) * mu2;
      const bool on = pos < 0.f;
      const float dd = on ? 1.f / Rr : 0.f, ar = -k_ * imp * pos;
      constexpr int NL = M::D::NL;
      store4(s.D + NL + 4 * r, dd, dd, dd, dd);
      store4(s.aref + NL + 4 * r, on ? ar - b_ * (vn + v1) : 0.f, on ? ar - b_ * (vn - v1) : 0.f, on ? ar - b_ * (vn + v2) : 0.f, on ? ar - b_ * (vn - v2) : 0.f);
      s.cdist[r] = lane_val(CP[3], l);
      for (int k = 0; k < 3; k++) s.cpos[3 * r + k] = lane_val(CP[k], l);
    });
    w.items(64, [&](int l) {   // -- the joint's limit row (leg lanes)
      const int d = l & 15, r = l >> 4;
      if (!(d >= 1 && d <= 3)) return;
      const int b = 3 * r + d + 1, ji = b - 1, i = b + 4, qa = b + 5, lr = m->dof_limrow[i];
      if (lr < 0) return;
      const float q = s.qpos[qa];
      const float dist_min = q - m->jnt_range[ji][0], dist_max = m->jnt_range[ji][1] - q;
      const float pos = dm::fminf_(dist_min, dist_max) - m->jnt_margin[ji];
      const float sgn = dist_min < dist_max ? 1.f : -1.f;
      float k_, b_, imp;
      kbi(m->kbi_tab[m->jnt_kbi[ji]], pos, k_, b_, imp);
      const float Rr = dm::fmaxf_(m->dof_invweight0[i] * (1.f - imp) / imp, MJ_MINVAL);
      const bool on = pos < 0.f;
      s.lsign[lr] = sgn;
      s.D[lr] = on ? 1.f / Rr : 0.f;
      s.aref[lr] = on ? -b_ * (sgn * lane_val(QVL, l)) - k_ * imp * pos : 0.f;
    });
  }
  DIAL_MARK(w, 19);
  // ---- rne: local body forces cfl = cinert cacc + cvel x* (cinert cvel)
  {
    vfloat T[6];
    w.per_lane_n(T, [&](int l, float* o) {
      const bool body = (l & 15) <= 3;
      float ci[10], ca[6], cv[6], f1[6], f2[6], f3[6];
      for (int k = 0; k < 10; k++) ci[k] = lane_val(X[k], l);
      for (int k = 0; k < 6; k++) { cv[k] = lane_val(VA[k], l); ca[k] = lane_val(VA[6 + k], l); }
      dm::inert_mul(f1, ci, ca);
      dm::inert_mul(f2, ci, cv);
      dm::motion_cross_force(f3, cv, f2);
      for (int k = 0; k < 6; k++) o[k] = body ? f1[k] + f3[k] : 0.f;
    });
    DIAL_UNROLL_FULL
    for (int k = 0; k < 6; k++) X[10 + k] = T[k];
  }
  // ---- subtree sums (smooth.crb, rne backward): leaf-to-root along the legs, then the trunk = own + the four hips
  // Round 6, second form.  The lanes above a calf hold zeros, so with T = X + shl1(X) (thigh + calf in the thigh lane, the calf's own in
  // the calf lane) R = X + shl1(T) is hip + (thigh + calf), thigh + calf, calf in the three leg lanes -- the association of the two
  // masked rounds this replaces, as 32 DPP-operand additions instead of 2 x 16 x (shift, add, select).  (Same sums on paper; the outputs
  // of the two builds differ in the last bits of the laterally symmetric components -- tools/gpu/bit_compare_steps.py -- so: rounding level.)  The trunk
  // copies (d = 0) and the lanes past the calf end up with sums nobody reads: the trunk's own terms are taken first.
  float X0[16];   // the trunk's own terms, taken before the sums below overwrite lane 0
  for (int k = 0; k < 16; k++) X0[k] = bcast(X[k], 0);
  {
    vfloat T[16];
    DIAL_UNROLL_FULL
    for (int k = 0; k < 16; k++) T[k] = X[k] + w.template row_shl<1>(X[k]);
    DIAL_UNROLL_FULL
    for (int k = 0; k < 16; k++) X[k] = X[k] + w.template row_shl<1>(T[k]);
  }
  float XT[16];   // the trunk's composite inertia and subtree force
  {
    vfloat H[16];
    w.per_lane_n(H, [&](int l, float* o) {
      const bool hip = (l & 15) == 1;
      for (int k = 0; k < 16; k++) o[k] = hip ? lane_val(X[k], l) : 0.f;
    });
    float hs[16];
    w.vsumN(H, hs);
    for (int k = 0; k < 16; k++) XT[k] = X0[k] + hs[k];
  }
  DIAL_MARK(w, 22);
  // ---- F_i = crb cdof_i, M = F . cdof over the ancestors (support.make_m), qfrc_smooth = passive - bias + actuator
  vfloat MO[11];   // columns 0..5 (trunk dofs) | own diagonal | parent dof | grandparent dof | qfrc_smooth | (unused)
  {
    vfloat P1[6], P2[6];
    DIAL_UNROLL_FULL
    for (int k = 0; k < 6; k++) { P1[k] = w.template row_shr<1>(CD[k]); P2[k] = w.template row_shr<2>(CD[k]); }
    w.per_lane_n(MO, [&](int l, float* o) {
      const int d = l & 15, r = l >> 4;
      const bool leg = d >= 1 && d <= 3, tdof = r == 0 && d >= 4 && d <= 9;
      const int b = leg ? 3 * r + d + 1 : 1, i = leg ? b + 4 : (tdof ? d - 4 : 0);
      float crb[10], cfrc[6], cd[6], f[6];
      for (int k = 0; k < 10; k++) crb[k] = leg ? lane_val(X[k], l) : XT[k];
      for (int k = 0; k < 6; k++) { cfrc[k] = leg ? lane_val(X[10 + k], l) : XT[10 + k]; cd[k] = lane_val(CD[k], l); }
      dm::inert_mul(f, crb, cd);
      const float arm = m->dof_armature[i];
      for (int j = 0; j < 6; j++) {
        float v = 0.f;
        if (j < 3) v = f[3 + j];
        else for (int k = 0; k < 6; k++) v += f[k] * cdT[j - 3][k];
        o[j] = (tdof && j == i) ? v + arm : v;
      }
      float own = 0.f, p1 = 0.f, p2 = 0.f, bias = 0.f;
      for (int k = 0; k < 6; k++) {
        own += f[k] * cd[k];
        p1 += f[k] * lane_val(P1[k], l);
        p2 += f[k] * lane_val(P2[k], l);
        bias += cd[k] * cfrc[k];
      }
      o[6] = own + arm; o[7] = p1; o[8] = p2;
      const float passive = -m->dof_damping[i] * lane_val(QVL, l);
      // (every operand fetched unconditionally: inside `if (ctrllimited)` / `if (isposition)` the fetches were four dependent
      //  LDS round trips under exec masks)
      const int a = m->dof_act[i];
      const int aa = a >= 0 ? a : 0;
      float c0 = s.ctrl[aa];
      const float lo = m->act_ctrlrange[aa][0], hi = m->act_ctrlrange[aa][1], kp = m->act_kp[aa];
      if constexpr (M::D::pre_ctrl) {
        if (w.jrow) {   // (rollouts: act2tau by the actuated dof's own lane, from the table's joint target -- no ctrl phase, no LDS round trip)
          const float jt = w.jrow[aa];
          if (m->position_control) c0 = jt;
          else {
            const float q_err = jt - s.qpos[7 + aa];
            c0 = dm::clip(m->kp[aa] * q_err - m->kd[aa] * s.qvel[6 + aa], m->tau_range[aa][0], m->tau_range[aa][1]);
          }
          if (a >= 0) s.ctrl[aa] = c0;   // (read by the seq-jump task's info update)
        }
      }
      const float qp = s.qpos[m->act_qposadr[aa]], gear = m->act_gear[aa];
      const float c = m->act_ctrllimited[aa] ? dm::clip(c0, lo, hi) : c0;
      const float force = m->act_isposition[aa] ? kp * (c - qp) : c;
      const float actf = a >= 0 ? gear * force : 0.f;
      o[9] = passive - bias + actf;
      o[10] = 0.f;
    });
  }
  DIAL_MARK(w, 23);
  // ---- the dofs' outputs
  w.items(64, [&](int l) {
    const int d = l & 15, r = l >> 4;
    const bool leg = d >= 1 && d <= 3, tdof = r == 0 && d >= 4 && d <= 9;
    const int b = leg ? 3 * r + d + 1 : 1;
    if (leg || tdof) {
      const int i = leg ? b + 4 : d - 4;
      for (int k = 0; k < 3; k++) store2(s.cdof + 6 * i + 2 * k, lane_val(CD[2 * k], l), lane_val(CD[2 * k + 1], l));
      const float qf = lane_val(MO[9], l);
      s.qfs[i] = qf;
      s.rhs[i] = qf;
      // (predicated stores under branches: redirecting the lanes that are off to a dump word instead measured the same,
      //  profiles/r04_ab_smooth_quad.txt)
      if constexpr (M::D::square) {
        for (int j = 0; j < 6; j++) {
          if (j <= i) { const float v = lane_val(MO[j], l); s.M[i * S + j] = v; s.M[j * S + i] = v; }
        }
        if (leg) {
          s.M[i * S + i] = lane_val(MO[6], l);
          if (d >= 2) { const float v = lane_val(MO[7], l); s.M[i * S + i - 1] = v; s.M[(i - 1) * S + i] = v; }
          if (d == 3) { const float v = lane_val(MO[8], l); s.M[i * S + i - 2] = v; s.M[(i - 2) * S + i] = v; }
        }
      } else {   // packed lower triangle (generic feature set)
        (void)S;
        for (int j = 0; j < 6; j++) {
          if (j <= i) s.M[tri_idx(i, j)] = lane_val(MO[j], l);
        }
        if (leg) {
          s.M[tri_idx(i, i)] = lane_val(MO[6], l);
          if (d >= 2) s.M[tri_idx(i, i - 1)] = lane_val(MO[7], l);
          if (d == 3) s.M[tri_idx(i, i - 2)] = lane_val(MO[8], l);
        }
      }
    }
  });
  if constexpr (!FUSED) {
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

// Once per kernel: what the stage above never rewrites -- the contact frames of the four plane-sphere contacts
// (collision_primitive make_frame(plane normal)), the structural zeros of the contact Jacobian, lsign of the contact rows.
template <class W, class M>
DIAL_DEV void init_quad(W& w, const M* m, const Ws& s) {
  w.items(M::D::NV * M::D::T, [&](int e) { s.Jc[e] = 0.f; });                 // J^T: a leg's dofs move no other leg's foot
  w.items(4 * M::D::NC, [&](int e) { s.lsign[M::D::NL + e] = 0.f; });        // (the solver reads a contact row's lsign as its zero word)
  w.items(M::D::NC, [&](int c) {
    const float gq[4] = {m->geom_quat[0][0], m->geom_quat[0][1], m->geom_quat[0][2], m->geom_quat[0][3]};
    float mat[9], fr[9];
    dm::quat_to_mat(mat, gq);
    const float n[3] = {mat[2], mat[5], mat[8]};
    make_frame(fr, n);
    for (int k = 0; k < 9; k++) s.cframe[9 * c + k] = fr[k];
  });
}

// Once per kernel, generic feature set on the quadruped tree: the poses of the bodies welded to the world (the crate), which no
// lane of the stage computes.
template <class W, class M>
DIAL_DEV void init_quad_gen(W& w, const M* m, const Ws& s) {
  w.items(M::D::NB - 14, [&](int e) {
    const int b = 14 + e;
    for (int k = 0; k < 3; k++) s.xpos[3 * b + k] = m->body_pos[b][k];
    for (int k = 0; k < 4; k++) s.xquat[4 * b + k] = m->body_quat[b][k];
  });
}

}  // namespace dial

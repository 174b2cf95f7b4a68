/*
 * dial_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE ONLY, never on the product path).
 *
 * A plain-C, sequential, dense restatement of the reference's DIAL-MPC inner loop:
 *   planner   : dial_mpc/core/dial_core.py:36-48,103-145,160-166   (reverse_once, rollout_us, shift)
 *   control   : dial_mpc/envs/base_env.py:38-66                     (act2joint, act2tau)
 *   env.step  : dial_mpc/envs/unitree_go2_env.py:126-261 (walk), :403-521 (seq_jump),
 *               dial_mpc/envs/unitree_h1_env.py:181-321 (H1 walk), :696-858 (H1 loco),
 *               dial_mpc/envs/manipulation.py:63-115 (Allegro reorient + act2joint override)
 *   helpers   : dial_mpc/utils/function_utils.py:7-43               (inv_rotate, get_foot_step)
 *   x / xd    : dial_mpc/deploy/dial_plan.py:45-61                  (the reference's own copy of
 *               brax.mjx.pipeline's derivation of x, xd from mjx.Data)
 * The physics inside `pipeline_step` lives in THIRD-PARTY code that is not vendored in the
 * reference and not installable here: brax (unpinned) -> mujoco.mjx (unpinned, setup.py:9-21).
 * It is restated below from the published MJX algorithm (function names in comments are the
 * MJX ones: smooth.kinematics, smooth.com_pos, smooth.crb, smooth.factor_m, collision_driver,
 * constraint.make_constraint, smooth.com_vel, passive, smooth.rne, forward.fwd_actuation,
 * forward.fwd_acceleration, solver.solve, forward.euler).
 *
 * PARITY STATUS: **parity unpinned** against the JAX reference -- the reference ships no
 * golden vectors/tests and cannot run in the build container.  What IS pinned:
 * the spline matrices (SciPy FITPACK), get_foot_step / noise-schedule KATs, model constants vs
 * SURVEY D, physics invariants, fp32-vs-fp64 self-consistency (tests/test_oracle_*.py).
 *
 * Build: see oracle/Makefile.  -DREAL=float -> liboracle_f32.so, -DREAL=double -> _f64.so.
 * Every exported array argument is `real*` (float32 or float64 according to the flavour).
 */
#include "../include/dial_mpc.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL double
#endif
typedef REAL real;
/* tools/opcount builds this file as C++ with REAL = an operation-counting type (FLOP count of the dense formulation) */
#ifdef DIAL_OPCOUNT
#define OPC(field) (g_ops.field++)
#else
#define OPC(field) ((void)0)
#endif

#define NB DIAL_MAX_BODY
#define NJ DIAL_MAX_JNT
#define NQ DIAL_MAX_Q
#define NV DIAL_MAX_V
#define NU DIAL_MAX_U
#define NG DIAL_MAX_GEOM
#define NS DIAL_MAX_SITE
#define NC DIAL_MAX_CON
#define NE DIAL_MAX_EFC

#define MJ_MINVAL ((real)1e-15)
#define MJ_MINIMP ((real)0.0001)
#define MJ_MAXIMP ((real)0.9999)
#define R_PI ((real)3.14159265358979323846)

static inline real r_sqrt(real x) { OPC(sqrt_); return (real)sqrt((double)x); }
static inline real r_sin(real x) { OPC(trans); return sizeof(real) == 4 ? (real)sinf((float)x) : (real)sin((double)x); }
static inline real r_cos(real x) { OPC(trans); return sizeof(real) == 4 ? (real)cosf((float)x) : (real)cos((double)x); }
static inline real r_atan2(real y, real x) { OPC(trans); return sizeof(real) == 4 ? (real)atan2f((float)y, (float)x) : (real)atan2((double)y, (double)x); }
static inline real r_asin(real x) { OPC(trans); return sizeof(real) == 4 ? (real)asinf((float)x) : (real)asin((double)x); }
static inline real r_pow(real x, real y) { OPC(trans); return sizeof(real) == 4 ? (real)powf((float)x, (float)y) : (real)pow((double)x, (double)y); }
static inline real r_abs(real x) { return x < 0 ? -x : x; }
static inline real r_min(real a, real b) { return a < b ? a : b; }
static inline real r_max(real a, real b) { return a > b ? a : b; }
static inline real r_clip(real x, real lo, real hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline real r_floor(real x) { return (real)floor((double)x); }
static inline real r_fma(real a, real b, real c) { OPC(mul); OPC(add); return sizeof(real) == 4 ? (real)fmaf((float)a, (float)b, (float)c) : (real)fma((double)a, (double)b, (double)c); }

/* ------------------------------------------------------------------ per-sample data (mjx.Data) */
typedef struct {
  real qpos[NQ], qvel[NV], qacc_warmstart[NV], ctrl[NU];
  real xpos[NB][3], xquat[NB][4], xmat[NB][9], xipos[NB][3], ximat[NB][9];
  real xanchor[NJ][3], xaxis[NJ][3];
  real geom_xpos[NG][3], geom_xmat[NG][9], site_xpos[NS][3];
  real subtree_com[NB][3], cinert[NB][10], cdof[NV][6], cvel[NB][6], cdof_dot[NV][6];
  real qM[NV][NV], qL[NV][NV];
  real con_dist[NC], con_pos[NC][3], con_frame[NC][9];
  real efc_J[NE][NV], efc_D[NE], efc_aref[NE], efc_force[NE];
  real efc_floss[NE];   /* frictionloss of a dry-friction row, 0 for every other row */
  int efc_on[NE];
  real qfrc_passive[NV], qfrc_bias[NV], qfrc_actuator[NV], qfrc_smooth[NV], qacc_smooth[NV];
  real qacc[NV], qfrc_constraint[NV];
  int solver_niter;
} odata;

/* ------------------------------------------------------------------ small math (mjx/_src/math.py) */
static void quat_mul(real* o, const real* u, const real* v) {
  real r0 = u[0] * v[0] - u[1] * v[1] - u[2] * v[2] - u[3] * v[3];
  real r1 = u[0] * v[1] + u[1] * v[0] + u[2] * v[3] - u[3] * v[2];
  real r2 = u[0] * v[2] - u[1] * v[3] + u[2] * v[0] + u[3] * v[1];
  real r3 = u[0] * v[3] + u[1] * v[2] - u[2] * v[1] + u[3] * v[0];
  o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
}
static void cross3(real* o, const real* a, const real* b) {
  real r0 = a[1] * b[2] - a[2] * b[1], r1 = a[2] * b[0] - a[0] * b[2], r2 = a[0] * b[1] - a[1] * b[0];
  o[0] = r0; o[1] = r1; o[2] = r2;
}
static real dot3(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
/* math.rotate: r = 2(u.v)u + (s^2-u.u)v + 2s(u x v) */
static void rotate(real* o, const real* vec, const real* q) {
  real s = q[0];
  const real* u = q + 1;
  real ud = dot3(u, vec), uu = dot3(u, u), c[3];
  cross3(c, u, vec);
  real r[3];
  for (int i = 0; i < 3; i++) r[i] = 2 * (ud * u[i]) + (s * s - uu) * vec[i] + 2 * s * c[i];
  o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
}
static void inv_rotate(real* o, const real* vec, const real* q) {
  real qc[4] = {q[0], -q[1], -q[2], -q[3]};
  rotate(o, vec, qc);
}
static void quat_to_mat(real* m, const real* q) {
  real w = q[0], x = q[1], y = q[2], z = q[3];
  m[0] = w * w + x * x - y * y - z * z; m[1] = 2 * (x * y - w * z); m[2] = 2 * (x * z + w * y);
  m[3] = 2 * (x * y + w * z); m[4] = w * w - x * x + y * y - z * z; m[5] = 2 * (y * z - w * x);
  m[6] = 2 * (x * z - w * y); m[7] = 2 * (y * z + w * x); m[8] = w * w - x * x - y * y + z * z;
}
static void normalize4(real* q) {
  real n = r_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  if (n > 0) for (int i = 0; i < 4; i++) q[i] /= n;
}
static void axis_angle_to_quat(real* q, const real* axis, real angle) {
  real s = r_sin(angle * (real)0.5), c = r_cos(angle * (real)0.5);
  q[0] = c; q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}
static void matvec3(real* o, const real* m, const real* v) {
  real r0 = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  real r1 = m[3] * v[0] + m[4] * v[1] + m[5] * v[2];
  real r2 = m[6] * v[0] + m[7] * v[1] + m[8] * v[2];
  o[0] = r0; o[1] = r1; o[2] = r2;
}
/* math.inert_mul: cinert (10) x motion (6: ang,lin) -> force (6: ang,lin) */
static void inert_mul(real* o, const real* I, const real* v) {
  real inr[9] = {I[0], I[3], I[4], I[3], I[1], I[5], I[4], I[5], I[2]};
  const real* pos = I + 6;
  real mass = I[9], a[3], c1[3], c2[3];
  matvec3(a, inr, v);
  cross3(c1, pos, v + 3);
  cross3(c2, pos, v);
  for (int i = 0; i < 3; i++) { o[i] = a[i] + c1[i]; o[3 + i] = mass * v[3 + i] - c2[i]; }
}
/* math.motion_cross(u, v) */
static void motion_cross(real* o, const real* u, const real* v) {
  real a[3], b[3], c[3];
  cross3(a, u, v);
  cross3(b, u + 3, v);
  cross3(c, u, v + 3);
  for (int i = 0; i < 3; i++) { o[i] = a[i]; o[3 + i] = b[i] + c[i]; }
}
/* math.motion_cross_force(v, f) */
static void motion_cross_force(real* o, const real* v, const real* f) {
  real a[3], b[3], c[3];
  cross3(a, v, f);
  cross3(b, v + 3, f + 3);
  cross3(c, v, f + 3);
  for (int i = 0; i < 3; i++) { o[i] = a[i] + b[i]; o[3 + i] = c[i]; }
}

/* ------------------------------------------------------------------ smooth.kinematics */
static void kinematics(const dial_model* m, odata* d) {
  for (int k = 0; k < 3; k++) d->xpos[0][k] = 0;
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  for (int b = 1; b < m->nbody; b++) {
    int p = m->body_parent[b];
    real pos[3], quat[4], bp[3] = {m->body_pos[b][0], m->body_pos[b][1], m->body_pos[b][2]};
    real bq[4] = {m->body_quat[b][0], m->body_quat[b][1], m->body_quat[b][2], m->body_quat[b][3]};
    rotate(pos, bp, d->xquat[p]);
    for (int k = 0; k < 3; k++) pos[k] += d->xpos[p][k];
    quat_mul(quat, d->xquat[p], bq);
    for (int ji = m->body_jntadr[b]; ji < m->body_jntadr[b] + m->body_jntnum[b]; ji++) {
      int qa = m->jnt_qposadr[ji];
      real jp[3] = {m->jnt_pos[ji][0], m->jnt_pos[ji][1], m->jnt_pos[ji][2]};
      real ja[3] = {m->jnt_axis[ji][0], m->jnt_axis[ji][1], m->jnt_axis[ji][2]};
      if (m->jnt_type[ji] == DIAL_JNT_FREE) {
        for (int k = 0; k < 3; k++) { pos[k] = d->qpos[qa + k]; d->xanchor[ji][k] = pos[k]; }
        d->xaxis[ji][0] = 0; d->xaxis[ji][1] = 0; d->xaxis[ji][2] = 1;
        for (int k = 0; k < 4; k++) quat[k] = d->qpos[qa + 3 + k];
        normalize4(quat);
        for (int k = 0; k < 4; k++) d->qpos[qa + 3 + k] = quat[k]; /* kinematics writes back the normalised quat */
      } else {
        real anchor[3], axis[3];
        rotate(anchor, jp, quat);
        for (int k = 0; k < 3; k++) anchor[k] += pos[k];
        rotate(axis, ja, quat);
        for (int k = 0; k < 3; k++) { d->xanchor[ji][k] = anchor[k]; d->xaxis[ji][k] = axis[k]; }
        if (m->jnt_type[ji] == DIAL_JNT_HINGE) {
          real angle = d->qpos[qa] - (real)m->qpos0[qa], qloc[4], t[3];
          axis_angle_to_quat(qloc, ja, angle);
          quat_mul(quat, quat, qloc);
          rotate(t, jp, quat);
          for (int k = 0; k < 3; k++) pos[k] = anchor[k] - t[k];
        } else { /* slide */
          real disp = d->qpos[qa] - (real)m->qpos0[qa];
          for (int k = 0; k < 3; k++) pos[k] += axis[k] * disp;
        }
      }
    }
    for (int k = 0; k < 3; k++) d->xpos[b][k] = pos[k];
    for (int k = 0; k < 4; k++) d->xquat[b][k] = quat[k];
  }
  for (int b = 0; b < m->nbody; b++) {
    quat_to_mat(d->xmat[b], d->xquat[b]);
    real ip[3] = {m->body_ipos[b][0], m->body_ipos[b][1], m->body_ipos[b][2]}, t[3], q[4];
    real iq[4] = {m->body_iquat[b][0], m->body_iquat[b][1], m->body_iquat[b][2], m->body_iquat[b][3]};
    rotate(t, ip, d->xquat[b]);
    for (int k = 0; k < 3; k++) d->xipos[b][k] = d->xpos[b][k] + t[k];
    quat_mul(q, d->xquat[b], iq);
    quat_to_mat(d->ximat[b], q);
  }
  for (int g = 0; g < m->ngeom; g++) {
    int b = m->geom_bodyid[g];
    real gp[3] = {m->geom_pos[g][0], m->geom_pos[g][1], m->geom_pos[g][2]}, t[3], q[4];
    real gq[4] = {m->geom_quat[g][0], m->geom_quat[g][1], m->geom_quat[g][2], m->geom_quat[g][3]};
    rotate(t, gp, d->xquat[b]);
    for (int k = 0; k < 3; k++) d->geom_xpos[g][k] = d->xpos[b][k] + t[k];
    quat_mul(q, d->xquat[b], gq);
    quat_to_mat(d->geom_xmat[g], q);
  }
  for (int s = 0; s < m->nsite; s++) {
    int b = m->site_bodyid[s];
    real sp[3] = {m->site_pos[s][0], m->site_pos[s][1], m->site_pos[s][2]}, t[3];
    rotate(t, sp, d->xquat[b]);
    for (int k = 0; k < 3; k++) d->site_xpos[s][k] = d->xpos[b][k] + t[k];
  }
}

/* ------------------------------------------------------------------ smooth.com_pos */
static void com_pos(const dial_model* m, odata* d) {
  real pos[NB][3], mass[NB];
  for (int b = 0; b < m->nbody; b++) {
    mass[b] = m->body_mass[b];
    for (int k = 0; k < 3; k++) pos[b][k] = d->xipos[b][k] * mass[b];
  }
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parent[b];
    mass[p] += mass[b];
    for (int k = 0; k < 3; k++) pos[p][k] += pos[b][k];
  }
  for (int b = 0; b < m->nbody; b++)
    for (int k = 0; k < 3; k++)
      d->subtree_com[b][k] = mass[b] < MJ_MINVAL ? d->xipos[b][k] : pos[b][k] / mass[b];
  for (int b = 0; b < m->nbody; b++) {
    const real* c = d->subtree_com[m->body_rootid[b]];
    real off[3] = {d->xipos[b][0] - c[0], d->xipos[b][1] - c[1], d->xipos[b][2] - c[2]};
    real mb = m->body_mass[b], I[9];
    const real* R = d->ximat[b];
    /* (ximat * inertia) @ ximat.T + h h^T mass, h h^T = |off|^2 1 - off off^T */
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) {
        real s = 0;
        for (int k = 0; k < 3; k++) s += R[3 * i + k] * (real)m->body_inertia[b][k] * R[3 * j + k];
        real hh = (i == j ? dot3(off, off) : 0) - off[i] * off[j];
        I[3 * i + j] = s + hh * mb;
      }
    real* ci = d->cinert[b];
    ci[0] = I[0]; ci[1] = I[4]; ci[2] = I[8]; ci[3] = I[1]; ci[4] = I[2]; ci[5] = I[5];
    ci[6] = off[0] * mb; ci[7] = off[1] * mb; ci[8] = off[2] * mb; ci[9] = mb;
  }
  for (int ji = 0; ji < m->njnt; ji++) {
    int b = m->jnt_bodyid[ji], da = m->jnt_dofadr[ji];
    const real* c = d->subtree_com[m->body_rootid[b]];
    real off[3] = {c[0] - d->xanchor[ji][0], c[1] - d->xanchor[ji][1], c[2] - d->xanchor[ji][2]};
    if (m->jnt_type[ji] == DIAL_JNT_FREE) {
      for (int i = 0; i < 3; i++)
        for (int k = 0; k < 6; k++) d->cdof[da + i][k] = (k == 3 + i) ? 1 : 0;
      for (int i = 0; i < 3; i++) { /* rows of xmat.T = columns of xmat */
        real a[3] = {d->xmat[b][i], d->xmat[b][3 + i], d->xmat[b][6 + i]}, cr[3];
        cross3(cr, a, off);
        for (int k = 0; k < 3; k++) { d->cdof[da + 3 + i][k] = a[k]; d->cdof[da + 3 + i][3 + k] = cr[k]; }
      }
    } else if (m->jnt_type[ji] == DIAL_JNT_HINGE) {
      real cr[3];
      cross3(cr, d->xaxis[ji], off);
      for (int k = 0; k < 3; k++) { d->cdof[da][k] = d->xaxis[ji][k]; d->cdof[da][3 + k] = cr[k]; }
    } else {
      for (int k = 0; k < 3; k++) { d->cdof[da][k] = 0; d->cdof[da][3 + k] = d->xaxis[ji][k]; }
    }
  }
}

/* ------------------------------------------------------------------ smooth.crb + factor_m (dense) */
static int cholesky(int n, real A[NV][NV], real L[NV][NV]) {
  for (int j = 0; j < n; j++) {
    real s = A[j][j];
    for (int k = 0; k < j; k++) s -= L[j][k] * L[j][k];
    if (!(s > 0)) return -1;
    real ljj = r_sqrt(s);
    L[j][j] = ljj;
    for (int i = j + 1; i < n; i++) {
      real t = A[i][j];
      for (int k = 0; k < j; k++) t -= L[i][k] * L[j][k];
      L[i][j] = t / ljj;
    }
    for (int i = 0; i < j; i++) L[i][j] = 0;
  }
  return 0;
}
static void cho_solve(int n, real L[NV][NV], const real* b, real* x) {
  real y[NV];
  for (int i = 0; i < n; i++) {
    real s = b[i];
    for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
  for (int i = n - 1; i >= 0; i--) {
    real s = y[i];
    for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
}
static void crb(const dial_model* m, odata* d) {
  real cb[NB][10];
  for (int b = 0; b < m->nbody; b++) for (int k = 0; k < 10; k++) cb[b][k] = d->cinert[b][k];
  for (int b = m->nbody - 1; b > 0; b--) {
    int p = m->body_parent[b];
    for (int k = 0; k < 10; k++) cb[p][k] += cb[b][k];
  }
  for (int k = 0; k < 10; k++) cb[0][k] = 0;
  int nv = m->nv;
  for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) d->qM[i][j] = 0;
  for (int i = 0; i < nv; i++) {
    real f[6];
    inert_mul(f, cb[m->dof_bodyid[i]], d->cdof[i]);
    for (int j = i; j >= 0; j = m->dof_parentid[j]) {
      real s = 0;
      for (int k = 0; k < 6; k++) s += f[k] * d->cdof[j][k];
      d->qM[i][j] = s; d->qM[j][i] = s;
    }
    d->qM[i][i] += (real)m->dof_armature[i];
  }
  cholesky(nv, d->qM, d->qL);
}

/* ------------------------------------------------------------------ collision_driver (static list) */
static void make_frame(real* frame, const real* a_in) {
  real a[3] = {a_in[0], a_in[1], a_in[2]}, n = r_sqrt(dot3(a, a));
  for (int k = 0; k < 3; k++) a[k] /= n;
  real b[3] = {0, 0, 0};
  if (-0.5 < a[1] && a[1] < 0.5) b[1] = 1; else b[2] = 1;
  real ab = dot3(a, b);
  for (int k = 0; k < 3; k++) b[k] -= a[k] * ab;
  n = r_sqrt(dot3(b, b));
  for (int k = 0; k < 3; k++) b[k] /= n;
  real c[3];
  cross3(c, a, b);
  for (int k = 0; k < 3; k++) { frame[k] = a[k]; frame[3 + k] = b[k]; frame[6 + k] = c[k]; }
}
/* math.closest_segment_point (MJX): point on segment [a, b] closest to pt */
static void closest_segment_point(real* o, const real* a, const real* b, const real* pt) {
  real ab[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, pa[3] = {pt[0] - a[0], pt[1] - a[1], pt[2] - a[2]};
  real t = dot3(pa, ab) / (dot3(ab, ab) + (real)1e-6);
  t = r_clip(t, 0, 1);
  for (int k = 0; k < 3; k++) o[k] = a[k] + t * ab[k];
}
/* math.closest_segment_to_segment_points (MJX) */
static void closest_segment_to_segment(real* best_a, real* best_b, const real* a0, const real* a1, const real* b0, const real* b1) {
  real da[3] = {a1[0] - a0[0], a1[1] - a0[1], a1[2] - a0[2]}, db[3] = {b1[0] - b0[0], b1[1] - b0[1], b1[2] - b0[2]};
  real len_a = r_sqrt(dot3(da, da)), len_b = r_sqrt(dot3(db, db));
  for (int k = 0; k < 3; k++) { da[k] = len_a > 0 ? da[k] / len_a : 0; db[k] = len_b > 0 ? db[k] / len_b : 0; }
  real half_a = len_a * (real)0.5, half_b = len_b * (real)0.5, a_mid[3], b_mid[3], trans[3];
  for (int k = 0; k < 3; k++) { a_mid[k] = a0[k] + da[k] * half_a; b_mid[k] = b0[k] + db[k] * half_b; trans[k] = a_mid[k] - b_mid[k]; }
  real dadb = dot3(da, db), dat = dot3(da, trans), dbt = dot3(db, trans);
  real denom = 1 - dadb * dadb;
  real orig_ta = (-dat + dadb * dbt) / (denom + (real)1e-6);
  real orig_tb = dbt + orig_ta * dadb;
  real ta = r_clip(orig_ta, -half_a, half_a), tb = r_clip(orig_tb, -half_b, half_b);
  real ba[3], bb[3], na[3], nb[3];
  for (int k = 0; k < 3; k++) { ba[k] = a_mid[k] + da[k] * ta; bb[k] = b_mid[k] + db[k] * tb; }
  closest_segment_point(na, a0, a1, bb);
  closest_segment_point(nb, b0, b1, ba);
  real e1[3] = {na[0] - bb[0], na[1] - bb[1], na[2] - bb[2]}, e2[3] = {nb[0] - ba[0], nb[1] - ba[1], nb[2] - ba[2]};
  real d1 = dot3(e1, e1), d2 = dot3(e2, e2);
  for (int k = 0; k < 3; k++) { best_a[k] = d1 < d2 ? na[k] : ba[k]; best_b[k] = d1 < d2 ? bb[k] : nb[k]; }
}
/* collision_primitive._sphere_sphere (MJX): the contact of two spheres, shared by sphere-capsule / capsule-capsule */
static void sphere_sphere(const real* p1, real r1, const real* p2, real r2, real* dist, real* pos, real* frame) {
  real n[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, len = r_sqrt(dot3(n, n));
  if (len == 0) { n[0] = 1; n[1] = 0; n[2] = 0; }
  else for (int k = 0; k < 3; k++) n[k] /= len;
  *dist = len - (r1 + r2);
  for (int k = 0; k < 3; k++) pos[k] = p1[k] + n[k] * (r1 + *dist * (real)0.5);
  make_frame(frame, n);
}
/* ---- box narrow phases (the crate scenes).  MJX sends boxes through collision_convex.py, which is third-party code that
 * is not under /root/reference; these functions restate the GEOMETRY of each pair (unique wherever the contact is), with a
 * fixed number of candidate contacts per pair -- include/dial_mpc.h DIAL_CON_*_BOX, DESIGN.md section 1.  Conventions are
 * MJX's: the normal points from geom1 into geom2, dist < 0 is penetration, pos lies midway between the two surfaces. */
typedef struct { real c[3], R[9], h[3]; } obox;   /* centre, world-from-local rotation (row-major), half sizes */
static void obox_of(const dial_model* m, const odata* d, int g, obox* b) {
  for (int k = 0; k < 3; k++) { b->c[k] = d->geom_xpos[g][k]; b->h[k] = (real)m->geom_size[g][k]; }
  for (int k = 0; k < 9; k++) b->R[k] = d->geom_xmat[g][k];
}
static void obox_axis(const obox* b, int k, real* a) { a[0] = b->R[k]; a[1] = b->R[3 + k]; a[2] = b->R[6 + k]; }
static void obox_local(const obox* b, const real* p, real* o) {
  real r[3] = {p[0] - b->c[0], p[1] - b->c[1], p[2] - b->c[2]};
  for (int k = 0; k < 3; k++) o[k] = b->R[k] * r[0] + b->R[3 + k] * r[1] + b->R[6 + k] * r[2];
}
static void obox_world_dir(const obox* b, const real* v, real* o) {
  for (int k = 0; k < 3; k++) o[k] = b->R[3 * k] * v[0] + b->R[3 * k + 1] * v[1] + b->R[3 * k + 2] * v[2];
}
/* a sphere against a box: the point of the box closest to the centre; a centre inside the box leaves through the nearest face */
static void sphere_box(const real* sc, real r, const obox* b, real* dist, real* pos, real* frame) {
  real p[3], nl[3], len2 = 0;
  obox_local(b, sc, p);
  for (int k = 0; k < 3; k++) { nl[k] = r_clip(p[k], -b->h[k], b->h[k]) - p[k]; len2 += nl[k] * nl[k]; }
  if (len2 > 0) {
    real len = r_sqrt(len2);
    for (int k = 0; k < 3; k++) nl[k] /= len;
    *dist = len - r;
  } else {
    int kb = 0;
    real best = b->h[0] - r_abs(p[0]);
    for (int k = 1; k < 3; k++) { real e = b->h[k] - r_abs(p[k]); if (e < best) { best = e; kb = k; } }
    nl[0] = nl[1] = nl[2] = 0;
    nl[kb] = p[kb] >= 0 ? -1 : 1;
    *dist = -best - r;
  }
  real n[3];
  obox_world_dir(b, nl, n);
  for (int k = 0; k < 3; k++) pos[k] = sc[k] + n[k] * (r + *dist * (real)0.5);
  make_frame(frame, n);
}
/* plane against a box: the sub-th lowest vertex (ties: lower vertex index first) */
static void plane_box(const real* n, const real* ppos, const obox* b, int sub, real* dist, real* pos, real* frame) {
  real v[8][3], dv[8];
  for (int i = 0; i < 8; i++) {
    real l[3] = {(i & 1) ? b->h[0] : -b->h[0], (i & 2) ? b->h[1] : -b->h[1], (i & 4) ? b->h[2] : -b->h[2]}, w[3];
    obox_world_dir(b, l, w);
    for (int k = 0; k < 3; k++) v[i][k] = b->c[k] + w[k];
    real df[3] = {v[i][0] - ppos[0], v[i][1] - ppos[1], v[i][2] - ppos[2]};
    dv[i] = dot3(df, n);
  }
  int pick = 0;
  for (int i = 0; i < 8; i++) {
    int rank = 0;
    for (int j = 0; j < 8; j++) rank += (dv[j] < dv[i] || (dv[j] == dv[i] && j < i)) ? 1 : 0;
    if (rank == sub) pick = i;
  }
  *dist = dv[pick];
  for (int k = 0; k < 3; k++) pos[k] = v[pick][k] - n[k] * (dv[pick] * (real)0.5);
  make_frame(frame, n);
}
/* squared distance from the segment point a0 + t (a1 - a0) (box frame) to the box, minimised over t in [0, 1]: the
 * function is convex and piecewise quadratic, its pieces end where the point crosses one of the six slab planes */
static real segment_box_closest_t(const real* a0, const real* a1, const real* h) {
  real bp[8];
  int nb = 0;
  bp[nb++] = 0;
  for (int k = 0; k < 3; k++) {
    real dk = a1[k] - a0[k];
    if (dk == 0) continue;
    for (int sg = -1; sg <= 1; sg += 2) {
      real t = ((real)sg * h[k] - a0[k]) / dk;
      if (t > 0 && t < 1) bp[nb++] = t;
    }
  }
  bp[nb++] = 1;
  for (int i = 1; i < nb; i++) {   /* insertion sort */
    real x = bp[i];
    int j = i - 1;
    while (j >= 0 && bp[j] > x) { bp[j + 1] = bp[j]; j--; }
    bp[j + 1] = x;
  }
  real best_t = 0, best_f = -1;
  for (int i = 0; i + 1 < nb; i++) {
    real t0 = bp[i], t1 = bp[i + 1], tm = (real)0.5 * (t0 + t1), A = 0, B = 0;
    real off[3];
    int st[3];
    for (int k = 0; k < 3; k++) {
      real x = a0[k] + tm * (a1[k] - a0[k]);
      st[k] = x > h[k] ? 1 : (x < -h[k] ? -1 : 0);
      off[k] = a0[k] - (real)st[k] * h[k];
      if (st[k]) { real dk = a1[k] - a0[k]; A += dk * dk; B += dk * off[k]; }
    }
    real t = A > 0 ? r_clip(-B / A, t0, t1) : t0, f = 0;
    for (int k = 0; k < 3; k++) if (st[k]) { real x = off[k] + t * (a1[k] - a0[k]); f += x * x; }
    /* a later piece replaces an earlier one only if it is better by more than rounding (a capsule parallel to a face keeps
     * its first end, a segment that passes through the box the point where it enters) */
    if (best_f < 0 || f < best_f * (1 - (real)1e-6) - (real)1e-12) { best_f = f; best_t = t; }
  }
  return best_t;
}
/* capsule against a box: sub 0 = the sphere at the segment point closest to the box, sub 1 = the sphere at the segment end
 * farther from that point (a capsule lying on a face touches with both, one standing on an end or crossing an edge with one).
 * When the capsule's AXIS itself enters the box (penetration deeper than the radius) the closest point degenerates to a whole
 * stretch of distance 0; the contact is then the point where the axis crosses the surface, with the normal of the face it
 * crosses and dist = -radius -- the continuation of the shallow case (closest point -> surface point, same face normal). */
static void capsule_box(const real* ctr, const real* axis, real hl, real r, const obox* b, int sub, real* dist, real* pos, real* frame) {
  real e0[3], e1[3], l0[3], l1[3];
  for (int k = 0; k < 3; k++) { e0[k] = ctr[k] - axis[k] * hl; e1[k] = ctr[k] + axis[k] * hl; }
  obox_local(b, e0, l0);
  obox_local(b, e1, l1);
  /* interior stretch [ta, tb] of the axis: intersection of the three slabs; ka / kb = the slab that bounds it */
  real ta = 0, tb = 1, t;
  int hit = 1, ka = -1, kb = -1, kface = -1;
  for (int k = 0; k < 3 && hit; k++) {
    real dk = l1[k] - l0[k];
    if (dk == 0) { if (r_abs(l0[k]) > b->h[k]) hit = 0; continue; }
    real t1 = (-b->h[k] - l0[k]) / dk, t2 = (b->h[k] - l0[k]) / dk;
    real lo = r_min(t1, t2), hi = r_max(t1, t2);
    if (lo > ta) { ta = lo; ka = k; }
    if (hi < tb) { tb = hi; kb = k; }
  }
  if (hit && ta <= tb) {
    if (ka >= 0) { t = ta; kface = ka; }          /* the axis enters through face ka */
    else if (kb >= 0) { t = tb; kface = kb; }     /* first end inside: where it leaves */
    else t = 0;                                   /* the whole axis is inside: plain end spheres */
  } else {
    t = segment_box_closest_t(l0, l1, b->h);
  }
  if (sub == 0 && kface >= 0) {
    real pl[3], nl[3] = {0, 0, 0}, n[3], pw[3];
    for (int k = 0; k < 3; k++) pl[k] = l0[k] + t * (l1[k] - l0[k]);
    nl[kface] = pl[kface] >= 0 ? -1 : 1;
    obox_world_dir(b, nl, n);
    for (int k = 0; k < 3; k++) pw[k] = e0[k] + t * (e1[k] - e0[k]);
    *dist = -r;
    for (int k = 0; k < 3; k++) pos[k] = pw[k] + n[k] * (r + *dist * (real)0.5);
    make_frame(frame, n);
    return;
  }
  real sc[3];
  if (sub == 0) for (int k = 0; k < 3; k++) sc[k] = e0[k] + t * (e1[k] - e0[k]);
  else for (int k = 0; k < 3; k++) sc[k] = t <= (real)0.5 ? e1[k] : e0[k];
  sphere_box(sc, r, b, dist, pos, frame);
}
/* box against box: separating-axis test over the 6 face normals and the 9 edge-edge directions.  Face axis: the incident
 * face of the other box is clipped against the side planes of the reference face, the sub-th deepest point is the
 * contact; edge axis: the closest points of the two edges (sub 0).  Candidates that do not exist are parked at dist = 1. */
static void box_box(const obox* A, const obox* B, int sub, real* dist, real* pos, real* frame) {
  real tw[3] = {B->c[0] - A->c[0], B->c[1] - A->c[1], B->c[2] - A->c[2]};
  {   /* bounding spheres more than 1 cm apart: every candidate is parked, the normal is the centre line */
    real gap = r_sqrt(dot3(tw, tw)) - r_sqrt(dot3(A->h, A->h)) - r_sqrt(dot3(B->h, B->h));
    if (gap > (real)0.01) {
      *dist = sub == 0 ? gap : 1;
      for (int k = 0; k < 3; k++) pos[k] = (real)0.5 * (A->c[k] + B->c[k]);
      make_frame(frame, tw);
      return;
    }
  }
  real ax[2][3][3];
  for (int k = 0; k < 3; k++) { obox_axis(A, k, ax[0][k]); obox_axis(B, k, ax[1][k]); }
  real Rr[3][3], Q[3][3], t[3];
  for (int i = 0; i < 3; i++) {
    t[i] = dot3(tw, ax[0][i]);
    for (int j = 0; j < 3; j++) { Rr[i][j] = dot3(ax[0][i], ax[1][j]); Q[i][j] = r_abs(Rr[i][j]); }
  }
  /* face axes */
  int best = -1;
  real sbest = 0;
  for (int i = 0; i < 3; i++) {
    real s = r_abs(t[i]) - (A->h[i] + B->h[0] * Q[i][0] + B->h[1] * Q[i][1] + B->h[2] * Q[i][2]);
    if (best < 0 || s > sbest) { best = i; sbest = s; }
  }
  for (int j = 0; j < 3; j++) {
    real tb = t[0] * Rr[0][j] + t[1] * Rr[1][j] + t[2] * Rr[2][j];
    real s = r_abs(tb) - (B->h[j] + A->h[0] * Q[0][j] + A->h[1] * Q[1][j] + A->h[2] * Q[2][j]);
    if (s > sbest) { best = 3 + j; sbest = s; }
  }
  /* edge axes: an edge pair wins only if it separates clearly better than the best face (no jitter between the two kinds) */
  real nbest[3] = {0, 0, 0};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      real L[3];
      cross3(L, ax[0][i], ax[1][j]);
      real len = r_sqrt(dot3(L, L));
      if (len < (real)1e-4) continue;
      for (int k = 0; k < 3; k++) L[k] /= len;
      real ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += A->h[k] * r_abs(dot3(L, ax[0][k])); rb += B->h[k] * r_abs(dot3(L, ax[1][k])); }
      real s = r_abs(dot3(tw, L)) - (ra + rb);
      if (s > sbest + (real)0.05 * r_abs(sbest) + (real)1e-5) { best = 6 + 3 * i + j; sbest = s; for (int k = 0; k < 3; k++) nbest[k] = L[k]; }
    }
  real mid[3] = {(real)0.5 * (A->c[0] + B->c[0]), (real)0.5 * (A->c[1] + B->c[1]), (real)0.5 * (A->c[2] + B->c[2])};
  if (sbest > (real)0.01) {   /* clearly apart */
    real n[3];
    if (best < 3) for (int k = 0; k < 3; k++) n[k] = ax[0][best][k] * (t[best] >= 0 ? 1 : -1);
    else if (best < 6) { real sg = dot3(tw, ax[1][best - 3]) >= 0 ? 1 : -1; for (int k = 0; k < 3; k++) n[k] = ax[1][best - 3][k] * sg; }
    else { real sg = dot3(tw, nbest) >= 0 ? 1 : -1; for (int k = 0; k < 3; k++) n[k] = nbest[k] * sg; }
    *dist = sub == 0 ? sbest : 1;
    for (int k = 0; k < 3; k++) pos[k] = mid[k];
    make_frame(frame, n);
    return;
  }
  if (best >= 6) {   /* edge - edge */
    int i = (best - 6) / 3, j = (best - 6) % 3;
    real n[3], sg = dot3(tw, nbest) >= 0 ? 1 : -1;
    for (int k = 0; k < 3; k++) n[k] = nbest[k] * sg;
    real pa[3] = {A->c[0], A->c[1], A->c[2]}, pb[3] = {B->c[0], B->c[1], B->c[2]};
    for (int k = 0; k < 3; k++) {
      if (k != i) { real sk = dot3(n, ax[0][k]) >= 0 ? 1 : -1; for (int q = 0; q < 3; q++) pa[q] += sk * A->h[k] * ax[0][k][q]; }
      if (k != j) { real sk = dot3(n, ax[1][k]) >= 0 ? 1 : -1; for (int q = 0; q < 3; q++) pb[q] -= sk * B->h[k] * ax[1][k][q]; }
    }
    const real *ua = ax[0][i], *ub = ax[1][j];
    real w[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    real uaub = dot3(ua, ub), q1 = dot3(ua, w), q2 = -dot3(ub, w), den = 1 - uaub * uaub;
    real al = (q1 + uaub * q2) / den, be = (uaub * q1 + q2) / den;
    al = r_clip(al, -A->h[i], A->h[i]);
    be = r_clip(be, -B->h[j], B->h[j]);
    real ca[3], cb[3], df[3];
    for (int k = 0; k < 3; k++) { ca[k] = pa[k] + al * ua[k]; cb[k] = pb[k] + be * ub[k]; df[k] = cb[k] - ca[k]; }
    *dist = sub == 0 ? dot3(df, n) : 1;
    for (int k = 0; k < 3; k++) pos[k] = (real)0.5 * (ca[k] + cb[k]);
    make_frame(frame, n);
    return;
  }
  /* face: reference box X (axis kx), incident box Y */
  const obox *X = best < 3 ? A : B, *Y = best < 3 ? B : A;
  const int xi = best < 3 ? 0 : 1, yi = 1 - xi, kx = best < 3 ? best : best - 3;
  real xy[3] = {Y->c[0] - X->c[0], Y->c[1] - X->c[1], Y->c[2] - X->c[2]};
  real nref[3], sgx = dot3(xy, ax[xi][kx]) >= 0 ? 1 : -1;
  for (int k = 0; k < 3; k++) nref[k] = ax[xi][kx][k] * sgx;
  int my = 0;
  real amax = -1;
  for (int k = 0; k < 3; k++) { real a = r_abs(dot3(nref, ax[yi][k])); if (a > amax) { amax = a; my = k; } }
  const int uy = (my + 1) % 3, vy = (my + 2) % 3, ux = (kx + 1) % 3, vx = (kx + 2) % 3;
  real sgy = dot3(nref, ax[yi][my]) >= 0 ? -1 : 1;   /* the incident face looks back at the reference box */
  real poly[2][10][3];   /* (a, b, d): reference-face coordinates and signed distance to the reference face */
  int np_ = 4;
  for (int q = 0; q < 4; q++) {
    real su = (q == 0 || q == 3) ? 1 : -1, sv = (q < 2) ? 1 : -1, pw[3];
    for (int k = 0; k < 3; k++) pw[k] = Y->c[k] + sgy * Y->h[my] * ax[yi][my][k] + su * Y->h[uy] * ax[yi][uy][k] + sv * Y->h[vy] * ax[yi][vy][k] - X->c[k];
    poly[0][q][0] = dot3(pw, ax[xi][ux]);
    poly[0][q][1] = dot3(pw, ax[xi][vx]);
    poly[0][q][2] = dot3(pw, nref) - X->h[kx];
  }
  int cur = 0;
  for (int pl = 0; pl < 4; pl++) {   /* Sutherland-Hodgman against  +-a <= h_u,  +-b <= h_v */
    const int co = pl >> 1;
    const real sg = (pl & 1) ? -1 : 1, lim = co == 0 ? X->h[ux] : X->h[vx];
    int no = 0;
    for (int q = 0; q < np_; q++) {
      const real* P = poly[cur][q];
      const real* Qn = poly[cur][(q + 1) % np_];
      real fp = sg * P[co] - lim, fq = sg * Qn[co] - lim;
      if (fp <= 0) { for (int k = 0; k < 3; k++) poly[1 - cur][no][k] = P[k]; no++; }
      if ((fp <= 0) != (fq <= 0)) {
        real w = fp / (fp - fq);
        for (int k = 0; k < 3; k++) poly[1 - cur][no][k] = P[k] + w * (Qn[k] - P[k]);
        no++;
      }
    }
    np_ = no;
    cur = 1 - cur;
    if (np_ == 0) break;
  }
  real n[3];
  for (int k = 0; k < 3; k++) n[k] = xi == 0 ? nref[k] : -nref[k];
  make_frame(frame, n);
  int pick = -1;
  for (int q = 0; q < np_; q++) {
    int rank = 0;
    for (int o = 0; o < np_; o++) rank += (poly[cur][o][2] < poly[cur][q][2] || (poly[cur][o][2] == poly[cur][q][2] && o < q)) ? 1 : 0;
    if (rank == sub) pick = q;
  }
  if (pick < 0) { *dist = 1; for (int k = 0; k < 3; k++) pos[k] = mid[k]; return; }
  const real* P = poly[cur][pick];
  *dist = P[2];
  for (int k = 0; k < 3; k++) pos[k] = X->c[k] + P[0] * ax[xi][ux][k] + P[1] * ax[xi][vx][k] + (X->h[kx] + P[2] * (real)0.5) * nref[k];
}
static void collision(const dial_model* m, odata* d) {
  for (int c = 0; c < m->ncon; c++) {
    int g1 = m->con_geom1[c], g2 = m->con_geom2[c];
    if (m->con_kind[c] >= DIAL_CON_PLANE_BOX) {
      obox b2;
      obox_of(m, d, g2, &b2);
      if (m->con_kind[c] == DIAL_CON_PLANE_BOX) {
        real n[3] = {d->geom_xmat[g1][2], d->geom_xmat[g1][5], d->geom_xmat[g1][8]};
        plane_box(n, d->geom_xpos[g1], &b2, m->con_sub[c], &d->con_dist[c], d->con_pos[c], d->con_frame[c]);
      } else if (m->con_kind[c] == DIAL_CON_SPHERE_BOX) {
        sphere_box(d->geom_xpos[g1], (real)m->geom_size[g1][0], &b2, &d->con_dist[c], d->con_pos[c], d->con_frame[c]);
      } else if (m->con_kind[c] == DIAL_CON_CAPSULE_BOX) {
        real ax1[3] = {d->geom_xmat[g1][2], d->geom_xmat[g1][5], d->geom_xmat[g1][8]};
        capsule_box(d->geom_xpos[g1], ax1, (real)m->geom_size[g1][1], (real)m->geom_size[g1][0], &b2, m->con_sub[c], &d->con_dist[c], d->con_pos[c], d->con_frame[c]);
      } else {
        obox b1;
        obox_of(m, d, g1, &b1);
        box_box(&b1, &b2, m->con_sub[c], &d->con_dist[c], d->con_pos[c], d->con_frame[c]);
      }
      continue;
    }
    if (m->con_kind[c] == DIAL_CON_SPHERE_CAPSULE || m->con_kind[c] == DIAL_CON_CAPSULE_CAPSULE) {
      real ax2[3] = {d->geom_xmat[g2][2], d->geom_xmat[g2][5], d->geom_xmat[g2][8]}, hl2 = m->geom_size[g2][1];
      real b0[3], b1[3], p1[3], p2[3];
      for (int k = 0; k < 3; k++) { b0[k] = d->geom_xpos[g2][k] - ax2[k] * hl2; b1[k] = d->geom_xpos[g2][k] + ax2[k] * hl2; }
      if (m->con_kind[c] == DIAL_CON_SPHERE_CAPSULE) {
        for (int k = 0; k < 3; k++) p1[k] = d->geom_xpos[g1][k];
        closest_segment_point(p2, b0, b1, p1);
      } else {
        real ax1[3] = {d->geom_xmat[g1][2], d->geom_xmat[g1][5], d->geom_xmat[g1][8]}, hl1 = m->geom_size[g1][1], a0[3], a1[3];
        for (int k = 0; k < 3; k++) { a0[k] = d->geom_xpos[g1][k] - ax1[k] * hl1; a1[k] = d->geom_xpos[g1][k] + ax1[k] * hl1; }
        closest_segment_to_segment(p1, p2, a0, a1, b0, b1);
      }
      sphere_sphere(p1, (real)m->geom_size[g1][0], p2, (real)m->geom_size[g2][0], &d->con_dist[c], d->con_pos[c], d->con_frame[c]);
      continue;
    }
    real n[3] = {d->geom_xmat[g1][2], d->geom_xmat[g1][5], d->geom_xmat[g1][8]};
    real ctr[3] = {d->geom_xpos[g2][0], d->geom_xpos[g2][1], d->geom_xpos[g2][2]};
    real radius = m->geom_size[g2][0];
    if (m->con_kind[c] == DIAL_CON_PLANE_SPHERE) {
      make_frame(d->con_frame[c], n);
    } else {
      real axis[3] = {d->geom_xmat[g2][2], d->geom_xmat[g2][5], d->geom_xmat[g2][8]};
      real na = dot3(n, axis), b[3], bn;
      for (int k = 0; k < 3; k++) b[k] = axis[k] - n[k] * na;
      bn = r_sqrt(dot3(b, b));
      if (bn < 0.5) {
        b[0] = 0; b[1] = 0; b[2] = 0;
        if (-0.5 < n[1] && n[1] < 0.5) b[1] = 1; else b[2] = 1;
      } else {
        for (int k = 0; k < 3; k++) b[k] /= bn;
      }
      real cr[3];
      cross3(cr, n, b);
      for (int k = 0; k < 3; k++) { d->con_frame[c][k] = n[k]; d->con_frame[c][3 + k] = b[k]; d->con_frame[c][6 + k] = cr[k]; }
      real sgn = m->con_kind[c] == DIAL_CON_PLANE_CAPSULE_P ? 1 : -1, hl = m->geom_size[g2][1];
      for (int k = 0; k < 3; k++) ctr[k] += sgn * axis[k] * hl;
    }
    real diff[3] = {ctr[0] - d->geom_xpos[g1][0], ctr[1] - d->geom_xpos[g1][1], ctr[2] - d->geom_xpos[g1][2]};
    real dist = dot3(diff, n) - radius;
    d->con_dist[c] = dist;
    for (int k = 0; k < 3; k++) d->con_pos[c][k] = ctr[k] - n[k] * (radius + (real)0.5 * dist);
  }
}

/* ------------------------------------------------------------------ constraint.make_constraint */
static void kbi(const dial_model* m, const real* solref, const real* solimp, real pos, real* k, real* b, real* imp) {
  real timeconst = solref[0], dampratio = solref[1];
  timeconst = r_max(timeconst, 2 * (real)m->timestep); /* refsafe */
  real dmin = r_clip(solimp[0], MJ_MINIMP, MJ_MAXIMP), dmax = r_clip(solimp[1], MJ_MINIMP, MJ_MAXIMP);
  real width = r_max(MJ_MINVAL, solimp[2]), mid = r_clip(solimp[3], MJ_MINIMP, MJ_MAXIMP);
  real power = r_max(1, solimp[4]);
  *k = 1 / (dmax * dmax * timeconst * timeconst * dampratio * dampratio);
  *b = 2 / (dmax * timeconst);
  if (solref[0] <= 0) *k = -solref[0] / (dmax * dmax);
  if (solref[1] <= 0) *b = -solref[1] / dmax;
  real x = r_abs(pos) / width;
  real ia = (1 / r_pow(mid, power - 1)) * r_pow(x, power);
  real ib = 1 - (1 / r_pow(1 - mid, power - 1)) * r_pow(1 - x, power);
  real y = x < mid ? ia : ib;
  real im = dmin + y * (dmax - dmin);
  im = r_clip(im, dmin, dmax);
  if (x > 1) im = dmax;
  *imp = im;
}
/* support.jac: translational Jacobian column i of `body` at world `point` */
static void jacp_col(const dial_model* m, const odata* d, int body, const real* point, int i, real* out) {
  out[0] = out[1] = out[2] = 0;
  /* is dof i an ancestor-or-self dof of body? */
  int b = body, ok = 0;
  int db = m->dof_bodyid[i];
  while (b > 0) { if (b == db) { ok = 1; break; } b = m->body_parent[b]; }
  if (!ok) return;
  const real* c = d->subtree_com[m->body_rootid[body]];
  real off[3] = {point[0] - c[0], point[1] - c[1], point[2] - c[2]}, cr[3];
  cross3(cr, d->cdof[i], off);
  for (int k = 0; k < 3; k++) out[k] = d->cdof[i][3 + k] + cr[k];
}
/* support.jac: rotational Jacobian column i of `body` */
static void jacr_col(const dial_model* m, const odata* d, int body, int i, real* out) {
  out[0] = out[1] = out[2] = 0;
  int b = body, db = m->dof_bodyid[i];
  while (b > 0) { if (b == db) { for (int k = 0; k < 3; k++) out[k] = d->cdof[i][k]; return; } b = m->body_parent[b]; }
}
/* first constraint row of contact c: limits, then 4 pyramid edges (pyramidal) or condim rows (elliptic) per contact */
static int efc_adr(const dial_model* m, int c) {
  int r = m->nlim + m->nfri;
  for (int k = 0; k < c; k++) r += m->cone == DIAL_CONE_ELLIPTIC ? m->con_dim[k] : 4;
  return r;
}
static void make_constraint(const dial_model* m, odata* d) {
  int nv = m->nv, r = 0;
  /* limits (constraint._instantiate_limit_slide_hinge) */
  for (int l = 0; l < m->nlim; l++, r++) {
    int ji = m->lim_jnt[l], qa = m->jnt_qposadr[ji], da = m->jnt_dofadr[ji];
    real q = d->qpos[qa];
    real dist_min = q - (real)m->jnt_range[ji][0], dist_max = (real)m->jnt_range[ji][1] - q;
    real pos = r_min(dist_min, dist_max) - (real)m->jnt_margin[ji];
    int on = pos < 0;
    for (int i = 0; i < nv; i++) d->efc_J[r][i] = 0;
    d->efc_on[r] = on;
    d->efc_D[r] = 0; d->efc_aref[r] = 0;
    if (!on) continue;
    real sgn = dist_min < dist_max ? 1 : -1;
    d->efc_J[r][da] = sgn;
    real solref[2] = {m->jnt_solref[ji][0], m->jnt_solref[ji][1]}, solimp[5];
    for (int k = 0; k < 5; k++) solimp[k] = m->jnt_solimp[ji][k];
    real k_, b_, imp;
    kbi(m, solref, solimp, pos, &k_, &b_, &imp);
    real R = r_max((real)m->dof_invweight0[da] * (1 - imp) / imp, MJ_MINVAL);
    real vel = sgn * d->qvel[da];
    d->efc_aref[r] = -b_ * vel - k_ * imp * pos;
    d->efc_D[r] = 1 / R;
  }
  for (int q = 0; q < m->nlim; q++) d->efc_floss[q] = 0;
  /* dry friction (constraint._instantiate_friction): one row per dof with frictionloss, J = e_dof, pos = 0 (the impedance is
   * solimp's d0), aref = -b * qvel; the row's cost is quadratic up to |force| = frictionloss and linear beyond (solver) */
  for (int q = 0; q < m->nfri; q++, r++) {
    int da = m->fri_dof[q];
    for (int i = 0; i < nv; i++) d->efc_J[r][i] = 0;
    d->efc_J[r][da] = 1;
    d->efc_on[r] = 1;
    real solref[2] = {m->fri_solref[q][0], m->fri_solref[q][1]}, solimp[5];
    for (int k = 0; k < 5; k++) solimp[k] = m->fri_solimp[q][k];
    real k_, b_, imp;
    kbi(m, solref, solimp, 0, &k_, &b_, &imp);
    real R = r_max((real)m->dof_invweight0[da] * (1 - imp) / imp, MJ_MINVAL);
    d->efc_aref[r] = -b_ * d->qvel[da];
    d->efc_D[r] = 1 / R;
    d->efc_floss[r] = (real)m->fri_loss[q];
  }
  for (int q = r; q < m->nefc; q++) d->efc_floss[q] = 0;
  /* elliptic contacts (constraint._efc_contact_elliptic): condim rows per contact = normal, 2 tangents, torsion,
   * 2 rolling; R of the friction rows follows from the normal row (impratio, friction ratios); the friction rows'
   * reference acceleration has no position term (pos_aref = 0) */
  if (m->cone == DIAL_CONE_ELLIPTIC) {
    for (int c = 0; c < m->ncon; c++) {
      real pos = d->con_dist[c] - (real)m->con_margin[c];
      int on = pos < 0, dim = m->con_dim[c];
      int b1 = m->con_body1[c], b2 = m->con_body2[c];
      real t = (real)m->body_invweight0[b1][0] + (real)m->body_invweight0[b2][0];
      real f0 = m->con_friction[c][0];
      real invw[6];
      invw[0] = t;
      invw[1] = t / (real)m->impratio;
      for (int j = 2; j < dim; j++) { real fj = m->con_friction[c][j - 1]; invw[j] = invw[1] * (f0 * f0) / (fj * fj); }
      real solref[2] = {m->con_solref[c][0], m->con_solref[c][1]}, solimp[5];
      for (int k = 0; k < 5; k++) solimp[k] = m->con_solimp[c][k];
      real k_, b_, imp;
      kbi(m, solref, solimp, pos, &k_, &b_, &imp);
      for (int j = 0; j < dim; j++, r++) {
        real vel = 0;
        for (int i = 0; i < nv; i++) {
          real c1[3], c2[3], diff[3];
          if (j < 3) { jacp_col(m, d, b1, d->con_pos[c], i, c1); jacp_col(m, d, b2, d->con_pos[c], i, c2); }
          else { jacr_col(m, d, b1, i, c1); jacr_col(m, d, b2, i, c2); }
          for (int k = 0; k < 3; k++) diff[k] = c2[k] - c1[k];
          real jv_ = on ? dot3(d->con_frame[c] + 3 * (j % 3), diff) : 0;
          d->efc_J[r][i] = jv_;
          vel += jv_ * d->qvel[i];
        }
        real R = r_max(invw[j] * (1 - imp) / imp, MJ_MINVAL);
        d->efc_on[r] = on;
        d->efc_aref[r] = on ? -b_ * vel - k_ * imp * (j == 0 ? pos : 0) : 0;
        d->efc_D[r] = on ? 1 / R : 0;
      }
    }
    return;
  }
  /* pyramidal contacts (constraint._instantiate_contact) */
  for (int c = 0; c < m->ncon; c++) {
    real pos = d->con_dist[c] - (real)m->con_margin[c];
    int on = pos < 0;
    int b1 = m->con_body1[c], b2 = m->con_body2[c];
    real dcon[3][NV];
    for (int i = 0; i < nv; i++) {
      real j1[3], j2[3], diff[3];
      jacp_col(m, d, b1, d->con_pos[c], i, j1);
      jacp_col(m, d, b2, d->con_pos[c], i, j2);
      for (int k = 0; k < 3; k++) diff[k] = j2[k] - j1[k];
      for (int a = 0; a < 3; a++) dcon[a][i] = dot3(d->con_frame[c] + 3 * a, diff);
    }
    real t = (real)m->body_invweight0[b1][0] + (real)m->body_invweight0[b2][0];
    real mu = m->con_friction[c][0];
    real invweight = t + mu * mu * t;
    invweight = invweight * 2 * mu * mu / (real)m->impratio;
    real solref[2] = {m->con_solref[c][0], m->con_solref[c][1]}, solimp[5];
    for (int k = 0; k < 5; k++) solimp[k] = m->con_solimp[c][k];
    real k_, b_, imp;
    kbi(m, solref, solimp, pos, &k_, &b_, &imp);
    real R = r_max(invweight * (1 - imp) / imp, MJ_MINVAL);
    for (int e = 0; e < 4; e++, r++) {
      int tan = 1 + e / 2;
      real f = (e % 2 == 0) ? (real)m->con_friction[c][tan - 1] : -(real)m->con_friction[c][tan - 1];
      d->efc_on[r] = on;
      real vel = 0;
      for (int i = 0; i < nv; i++) {
        real j = on ? dcon[0][i] + dcon[tan][i] * f : 0;
        d->efc_J[r][i] = j;
        vel += j * d->qvel[i];
      }
      d->efc_aref[r] = on ? -b_ * vel - k_ * imp * pos : 0;
      d->efc_D[r] = on ? 1 / R : 0;
    }
  }
}

/* ------------------------------------------------------------------ smooth.com_vel / passive / rne */
static void com_vel(const dial_model* m, odata* d) {
  for (int k = 0; k < 6; k++) d->cvel[0][k] = 0;
  for (int b = 1; b < m->nbody; b++) {
    real cvel[6];
    for (int k = 0; k < 6; k++) cvel[k] = d->cvel[m->body_parent[b]][k];
    for (int ji = m->body_jntadr[b]; ji < m->body_jntadr[b] + m->body_jntnum[b]; ji++) {
      int da = m->jnt_dofadr[ji];
      if (m->jnt_type[ji] == DIAL_JNT_FREE) {
        for (int i = 0; i < 3; i++) for (int k = 0; k < 6; k++) cvel[k] += d->cdof[da + i][k] * d->qvel[da + i];
        for (int i = 0; i < 3; i++) for (int k = 0; k < 6; k++) d->cdof_dot[da + i][k] = 0;
        for (int i = 3; i < 6; i++) motion_cross(d->cdof_dot[da + i], cvel, d->cdof[da + i]);
        for (int i = 3; i < 6; i++) for (int k = 0; k < 6; k++) cvel[k] += d->cdof[da + i][k] * d->qvel[da + i];
      } else {
        motion_cross(d->cdof_dot[da], cvel, d->cdof[da]);
        for (int k = 0; k < 6; k++) cvel[k] += d->cdof[da][k] * d->qvel[da];
      }
    }
    for (int k = 0; k < 6; k++) d->cvel[b][k] = cvel[k];
  }
}
static void rne(const dial_model* m, odata* d) {
  real cacc[NB][6], cfrc[NB][6];
  for (int k = 0; k < 3; k++) { cacc[0][k] = 0; cacc[0][3 + k] = -(real)m->gravity[k]; }
  for (int b = 1; b < m->nbody; b++) {
    for (int k = 0; k < 6; k++) cacc[b][k] = cacc[m->body_parent[b]][k];
    for (int i = m->body_dofadr[b]; i < m->body_dofadr[b] + m->body_dofnum[b]; i++)
      for (int k = 0; k < 6; k++) cacc[b][k] += d->cdof_dot[i][k] * d->qvel[i];
  }
  for (int b = 0; b < m->nbody; b++) {
    real f1[6], f2[6], f3[6];
    inert_mul(f1, d->cinert[b], cacc[b]);
    inert_mul(f2, d->cinert[b], d->cvel[b]);
    motion_cross_force(f3, d->cvel[b], f2);
    for (int k = 0; k < 6; k++) cfrc[b][k] = f1[k] + f3[k];
  }
  for (int b = m->nbody - 1; b > 0; b--)
    for (int k = 0; k < 6; k++) cfrc[m->body_parent[b]][k] += cfrc[b][k];
  for (int i = 0; i < m->nv; i++) {
    real s = 0;
    for (int k = 0; k < 6; k++) s += d->cdof[i][k] * cfrc[m->dof_bodyid[i]][k];
    d->qfrc_bias[i] = s;
  }
}

/* ------------------------------------------------------------------ solver.solve (Newton, dense) */
/* optional decision trace of the current thread (oracle_rollout_trace): which discrete choices the solver made */
#define TRACE_N 8
static __thread int* g_trace = 0; /* [use_warm, niter, nactive_start, nactive_end, ls_iters_total, improved_mask, ncontact_on, nlimit_on] */
typedef struct {
  real qacc[NV], Ma[NV], Jaref[NE], grad[NV], Mgrad[NV], search[NV], qfrc_constraint[NV], efc_force[NE];
  int active[NE];
  real gauss, cost, prev_cost;
  /* elliptic cones: per contact zone (0 top, 1 middle, 2 bottom) and the middle-zone quantities */
  int zone[NC];
  real cU[NC][6], cN[NC], cT[NC], cDm[NC], cmu[NC];
} sctx;

static void mul_m(const dial_model* m, const odata* d, const real* v, real* out) {
  for (int i = 0; i < m->nv; i++) {
    real s = 0;
    for (int j = 0; j < m->nv; j++) s += d->qM[i][j] * v[j];
    out[i] = s;
  }
}
/* solver._update_constraint for elliptic cones: zone of every contact, forces, cost of the constraint part */
static real update_constraint_elliptic(const dial_model* m, const odata* d, sctx* c) {
  real cost = 0;
  for (int r = 0; r < m->nlim; r++) {
    c->active[r] = c->Jaref[r] < 0;
    c->efc_force[r] = d->efc_D[r] * -c->Jaref[r] * (c->active[r] ? 1 : 0);
    cost += (real)0.5 * d->efc_D[r] * c->Jaref[r] * c->Jaref[r] * (c->active[r] ? 1 : 0);
  }
  int r0 = m->nlim + m->nfri;
  for (int k = 0; k < m->ncon; k++) {
    int dim = m->con_dim[k];
    real mu = (real)m->con_friction[k][0] / r_sqrt((real)m->impratio);
    real U[6], tsqr = 0;
    U[0] = c->Jaref[r0] * mu;
    for (int j = 1; j < dim; j++) { U[j] = c->Jaref[r0 + j] * (real)m->con_friction[k][j - 1]; tsqr += U[j] * U[j]; }
    real N = U[0], T = r_sqrt(tsqr);
    int bottom = (tsqr <= 0 && N < 0) || (tsqr > 0 && mu * N + T <= 0);
    int middle = tsqr > 0 && N < mu * T && mu * N + T > 0;
    c->zone[k] = bottom ? 2 : (middle ? 1 : 0);
    c->cmu[k] = mu; c->cN[k] = N; c->cT[k] = T;
    for (int j = 0; j < 6; j++) c->cU[k][j] = j < dim ? U[j] : 0;
    real Dm = d->efc_D[r0] / r_max(mu * mu * (1 + mu * mu), MJ_MINVAL);
    c->cDm[k] = Dm;
    for (int j = 0; j < dim; j++) { c->active[r0 + j] = bottom; c->efc_force[r0 + j] = 0; }
    if (bottom) {
      for (int j = 0; j < dim; j++) {
        c->efc_force[r0 + j] = -d->efc_D[r0 + j] * c->Jaref[r0 + j];
        cost += (real)0.5 * d->efc_D[r0 + j] * c->Jaref[r0 + j] * c->Jaref[r0 + j];
      }
    } else if (middle) {
      real nmt = N - mu * T;
      cost += (real)0.5 * Dm * nmt * nmt;
      real fn = -Dm * nmt * mu;
      c->efc_force[r0] = fn;
      for (int j = 1; j < dim; j++) c->efc_force[r0 + j] = -fn / T * U[j] * (real)m->con_friction[k][j - 1];
    }
    r0 += dim;
  }
  return cost;
}
static void update_constraint(const dial_model* m, const odata* d, sctx* c) {
  int nv = m->nv, ne = m->nefc;
  if (m->cone == DIAL_CONE_ELLIPTIC) {
    real ccost = update_constraint_elliptic(m, d, c);
    for (int i = 0; i < nv; i++) {
      real s = 0;
      for (int r = 0; r < ne; r++) s += d->efc_J[r][i] * c->efc_force[r];
      c->qfrc_constraint[i] = s;
    }
    real gauss = 0;
    for (int i = 0; i < nv; i++) gauss += (c->Ma[i] - d->qfrc_smooth[i]) * (c->qacc[i] - d->qacc_smooth[i]);
    gauss *= (real)0.5;
    c->gauss = gauss;
    c->prev_cost = c->cost;
    c->cost = ccost + gauss;
    return;
  }
  real fcost = 0;   /* cost of the dry-friction rows in their LINEAR zones (solver._update_constraint) */
  for (int r = 0; r < ne; r++) {
    if (d->efc_floss[r] > 0) {
      /* quadratic while |D Jaref| < frictionloss, i.e. |Jaref| < rf = R * frictionloss; beyond: force = -+frictionloss and
       * cost = frictionloss * (-0.5 rf -+ Jaref) */
      real f = d->efc_floss[r], rf = f / d->efc_D[r], j = c->Jaref[r];
      int neg = j <= -rf, pos = j >= rf;
      c->active[r] = !neg && !pos;
      c->efc_force[r] = neg ? f : (pos ? -f : d->efc_D[r] * -j);
      if (neg) fcost += f * (-(real)0.5 * rf - j);
      if (pos) fcost += f * (-(real)0.5 * rf + j);
      continue;
    }
    c->active[r] = c->Jaref[r] < 0;
    c->efc_force[r] = d->efc_D[r] * -c->Jaref[r] * (c->active[r] ? 1 : 0);
  }
  for (int i = 0; i < nv; i++) {
    real s = 0;
    for (int r = 0; r < ne; r++) s += d->efc_J[r][i] * c->efc_force[r];
    c->qfrc_constraint[i] = s;
  }
  real gauss = 0;
  for (int i = 0; i < nv; i++) gauss += (c->Ma[i] - d->qfrc_smooth[i]) * (c->qacc[i] - d->qacc_smooth[i]);
  gauss *= (real)0.5;
  real cost = 0;
  for (int r = 0; r < ne; r++) cost += d->efc_D[r] * c->Jaref[r] * c->Jaref[r] * (c->active[r] ? 1 : 0);
  cost = (real)0.5 * cost + fcost + gauss;
  c->gauss = gauss;
  c->prev_cost = c->cost;
  c->cost = cost;
}
static void update_gradient(const dial_model* m, const odata* d, sctx* c) {
  int nv = m->nv, ne = m->nefc;
  for (int i = 0; i < nv; i++) c->grad[i] = c->Ma[i] - d->qfrc_smooth[i] - c->qfrc_constraint[i];
  static __thread real H[NV][NV], L[NV][NV];
  for (int i = 0; i < nv; i++)
    for (int j = 0; j < nv; j++) {
      real s = 0;
      for (int r = 0; r < ne; r++) if (c->active[r]) s += d->efc_J[r][i] * d->efc_D[r] * d->efc_J[r][j];
      H[i][j] = d->qM[i][j] + s;
    }
  if (m->cone == DIAL_CONE_ELLIPTIC) {
    /* cone Hessian of the middle-zone contacts: H += J_c^T Hc J_c,
     * Hc = Dm diag(mu, f) [[1, -mu U^T / T], [-mu U / T, mu N / T^3 U U^T + (mu^2 - mu N / T) I]] diag(mu, f) */
    int r0 = m->nlim + m->nfri;
    for (int k = 0; k < m->ncon; k++) {
      int dim = m->con_dim[k];
      if (c->zone[k] == 1) {
        real mu = c->cmu[k], N = c->cN[k], T = r_max(c->cT[k], MJ_MINVAL), TTT = r_max(T * T * T, MJ_MINVAL), Dm = c->cDm[k];
        real Hc[6][6], fri[6];
        fri[0] = mu;
        for (int a = 1; a < dim; a++) fri[a] = (real)m->con_friction[k][a - 1];
        for (int a = 0; a < dim; a++)
          for (int b = 0; b < dim; b++) {
            real v;
            if (a == 0 && b == 0) v = 1;
            else if (a == 0) v = -mu / T * c->cU[k][b];
            else if (b == 0) v = -mu / T * c->cU[k][a];
            else v = mu * N / TTT * c->cU[k][a] * c->cU[k][b] + (a == b ? mu * mu - mu * N / T : 0);
            Hc[a][b] = v * Dm * fri[a] * fri[b];
          }
        for (int i = 0; i < nv; i++)
          for (int j = 0; j < nv; j++) {
            real s = 0;
            for (int a = 0; a < dim; a++) {
              real t = 0;
              for (int b = 0; b < dim; b++) t += Hc[a][b] * d->efc_J[r0 + b][j];
              s += d->efc_J[r0 + a][i] * t;
            }
            H[i][j] += s;
          }
      }
      r0 += dim;
    }
  }
  cholesky(nv, H, L);
  cho_solve(nv, L, c->grad, c->Mgrad);
}
static void ctx_create(const dial_model* m, const odata* d, const real* qacc, sctx* c, int grad) {
  int nv = m->nv, ne = m->nefc;
  for (int i = 0; i < nv; i++) c->qacc[i] = qacc[i];
  for (int r = 0; r < ne; r++) {
    real s = 0;
    for (int i = 0; i < nv; i++) s += d->efc_J[r][i] * qacc[i];
    c->Jaref[r] = s - d->efc_aref[r];
  }
  mul_m(m, d, qacc, c->Ma);
  c->cost = (real)INFINITY; c->prev_cost = 0; c->gauss = 0;
  for (int i = 0; i < nv; i++) { c->grad[i] = 0; c->Mgrad[i] = 0; c->search[i] = 0; }
  update_constraint(m, d, c);
  if (grad) {
    update_gradient(m, d, c);
    for (int i = 0; i < nv; i++) c->search[i] = -c->Mgrad[i];
  }
}
typedef struct { real alpha, cost, deriv_0, deriv_1; } lspoint;
static lspoint ls_point(int ne, const odata* d, const sctx* c, real alpha, const real* jv, real quad[][3], const real* quad_gauss) {
  real qt[3] = {quad_gauss[0], quad_gauss[1], quad_gauss[2]};
  for (int r = 0; r < ne; r++) {
    real x = c->Jaref[r] + alpha * jv[r];
    if (d->efc_floss[r] > 0) {   /* dry friction (solver._eval_pt): quadratic inside |x| < rf, linear outside */
      real f = d->efc_floss[r], rf = f / d->efc_D[r];
      if (x <= -rf) { qt[0] += f * (-(real)0.5 * rf - c->Jaref[r]); qt[1] += -f * jv[r]; }
      else if (x >= rf) { qt[0] += f * (-(real)0.5 * rf + c->Jaref[r]); qt[1] += f * jv[r]; }
      else { qt[0] += quad[r][0]; qt[1] += quad[r][1]; qt[2] += quad[r][2]; }
      continue;
    }
    if (x < 0) { qt[0] += quad[r][0]; qt[1] += quad[r][1]; qt[2] += quad[r][2]; }
  }
  lspoint p;
  p.alpha = alpha;
  p.cost = alpha * alpha * qt[2] + alpha * qt[1] + qt[0];
  /* The slope is evaluated with ONE rounding (fused multiply-add), as XLA does on the reference's platform.  With two
   * roundings (this file is compiled -ffp-contract=off) the slope at a Newton point alpha = -q1 / (2 q2) evaluates to
   * exactly 0 about half of the time; `_in_bracket` rejects a zero-slope candidate and the 5-iteration search falls
   * back to bisection -- a rounding lottery that a contracting compiler (XLA, hipcc) does not play. */
  p.deriv_0 = r_fma(2 * alpha, qt[2], qt[1]);
  p.deriv_1 = 2 * qt[2] + (qt[2] == 0 ? MJ_MINVAL : 0);
  return p;
}
/* solver._eval_pt_elliptic: limit rows as usual, contacts by zone at alpha; quad_c = per-contact sum of its rows'
 * quadratics (bottom zone), cone = (U0, V0, UU, UV, VV, Dm, mu) per contact */
static lspoint ls_point_elliptic(const dial_model* m, const sctx* c, real alpha, const real* jv, real quad[][3],
                                 const real* quad_gauss, real quad_c[][3], real cone[][7]) {
  real qt[3] = {quad_gauss[0], quad_gauss[1], quad_gauss[2]};
  for (int r = 0; r < m->nlim; r++) {
    real x = c->Jaref[r] + alpha * jv[r];
    if (x < 0) { qt[0] += quad[r][0]; qt[1] += quad[r][1]; qt[2] += quad[r][2]; }
  }
  real ccost = 0, cd0 = 0, cd1 = 0;
  for (int k = 0; k < m->ncon; k++) {
    real u0 = cone[k][0], v0 = cone[k][1], uu = cone[k][2], uv = cone[k][3], vv = cone[k][4], dm = cone[k][5], mu = cone[k][6];
    real n = u0 + alpha * v0;
    real tsqr = uu + alpha * (2 * uv + alpha * vv);
    real t = r_sqrt(tsqr);
    int bottom = (tsqr <= 0 && n < 0) || (tsqr > 0 && mu * n + t <= 0);
    int middle = tsqr > 0 && n < mu * t && mu * n + t > 0;
    if (bottom) { qt[0] += quad_c[k][0]; qt[1] += quad_c[k][1]; qt[2] += quad_c[k][2]; }
    if (middle) {
      real n1 = v0, t1 = (uv + alpha * vv) / t, t2 = vv / t - (uv + alpha * vv) * t1 / (t * t);
      real nmt = n - mu * t;
      ccost += (real)0.5 * dm * nmt * nmt;
      cd0 += dm * nmt * (n1 - mu * t1);
      cd1 += dm * ((n1 - mu * t1) * (n1 - mu * t1) - nmt * mu * t2);
    }
  }
  lspoint p;
  p.alpha = alpha;
  p.cost = alpha * alpha * qt[2] + alpha * qt[1] + qt[0] + ccost;
  p.deriv_0 = r_fma(2 * alpha, qt[2], qt[1]) + cd0;
  p.deriv_1 = 2 * qt[2] + cd1;
  if (p.deriv_1 == 0) p.deriv_1 = MJ_MINVAL;
  return p;
}
static real vnorm(int n, const real* v) {
  real s = 0;
  for (int i = 0; i < n; i++) s += v[i] * v[i];
  return r_sqrt(s);
}
static void linesearch(const dial_model* m, const odata* d, sctx* c) {
  int nv = m->nv, ne = m->nefc;
  real scale = (real)m->meaninertia * (real)(nv > 1 ? nv : 1);
  real smag = vnorm(nv, c->search) * scale;
  real gtol = (real)m->tolerance * (real)m->ls_tolerance * smag;
  real mv[NV], jv[NE], quad[NE][3], quad_gauss[3];
  mul_m(m, d, c->search, mv);
  for (int r = 0; r < ne; r++) {
    real s = 0;
    for (int i = 0; i < nv; i++) s += d->efc_J[r][i] * c->search[i];
    jv[r] = s;
  }
  real s1 = 0, s2 = 0, s3 = 0;
  for (int i = 0; i < nv; i++) { s1 += c->search[i] * c->Ma[i]; s2 += c->search[i] * d->qfrc_smooth[i]; s3 += c->search[i] * mv[i]; }
  quad_gauss[0] = c->gauss; quad_gauss[1] = s1 - s2; quad_gauss[2] = (real)0.5 * s3;
  for (int r = 0; r < ne; r++) {
    quad[r][0] = (real)0.5 * c->Jaref[r] * c->Jaref[r] * d->efc_D[r];
    quad[r][1] = jv[r] * c->Jaref[r] * d->efc_D[r];
    quad[r][2] = (real)0.5 * jv[r] * jv[r] * d->efc_D[r];
  }
  static __thread real quad_c[NC][3], cone[NC][7];
  const int ell = m->cone == DIAL_CONE_ELLIPTIC;
  if (ell) {
    int r0 = m->nlim + m->nfri;
    for (int k = 0; k < m->ncon; k++) {
      int dim = m->con_dim[k];
      real mu = (real)m->con_friction[k][0] / r_sqrt((real)m->impratio);
      quad_c[k][0] = quad_c[k][1] = quad_c[k][2] = 0;
      for (int j = 0; j < dim; j++) for (int q = 0; q < 3; q++) quad_c[k][q] += quad[r0 + j][q];
      real uu = 0, uv = 0, vv = 0;
      for (int j = 1; j < dim; j++) {
        real f = (real)m->con_friction[k][j - 1], u = c->Jaref[r0 + j] * f, v = jv[r0 + j] * f;
        uu += u * u; uv += u * v; vv += v * v;
      }
      cone[k][0] = c->Jaref[r0] * mu; cone[k][1] = jv[r0] * mu; cone[k][2] = uu; cone[k][3] = uv; cone[k][4] = vv;
      cone[k][5] = d->efc_D[r0] / r_max(mu * mu * (1 + mu * mu), MJ_MINVAL); cone[k][6] = mu;
      r0 += dim;
    }
  }
#define LS_POINT(a) (ell ? ls_point_elliptic(m, c, (a), jv, quad, quad_gauss, quad_c, cone) : ls_point(ne, d, c, (a), jv, quad, quad_gauss))
  lspoint p0 = LS_POINT(0);
  lspoint lo = LS_POINT(p0.alpha - p0.deriv_0 / p0.deriv_1), hi;
  if (lo.deriv_0 < p0.deriv_0) { hi = p0; } else { hi = lo; lo = p0; }
  /* bracket refinement of current MJX (the release line that supports elliptic cones, which the reference's Allegro
   * env needs): a candidate y replaces a bracket end x only when it lies on the same side of the minimum and closer
   * to it (`_in_bracket`); each end is offered its own Newton step, the mid-point and the OTHER end's Newton step. */
#define IN_BRACKET(x, y) ((((x).deriv_0 < (y).deriv_0) && ((y).deriv_0 < 0)) || (((x).deriv_0 > (y).deriv_0) && ((y).deriv_0 > 0)))
  int swap = 1, ls_iter = 0;
  for (;;) {
    int done = ls_iter >= m->ls_iterations;
    done |= !swap;
    done |= (lo.deriv_0 < 0) && (lo.deriv_0 > -gtol);
    done |= (hi.deriv_0 > 0) && (hi.deriv_0 < gtol);
    if (done) break;
    lspoint lo_next = LS_POINT(lo.alpha - lo.deriv_0 / lo.deriv_1);
    lspoint hi_next = LS_POINT(hi.alpha - hi.deriv_0 / hi.deriv_1);
    lspoint mid = LS_POINT((real)0.5 * (lo.alpha + hi.alpha));
    if (m->ls_rule == DIAL_LS_SWAP) { /* the rule of MJX <= 3.1.3 */
      int swap_lo_next = (lo.deriv_0 > 0) || (lo.deriv_0 < lo_next.deriv_0);
      if (swap_lo_next) lo = lo_next;
      int swap_lo_mid = (mid.deriv_0 < 0) && (lo.deriv_0 < mid.deriv_0);
      if (swap_lo_mid) lo = mid;
      int swap_hi_next = (hi.deriv_0 < 0) || (hi.deriv_0 > hi_next.deriv_0);
      if (swap_hi_next) hi = hi_next;
      int swap_hi_mid = (mid.deriv_0 > 0) && (hi.deriv_0 > mid.deriv_0);
      if (swap_hi_mid) hi = mid;
      swap = swap_lo_next || swap_lo_mid || swap_hi_next || swap_hi_mid;
      ls_iter++;
      continue;
    }
    int s1 = IN_BRACKET(lo, lo_next);
    if (s1) lo = lo_next;
    int s2 = IN_BRACKET(lo, mid);
    if (s2) lo = mid;
    int s3 = IN_BRACKET(lo, hi_next);
    if (s3) lo = hi_next;
    int s4 = IN_BRACKET(hi, hi_next);
    if (s4) hi = hi_next;
    int s5 = IN_BRACKET(hi, mid);
    if (s5) hi = mid;
    int s6 = IN_BRACKET(hi, lo_next);
    if (s6) hi = lo_next;
    swap = s1 || s2 || s3 || s4 || s5 || s6;
    ls_iter++;
  }
#undef IN_BRACKET
  int improved = (lo.cost < p0.cost) || (hi.cost < p0.cost);
  real alpha = lo.cost < hi.cost ? lo.alpha : hi.alpha;
  if (g_trace) { g_trace[4] += ls_iter; g_trace[5] = (g_trace[5] << 1) | improved; }
  if (improved) {
    for (int i = 0; i < nv; i++) { c->qacc[i] += c->search[i] * alpha; c->Ma[i] += mv[i] * alpha; }
    for (int r = 0; r < ne; r++) c->Jaref[r] += jv[r] * alpha;
  }
#undef LS_POINT
}
static void solve(const dial_model* m, odata* d) {
  int nv = m->nv;
  static __thread sctx warm, smth, c;
  ctx_create(m, d, d->qacc_warmstart, &warm, 0);
  ctx_create(m, d, d->qacc_smooth, &smth, 0);
  const real* q0 = warm.cost < smth.cost ? d->qacc_warmstart : d->qacc_smooth;
  ctx_create(m, d, q0, &c, 1);
  if (g_trace) {
    g_trace[0] = warm.cost < smth.cost;
    g_trace[2] = 0;
    for (int r = 0; r < m->nefc; r++) g_trace[2] += c.active[r] && d->efc_D[r] > 0;
    g_trace[4] = 0; g_trace[5] = 0; g_trace[6] = 0; g_trace[7] = 0;
    for (int r = 0; r < m->nefc; r++) { if (d->efc_D[r] > 0) { if (r < m->nlim) g_trace[7]++; else g_trace[6]++; } }
  }
  real scale = 1 / ((real)m->meaninertia * (real)(nv > 1 ? nv : 1));
  int niter = 0;
  for (;;) {
    if (m->iterations != 1) { /* lax.while_loop cond (iterations == 1 runs the body once, unconditionally) */
      real improvement = scale * (c.prev_cost - c.cost);
      real gradient = scale * vnorm(nv, c.grad);
      int done = niter >= m->iterations;
      done |= improvement < (real)m->tolerance;
      done |= gradient < (real)m->tolerance;
      if (done) break;
    } else if (niter >= 1) break;
    linesearch(m, d, &c);
    update_constraint(m, d, &c);
    update_gradient(m, d, &c);
    for (int i = 0; i < nv; i++) c.search[i] = -c.Mgrad[i];
    niter++;
  }
  d->solver_niter = niter;
  if (g_trace) {
    g_trace[1] = niter;
    g_trace[3] = 0;
    for (int r = 0; r < m->nefc; r++) g_trace[3] += c.active[r] && d->efc_D[r] > 0;
  }
  for (int i = 0; i < nv; i++) { d->qacc[i] = c.qacc[i]; d->qacc_warmstart[i] = c.qacc[i]; d->qfrc_constraint[i] = c.qfrc_constraint[i]; }
  for (int r = 0; r < m->nefc; r++) d->efc_force[r] = c.efc_force[r];
}

/* ------------------------------------------------------------------ forward.forward / euler / step */
static void forward(const dial_model* m, odata* d) {
  int nv = m->nv;
  kinematics(m, d);
  com_pos(m, d);
  crb(m, d);
  collision(m, d);
  make_constraint(m, d);
  com_vel(m, d);
  for (int i = 0; i < nv; i++) d->qfrc_passive[i] = -(real)m->dof_damping[i] * d->qvel[i];
  rne(m, d);
  /* fwd_actuation */
  for (int i = 0; i < nv; i++) d->qfrc_actuator[i] = 0;
  for (int a = 0; a < m->nu; a++) {
    real ctrl = d->ctrl[a];
    if (m->act_ctrllimited[a]) ctrl = r_clip(ctrl, (real)m->act_ctrlrange[a][0], (real)m->act_ctrlrange[a][1]);
    real force = m->act_isposition[a] ? (real)m->act_kp[a] * (ctrl - d->qpos[m->act_qposadr[a]]) : ctrl;
    d->qfrc_actuator[m->act_dofadr[a]] += (real)m->act_gear[a] * force;
  }
  /* fwd_acceleration */
  for (int i = 0; i < nv; i++) d->qfrc_smooth[i] = d->qfrc_passive[i] - d->qfrc_bias[i] + d->qfrc_actuator[i];
  cho_solve(nv, d->qL, d->qfrc_smooth, d->qacc_smooth);
  if (m->nefc == 0) {
    for (int i = 0; i < nv; i++) d->qacc[i] = d->qacc_smooth[i];
    return;
  }
  solve(m, d);
}
static void euler(const dial_model* m, odata* d) {
  int nv = m->nv;
  real dt = m->timestep, qacc[NV];
  for (int i = 0; i < nv; i++) qacc[i] = d->qacc[i];
  if (m->eulerdamp) {
    static __thread real H[NV][NV], L[NV][NV];
    real f[NV];
    for (int i = 0; i < nv; i++) {
      for (int j = 0; j < nv; j++) H[i][j] = d->qM[i][j];
      H[i][i] += dt * (real)m->dof_damping[i];
      f[i] = d->qfrc_smooth[i] + d->qfrc_constraint[i];
    }
    cholesky(nv, H, L);
    cho_solve(nv, L, f, qacc);
  }
  for (int i = 0; i < nv; i++) d->qvel[i] += qacc[i] * dt;
  for (int ji = 0; ji < m->njnt; ji++) {
    int qa = m->jnt_qposadr[ji], da = m->jnt_dofadr[ji];
    if (m->jnt_type[ji] == DIAL_JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += dt * d->qvel[da + k];
      /* math.quat_integrate */
      real v[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]};
      real nrm = r_sqrt(dot3(v, v)), axis[3] = {1, 0, 0}, qr[4], qn[4];
      if (nrm > 0) for (int k = 0; k < 3; k++) axis[k] = v[k] / nrm;
      axis_angle_to_quat(qr, axis, dt * nrm);
      quat_mul(qn, d->qpos + qa + 3, qr);
      normalize4(qn);
      for (int k = 0; k < 4; k++) d->qpos[qa + 3 + k] = qn[k];
    } else {
      d->qpos[qa] += dt * d->qvel[da];
    }
  }
}

/* ------------------------------------------------------------------ env: control (base_env.py:38-66) */
static void act2joint(const dial_model* m, const dial_task* t, const real* act, real* jt) {
  for (int a = 0; a < m->nu; a++) {
    real an = (act[a] * (real)t->action_scale + (real)1.0) / (real)2.0;
    /* joint_offset: AllegroReorientEnv.act2joint adds the keyframe pose init_q[7:] (manipulation.py:107-109); 0 elsewhere */
    real v = ((real)t->joint_range[a][0] + (real)t->joint_offset[a]) + an * ((real)t->joint_range[a][1] - (real)t->joint_range[a][0]);
    jt[a] = r_clip(v, (real)t->phys_range[a][0], (real)t->phys_range[a][1]);
  }
}
static void act2tau(const dial_model* m, const dial_task* t, const real* act, const real* qpos, const real* qvel, real* tau) {
  real jt[NU];
  act2joint(m, t, act, jt);
  for (int a = 0; a < m->nu; a++) {
    real q_err = jt[a] - qpos[7 + a];
    real v = (real)t->kp[a] * q_err - (real)t->kd[a] * qvel[6 + a];
    tau[a] = r_clip(v, (real)t->tau_range[a][0], (real)t->tau_range[a][1]);
  }
}

/* function_utils.py:18-43 */
static real foot_step_height(real tt, real footphase, real duty) {
  real two_pi = 2 * R_PI;
  real x = tt + R_PI - footphase;
  real angle = x - two_pi * r_floor(x / two_pi) - R_PI; /* jnp `%` is a floored modulo */
  if (duty < 1) angle = angle * (real)0.5 / (1 - duty);
  real clipped = r_clip(angle, -R_PI / 2, R_PI / 2);
  real value = duty < 1 ? r_cos(clipped) : 0;
  return r_abs(value) >= (real)1e-6 ? r_abs(value) : 0;
}
static void get_foot_step(const dial_task* t, real time, real* h) {
  for (int f = 0; f < t->nfeet; f++)
    h[f] = (real)t->gait_amp * foot_step_height(time * 2 * R_PI * (real)t->gait_cadence + R_PI,
                                                2 * R_PI * (real)t->gait_phase[f], (real)t->gait_duty);
}
/* brax.math.quat_to_euler(q)[2] */
static real quat_yaw(const real* q) {
  return r_atan2(-2 * q[1] * q[2] + 2 * q[0] * q[3], q[1] * q[1] + q[0] * q[0] - q[3] * q[3] - q[2] * q[2]);
}

/* state <-> odata */
static void load_state(const dial_model* m, const real* state, odata* d, real* info) {
  for (int i = 0; i < m->nq; i++) d->qpos[i] = state[i];
  for (int i = 0; i < m->nv; i++) { d->qvel[i] = state[m->nq + i]; d->qacc_warmstart[i] = state[m->nq + m->nv + i]; }
  for (int i = 0; i < DIAL_INFO_N; i++) info[i] = state[m->nq + 2 * m->nv + i];
}
static void store_state(const dial_model* m, const odata* d, const real* info, real* state) {
  for (int i = 0; i < m->nq; i++) state[i] = d->qpos[i];
  for (int i = 0; i < m->nv; i++) { state[m->nq + i] = d->qvel[i]; state[m->nq + m->nv + i] = d->qacc_warmstart[i]; }
  for (int i = 0; i < DIAL_INFO_N; i++) state[m->nq + 2 * m->nv + i] = info[i];
}

/* One env.step on (d, info).  Returns the reward. */
static real env_step(const dial_model* m, const dial_task* t, odata* d, real* info, const real* action) {
  real dt = t->dt;
  real ctrl[NU];
  if (t->position_control) act2joint(m, t, action, ctrl);
  else act2tau(m, t, action, d->qpos, d->qvel, ctrl);
  for (int a = 0; a < m->nu; a++) d->ctrl[a] = ctrl[a];
  for (int f = 0; f < t->n_frames; f++) { forward(m, d); euler(m, d); } /* pipeline_step */
  /* x, xd (brax.mjx.pipeline; reference copy dial_plan.py:53-58): from the PRE-integration forward pass */
  int tb = t->torso_x + 1, ub = t->upright_x + 1;
  const real* rot_t = d->xquat[tb];
  const real* c = d->subtree_com[m->body_rootid[tb]];
  real off[3] = {d->xpos[tb][0] - c[0], d->xpos[tb][1] - c[1], d->xpos[tb][2] - c[2]};
  real ang[3] = {d->cvel[tb][0], d->cvel[tb][1], d->cvel[tb][2]}, cr[3], vel[3];
  cross3(cr, off, ang);
  for (int k = 0; k < 3; k++) vel[k] = d->cvel[tb][3 + k] - cr[k];
  real step = info[DIAL_INFO_STEP];
  real reward = 0;
  real up[3] = {0, 0, 1}, vec[3];
  rotate(vec, up, d->xquat[ub]);
  real reward_upright = -((vec[0] - 0) * (vec[0] - 0) + (vec[1] - 0) * (vec[1] - 0) + (vec[2] - 1) * (vec[2] - 1));
  real yaw = quat_yaw(rot_t);
  if (t->kind == DIAL_TASK_ALLEGRO) { /* manipulation.py:75-100; torso_x = the object body */
    real e_ang = 0, e_pos = 0, e_jnt = 0;
    for (int k = 0; k < 3; k++) {
      real a = ang[k] * R_PI / (real)180.0 - info[DIAL_INFO_ANG_VEL_TAR + k];
      real p = d->xpos[tb][k] - info[DIAL_INFO_POS_TAR + k];
      e_ang += a * a;
      e_pos += p * p;
    }
    for (int a = 0; a < m->nu; a++) { real e = d->qpos[7 + a] - (real)t->joint_offset[a]; e_jnt += e * e; }
    reward = -e_ang * (real)1.0 + -e_pos * (real)5.0 + -e_jnt * (real)0.1;
    info[DIAL_INFO_DONE] = step >= 100 ? 1 : 0;
    info[DIAL_INFO_STEP] = step + 1;
    info[DIAL_INFO_REWARD] = reward;
    return reward;
  }
  if (t->kind == DIAL_TASK_GO2_CRATE) { /* UnitreeGo2CrateEnv.step, unitree_go2_env.py:679-795 */
    real z_tar[DIAL_MAX_FEET], reward_gaits = 0;
    get_foot_step(t, step * dt, z_tar);                                             /* :697-703 */
    for (int f = 0; f < t->nfeet; f++) { real e = (z_tar[f] - d->site_xpos[t->feet_site[f]][2]) / (real)0.05; reward_gaits += e * e; }
    reward_gaits = -reward_gaits;
    real R[9], head[3], reward_pos = 0;                                             /* :705-719 */
    quat_to_mat(R, rot_t);
    for (int k = 0; k < 3; k++) {
      real pos_tar = info[DIAL_INFO_POS_TAR + k] + info[DIAL_INFO_VEL_TAR + k] * dt * step;
      head[k] = d->xpos[tb][k] + (R[3 * k] * (real)t->head_vec[0] + R[3 * k + 1] * (real)t->head_vec[1] + R[3 * k + 2] * (real)t->head_vec[2]);
      reward_pos += (head[k] - pos_tar) * (head[k] - pos_tar);
    }
    reward_pos = -reward_pos;
    real reward_yaw = -(yaw - info[DIAL_INFO_YAW_TAR]) * (yaw - info[DIAL_INFO_YAW_TAR]);   /* :724-727 */
    real reward_vel = 0, reward_height, reward_energy = 0;                          /* :730-740 */
    for (int k = 0; k < 3; k++) { real e = vel[k] - info[DIAL_INFO_VEL_TAR + k]; reward_vel += e * e; }
    reward_vel = -reward_vel;
    reward_height = -(d->xpos[tb][2] - info[DIAL_INFO_POS_TAR + 2]) * (d->xpos[tb][2] - info[DIAL_INFO_POS_TAR + 2]);
    for (int a = 0; a < m->nu; a++) { real e = r_max(ctrl[a] * d->qvel[6 + a] / (real)160.0, 0); reward_energy += e * e; }
    reward_energy = -reward_energy;
    /* pitch / roll (:741-746) are multiplied by 0.0 below; quat_to_euler is finite for every unit quaternion, so they are
     * left out (an exact zero is added either way) */
    real reward_contact = 0, penalty_contact = 0;                                   /* :748-767 */
    for (int c2 = 0; c2 < m->ncon; c2++) {
      int pen = d->con_dist[c2] <= (real)0.001;
      for (int i = 0; i < 4; i++)
        if (c2 == i) {   /* `penalty_contact.at[i]`: upstream indexes the penalty mask by the FOOT number, not by the contact */
          const real* cp = d->con_pos[t->crate_contact[i]];
          int cond = cp[0] > (real)t->crate_region[0] && cp[0] < (real)t->crate_region[1] && cp[1] > (real)t->crate_region[2] &&
                     cp[1] < (real)t->crate_region[3] && cp[2] > (real)t->crate_region[4] && cp[2] < (real)t->crate_region[5];
          pen = pen && !cond;
        }
      penalty_contact += pen ? 1 : 0;
    }
    for (int i = 0; i < 4; i++) {
      const real* cp = d->con_pos[t->crate_contact[i]];
      int cond = cp[0] > (real)t->crate_region[0] && cp[0] < (real)t->crate_region[1] && cp[1] > (real)t->crate_region[2] &&
                 cp[1] < (real)t->crate_region[3] && cp[2] > (real)t->crate_region[4] && cp[2] < (real)t->crate_region[5];
      reward_contact += cond ? 1 : 0;
    }
    reward = reward_gaits * (real)0.0 + reward_pos * (real)1.0 + reward_upright * (real)0.01 + reward_yaw * (real)0.3 +
             reward_vel * (real)0.0 + reward_height * (real)0.0 + reward_energy * (real)0.0000 + (real)0.0 + (real)0.0 +
             reward_contact * (real)0.02 - penalty_contact * (real)0.0;             /* :770-783 */
    info[DIAL_INFO_DONE] = 0;
    info[DIAL_INFO_STEP] = step + 1;
    info[DIAL_INFO_REWARD] = reward;
    return reward;
  }
  if (t->kind == DIAL_TASK_GO2_WALK || t->kind == DIAL_TASK_H1_WALK || t->kind == DIAL_TASK_H1_LOCO || t->kind == DIAL_TASK_H1_PUSH_CRATE) {
    /* unitree_go2_env.py:142-162 / unitree_h1_env.py:196-217: target ramp uses the PRE-increment step */
    for (int k = 0; k < 3; k++) {
      real v = t->cmd_vel[k], a = t->cmd_ang_vel[k];
      /* :150-155: `lax.cond(randomize_target & (step % 500 == 0), randomize, dont_randomize)` -- the sampled command
       * holds for THIS step only (it is not stored), every other step uses the default */
      if (t->randomize_tasks && t->n_cmd > 0 && ((long)step) % 500 == 0) {
        const float* c = t->cmd_table[(((long)step) / 500) % t->n_cmd];
        v = k < 2 ? (real)c[k] : 0;          /* new_lin_vel_cmd = [vx, vy, 0] */
        a = k == 2 ? (real)c[2] : 0;         /* new_ang_vel_cmd = [0, 0, vyaw] */
      }
      info[DIAL_INFO_VEL_TAR + k] = r_min(v * step * dt / (real)t->ramp_up_time, v);
      info[DIAL_INFO_ANG_VEL_TAR + k] = r_min(a * step * dt / (real)t->ramp_up_time, a);
    }
    real z_tar[DIAL_MAX_FEET], z_feet[DIAL_MAX_FEET], reward_gaits = 0;
    get_foot_step(t, step * dt, z_tar);
    int contact[DIAL_MAX_FEET];
    for (int f = 0; f < t->nfeet; f++) {
      real zs = d->site_xpos[t->feet_site[f]][2];
      real fz;
      if (t->kind == DIAL_TASK_GO2_WALK) {
        z_feet[f] = zs;                                           /* unitree_go2_env.py:166 */
        reward_gaits += ((z_tar[f] - z_feet[f]) / (real)0.05) * ((z_tar[f] - z_feet[f]) / (real)0.05);
        fz = zs - (real)t->foot_radius;                           /* :178 */
      } else if (t->kind == DIAL_TASK_H1_PUSH_CRATE) {
        /* unitree_h1_env.py:474-480: min(contact.dist[2:4]), min(contact.dist[6:8]) = the two floor contacts of each foot
         * capsule (include/dial_mpc.h: dial_task.pc_foot_contact) */
        z_feet[f] = r_min(d->con_dist[t->pc_foot_contact[f][0]], d->con_dist[t->pc_foot_contact[f][1]]);
        reward_gaits += (z_tar[f] - z_feet[f]) * (z_tar[f] - z_feet[f]);
        fz = zs;
      } else if (t->kind == DIAL_TASK_H1_WALK) {
        z_feet[f] = r_min(d->con_dist[2 * f], d->con_dist[2 * f + 1]); /* unitree_h1_env.py:230-235 */
        reward_gaits += (z_tar[f] - z_feet[f]) * (z_tar[f] - z_feet[f]);
        fz = zs;                                                  /* :240 */
      } else { /* H1 loco: two capsules (= four contacts) per foot, unitree_h1_env.py:746-752 */
        real z4 = d->con_dist[4 * f];
        for (int q = 1; q < 4; q++) z4 = r_min(z4, d->con_dist[4 * f + q]);
        z_feet[f] = z4;
        reward_gaits += (z_tar[f] - z_feet[f]) * (z_tar[f] - z_feet[f]);
        fz = zs;
      }
      contact[f] = fz < (real)1e-3;
    }
    reward_gaits = -reward_gaits;
    real yaw_tar = info[DIAL_INFO_YAW_TAR] + info[DIAL_INFO_ANG_VEL_TAR + 2] * dt * step;
    real d_yaw = yaw - yaw_tar;
    real wy = r_atan2(r_sin(d_yaw), r_cos(d_yaw));
    real reward_yaw = -(wy * wy);
    real vb[3], ab[3], angs[3] = {ang[0] * R_PI / (real)180.0, ang[1] * R_PI / (real)180.0, ang[2] * R_PI / (real)180.0};
    inv_rotate(vb, vel, rot_t);
    inv_rotate(ab, angs, rot_t);
    real reward_vel = -((vb[0] - info[DIAL_INFO_VEL_TAR]) * (vb[0] - info[DIAL_INFO_VEL_TAR]) +
                        (vb[1] - info[DIAL_INFO_VEL_TAR + 1]) * (vb[1] - info[DIAL_INFO_VEL_TAR + 1]));
    real reward_ang_vel = -((ab[2] - info[DIAL_INFO_ANG_VEL_TAR + 2]) * (ab[2] - info[DIAL_INFO_ANG_VEL_TAR + 2]));
    real dh = d->xpos[tb][2] - info[DIAL_INFO_POS_TAR + 2];
    real reward_height = -(dh * dh);
    if (t->kind == DIAL_TASK_GO2_WALK) { /* unitree_go2_env.py:227-239 */
      reward = reward_gaits * (real)0.1 + reward_upright * (real)0.5 + reward_yaw * (real)0.3 +
               reward_vel * (real)1.0 + reward_ang_vel * (real)1.0 + reward_height * (real)1.0;
    } else if (t->kind == DIAL_TASK_H1_LOCO) { /* unitree_h1_env.py:795-827 */
      real e3 = 0;
      for (int k = 0; k < 3; k++) { real e = ab[k] - info[DIAL_INFO_ANG_VEL_TAR + k]; e3 += e * e; }
      real reward_ang3 = -e3;
      real reward_foot_level = 0;
      for (int f = 0; f < 2; f++) { /* site_xmat @ [0,0,1] = third column of the site frame */
        int sb = m->site_bodyid[t->feet_site[f]];
        real sq[4] = {m->site_quat[t->feet_site[f]][0], m->site_quat[t->feet_site[f]][1], m->site_quat[t->feet_site[f]][2], m->site_quat[t->feet_site[f]][3]};
        real q[4], mat[9];
        quat_mul(q, d->xquat[sb], sq);
        quat_to_mat(mat, q);
        real v[3] = {mat[2], mat[5], mat[8]};
        reward_foot_level += (v[0] - 0) * (v[0] - 0) + (v[1] - 0) * (v[1] - 0) + (v[2] - 1) * (v[2] - 1);
      }
      reward_foot_level = -reward_foot_level;
      real reward_energy = 0;
      for (int a = 0; a < m->nu; a++) { real e = ctrl[a] / (real)t->tau_range[a][1] * d->qvel[6 + a] / (real)160.0; reward_energy += e * e; }
      reward_energy = -reward_energy;
      reward = reward_gaits * (real)10.0 + reward_upright * (real)0.5 + reward_yaw * (real)0.5 +
               reward_vel * (real)1.0 + reward_ang3 * (real)1.0 + reward_height * (real)0.5 +
               reward_foot_level * (real)0.02 + reward_energy * (real)0.01;
    } else if (t->kind == DIAL_TASK_H1_PUSH_CRATE) { /* unitree_h1_env.py:520-548 */
      real reward_energy = 0;
      for (int a = 0; a < m->nu; a++) { real e = ctrl[a] / (real)t->tau_range[a][1]; reward_energy += e * e; }
      reward_energy = -reward_energy;
      /* :525-531: hands on the crate (below 1.1 m) count, every other part of the robot touching it is penalised */
      real reward_contact = 0;
      for (int q = 0; q < 2; q++) {
        int cc = t->pc_wanted[q];
        reward_contact += (d->con_dist[cc] < (real)1e-3 && d->con_pos[cc][2] < (real)t->pc_wanted_zmax) ? 1 : 0;
      }
      for (int q = 0; q < t->pc_n_unwanted; q++) reward_contact -= d->con_dist[t->pc_unwanted[q]] < (real)1e-3 ? 1 : 0;
      /* reward_air_time, reward_pos and reward_alive carry the weight 0.0 upstream and are finite: left out */
      reward = reward_gaits * (real)5.0 + reward_upright * (real)0.01 + reward_yaw * (real)0.1 +
               reward_vel * (real)1.0 + reward_ang_vel * (real)1.0 + reward_height * (real)0.5 +
               reward_energy * (real)0.01 + reward_contact * (real)0.05;
    } else { /* unitree_h1_env.py:282-298 */
      real reward_energy = 0;
      for (int a = 0; a < m->nu; a++) { real e = ctrl[a] / (real)t->tau_range[a][1]; reward_energy += e * e; }
      reward_energy = -reward_energy;
      reward = reward_gaits * (real)5.0 + reward_upright * (real)0.5 + reward_yaw * (real)0.1 +
               reward_vel * (real)1.0 + reward_ang_vel * (real)1.0 + reward_height * (real)0.5 +
               reward_energy * (real)0.01;
    }
    /* info update (unitree_go2_env.py:176-182,251-256) */
    for (int f = 0; f < t->nfeet; f++) {
      int filt = contact[f] || (info[DIAL_INFO_LAST_CONTACT + f] != 0);
      info[DIAL_INFO_AIR_TIME + f] = (info[DIAL_INFO_AIR_TIME + f] + dt) * (filt ? 0 : 1);
      info[DIAL_INFO_LAST_CONTACT + f] = contact[f] ? 1 : 0;
    }
  } else { /* DIAL_TASK_GO2_SEQ_JUMP, unitree_go2_env.py:423-496 */
    int stage = (int)info[DIAL_INFO_STAGE];
    real rp = 0;
    for (int k = 0; k < 3; k++) { real e = d->xpos[tb][k] - (real)t->pose_targets[stage][k]; rp += e * e; }
    real reward_pos = -rp;
    real ey = yaw - (real)t->yaw_targets[stage];
    real reward_yaw = -(ey * ey);
    real reward_contact = 0, penalty_contact = 0;
    for (int i = 0; i < 4; i++) {
      int pen = d->con_dist[i] <= (real)0.001;
      for (int j = 0; j < t->n_stage; j++) {
        real dx = d->con_pos[i][0] - (real)t->contact_targets[j][i][0];
        real dy = d->con_pos[i][1] - (real)t->contact_targets[j][i][1];
        int cond = (dx * dx + dy * dy) <= (real)t->contact_radius[j][i] * (real)t->contact_radius[j][i];
        real val = (j == stage ? 1 : 0) * r_clip(d->con_dist[i] * (real)-1.0 + (real)1.0, 0, 1);
        reward_contact += cond ? val : 0;
        pen = pen && !cond;
      }
      penalty_contact += pen ? 1 : 0;
    }
    reward = reward_pos * (real)1.0 + reward_upright * (real)1.0 + reward_yaw * (real)0.3 +
             reward_contact * (real)0.1 - penalty_contact * (real)0.1 + (real)1.0 * (real)10.0;
    for (int a = 0; a < m->nu; a++) info[DIAL_INFO_LAST_CTRL + a] = ctrl[a];
  }
  /* done (unitree_go2_env.py:242-248, 498-505; unitree_h1_env.py:300-308) */
  real upv[3];
  rotate(upv, up, rot_t);
  int done = upv[2] < 0;
  for (int a = 0; a < m->nu; a++) {
    real q = d->qpos[7 + a];
    done |= q < (real)t->joint_range[a][0];
    done |= q > (real)t->joint_range[a][1];
  }
  done |= d->xpos[tb][2] < (real)t->done_height;
  info[DIAL_INFO_DONE] = done ? 1 : 0;
  info[DIAL_INFO_STEP] = step + 1;
  if (t->kind == DIAL_TASK_GO2_SEQ_JUMP) { /* :512-515, uses the incremented step */
    real st = r_floor(info[DIAL_INFO_STEP] * dt / (real)t->jump_dt);
    info[DIAL_INFO_STAGE] = r_min(st, (real)(t->n_stage - 1));
  }
  info[DIAL_INFO_REWARD] = reward;
  return reward;
}

/* ================================================================== exported API */
int oracle_real_bytes(void) { return (int)sizeof(real); }
int oracle_abi_sizes(int* a, int* b, int* c) { *a = (int)sizeof(dial_model); *b = (int)sizeof(dial_task); *c = (int)sizeof(dial_cfg); return 0; }

/* env.reset: qpos/qvel -> packed state (pipeline_init = mjx.forward with ctrl = 0) */
int oracle_env_reset(const dial_model* m, const dial_task* t, const real* qpos, const real* qvel, real* state,
                     real* xpos_out, real* xquat_out) {
  odata* d = (odata*)calloc(1, sizeof(odata));
  real info[DIAL_INFO_N];
  memset(info, 0, sizeof(info));
  for (int i = 0; i < m->nq; i++) d->qpos[i] = qpos[i];
  for (int i = 0; i < m->nv; i++) d->qvel[i] = qvel[i];
  forward(m, d);
  for (int k = 0; k < 3; k++) { info[DIAL_INFO_POS_TAR + k] = t->init_pos_tar[k]; info[DIAL_INFO_ANG_VEL_TAR + k] = t->init_ang_vel_tar[k]; }
  store_state(m, d, info, state);
  if (xpos_out) for (int b = 1; b < m->nbody; b++) for (int k = 0; k < 3; k++) xpos_out[(b - 1) * 3 + k] = d->xpos[b][k];
  if (xquat_out) for (int b = 1; b < m->nbody; b++) for (int k = 0; k < 4; k++) xquat_out[(b - 1) * 4 + k] = d->xquat[b][k];
  free(d);
  return 0;
}

int oracle_env_step(const dial_model* m, const dial_task* t, real* state, const real* action, real* xpos_out,
                    real* xquat_out, real* ctrl_out) {
  odata* d = (odata*)calloc(1, sizeof(odata));
  real info[DIAL_INFO_N];
  load_state(m, state, d, info);
  env_step(m, t, d, info, action);
  store_state(m, d, info, state);
  if (xpos_out) for (int b = 1; b < m->nbody; b++) for (int k = 0; k < 3; k++) xpos_out[(b - 1) * 3 + k] = d->xpos[b][k];
  if (xquat_out) for (int b = 1; b < m->nbody; b++) for (int k = 0; k < 4; k++) xquat_out[(b - 1) * 4 + k] = d->xquat[b][k];
  if (ctrl_out) for (int a = 0; a < m->nu; a++) ctrl_out[a] = d->ctrl[a];
  free(d);
  return 0;
}

/* rollout_us_vmap (dial_core.py:36-42,80-81): us [B,T,nu] -> rewss [B,T], qss [B,T,nq], qdss [B,T,nv],
 * xposs [B,T,(nbody-1)*3]; optional outputs may be NULL. */
int oracle_rollout(const dial_model* m, const dial_task* t, const real* state, const real* us, int B, int T,
                   real* rewss, real* qss, real* qdss, real* xposs) {
  int nq = m->nq, nv = m->nv, nu = m->nu, nx = (m->nbody - 1) * 3;
#pragma omp parallel
  {
    odata* d = (odata*)calloc(1, sizeof(odata));
    real info[DIAL_INFO_N];
#pragma omp for schedule(dynamic, 4)
    for (int n = 0; n < B; n++) {
      load_state(m, state, d, info);
      for (int s = 0; s < T; s++) {
        real rew = env_step(m, t, d, info, us + ((size_t)n * T + s) * nu);
        size_t o = (size_t)n * T + s;
        rewss[o] = rew;
        if (qss) for (int i = 0; i < nq; i++) qss[o * nq + i] = d->qpos[i];
        if (qdss) for (int i = 0; i < nv; i++) qdss[o * nv + i] = d->qvel[i];
        if (xposs) for (int b = 1; b < m->nbody; b++) for (int k = 0; k < 3; k++) xposs[o * nx + (b - 1) * 3 + k] = d->xpos[b][k];
      }
    }
    free(d);
  }
  return 0;
}

/* One rollout with the solver's decision trace: trace [T][TRACE_N] ints per env.step (last physics sub-step). */
/* noise_mag > 0: before EVERY step qpos, qvel and qacc_warmstart are multiplied by 1 + noise_mag * 2^-24 * u,
 * u ~ U(-1,1) from a 64-bit LCG seeded with noise_seed -- a rounding-level perturbation applied where the next
 * solver decision is taken (a perturbation of the start state alone is forgotten by the dissipative contact
 * dynamics and, for the warm start, after a single step).  Used by the knife-edge witness of the parity tests. */
int oracle_rollout_trace(const dial_model* m, const dial_task* t, const real* state, const real* us, int T, int* trace,
                         real* rewss, real* qss, real* qdss, unsigned long long noise_seed, double noise_mag) {
  odata* d = (odata*)calloc(1, sizeof(odata));
  real info[DIAL_INFO_N];
  load_state(m, state, d, info);
  unsigned long long lcg = noise_seed * 6364136223846793005ULL + 1442695040888963407ULL;
  for (int s = 0; s < T; s++) {
    if (noise_mag > 0) {
      const double sc = noise_mag * 5.9604644775390625e-08;
#define ORACLE_JITTER(x) do { lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; \
        double u_ = (double)(lcg >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0; (x) = (real)((double)(x) * (1.0 + sc * u_)); } while (0)
      for (int i = 0; i < m->nq; i++) ORACLE_JITTER(d->qpos[i]);
      for (int i = 0; i < m->nv; i++) { ORACLE_JITTER(d->qvel[i]); ORACLE_JITTER(d->qacc_warmstart[i]); }
#undef ORACLE_JITTER
    }
    g_trace = trace + s * TRACE_N;
    for (int k = 0; k < TRACE_N; k++) g_trace[k] = 0;
    real rew = env_step(m, t, d, info, us + (size_t)s * m->nu);
    g_trace = 0;
    if (rewss) rewss[s] = rew;
    if (qss) for (int i = 0; i < m->nq; i++) qss[s * m->nq + i] = d->qpos[i];
    if (qdss) for (int i = 0; i < m->nv; i++) qdss[s * m->nv + i] = d->qvel[i];
  }
  free(d);
  return 0;
}

/* oracle_rollout with the rounding-level jitter of oracle_rollout_trace applied to EVERY rollout (rollout n draws from
 * an LCG seeded with noise_seed + n): one member of the "jitter ensemble" the distribution-level parity gates are
 * calibrated on -- how far do the aggregates of reverse_once move when nothing but <= noise_mag ulp of the state
 * changes before each step?  (tests/conftest.py: jitter_envelope) */
int oracle_rollout_jitter(const dial_model* m, const dial_task* t, const real* state, const real* us, int B, int T,
                          real* rewss, real* qss, real* qdss, real* xposs, unsigned long long noise_seed,
                          double noise_mag) {
  int nq = m->nq, nv = m->nv, nu = m->nu, nx = (m->nbody - 1) * 3;
  const double sc = noise_mag * 5.9604644775390625e-08;
#pragma omp parallel
  {
    odata* d = (odata*)calloc(1, sizeof(odata));
    real info[DIAL_INFO_N];
#pragma omp for schedule(dynamic, 4)
    for (int n = 0; n < B; n++) {
      load_state(m, state, d, info);
      unsigned long long lcg = (noise_seed + (unsigned long long)n) * 6364136223846793005ULL + 1442695040888963407ULL;
      for (int s = 0; s < T; s++) {
#define ORACLE_JITTER(x) do { lcg = lcg * 6364136223846793005ULL + 1442695040888963407ULL; \
        double u_ = (double)(lcg >> 11) * (1.0 / 9007199254740992.0) * 2.0 - 1.0; (x) = (real)((double)(x) * (1.0 + sc * u_)); } while (0)
        for (int i = 0; i < nq; i++) ORACLE_JITTER(d->qpos[i]);
        for (int i = 0; i < nv; i++) { ORACLE_JITTER(d->qvel[i]); ORACLE_JITTER(d->qacc_warmstart[i]); }
#undef ORACLE_JITTER
        real rew = env_step(m, t, d, info, us + ((size_t)n * T + s) * nu);
        size_t o = (size_t)n * T + s;
        rewss[o] = rew;
        if (qss) for (int i = 0; i < nq; i++) qss[o * nq + i] = d->qpos[i];
        if (qdss) for (int i = 0; i < nv; i++) qdss[o * nv + i] = d->qvel[i];
        if (xposs) for (int b = 1; b < m->nbody; b++) for (int k = 0; k < 3; k++) xposs[o * nx + (b - 1) * 3 + k] = d->xpos[b][k];
      }
    }
    free(d);
  }
  return 0;
}

/* reverse_once (dial_core.py:103-145).  eps [N,Hn+1,nu]; noise_scale [ns] (ns = Hn+1 or 1).
 * Outputs: Ybar_out [Hn+1,nu], rews [N+1], qbar [T,nq], qdbar [T,nv], xbar [T,nx], and the optional
 * full intermediates us_out [N+1,T,nu], rewss_out [N+1,T], weights_out [N+1] for stage-wise parity tests. */
int oracle_reverse_once(const dial_model* m, const dial_task* t, const dial_cfg* cfg, const real* state,
                        const real* Ybar_in, const real* noise_scale, int ns, const real* eps, real* Ybar_out,
                        real* rews, real* qbar, real* qdbar, real* xbar, real* us_out, real* rewss_out,
                        real* weights_out) {
  int N = cfg->Nsample, Hn1 = cfg->Hnode + 1, T = cfg->Hsample + 1, B = N + 1;
  int nq = m->nq, nv = m->nv, nu = m->nu, nx = (m->nbody - 1) * 3;
  real* Y0s = (real*)malloc(sizeof(real) * (size_t)B * Hn1 * nu);
  real* us = (real*)malloc(sizeof(real) * (size_t)B * T * nu);
  real* rewss = (real*)malloc(sizeof(real) * (size_t)B * T);
  real* qss = (real*)malloc(sizeof(real) * (size_t)B * T * nq);
  real* qdss = (real*)malloc(sizeof(real) * (size_t)B * T * nv);
  real* xss = (real*)malloc(sizeof(real) * (size_t)B * T * nx);
  real* w = (real*)malloc(sizeof(real) * (size_t)B);
#pragma omp parallel for schedule(static)
  for (int n = 0; n < B; n++)
    for (int k = 0; k < Hn1; k++)
      for (int a = 0; a < nu; a++) {
        real v;
        if (n < N) {
          real sc = noise_scale[ns == 1 ? 0 : k];
          v = eps[((size_t)n * Hn1 + k) * nu + a] * sc + Ybar_in[k * nu + a]; /* :110 */
          if (k == 0) v = Ybar_in[a];                                        /* :112 */
        } else {
          v = Ybar_in[k * nu + a];                                           /* :114 */
        }
        Y0s[((size_t)n * Hn1 + k) * nu + a] = r_clip(v, -1, 1);               /* :115 */
      }
#pragma omp parallel for schedule(static)
  for (int n = 0; n < B; n++) /* node2u (:117): us = W @ Y0s */
    for (int s = 0; s < T; s++)
      for (int a = 0; a < nu; a++) {
        real acc = 0;
        for (int k = 0; k < Hn1; k++) acc += (real)cfg->W[s][k] * Y0s[((size_t)n * Hn1 + k) * nu + a];
        us[((size_t)n * T + s) * nu + a] = acc;
      }
  oracle_rollout(m, t, state, us, B, T, rewss, qss, qdss, xss);
  real rew_bar = 0; /* :121 */
  for (int s = 0; s < T; s++) rew_bar += rewss[(size_t)(B - 1) * T + s];
  rew_bar /= T;
  real mean = 0;
  for (int n = 0; n < B; n++) { /* :125 */
    real s_ = 0;
    for (int s = 0; s < T; s++) s_ += rewss[(size_t)n * T + s];
    rews[n] = s_ / T;
    mean += rews[n];
  }
  mean /= B;
  real var = 0;
  for (int n = 0; n < B; n++) var += (rews[n] - mean) * (rews[n] - mean);
  real std = r_sqrt(var / B); /* ddof = 0 */
  real mx = -(real)INFINITY;
  for (int n = 0; n < B; n++) { /* :126 */
    w[n] = (rews[n] - rew_bar) / std / (real)cfg->temp_sample;
    if (w[n] > mx) mx = w[n];
  }
  real den = 0;
  for (int n = 0; n < B; n++) { w[n] = (real)exp((double)(w[n] - mx)); den += w[n]; } /* :128 softmax */
  for (int n = 0; n < B; n++) w[n] /= den;
  /* weighted means (:132-135): every output entry sums over the samples in index order (one thread per entry: the
   * OpenMP split does not touch the summation order) */
#pragma omp parallel for schedule(static)
  for (int i = 0; i < Hn1 * nu; i++) { real s_ = 0; for (int n = 0; n < B; n++) s_ += w[n] * Y0s[(size_t)n * Hn1 * nu + i]; Ybar_out[i] = s_; }
  if (qbar) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < T * nq; i++) { real s_ = 0; for (int n = 0; n < B; n++) s_ += w[n] * qss[(size_t)n * T * nq + i]; qbar[i] = s_; }
  }
  if (qdbar) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < T * nv; i++) { real s_ = 0; for (int n = 0; n < B; n++) s_ += w[n] * qdss[(size_t)n * T * nv + i]; qdbar[i] = s_; }
  }
  if (xbar) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < T * nx; i++) { real s_ = 0; for (int n = 0; n < B; n++) s_ += w[n] * xss[(size_t)n * T * nx + i]; xbar[i] = s_; }
  }
  if (us_out) memcpy(us_out, us, sizeof(real) * (size_t)B * T * nu);
  if (rewss_out) memcpy(rewss_out, rewss, sizeof(real) * (size_t)B * T);
  if (weights_out) memcpy(weights_out, w, sizeof(real) * (size_t)B);
  free(Y0s); free(us); free(rewss); free(qss); free(qdss); free(xss); free(w);
  return 0;
}

/* shift (dial_core.py:160-166): u = W Y; roll(-1); u[-1] = 0; Y = V u */
int oracle_shift(const dial_model* m, const dial_cfg* cfg, real* Y) {
  int Hn1 = cfg->Hnode + 1, T = cfg->Hsample + 1, nu = m->nu;
  real u[DIAL_MAX_T][NU], u2[DIAL_MAX_T][NU];
  for (int s = 0; s < T; s++)
    for (int a = 0; a < nu; a++) {
      real acc = 0;
      for (int k = 0; k < Hn1; k++) acc += (real)cfg->W[s][k] * Y[k * nu + a];
      u[s][a] = acc;
    }
  for (int s = 0; s < T; s++) for (int a = 0; a < nu; a++) u2[s][a] = s + 1 < T ? u[s + 1][a] : 0;
  for (int k = 0; k < Hn1; k++)
    for (int a = 0; a < nu; a++) {
      real acc = 0;
      for (int s = 0; s < T; s++) acc += (real)cfg->V[k][s] * u2[s][a];
      Y[k * nu + a] = acc;
    }
  return 0;
}

/* Debug/diagnostic dump of one forward pass for unit tests: runs mjx.forward on (qpos, qvel, ctrl, warm)
 * and returns intermediates.  Any pointer may be NULL. */
int oracle_forward_dump(const dial_model* m, const real* qpos, const real* qvel, const real* ctrl, const real* warm,
                        real* qM, real* qfrc_bias, real* qacc_smooth, real* qacc, real* efc_force, real* con_dist,
                        real* con_pos, real* subtree_com, real* site_xpos, real* cvel, real* xpos, real* xquat,
                        real* efc_J, real* efc_aref, real* efc_D, int* niter) {
  odata* d = (odata*)calloc(1, sizeof(odata));
  for (int i = 0; i < m->nq; i++) d->qpos[i] = qpos[i];
  for (int i = 0; i < m->nv; i++) { d->qvel[i] = qvel[i]; d->qacc_warmstart[i] = warm ? warm[i] : 0; }
  for (int a = 0; a < m->nu; a++) d->ctrl[a] = ctrl ? ctrl[a] : 0;
  forward(m, d);
  int nv = m->nv;
  if (qM) for (int i = 0; i < nv; i++) for (int j = 0; j < nv; j++) qM[i * nv + j] = d->qM[i][j];
  if (qfrc_bias) for (int i = 0; i < nv; i++) qfrc_bias[i] = d->qfrc_bias[i];
  if (qacc_smooth) for (int i = 0; i < nv; i++) qacc_smooth[i] = d->qacc_smooth[i];
  if (qacc) for (int i = 0; i < nv; i++) qacc[i] = d->qacc[i];
  if (efc_force) for (int r = 0; r < m->nefc; r++) efc_force[r] = d->efc_force[r];
  if (con_dist) for (int c = 0; c < m->ncon; c++) con_dist[c] = d->con_dist[c];
  if (con_pos) for (int c = 0; c < m->ncon; c++) for (int k = 0; k < 3; k++) con_pos[c * 3 + k] = d->con_pos[c][k];
  if (subtree_com) for (int b = 0; b < m->nbody; b++) for (int k = 0; k < 3; k++) subtree_com[b * 3 + k] = d->subtree_com[b][k];
  if (site_xpos) for (int s = 0; s < m->nsite; s++) for (int k = 0; k < 3; k++) site_xpos[s * 3 + k] = d->site_xpos[s][k];
  if (cvel) for (int b = 0; b < m->nbody; b++) for (int k = 0; k < 6; k++) cvel[b * 6 + k] = d->cvel[b][k];
  if (xpos) for (int b = 0; b < m->nbody; b++) for (int k = 0; k < 3; k++) xpos[b * 3 + k] = d->xpos[b][k];
  if (xquat) for (int b = 0; b < m->nbody; b++) for (int k = 0; k < 4; k++) xquat[b * 4 + k] = d->xquat[b][k];
  if (efc_J) for (int r = 0; r < m->nefc; r++) for (int i = 0; i < nv; i++) efc_J[r * nv + i] = d->efc_J[r][i];
  if (efc_aref) for (int r = 0; r < m->nefc; r++) efc_aref[r] = d->efc_aref[r];
  if (efc_D) for (int r = 0; r < m->nefc; r++) efc_D[r] = d->efc_D[r];
  if (niter) *niter = d->solver_niter;
  free(d);
  return 0;
}

/* box narrow-phase hook (tests/test_box_collisions.py): geoms as (pos[3], mat[9] row-major, size[3]) */
int oracle_box_contact(int kind, int sub, const real* g1, const real* g2, real* dist, real* pos, real* frame) {
  obox b1, b2;
  for (int k = 0; k < 3; k++) { b1.c[k] = g1[k]; b1.h[k] = g1[12 + k]; b2.c[k] = g2[k]; b2.h[k] = g2[12 + k]; }
  for (int k = 0; k < 9; k++) { b1.R[k] = g1[3 + k]; b2.R[k] = g2[3 + k]; }
  real ax1[3] = {g1[5], g1[8], g1[11]};
  if (kind == DIAL_CON_PLANE_BOX) plane_box(ax1, g1, &b2, sub, dist, pos, frame);
  else if (kind == DIAL_CON_SPHERE_BOX) sphere_box(g1, g1[12], &b2, dist, pos, frame);
  else if (kind == DIAL_CON_CAPSULE_BOX) capsule_box(g1, ax1, g1[13], g1[12], &b2, sub, dist, pos, frame);
  else if (kind == DIAL_CON_BOX_BOX) box_box(&b1, &b2, sub, dist, pos, frame);
  else return -1;
  return 0;
}

/* get_foot_step KAT hook */
int oracle_foot_step(const dial_task* t, real time, real* h) { get_foot_step(t, time, h); return 0; }

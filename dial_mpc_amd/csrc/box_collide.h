// box_collide.h -- narrow phases of box geoms (the crate scenes: include/dial_mpc.h DIAL_CON_*_BOX).
//
// MJX sends boxes through collision_convex.py -- third-party code that is not part of the reference checkout and whose
// manifold selection changed between releases.  What is implemented here is the GEOMETRY of each pair (unique wherever
// the contact is unique) with a fixed number of candidate contacts per pair, in MJX's conventions: the normal points
// from geom1 into geom2, dist < 0 is penetration, pos lies midway between the two surfaces.  One lane evaluates one
// candidate; the candidates of one pair recompute the pair (a box pair is <= 4 lanes of the contact phase).
// Only the generic instantiation carries this code (rollout_body.h: `if constexpr (!is_static)`).
#pragma once

// distance beyond which the second broad phases park a box candidate (a lower bound of the true distance is enough then: no
// rows, nothing reads the position); dial_create rejects box contacts whose margin reaches it
#ifndef DIAL_BOX_PARK_DIST
#define DIAL_BOX_PARK_DIST 0.01f
#endif

namespace dial {

struct BoxG {
  float c[3], q[4], h[3];   // centre, orientation (world-from-box quaternion), half sizes
};

// NO dynamically indexed private arrays in this file: an array indexed by a run-time value cannot live in registers, the
// compiler puts it in scratch memory, and round 3's generic kernel carried 544 B of it per lane -- 865 MB of HBM traffic
// per launch for 19.6 MB of algorithmic bytes (profiles/r03_pmc_unitree_go2_crate_climb.json).  Run-time choices among three
// values are select chains (pick3), sorted lists are sorting networks on named registers, and the one structure that
// really is a list of run-time length -- the clipping polygons of box_box -- lives in a lane-private slice of LDS.
#ifdef DIAL_EMU
#define DIAL_UNROLL_FULL
#else
#define DIAL_UNROLL_FULL _Pragma("unroll")
#endif
DIAL_DEV float pick3(float a0, float a1, float a2, int k) { return k == 0 ? a0 : (k == 1 ? a1 : a2); }
DIAL_DEV float pick3v(const float* a, int k) { return pick3(a[0], a[1], a[2], k); }                  // a[k], a = 3 registers
DIAL_DEV void pick_row(float* o, const float (*a)[3], int k) {                                       // o = a[k][:]
  for (int c = 0; c < 3; c++) o[c] = pick3(a[0][c], a[1][c], a[2][c], k);
}
DIAL_DEV void cswap(float& a, float& b) { const float lo = a < b ? a : b, hi = a < b ? b : a; a = lo; b = hi; }
// words of LDS one box-box candidate needs for its two clipping polygons (<= 10 vertices x 3 each)
#define DIAL_BOX_POLY_WORDS 60

DIAL_DEV void box_axes(const BoxG& b, float ax[3][3]) {   // ax[k] = world direction of the box's k-th axis
  float mat[9];
  dm::quat_to_mat(mat, b.q);
  for (int k = 0; k < 3; k++) { ax[k][0] = mat[k]; ax[k][1] = mat[3 + k]; ax[k][2] = mat[6 + k]; }
}

// sphere (centre sc, radius r) against a box
DIAL_DEV void sphere_box(const float* sc, float r, const BoxG& b, float& dist, float* pos, float* fr) {
  const float rel[3] = {sc[0] - b.c[0], sc[1] - b.c[1], sc[2] - b.c[2]};
  float p[3], nl[3];
  dm::inv_rotate(p, rel, b.q);
  float len2 = 0.f, slack = 0.f;
  int kn = 0;
  for (int k = 0; k < 3; k++) {
    const float over = dm::absf(p[k]) - b.h[k];          // > 0: outside the slab of axis k
    const float o = over > 0.f ? over : 0.f;
    nl[k] = p[k] > 0.f ? -o : o;                         // from the centre towards the box
    len2 += o * o;
    if (k == 0 || over > slack) { slack = over; kn = k; }   // inside: the face that is nearest (largest negative `over`)
  }
  if (len2 > 0.f) {
    const float len = DM_SQRT(len2);
    for (int k = 0; k < 3; k++) nl[k] /= len;
    dist = len - r;
  } else {
    const float sgn = pick3v(p, kn) >= 0.f ? -1.f : 1.f;
    for (int k = 0; k < 3; k++) nl[k] = k == kn ? sgn : 0.f;
    dist = slack - r;                                    // slack = -(depth below the nearest face)
  }
  float n[3];
  dm::rotate(n, nl, b.q);
  for (int k = 0; k < 3; k++) pos[k] = sc[k] + n[k] * (r + dist * 0.5f);
  make_frame(fr, n);
}

// plane (normal n through ppos) against a box: the sub-th lowest vertex.  With e_k = h_k |n . axis_k| the vertex heights
// are base -+ e_0 -+ e_1 -+ e_2; sorted: all minus, then the smallest e flipped, ... -- ranked by counting, ties by index
DIAL_DEV void plane_box(const float* n, const float* ppos, const BoxG& b, int sub, float& dist, float* pos, float* fr) {
  float ax[3][3], e[3];
  box_axes(b, ax);
  const float rel[3] = {b.c[0] - ppos[0], b.c[1] - ppos[1], b.c[2] - ppos[2]};
  const float base = dm::dot3(rel, n);
  for (int k = 0; k < 3; k++) e[k] = b.h[k] * dm::dot3(n, ax[k]);
  {   // broad phase: the LOWEST vertex more than 1 cm above the plane -- all four candidates parked (dist >= margin: no rows; nothing
      // reads the position of a box-plane contact that does not touch).  The trunk over the floor: every step of a standing robot.
    const float lowest = base - (dm::absf(e[0]) + dm::absf(e[1]) + dm::absf(e[2]));
    if (lowest > DIAL_BOX_PARK_DIST) {
      dist = lowest;
      for (int k = 0; k < 3; k++) pos[k] = b.c[k];
      make_frame(fr, n);
      return;
    }
  }
  float hv[8];
  DIAL_UNROLL_FULL
  for (int i = 0; i < 8; i++) hv[i] = base + ((i & 1) ? e[0] : -e[0]) + ((i & 2) ? e[1] : -e[1]) + ((i & 4) ? e[2] : -e[2]);
  int pick = 0;
  DIAL_UNROLL_FULL
  for (int i = 0; i < 8; i++) {
    int rank = 0;
    DIAL_UNROLL_FULL
    for (int j = 0; j < 8; j++) rank += (hv[j] < hv[i] || (hv[j] == hv[i] && j < i)) ? 1 : 0;
    pick = rank == sub ? i : pick;
  }
  float v[3];
  for (int k = 0; k < 3; k++)
    v[k] = b.c[k] + ((pick & 1) ? b.h[0] : -b.h[0]) * ax[0][k] + ((pick & 2) ? b.h[1] : -b.h[1]) * ax[1][k] + ((pick & 4) ? b.h[2] : -b.h[2]) * ax[2][k];
  const float d[3] = {v[0] - ppos[0], v[1] - ppos[1], v[2] - ppos[2]};
  dist = dm::dot3(d, n);
  for (int k = 0; k < 3; k++) pos[k] = v[k] - n[k] * (dist * 0.5f);
  make_frame(fr, n);
}

// parameter t in [0, 1] of the segment point l0 + t (l1 - l0) (box frame) closest to the box: the squared distance is convex
// and piecewise quadratic in t; its pieces end where the point crosses one of the six slab planes.  A later piece replaces
// an earlier one only if it is better by more than rounding (a capsule parallel to a face keeps its first end)
DIAL_DEV float segment_box_t(const float* l0, const float* l1, const float* h) {
  // the cuts 0 < t < 1 where the point crosses a slab plane, sorted: eight NAMED values (0, up to six crossings, 1; unused
  // entries are parked at 2 so that they sort to the end) through a 19-exchange sorting network -- the insertion sort over a
  // dynamically indexed array this replaces lived in scratch memory.  Sorted values are what the pieces below depend on, so
  // the result is the insertion sort's.
  float c0 = 0.f, c1 = 2.f, c2 = 2.f, c3 = 2.f, c4 = 2.f, c5 = 2.f, c6 = 2.f, c7 = 1.f;
  int nc = 2;
  {
    float* const slot[3][2] = {{&c1, &c2}, {&c3, &c4}, {&c5, &c6}};
    DIAL_UNROLL_FULL
    for (int k = 0; k < 3; k++) {
      const float dk = l1[k] - l0[k];
      if (dk == 0.f) continue;
      const float ta = (h[k] - l0[k]) / dk, tb = (-h[k] - l0[k]) / dk;
      const float tlo = ta < tb ? ta : tb, thi = ta < tb ? tb : ta;
      if (tlo > 0.f && tlo < 1.f) { *slot[k][0] = tlo; nc++; }
      if (thi > 0.f && thi < 1.f) { *slot[k][1] = thi; nc++; }
    }
  }
  // optimal 8-input network (19 compare-exchanges)
  cswap(c0, c1); cswap(c2, c3); cswap(c4, c5); cswap(c6, c7);
  cswap(c0, c2); cswap(c1, c3); cswap(c4, c6); cswap(c5, c7);
  cswap(c1, c2); cswap(c5, c6); cswap(c0, c4); cswap(c3, c7);
  cswap(c1, c5); cswap(c2, c6);
  cswap(c1, c4); cswap(c3, c6);
  cswap(c2, c4); cswap(c3, c5);
  cswap(c3, c4);
  const float cut[8] = {c0, c1, c2, c3, c4, c5, c6, c7};   // statically indexed below
  float best_t = 0.f, best_f = -1.f;
  DIAL_UNROLL_FULL
  for (int i = 0; i + 1 < 8; i++) {
    if (i + 1 >= nc) continue;           // (pieces beyond the last real cut: the parked entries)
    const float t0 = cut[i], t1 = cut[i + 1], tm = 0.5f * (t0 + t1);
    float A = 0.f, B = 0.f, C = 0.f;    // f(t) = A t^2 + 2 B t + C over this piece
    for (int k = 0; k < 3; k++) {
      const float dk = l1[k] - l0[k], x = l0[k] + tm * dk;
      const float wall = x > h[k] ? h[k] : (x < -h[k] ? -h[k] : x);
      if (wall != x) { const float o = l0[k] - wall; A += dk * dk; B += dk * o; C += o * o; }
    }
    const float t = A > 0.f ? dm::clip(-B / A, t0, t1) : t0;
    const float fr = (A * t + 2.f * B) * t + C, f = fr > 0.f ? fr : 0.f;   // (cancellation can leave a tiny negative value)
    if (best_f < 0.f || f < best_f * (1.f - 1e-6f) - 1e-12f) { best_f = f; best_t = t; }
  }
  return best_t;
}

// capsule against a box: sub 0 = sphere at the segment point closest to the box, sub 1 = sphere at the end farther from it.
// When the capsule's AXIS enters the box (penetration deeper than the radius) "closest" degenerates to a stretch of distance
// 0: the contact is then where the axis crosses the surface, with the normal of the face it crosses and dist = -radius --
// the continuation of the shallow case (closest point -> surface point, same face normal).
DIAL_DEV void capsule_box(const float* ctr, const float* axis, float hl, float r, const BoxG& b, int sub, float& dist, float* pos, float* fr) {
  {   // broad phase, as in box_box: bounding spheres more than 1 cm apart -- both candidates parked (dist >= margin: no rows, and
      // nothing reads the position of a capsule contact that does not touch); the normal is the centre line.  All the capsule
      // lanes of a wavefront take this exit while the robot is away from the box, and the ~600-instruction routine is skipped.
    const float tw[3] = {b.c[0] - ctr[0], b.c[1] - ctr[1], b.c[2] - ctr[2]};
    const float gap = DM_SQRT(dm::dot3(tw, tw)) - (hl + r) - DM_SQRT(dm::dot3(b.h, b.h));
    if (gap > DIAL_BOX_PARK_DIST) {
      dist = sub == 0 ? gap : 1.f;
      for (int k = 0; k < 3; k++) pos[k] = 0.5f * (ctr[k] + b.c[k]);
      make_frame(fr, tw);
      return;
    }
  }
  float e0[3], e1[3], r0[3], r1[3], l0[3], l1[3];
  for (int k = 0; k < 3; k++) { e0[k] = ctr[k] - axis[k] * hl; e1[k] = ctr[k] + axis[k] * hl; r0[k] = e0[k] - b.c[k]; r1[k] = e1[k] - b.c[k]; }
  dm::inv_rotate(l0, r0, b.q);
  dm::inv_rotate(l1, r1, b.q);
  {   // second broad phase, in the box's frame: the capsule more than 1 cm outside one of the box's three slabs (a separating
      // FACE axis: the true distance is at least that gap).  The bounding spheres above overlap whenever the robot stands next to
      // the box; the slabs separate all but the capsules that are about to touch.
    float sep = -1.f;
    for (int k = 0; k < 3; k++) {
      const float lo = dm::fminf_(l0[k], l1[k]) - r, hi = dm::fmaxf_(l0[k], l1[k]) + r;
      sep = dm::fmaxf_(sep, dm::fmaxf_(lo - b.h[k], -b.h[k] - hi));
    }
    if (sep > DIAL_BOX_PARK_DIST) {
      dist = sub == 0 ? sep : 1.f;
      for (int k = 0; k < 3; k++) pos[k] = 0.5f * (ctr[k] + b.c[k]);
      const float tw[3] = {b.c[0] - ctr[0], b.c[1] - ctr[1], b.c[2] - ctr[2]};
      make_frame(fr, tw);
      return;
    }
  }
  // interior stretch [ta, tb] of the axis (intersection of the three slabs) and the slabs that bound it
  float ta = 0.f, tb = 1.f;
  int ka = -1, kb = -1;
  bool hit = true;
  for (int k = 0; k < 3; k++) {
    const float dk = l1[k] - l0[k];
    if (dk == 0.f) { hit = hit && !(dm::absf(l0[k]) > b.h[k]); continue; }
    const float t1 = (-b.h[k] - l0[k]) / dk, t2 = (b.h[k] - l0[k]) / dk;
    const float lo = dm::fminf_(t1, t2), hi = dm::fmaxf_(t1, t2);
    if (lo > ta) { ta = lo; ka = k; }
    if (hi < tb) { tb = hi; kb = k; }
  }
  const bool through = hit && ta <= tb;
  const int kface = through ? (ka >= 0 ? ka : kb) : -1;          // enters through face ka, else (first end inside) leaves through kb
  const float t = through ? (ka >= 0 ? ta : (kb >= 0 ? tb : 0.f)) : segment_box_t(l0, l1, b.h);
  if (sub == 0 && kface >= 0) {
    float nl[3], n[3];
    const float l0f = pick3v(l0, kface), l1f = pick3v(l1, kface);
    const float sgn = l0f + t * (l1f - l0f) >= 0.f ? -1.f : 1.f;
    for (int k = 0; k < 3; k++) nl[k] = k == kface ? sgn : 0.f;
    dm::rotate(n, nl, b.q);
    dist = -r;
    for (int k = 0; k < 3; k++) pos[k] = e0[k] + t * (e1[k] - e0[k]) + n[k] * (r + dist * 0.5f);
    make_frame(fr, n);
    return;
  }
  float sc[3];
  for (int k = 0; k < 3; k++) sc[k] = sub == 0 ? e0[k] + t * (e1[k] - e0[k]) : (t <= 0.5f ? e1[k] : e0[k]);
  sphere_box(sc, r, b, dist, pos, fr);
}

// box against box (separating-axis test; see the description at DIAL_CON_BOX_BOX and oracle-independent notes in DESIGN.md).
// Everything is done in A's frame: C = RA^T RB, t = RA^T (cB - cA).
// `poly`: DIAL_BOX_POLY_WORDS floats of LDS private to the calling lane (the two clipping polygons of a face contact).
DIAL_DEV void box_box(const BoxG& A, const BoxG& B, int sub, float& dist, float* pos, float* fr, float* poly) {
  const float tw[3] = {B.c[0] - A.c[0], B.c[1] - A.c[1], B.c[2] - A.c[2]};
  {   // bounding spheres more than 1 cm apart: nothing to do (all candidates parked; the normal is the centre line)
    const float gap = DM_SQRT(dm::dot3(tw, tw)) - DM_SQRT(dm::dot3(A.h, A.h)) - DM_SQRT(dm::dot3(B.h, B.h));
    if (gap > DIAL_BOX_PARK_DIST) {
      dist = sub == 0 ? gap : 1.f;
      for (int k = 0; k < 3; k++) pos[k] = 0.5f * (A.c[k] + B.c[k]);
      make_frame(fr, tw);
      return;
    }
  }
  float axA[3][3], axB[3][3], C[3][3], Q[3][3], t[3];
  box_axes(A, axA);
  box_axes(B, axB);
  DIAL_UNROLL_FULL
  for (int i = 0; i < 3; i++) {
    t[i] = dm::dot3(tw, axA[i]);
    DIAL_UNROLL_FULL
    for (int j = 0; j < 3; j++) { C[i][j] = dm::dot3(axA[i], axB[j]); Q[i][j] = dm::absf(C[i][j]); }
  }
  int best = -1;
  float sbest = 0.f;
  DIAL_UNROLL_FULL
  for (int i = 0; i < 3; i++) {          // faces of A
    const float sep = dm::absf(t[i]) - (A.h[i] + B.h[0] * Q[i][0] + B.h[1] * Q[i][1] + B.h[2] * Q[i][2]);
    if (best < 0 || sep > sbest) { best = i; sbest = sep; }
  }
  DIAL_UNROLL_FULL
  for (int j = 0; j < 3; j++) {          // faces of B
    const float tb = t[0] * C[0][j] + t[1] * C[1][j] + t[2] * C[2][j];
    const float sep = dm::absf(tb) - (B.h[j] + A.h[0] * Q[0][j] + A.h[1] * Q[1][j] + A.h[2] * Q[2][j]);
    if (sep > sbest) { best = 3 + j; sbest = sep; }
  }
  // second broad phase: a FACE axis separates the boxes by more than 1 cm (the true distance is at least that): all candidates
  // parked before the nine edge axes and the clipping.  The bounding spheres above overlap whenever the robot is near the crate
  // (its half-diagonal is 0.63 m); the trunk over the crate's top face, or the torso beside it, is separated on a face axis.
  if (sbest > DIAL_BOX_PARK_DIST) {
    dist = sub == 0 ? sbest : 1.f;
    for (int k = 0; k < 3; k++) pos[k] = 0.5f * (A.c[k] + B.c[k]);
    make_frame(fr, tw);
    return;
  }
  float nedge[3] = {0.f, 0.f, 0.f};       // world direction of the winning edge axis
  DIAL_UNROLL_FULL
  for (int i = 0; i < 3; i++) {
    DIAL_UNROLL_FULL
    for (int j = 0; j < 3; j++) {
      float L[3];
      dm::cross3(L, axA[i], axB[j]);
      const float len = DM_SQRT(dm::dot3(L, L));
      if (len < 1e-4f) continue;
      for (int k = 0; k < 3; k++) L[k] /= len;
      float ra = 0.f, rb = 0.f;
      for (int k = 0; k < 3; k++) { ra += A.h[k] * dm::absf(dm::dot3(L, axA[k])); rb += B.h[k] * dm::absf(dm::dot3(L, axB[k])); }
      const float sep = dm::absf(dm::dot3(tw, L)) - (ra + rb);
      if (sep > sbest + 0.05f * dm::absf(sbest) + 1e-5f) { best = 6 + 3 * i + j; sbest = sep; for (int k = 0; k < 3; k++) nedge[k] = L[k]; }
    }
  }
  const float mid[3] = {0.5f * (A.c[0] + B.c[0]), 0.5f * (A.c[1] + B.c[1]), 0.5f * (A.c[2] + B.c[2])};
  float n[3];
  if (best < 3) {
    float ab[3];
    pick_row(ab, axA, best);
    const float sg = pick3v(t, best) >= 0.f ? 1.f : -1.f;
    for (int k = 0; k < 3; k++) n[k] = ab[k] * sg;
  } else if (best < 6) {
    float bb[3];
    pick_row(bb, axB, best - 3);
    const float sg = dm::dot3(tw, bb) >= 0.f ? 1.f : -1.f;
    for (int k = 0; k < 3; k++) n[k] = bb[k] * sg;
  } else { const float sg = dm::dot3(tw, nedge) >= 0.f ? 1.f : -1.f; for (int k = 0; k < 3; k++) n[k] = nedge[k] * sg; }
  make_frame(fr, n);
  if (sbest > DIAL_BOX_PARK_DIST) {                    // clearly apart
    dist = sub == 0 ? sbest : 1.f;
    for (int k = 0; k < 3; k++) pos[k] = mid[k];
    return;
  }
  if (best >= 6) {                        // edge - edge: closest points of the two supporting edges
    const int i = (best - 6) / 3, j = (best - 6) - 3 * i;
    float pa[3] = {A.c[0], A.c[1], A.c[2]}, pb[3] = {B.c[0], B.c[1], B.c[2]};
    DIAL_UNROLL_FULL
    for (int k = 0; k < 3; k++) {
      if (k != i) { const float sk = dm::dot3(n, axA[k]) >= 0.f ? A.h[k] : -A.h[k]; for (int q = 0; q < 3; q++) pa[q] += sk * axA[k][q]; }
      if (k != j) { const float sk = dm::dot3(n, axB[k]) >= 0.f ? B.h[k] : -B.h[k]; for (int q = 0; q < 3; q++) pb[q] -= sk * axB[k][q]; }
    }
    float ai[3], bj[3];
    pick_row(ai, axA, i);
    pick_row(bj, axB, j);
    const float ahi = pick3v(A.h, i), bhj = pick3v(B.h, j);
    const float wv[3] = {pb[0] - pa[0], pb[1] - pa[1], pb[2] - pa[2]};
    const float uu = dm::dot3(ai, bj), q1 = dm::dot3(ai, wv), q2 = -dm::dot3(bj, wv), den = 1.f - uu * uu;
    const float al = dm::clip((q1 + uu * q2) / den, -ahi, ahi), be = dm::clip((uu * q1 + q2) / den, -bhj, bhj);
    float gap = 0.f;
    for (int k = 0; k < 3; k++) {
      const float ca = pa[k] + al * ai[k], cb = pb[k] + be * bj[k];
      gap += (cb - ca) * n[k];
      pos[k] = 0.5f * (ca + cb);
    }
    dist = sub == 0 ? gap : 1.f;
    return;
  }
  // face contact: reference box X (face axis kx, outward normal nref towards Y), incident box Y -- copies selected value by
  // value (a pointer to one of two private arrays would make both of them memory objects)
  const bool a_ref = best < 3;
  float aX[3][3], aY[3][3], Xh[3], Yh[3], Xc[3], Yc[3];
  DIAL_UNROLL_FULL
  for (int r = 0; r < 3; r++) {
    Xh[r] = a_ref ? A.h[r] : B.h[r]; Yh[r] = a_ref ? B.h[r] : A.h[r];
    Xc[r] = a_ref ? A.c[r] : B.c[r]; Yc[r] = a_ref ? B.c[r] : A.c[r];
    DIAL_UNROLL_FULL
    for (int c = 0; c < 3; c++) { aX[r][c] = a_ref ? axA[r][c] : axB[r][c]; aY[r][c] = a_ref ? axB[r][c] : axA[r][c]; }
  }
  const int kx = a_ref ? best : best - 3, ux = (kx + 1) % 3, vx = (kx + 2) % 3;
  float nref[3];
  for (int k = 0; k < 3; k++) nref[k] = a_ref ? n[k] : -n[k];
  int my = 0;
  float amax = -1.f;
  DIAL_UNROLL_FULL
  for (int k = 0; k < 3; k++) { const float a = dm::absf(dm::dot3(nref, aY[k])); if (a > amax) { amax = a; my = k; } }
  const int uy = (my + 1) % 3, vy = (my + 2) % 3;
  float aXu[3], aXv[3], aYm[3], aYu[3], aYv[3];
  pick_row(aXu, aX, ux); pick_row(aXv, aX, vx);
  pick_row(aYm, aY, my); pick_row(aYu, aY, uy); pick_row(aYv, aY, vy);
  const float Xhk = pick3v(Xh, kx), Xhu = pick3v(Xh, ux), Xhv = pick3v(Xh, vx);
  const float Yhm = pick3v(Yh, my), Yhu = pick3v(Yh, uy), Yhv = pick3v(Yh, vy);
  const float sgy = dm::dot3(nref, aYm) >= 0.f ? -Yhm : Yhm;   // the incident face looks back at X
  // incident face in reference-face coordinates (a, b) and signed distance d to the reference face
  float fc[3], cu[3], cv[3];   // face centre, half edges
  {
    float cw[3];
    for (int k = 0; k < 3; k++) cw[k] = Yc[k] + sgy * aYm[k] - Xc[k];
    fc[0] = dm::dot3(cw, aXu); fc[1] = dm::dot3(cw, aXv); fc[2] = dm::dot3(cw, nref) - Xhk;
    cu[0] = Yhu * dm::dot3(aYu, aXu); cu[1] = Yhu * dm::dot3(aYu, aXv); cu[2] = Yhu * dm::dot3(aYu, nref);
    cv[0] = Yhv * dm::dot3(aYv, aXu); cv[1] = Yhv * dm::dot3(aYv, aXv); cv[2] = Yhv * dm::dot3(aYv, nref);
  }
  // the two polygons of the clipping passes, 10 vertices x 3 each, in the lane's slice of LDS (run-time length, run-time index)
  float* const pa_ = poly;
  float* const pb_ = poly + 30;
  int np_ = 4;
  for (int k = 0; k < 3; k++) {
    pa_[0 + k] = fc[k] + cu[k] + cv[k]; pa_[3 + k] = fc[k] - cu[k] + cv[k];
    pa_[6 + k] = fc[k] - cu[k] - cv[k]; pa_[9 + k] = fc[k] + cu[k] - cv[k];
  }
  for (int pl = 0; pl < 4 && np_ > 0; pl++) {   // clip against a <= hu, -a <= hu, b <= hv, -b <= hv
    const int co = pl >> 1;
    const float sg = (pl & 1) ? -1.f : 1.f, lim = co == 0 ? Xhu : Xhv;
    const float* src = (pl & 1) ? pb_ : pa_;
    float* dst = (pl & 1) ? pa_ : pb_;
    int no = 0;
    for (int q = 0; q < np_; q++) {
      const int q2 = q + 1 < np_ ? q + 1 : 0;
      const float sq0 = src[3 * q], sq1 = src[3 * q + 1], sq2 = src[3 * q + 2];
      const float sr0 = src[3 * q2], sr1 = src[3 * q2 + 1], sr2 = src[3 * q2 + 2];
      const float fp = sg * (co == 0 ? sq0 : sq1) - lim, fq = sg * (co == 0 ? sr0 : sr1) - lim;
      if (fp <= 0.f) { dst[3 * no] = sq0; dst[3 * no + 1] = sq1; dst[3 * no + 2] = sq2; no++; }
      if ((fp <= 0.f) != (fq <= 0.f)) {
        const float wgt = fp / (fp - fq);
        dst[3 * no] = sq0 + wgt * (sr0 - sq0); dst[3 * no + 1] = sq1 + wgt * (sr1 - sq1); dst[3 * no + 2] = sq2 + wgt * (sr2 - sq2);
        no++;
      }
    }
    np_ = no;
  }
  // after the four passes the polygon is back in pa_ (each pass ping-pongs)
  int pick = -1;
  for (int q = 0; q < np_; q++) {
    int rank = 0;
    const float zq = pa_[3 * q + 2];
    for (int o = 0; o < np_; o++) { const float zo = pa_[3 * o + 2]; rank += (zo < zq || (zo == zq && o < q)) ? 1 : 0; }
    pick = rank == sub ? q : pick;
  }
  if (pick < 0) { dist = 1.f; for (int k = 0; k < 3; k++) pos[k] = mid[k]; return; }
  const float pk0 = pa_[3 * pick], pk1 = pa_[3 * pick + 1], pk2 = pa_[3 * pick + 2];
  dist = pk2;
  for (int k = 0; k < 3; k++) pos[k] = Xc[k] + pk0 * aXu[k] + pk1 * aXv[k] + (Xhk + pk2 * 0.5f) * nref[k];
}

}  // namespace dial

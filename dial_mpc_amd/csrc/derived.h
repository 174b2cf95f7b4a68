// derived.h -- topology tables derived from dial_model on the host (dial_create) and read by the
// kernels, plus the per-wavefront LDS workspace layout.  Internal to the library (not ABI).
#pragma once
#include <stdint.h>
#include "../../include/dial_mpc.h"

#define DIAL_MAX_TRI ((DIAL_MAX_V * (DIAL_MAX_V + 1)) / 2)

struct dial_derived {
  int32_t nlevel;                          // max body depth
  int32_t lvl_start[DIAL_MAX_BODY + 1];    // bodies of depth d: lvl_body[lvl_start[d-1] .. lvl_start[d])
  int32_t lvl_body[DIAL_MAX_BODY];
  uint32_t body_ancmask[DIAL_MAX_BODY];    // bit i set: dof i moves body b (ancestor-or-own dof)
  uint32_t dof_ancmask[DIAL_MAX_V];        // bit j set: dof j is an ancestor-or-self of dof i
  int32_t dof_act[DIAL_MAX_V];             // actuator acting on dof i, or -1
  int32_t dof_limrow[DIAL_MAX_V];          // limit row of dof i, or -1
  int32_t ntri;                            // nv*(nv+1)/2
  uint16_t tri[DIAL_MAX_TRI];              // lower-triangle entries, (i << 8) | j, row-major
  int32_t ws_words;                        // LDS words per wavefront
};


// ---- per-wavefront LDS workspace (pointers into one float array) -----------------------------
struct Ws {
  float *qpos, *qvel, *warm, *info, *ctrl, *act, *Y;
  float *xpos, *xquat, *xmat, *xipos, *ximat, *xanchor, *xaxis, *gpos, *gaxis, *spos, *com;
  float *cinert, *cdof, *cvel, *cdofdot, *cacc, *crb, *cfl, *cfrc, *Fd;
  float *M, *L, *H;
  float *cdist, *cpos, *cframe, *Jc;
  float *D, *aref, *lsign, *Jaref, *JarefW, *JarefS, *jv, *frc, *quad;
  float *qfs, *qas, *qacc, *Ma, *MaW, *MaS, *grad, *search, *mv, *qfc, *rhs, *ysol;
};

#if defined(__HIPCC__)
#define WS_HD __host__ __device__ inline
#else
#define WS_HD inline
#endif

// Carve the workspace out of `base`; returns the number of words used.  Used with base = nullptr on
// the host to size the dynamic LDS allocation.
WS_HD int ws_carve(Ws& s, float* base, int nq, int nv, int nu, int nbody, int njnt, int ngeom, int nsite,
                   int ncon, int nefc, int nnode) {
  int o = 0;
#define WS_TAKE(name, n) s.name = base + o; o += (((n) + 3) & ~3);
  WS_TAKE(qpos, nq) WS_TAKE(qvel, nv) WS_TAKE(warm, nv) WS_TAKE(info, DIAL_INFO_N) WS_TAKE(ctrl, nu)
  WS_TAKE(act, nu) WS_TAKE(Y, nnode * nu)
  WS_TAKE(xpos, nbody * 3) WS_TAKE(xquat, nbody * 4) WS_TAKE(xmat, nbody * 9) WS_TAKE(xipos, nbody * 3)
  WS_TAKE(ximat, nbody * 9) WS_TAKE(xanchor, njnt * 3) WS_TAKE(xaxis, njnt * 3) WS_TAKE(gpos, ngeom * 3)
  WS_TAKE(gaxis, ngeom * 3) WS_TAKE(spos, nsite * 3) WS_TAKE(com, nbody * 3)
  WS_TAKE(cinert, nbody * 10) WS_TAKE(cdof, nv * 6) WS_TAKE(cvel, nbody * 6) WS_TAKE(cdofdot, nv * 6)
  WS_TAKE(cacc, nbody * 6) WS_TAKE(crb, nbody * 10) WS_TAKE(cfl, nbody * 6) WS_TAKE(cfrc, nbody * 6)
  WS_TAKE(Fd, nv * 6)
  WS_TAKE(M, nv * nv) WS_TAKE(L, nv * nv) WS_TAKE(H, nv * nv)
  WS_TAKE(cdist, ncon) WS_TAKE(cpos, ncon * 3) WS_TAKE(cframe, ncon * 9) WS_TAKE(Jc, ncon * 3 * nv)
  WS_TAKE(D, nefc) WS_TAKE(aref, nefc) WS_TAKE(lsign, nefc) WS_TAKE(Jaref, nefc) WS_TAKE(JarefW, nefc)
  WS_TAKE(JarefS, nefc) WS_TAKE(jv, nefc) WS_TAKE(frc, nefc) WS_TAKE(quad, nefc * 3)
  WS_TAKE(qfs, nv) WS_TAKE(qas, nv) WS_TAKE(qacc, nv) WS_TAKE(Ma, nv) WS_TAKE(MaW, nv) WS_TAKE(MaS, nv)
  WS_TAKE(grad, nv) WS_TAKE(search, nv) WS_TAKE(mv, nv) WS_TAKE(qfc, nv) WS_TAKE(rhs, nv) WS_TAKE(ysol, nv)
#undef WS_TAKE
  return o;
}

// Host: build the derived tables.  Returns 0 or a negative DIAL_ERR_* code.
static inline int dial_build_derived(const dial_model* m, dial_derived* dv) {
  if (m->nv > 32 || m->nbody > DIAL_MAX_BODY || m->nv > DIAL_MAX_V) return DIAL_ERR_ARG;
  int nlevel = 0;
  for (int b = 1; b < m->nbody; b++) nlevel = m->body_depth[b] > nlevel ? m->body_depth[b] : nlevel;
  dv->nlevel = nlevel;
  int k = 0;
  dv->lvl_start[0] = 0;
  for (int d = 1; d <= nlevel; d++) {
    for (int b = 1; b < m->nbody; b++)
      if (m->body_depth[b] == d) dv->lvl_body[k++] = b;
    dv->lvl_start[d] = k;
  }
  for (int i = 0; i < m->nv; i++) {
    uint32_t mask = 0;
    for (int j = i; j >= 0; j = m->dof_parentid[j]) mask |= (1u << j);
    dv->dof_ancmask[i] = mask;
    dv->dof_act[i] = -1;
    dv->dof_limrow[i] = -1;
  }
  for (int b = 0; b < m->nbody; b++) {
    uint32_t mask = 0;
    int bb = b;
    while (bb > 0 && m->body_dofnum[bb] == 0) bb = m->body_parent[bb];
    if (bb > 0) mask = dv->dof_ancmask[m->body_dofadr[bb] + m->body_dofnum[bb] - 1];
    dv->body_ancmask[b] = mask;
  }
  for (int a = 0; a < m->nu; a++) {
    if (dv->dof_act[m->act_dofadr[a]] != -1) return DIAL_ERR_UNSUPPORTED;  // one actuator per dof
    dv->dof_act[m->act_dofadr[a]] = a;
  }
  for (int l = 0; l < m->nlim; l++) dv->dof_limrow[m->jnt_dofadr[m->lim_jnt[l]]] = l;
  int t = 0;
  for (int i = 0; i < m->nv; i++)
    for (int j = 0; j <= i; j++) dv->tri[t++] = (uint16_t)((i << 8) | j);
  dv->ntri = t;
  Ws s;
  dv->ws_words = ws_carve(s, (float*)0, m->nq, m->nv, m->nu, m->nbody, m->njnt, m->ngeom, m->nsite, m->ncon,
                          m->nefc, DIAL_MAX_NODE);
  return DIAL_OK;
}

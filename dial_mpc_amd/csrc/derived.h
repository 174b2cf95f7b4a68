// derived.h -- host-side topology tables derived from dial_model, the conversion of the capacity-sized ABI
// structs into the dimension-specialised CModel<D>, and the per-wavefront LDS workspace layout.
// Internal to the library (not ABI).
#pragma once
#include <cmath>
#include <cstring>
#include "cmodel.h"

#define DIAL_MAX_TRI ((DIAL_MAX_V * (DIAL_MAX_V + 1)) / 2)
#define DIAL_MAX_HITEM 512
#define DIAL_ZTAB_W 8   /* words per control step of Ws::ztab */

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
  // H = M + J^T D J work list: one item per (matrix entry, chunk of <= 4 contacts that touch it).  An entry hit
  // by more contacts than the chunk size is split over P = 2 or 4 adjacent lanes whose partial sums are combined
  // with quad DPP adds.  Packed as  i | j<<5 | pcode<<10 (P = 1,2,4) | writer<<12 | limdiag<<13 | n<<14 |
  // c0<<17 | c1<<20 | c2<<23 | c3<<26.  Groups are ordered P = 4, 2, 1 (so they never straddle a quad) and the
  // P = 1 items by decreasing n; hpass_n[p] = max n among items 64p .. 64p+63.
  int32_t nhitem;
  uint8_t hpass_n[8];
  uint32_t hitem[DIAL_MAX_HITEM];
};

// Host: build the derived tables.  Returns 0 or a negative DIAL_ERR_* code.
static inline int dial_build_derived(const dial_model* m, dial_derived* dv) {
  if (m->nv > 32 || m->nbody > DIAL_MAX_BODY || m->nv > DIAL_MAX_V) return DIAL_ERR_ARG;
  for (int b = 0; b < m->nbody; b++) {   // root-to-body dof paths must fit the ancestor lists (NANC = 12)
    int depth_dofs = 0;
    for (int bb = b; bb > 0; bb = m->body_parent[bb]) depth_dofs += m->body_dofnum[bb];
    if (depth_dofs > 12) return DIAL_ERR_UNSUPPORTED;
  }
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
  // lower-triangle entries that can be non-zero: M[i][j] and (for world-only contacts) H[i][j] vanish unless
  // dof j is an ancestor of dof i (branch-induced sparsity)
  // A contact between two MOVING bodies (push crate: robot vs the sliding crate) couples dofs of different branches: H loses the
  // branch-induced sparsity, every lower-triangle entry can be non-zero (M itself stays sparse: its assembly tests the ancestor
  // relation per entry).  Such models run on the generic instantiation, whose factorisation is dense anyway.
  bool coupled = false;
  for (int c = 0; c < m->ncon; c++) coupled = coupled || (dv->body_ancmask[m->con_body1[c]] != 0 && dv->body_ancmask[m->con_body2[c]] != 0);
  int t = 0;
  for (int i = 0; i < m->nv; i++)
    for (int j = 0; j <= i; j++)
      if (coupled || ((dv->dof_ancmask[i] >> j) & 1u)) dv->tri[t++] = (uint16_t)((i << 8) | j);
  dv->ntri = t;
  dv->nhitem = 0;
  for (int p = 0; p < 8; p++) dv->hpass_n[p] = 0;
  if (m->cone == DIAL_CONE_ELLIPTIC) { dv->nhitem = 0; return DIAL_OK; }   // solver_cone.h assembles H per contact
  if (coupled) return DIAL_OK;   // no sparse H work list: the generic instantiation assembles H over `tri` (dense here)
  // ---- H work list.  Contact c (world vs body2) touches dof i iff i moves body2; j is an ancestor of i, so
  // entry (i, j) is touched by exactly the contacts that touch i.
  if (m->ncon <= 8) {
    int chunk = 4;
    for (int i = 0; i < m->nv; i++) {
      int n = 0;
      for (int c = 0; c < m->ncon; c++) n += ((dv->body_ancmask[m->con_body1[c]] | dv->body_ancmask[m->con_body2[c]]) >> i) & 1u;
      if (n > 0 && n < chunk) chunk = n;
    }
    int nh = 0;
    bool fits = true;
    // stage 0: groups of 4 lanes, stage 1: groups of 2, stages 2..6: single-lane items with n = 4, 3, 2, 1, 0
    for (int stage = 0; stage < 7 && fits; stage++) {
      for (int e = 0; e < t && fits; e++) {
        const int i = dv->tri[e] >> 8, j = dv->tri[e] & 0xff;
        int cl[8], n = 0;
        for (int c = 0; c < m->ncon; c++)
          if (((dv->body_ancmask[m->con_body1[c]] | dv->body_ancmask[m->con_body2[c]]) >> i) & 1u) cl[n++] = c;
        const int parts = n == 0 ? 1 : (n + chunk - 1) / chunk;
        if (parts > 4) { fits = false; break; }
        const int P = parts == 1 ? 1 : (parts == 2 ? 2 : 4);
        const int want = P == 4 ? 0 : (P == 2 ? 1 : 2 + (4 - n));
        if (want != stage) continue;
        for (int q = 0; q < P; q++) {
          if (nh >= DIAL_MAX_HITEM) { fits = false; break; }
          uint32_t h = (uint32_t)i | ((uint32_t)j << 5) | ((uint32_t)(P == 1 ? 0 : (P == 2 ? 1 : 2)) << 10);
          if (q == 0) {
            h |= 1u << 12;
            if (i == j && dv->dof_limrow[i] >= 0) h |= 1u << 13;
          }
          int nq = 0;
          for (int k = q * chunk; k < n && k < (q + 1) * chunk; k++) h |= (uint32_t)cl[k] << (17 + 3 * nq++);
          h |= (uint32_t)nq << 14;
          if (nq > dv->hpass_n[nh >> 6]) dv->hpass_n[nh >> 6] = (uint8_t)nq;
          dv->hitem[nh++] = h;
        }
      }
    }
    if (nh > 8 * 64) fits = false;
    dv->nhitem = fits ? nh : 0;
  }
  return DIAL_OK;
}

// Host: does the derived H work list fit the instantiation's capacity (square layout only)?
template <class D>
static inline bool derived_fits(const dial_derived* dv) {
  if constexpr (D::ell) return true;
  else if constexpr (D::square) return dv->nhitem > 0 && dv->nhitem <= D::NHI;
  else return true;
}

// Host: does an elliptic model fit the per-contact tables of the instantiation?
template <class D>
static inline bool ell_fits(const dial_model* m, const dial_derived* dv) {
  if constexpr (!D::ell) return m->cone != DIAL_CONE_ELLIPTIC;
  else {
    if (m->cone != DIAL_CONE_ELLIPTIC) return false;
    int ne = m->nlim, jcw = 0, dofc[DIAL_MAX_V] = {0};
    for (int c = 0; c < m->ncon; c++) {
      if (m->con_dim[c] != 3 && m->con_dim[c] != 6) return false;
      const uint32_t mask = dv->body_ancmask[m->con_body1[c]] | dv->body_ancmask[m->con_body2[c]];
      int nd = 0;
      for (int i = 0; i < m->nv; i++) if ((mask >> i) & 1u) { nd++; if (++dofc[i] > D::NDC) return false; }
      if (nd > D::NCD || (nd & 1)) return false;   // (even: the row products fetch the compact Jacobian in 8-byte pairs)
      ne += m->con_dim[c];
      jcw += m->con_dim[c] * nd;
    }
    for (int i = 0; i < m->nv; i++) {   // solver_cone.h: NBLK = 6 -- rows of M at most 6 wide (a free body; four-joint finger chains)
      uint32_t mask = dv->dof_ancmask[i] | (1u << i);
      for (int j = 0; j < m->nv; j++) if ((dv->dof_ancmask[j] >> i) & 1u) mask |= 1u << j;
      int lo = 0, hi = m->nv;
      while (!((mask >> lo) & 1u)) lo++;
      while (!((mask >> (hi - 1)) & 1u)) hi--;
      if (hi - lo > 6) return false;
    }
    return ne == D::NE && ne == m->nefc && jcw == D::JCW;
  }
}

// Host: capacity-sized ABI structs -> CModel<D>.  The caller has checked dims_match<D>() for static D.
// constraint._kbi's position-independent part for one row (CModel::jnt_kbi / con_kbi / fri_kbi), in the reference's own fp32 order of
// operations and with true divisions (the device evaluated the same expressions with v_rcp in every step)
static inline void kbi_row(float* o, const float* solref, const float* solimp, float timestep) {
  const float MINIMP = 0.0001f, MAXIMP = 0.9999f, MINVAL = 1e-15f;
  const auto clipf = [](float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); };
  volatile float timeconst = solref[0] > 2.f * timestep ? solref[0] : 2.f * timestep, dampratio = solref[1];   // (volatile: no fast-math re-association on the host)
  volatile float dmin = clipf(solimp[0], MINIMP, MAXIMP), dmax = clipf(solimp[1], MINIMP, MAXIMP);
  volatile float width = solimp[2] > MINVAL ? solimp[2] : MINVAL, mid = clipf(solimp[3], MINIMP, MAXIMP);
  volatile float power = solimp[4] > 1.f ? solimp[4] : 1.f;
  volatile float d2 = dmax * dmax;
  volatile float den = d2 * timeconst; den = den * timeconst; den = den * dampratio; den = den * dampratio;
  volatile float k = 1.f / den, dt = dmax * timeconst, b = 2.f / dt;
  if (solref[0] <= 0.f) k = -solref[0] / d2;
  if (solref[1] <= 0.f) b = -solref[1] / dmax;
  volatile float omm = 1.f - mid;
  o[0] = k; o[1] = b; o[2] = dmin; o[3] = dmax; o[4] = 1.f / width; o[5] = mid;
  if (power == 2.f) { o[6] = 1.f / mid; o[7] = 1.f / omm; }
  else { volatile float pa = std::pow((float)mid, (float)power - 1.f), pb = std::pow((float)omm, (float)power - 1.f); o[6] = 1.f / pa; o[7] = 1.f / pb; }
  o[8] = power; o[9] = 0.f; o[10] = 0.f; o[11] = 0.f;
}

// distinct (solref, solimp) rows among the model's limit rows, contacts and dry-friction rows (CModel::kbi_tab's occupancy)
static inline int kbi_unique_rows(const dial_model* m) {
  int n = 0;
  static thread_local float ref[DIAL_MAX_JNT + DIAL_MAX_CON + DIAL_MAX_FRI][7];
  const auto intern = [&](const float* solref, const float* solimp) {
    float key[7] = {solref[0], solref[1], solimp[0], solimp[1], solimp[2], solimp[3], solimp[4]};
    for (int r = 0; r < n; r++) if (memcmp(ref[r], key, sizeof(key)) == 0) return;
    memcpy(ref[n++], key, sizeof(key));
  };
  for (int l = 0; l < m->nlim; l++) { const int j = m->lim_jnt[l]; intern(m->jnt_solref[j], m->jnt_solimp[j]); }
  for (int c = 0; c < m->ncon; c++) intern(m->con_solref[c], m->con_solimp[c]);
  for (int q = 0; q < m->nfri && q < DIAL_MAX_FRI; q++) intern(m->fri_solref[q], m->fri_solimp[q]);
  return n;
}

template <class D>
static inline void fill_cmodel(CModel<D>* c, const dial_model* m, const dial_task* t, const dial_derived* dv) {
  CModel<D>& o = *c;
  for (size_t i = 0; i < sizeof(o); i++) ((char*)c)[i] = 0;
  o.nq = m->nq; o.nv = m->nv; o.nu = m->nu; o.nbody = m->nbody; o.njnt = m->njnt; o.ngeom = m->ngeom;
  o.nsite = m->nsite; o.ncon = m->ncon; o.nlim = m->nlim; o.nefc = m->nefc;
  o.iterations = m->iterations; o.ls_iterations = m->ls_iterations; o.ls_rule = m->ls_rule; o.nlevel = dv->nlevel; o.ntri = dv->ntri;
  o.timestep = m->timestep; o.tolerance = m->tolerance; o.ls_tolerance = m->ls_tolerance;
  o.impratio = m->impratio; o.meaninertia = m->meaninertia;
  for (int k = 0; k < 3; k++) o.gravity[k] = m->gravity[k];
  o.kin_fast = 1;
  for (int b = 0; b < m->nbody; b++) if (m->body_jntnum[b] > 1) o.kin_fast = 0;
  for (int b = 0; b < m->nbody; b++) {
    o.body_parent[b] = m->body_parent[b]; o.body_jntadr[b] = m->body_jntadr[b]; o.body_jntnum[b] = m->body_jntnum[b];
    o.body_dofadr[b] = m->body_dofadr[b]; o.body_dofnum[b] = m->body_dofnum[b];
    o.body_subtree_end[b] = m->body_subtree_end[b]; o.body_rootid[b] = m->body_rootid[b];
    o.body_ancmask[b] = dv->body_ancmask[b];
    {
      int na = 0;
      for (int i = 0; i < m->nv && na < D::NANC; i++)
        if ((dv->body_ancmask[b] >> i) & 1u) { if constexpr (D::phase_tabs) o.body_anc[b][na] = (uint8_t)i; na++; }
      o.body_nanc[b] = na;
      int flags = 0;
      if (m->body_quat[b][0] == 1.f && m->body_quat[b][1] == 0.f && m->body_quat[b][2] == 0.f && m->body_quat[b][3] == 0.f) flags |= 1;
      bool origin = true;
      for (int ji = m->body_jntadr[b]; ji >= 0 && ji < m->body_jntadr[b] + m->body_jntnum[b]; ji++)
        origin = origin && m->jnt_pos[ji][0] == 0.f && m->jnt_pos[ji][1] == 0.f && m->jnt_pos[ji][2] == 0.f;
      if (origin) flags |= 2;
      if (m->body_jntnum[b] == 1 && m->jnt_type[m->body_jntadr[b]] == DIAL_JNT_FREE) flags |= 4;
      o.body_flags[b] = flags;
    }
    for (int k = 0; k < 3; k++) { o.body_pos[b][k] = m->body_pos[b][k]; o.body_ipos[b][k] = m->body_ipos[b][k]; o.body_inertia[b][k] = m->body_inertia[b][k]; }
    for (int k = 0; k < 4; k++) { o.body_quat[b][k] = m->body_quat[b][k]; o.body_iquat[b][k] = m->body_iquat[b][k]; }
    o.body_mass[b] = m->body_mass[b]; o.body_invweight0[b] = m->body_invweight0[b][0];
    o.lvl_body[b] = dv->lvl_body[b];
    o.body_depth[b] = m->body_depth[b];
  }
  for (int b = 0; b <= m->nbody && b <= D::NB; b++) o.lvl_start[b] = dv->lvl_start[b < DIAL_MAX_BODY + 1 ? b : DIAL_MAX_BODY];
  {  // chains: one per leaf body, listed root first; models that exceed the table sizes get nchain = 0
    int nch = 0;
    bool ok = true;
    for (int b = 1; b < m->nbody && ok; b++) {
      bool leaf = true;
      for (int c2 = b + 1; c2 < m->nbody; c2++) leaf = leaf && m->body_parent[c2] != b;
      if (!leaf) continue;
      int path[64], len = 0;
      for (int bb = b; bb > 0 && len < 64; bb = m->body_parent[bb]) path[len++] = bb;
      if (nch >= D::NCHAIN || len > D::CHAINLEN) { ok = false; break; }
      o.chain_len[nch] = len;
      for (int q = 0; q < len; q++) o.chain_body[nch][q] = (uint8_t)path[len - 1 - q];
      nch++;
    }
    o.nchain = ok ? nch : 0;
    // the exclusive tail of every chain (bodies below its last branching body) and the "shared" bodies that no
    // tail covers -- the branching bodies and everything above them -- listed deepest first with their children
    o.nshared = -1;
    if (ok) {
      int nchild[DIAL_MAX_BODY] = {0};
      bool covered[DIAL_MAX_BODY] = {false};
      for (int b = 1; b < m->nbody; b++) nchild[m->body_parent[b]]++;
      for (int c = 0; c < nch; c++) {
        int q = o.chain_len[c];
        while (q > 0 && nchild[o.chain_body[c][q - 1]] < 2) q--;   // first body after the last branching one
        o.chain_excl[c] = q;
        for (int r = q; r < o.chain_len[c]; r++) covered[o.chain_body[c][r]] = true;
      }
      int ns = 0;
      bool fits = true;
      for (int b = m->nbody - 1; b >= 1 && fits; b--) {      // DFS numbering: higher index first = deepest first
        if (covered[b]) continue;
        if (ns >= 4 || nchild[b] > 4) { fits = false; break; }
        o.shared_body[ns] = b;
        int k = 0;
        for (int c2 = b + 1; c2 < m->nbody; c2++)
          if (m->body_parent[c2] == b) o.shared_child[ns][k++] = (uint8_t)c2;
        o.shared_nchild[ns] = k;
        ns++;
      }
      if (fits) o.nshared = ns;
    }
  }
  if constexpr ((!D::gen && D::square && RowsOf<typename D::Topo>::maxd > 0) || D::rows_gen) {
    int ms, md;
    rows_build(m, &o.rows, RowsOf<typename D::Topo>::maxd, &ms, &md, RowsOf<typename D::Topo>::static_root, D::rows_gen);   // (dims_match<D> has checked that it succeeds)
  }
  for (int j = 0; j < m->njnt; j++) {
    o.jnt_type[j] = m->jnt_type[j]; o.jnt_qposadr[j] = m->jnt_qposadr[j]; o.jnt_dofadr[j] = m->jnt_dofadr[j];
    o.jnt_bodyid[j] = m->jnt_bodyid[j]; o.jnt_margin[j] = m->jnt_margin[j];
    for (int k = 0; k < 3; k++) { o.jnt_pos[j][k] = m->jnt_pos[j][k]; o.jnt_axis[j][k] = m->jnt_axis[j][k]; }
    for (int k = 0; k < 2; k++) o.jnt_range[j][k] = m->jnt_range[j][k];
    o.jnt_kbi[j] = 0;
  }
  for (int i = 0; i < m->nq; i++) o.qpos0[i] = m->qpos0[i];
  for (int i = 0; i < m->nv; i++) {
    o.dof_bodyid[i] = m->dof_bodyid[i]; o.dof_jntid[i] = m->dof_jntid[i]; o.dof_act[i] = dv->dof_act[i];
    o.dof_limrow[i] = dv->dof_limrow[i]; o.dof_ancmask[i] = dv->dof_ancmask[i];
    o.dof_descmask[i] = 0;
    for (int j = 0; j < m->nv; j++) if ((dv->dof_ancmask[j] >> i) & 1u) o.dof_descmask[i] |= (1u << j);
    o.dof_armature[i] = m->dof_armature[i]; o.dof_damping[i] = m->dof_damping[i]; o.dof_invweight0[i] = m->dof_invweight0[i];
  }
  for (int i = 0; i < m->nv; i++) {   // column range of row i of M that can be non-zero: from its first ancestor dof to its last
    // descendant dof (depth-first numbering).  Entries of that range that belong to a sibling branch are exact zeros.
    const uint32_t mask = o.dof_ancmask[i] | o.dof_descmask[i] | (1u << i);
    int lo = 0, hi = m->nv;
    while (!((mask >> lo) & 1u)) lo++;
    while (!((mask >> (hi - 1)) & 1u)) hi--;
    o.dof_blk0[i] = lo; o.dof_blk1[i] = hi;
  }
  if constexpr (D::phase_tabs) for (int e = 0; e < dv->ntri; e++) o.tri[e] = dv->tri[e];
  if constexpr (D::square && !D::ell) {
    static_assert(D::NHI % 64 == 0 && D::NV * D::T < 1024 && D::NV * D::S < 1024 && D::NE + 4 < 64, "hrec field widths");
    o.nhitem = dv->nhitem <= D::NHI ? dv->nhitem : 0;
    for (int p = 0; p < 8; p++) o.hpass_n[p] = dv->hpass_n[p];
    for (int e = 0; e < D::NHI; e++) {
      uint32_t w0 = 0, w1 = (uint32_t)(D::NLP + 4 * D::NC) << 20;   // no-op item: reads valid words, writes nothing
      if (e < o.nhitem) {
        const uint32_t h = dv->hitem[e];
        const uint32_t i = h & 31u, j = (h >> 5) & 31u, pc = (h >> 10) & 3u, wr = (h >> 12) & 1u, ld = (h >> 13) & 1u;
        const uint32_t n = (h >> 14) & 7u;
        w0 = (i * D::T) | ((j * D::T) << 10);
        for (uint32_t q = 0; q < 4; q++) w0 |= ((h >> (17 + 3 * q)) & 7u) << (20 + 3 * q);
        const uint32_t lim = ld ? (uint32_t)dv->dof_limrow[i] : (uint32_t)(D::NLP + 4 * D::NC);
        w1 = (i * D::S + j) | ((j * D::S + i) << 10) | (lim << 20) | (pc << 26) | (wr << 28) | (n << 29);
      }
      o.hrec[e][0] = w0;
      o.hrec[e][1] = w1;
    }
  }
  for (int g = 0; g < m->ngeom; g++) {
    o.geom_bodyid[g] = m->geom_bodyid[g];
    for (int k = 0; k < 3; k++) { o.geom_pos[g][k] = m->geom_pos[g][k]; o.geom_size[g][k] = m->geom_size[g][k]; }
    for (int k = 0; k < 4; k++) o.geom_quat[g][k] = m->geom_quat[g][k];
  }
  o.quad_site_is_geom = 0;
  if constexpr (D::quad_stage) {
    bool same = m->nsite >= 5 && m->ngeom >= 5;
    for (int r = 1; same && r <= 4; r++) for (int k = 0; k < 3; k++) same = same && m->site_pos[r][k] == m->geom_pos[r][k];
    o.quad_site_is_geom = same ? 1 : 0;
  }
  {   // (math.quat_to_3x3's third column, dmath.h: quat_to_mat)
    const float w = m->geom_quat[0][0], x = m->geom_quat[0][1], y = m->geom_quat[0][2], z = m->geom_quat[0][3];
    o.geom0_normal[0] = 2.f * (x * z + w * y); o.geom0_normal[1] = 2.f * (y * z - w * x); o.geom0_normal[2] = w * w - x * x - y * y + z * z;
  }
  for (int s = 0; s < m->nsite; s++) {
    o.site_bodyid[s] = m->site_bodyid[s];
    for (int k = 0; k < 3; k++) o.site_pos[s][k] = m->site_pos[s][k];
    for (int k = 0; k < 4; k++) o.site_quat[s][k] = m->site_quat[s][k];
  }
  for (int cidx = 0; cidx < m->ncon; cidx++) {
    o.con_kind[cidx] = m->con_kind[cidx]; o.con_geom1[cidx] = m->con_geom1[cidx]; o.con_geom2[cidx] = m->con_geom2[cidx];
    o.con_body1[cidx] = m->con_body1[cidx]; o.con_body2[cidx] = m->con_body2[cidx]; o.con_margin[cidx] = m->con_margin[cidx];
    for (int k = 0; k < 5; k++) o.con_friction[cidx][k] = m->con_friction[cidx][k];
    {   // constraint._efc_contact_pyramidal / _elliptic: the rows' inverse weights, in the reference's own order of operations
      const float t = m->body_invweight0[m->con_body1[cidx]][0] + m->body_invweight0[m->con_body2[cidx]][0], mu = m->con_friction[cidx][0];
      if constexpr (D::ell) { o.con_invw[cidx][0] = t; o.con_invw[cidx][1] = t / m->impratio; }
      else {
        float invweight = t + mu * mu * t;
        invweight = invweight * 2.f * mu * mu / m->impratio;
        o.con_invw[cidx][0] = invweight;
      }
    }
  }
  {   // the impedance table: one row per distinct (solref, solimp) among the limit rows, the contacts and the dry-friction rows
    int n = 0;
    float ref[CModel<D>::NKBI][7];
    const auto intern = [&](const float* solref, const float* solimp) -> uint8_t {
      float key[7] = {solref[0], solref[1], solimp[0], solimp[1], solimp[2], solimp[3], solimp[4]};
      for (int r = 0; r < n; r++) if (memcmp(ref[r], key, sizeof(key)) == 0) return (uint8_t)r;
      if (n >= CModel<D>::NKBI) return 0;   // (dial_create has checked kbi_unique_rows() against the instantiation's capacity)
      memcpy(ref[n], key, sizeof(key));
      kbi_row(o.kbi_tab[n], solref, solimp, m->timestep);
      return (uint8_t)n++;
    };
    for (int l = 0; l < m->nlim; l++) { const int j = m->lim_jnt[l]; o.jnt_kbi[j] = intern(m->jnt_solref[j], m->jnt_solimp[j]); }
    for (int c = 0; c < m->ncon; c++) o.con_kbi[c] = intern(m->con_solref[c], m->con_solimp[c]);
    if constexpr (D::gen) for (int q = 0; q < m->nfri && q < DIAL_MAX_FRI; q++) o.fri_kbi[q] = intern(m->fri_solref[q], m->fri_solimp[q]);
  }
  o.cone = m->cone; o.eulerdamp = m->eulerdamp;
  if constexpr (D::gen) {
    int nbb = 0;
    for (int cidx = 0; cidx < m->ncon; cidx++) {
      o.con_sub[cidx] = m->con_sub[cidx];
      o.con_bbslot[cidx] = m->con_kind[cidx] == DIAL_CON_BOX_BOX ? nbb++ : 0;
    }
    for (int f = 0; f < DIAL_MAX_FEET; f++) o.crate_contact[f] = t->crate_contact[f];
    for (int k = 0; k < 6; k++) o.crate_region[k] = t->crate_region[k];
    for (int k = 0; k < 3; k++) o.head_vec[k] = t->head_vec[k];
    o.nfri = m->nfri;
    for (int i = 0; i < m->nv; i++) o.dof_frirow[i] = -1;
    for (int q = 0; q < m->nfri && q < DIAL_MAX_FRI; q++) {
      o.fri_dof[q] = m->fri_dof[q]; o.fri_loss[q] = m->fri_loss[q];
      o.dof_frirow[m->fri_dof[q]] = m->nlim + q;
    }
    for (int f = 0; f < 2; f++) { o.pc_wanted[f] = t->pc_wanted[f]; for (int k = 0; k < 2; k++) o.pc_foot_contact[f][k] = t->pc_foot_contact[f][k]; }
    o.pc_n_unwanted = t->pc_n_unwanted;
    for (int k = 0; k < 16; k++) o.pc_unwanted[k] = t->pc_unwanted[k];
    o.pc_wanted_zmax = t->pc_wanted_zmax;
  }
  if constexpr (D::ell) {
    int adr = m->nlim, joff = 0;
    for (int i = 0; i < m->nv; i++) o.dof_ncon[i] = 0;
    {
      int e = 0;
      for (int a = 0; a < D::NCD; a++) for (int b = 0; b <= a; b++) { o.pair_a[e] = (uint8_t)a; o.pair_b[e] = (uint8_t)b; e++; }
    }
    for (int cidx = 0; cidx < m->ncon; cidx++) {
      o.con_dim[cidx] = m->con_dim[cidx];
      o.con_adr[cidx] = adr;
      adr += m->con_dim[cidx];
      const uint32_t mask = dv->body_ancmask[m->con_body1[cidx]] | dv->body_ancmask[m->con_body2[cidx]];
      int nd = 0;
      for (int i = 0; i < m->nv; i++)
        if ((mask >> i) & 1u) {
          if (nd < D::NCD) o.con_dof[cidx][nd] = (uint8_t)i;
          if (o.dof_ncon[i] < D::NDC) o.dof_con[i][o.dof_ncon[i]] = (uint16_t)(cidx | (nd << 8));
          o.dof_ncon[i]++;
          nd++;
        }
      for (int i = 0; i < m->nv; i++) o.con_dofpos[cidx][i] = 255;
      for (int q = 0; q < nd && q < D::NCD; q++) o.con_dofpos[cidx][o.con_dof[cidx][q]] = (uint8_t)q;
      o.con_ndof[cidx] = nd;
      o.con_joff[cidx] = joff;
      joff += m->con_dim[cidx] * nd;
    }
  }
  for (int l = 0; l < m->nlim; l++) o.lim_jnt[l] = m->lim_jnt[l];
  for (int a = 0; a < m->nu; a++) {
    o.act_qposadr[a] = m->act_qposadr[a]; o.act_ctrllimited[a] = m->act_ctrllimited[a];
    o.act_isposition[a] = m->act_isposition[a]; o.act_gear[a] = m->act_gear[a]; o.act_kp[a] = m->act_kp[a];
    o.act_ctrlrange[a][0] = m->act_ctrlrange[a][0]; o.act_ctrlrange[a][1] = m->act_ctrlrange[a][1];
    o.kp[a] = t->kp[a]; o.kd[a] = t->kd[a];
    for (int k = 0; k < 2; k++) { o.joint_range[a][k] = t->joint_range[a][k]; o.phys_range[a][k] = t->phys_range[a][k]; o.tau_range[a][k] = t->tau_range[a][k]; }
    o.joint_offset[a] = t->joint_offset[a];
  }
  o.kind = t->kind; o.n_frames = t->n_frames; o.position_control = t->position_control; o.torso_x = t->torso_x;
  o.upright_x = t->upright_x; o.nfeet = t->nfeet; o.n_stage = t->n_stage;
  o.randomize_tasks = t->randomize_tasks && t->n_cmd > 0; o.n_cmd = t->n_cmd;
  for (int f = 0; f < DIAL_MAX_FEET; f++) { o.feet_site[f] = t->feet_site[f]; o.gait_phase[f] = t->gait_phase[f]; }
  o.dt = t->dt; o.action_scale = t->action_scale; o.foot_radius = t->foot_radius; o.gait_duty = t->gait_duty;
  o.gait_cadence = t->gait_cadence; o.gait_amp = t->gait_amp; o.ramp_up_time = t->ramp_up_time;
  o.done_height = t->done_height; o.jump_dt = t->jump_dt;
  for (int k = 0; k < 3; k++) { o.cmd_vel[k] = t->cmd_vel[k]; o.cmd_ang_vel[k] = t->cmd_ang_vel[k]; o.init_pos_tar[k] = t->init_pos_tar[k]; o.init_ang_vel_tar[k] = t->init_ang_vel_tar[k]; }
}

// ---- per-wavefront LDS workspace (pointers into one float array) -----------------------------
// Arrays that are dead before the constraint solver starts share their storage with arrays that only live
// inside the solver ("union" below); symmetric matrices are stored as packed lower triangles (generic path) or
// as full nv x S squares (`square`, the dimension-specialised instantiations).
struct Ws {
  float *qpos, *qvel, *warm, *info, *ctrl, *act, *Y, *ztar, *rpart;
  float *xpos, *xquat, *spos, *com, *cvel, *cdof;
  float *M, *L;
  float *cdist, *cpos, *cframe, *Jc;
  float *D, *aref, *lsign, *Jaref, *qfs, *qas, *qacc, *Ma, *rhs;
  // dynamics temporaries (dead after the contact-Jacobian phase) ...
  float *xmat, *xipos, *ximat, *xanchor, *xaxis, *gpos, *gaxis, *cinert, *cdofdot, *cacc, *crb, *cfl, *cfrc, *Fd;
  // ... aliased by solver-only arrays
  float *H, *JarefW, *JarefS, *jv, *frc, *quad, *MaW, *MaS, *grad, *search, *mv, *qfc, *ysol;
  // elliptic models (solver_cone.h): contact-on flags, per-contact cone Hessian weights, per-dof vectors that the row
  // products gather from
  float *con_on, *cwd, *cwa, *cwb, *ccf, *vec0, *vec1, *ulist;
  // generic instantiation: the contacts that touch, compacted (rollout_body.h: con_of)
  float *clist, *sq;   // sq: DIAL_MAX_V x DIAL_MAX_V square the register Cholesky reads (rollout_body.h: solve_spd_reg)
  float *cmu;          // friction coefficients (mu1, mu2) of the compact contacts: the solver's loops read them by compact index
  // generic instantiation, rollout kernel: the Jacobian and the per-row arrays above are sized for con_cap touching contacts
  // (0 = for all ncon); `ovf` = this sample's full-size copy of them in global memory (ws_overflow), used when more touch
  float* ovf;
  int con_cap;
  // control tables of a whole rollout, built once in its prologue (rollout_driver.h; instantiations with Dims::pre_ctrl):
  // jtab[t * nu + a] = the joint target act2joint(u_t)[a] of control step t, ztab[t * DIAL_ZTAB_W + f] = the gait clock's foot
  // height of step t (f < 4) and the step's ramped velocity targets (4, 5: v_x, v_y; 6: yaw rate; 7: yaw) -- none depends on the
  // state, so K2 / act2joint / get_foot_step / the ramps leave the per-step dependence chain
  float *jtab, *ztab;
};

#if defined(__HIPCC__)
#define WS_HD __host__ __device__ inline
#else
#define WS_HD inline
#endif

WS_HD int tri_idx(int i, int j) { return (i * (i + 1)) / 2 + j; }   // i >= j

// Carve the workspace out of `base`; returns the number of words used (call with base = nullptr to size
// the dynamic LDS allocation).  `with_L`: keep a packed Cholesky factor in LDS (LDS solver path).
// `square`: the register solver's square layout (Dims::square): M and H are nv x S squares, Jc holds the dof-major
// pyramid rows J^T[i][4c + e], the transpose scratch L aliases H, frc is padded so that the contact weights start
// 16-byte aligned.
// `con_cap` (generic pyramidal layout only, 0 = off): size the contact Jacobian and the per-row arrays for that many TOUCHING
// contacts instead of all ncon candidates (crate scene: 52 candidates, 4-8 touch; 31.6 KB -> 17.4 KB per wavefront at a cap of
// 14, i.e. 9 instead of 5 wavefronts per CU); a sample that touches with more runs on ws_overflow's arrays.
// `sq_n`: dimension of the generic solver's dense square `sq` (Dims::NVP).
WS_HD int ws_carve(Ws& s, float* base, int nq, int nv, int nu, int nbody, int njnt, int ngeom, int nsite,
                   int ncon_all, int nefc_all, int nnode, bool with_L, bool square = false, int ell_jcw = 0, int con_cap = 0,
                   int sq_n = DIAL_MAX_V, int tab_steps = 0) {
  int o = 0;
  const bool capped = con_cap > 0 && con_cap < ncon_all && !square && ell_jcw == 0;
  const int ncon = ncon_all;                                            // arrays indexed by the model's contact index
  const int nefc = capped ? nefc_all - 4 * (ncon_all - con_cap) : nefc_all;   // arrays indexed by (compact) row
  s.con_cap = capped ? con_cap : 0;
  s.ovf = nullptr;
  const int ntri = square ? nv * ((nv + 3) & ~3) : (nv * (nv + 1)) / 2;
  const int njc = ell_jcw > 0 ? ell_jcw : (square ? nv * 4 * ncon : (capped ? con_cap : ncon) * 3 * nv);
  const int ell = ell_jcw > 0 ? 1 : 0;
#define WS_TAKE(name, n) s.name = base + o; o += (((n) + 3) & ~3);
  WS_TAKE(qpos, nq) WS_TAKE(qvel, nv) WS_TAKE(warm, nv) WS_TAKE(info, DIAL_INFO_N) WS_TAKE(ctrl, nu)
  WS_TAKE(act, nu) WS_TAKE(ztar, DIAL_MAX_FEET) WS_TAKE(rpart, 10)
  WS_TAKE(xpos, nbody * 3) WS_TAKE(xquat, nbody * 4) WS_TAKE(spos, nsite * 3) WS_TAKE(com, nbody * 3)
  WS_TAKE(cvel, nbody * 6) WS_TAKE(cdof, nv * 6)
  WS_TAKE(M, ntri)
  WS_TAKE(cdist, ncon) WS_TAKE(cpos, ncon * 3) WS_TAKE(cframe, ncon * 9) WS_TAKE(Jc, njc)
  WS_TAKE(D, nefc) WS_TAKE(aref, nefc) WS_TAKE(lsign, nefc) WS_TAKE(Jaref, nefc)
  WS_TAKE(qfs, nv) WS_TAKE(qas, nv) WS_TAKE(qacc, nv) WS_TAKE(Ma, nv) WS_TAKE(rhs, nv)
  WS_TAKE(con_on, ell * ncon) WS_TAKE(qfc, ell * nv) WS_TAKE(ulist, ell * (nefc > 0 ? 68 : 0))
  WS_TAKE(clist, with_L ? ncon : 0) WS_TAKE(cmu, with_L ? 2 * (capped ? con_cap : ncon) : 0)
  const int u0 = o;
  // A1: dead after the cinert/cdof phase ...
  WS_TAKE(xmat, 0) WS_TAKE(xipos, nbody * 3) WS_TAKE(ximat, nbody * 9) WS_TAKE(xanchor, njnt * 3)
  WS_TAKE(xaxis, njnt * 3)
  const int a1_end = o;
  o = u0;   // ... so the velocity-dependent temporaries written after that phase reuse it
  WS_TAKE(cdofdot, nv * 6) WS_TAKE(cacc, nbody * 6) WS_TAKE(cfl, nbody * 6) WS_TAKE(cfrc, nbody * 6)
  o = o > a1_end ? o : a1_end;
  WS_TAKE(gpos, ngeom * 3) WS_TAKE(gaxis, ngeom * 3)
  const int ci0 = o;
  WS_TAKE(cinert, nbody * 10)
  const int ci1 = o;
  o = ci0;  // F_i = crb * cdof is written after the last read of cinert
  WS_TAKE(Fd, nv * 6)
  o = o > ci1 ? o : ci1;
  WS_TAKE(crb, nbody * 10)
  const int u1 = o;
  o = u0;
  WS_TAKE(H, ntri) WS_TAKE(jv, nefc) WS_TAKE(frc, nefc + 4)
  // `tab_steps` = T (instantiations whose position / velocity stage runs in registers and touches none of the A1 temporaries this
  // region aliases -- Dims::pre_ctrl --, else 0): the rollout's control tables live behind the solver's arrays for its whole length
  WS_TAKE(jtab, tab_steps * nu) WS_TAKE(ztab, tab_steps * DIAL_ZTAB_W)
  WS_TAKE(cwd, ell * ncon * 6) WS_TAKE(cwa, ell * ncon * 6) WS_TAKE(cwb, ell * ncon * 6) WS_TAKE(ccf, ell * ncon * 4)
  WS_TAKE(vec0, ell * nv) WS_TAKE(vec1, ell * nv)
  const int ls = with_L ? 1 : 0;   // the rest is LDS-solver state; the register solver keeps it in VGPRs
  WS_TAKE(JarefW, ls * nefc) WS_TAKE(JarefS, ls * nefc) WS_TAKE(quad, ls * nefc * 3) WS_TAKE(MaW, ls * nv)
  WS_TAKE(MaS, ls * nv) WS_TAKE(grad, ls * nv) WS_TAKE(search, ls * nv) WS_TAKE(mv, ls * nv)
  if (!ell) { WS_TAKE(qfc, ls * nv) }
  WS_TAKE(ysol, ls * nv)
#ifdef DIAL_LDS_CHOL
  WS_TAKE(L, with_L ? ntri : 0)   // packed Cholesky factor of the LDS Cholesky (reference implementation, rollout_body.h: solve_spd)
#else
  WS_TAKE(L, 0)                   // (the register L D L^T keeps its factor in VGPRs and the square `sq`)
#endif
  WS_TAKE(sq, with_L ? sq_n * ((sq_n + 3) & ~3) : 0)
  o = o > u1 ? o : u1;
  WS_TAKE(Y, nnode * nu)   // last: its size is the only run-time quantity, every other offset is a constant
#undef WS_TAKE
  return o;
}

// Full-size copies (all ncon candidates, all nefc rows) of the arrays ws_carve sizes by con_cap; returns the words used.
WS_HD int ws_overflow(Ws& s, float* base, int nv, int ncon, int nefc) {
  int o = 0;
#define WS_TAKE(name, n) s.name = base + o; o += (((n) + 3) & ~3);
  WS_TAKE(Jc, ncon * 3 * nv)
  WS_TAKE(D, nefc) WS_TAKE(aref, nefc) WS_TAKE(lsign, nefc) WS_TAKE(Jaref, nefc)
  WS_TAKE(jv, nefc) WS_TAKE(frc, nefc + 4) WS_TAKE(JarefW, nefc) WS_TAKE(JarefS, nefc) WS_TAKE(quad, nefc * 3) WS_TAKE(cmu, 2 * ncon)
#undef WS_TAKE
  return o;
}

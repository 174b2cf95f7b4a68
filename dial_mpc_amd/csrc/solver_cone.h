// solver_cone.h -- Newton solver (solver.solve of MJX) for models with ELLIPTIC friction cones (Dims::ell: the
// Allegro hand, 16 limit rows + 14 condim-3 + 5 condim-6 contacts = 88 rows, more than a wavefront has lanes).
//
// One wavefront owns the sample, so every data-dependent decision (contact on / off, cone zone, convergence) is
// wave-uniform and costs no divergence.  The solver is organised around CONTACTS, not rows:
//   * "unit" lane u < NL + NC owns limit row u or contact u - NL: zone classification, forces, cost, the cone's
//     Hessian weights and the line-search terms of its unit (35 lanes for Allegro);
//   * dof lane i < NV keeps qacc, Ma, grad, search, M search in registers;
//   * row products J v run over (contact, row) items with the compact Jacobian (only the <= 10 dofs that move the
//     contact's two bodies), J^T f over the static contact list of every dof;
//   * H = M + sum_c J_c^T W_c J_c is accumulated contact by contact over (dof, dof) pairs of the contact's dof set;
//     contacts that are off (dist >= margin) or in the cone's top zone are skipped outright -- with the ball resting
//     on three fingertips 3 of the 19 contacts do any work;
//   * the units that can contribute at all (contacts that are on, limit rows past their range) are compacted into a
//     list once per solve (ballot + mbcnt): row products, the Jaref update and the line search run over that list --
//     with <= 16 such units the three trial points of a line-search iteration are evaluated in ONE pass by three
//     16-lane groups (six interleaved 16-lane DPP reductions per pass);
//   * H couples every dof with every other, so it is factorised by the register L D L^T of solver_reg.h with the
//     dense elimination order; M (and M + dt B of the implicit damping) keep the tree order.
//
// Per contact (rows r0 .. r0+dim-1, friction f, mu = f_0 / sqrt(impratio)), with U = (mu r_0, f_1 r_1, ...),
// N = U_0, T = |U_1..|:  top zone (N >= mu T): nothing;  bottom zone (mu N + T <= 0): the rows are plain quadratic
// rows;  middle zone: cost 1/2 Dm (N - mu T)^2, Dm = D_0 / (mu^2 (1 + mu^2)), force and Hessian
//   Hc = Dm F [[1, c0 U^T], [c0 U, c2 U U^T + c1 I]] F,  F = diag(mu, f),  c0 = -mu / T, c1 = mu^2 - mu N / T, c2 = mu N / T^3
// (solver._update_constraint / _update_gradient / _eval_pt_elliptic of MJX; the oracle restates the same formulas).
#pragma once
// (included from rollout_body.h after solver_reg.h)

namespace dial {

template <class W, class M>
DIAL_DEV void solver_cone(W& w, const M* m, const Ws& s) {
  using D = typename M::D;
  constexpr int NV = D::NV, NC = D::NC, NL = D::NL, S = D::S, NCD = D::NCD, NU_ = NL + NC;
  static_assert(D::ell && D::square && NU_ <= 64 && NV <= 32, "unit lanes / dof lanes must fit one wavefront");
  constexpr int NBLK = 6;   // widest diagonal block of M: a free body (checked on the host: ell_fits)
  w.begin_region();
  const vfloat vzero = vsplat(0.f);
  const vbool isdof = w.lane_lt(NV);
  const float mu_scale = 1.f / DM_SQRT(m->impratio);   // mu = friction_0 / sqrt(impratio)

  // ---- units that can contribute: limit rows past their range (D > 0) and contacts that are on
  const int n_on = w.compact(NU_, [&](int u) { return u < NL ? s.D[u] > 0.f : s.con_on[u - NL] != 0.f; }, s.ulist);
  // ---- row products  out[r] = J_r . vec  over the rows of those units (two vectors at once when vecB != nullptr)
  auto row_products = [&](const float* vecA, float* outA, const float* vecB, float* outB) {
    w.items(6 * n_on, [&](int it) {
      const int idx = it / 6, k = it - 6 * idx, u = (int)s.ulist[idx];
      if (u < NL) {
        if (k != 0) return;
        const int dof = m->jnt_dofadr[m->lim_jnt[u]];
        outA[u] = s.lsign[u] * vecA[dof];
        if (vecB) outB[u] = s.lsign[u] * vecB[dof];
        return;
      }
      const int c = u - NL;
      if (k >= m->con_dim[c]) return;
      const int r = m->con_adr[c] + k, nd = m->con_ndof[c];
      const float* J = s.Jc + m->con_joff[c] + k * nd;
      // The row of the compact Jacobian in 8-byte pairs (dim * nd and nd are even: rows start 8-byte aligned) and the
      // contact's dof list as three 32-bit words, with a FIXED trip count (pairs >= nd masked): ~18 independent LDS
      // fetches issued up front instead of two dependent round trips per column of a loop of unknown length
      const uint32_t* dw = reinterpret_cast<const uint32_t*>(m->con_dof[c]);
      const uint32_t dws[3] = {dw[0], dw[1], dw[2]};
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int t = 0; t < (NCD + 1) / 2; t++) {
        const bool on = 2 * t < nd;
        float j0, j1;
        load2(J + (on ? 2 * t : 0), j0, j1);
        const int i0 = (int)((dws[(2 * t) >> 2] >> (8 * ((2 * t) & 3))) & 255u), i1 = (int)((dws[(2 * t + 1) >> 2] >> (8 * ((2 * t + 1) & 3))) & 255u);
        const int g0 = on ? i0 : 0, g1 = on ? i1 : 0;
        float ta = a + j0 * vecA[g0];
        ta = ta + j1 * vecA[g1];
        a = on ? ta : a;
        if (vecB) { float tb = b + j0 * vecB[g0]; tb = tb + j1 * vecB[g1]; b = on ? tb : b; }
      }
      outA[r] = a;
      if (vecB) outB[r] = b;
    });
  };
  // rows of the contributing units: r -> f(r)
  auto for_on_rows = [&](auto f) {
    w.items(6 * n_on, [&](int it) {
      const int idx = it / 6, k = it - 6 * idx, u = (int)s.ulist[idx];
      if (u < NL) { if (k == 0) f(u); return; }
      const int c = u - NL;
      if (k < m->con_dim[c]) f(m->con_adr[c] + k);
    });
  };
  // ================================================================ ROW layout of the contributing units (round 6)
  // With <= 8 contributing units (the rule: 2.4 on average with the ball in the hand) lane 8 idx + k owns ROW k of unit ulist[idx]
  // (a limit row is a one-row unit), and everything that walked a unit's <= 6 rows serially in ONE lane -- the zone / force / weight
  // evaluation of every Newton iteration, the two warm-start costs of every solve, the line search's per-unit coefficients -- is one
  // pass of straight-line code per row lane plus sums over aligned groups of eight lanes (three DPP steps: wave.h seg8_sumN).  The
  // row's Jaref lives in a register of its lane for the whole solve (the LDS copy is kept for the dof-side consumers).  The unit-lane
  // code below stays as the path for more than 8 units.  Lone rollout: zones 3.2 k -> , line-search set-up 5.3 k -> , warm start
  // 7.2 k -> cycles (profiles/r06_sections_allegro_reorient_cycles.txt); sums over a unit's rows associate differently: rounding level.
  const bool r8 = n_on <= 8;
  // per row lane: [0] row index | kind << 7 | (contact + 1) << 10 as an exact float (kind: 0 idle, 1 limit row, 2 contact row 0, 3 contact
  // row k > 0; one register instead of three: the queue kernel sits on the 168-VGPR budget), [1] friction factor of the row (k = 0: mu;
  // limit row: 1), [2] D of the row, [3] mu of the unit (0: limit row), [4] Dm of the unit
  vfloat RL[5];
  const auto rl_kind = [&](int l) { return ((int)lane_val(RL[0], l) >> 7) & 7; };
  const auto rl_row = [&](int l) { return (int)lane_val(RL[0], l) & 127; };
  const auto rl_con = [&](int l) { return ((int)lane_val(RL[0], l) >> 10) - 1; };
  static_assert(D::NE <= 128 && D::NC < 1000, "packed row code");
  if (r8) w.per_lane_n(RL, [&](int l, float* o) {
    for (int k = 0; k < 5; k++) o[k] = 0.f;
    const int idx = l >> 3, k = l & 7;
    if (idx >= n_on) return;
    const int u = (int)s.ulist[idx];
    if (u < NL) {
      if (k != 0) return;
      o[0] = (float)(u | (1 << 7)); o[1] = 1.f; o[2] = s.D[u];
      return;
    }
    const int c = u - NL, dim = m->con_dim[c], r0 = m->con_adr[c];
    if (k >= dim) return;
    const float mu = m->con_friction[c][0] * mu_scale;
    o[0] = (float)((r0 + k) | ((k == 0 ? 2 : 3) << 7) | ((c + 1) << 10));
    o[1] = k == 0 ? mu : m->con_friction[c][k > 0 ? k - 1 : 0];
    o[2] = s.D[r0 + k];
    o[3] = mu;
    o[4] = s.D[r0] / dm::fmaxf_(mu * mu * (1.f + mu * mu), MJ_MINVAL);
  });
  // J_r . vec for the row of every lane (two vectors at once when vecB != nullptr): the inner product of row_products below
  auto row_products_r8 = [&](const float* vecA, const float* vecB, vfloat& outA, vfloat& outB) {
    vfloat ab[2];
    w.per_lane_n(ab, [&](int l, float* o) {
      o[0] = 0.f; o[1] = 0.f;
      const int kind = rl_kind(l);
      if (kind == 0) return;
      const int r = rl_row(l);
      if (kind == 1) {
        const int dof = m->jnt_dofadr[m->lim_jnt[r]];
        o[0] = s.lsign[r] * vecA[dof];
        if (vecB) o[1] = s.lsign[r] * vecB[dof];
        return;
      }
      const int c = rl_con(l), k = l & 7, nd = m->con_ndof[c];
      const float* J = s.Jc + m->con_joff[c] + k * nd;
      const uint32_t* dw = reinterpret_cast<const uint32_t*>(m->con_dof[c]);
      const uint32_t dws[3] = {dw[0], dw[1], dw[2]};
      float a = 0.f, b = 0.f;
#pragma unroll
      for (int t = 0; t < (NCD + 1) / 2; t++) {
        const bool on = 2 * t < nd;
        float j0, j1;
        load2(J + (on ? 2 * t : 0), j0, j1);
        const int i0 = (int)((dws[(2 * t) >> 2] >> (8 * ((2 * t) & 3))) & 255u), i1 = (int)((dws[(2 * t + 1) >> 2] >> (8 * ((2 * t + 1) & 3))) & 255u);
        const int g0 = on ? i0 : 0, g1 = on ? i1 : 0;
        float ta = a + j0 * vecA[g0];
        ta = ta + j1 * vecA[g1];
        a = on ? ta : a;
        if (vecB) { float tb = b + j0 * vecB[g0]; tb = tb + j1 * vecB[g1]; b = on ? tb : b; }
      }
      o[0] = a; o[1] = b;
    });
    outA = ab[0]; outB = ab[1];
  };
  // the unit evaluation in the row layout: vja = the row values; returns every lane's share of the cost (their wave sum is the total).
  // STORE: forces, zone, Hessian weights -- what unit_cost(.., true) writes for the contributing units
  auto unit_eval_r8 = [&](const vfloat& vja, bool store) -> vfloat {
    vfloat t[2];   // tangential part |U_1..|^2 and the normal part U_0 of the lane's unit, in every lane of its group of eight
    w.per_lane_n(t, [&](int l, float* o) {
      const int kind = rl_kind(l);
      const float U = lane_val(vja, l) * lane_val(RL[1], l);
      o[0] = kind == 3 ? U * U : 0.f;
      o[1] = (kind == 1 || kind == 2) ? U : 0.f;
    });
    w.seg8_sumN(t);
    const vfloat out = w.per_lane([&](int l) -> float {
      const int kind = rl_kind(l);
      if (kind == 0) return 0.f;
      const float jr = lane_val(vja, l), fr = lane_val(RL[1], l), Dk = lane_val(RL[2], l), mu = lane_val(RL[3], l), Dm = lane_val(RL[4], l);
      const float U = jr * fr, N = lane_val(t[1], l);
      float tsqr = lane_val(t[0], l);
      tsqr = tsqr >= DM_FLT_MIN ? tsqr : 0.f;   // (denormal = zero, as in unit_cost)
      const float T = DM_SQRT(tsqr);
      const bool bottom = (tsqr <= 0.f && N < 0.f) || (tsqr > 0.f && mu * N + T <= 0.f);
      const bool middle = tsqr > 0.f && N < mu * T && mu * N + T > 0.f;
      const float nmt = N - mu * T;
      float cost = 0.f;
      if (bottom) cost = 0.5f * Dk * jr * jr;
      else if (middle && kind == 2) cost = 0.5f * Dm * nmt * nmt;
      if (store) {
        const int r = rl_row(l), c = rl_con(l), k = l & 7;
        float f = 0.f;
        if (bottom) f = -Dk * jr;
        else if (middle) { const float fn = -Dm * nmt * mu; f = kind == 2 ? fn : -fn / T * U * fr; }
        s.frc[r] = f;
        if (kind >= 2) {
          if (kind == 2) s.lsign[r] = bottom ? 2.f : (middle ? 1.f : 0.f);
          if (bottom) s.cwd[6 * c + k] = Dk;
          else if (middle) {
            s.cwa[6 * c + k] = fr; s.cwb[6 * c + k] = U;
            if (kind == 2) {
              const float Tg = dm::fmaxf_(T, MJ_MINVAL), TTT = dm::fmaxf_(Tg * Tg * Tg, MJ_MINVAL);
              s.ccf[4 * c] = Dm;
              s.ccf[4 * c + 1] = -mu / Tg;               // c0
              s.ccf[4 * c + 2] = mu * mu - mu * N / Tg;  // c1
              s.ccf[4 * c + 3] = mu * N / TTT;           // c2
            }
          }
        }
      }
      return cost;
    });
    return out;
  };
  vfloat vJaR = vzero;   // the row lanes' Jaref (r8)

  // ---- unit evaluation: cost of limit row / contact u at the row values ja[]; STORE additionally writes the forces,
  // the zone (lsign of the contact's first row doubles as storage) and the Hessian weights
  auto unit_cost = [&](int u, const float* ja, const float* aref_or_null, bool store) -> float {
    // value of row r: ja[r] - aref[r] when aref is given (initial points), ja[r] otherwise
    auto val = [&](int r) { return aref_or_null ? ja[r] - aref_or_null[r] : ja[r]; };
    if (u < NL) {
      const float d = s.D[u];
      if (d == 0.f) { if (store) s.frc[u] = 0.f; return 0.f; }   // inside its range: the row is off (and was never written)
      const float j = val(u);
      const bool act = j < 0.f;
      if (store) s.frc[u] = act ? -d * j : 0.f;
      return act ? 0.5f * d * j * j : 0.f;
    }
    const int c = u - NL, r0 = m->con_adr[c], dim = m->con_dim[c];
    if (s.con_on[c] == 0.f) {
      if (store) {
        s.lsign[r0] = 0.f;
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < dim) s.frc[r0 + k] = 0.f;
      }
      return 0.f;
    }
    const float mu = m->con_friction[c][0] * mu_scale;
    // all (<= 6) rows of the contact with a fixed trip count, rows >= dim masked (they re-read row 0): the loads issue
    // back to back and jr / fr / U stay in registers -- the dim-bounded loops paid a dependent LDS round trip per row
    float U[6], jr[6], fr[6], Dk[6], tsqr = 0.f;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const int rk = r0 + (k < dim ? k : 0);
      jr[k] = val(rk);
      Dk[k] = s.D[rk];
      fr[k] = k == 0 ? mu : m->con_friction[c][k - 1];
      U[k] = jr[k] * fr[k];
    }
#pragma unroll
    for (int k = 1; k < 6; k++) { const float t2 = tsqr + U[k] * U[k]; tsqr = k < dim ? t2 : tsqr; }
    // A tangential part below the smallest NORMAL fp32 (a ball that has come to rest: |U_t|^2 decays through 1e-38) counts as
    // zero: the hardware's sqrt / rcp / rsq flush denormal arguments, so with `tsqr > 0` as the test T came out 0 for a
    // positive denormal tsqr and the middle zone's -fn / T * U[k] was inf * 0 -- NaN in every rollout of the closed loop once
    // the ball lay still on the floor (found when the row layout changed the rounding of the decay; the hazard was there before).
    tsqr = tsqr >= DM_FLT_MIN ? tsqr : 0.f;
    const float N = U[0], T = DM_SQRT(tsqr);
    const bool bottom = (tsqr <= 0.f && N < 0.f) || (tsqr > 0.f && mu * N + T <= 0.f);
    const bool middle = tsqr > 0.f && N < mu * T && mu * N + T > 0.f;
    const float Dm = Dk[0] / dm::fmaxf_(mu * mu * (1.f + mu * mu), MJ_MINVAL);
    float cost = 0.f;
    if (bottom) {
#pragma unroll
      for (int k = 0; k < 6; k++) { const float t2 = cost + 0.5f * Dk[k] * jr[k] * jr[k]; cost = k < dim ? t2 : cost; }
    } else if (middle) {
      const float nmt = N - mu * T;
      cost = 0.5f * Dm * nmt * nmt;
    }
    if (store) {
      s.lsign[r0] = bottom ? 2.f : (middle ? 1.f : 0.f);
      if (bottom) {
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < dim) { s.frc[r0 + k] = -Dk[k] * jr[k]; s.cwd[6 * c + k] = Dk[k]; }
      } else if (middle) {
        const float nmt = N - mu * T, fn = -Dm * nmt * mu;
        s.frc[r0] = fn;
#pragma unroll
        for (int k = 1; k < 6; k++) if (k < dim) s.frc[r0 + k] = -fn / T * U[k] * fr[k];
        const float Tg = dm::fmaxf_(T, MJ_MINVAL), TTT = dm::fmaxf_(Tg * Tg * Tg, MJ_MINVAL);
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < dim) { s.cwa[6 * c + k] = fr[k]; s.cwb[6 * c + k] = U[k]; }
        s.ccf[4 * c] = Dm;
        s.ccf[4 * c + 1] = -mu / Tg;             // c0
        s.ccf[4 * c + 2] = mu * mu - mu * N / Tg;  // c1
        s.ccf[4 * c + 3] = mu * N / TTT;         // c2
      } else {
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < dim) s.frc[r0 + k] = 0.f;
      }
    }
    return cost;
  };

  // ---- persistent registers of the dof lanes
  const vfloat vqfs = w.per_lane([&](int l) { return l < NV ? s.qfs[l] : 0.f; });
  const vfloat vqas = w.per_lane([&](int l) { return l < NV ? s.qas[l] : 0.f; });
  const vfloat vwarm = w.per_lane([&](int l) { return l < NV ? s.warm[l] : 0.f; });
  // M v for the dof lanes, v gathered from LDS (M is block diagonal; full-row fetches keep it simple)
  auto mul_m = [&](const float* vec) {
    return w.per_lane([&](int l) {
      if (l >= NV) return 0.f;
      float acc = 0.f;   // M is block diagonal: the columns of the dof's own kinematic tree (<= NBLK, fixed trip count)
      const int j0 = m->dof_blk0[l], j1 = m->dof_blk1[l];
#pragma unroll
      for (int t = 0; t < NBLK; t++) {
        const bool on = j0 + t < j1;
        const int j = on ? j0 + t : j0;
        const float t2 = acc + s.M[l * S + j] * vec[j];
        acc = on ? t2 : acc;
      }
      return acc;
    });
  };

  // ---- warm-start selection: cost at qacc_warmstart vs cost at qacc_smooth
  w.items(NV, [&](int i) { s.vec0[i] = s.warm[i]; s.vec1[i] = s.qas[i]; });
  vfloat jaW = vzero, jaS = vzero;   // (r8) J warm - aref, J qacc_smooth - aref of the lane's row
  if (r8) {
    row_products_r8(s.vec0, s.vec1, jaW, jaS);
    const vfloat ar = w.per_lane([&](int l) { return rl_kind(l) != 0 ? s.aref[rl_row(l)] : 0.f; });
    jaW = jaW - ar; jaS = jaS - ar;
  } else {
    row_products(s.vec0, s.Jaref, s.vec1, s.jv);     // J warm, J qacc_smooth (aref subtracted on the fly below)
  }
  const vfloat maW = mul_m(s.vec0), maS = mul_m(s.vec1);
  float cw, gw, cs, gs;
  {
    vfloat t[4];
    if (r8) {
      t[0] = unit_eval_r8(jaW, false);
      t[2] = unit_eval_r8(jaS, false);
    } else {
      t[0] = w.per_lane([&](int l) { return l < NU_ ? unit_cost(l, s.Jaref, s.aref, false) : 0.f; });
      t[2] = w.per_lane([&](int l) { return l < NU_ ? unit_cost(l, s.jv, s.aref, false) : 0.f; });
    }
    t[1] = (maW - vqfs) * (vwarm - vqas);
    t[3] = (maS - vqfs) * (vqas - vqas);
    float r[4];
    w.vsumN(t, r);
    cw = r[0]; gw = r[1]; cs = r[2]; gs = r[3];
  }
  const float cost_w = cw + 0.5f * gw, cost_s = cs + 0.5f * gs;
  const bool use_warm = cost_w < cost_s;
  vfloat vqacc = use_warm ? vwarm : vqas;
  vfloat vMa = use_warm ? maW : maS;
  if (r8) {
    vJaR = use_warm ? jaW : jaS;
    // the LDS copy (H's limit rows read it) and the forces of the limit rows that are OFF (D = 0: not in the list, never written
    // by the row lanes; J^T f reads the limit row of every dof)
    w.items(64 + NL, [&](int it) {
      if (it < 64) { if (rl_kind(it) != 0) s.Jaref[rl_row(it)] = lane_val(vJaR, it); }
      else if (s.D[it - 64] == 0.f) s.frc[it - 64] = 0.f;
    });
  } else {
    for_on_rows([&](int r) { s.Jaref[r] = (use_warm ? s.Jaref[r] : s.jv[r]) - s.aref[r]; });
  }
  float cost = use_warm ? cost_w : cost_s;
  float gauss = use_warm ? 0.5f * gw : 0.5f * gs;
  float prev_cost = INFINITY;
  const float scale = 1.f / (m->meaninertia * (float)(NV > 1 ? NV : 1));
  const bool rule_swap = m->ls_rule == DIAL_LS_SWAP;
  const int max_iter = DM_UNIFORM_I(m->iterations), max_ls = DM_UNIFORM_I(m->ls_iterations);
  const float tol = m->tolerance, ls_tol = m->ls_tolerance, meaninertia = m->meaninertia;

  DIAL_MARK(w, 14);
  int niter = 0;
  for (;;) {
    // ---- _update_constraint: zones, forces, Hessian weights (unit lanes); cost
    const vfloat ucost = r8 ? unit_eval_r8(vJaR, true) : w.per_lane([&](int l) { return l < NU_ ? unit_cost(l, s.Jaref, nullptr, true) : 0.f; });
    w.fence();
    DIAL_MARK(w, 12);
    // ---- J^T f per dof lane: own limit row + the contacts that move the dof
    const vfloat qfc = w.per_lane([&](int l) {
      if (l >= NV) return 0.f;
      const int lr = m->dof_limrow[l];
      float acc = lr >= 0 ? s.lsign[lr] * s.frc[lr] : 0.f;
      for (int idx = 0; idx < n_on; idx++) {          // wave-uniform loop over the contributing units
        const int u = (int)s.ulist[idx];
        if (u < NL) continue;
        const int c = u - NL, r0 = m->con_adr[c];
        if (s.lsign[r0] == 0.f) continue;             // top zone: no force (uniform)
        const int a = m->con_dofpos[c][l];            // where dof l sits in the contact's dof list, or 255
        if (a == 255) continue;
        const int nd = m->con_ndof[c], dim = m->con_dim[c];
        const float* J = s.Jc + m->con_joff[c] + a;
        for (int k = 0; k < dim; k++) acc += J[k * nd] * s.frc[r0 + k];
      }
      return acc;
    });
    const vfloat vgrad = vsel(isdof, vMa - vqfs - qfc, vzero);
    DIAL_MARK(w, 13);
    float gn;
    {
      vfloat t[3] = {ucost, (vMa - vqfs) * (vqacc - vqas), vgrad * vgrad};
      float r[3];
      w.vsumN(t, r);
      if (niter > 0) {
        gauss = 0.5f * r[1];
        prev_cost = cost;
        cost = r[0] + gauss;
      }
      gn = r[2];
    }
    DIAL_MARK(w, 4);
    bool done;
    if (max_iter != 1) {
      const float improvement = scale * (prev_cost - cost), gradient = scale * DM_SQRT(gn);
      done = niter >= max_iter || improvement < tol || gradient < tol;
    } else {
      done = niter >= 1;
    }
    if (done) {
      w.items(NV, [&](int i) { s.qfc[i] = lane_val(qfc, i); });   // qfrc_constraint of the final point (Euler damping)
      break;
    }

    // ---- H = M + limit rows + per-contact blocks
    w.items(NV * S / 4, [&](int e) {   // 16-byte copies
      const float a = s.M[4 * e], b = s.M[4 * e + 1], c2 = s.M[4 * e + 2], d2 = s.M[4 * e + 3];
      s.H[4 * e] = a; s.H[4 * e + 1] = b; s.H[4 * e + 2] = c2; s.H[4 * e + 3] = d2;
    });
    w.items(NL, [&](int r) {
      if (s.D[r] > 0.f && s.Jaref[r] < 0.f) { const int i = m->jnt_dofadr[m->lim_jnt[r]]; s.H[i * S + i] += s.D[r]; }
    });
    DIAL_MARK(w, 24);
    for (int idx = 0; idx < n_on; idx++) {           // the contributing units (compacted once per solve), not all NC contacts
      const int u = DM_UNIFORM_I((int)s.ulist[idx]);
      if (u < NL) continue;                          // limit row: added above (wave-uniform)
      const int c = u - NL;
      const float zone = s.lsign[m->con_adr[c]];
      if (zone == 0.f) continue;                    // top zone: no curvature
      const int nd = m->con_ndof[c], dim = m->con_dim[c];
      const float* J = s.Jc + m->con_joff[c];
      w.items(NCD * (NCD + 1) / 2, [&](int it) {
        const int a = m->pair_a[it], b = m->pair_b[it];   // a >= b
        if (a >= nd) return;
        float acc;
        if (zone == 2.f) {                          // bottom zone: plain quadratic rows
          acc = 0.f;
#pragma unroll
          for (int k = 0; k < 6; k++) {
            const int kk = k < dim ? k : 0;
            const float t2 = acc + (J[kk * nd + a] * s.cwd[6 * c + kk]) * J[kk * nd + b];
            acc = k < dim ? t2 : acc;
          }
        } else {                                    // middle zone: cone Hessian
          const float Dm = s.ccf[4 * c], c0 = s.ccf[4 * c + 1], c1 = s.ccf[4 * c + 2], c2 = s.ccf[4 * c + 3];
          const float x0 = s.cwa[6 * c] * J[a], y0 = s.cwa[6 * c] * J[b];
          float ux = 0.f, uy = 0.f, xty = 0.f;
#pragma unroll
          for (int k = 1; k < 6; k++) {
            const int kk = k < dim ? k : 0;
            const float xk = s.cwa[6 * c + kk] * J[kk * nd + a], yk = s.cwa[6 * c + kk] * J[kk * nd + b], uk = s.cwb[6 * c + kk];
            const float t_ux = ux + uk * xk, t_uy = uy + uk * yk, t_xy = xty + xk * yk;
            ux = k < dim ? t_ux : ux;
            uy = k < dim ? t_uy : uy;
            xty = k < dim ? t_xy : xty;
          }
          acc = Dm * (x0 * y0 + c0 * (x0 * uy + ux * y0) + c2 * (ux * uy) + c1 * xty);
        }
        const int i = m->con_dof[c][a], j = m->con_dof[c][b];
        s.H[i * S + j] += acc;
        if (a != b) s.H[j * S + i] += acc;
      });
    }
    DIAL_MARK(w, 5);
    const vfloat vsearch = vzero - reg_chol<D, TopoDense>(w, m, s.H, vgrad, s.H);
    DIAL_MARK(w, 6);

    // ---- solver._linesearch
    w.begin_region();
    w.items(NV, [&](int i) { s.vec0[i] = lane_val(vsearch, i); });
    vfloat vjvR = vzero, vdummy = vzero;   // (r8) J search of the lane's row
    if (r8) row_products_r8(s.vec0, nullptr, vjvR, vdummy);
    else row_products(s.vec0, s.jv, nullptr, nullptr);
    const vfloat vmv = mul_m(s.vec0);
    float sn2, s1, s2;
    {
      vfloat t[3] = {vsearch * vsearch, vsearch * vMa - vsearch * vqfs, vsearch * vmv};
      float r[3];
      w.vsumN(t, r);
      sn2 = r[0]; s1 = r[1]; s2 = r[2];
    }
    const float smag = DM_SQRT(sn2) * meaninertia * (float)(NV > 1 ? NV : 1);
    const float gtol = tol * ls_tol * smag;
    const float qg0 = gauss, qg1 = s1, qg2 = 0.5f * s2;
    // per-unit line-search registers: limit row: (Jaref, jv | q0 q1 q2); contact: (u0 v0 uu uv vv Dm mu | quad_c).
    // fast layout (<= 16 contributing units): lane (g, j) = (lane >> 4, lane & 15), g < 3, holds unit ulist[j] -- the
    // three 16-lane groups evaluate the three trial points of an iteration in one pass; otherwise lane u holds unit u
    const bool fast = n_on <= 16;
    auto unit_of_lane = [&](int l) -> int {
      if (fast) { const int j = l & 15; return (l < 48 && j < n_on) ? (int)s.ulist[j] : -1; }
      return l < NU_ ? l : -1;
    };
    vfloat L[10];
    if (r8) {
      // the units' coefficients from their row lanes: partial terms per row, sums over the group of eight, then every line-search
      // lane (g, j) fetches unit j's ten words from lane 8 j (ds_bpermute)
      vfloat P[8];   // q0 q1 q2 | uu uv vv | u0 v0
      w.per_lane_n(P, [&](int l, float* o) {
        for (int k = 0; k < 8; k++) o[k] = 0.f;
        const int kind = rl_kind(l);
        if (kind == 0) return;
        const float ja = lane_val(vJaR, l), jv = lane_val(vjvR, l), d = lane_val(RL[2], l), f = lane_val(RL[1], l);
        o[0] = 0.5f * ja * ja * d; o[1] = jv * ja * d; o[2] = 0.5f * jv * jv * d;
        if (kind == 3) { const float a = ja * f, b = jv * f; o[3] = a * a; o[4] = a * b; o[5] = b * b; }
        if (kind == 1) { o[6] = ja; o[7] = jv; }                  // limit row: (Jaref, jv)
        if (kind == 2) { o[6] = ja * f; o[7] = jv * f; }          // contact: (Jaref_0 mu, jv_0 mu)   (f = mu on row 0)
      });
      w.seg8_sumN(P);
      const vfloat src[10] = {P[6], P[7], P[3], P[4], P[5], RL[4], RL[3], P[0], P[1], P[2]};
#pragma unroll
      for (int k = 0; k < 10; k++) {
        const vfloat g = w.gather64(src[k], [&](int l) { return 8 * (l & 15); });
        L[k] = w.per_lane([&](int l) { return (l < 48 && (l & 15) < n_on) ? lane_val(g, l) : 0.f; });
      }
      // (s.jv keeps the rows' J search for nothing but the profile build's dumps; the Jaref update below runs in the row lanes)
    } else
    w.per_lane_n(L, [&](int l, float* o) {
      for (int k = 0; k < 10; k++) o[k] = 0.f;
      const int u = unit_of_lane(l);
      if (u < 0) return;
      if (u < NL) {
        const float ja = s.Jaref[u], jv = s.jv[u], d = s.D[u];
        if (d == 0.f) return;
        o[0] = ja; o[1] = jv;
        o[7] = 0.5f * ja * ja * d; o[8] = jv * ja * d; o[9] = 0.5f * jv * jv * d;
        return;
      }
      const int c = u - NL;
      if (s.con_on[c] == 0.f) return;
      const int r0 = m->con_adr[c], dim = m->con_dim[c];
      const float mu = m->con_friction[c][0] * mu_scale;
      float uu = 0.f, uv = 0.f, vv = 0.f, q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
      for (int k = 0; k < 6; k++) {                 // fixed trip count, rows >= dim masked (see unit_cost)
        const bool on = k < dim;
        const int rk = r0 + (on ? k : 0);
        const float ja = s.Jaref[rk], jv = s.jv[rk], d = s.D[rk];
        const float t0 = q0 + 0.5f * ja * ja * d, t1 = q1 + jv * ja * d, t2 = q2 + 0.5f * jv * jv * d;
        q0 = on ? t0 : q0; q1 = on ? t1 : q1; q2 = on ? t2 : q2;
        if (k > 0) {
          const float f = m->con_friction[c][k - 1], a = ja * f, b = jv * f;
          const float tuu = uu + a * a, tuv = uv + a * b, tvv = vv + b * b;
          uu = on ? tuu : uu; uv = on ? tuv : uv; vv = on ? tvv : vv;
        }
      }
      o[0] = s.Jaref[r0] * mu; o[1] = s.jv[r0] * mu; o[2] = uu; o[3] = uv; o[4] = vv;
      o[5] = s.D[r0] / dm::fmaxf_(mu * mu * (1.f + mu * mu), MJ_MINVAL); o[6] = mu;
      o[7] = q0; o[8] = q1; o[9] = q2;
    });
    DIAL_MARK(w, 25);
    // is the lane's unit a limit row?  (mu = L[6] is > 0 exactly for contacts)
    // the six sums of one unit at alpha: quadratic part (q0 q1 q2) + cone part (cost, slope, curvature)
    // Straight-line code for every lane: a limit row is the degenerate contact mu = 0, uu = uv = vv = 0 (tsqr = 0, so the
    // "bottom" test reduces to Jaref + alpha jv < 0 and "middle" is never true), an idle lane holds zeros everywhere.
    // Three divergent branches per evaluation (limit / bottom / middle lanes ran one after the other) become selects.
    auto unit_terms = [&](int l, float alpha, float* o) {
      const float mu = lane_val(L[6], l);
      const float u0 = lane_val(L[0], l), v0 = lane_val(L[1], l), uu = lane_val(L[2], l), uv = lane_val(L[3], l), vv = lane_val(L[4], l);
      const float dmc = lane_val(L[5], l);
      const float n = u0 + alpha * v0;
      const float tsqr_raw = uu + alpha * (2.f * uv + alpha * vv);
      const float tsqr = tsqr_raw >= DM_FLT_MIN ? tsqr_raw : 0.f;   // (denormal = zero, as in unit_cost)
      // T and 1 / T from one v_rsq and a Newton correction each (both then within an ulp of sqrt / divide: the zone
      // tests below sit on the oracle's decisions) instead of a square root and three divisions
      const float rt0 = tsqr > 0.f ? fast_rsqrt(tsqr) : 0.f;
      const float tt0 = tsqr * rt0;
      const float tt = DM_FMA(DM_FMA(-tt0, tt0, tsqr), 0.5f * rt0, tt0);
      const float rt = DM_FMA(rt0, DM_FMA(-tt, rt0, 1.f), rt0);
      const bool bottom = (tsqr <= 0.f && n < 0.f) || (tsqr > 0.f && mu * n + tt <= 0.f);
      const bool middle = tsqr > 0.f && n < mu * tt && mu * n + tt > 0.f;
      o[0] = bottom ? lane_val(L[7], l) : 0.f;
      o[1] = bottom ? lane_val(L[8], l) : 0.f;
      o[2] = bottom ? lane_val(L[9], l) : 0.f;
      const float w1 = uv + alpha * vv;
      const float n1 = v0, t1 = w1 * rt, t2 = vv * rt - w1 * t1 * (rt * rt);
      const float nmt = n - mu * tt, g = n1 - mu * t1;
      o[3] = middle ? 0.5f * dmc * nmt * nmt : 0.f;
      o[4] = middle ? dmc * nmt * g : 0.f;
      o[5] = middle ? dmc * (g * g - nmt * mu * t2) : 0.f;
    };
    // cost / slope / curvature of a point from its six sums, packed into the four words of ls_bracket.h
    auto finish = [&](float alpha, const float* r, float* o) {
      const float q0 = r[0] + qg0, q1 = r[1] + qg1, q2 = r[2] + qg2;
      const float cost = alpha * alpha * q2 + alpha * q1 + q0 + r[3];
      const float d0 = DM_FMA(2.f * alpha, q2, q1) + r[4];   // single-rounding slope, see rollout_body.h
      float d1 = 2.f * q2 + r[5];
      if (d1 == 0.f) d1 = MJ_MINVAL;
      ls_pack(alpha, cost, d0, d1, o[0], o[1], o[2], o[3]);
    };
    auto ls_point = [&](float alpha) {
      vfloat t[6];
      w.per_lane_n(t, [&](int l, float* o) {
        unit_terms(l, alpha, o);
        if (fast && l >= 16) for (int k = 0; k < 6; k++) o[k] = 0.f;   // the groups hold copies: count one
      });
      float r[6], o[4];
      w.vsumN(t, r);
      finish(alpha, r, o);
      LsPt p;
      p.alpha = fbits(o[0]); p.nalpha = fbits(o[1]); p.cost = fbits(o[2]); p.d0 = fbits(o[3]);
      return p;
    };
    // res: the packed words (alpha, nalpha, cost key, d0 key) of the three points of an iteration, each in the lanes of its
    // 16-lane group; ls_update_lazy broadcasts the slope keys and fetches the rest of the winners only (ls_bracket.h)
    vfloat res[4];
    auto ls_eval3 = [&](float a0, float a1, float a2) {
      if (!fast) {   // > 16 contributing units: three full-wave passes, results parked in the groups' lanes
        const LsPt P0 = ls_point(a0), P1 = ls_point(a1), P2 = ls_point(a2);
        w.per_lane_n(res, [&](int l, float* o) {
          const LsPt& P = l < 16 ? P0 : (l < 32 ? P1 : P2);
          o[0] = bitsf(P.alpha); o[1] = bitsf(P.nalpha); o[2] = bitsf(P.cost); o[3] = bitsf(P.d0);
        });
        return;
      }
      vfloat t[6];
      // (two opaque v_cndmask selects: left to itself the compiler stores the three trial steps to a scratch array and
      // loads a[lane >> 4] back -- a memory round trip in every line-search iteration)
      const auto group_alpha = [a0, a1, a2](int l) {
        float a = a2;
        DM_OPAQUE(a);
        if (l < 32) a = a1;
        DM_OPAQUE(a);
        if (l < 16) a = a0;
        return a;
      };
      w.per_lane_n(t, [&](int l, float* o) { unit_terms(l, group_alpha(l), o); });
      w.row16_sumN(t);
      // the group's point finished lane-wise (every lane of a group holds the six sums)
      w.per_lane_n(res, [&](int l, float* o) {
        const float r6[6] = {lane_val(t[0], l), lane_val(t[1], l), lane_val(t[2], l), lane_val(t[3], l), lane_val(t[4], l), lane_val(t[5], l)};
        finish(group_alpha(l), r6, o);
      });
    };
    const LsPt p0 = ls_point(0.f);
    LsPt lo, hi;
    ls_open(p0, ls_point(bitsf(p0.nalpha)), lo, hi);
    const int kg = DM_UNIFORM_I(fkey(gtol)), kng = DM_UNIFORM_I(fkey(-gtol));
    DIAL_MARK(w, 26);
    const LsGate gate = ls_gate(kg, kng);   // (one scalar compare + branch per loop condition, see solver_reg.h)
    int ls_iter = 0;
    while (ls_iter < max_ls) {
      DM_NOFOLD();
      if (ls_converged_lo(lo, gate)) break;
      DM_NOFOLD();
      if (ls_converged_hi(hi, gate)) break;
      ls_eval3(bitsf(lo.nalpha), bitsf(hi.nalpha), 0.5f * (bitsf(lo.alpha) + bitsf(hi.alpha)));   // groups: lo_next, hi_next, mid
      const bool swap = ls_update_lazy<!D::gen>(rule_swap, lo, hi, fbits(bcast(res[3], 0)), fbits(bcast(res[3], 16)), fbits(bcast(res[3], 32)), 0, 16, 32,
                                                [&](int word, int lane) { return fbits(bcast(res[word], lane)); });
      ls_iter++;
      if (!swap) break;
    }
    float alpha;
    const bool improved = ls_result(p0, lo, hi, alpha);
    if (improved) {
      vqacc = vqacc + vsearch * alpha;
      vMa = vMa + vmv * alpha;
      if (r8) {
        vJaR = vJaR + vjvR * alpha;
        w.items(64, [&](int l) { if (rl_kind(l) != 0) s.Jaref[rl_row(l)] = lane_val(vJaR, l); });
      } else {
        for_on_rows([&](int r) { s.Jaref[r] += s.jv[r] * alpha; });
      }
    }
    niter++;
    w.work++;
#ifdef DIAL_PROFILE
    if (w.lane == 0 && w.acc) { w.acc[30] += ls_iter; w.acc[31] += 1; w.acc[29] += fast ? 1 : 0; }
#endif
    DIAL_MARK(w, 7);
  }
#ifdef DIAL_PROFILE
  if (w.lane == 0 && w.acc) { w.acc[28] += 1; w.acc[27] += n_on; }
#endif
  w.items(NV, [&](int i) {
    const float q = lane_val(vqacc, i);
    s.qacc[i] = q;
    s.warm[i] = q;
  });
  DIAL_MARK(w, 8);
}

}  // namespace dial

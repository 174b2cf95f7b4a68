// solver_reg2.h -- the register-resident Newton solver of solver_reg.h on 32 LANES, for the kernel that runs TWO samples per
// wavefront (wave.h: WaveH).  Same iterates, same arithmetic in the same order -- the one-sample solver is the reference the
// emulator and the GPU tests compare against bit for bit -- on a layout that never leaves the half:
//
//   lanes [0, NV)      dof i: row i of M in R[NV]; its joint-limit row, if it has one               ("limit slot")
//   lanes [0, 4 NC)    ALSO pyramid edge e of contact c (lane 4 c + e), in registers of their own   ("contact slot"):
//                      the row is kept COMPACT -- the six trunk columns and the three columns of the contact's own leg
//                      (a foot contact's Jacobian has no other non-zeros: solver_reg.h multiplies the rest by exact zeros)
//
// Broadcasts are VGPR operands: dup_rows (one v_permlane16_swap) makes the half's even / odd row visible in both rows, DPP
// row_newbcast picks the lane -- where solver_reg.h pays v_readlane + the SGPR-operand hazard per pivot, this layout pays one
// swap per VECTOR and a DPP mov per use, and serves two samples.  The leg columns of a contact row are a lane-indexed gather
// (ds_bpermute).  Wave-uniform scalars of solver_reg.h (costs, step sizes, the bracket of the line search, loop exits) are
// per-lane values that agree within the half; where the two samples disagree (one has converged, one re-uses its factor) the
// EXEC mask serialises the two paths like any divergent code.
// The line search evaluates TWO trial points per pass in two 16-lane groups (solver_reg.h: three in three groups): an iteration
// is one pass for (lo_next, hi_next) and one for the mid-point; the rows keep their positions inside the 16-lane groups, so the
// sums associate identically.  The bracket update runs lane-wise on the integer keys of ls_bracket.h (ls_update).
#pragma once
// (included from rollout_body.h after solver_reg.h)

#ifndef DIAL_PAIR_HBATCH
#define DIAL_PAIR_HBATCH 3   // work-list passes of 32 records whose LDS fetches are in flight together (H assembly below)
#endif

namespace dial {

template <class W, class M>
DIAL_DEV void solver_reg2(W& w, const M* m, const Ws& s) {
  constexpr int NV = M::D::NV, NC = M::D::NC, NL = M::D::NL;
  static_assert(W::half2, "32-lane layout: the half-wave execution model");
  static_assert(NV <= 32 && 4 * NC <= 16 && NV == 6 + 3 * NC, "free trunk + one three-dof leg per contact; contact rows in one DPP row");
#ifdef DIAL_PAIR_SOLVER_SCOPE
  DIAL_LANE_SCOPE(w);
#endif
  w.begin_region();
  const vbool isdof = w.lane_lt(NV), iscon = w.lane_lt(4 * NC);
  const vfloat vzero = vsplat(0.f);
  const auto lim_of = [&](int l) -> int { return l < NV ? m->dof_limrow[l] : -1; };

  // ---- persistent registers
  vfloat R[NV];      // dof lane i: row i of M
  vfloat RJt[6];     // contact lane r = 4 c + e: the trunk columns of J_r = Jn +- mu Jt ...
  vfloat RJl[3];     // ... and the columns of leg c's dofs 6 + 3 c + q
  {
    // (offsets from ONE base pointer rather than a select of pointers: keeps the accesses in the LDS address space; idle lanes
    //  re-read a word that holds 0 -- lsign of a contact row)
    constexpr int S = M::D::S, T = M::D::T;
    const int ojc = (int)(s.Jc - s.M), ozero = (int)(s.lsign - s.M) + NL;
#pragma unroll
    for (int j = 0; j < NV; j++) R[j] = w.per_lane([&](int l) { return s.M[l < NV ? l * S + j : ozero]; });
#pragma unroll
    for (int j = 0; j < 6; j++) RJt[j] = w.per_lane([&](int l) { return s.M[l < 4 * NC ? ojc + j * T + l : ozero]; });
#pragma unroll
    for (int q = 0; q < 3; q++) RJl[q] = w.per_lane([&](int l) { return s.M[l < 4 * NC ? ojc + (6 + 3 * (l >> 2) + q) * T + l : ozero]; });
  }
  const vfloat vD = w.per_lane([&](int l) { const int r = lim_of(l); return r >= 0 ? s.D[r] : 0.f; });
  const vfloat varef = w.per_lane([&](int l) { const int r = lim_of(l); return r >= 0 ? s.aref[r] : 0.f; });
  const vfloat vls = w.per_lane([&](int l) { const int r = lim_of(l); return r >= 0 ? s.lsign[r] : 0.f; });
  const vfloat cD = w.per_lane([&](int l) { return l < 4 * NC ? s.D[NL + l] : 0.f; });
  const vfloat caref = w.per_lane([&](int l) { return l < 4 * NC ? s.aref[NL + l] : 0.f; });
  const vfloat vqfs = w.per_lane([&](int l) { return l < NV ? s.qfs[l] : 0.f; });
  const vfloat vqas = w.per_lane([&](int l) { return l < NV ? s.qas[l] : 0.f; });
  const vfloat vwarm = w.per_lane([&](int l) { return l < NV ? s.warm[l] : 0.f; });

  // pM[dof lane i] = (M v)_i, pJ[contact lane r] = (J v)_r.  Three partial sums each, column j in chain j % 3, in ascending j:
  // exactly the chains of solver_reg.h's sweep minus its exact-zero terms.
  auto dotR = [&](const vfloat& v, vfloat& pM, vfloat& pJ) {
    vfloat vX, vY;
    w.dup_rows(v, vX, vY);
    vfloat accM[3] = {vzero, vzero, vzero}, accJ[3] = {vzero, vzero, vzero};
    static_for<0, NV>([&](auto JJ) {
      constexpr int j = JJ;
      accM[j % 3] = w.template fma_pick<j>(accM[j % 3], vX, vY, R[j]);
      if constexpr (j < 6) accJ[j % 3] = w.template fma_pick<j>(accJ[j % 3], vX, vY, RJt[j]);
    });
    static_for<0, 3>([&](auto QQ) {
      constexpr int q = QQ;
      const vfloat vl = w.gather(v, [&](int l) { return l < 4 * NC ? 6 + 3 * (l >> 2) + q : 0; });
      accJ[q] = accJ[q] + RJl[q] * vl;
    });
    pM = (accM[0] + accM[1]) + accM[2];
    pJ = (accJ[0] + accJ[1]) + accJ[2];
  };
  auto cost_terms = [&](const vfloat& D_, const vfloat& ja) { return vsel(vlt0(ja), D_ * ja * ja, vzero); };
  // the sum over all rows as solver_reg.h's 64-lane reduction forms it: (contact rows) + (limit rows)
  auto sum_rows = [&](float rc, float rl) { return rc + rl; };

  // ---- warm-start selection (solver.solve): cost at qacc_warmstart vs cost at qacc_smooth
  vfloat pW, jW, pS, jS;
  dotR(vwarm, pW, jW);
  dotR(vqas, pS, jS);
  const vfloat jaW = vls * vwarm - varef, jaS = vls * vqas - varef;          // limit slot
  const vfloat cjaW = vsel(iscon, jW, vzero) - caref, cjaS = vsel(iscon, jS, vzero) - caref;   // contact slot
  const vfloat maW = vsel(isdof, pW, vzero), maS = vsel(isdof, pS, vzero);
  float cw, gw, cs, gs;
  {
    vfloat t[6] = {cost_terms(vD, jaW), cost_terms(cD, cjaW), (maW - vqfs) * (vwarm - vqas),
                   cost_terms(vD, jaS), cost_terms(cD, cjaS), (maS - vqfs) * (vqas - vqas)};
    float r[6];
    w.vsumN(t, r);
    cw = sum_rows(r[1], r[0]); gw = r[2]; cs = sum_rows(r[4], r[3]); gs = r[5];
  }
  const float cost_w = 0.5f * cw + 0.5f * gw, cost_s = 0.5f * cs + 0.5f * gs;
  const bool use_warm = cost_w < cost_s;
  vfloat vqacc = use_warm ? vwarm : vqas;
  vfloat vMa = use_warm ? maW : maS;
  vfloat vJa = use_warm ? jaW : jaS;      // limit slot
  vfloat cJa = use_warm ? cjaW : cjaS;    // contact slot
  float cost = use_warm ? cost_w : cost_s;
  float gauss = use_warm ? 0.5f * gw : 0.5f * gs;
  float prev_cost = INFINITY;
  const float scale = 1.f / (m->meaninertia * (float)(NV > 1 ? NV : 1));
  const bool rule_swap = m->ls_rule == DIAL_LS_SWAP;
  const int max_iter = DM_UNIFORM_I(m->iterations), max_ls = DM_UNIFORM_I(m->ls_iterations);
  const float tol = m->tolerance, ls_tol = m->ls_tolerance, meaninertia = m->meaninertia;

  int niter = 0;
  unsigned long long act_prev_l = 0, act_prev_c = 0;
  vfloat h_dinv = vzero;
  bool h_valid = false;
  for (;;) {
    // ---- _update_constraint: forces; _update_gradient: grad = Ma - qfrc_smooth - J^T f
    const vbool act = vlt0(vJa), cact = vlt0(cJa);
    const vfloat vf = vsel(act, vD * (vzero - vJa), vzero);
    const vfloat cf = vsel(cact, cD * (vzero - cJa), vzero);
    vfloat qfc = vls * vf;  // limit row of the own dof
    {
      vfloat fX, fY;
      w.dup_rows(cf, fX, fY);   // (the contact rows sit in the half's even row)
      static_for<0, NC>([&](auto Cc) {   // J^T f from the dof-major pyramid rows: one 16-byte fetch per contact
        constexpr int c = Cc;
        vfloat jr[4];
        w.per_lane4([&](int l) { return s.Jc + (l < NV ? l : 0) * M::D::T + 4 * c; }, jr[0], jr[1], jr[2], jr[3]);
        qfc = qfc + ((jr[0] * w.template row_bcast<4 * c>(fX) + jr[1] * w.template row_bcast<4 * c + 1>(fX)) +
                     (jr[2] * w.template row_bcast<4 * c + 2>(fX) + jr[3] * w.template row_bcast<4 * c + 3>(fX)));
      });
    }
    const vfloat vgrad = vsel(isdof, vMa - vqfs - qfc, vzero);
    float gn = 0.f;
    if (niter > 0) {
      vfloat t[4] = {cost_terms(vD, vJa), cost_terms(cD, cJa), (vMa - vqfs) * (vqacc - vqas), vgrad * vgrad};
      float r[4];
      w.vsumN(t, r);
      gauss = 0.5f * r[2];
      prev_cost = cost;
      cost = 0.5f * sum_rows(r[1], r[0]) + gauss;
      gn = r[3];
    } else if (max_iter != 1) {
      gn = w.vsum(vgrad * vgrad);
    }
    DIAL_MARK(w, 4);
    bool done;
    if (max_iter != 1) {
      const float improvement = scale * (prev_cost - cost), gradient = scale * DM_SQRT(gn);
      done = niter >= max_iter || improvement < tol || gradient < tol;
    } else {
      done = niter >= 1;
    }
    if (done) break;

    // ---- Newton direction: H = M + J^T diag(D*active) J in LDS (lane per entry), Cholesky in registers
    const vfloat vwgt = vsel(act, vD, vzero), cwgt = vsel(cact, cD, vzero);
#ifdef DIAL_NO_FACTOR_REUSE
    const bool reuse = false;
#else
    const unsigned long long act_l = w.mask(vlt0(vzero - vwgt)), act_c = w.mask(vlt0(vzero - cwgt));   // rows that carry weight
    const bool reuse = h_valid && act_l == act_prev_l && act_c == act_prev_c;
    act_prev_l = act_l;
    act_prev_c = act_c;
    h_valid = true;
#endif
    vfloat vsearch;
    if (reuse) {
      vsearch = vzero - reg_chol<typename M::D, typename M::D::Topo, true>(w, m, s.H, vgrad, s.H, &h_dinv);
      DIAL_MARK(w, 5);
    } else {
    {
      // row weights: limit rows at frc[0, NL), contact rows 16-byte aligned at frc[NLP, NLP + 4 NC), then a zero word
      constexpr int NLP = M::D::NLP, NP = M::D::NHI / 32;   // passes of 32 work-list records (solver_reg.h: of 64)
      w.items(32, [&](int l) {
        const int r = lim_of(l);
        if (r >= 0) s.frc[r] = lane_val(vwgt, l);
        if (l < 4 * NC) s.frc[NLP + l] = lane_val(cwgt, l);
        if (l == 31) s.frc[NLP + 4 * NC] = 0.f;
      });
      // the work list of solver_reg.h, record p * 32 + lane in pass p: the same items, the same quads.  Three passes at a time:
      // every lane accumulates its item of every pass of the batch (the LDS latencies of the passes overlap), partial sums of
      // split entries are combined inside quads, then one phase writes the batch's entries.
      constexpr int NB = DIAL_PAIR_HBATCH;
      static_assert(NP % NB == 0, "work-list capacity: a multiple of 96 records");
      static_for<0, NP / NB>([&](auto BATCH) {
        constexpr int p0 = BATCH * NB;
        vfloat part[NB], tot[NB];
        static_for<0, 4>([&](auto Qq) {
          constexpr int q = Qq;
          bool any = false;
          static_for<0, NB>([&](auto PP) { any = any || q < m->hpass_n[(p0 + PP) / 2]; });
          if (any) {   // wave-uniform: round q of the contact lists (Go2: one round)
            vfloat ji[NB][4], jj[NB][4], dd[NB][4];
            static_for<0, NB>([&](auto PP) {
              constexpr int pp = PP, pass = p0 + PP;
              const auto c4 = [&](int l) { return (int)((m->hrec[pass * 32 + l][0] >> (20 + 3 * q)) & 7u) * 4; };
              w.per_lane4([&](int l) { return s.Jc + (m->hrec[pass * 32 + l][0] & 1023u) + c4(l); }, ji[pp][0], ji[pp][1], ji[pp][2], ji[pp][3]);
              w.per_lane4([&](int l) { return s.Jc + ((m->hrec[pass * 32 + l][0] >> 10) & 1023u) + c4(l); }, jj[pp][0], jj[pp][1], jj[pp][2], jj[pp][3]);
              w.per_lane4([&](int l) { return s.frc + NLP + c4(l); }, dd[pp][0], dd[pp][1], dd[pp][2], dd[pp][3]);
            });
            static_for<0, NB>([&](auto PP) {
              constexpr int pp = PP, pass = p0 + PP;
              const vfloat t = ((ji[pp][0] * dd[pp][0]) * jj[pp][0] + (ji[pp][1] * dd[pp][1]) * jj[pp][1]) +
                               ((ji[pp][2] * dd[pp][2]) * jj[pp][2] + (ji[pp][3] * dd[pp][3]) * jj[pp][3]);
              const vfloat tq = w.per_lane([&](int l) { return q < (int)(m->hrec[pass * 32 + l][1] >> 29) ? lane_val(t, l) : 0.f; });
              if constexpr (q == 0) part[pp] = tq; else part[pp] = part[pp] + tq;
            });
          }
        });
        static_for<0, NB>([&](auto PP) {
          constexpr int pp = PP, pass = p0 + PP;
          const vfloat pair = part[pp] + w.quad_xor1(part[pp]);
          const vfloat quad = pair + w.quad_xor2(pair);
          tot[pp] = w.per_lane([&](int l) {
            const float t1 = lane_val(part[pp], l), t2 = lane_val(pair, l), t4 = lane_val(quad, l);
            const uint32_t pc = (m->hrec[pass * 32 + l][1] >> 26) & 3u;
            return pc == 0 ? t1 : (pc == 1 ? t2 : t4);
          });
        });
        w.items(32, [&](int l) {
          static_for<0, NB>([&](auto PP) {
            constexpr int pp = PP, pass = p0 + PP;
            const uint32_t w1 = m->hrec[pass * 32 + l][1];
            const float v = (s.M[w1 & 1023u] + lane_val(tot[pp], l)) + s.frc[(w1 >> 20) & 63u];
            if ((w1 >> 28) & 1u) {
              s.H[w1 & 1023u] = v;
              s.H[(w1 >> 10) & 1023u] = v;
            }
          });
        });
      });
    }
    DIAL_MARK(w, 5);
    vsearch = vzero - reg_chol<typename M::D>(w, m, s.H, vgrad, s.H, &h_dinv);
    }
    DIAL_MARK(w, 6);

    // ---- solver._linesearch
    w.begin_region();
    vfloat pv, cjv;
    dotR(vsearch, pv, cjv);
    const vfloat vmv = vsel(isdof, pv, vzero);
    const vfloat vjv = vls * vsearch;          // limit slot
    cjv = vsel(iscon, cjv, vzero);             // contact slot
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
    // Line-search layout: lane (g, l) = (lane >> 4, lane & 15) owns rows l, l + 16 for the trial point of group g.
    // Re-layout through LDS (rows are indexed by r there).
    w.items(32, [&](int l) {
      const int r = lim_of(l);
      if (r >= 0) { s.Jaref[r] = lane_val(vJa, l); s.jv[r] = lane_val(vjv, l); }
      if (l < 4 * NC) { s.Jaref[NL + l] = lane_val(cJa, l); s.jv[NL + l] = lane_val(cjv, l); }
    });
    constexpr int NE = M::D::NE, RPL = (NE + 15) / 16;   // rows per line-search lane (Go2: 2)
    const vbool g0 = w.lane_lt(16);
    vfloat lJa[RPL], ljv[RPL], Q0[RPL], Q1[RPL], Q2[RPL];
#pragma unroll
    for (int q = 0; q < RPL; q++) {
      lJa[q] = w.per_lane([&](int l) { const int r = (l & 15) + 16 * q; return r < NE ? s.Jaref[r] : 0.f; });
      ljv[q] = w.per_lane([&](int l) { const int r = (l & 15) + 16 * q; return r < NE ? s.jv[r] : 0.f; });
      const vfloat lD = w.per_lane([&](int l) { const int r = (l & 15) + 16 * q; return r < NE ? s.D[r] : 0.f; });
      const vfloat dja = lD * lJa[q], djv = lD * ljv[q];
      Q0[q] = (lJa[q] * 0.5f) * dja;
      Q1[q] = ljv[q] * dja;
      Q2[q] = (ljv[q] * 0.5f) * djv;
    }
    // One pass evaluates up to THREE trial points: every lane the point of its group (`a_ab`: group 0 lo_next, group 1 hi_next) AND
    // the mid-point (`a_c`, by both groups redundantly) -- per row and point one FMA, one compare and three selected adds into the
    // sums (0.5 D Jaref^2, D jv Jaref, 0.5 D jv^2 over the active rows), six interleaved 16-lane DPP reductions, and every lane
    // finishes both of its points: cost, slope, the point's own Newton step and the integer keys of ls_bracket.h.  Each point's
    // sums associate exactly as in solver_reg.h (rows l and l + 16 in lane l of a 16-lane group).
    auto finish = [&](const vfloat& va, vfloat s0, vfloat s1v, vfloat s2v, vfloat (&pk)[4]) {
      const vfloat q0 = s0 + vsplat(qg0), q1 = s1v + vsplat(qg1), q2 = s2v + vsplat(qg2);
      const vfloat vcost = (va * va) * q2 + va * q1 + q0;
      const vfloat vd0 = vfma(va * 2.f, q2, q1);   // single rounding: see the line search of rollout_body.h
      const vfloat vd1 = q2 * 2.f + vsel(veq0(q2), vsplat(MJ_MINVAL), vzero);
      w.per_lane_n(pk, [&](int l, float* o) {
        ls_pack(lane_val(va, l), lane_val(vcost, l), lane_val(vd0, l), lane_val(vd1, l), o[0], o[1], o[2], o[3]);
      });
    };
    auto to_pt = [&](const vfloat (&pk)[4]) {
      LsPt p;
      p.alpha = fbits(lane_val(pk[0], 0)); p.nalpha = fbits(lane_val(pk[1], 0)); p.cost = fbits(lane_val(pk[2], 0)); p.d0 = fbits(lane_val(pk[3], 0));
      return p;
    };
    auto ls_eval1 = [&](float a) {   // one point, evaluated by both groups (identical results): every lane of the half has it
      const vfloat va = vsplat(a);
      vfloat t[3] = {vzero, vzero, vzero};
#pragma unroll
      for (int q = 0; q < RPL; q++) {
        const vfloat act_ = vsel(vlt0(lJa[q] + ljv[q] * va), vsplat(1.f), vzero);
        t[0] = t[0] + act_ * Q0[q];
        t[1] = t[1] + act_ * Q1[q];
        t[2] = t[2] + act_ * Q2[q];
      }
      w.row16_sumN(t);
      vfloat pk[4];
      finish(va, t[0], t[1], t[2], pk);
      return to_pt(pk);
    };
    DIAL_MARK(w, 26);   // line-search set-up (J v, M v, sums, re-layout)
    const LsPt p0 = ls_eval1(0.f);
    const LsPt p1 = ls_eval1(bitsf(p0.nalpha));
    LsPt lo, hi;
    ls_open(p0, p1, lo, hi);
    const int kg = fkey(gtol), kng = fkey(-gtol);
    bool swap = true;
    int ls_iter = 0;
    for (;;) {
      const bool ls_done = (ls_iter >= max_ls) | !swap | ls_converged(lo, hi, kg, kng);
      if (ls_done) break;
      const vfloat va = vsel(g0, vsplat(bitsf(lo.nalpha)), vsplat(bitsf(hi.nalpha)));   // groups: lo_next, hi_next
      const vfloat vc = vsplat(0.5f * (bitsf(lo.alpha) + bitsf(hi.alpha)));             // both groups: mid
      vfloat t[6] = {vzero, vzero, vzero, vzero, vzero, vzero};
#pragma unroll
      for (int q = 0; q < RPL; q++) {
        const vfloat act_a = vsel(vlt0(lJa[q] + ljv[q] * va), vsplat(1.f), vzero);
        const vfloat act_c = vsel(vlt0(lJa[q] + ljv[q] * vc), vsplat(1.f), vzero);
        t[0] = t[0] + act_a * Q0[q];
        t[1] = t[1] + act_a * Q1[q];
        t[2] = t[2] + act_a * Q2[q];
        t[3] = t[3] + act_c * Q0[q];
        t[4] = t[4] + act_c * Q1[q];
        t[5] = t[5] + act_c * Q2[q];
      }
      w.row16_sumN(t);
      vfloat pkab[4], pkc[4], pkA[4], pkB[4];
      finish(va, t[0], t[1], t[2], pkab);
      finish(vc, t[3], t[4], t[5], pkc);
#pragma unroll
      for (int k = 0; k < 4; k++) w.dup_rows(pkab[k], pkA[k], pkB[k]);   // group 0's point / group 1's point, to the whole half
      // the bracket update on the slope keys alone (ls_bracket.h: ls_update_lazy); `lane` 0 / 1 / 2 names the winner: lo_next, hi_next, mid
      swap = ls_update_lazy(rule_swap, lo, hi, fbits(lane_val(pkA[3], 0)), fbits(lane_val(pkB[3], 0)), fbits(lane_val(pkc[3], 0)), 0, 1, 2,
                            [&](int word, int which) {
                              const int a_ = fbits(lane_val(pkA[word], 0)), b_ = fbits(lane_val(pkB[word], 0)), c_ = fbits(lane_val(pkc[word], 0));
                              return which == 0 ? a_ : (which == 1 ? b_ : c_);
                            });
      ls_iter++;
    }
    float alpha;
    const bool improved = ls_result(p0, lo, hi, alpha);
    if (improved) {
      vqacc = vqacc + vsearch * alpha;
      vMa = vMa + vmv * alpha;
      vJa = vJa + vjv * alpha;
      cJa = cJa + cjv * alpha;
    }
    niter++;
    DIAL_MARK(w, 7);
  }
  w.items(NV, [&](int i) {
    const float q = lane_val(vqacc, i);
    s.qacc[i] = q;
    s.warm[i] = q;
  });
  DIAL_MARK(w, 8);
}

}  // namespace dial

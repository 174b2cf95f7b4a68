// emu.cpp -- host "wave emulator" build of the kernel body (TEST INFRASTRUCTURE ONLY).
//
// Compiles dial_mpc_amd/csrc/rollout_driver.h with -DDIAL_EMU: lanes run sequentially and, when
// check_races != 0, every phase is executed in both item orders from the same LDS snapshot to expose
// intra-phase dependences (see wave.h).  This lets `pytest -m "not gpu"` compare the exact kernel
// logic with the oracle on a machine without a GPU.  It is never loaded by dial_mpc_amd/.
//
// `path`: 0 = pick the dimension-specialised instantiation when the model matches one (what the HIP
// library does), 1 = force the generic (capacity-dimension) instantiation.
#define DIAL_EMU 1
#include "../../dial_mpc_amd/csrc/rollout_driver.h"

#include <cstdlib>
#include <vector>
#include <xmmintrin.h>

namespace {

template <class D>
struct Runner {
  CModel<D> cm;
  int ws_words, con_cap = 0, ovf_words = 0;
  Runner(const dial_model* m, const dial_task* t, const dial_derived* dv) {
    fill_cmodel(&cm, m, t, dv);
    Ws s;
    // DIAL_EMU_CON_CAP=k: the GPU rollout kernel's capped workspace (derived.h: ws_carve) + overflow area, on the host
    if (const char* e = std::getenv("DIAL_EMU_CON_CAP")) con_cap = D::gen ? std::atoi(e) : 0;
    ws_words = ws_carve(s, (float*)0, m->nq, m->nv, m->nu, m->nbody, m->njnt, m->ngeom, m->nsite, m->ncon, m->nefc,
                        DIAL_MAX_NODE, dial::kNeedL<D>, D::square, D::ell ? D::JCW : 0, con_cap, D::NVP, D::pre_ctrl ? DIAL_MAX_T : 0);
    if (con_cap > 0) ovf_words = ws_overflow(s, (float*)0, m->nv, m->ncon, m->nefc);
  }
  void setup(std::vector<float>& lds, Ws& s, Wave& w, int check_races) const {
    // (the overflow area lives behind the LDS image in the same vector: the race detector then sees both)
    lds.assign(ws_words + ovf_words, 0.f);
    ws_carve(s, lds.data(), cm.nq, cm.nv, cm.nu, cm.nbody, cm.njnt, cm.ngeom, cm.nsite, cm.ncon, cm.nefc,
             DIAL_MAX_NODE, dial::kNeedL<D>, D::square, D::ell ? D::JCW : 0, con_cap, D::NVP, D::pre_ctrl ? DIAL_MAX_T : 0);
    if (s.con_cap > 0) s.ovf = lds.data() + ws_words;
    w.lds = lds.data();
    w.lds_words = ws_words + ovf_words;
    w.check_races = check_races != 0;
    // debugging aids: DIAL_EMU_NANCHECK=1 reports the first phase that leaves a non-finite value in LDS; DIAL_EMU_FTZ=1 flushes
    // fp32 denormals (inputs and results) like the GPU's sqrt / rcp / rsq do -- x / denormal = inf, sqrt(denormal) = 0
    w.nancheck = std::getenv("DIAL_EMU_NANCHECK") != nullptr;
    if (std::getenv("DIAL_EMU_FTZ")) _mm_setcsr(_mm_getcsr() | 0x8040);
    if (w.nancheck) {
      static bool once = false;
      if (!once) {
        once = true;
#define EMU_OFF(name) std::fprintf(stderr, "[wave_emu] %-8s at word %d\n", #name, (int)(s.name - lds.data()));
        EMU_OFF(qpos) EMU_OFF(qvel) EMU_OFF(warm) EMU_OFF(info) EMU_OFF(ctrl) EMU_OFF(xpos) EMU_OFF(xquat) EMU_OFF(com) EMU_OFF(cvel) EMU_OFF(cdof)
        EMU_OFF(M) EMU_OFF(cdist) EMU_OFF(cpos) EMU_OFF(cframe) EMU_OFF(Jc) EMU_OFF(D) EMU_OFF(aref) EMU_OFF(lsign) EMU_OFF(Jaref) EMU_OFF(qfs)
        EMU_OFF(qas) EMU_OFF(qacc) EMU_OFF(Ma) EMU_OFF(rhs) EMU_OFF(con_on) EMU_OFF(qfc) EMU_OFF(ulist) EMU_OFF(H) EMU_OFF(jv) EMU_OFF(frc)
        EMU_OFF(cwd) EMU_OFF(cwa) EMU_OFF(cwb) EMU_OFF(ccf) EMU_OFF(vec0) EMU_OFF(vec1) EMU_OFF(gpos) EMU_OFF(gaxis) EMU_OFF(Y)
#undef EMU_OFF
      }
    }
  }
};

// WV: Wave (one sample per wavefront, the 64-lane layouts) or WaveH (one half of the two-samples-per-wavefront kernel: the
// 32-lane layouts of smooth_quad2.h / solver_reg2.h).  tree: wave sums in the GPU's association (wave.h: emu_row_tree) -- what
// lets the two layouts be compared bit for bit.
template <class D, class WV = Wave>
int run_rollout(const dial_model* m, const dial_task* t, const dial_derived* dv, const dial_cfg* cfg,
                const dial::RolloutIO& io, int B, int check_races, bool tree = false) {
  Runner<D> r(m, t, dv);
  int races = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : races)
  for (int n = 0; n < B; n++) {
    std::vector<float> lds;
    Ws s;
    WV w;
    r.setup(lds, s, w, check_races);
    w.tree_sums = tree;
    dial::rollout_sample<true>(w, &r.cm, t, cfg, s, io, n);
    races += w.races;
  }
  return races;
}
template <class D>
int run_env_step(const dial_model* m, const dial_task* t, const dial_derived* dv, float* state, const float* action,
                 float* xpos, float* xquat, float* ctrl, int check_races) {
  Runner<D> r(m, t, dv);
  std::vector<float> lds;
  Ws s;
  Wave w;
  r.setup(lds, s, w, check_races);
  dial::env_step_single(w, &r.cm, t, s, state, action, xpos, xquat, ctrl);
  return w.races;
}
template <class D>
int run_env_reset(const dial_model* m, const dial_task* t, const dial_derived* dv, const float* qpos,
                  const float* qvel, float* state, float* xpos, float* xquat, int check_races) {
  Runner<D> r(m, t, dv);
  std::vector<float> lds;
  Ws s;
  Wave w;
  r.setup(lds, s, w, check_races);
  dial::env_reset_single(w, &r.cm, s, qpos, qvel, state, xpos, xquat);
  return w.races;
}

#define DISPATCH(path, m, CALL)                                        \
  if ((m)->cone == DIAL_CONE_ELLIPTIC) {                                \
    if (dims_match<DimsAllegro>(m) && ell_fits<DimsAllegro>(m, &dv)) return CALL(DimsAllegro); \
    return DIAL_ERR_UNSUPPORTED;   /* elliptic cones: dimension-specialised instantiations only */ \
  }                                                                    \
  if ((m)->eulerdamp) return DIAL_ERR_UNSUPPORTED;                     \
  if ((path) == 0 && dims_match<DimsGo2>(m)) return CALL(DimsGo2);     \
  if ((path) == 0 && dims_match<DimsH1>(m)) return CALL(DimsH1);       \
  if ((path) == 0 && dims_match<DimsH1Loco>(m)) return CALL(DimsH1Loco); \
  if ((path) == 0 && dims_match<DimsGo2Crate>(m)) return CALL(DimsGo2Crate); \
  if ((path) == 0 && dims_match<DimsH1PushCrate>(m)) return CALL(DimsH1PushCrate); \
  return CALL(DimsMax);

}  // namespace

extern "C" {

int emu_rollout(const dial_model* m, const dial_task* t, const dial_cfg* cfg, const float* state, const float* us,
                const float* eps, const float* Ybar, const float* noise_scale, int ns, int n_noise, int B, int T,
                int Hn1, float* Y0s, float* rewss, float* rews, float* qss, float* qdss, float* xss,
                int check_races, int path, float* trace) {
  dial_derived dv;
  int rc = dial_build_derived(m, &dv);
  if (rc) return rc;
  dial::RolloutIO io{state, us, eps, Ybar, noise_scale, ns, n_noise, T, Hn1, Y0s, rewss, rews, qss, qdss, xss, nullptr, 0, 0u, 0u, 0u, 0};
  io.trace = trace;   // [B,T,nstate] packed state after every env.step, or nullptr
  // path 2: the Go2's 32-lane (half-wave) layouts; path 3: its 64-lane layouts with the GPU's summation order (the reference
  // path 2 must equal bit for bit)
  if (path == 2 || path == 3) {
    if (!dims_match<DimsGo2>(m)) return DIAL_ERR_UNSUPPORTED;
    if (path == 2) return run_rollout<DimsGo2, WaveH>(m, t, &dv, cfg, io, B, check_races, true);
    return run_rollout<DimsGo2, Wave>(m, t, &dv, cfg, io, B, check_races, true);
  }
#define CALL(D) run_rollout<D>(m, t, &dv, cfg, io, B, check_races)
  DISPATCH(path, m, CALL)
#undef CALL
}

int emu_env_step(const dial_model* m, const dial_task* t, float* state, const float* action, float* xpos,
                 float* xquat, float* ctrl, int check_races, int path) {
  dial_derived dv;
  int rc = dial_build_derived(m, &dv);
  if (rc) return rc;
#define CALL(D) run_env_step<D>(m, t, &dv, state, action, xpos, xquat, ctrl, check_races)
  DISPATCH(path, m, CALL)
#undef CALL
}

int emu_env_reset(const dial_model* m, const dial_task* t, const float* qpos, const float* qvel, float* state,
                  float* xpos, float* xquat, int check_races, int path) {
  dial_derived dv;
  int rc = dial_build_derived(m, &dv);
  if (rc) return rc;
#define CALL(D) run_env_reset<D>(m, t, &dv, qpos, qvel, state, xpos, xquat, check_races)
  DISPATCH(path, m, CALL)
#undef CALL
}

int emu_sizes(const dial_model* m, int* cmodel_bytes, int* ws_words) {
  dial_derived dv;
  if (dial_build_derived(m, &dv)) return -1;
  dial_task t{};
  if (dims_match<DimsGo2>(m)) { Runner<DimsGo2> r(m, &t, &dv); *cmodel_bytes = (int)sizeof(r.cm); *ws_words = r.ws_words; return 1; }
  if (dims_match<DimsH1>(m)) { Runner<DimsH1> r(m, &t, &dv); *cmodel_bytes = (int)sizeof(r.cm); *ws_words = r.ws_words; return 2; }
  if (dims_match<DimsH1Loco>(m)) { Runner<DimsH1Loco> r(m, &t, &dv); *cmodel_bytes = (int)sizeof(r.cm); *ws_words = r.ws_words; return 3; }
  if (dims_match<DimsAllegro>(m) && ell_fits<DimsAllegro>(m, &dv)) { Runner<DimsAllegro> r(m, &t, &dv); *cmodel_bytes = (int)sizeof(r.cm); *ws_words = r.ws_words; return 4; }
  if (dims_match<DimsGo2Crate>(m)) { Runner<DimsGo2Crate> r(m, &t, &dv); *cmodel_bytes = (int)sizeof(r.cm); *ws_words = r.ws_words; return 5; }
  if (dims_match<DimsH1PushCrate>(m)) { Runner<DimsH1PushCrate> r(m, &t, &dv); *cmodel_bytes = (int)sizeof(r.cm); *ws_words = r.ws_words; return 6; }
  Runner<DimsMax> r(m, &t, &dv);
  *cmodel_bytes = (int)sizeof(r.cm);
  *ws_words = r.ws_words;
  return 0;
}
// the kernel's box narrow phases (csrc/box_collide.h) on one pair: geoms as (pos[3], quat[4], size[3])
int emu_box_contact(int kind, int sub, const float* g1, const float* g2, float* dist, float* pos, float* frame) {
  dial::BoxG b1, b2;
  for (int k = 0; k < 3; k++) { b1.c[k] = g1[k]; b1.h[k] = g1[7 + k]; b2.c[k] = g2[k]; b2.h[k] = g2[7 + k]; }
  for (int k = 0; k < 4; k++) { b1.q[k] = g1[3 + k]; b2.q[k] = g2[3 + k]; }
  float mat[9];
  dm::quat_to_mat(mat, b1.q);
  const float ax1[3] = {mat[2], mat[5], mat[8]};
  if (kind == DIAL_CON_PLANE_BOX) dial::plane_box(ax1, g1, b2, sub, *dist, pos, frame);
  else if (kind == DIAL_CON_SPHERE_BOX) dial::sphere_box(g1, g1[7], b2, *dist, pos, frame);
  else if (kind == DIAL_CON_CAPSULE_BOX) dial::capsule_box(g1, ax1, g1[8], g1[7], b2, sub, *dist, pos, frame);
  else if (kind == DIAL_CON_BOX_BOX) { float poly[DIAL_BOX_POLY_WORDS]; dial::box_box(b1, b2, sub, *dist, pos, frame, poly); }
  else return -1;
  return 0;
}
// the bracket update of the line search on integer slope keys (csrc/ls_bracket.h: ls_update_lazy): n cases of
// (lo.d0, hi.d0, k_lo_next, k_hi_next, k_mid) -> (lo.d0', hi.d0', lo_sel, hi_sel, any); sel = 0 / 1 / 2 names the winner
// (lo_next / hi_next / mid), negative: the end kept its point
int emu_ls_update(int rule_swap, int n, const int* in5, int* out5) {
  for (int i = 0; i < n; i++) {
    dial::LsPt lo{0, 0, 0, in5[5 * i]}, hi{0, 0, 0, in5[5 * i + 1]};
    int lo_lane = -1, hi_lane = -1, calls = 0;
    const auto fetch = [&](int word, int lane) { if (word == 0) (calls++ == 0 ? lo_lane : hi_lane) = lane; return 1000 + lane; };
    // rule_swap 2: `_in_bracket` in the boolean form the capacity-dimension kernel keeps (ls_bracket.h: MINMAX = false)
    const bool any = rule_swap == 2 ? dial::ls_update_lazy<false>(false, lo, hi, in5[5 * i + 2], in5[5 * i + 3], in5[5 * i + 4], 0, 1, 2, fetch)
                                    : dial::ls_update_lazy<true>(rule_swap != 0, lo, hi, in5[5 * i + 2], in5[5 * i + 3], in5[5 * i + 4], 0, 1, 2, fetch);
    // (the fetches run for both ends whether they moved or not; which lane each END took is read off its new alpha word)
    out5[5 * i] = lo.d0; out5[5 * i + 1] = hi.d0;
    out5[5 * i + 2] = lo.alpha >= 1000 ? lo.alpha - 1000 : -1;
    out5[5 * i + 3] = hi.alpha >= 1000 ? hi.alpha - 1000 : -1;
    out5[5 * i + 4] = any ? 1 : 0;
    (void)lo_lane; (void)hi_lane;
  }
  return 0;
}
// the loop's done-test in both forms (ls_bracket.h): n cases of (lo.d0, hi.d0, kg, kng) -> out[i] = ls_converged | (lo / hi range-compare form) << 1
int emu_ls_converged(int n, const int* in4, int* out) {
  for (int i = 0; i < n; i++) {
    dial::LsPt lo{0, 0, 0, in4[4 * i]}, hi{0, 0, 0, in4[4 * i + 1]};
    const int kg = in4[4 * i + 2], kng = in4[4 * i + 3];
    const dial::LsGate g = dial::ls_gate(kg, kng);
    out[i] = (dial::ls_converged(lo, hi, kg, kng) ? 1 : 0) | ((dial::ls_converged_lo(lo, g) || dial::ls_converged_hi(hi, g)) ? 2 : 0);
  }
  return 0;
}
}

// emu.cpp -- host "wave emulator" build of the kernel body (TEST INFRASTRUCTURE ONLY).
//
// Compiles dial_mpc_amd/csrc/rollout_driver.h with -DDIAL_EMU: lanes run sequentially and, when
// check_races != 0, every phase is executed in both item orders from the same LDS snapshot to expose
// intra-phase dependences (see wave.h).  This lets `pytest -m "not gpu"` compare the exact kernel
// logic with the oracle on a machine without a GPU.  It is never loaded by dial_mpc_amd/.
#define DIAL_EMU 1
#include "../../dial_mpc_amd/csrc/rollout_driver.h"

#include <vector>

extern "C" {

int emu_rollout(const dial_model* m, const dial_task* t, const dial_cfg* cfg, const float* state, const float* us,
                const float* eps, const float* Ybar, const float* noise_scale, int ns, int n_noise, int B, int T,
                int Hn1, float* Y0s, float* rewss, float* rews, float* qss, float* qdss, float* xss,
                int check_races) {
  dial_derived dv;
  int rc = dial_build_derived(m, &dv);
  if (rc) return rc;
  if (m->eulerdamp) return DIAL_ERR_UNSUPPORTED;
  int races = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : races)
  for (int n = 0; n < B; n++) {
    std::vector<float> lds(dv.ws_words, 0.f);
    Ws s;
    ws_carve(s, lds.data(), m->nq, m->nv, m->nu, m->nbody, m->njnt, m->ngeom, m->nsite, m->ncon, m->nefc,
             DIAL_MAX_NODE);
    Wave w;
    w.lds = lds.data();
    w.lds_words = dv.ws_words;
    w.check_races = check_races != 0;
    dial::RolloutIO io{state, us, eps, Ybar, noise_scale, ns, n_noise, T, Hn1, Y0s, rewss, rews, qss, qdss, xss, nullptr};
    dial::rollout_sample(w, m, t, &dv, cfg, s, io, n);
    races += w.races;
  }
  return races;
}

int emu_env_step(const dial_model* m, const dial_task* t, float* state, const float* action, float* xpos,
                 float* xquat, float* ctrl, int check_races) {
  dial_derived dv;
  int rc = dial_build_derived(m, &dv);
  if (rc) return rc;
  std::vector<float> lds(dv.ws_words, 0.f);
  Ws s;
  ws_carve(s, lds.data(), m->nq, m->nv, m->nu, m->nbody, m->njnt, m->ngeom, m->nsite, m->ncon, m->nefc, DIAL_MAX_NODE);
  Wave w;
  w.lds = lds.data(); w.lds_words = dv.ws_words; w.check_races = check_races != 0;
  dial::env_step_single(w, m, t, &dv, s, state, action, xpos, xquat, ctrl);
  return w.races;
}

int emu_env_reset(const dial_model* m, const dial_task* t, const float* qpos, const float* qvel, float* state,
                  float* xpos, float* xquat, int check_races) {
  dial_derived dv;
  int rc = dial_build_derived(m, &dv);
  if (rc) return rc;
  std::vector<float> lds(dv.ws_words, 0.f);
  Ws s;
  ws_carve(s, lds.data(), m->nq, m->nv, m->nu, m->nbody, m->njnt, m->ngeom, m->nsite, m->ncon, m->nefc, DIAL_MAX_NODE);
  Wave w;
  w.lds = lds.data(); w.lds_words = dv.ws_words; w.check_races = check_races != 0;
  dial::env_reset_single(w, m, t, &dv, s, qpos, qvel, state, xpos, xquat);
  return w.races;
}

int emu_ws_words(const dial_model* m) {
  dial_derived dv;
  if (dial_build_derived(m, &dv)) return -1;
  return dv.ws_words;
}
}

// kern_family.hip -- the rollout / env.step / env.reset kernels of ONE robot family (-DDIAL_FAMILY=0..7), see kernel_list.h.
#include "kernel_list.h"
#define DIAL_X(D, WPB, OCC, Q, TR) \
  template __global__ void rollout_kernel<D, WPB, OCC, Q, TR>(const CModel<D>*, const dial_task*, const dial_cfg*, dial::RolloutIO, int, int, int*);
#define DIAL_XE(D)                                                                                                       \
  template __global__ void env_step_kernel<D>(const CModel<D>*, const dial_task*, float*, const float*, float*, float*, float*); \
  template __global__ void env_reset_kernel<D>(const CModel<D>*, const float*, const float*, float*, float*, float*);
#define DIAL_X2(D, WPB, OCC, Q, MI) \
  template __global__ void rollout_kernel2<D, WPB, OCC, Q, MI>(const CModel<D>*, const dial_task*, const dial_cfg*, dial::RolloutIO, int, int, int*);
#if DIAL_FAMILY == 0
DIAL_KERNELS_GO2(DIAL_X, DIAL_XE)
#elif DIAL_FAMILY == 7
DIAL_KERNELS2_GO2(DIAL_X2)
#elif DIAL_FAMILY == 1
DIAL_KERNELS_H1(DIAL_X, DIAL_XE)
#elif DIAL_FAMILY == 2
DIAL_KERNELS_H1LOCO(DIAL_X, DIAL_XE)
#elif DIAL_FAMILY == 3
DIAL_KERNELS_ALLEGRO(DIAL_X, DIAL_XE)
#elif DIAL_FAMILY == 4
DIAL_KERNELS_GENERIC(DIAL_X, DIAL_XE)
#elif DIAL_FAMILY == 5
DIAL_KERNELS_GO2CRATE(DIAL_X, DIAL_XE)
#elif DIAL_FAMILY == 6
DIAL_KERNELS_H1PUSHCRATE(DIAL_X, DIAL_XE)
#else
#error "DIAL_FAMILY must be 0 .. 7"
#endif

// probe.hip -- instantiate ONE rollout kernel for ISA inspection (tools/isa/probe.sh): VGPR / SGPR / scratch / spill
// counts and the instruction stream of a single instantiation in ~20 s instead of the whole library's 90 s.
#include "../../dial_mpc_amd/csrc/rollout_kernel.h"
#ifndef PROBE_D
#define PROBE_D DimsMax
#endif
#ifndef PROBE_WPB
#define PROBE_WPB 1
#endif
#ifndef PROBE_OCC
#define PROBE_OCC 3
#endif
#ifndef PROBE_QUEUE
#define PROBE_QUEUE false
#endif
template __global__ void rollout_kernel<PROBE_D, PROBE_WPB, PROBE_OCC, PROBE_QUEUE>(const CModel<PROBE_D>*, const dial_task*, const dial_cfg*,
                                                                                    dial::RolloutIO, int, int, int*);

#ifdef PROBE_PAIR
template __global__ void rollout_kernel2<PROBE_D, PROBE_WPB, PROBE_OCC, PROBE_QUEUE>(const CModel<PROBE_D>*, const dial_task*, const dial_cfg*,
                                                                                     dial::RolloutIO, int, int, int*);
#endif

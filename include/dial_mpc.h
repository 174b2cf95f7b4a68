/*
 * dial_mpc.h -- C ABI of the MI355X-native DIAL-MPC inner loop.
 *
 * This header is the single source of truth for
 *   (1) the plain-old-data structs that describe a compiled robot model, an
 *       environment task and the planner configuration, and
 *   (2) the entry points of libdialhip.so (the HIP product path) and of
 *       liboracle_f32/f64.so (the CPU oracle, test infrastructure only).
 * dial_mpc_amd/_abi.py parses THIS file to build the ctypes mirrors, so a field
 * added here is automatically visible from Python.  Keep the struct syntax
 * simple: one field per line, `int32_t`/`float` only, array extents as DIAL_*
 * macros or literals.
 *
 * Reference interfaces replaced (paths relative to LeCAR-Lab/dial-mpc):
 *   dial_rollout        <- MBDPI.rollout_us_vmap      dial_mpc/core/dial_core.py:36-42,80-81
 *   dial_reverse_once   <- MBDPI.reverse_once         dial_mpc/core/dial_core.py:103-145
 *   dial_shift          <- MBDPI.shift                dial_mpc/core/dial_core.py:160-166
 *   dial_env_step       <- BaseEnv/<Env>.step         dial_mpc/envs/unitree_go2_env.py:126-261,
 *                                                     :403-521, dial_mpc/envs/unitree_h1_env.py:181-321,
 *                                                     :696-858 (H1 loco), dial_mpc/envs/manipulation.py:63-115
 *                                                     (Allegro in-hand reorientation + its act2joint override),
 *                                                     unitree_go2_env.py:679-795 (crate climb),
 *                                                     unitree_h1_env.py:418-566 (push crate)
 *   dial_env_reset      <- <Env>.reset + pipeline_init  dial_mpc/envs/unitree_go2_env.py:101-124
 *   dial_model          <- brax System / mujoco MjModel built by BaseEnv.make_system
 *                                                     dial_mpc/envs/base_env.py:15-29
 *   dial_task           <- <Env>Config dataclasses    dial_mpc/envs/unitree_go2_env.py:25-33,
 *                                                     dial_mpc/config/base_env_config.py:4-20
 *   dial_cfg            <- DialConfig                 dial_mpc/core/dial_config.py:4-23
 *
 * All device pointers are plain `float*`/`const float*` into HBM owned by the
 * caller (PyTorch-ROCm tensors via .data_ptr()); `stream` is a hipStream_t passed
 * as void* so that this header has no HIP dependency.
 */
#ifndef DIAL_MPC_H
#define DIAL_MPC_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- static capacity limits (all robots in the reference fit) --------------- */
#define DIAL_MAX_BODY 24   /* incl. world body 0 (H1 walk: 21)                   */
#define DIAL_MAX_JNT 24
#define DIAL_MAX_Q 32
#define DIAL_MAX_V 28      /* H1 walk: 25                                        */
#define DIAL_MAX_U 20      /* H1 walk: 19                                        */
#define DIAL_MAX_GEOM 20   /* collision geoms only (Go2 crate scene: 15 robot geoms + floor + crate) */
#define DIAL_MAX_SITE 8
#define DIAL_MAX_CON 56    /* static contact list (Go2: 4, H1 walk: 4, Allegro: 19, Go2 crate scene: 52) */
#define DIAL_MAX_EFC 240   /* constraint rows: limits + 4 per pyramidal / condim per elliptic contact (Allegro: 88, crate scene: 220) */
#define DIAL_MAX_LIM 24    /* limited hinge joints                               */
#define DIAL_MAX_FRI 4     /* dofs with frictionloss (push-crate scene: the crate's slide joint) */
#define DIAL_MAX_FEET 4
#define DIAL_MAX_STAGE 12  /* seq-jump stages                                    */
#define DIAL_MAX_T 36      /* Hsample+1                                          */
#define DIAL_MAX_NODE 10   /* Hnode+1                                            */
#define DIAL_INFO_N 48     /* floats of env info in the packed state             */
#define DIAL_MAX_CMD 16    /* randomize_tasks: velocity commands of the episodes step / 500 = 0, 1, ... (wraps)  */

/* joint types (MuJoCo numbering) */
#define DIAL_JNT_FREE 0
#define DIAL_JNT_BALL 1
#define DIAL_JNT_SLIDE 2
#define DIAL_JNT_HINGE 3

/* geom types (MuJoCo numbering, subset) */
#define DIAL_GEOM_PLANE 0
#define DIAL_GEOM_SPHERE 2
#define DIAL_GEOM_CAPSULE 3
#define DIAL_GEOM_BOX 6

/* static contact kinds */
#define DIAL_CON_PLANE_SPHERE 0
#define DIAL_CON_PLANE_CAPSULE_P 1 /* capsule end  +axis*halflen */
#define DIAL_CON_PLANE_CAPSULE_N 2 /* capsule end  -axis*halflen */
#define DIAL_CON_SPHERE_CAPSULE 3  /* geom1 sphere, geom2 capsule (MJX sphere_capsule)        */
#define DIAL_CON_CAPSULE_CAPSULE 4 /* MJX capsule_capsule: closest points of the two segments */
/* Box narrow phases (the crate scenes).  MJX routes boxes through collision_convex.py, which is NOT in
 * /root/reference (third-party, un-vendored) and whose manifold selection differs between releases; what follows is a
 * restatement of the GEOMETRY (unique wherever the contact is unique), not of MJX's vertex bookkeeping -- see DESIGN.md
 * section 1 "crate scenes".  A pair yields a FIXED number of candidate contacts (con_sub = 0, 1, ...), as MJX's do:
 * candidates that do not touch carry dist >= margin and produce zero rows.                                          */
#define DIAL_CON_PLANE_BOX 5    /* geom1 plane, geom2 box: con_sub = k-th lowest of the 8 vertices (k < 4)              */
#define DIAL_CON_SPHERE_BOX 6   /* geom1 sphere, geom2 box: closest point of the box to the centre                      */
#define DIAL_CON_CAPSULE_BOX 7  /* geom1 capsule, geom2 box: con_sub 0 = the segment point closest to the box, 1 = the
                                   segment end farther from that point (both as spheres of the capsule's radius); bounding
                                   spheres more than 1 cm apart: both candidates parked at dist > 0 (as DIAL_CON_BOX_BOX) */
#define DIAL_CON_BOX_BOX 8      /* separating-axis test; face contact: incident face clipped against the reference face,
                                   con_sub = k-th deepest point (k < 4); edge contact: one point (con_sub 0)            */

/* bracket-update rule of the Newton solver's line search (solver._linesearch of MJX).  The reference pins no MJX
 * version (setup.py:9-21); its Allegro env needs elliptic cones, i.e. a release (>= 3.1.4) that carries the
 * `_in_bracket` rule, and tools/reference_env.txt pins 3.2.7 -- so DIAL_LS_IN_BRACKET is what dial_mpc_amd/mjcf.py
 * writes into EVERY compiled model.  The older rule stays selectable per model because it is well-conditioned under
 * truncation (ls_iterations = 5): with it GPU and oracle can be compared rollout by rollout at the envs' real solver
 * settings (tests), whereas `_in_bracket` rejects zero-slope candidates and turns rounding noise into different
 * iterates -- under it the product outputs are gated at the distribution level (tests/conftest.py).               */
#define DIAL_LS_SWAP 0        /* MJX <= 3.1.3: swap_lo_next / swap_lo_mid / swap_hi_next / swap_hi_mid            */
#define DIAL_LS_IN_BRACKET 1  /* MJX >= 3.1.4: _in_bracket(x, y), each end offered lo_next, mid, hi_next          */

/* friction cone (mjtCone) */
#define DIAL_CONE_PYRAMIDAL 0
#define DIAL_CONE_ELLIPTIC 1

/* task kinds */
#define DIAL_TASK_GO2_WALK 0
#define DIAL_TASK_GO2_SEQ_JUMP 1
#define DIAL_TASK_H1_WALK 2
#define DIAL_TASK_H1_LOCO 3
#define DIAL_TASK_ALLEGRO 4
#define DIAL_TASK_GO2_CRATE 5  /* UnitreeGo2CrateEnv.step (unitree_go2_env.py:679-795) */
#define DIAL_TASK_H1_PUSH_CRATE 6  /* UnitreeH1PushCrateEnv.step (unitree_h1_env.py:418-566) */

/* packed-state info slots (floats; integers are stored as exactly representable floats) */
#define DIAL_INFO_STEP 0
#define DIAL_INFO_POS_TAR 1      /* 3 */
#define DIAL_INFO_VEL_TAR 4      /* 3 */
#define DIAL_INFO_ANG_VEL_TAR 7  /* 3 */
#define DIAL_INFO_YAW_TAR 10
#define DIAL_INFO_LAST_CONTACT 11 /* 4 */
#define DIAL_INFO_AIR_TIME 15     /* 4 */
#define DIAL_INFO_STAGE 19
#define DIAL_INFO_DONE 20
#define DIAL_INFO_REWARD 21
#define DIAL_INFO_LAST_CTRL 22    /* DIAL_MAX_U */

/* error codes */
#define DIAL_OK 0
#define DIAL_ERR_ARG (-1)
#define DIAL_ERR_HIP (-2)
#define DIAL_ERR_UNSUPPORTED (-3)

/* Compiled robot model: what mujoco.MjModel / brax System holds for the hot path.
 * Produced by dial_mpc_amd/mjcf.py from the MJCF.  Body 0 is the world.          */
typedef struct dial_model {
  int32_t nq;
  int32_t nv;
  int32_t nu;
  int32_t nbody;
  int32_t njnt;
  int32_t ngeom;
  int32_t nsite;
  int32_t ncon;
  int32_t nlim;
  int32_t nefc;
  int32_t iterations;
  int32_t ls_iterations;
  int32_t eulerdamp;
  int32_t cone;
  int32_t ls_rule;
  float timestep;
  float gravity[3];
  float tolerance;
  float ls_tolerance;
  float impratio;
  float meaninertia;
  int32_t body_parent[DIAL_MAX_BODY];
  int32_t body_jntadr[DIAL_MAX_BODY];
  int32_t body_jntnum[DIAL_MAX_BODY];
  int32_t body_dofadr[DIAL_MAX_BODY];
  int32_t body_dofnum[DIAL_MAX_BODY];
  int32_t body_depth[DIAL_MAX_BODY];
  int32_t body_subtree_end[DIAL_MAX_BODY];
  int32_t body_rootid[DIAL_MAX_BODY];
  float body_pos[DIAL_MAX_BODY][3];
  float body_quat[DIAL_MAX_BODY][4];
  float body_ipos[DIAL_MAX_BODY][3];
  float body_iquat[DIAL_MAX_BODY][4];
  float body_mass[DIAL_MAX_BODY];
  float body_inertia[DIAL_MAX_BODY][3];
  float body_invweight0[DIAL_MAX_BODY][2];
  int32_t jnt_type[DIAL_MAX_JNT];
  int32_t jnt_qposadr[DIAL_MAX_JNT];
  int32_t jnt_dofadr[DIAL_MAX_JNT];
  int32_t jnt_bodyid[DIAL_MAX_JNT];
  int32_t jnt_limited[DIAL_MAX_JNT];
  float jnt_pos[DIAL_MAX_JNT][3];
  float jnt_axis[DIAL_MAX_JNT][3];
  float jnt_range[DIAL_MAX_JNT][2];
  float jnt_solref[DIAL_MAX_JNT][2];
  float jnt_solimp[DIAL_MAX_JNT][5];
  float jnt_margin[DIAL_MAX_JNT];
  float qpos0[DIAL_MAX_Q];
  float key_qpos[DIAL_MAX_Q];
  int32_t dof_bodyid[DIAL_MAX_V];
  int32_t dof_jntid[DIAL_MAX_V];
  int32_t dof_parentid[DIAL_MAX_V];
  float dof_armature[DIAL_MAX_V];
  float dof_damping[DIAL_MAX_V];
  float dof_invweight0[DIAL_MAX_V];
  int32_t geom_type[DIAL_MAX_GEOM];
  int32_t geom_bodyid[DIAL_MAX_GEOM];
  float geom_pos[DIAL_MAX_GEOM][3];
  float geom_quat[DIAL_MAX_GEOM][4];
  float geom_size[DIAL_MAX_GEOM][3];
  int32_t site_bodyid[DIAL_MAX_SITE];
  float site_pos[DIAL_MAX_SITE][3];
  float site_quat[DIAL_MAX_SITE][4];
  int32_t con_kind[DIAL_MAX_CON];
  int32_t con_geom1[DIAL_MAX_CON];
  int32_t con_geom2[DIAL_MAX_CON];
  int32_t con_body1[DIAL_MAX_CON];
  int32_t con_body2[DIAL_MAX_CON];
  int32_t con_dim[DIAL_MAX_CON];
  int32_t con_sub[DIAL_MAX_CON];   /* which of the pair's candidate contacts (box kinds), 0 elsewhere */
  float con_friction[DIAL_MAX_CON][5];
  float con_solref[DIAL_MAX_CON][2];
  float con_solimp[DIAL_MAX_CON][5];
  float con_margin[DIAL_MAX_CON];
  int32_t lim_jnt[DIAL_MAX_LIM];
  /* dofs with dry friction (joint frictionloss; constraint._instantiate_friction of MJX): one constraint row each, placed
   * between the limit rows and the contact rows (nefc = nlim + nfri + contact rows); solref / solimp = the joint's
   * solreffriction / solimpfriction                                                                                  */
  int32_t nfri;
  int32_t fri_dof[DIAL_MAX_FRI];
  float fri_loss[DIAL_MAX_FRI];
  float fri_solref[DIAL_MAX_FRI][2];
  float fri_solimp[DIAL_MAX_FRI][5];
  int32_t act_dofadr[DIAL_MAX_U];
  int32_t act_qposadr[DIAL_MAX_U];
  int32_t act_ctrllimited[DIAL_MAX_U];
  int32_t act_isposition[DIAL_MAX_U];
  float act_gear[DIAL_MAX_U];
  float act_kp[DIAL_MAX_U];
  float act_ctrlrange[DIAL_MAX_U][2];
} dial_model;

/* Environment task: <Env>Config + the constants the env classes hard-code.        */
typedef struct dial_task {
  int32_t kind;
  int32_t n_frames;
  int32_t position_control;
  int32_t torso_x;
  int32_t upright_x;
  int32_t nfeet;
  int32_t feet_site[DIAL_MAX_FEET];
  int32_t n_stage;
  float dt;
  float action_scale;
  float kp[DIAL_MAX_U];
  float kd[DIAL_MAX_U];
  float joint_range[DIAL_MAX_U][2];
  float phys_range[DIAL_MAX_U][2];
  float tau_range[DIAL_MAX_U][2];
  float foot_radius;           /* joint_offset (below): added to the joint target before the clip -- the keyframe pose
                                  init_q[7:] of AllegroReorientEnv.act2joint (manipulation.py:107-109), 0 elsewhere */
  float gait_duty;
  float gait_cadence;
  float gait_amp;
  float gait_phase[DIAL_MAX_FEET];
  float cmd_vel[3];
  float cmd_ang_vel[3];
  float ramp_up_time;
  float done_height;
  float jump_dt;
  float contact_targets[DIAL_MAX_STAGE][4][3];
  float contact_radius[DIAL_MAX_STAGE][4];
  float pose_targets[DIAL_MAX_STAGE][3];
  float yaw_targets[DIAL_MAX_STAGE];
  float init_pos_tar[3];
  float init_ang_vel_tar[3];
  float joint_offset[DIAL_MAX_U];
  /* BaseEnvConfig.randomize_tasks of the walking envs (unitree_go2_env.py:142-155, unitree_h1_env.py:199-212,
   * :436-449, :718-731): on a step with info.step % 500 == 0 the command is `sample_command(rng)`, on EVERY other step
   * the default command (upstream does not store the sampled one) -- inside planner rollouts as well, since they run
   * env.step.  The draw crosses the boundary as DATA, like the planner's noise (JAX's key stream is version dependent):
   * cmd_table[e] = (vx, vy, vyaw) of episode e = step / 500 (mod n_cmd), filled by the host (dial_mpc_amd/envs) from a
   * documented generator with sample_command's ranges, or with values exported from a reference run.            */
  int32_t randomize_tasks;
  int32_t n_cmd;
  float cmd_table[DIAL_MAX_CMD][3];
  /* crate climb (unitree_go2_env.py:741-766): reward_contact counts the feet whose contact with the crate lies inside
   * the box `crate_region` = (x0, x1, y0, y1, z0, z1) -- upstream's cond on contact.pos[contact_indices[i]].  Upstream
   * hard-codes the indices [16, 17, 18, 19] of ITS MJX release's contact array; here the env class looks the four
   * foot-sphere / crate contacts up by geom identity (`crate_contact`, indices into this model's static contact list).
   * The device evaluates the four terms of :770-783 that carry a non-zero weight (head position, upright, yaw, feet on the
   * crate); the seven others (pitch, roll, vel, ang_vel, height, energy, penalty_contact) are multiplied by 0.0 upstream and
   * are not computed: identical rewards as long as they are finite, which the oracle -- it restates all eleven -- checks. */
  int32_t crate_contact[DIAL_MAX_FEET];
  float crate_region[6];
  float head_vec[3];           /* head_pos = torso pos + R head_vec (:717-718)                                         */
  /* push crate (unitree_h1_env.py:474-480, 525-531): upstream reads its MJX release's contact array by position --
   * z_feet = min(dist[2:4]), min(dist[6:8]); wanted_contacts = [26, 27]; unwanted_contacts = 14 .. 25.  Enumerated in
   * geom-pair order those are: the two floor contacts of each FOOT capsule; the two HAND spheres against the crate; every
   * other robot geom against the crate.  The env class looks them up by geom identity in this model's contact list.  */
  int32_t pc_foot_contact[2][2];
  int32_t pc_wanted[2];
  int32_t pc_n_unwanted;
  int32_t pc_unwanted[16];
  float pc_wanted_zmax;        /* 1.1 (:529)                                                                           */
} dial_task;

/* Planner configuration: DialConfig + the constant spline matrices
 * W = node2u(I) [(Hsample+1) x (Hnode+1)] and V = u2node(I) [(Hnode+1) x (Hsample+1)]. */
typedef struct dial_cfg {
  int32_t Nsample;
  int32_t Hsample;
  int32_t Hnode;
  float temp_sample;
  float W[DIAL_MAX_T][DIAL_MAX_NODE];
  float V[DIAL_MAX_NODE][DIAL_MAX_T];
} dial_cfg;

/* packed state: [qpos nq | qvel nv | qacc_warmstart nv | info DIAL_INFO_N] floats */
static inline int dial_state_size(int nq, int nv) { return nq + 2 * nv + DIAL_INFO_N; }

/* =============================== product C ABI (libdialhip.so) ================= */
typedef struct dial_ctx dial_ctx; /* opaque; one per (device, model, task, cfg).  NOT thread-safe, and ONE STREAM AT A
                                   * TIME: a context owns one set of scratch tensors, one rollout-queue head and one relay
                                   * turn flag, so two launches of the same context must never be in flight on different
                                   * streams (use one context per stream / per Python thread).                          */

/* Launch-shape / measurement options of a context (dial_create_ex).  Every field defaults to 0 = the shipped behaviour,
 * which is what dial_create / dial_create_sharded use.  NONE of them changes a result bit: the GPU suite compares each one
 * against the default launch (tests/test_gpu_parity.py, tests/test_gpu_crate.py).  libdialhip.so reads NO environment
 * variables: the way a launch is put on the chip depends on (model, task, cfg, these options) and nothing else.        */
typedef struct dial_options {
  int32_t force_generic;      /* 1: capacity-dimension (generic) kernel instantiation instead of the robot's own -- what a user model
                                 that is none of the seven runs on.  Its price, measured once with the Go2 (round 6, N = 2048 H = 16): 1.076
                                 ms per iteration against 0.347 ms on the robot's own kernel, i.e. 3.1 x (run-time dimensions, packed
                                 triangles, the LDS phase version of the position stage; profiles/r06_bench_go2_on_capacity_dimension_kernel.json) */
  int32_t con_cap;            /* generic instantiation, models with many candidate contacts: touching contacts the LDS
                                 workspace of a rollout wavefront holds (samples beyond run on an overflow area in global
                                 memory, bit-identically).  0: chosen per model (largest of 16 .. 8 that keeps nine wavefronts
                                 per CU), > 0: this cap, < 0: no cap (full-size workspace)                                 */
  int32_t no_queue;           /* 1: one wavefront per rollout at any batch size (no rollout queue)                           */
  int32_t no_relay;           /* 1: the mean trajectory runs as one more wavefront (no relay)                                */
  int32_t relay_always;       /* 1: relay at any N (default: only when N fills the SIMDs evenly)                             */
  int32_t relay_steps;        /* control steps per relay piece, 1 .. 16 (0: default 3)                                       */
  int32_t no_split_mask;      /* bit i: no split launch for kernel instantiation i (2 = H1, 4 = Allegro); -1: never split   */
  int32_t debug_relay_stall;  /* TEST HOOK k >= 1: relay piece k - 1 never hands over (its successors time out; dial_status) */
  int32_t no_slice;           /* 1: batches beyond the resident set of a model with data-dependent rollout lengths (elliptic
                                 cones: the Allegro hand) go through the plain rollout queue (whole rollouts) instead of the
                                 time-sliced one (pieces of `slice_steps` control steps, handed on through global memory)   */
  int32_t slice_steps;        /* control steps per piece of the time-sliced queue, 1 .. 16 (0: default 3)                    */
  int32_t no_lag_priority;    /* 1: pseudo-random fair SIMD sharing also for models with data-dependent rollout lengths
                                 (default there: the rollouts that are behind get the higher issue priority)                */
  int32_t no_mean_inline;     /* 1: Go2 batches beyond the resident set run the mean trajectory as an ordinary queue item (the
                                 last one: alone at the lone-wavefront pace) instead of interleaving its steps with the first
                                 T wavefronts' own                                                                          */
  int32_t no_spread;          /* 1: a Go2 batch between the small-batch limit and the large-batch kernel's resident set fills
                                 workgroup after workgroup (some CUs with 16 wavefronts, some with 8) instead of being dealt
                                 round-robin over the whole resident grid                                                     */
  int32_t pair_mode;          /* Go2: TWO rollouts per wavefront, one per 32-lane half (rollout_kernel2).  0 = for batches beyond
                                 DIAL_GO2_PAIR_MIN_B = 2304 rollouts (the one-sample kernels' resident set), where the SIMDs are short of
                                 issue slots (N = 65536: 7.2 -> 10.2 M rollouts/s); 1 = never (the one-rollout-per-wavefront kernels
                                 at any batch size); 2 = always.  The same arithmetic in the same order per rollout: bit-identical on the
                                 build without fused multiply-add contraction, at rounding level (which products the compiler fuses) on
                                 the product build.  CONSEQUENCE (product build): the kernel family is chosen from the batch a call
                                 launches, so a rollout's bits depend on it -- the same N sharded over ranks whose shards fall on the
                                 other side of 2304 (N = 4096 over 2 or 4 ranks) equals the fused run at rounding level, not bit for
                                 bit; ranks of one run always agree with each other (same shard size).  pair_mode 1 or 2 on every
                                 context restores bit-equality across shardings.                                                  */
} dial_options;

/* host pointers; copies model/task/cfg to the device and allocates scratch for
 * cfg->Nsample+1 rollouts.  Fails with DIAL_ERR_HIP when no HIP device is usable.
 * Which kernel instantiation runs is decided here from (model, task, options): the robot's own dimension-specialised one when the
 * model matches it -- dimensions, topology, task kind, one physics step per control step for the Go2's, and at most eight distinct
 * (solref, solimp) parameter sets among its limit rows / contacts / dry-friction rows (their impedance constants live in a shared
 * table of the staged model constants) -- otherwise the capacity-dimension instantiation (options.force_generic has its price). */
int dial_create(dial_ctx** out, const dial_model* model, const dial_task* task,
                const dial_cfg* cfg, int device);
/* Same, for one rank of a sample-sharded run: the rollout scratch is sized for n_local_cap (+1 mean-trajectory)
 * rollouts instead of cfg->Nsample + 1 -- at BASELINE config 5 (N = 65536 over 8 GPUs) 41 MB instead of 330 MB per
 * rank; the global reward / weight arrays still hold cfg->Nsample + 1 entries.  0 <= n_local_cap <= cfg->Nsample. */
int dial_create_sharded(dial_ctx** out, const dial_model* model, const dial_task* task,
                        const dial_cfg* cfg, int device, int n_local_cap);
/* The general form: n_local_cap < 0 means cfg->Nsample (cfg may be NULL: env.step / env.reset only); opts may be NULL. */
int dial_create_ex(dial_ctx** out, const dial_model* model, const dial_task* task, const dial_cfg* cfg, int device,
                   int n_local_cap, const dial_options* opts);
void dial_destroy(dial_ctx* ctx);
const char* dial_last_error(const dial_ctx* ctx); /* ctx may be NULL: last global error */

/* K3. Semantics of MBDPI.rollout_us_vmap (dial_core.py:80-81): B rollouts of T = Hsample+1
 * env.steps from the SAME packed state.  us:[B,T,nu]; rewss:[B,T]; qss:[B,T,nq];
 * qdss:[B,T,nv]; xposs:[B,T,(nbody-1)*3].  qss/qdss/xposs may be NULL.  All device ptrs. */
int dial_rollout(dial_ctx* ctx, const float* state, const float* us, int B, float* rewss,
                 float* qss, float* qdss, float* xposs, void* stream);

/* K1..K4. Semantics of MBDPI.reverse_once (dial_core.py:103-145) for the sample shard
 * [n_begin, n_begin+n_local) of cfg.Nsample (single GPU: 0, Nsample); the mean trajectory is
 * always rolled out as an extra sample.  eps:[n_local,Hnode+1,nu] standard-normal draws (the
 * reference's jax.random.normal, passed as data); noise_scale:[ns] with ns = Hnode+1 or 1.
 * Outputs: Ybar_out:[Hnode+1,nu], rews:[n_local+1] (last = mean trajectory), qbar:[T,nq],
 * qdbar:[T,nv], xbar:[T,(nbody-1)*3]; with n_local < Nsample the *_out tensors hold this
 * shard's partial (un-normalised) sums -- see dial_shard_reduce.
 * qbar, qdbar and xbar may ALL be NULL: the mean action alone is formed and the rollouts do not materialise their per-step
 * states (what the drivers need from every annealing iteration of a plan but the last, dial_core.py:262-264).
 * Degenerate case, bug-compatible with the reference: when all N+1 mean rewards are identical, std(rews) = 0 and
 * dial_core.py:126 divides 0 by 0 -- every weight and with it Ybar_out / qbar / qdbar / xbar is NaN.             */
int dial_reverse_once(dial_ctx* ctx, const float* state, const float* Ybar_in,
                      const float* noise_scale, int ns, const float* eps, float* Ybar_out,
                      float* rews, float* qbar, float* qdbar, float* xbar, void* stream);

/* Multi-GPU split of reverse_once: phase A rolls out this rank's shard and writes
 * its per-sample mean rewards; the caller all-gathers rews over ranks; phase B forms the global
 * weights redundantly on every rank and this rank's partial weighted sums, which the caller
 * all-reduces (sum).  packed_out: [ (Hnode+1)*nu | T*nq | T*nv | T*(nbody-1)*3 ] floats.     */
/* with_mean: bit 0 = also roll out the mean trajectory (every rank does); bit 1 (DIAL_SHARD_LEAN) = the iteration will be finished by
 * dial_shard_ybar* (mean action only): the rollouts then write neither their per-step states nor their candidate nodes.       */
#define DIAL_SHARD_LEAN 2
int dial_shard_rollout(dial_ctx* ctx, const float* state, const float* Ybar_in,
                       const float* noise_scale, int ns, const float* eps, int n_local,
                       int with_mean, float* rews_local, void* stream);
int dial_shard_reduce(dial_ctx* ctx, const float* rews_all, int n_total, int n_begin,
                      int n_local, int with_mean, float* packed_out, void* stream);
/* Single-collective variant (SURVEY 8e option (a)): after the all-gather of the rewards every rank forms the
 * COMPLETE weighted mean action locally -- the candidate nodes of all n_total samples are regenerated from
 * the full noise array eps_all:[n_total,Hnode+1,nu] (every rank holds it), so no all-reduce is needed.
 * Ybar_out:[Hnode+1,nu] is bit-identical on all ranks.                                               */
int dial_shard_ybar(dial_ctx* ctx, const float* rews_all, int n_total, const float* eps_all,
                    const float* Ybar_in, const float* noise_scale, int ns, float* Ybar_out, void* stream);

/* In-kernel noise (production path): the standard-normal draws are generated inside the
 * rollout kernel by Philox4x32-10 + Box-Muller keyed by (seed, counter = annealing-iteration index, global
 * sample index, element) -- no eps tensor crosses HBM and any rank can regenerate any sample's noise.
 * dial_rng_fill materialises exactly that noise (eps_out:[n_count,Hnode+1,nu], samples n_begin..) so that a run
 * can be replayed through dial_reverse_once / the oracle for parity.                                   */
int dial_reverse_once_rng(dial_ctx* ctx, const float* state, const float* Ybar_in, const float* noise_scale,
                          int ns, uint64_t seed, uint32_t counter, float* Ybar_out, float* rews, float* qbar,
                          float* qdbar, float* xbar, void* stream);
int dial_shard_rollout_rng(dial_ctx* ctx, const float* state, const float* Ybar_in, const float* noise_scale,
                           int ns, uint64_t seed, uint32_t counter, int n_begin, int n_local, int with_mean,
                           float* rews_local, void* stream);
int dial_rng_fill(dial_ctx* ctx, uint64_t seed, uint32_t counter, int n_begin, int n_count, float* eps_out,
                  void* stream);
/* dial_shard_ybar with the noise of ALL n_total samples regenerated inside the kernel (same Philox keys as
 * dial_shard_rollout_rng): no noise array exists anywhere in a sharded run.                              */
int dial_shard_ybar_rng(dial_ctx* ctx, const float* rews_all, int n_total, uint64_t seed, uint32_t counter,
                        const float* Ybar_in, const float* noise_scale, int ns, float* Ybar_out, void* stream);
/* gathered:[world][per+1] (the all-gather of every rank's dial_shard_rollout output) -> rews_all:[n_total+1]
 * = [all noisy samples | mean trajectory], the layout dial_shard_reduce / dial_shard_ybar* consume.      */
int dial_shard_pack_rewards(dial_ctx* ctx, const float* gathered, int world, int per, int n_total,
                            float* rews_all, void* stream);
/* The sharded iteration's phase B straight from the all-gather's receive buffer (round 6): the packing step runs inside the weights
 * kernel (rews_all_out:[n_total+1] receives what dial_shard_pack_rewards would have written), so a lean iteration is the rollout
 * launch + the all-gather + TWO launches (weights, weighted mean action), a full one the rollout + all-gather + two launches
 * (weights, weighted sums) + the all-reduce.  Same results as dial_shard_pack_rewards followed by dial_shard_ybar[_rng] /
 * dial_shard_reduce.                                                                                                    */
int dial_shard_ybar_gathered(dial_ctx* ctx, const float* gathered, int world, int per, int n_total, const float* eps_all,
                             const float* Ybar_in, const float* noise_scale, int ns, float* rews_all_out, float* Ybar_out,
                             void* stream);
int dial_shard_ybar_gathered_rng(dial_ctx* ctx, const float* gathered, int world, int per, int n_total, uint64_t seed,
                                 uint32_t counter, const float* Ybar_in, const float* noise_scale, int ns,
                                 float* rews_all_out, float* Ybar_out, void* stream);
int dial_shard_reduce_gathered(dial_ctx* ctx, const float* gathered, int world, int per, int n_total, int n_begin,
                               int n_local, int with_mean, float* rews_all_out, float* packed_out, void* stream);

/* K5. MBDPI.shift (dial_core.py:160-166): Y:[Hnode+1,nu] in place. */
int dial_shift(dial_ctx* ctx, float* Y, void* stream);

/* K6. env.step on the true state (B = 1), state updated in place; xpos_out:[(nbody-1)*3],
 * xquat_out:[(nbody-1)*4], ctrl_out:[nu] optional (may be NULL).                    */
int dial_env_step(dial_ctx* ctx, float* state, const float* action, float* xpos_out,
                  float* xquat_out, float* ctrl_out, void* stream);

/* env.reset: state <- (qpos, qvel, zeros) followed by mjx.forward (pipeline_init), info
 * initialised for the task.  qpos:[nq], qvel:[nv] device pointers.                  */
int dial_env_reset(dial_ctx* ctx, const float* qpos, const float* qvel, float* state,
                   float* xpos_out, float* xquat_out, void* stream);
/* The same for n states in ONE launch (one workgroup each): qpos:[n,nq], qvel:[n,nv], states:[n,nstate],
 * xpos_out:[n,(nbody-1)*3] / xquat_out:[n,(nbody-1)*4] optional.  What a driver uses to seed many synthetic or
 * randomised start states (the reference jits env.reset and calls it once per state: dial_mpc/core/dial_core.py:222,227). */
int dial_env_reset_batch(dial_ctx* ctx, const float* qpos, const float* qvel, float* states,
                         float* xpos_out, float* xquat_out, int n, void* stream);

/* timing hook for bench.py: average duration in ms of the last n launches of the rollout
 * kernel, measured with hipEvents on the launch stream (enable with dial_set_timing).   */
int dial_set_timing(dial_ctx* ctx, int enable);
int dial_get_rollout_ms(dial_ctx* ctx, double* total_ms, int* launches);

/* Sticky status of the context's ASYNCHRONOUS work, readable without synchronising: DIAL_OK, or DIAL_ERR_HIP once if an
 * earlier rollout launch gave up (a piece of the mean-trajectory relay that never got its turn -- bounded wait, ~0.2 s --
 * marks the launch invalid instead of hanging the GPU; the mean trajectory's reward then carries a NaN bit pattern).
 * Every compute entry point performs the same check first.  Drivers call it after their own synchronisation point,
 * before they publish a plan (deploy/dial_plan.py, core/dial_core.py).                                              */
int dial_status(dial_ctx* ctx);

/* Diagnostics for the parity tests (NULL, the default, in production): while `trace` is set, every rollout launch of the
 * context also writes the packed state [qpos|qvel|qacc_warmstart|info] AFTER each env.step -- trace:[rows,Hsample+1,
 * dial_state_size(nq,nv)] device floats, row = rollout index of the launch; launches with more rollouts than `rows` fail
 * with DIAL_ERR_ARG.  This exposes the per-step qacc_warmstart and env info, so that the oracle can be restarted from the
 * GPU's OWN state after step t and compared with the GPU's step t + 1 (tests/conftest.py: transition_parity).          */
int dial_set_state_trace(dial_ctx* ctx, float* trace, int rows);

/* ABI self-description used by tests: sizeof of the three structs. */
int dial_abi_sizes(int* model_bytes, int* task_bytes, int* cfg_bytes);

#ifdef __cplusplus
}
#endif
#endif /* DIAL_MPC_H */

// cmodel.h -- dimension-specialised, compact copies of (dial_model, dial_task, derived tables).
//
// The C ABI carries capacity-sized structs (include/dial_mpc.h).  The kernels are instantiated per robot
// with compile-time dimensions (Go2, H1) so that (a) the constants fit in LDS next to the per-sample state
// (CModel<Go2> is 5.4 KB instead of 8.9 KB + 3 KB + 2.6 KB), (b) every LDS address is an immediate offset and
// (c) loops over dofs / rows unroll, which the register-resident linear algebra needs.  A generic
// instantiation (capacity dimensions, constants read from global memory) serves any other model.
#pragma once
#include <stdint.h>
#include <type_traits>
#include "../../include/dial_mpc.h"

// ---- compile-time dof-tree topology of a robot (enables branch-induced sparsity in the factorisations)
struct TopoDense {
  static constexpr bool dense = true;
  static constexpr bool anc(int, int) { return true; }
};
template <int N>
struct ParentTable { int8_t p[N]; };
template <int N>
constexpr bool topo_anc(const ParentTable<N>& t, int i, int j) {   // is dof j an ancestor-or-self of dof i ?
  while (i >= 0) {
    if (i == j) return true;
    i = t.p[i];
  }
  return false;
}
struct TopoGo2 {   // free base (0..5) + 4 legs of 3 hinges hanging off dof 5
  static constexpr bool dense = false;
  static constexpr ParentTable<18> T{{-1, 0, 1, 2, 3, 4, 5, 6, 7, 5, 9, 10, 5, 12, 13, 5, 15, 16}};
  static constexpr bool anc(int i, int j) { return topo_anc(T, i, j); }
};
struct TopoH1 {    // free pelvis, 2 legs of 5, torso (dof 16), 2 arms of 4 hanging off the torso
  static constexpr bool dense = false;
  static constexpr ParentTable<25> T{{-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 5, 11, 12, 13, 14, 5, 16, 17, 18, 19, 16, 21, 22, 23}};
  static constexpr bool anc(int i, int j) { return topo_anc(T, i, j); }
};
struct TopoAllegro {  // free object (0..5) and four independent 4-hinge fingers; the palm is welded to the world
  static constexpr bool dense = false;
  static constexpr ParentTable<22> T{{-1, 0, 1, 2, 3, 4, -1, 6, 7, 8, -1, 10, 11, 12, -1, 14, 15, 16, -1, 18, 19, 20}};
  static constexpr bool anc(int i, int j) { return topo_anc(T, i, j); }
};
struct TopoH1PushCrate {   // the H1's tree (TopoH1) and the crate's slide dof (25), a root of its own
  static constexpr bool dense = false;
  static constexpr ParentTable<26> T{{-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 5, 11, 12, 13, 14, 5, 16, 17, 18, 19, 16, 21, 22, 23, -1}};
  static constexpr bool anc(int i, int j) { return topo_anc(T, i, j); }
};
struct TopoH1Loco {  // free pelvis, 2 legs of 5, torso yaw (dof 16); the arms are welded to the torso
  static constexpr bool dense = false;
  static constexpr ParentTable<17> T{{-1, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 5, 11, 12, 13, 14, 5}};
  static constexpr bool anc(int i, int j) { return topo_anc(T, i, j); }
};

// ---- row layout of the register-resident position / velocity stage (smooth_rows.h): robots that are one kinematic tree
// under a free root.  maxd = deepest chain below the root (0: the robot does not use the stage); merge_dst / merge_src = the
// lanes of the one body that lies on two chains (owner / copy), -1: none.  rows_build (below) derives the same from the model.
template <class Topo>
struct RowsOf { static constexpr int maxd = 0, merge_src = -1, merge_dst = -1; static constexpr bool static_root = false; };
template <>
struct RowsOf<TopoH1> { static constexpr int maxd = 5, merge_src = 49, merge_dst = 33; static constexpr bool static_root = false; };       // torso: row 2 owns it, row 3 copies it
template <>
struct RowsOf<TopoH1PushCrate> { static constexpr int maxd = 5, merge_src = 49, merge_dst = 33; static constexpr bool static_root = false; };   // (+ the crate: a tree of its own, outside the rows)
template <>
struct RowsOf<TopoH1Loco> { static constexpr int maxd = 5, merge_src = 49, merge_dst = 33; static constexpr bool static_root = false; };   // (arms welded: chain members without dofs)
// Allegro: the free object on its own (body 1), four fingers of four hinges + a welded tip under the palm, which is welded to the world
template <>
struct RowsOf<TopoAllegro> { static constexpr int maxd = 5, merge_src = -1, merge_dst = -1; static constexpr bool static_root = true; };
#define ROWS_BODY 1    /* the lane carries a body of its chain (copies included)            */
#define ROWS_OWNER 2   /* ... and is the one that stores it and counts its mass / forces    */
#define ROWS_JOINT 4   /* the body has a hinge (copies included: they propagate velocities) */
#define ROWS_DOF 8     /* owner of a hinge: writes the dof's row of M, qfrc_smooth, cdof    */
#define ROWS_TDOF 16   /* stands for one of the free body's six dofs                        */
#define ROWS_SOLO 32   /* static-root layout: the free body, a tree of its own (lane 15)   */
struct RowTab {
  uint8_t body[64], flags[64], dof[64], geom[64][2], site[64];
};
struct RowTabNone {};
// SQUARE: M / H live in LDS as full nv x S squares (S = nv rounded up to 4: every row is ds_read_b128-able and no
// index arithmetic or sparsity masks are needed when rows are fetched into registers), the contact Jacobian as
// dof-major pyramid rows (J^T[i][4c + e]) and H is assembled from a contact-sparse work list (NHI = its capacity).
// Costs LDS, so it is opt-in per robot.
// ELL: elliptic friction cones (condim rows per contact, NE = NE_ELL_ rows in total): the contact Jacobian is stored
// compactly (per contact only the dofs that move its two bodies, JCW_ words in total) and the Newton solver is
// solver_cone.h instead of solver_reg.h; H couples all dofs, so it is factorised with the dense elimination order.
// GEN: the GENERIC FEATURE SET -- any pyramidal model within the capacities: run-time contact compaction (the constraint
// section works on the contacts that touch), the LDS Newton solver with a dense register L D L^T, box narrow phases, dry-friction
// rows, the crate tasks, a capped LDS workspace with an overflow area.  Independent of STATIC (compile-time dimensions,
// constants staged in LDS): DimsMax is generic at run-time dimensions with its constants in global memory; DimsGo2Crate /
// DimsH1PushCrate (round 4) are the crate scenes' own instantiations -- generic features, compile-time dimensions, constants in
// LDS shared by the nine wavefronts of a workgroup.
template <bool STATIC, int NQ_, int NV_, int NU_, int NB_, int NJ_, int NG_, int NS_, int NC_, int NL_,
          class Topo_ = TopoDense, bool SQUARE_ = false, int NHI_ = 2, bool ELL_ = false, int NE_ELL_ = 0, int JCW_ = 4,
          bool GEN_ = !STATIC, int NFRI_ = -1, bool HDENSE_ = true>
struct Dims {
  using Topo = Topo_;
  static constexpr bool is_static = STATIC;
  static constexpr bool gen = GEN_;
  static constexpr int NFRI = GEN_ ? NFRI_ : 0;   // dry-friction rows: compile-time count, -1 = run time (capacity-dimension kernel)
  static constexpr int NVP = STATIC ? NV_ : DIAL_MAX_V;   // dimension of the generic solver's register L D L^T
  // generic feature set at compile-time dimensions: M is factorised with the dof tree's fill-free order (Topo); H = M + J^T D J
  // keeps that sparsity as long as every contact is against the world (crate climb: the crate is welded) and loses it when a
  // contact couples two moving bodies (push crate: robot against the sliding crate)
  static constexpr bool h_dense = HDENSE_;
  static constexpr int NQ = NQ_, NV = NV_, NU = NU_, NB = NB_, NJ = NJ_, NG = NG_, NS = NS_, NC = NC_, NL = NL_;
  static constexpr bool ell = ELL_;
  static constexpr int NE = ELL_ ? NE_ELL_ : NL_ + 4 * NC_;
  static constexpr int JCW = JCW_;            // words of the compact contact Jacobian (elliptic models)
  static constexpr int NCE = ELL_ ? NC_ : 1;  // extent of the per-contact tables only elliptic models carry
  static constexpr int NVE = ELL_ ? NV_ : 1;
  static constexpr int NCD = 10;              // max dofs that move the two bodies of one contact (Allegro: 6 + 4)
  static constexpr int NDC = 8;               // max contacts that touch one dof (Allegro: 6)
  static constexpr int NSA = NS_ > 0 ? NS_ : 1;   // extent of the site tables (no zero-length arrays)
  static constexpr bool square = SQUARE_;
  static constexpr int NHI = NHI_;            // capacity of the H work list (square layout)
  static constexpr int S = (NV_ + 3) & ~3;    // row stride of the square matrices
  static constexpr int T = 4 * NC_;           // row stride of the dof-major pyramid Jacobian
  static constexpr int NLP = (NL_ + 3) & ~3;  // contact rows start here in the 16-byte aligned row-weight array
  static constexpr int NTRI = NV_ * (NV_ + 1) / 2;
  static constexpr int NANC = 12;   // max dofs on a root-to-body path (Go2: 9, H1: 11)
  static constexpr int NCHAIN = 8;  // max root-to-leaf chains (Go2: 4 legs, H1: 2 legs + 2 arms)
  static constexpr int CHAINLEN = 8;  // max bodies on a chain (Go2: 4, H1: 6)
  // the tables only the LDS-phase version of forward()'s position / velocity stage reads (lower-triangle entry list, ancestor
  // lists): dropped from the LDS-resident constants of the robots whose stage runs in registers (smooth_quad.h / smooth_rows.h)
#ifdef DIAL_NO_QUAD
  static constexpr bool quad_stage = false;
#else
  static constexpr bool quad_stage = !GEN_ && !ELL_ && SQUARE_ && std::is_same<Topo_, TopoGo2>::value;
#endif
#ifdef DIAL_NO_ROWS
  static constexpr bool rows_stage = false;
#else
  static constexpr bool rows_stage = !GEN_ && SQUARE_ && RowsOf<Topo_>::maxd > 0;
#endif
  // the row layout under the GENERIC feature set (push crate: the H1's tree in registers; the crate on its slide joint -- a second
  // tree of one body -- as a one-lane phase; geom frames / collisions / rows generic), M as the packed lower triangle
#if defined(DIAL_NO_ROWS) || defined(DIAL_NO_ROWS_GEN)
  static constexpr bool rows_gen = false;
#else
  static constexpr bool rows_gen = GEN_ && STATIC && !ELL_ && !SQUARE_ && RowsOf<Topo_>::maxd > 0;
#endif
  // the quadruped register stage WITHOUT its fused foot contacts, for the generic feature set on the Go2's tree (crate climb:
  // 17 geoms, 52 candidate contacts): bodies / dofs in registers, then the generic geom frames, collisions and constraint rows
#if defined(DIAL_NO_QUAD) || defined(DIAL_NO_QUAD_GEN)
  static constexpr bool quad_gen = false;
#else
  static constexpr bool quad_gen = GEN_ && STATIC && !ELL_ && !SQUARE_ && std::is_same<Topo_, TopoGo2>::value;
#endif
  static constexpr bool phase_tabs = !quad_stage && !rows_stage && !rows_gen && !quad_gen;
  // the rollout's controls, joint targets and gait clock as tables built once in its prologue (rollout_driver.h; derived.h: Ws::jtab),
  // the per-step outputs stored by the phases that produce them: the instantiations whose position / velocity stage leaves the A1
  // temporaries of the workspace unused (the tables live there) and whose control step is ONE physics step (dial_create checks)
#ifdef DIAL_NO_PRE_CTRL
  static constexpr bool pre_ctrl = false;
#else
  static constexpr bool pre_ctrl = quad_stage;
#endif
};
using DimsGo2 = Dims<true, 19, 18, 12, 14, 13, 5, 5, 4, 12, TopoGo2, true, 192>;
using DimsH1 = Dims<true, 26, 25, 19, 21, 20, 3, 3, 4, 19, TopoH1, true, 256>;
using DimsH1Loco = Dims<true, 18, 17, 11, 21, 12, 5, 3, 8, 11, TopoH1Loco, true, 192>;
// Allegro: 19 contacts (14 x condim 3 + 5 x condim 6 = 72 rows) + 16 limits; compact Jacobian 8x3x4 + 6x3x8 + 6x6 + 4x6x10
using DimsAllegro = Dims<true, 23, 22, 16, 23, 17, 6, 0, 19, 16, TopoAllegro, true, 64, true, 88, 516>;
// the crate scenes: Go2 + floor + welded crate (52 candidate contacts), H1 + crate on a slide joint (28)
using DimsGo2Crate = Dims<true, 19, 18, 12, 15, 13, 17, 5, 52, 12, TopoGo2, false, 2, false, 0, 4, true, 0, false>;
using DimsH1PushCrate = Dims<true, 27, 26, 19, 22, 21, 9, 3, 28, 19, TopoH1PushCrate, false, 2, false, 0, 4, true, 1, true>;
using DimsMax = Dims<false, DIAL_MAX_Q, DIAL_MAX_V, DIAL_MAX_U, DIAL_MAX_BODY, DIAL_MAX_JNT, DIAL_MAX_GEOM,
                     DIAL_MAX_SITE, DIAL_MAX_CON, DIAL_MAX_LIM>;

// compile-time loop: f(std::integral_constant<int, I>) for I in [B, E)
#include <type_traits>
template <int B, int E, class F>
#if defined(__HIPCC__)
__host__ __device__ __forceinline__
#else
inline
#endif
void static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

// What only the generic instantiation carries (box narrow phases and the crate task: include/dial_mpc.h).  An EMPTY base for
#define DIAL_KBI_WORDS 12   /* rows of the impedance table: nine words, padded to whole 16-byte fetches */
#define DIAL_KBI_ROWS 8     /* distinct (solref, solimp) rows a dimension-specialised instantiation holds */

// the dimension-specialised instantiations: their constants live in LDS, which is on the residency edge.
template <class D, bool GENERIC = D::gen>
struct CModelGeneric {};
template <class D>
struct CModelGeneric<D, true> {
  int32_t con_sub[D::NC];
  int32_t con_bbslot[D::NC];   // box-box candidates: which lane-private slice of the polygon scratch (box_collide.h: box_box)
  int32_t crate_contact[DIAL_MAX_FEET];
  float crate_region[6], head_vec[3];
  // dry friction (joint frictionloss): rows [nlim, nlim + nfri) of the constraint list, and the push-crate task's contacts
  int32_t nfri, fri_dof[DIAL_MAX_FRI], dof_frirow[D::NV];
  float fri_loss[DIAL_MAX_FRI];
  uint8_t fri_kbi[(DIAL_MAX_FRI + 3) & ~3];   // row of CModel::kbi_tab
  int32_t pc_foot_contact[2][2], pc_wanted[2], pc_n_unwanted, pc_unwanted[16];
  float pc_wanted_zmax;
};

// Everything one env.step reads that is constant across samples and steps.
template <class D_>
struct CModel : CModelGeneric<D_> {
  using D = D_;
  // ---- scalars
  int32_t nq, nv, nu, nbody, njnt, ngeom, nsite, ncon, nlim, nefc;
  int32_t iterations, ls_iterations, ls_rule, nlevel, ntri;
  float timestep, gravity[3], tolerance, ls_tolerance, impratio, meaninertia;
  // ---- bodies
  int32_t body_parent[D::NB], body_jntadr[D::NB], body_jntnum[D::NB], body_dofadr[D::NB], body_dofnum[D::NB];
  int32_t body_subtree_end[D::NB], body_rootid[D::NB];
  uint32_t body_ancmask[D::NB];
  int32_t body_nanc[D::NB];              // number of ancestor-or-own dofs of body b ...
  uint8_t body_anc[D::phase_tabs ? D::NB : 1][D::NANC];      // ... and their indices, root first (phase version of forward() only)
  int32_t body_flags[D::NB];             // bit 0: body_quat is identity, bit 1: all joint anchors at the body origin, bit 2: free joint
  int32_t kin_fast;                      // every body has at most one joint: parent-independent local transforms (forward())
  float body_pos[D::NB][3], body_quat[D::NB][4], body_ipos[D::NB][3], body_iquat[D::NB][4];
  float body_mass[D::NB], body_inertia[D::NB][3], body_invweight0[D::NB];
  int32_t lvl_start[D::NB + 1], lvl_body[D::NB], body_depth[D::NB];
  // root-to-leaf chains: prefix sums along a chain give every ancestor sum (cvel, cacc) in one sweep
  int32_t nchain, chain_len[D::NCHAIN];
  uint8_t chain_body[D::NCHAIN][D::CHAINLEN];
  // subtree sums (crb, cfrc) by suffix sums up the part of each chain that belongs to it alone (from
  // chain_excl[c] to the leaf), then the few bodies no tail covers (the branching bodies and everything above
  // them), deepest first, from their children; nshared < 0: not available
  int32_t chain_excl[D::NCHAIN];
  int32_t nshared, shared_body[4], shared_nchild[4];
  uint8_t shared_child[4][4];
  // ---- joints
  int32_t jnt_type[D::NJ], jnt_qposadr[D::NJ], jnt_dofadr[D::NJ], jnt_bodyid[D::NJ];
  float jnt_pos[D::NJ][3], jnt_axis[D::NJ][3], jnt_range[D::NJ][2];
  // The position-independent part of constraint._kbi (k, b, the impedance curve's constants) of the limit rows / contacts / dry-friction
  // rows, evaluated ONCE on the host from solref / solimp / timestep (derived.h: kbi_row) instead of in every physics step by every row
  // lane: k b dmin dmax | 1/width mid 1/mid^(p-1) 1/(1-mid)^(p-1) | power -- round 6, ~35 issue slots (five of them v_rcp) per call.
  // Rows with the same solref / solimp share one table row (a robot has two or three distinct ones; the constants live in LDS, which
  // is on the residency edge: DESIGN.md 5c) -- jnt_kbi / con_kbi / fri_kbi hold the row's index.  More distinct rows than NKBI: the
  // capacity-dimension kernel (dial_create: kbi_unique_rows).
  static constexpr int NKBI = D::is_static ? DIAL_KBI_ROWS : D::NJ + D::NC + DIAL_MAX_FRI;
  alignas(16) float kbi_tab[NKBI][DIAL_KBI_WORDS];
  uint8_t jnt_kbi[(D::NJ + 3) & ~3], con_kbi[(D::NC + 3) & ~3];
  float jnt_margin[D::NJ], qpos0[D::NQ];
  // ---- dofs
  int32_t dof_bodyid[D::NV], dof_jntid[D::NV], dof_act[D::NV], dof_limrow[D::NV];
  uint32_t dof_ancmask[D::NV];           // bit j: dof j is an ancestor-or-self of dof i
  uint32_t dof_descmask[D::NV];          // bit j: dof j is a descendant-or-self of dof i
  int32_t dof_blk0[D::NV], dof_blk1[D::NV];   // dofs of the same kinematic tree: the non-zero columns of row i of M
  float dof_armature[D::NV], dof_damping[D::NV], dof_invweight0[D::NV];
  uint16_t tri[D::phase_tabs ? D::NTRI + (D::NTRI & 1) : 2];   // (phase version of forward() and the generic solver only)
  // H work list (square layout), derived from dial_derived::hitem and padded with no-op items to whole passes:
  //   hrec[it][0] = i*T | (j*T) << 10 | c0 << 20 | c1 << 23 | c2 << 26 | c3 << 29      (word offsets into J^T)
  //   hrec[it][1] = i*S+j | (j*S+i) << 10 | limit-row weight index << 20 (NE+3: a zero word) | pcode << 26 |
  //                 writer << 28 | n << 29
  int32_t nhitem;
  uint8_t hpass_n[8];
  uint32_t hrec[D::NHI][2];
  // ---- geoms / sites / contacts / limits / actuators
  int32_t geom_bodyid[D::NG];
  float geom_pos[D::NG][3], geom_quat[D::NG][4], geom_size[D::NG][3];
  int32_t quad_site_is_geom;             // quadruped stage: every foot's site sits at its foot geom's centre (site_pos[1 + r] == geom_pos[1 + r]): one rotation serves both
  float geom0_normal[3];                 // third column of geom 0's rotation (the floor plane's normal: smooth_quad.h evaluated it per step)
  int32_t site_bodyid[D::NSA];
  float site_pos[D::NSA][3], site_quat[D::NSA][4];
  int32_t con_kind[D::NC], con_geom1[D::NC], con_geom2[D::NC], con_body1[D::NC], con_body2[D::NC];
  float con_friction[D::NC][5], con_margin[D::NC];
  // pyramidal cones: the rows' invweight (_efc_contact_pyramidal) | elliptic: body_invweight0[body1] + [body2], the same / impratio
  float con_invw[D::NC][D::ell ? 2 : 1];
  // elliptic models: rows / dofs of every contact and the contacts of every dof (static, from the body tree)
  int32_t cone, eulerdamp;
  int32_t con_dim[D::NCE], con_adr[D::NCE];      // condim and first constraint row
  int32_t con_ndof[D::NCE], con_joff[D::NCE];    // dofs that move body1 or body2; word offset of J_c (dim x ndof, row-major)
  uint8_t con_dof[D::NCE][(D::NCD + 3) & ~3];   // rows padded to whole words: the row products fetch them as 32-bit words
  int32_t dof_ncon[D::NVE];
  uint16_t dof_con[D::NVE][D::NDC];              // contact | (index of the dof in the contact's dof list) << 8
  uint8_t con_dofpos[D::NCE][(D::NVE + 3) & ~3]; // index of dof i in the contact's dof list, 255 = the dof does not move it
  uint8_t pair_a[D::ell ? 56 : 4], pair_b[D::ell ? 56 : 4];   // the unordered pairs (a >= b) of a contact's <= 10 dofs
  int32_t lim_jnt[D::NL];
  int32_t act_qposadr[D::NU], act_ctrllimited[D::NU], act_isposition[D::NU];
  float act_gear[D::NU], act_kp[D::NU], act_ctrlrange[D::NU][2];
  // ---- task (dial_task; the seq-jump stage tables stay in the global dial_task)
  int32_t kind, n_frames, position_control, torso_x, upright_x, nfeet, n_stage, feet_site[DIAL_MAX_FEET], randomize_tasks, n_cmd;
  float dt, action_scale, foot_radius, gait_duty, gait_cadence, gait_amp, gait_phase[DIAL_MAX_FEET];
  float cmd_vel[3], cmd_ang_vel[3], ramp_up_time, done_height, jump_dt, init_pos_tar[3], init_ang_vel_tar[3];
  float kp[D::NU], kd[D::NU], joint_range[D::NU][2], phys_range[D::NU][2], tau_range[D::NU][2], joint_offset[D::NU];
  // ---- lane layout of the register-resident position / velocity stage (smooth_rows.h), robots that use it
  typename std::conditional<((!D::gen && D::square && RowsOf<typename D::Topo>::maxd > 0) || D::rows_gen), RowTab, RowTabNone>::type rows;
};

// ---- runtime / compile-time dimension accessors
#if defined(__HIPCC__)
#define CM_HD __host__ __device__ inline
#else
#define CM_HD inline
#endif
#define CM_DIM(fn, FIELD, field)                                   \
  template <class M>                                               \
  CM_HD constexpr int fn(const M* m) {                             \
    if constexpr (M::D::is_static) return M::D::FIELD;             \
    else return m->field;                                          \
  }
CM_DIM(dim_nq, NQ, nq)
CM_DIM(dim_nv, NV, nv)
CM_DIM(dim_nu, NU, nu)
CM_DIM(dim_nb, NB, nbody)
CM_DIM(dim_nj, NJ, njnt)
CM_DIM(dim_ng, NG, ngeom)
CM_DIM(dim_ns, NS, nsite)
CM_DIM(dim_nc, NC, ncon)
CM_DIM(dim_nl, NL, nlim)
CM_DIM(dim_ne, NE, nefc)
CM_DIM(dim_ntri, NTRI, ntri)
#undef CM_DIM
// dofs with dry friction: the generic instantiation only (the dimension-specialised robots have none)
template <class M>
CM_HD constexpr int dim_nf(const M* m) {
  if constexpr (M::D::NFRI >= 0) return M::D::NFRI;
  else return m->nfri;
}

// Host: the row layout of smooth_rows.h for a model (false: the model is not one tree of hinge / welded bodies under a free
// root with at most four chains of <= maxd bodies, at most one body shared by two chains (directly below the root), at most two
// geoms and one site per body, plane-sphere / plane-capsule contacts).
// extra_trees (the generic feature set's use of the layout): bodies that are roots of their own -- a slide joint on the world, no
// children, no sites (the push-crate scene's crate) -- stay outside the rows (smooth_rows.h runs them as a one-lane phase); geoms are
// not assigned to lanes at all (the generic geom phase computes every frame) and any contact kind is allowed.
static inline bool rows_build(const dial_model* m, RowTab* t, int maxd, int* merge_src, int* merge_dst, bool static_root = false,
                              bool extra_trees = false) {
  for (int l = 0; l < 64; l++) { t->body[l] = 0; t->flags[l] = 0; t->dof[l] = 0; t->geom[l][0] = 255; t->geom[l][1] = 255; t->site[l] = 255; }
  *merge_src = -1; *merge_dst = -1;
  if (maxd < 1 || maxd > 7 || m->nbody < 2 || m->nbody > 64) return false;
  if (static_root) {
    // body 1 = the free body on its own, body 2 = the chains' root, welded to the world; everything else hangs off body 2
    if (m->nbody < 3 || m->body_parent[1] != 0 || m->body_jntnum[1] != 1 || m->body_jntadr[1] != 0 || m->jnt_type[0] != DIAL_JNT_FREE ||
        m->jnt_qposadr[0] != 0 || m->jnt_dofadr[0] != 0 || m->body_dofadr[1] != 0 || m->body_rootid[1] != 1 ||
        m->body_parent[2] != 0 || m->body_jntnum[2] != 0 || m->body_rootid[2] != 2)
      return false;
    for (int b = 3; b < m->nbody; b++) {
      if (m->body_parent[b] < 2 || m->body_rootid[b] != 2 || m->body_jntnum[b] > 1) return false;
      if (m->body_jntnum[b] == 1 && m->jnt_type[m->body_jntadr[b]] != DIAL_JNT_HINGE) return false;
    }
    int owner[64];
    for (int b = 0; b < 64; b++) owner[b] = -1;
    int row = 0;
    for (int b = 2; b < m->nbody; b++) {
      bool leaf = true;
      for (int c2 = b + 1; c2 < m->nbody; c2++) leaf = leaf && m->body_parent[c2] != b;
      if (!leaf) continue;
      int path[64], len = 0;
      for (int bb = b; bb > 0 && len < 64; bb = m->body_parent[bb]) path[len++] = bb;
      if (row >= 4 || len - 1 > maxd) return false;
      for (int d = 0; d < len; d++) {
        const int bb = path[len - 1 - d], l = 16 * row + d;
        t->body[l] = (uint8_t)bb;
        t->flags[l] |= ROWS_BODY;
        if (owner[bb] < 0) { owner[bb] = l; t->flags[l] |= ROWS_OWNER; }
        else if (d >= 1) return false;   // no body on two chains in this form
        if (d >= 1 && m->body_jntnum[bb] == 1) {
          t->flags[l] |= ROWS_JOINT | ROWS_DOF;
          t->dof[l] = (uint8_t)m->body_dofadr[bb];
        }
      }
      row++;
    }
    t->body[15] = 1; t->flags[15] = ROWS_BODY | ROWS_OWNER | ROWS_SOLO; owner[1] = 15;
    for (int k = 0; k < 6; k++) { t->flags[8 + k] = ROWS_TDOF; t->dof[8 + k] = (uint8_t)k; }
    for (int g = 0; g < m->ngeom; g++) {
      const int b = m->geom_bodyid[g], l = b == 0 ? 14 : owner[b];
      if (l < 0) return false;
      if (t->geom[l][0] == 255) t->geom[l][0] = (uint8_t)g;
      else if (t->geom[l][1] == 255) t->geom[l][1] = (uint8_t)g;
      else return false;
    }
    for (int si = 0; si < m->nsite; si++) {
      const int b = m->site_bodyid[si];
      if (b == 0 || owner[b] < 0 || t->site[owner[b]] != 255) return false;
      t->site[owner[b]] = (uint8_t)si;
    }
    for (int c = 0; c < m->ncon; c++) if (m->con_kind[c] >= DIAL_CON_PLANE_BOX) return false;
    return true;
  }
  if (m->body_parent[1] != 0 || m->body_jntnum[1] != 1 || m->body_jntadr[1] != 0 || m->jnt_type[0] != DIAL_JNT_FREE ||
      m->jnt_qposadr[0] != 0 || m->jnt_dofadr[0] != 0 || m->body_dofadr[1] != 0 || m->body_rootid[1] != 1)
    return false;
  int nb_tree = m->nbody;   // bodies [1, nb_tree) form the tree under the free root
  if (extra_trees) {
    while (nb_tree > 2 && m->body_parent[nb_tree - 1] == 0) {
      const int b = nb_tree - 1;
      if (m->body_jntnum[b] != 1 || m->jnt_type[m->body_jntadr[b]] != DIAL_JNT_SLIDE || m->body_rootid[b] != b || m->body_dofnum[b] != 1) return false;
      nb_tree--;
    }
    if (m->nbody - nb_tree > 1) return false;   // (one such body: smooth_rows.h's solo phase)
    for (int si = 0; si < m->nsite; si++) if (m->site_bodyid[si] >= nb_tree) return false;
  }
  for (int b = 2; b < nb_tree; b++) {
    if (m->body_parent[b] < 1 || m->body_rootid[b] != 1 || m->body_jntnum[b] > 1) return false;
    if (m->body_jntnum[b] == 1 && m->jnt_type[m->body_jntadr[b]] != DIAL_JNT_HINGE) return false;
  }
  int owner[64];
  for (int b = 0; b < 64; b++) owner[b] = -1;
  int row = 0;
  for (int b = 1; b < nb_tree; b++) {
    bool leaf = true;
    for (int c2 = b + 1; c2 < m->nbody; c2++) leaf = leaf && m->body_parent[c2] != b;
    if (!leaf) continue;
    int path[64], len = 0;
    for (int bb = b; bb > 0 && len < 64; bb = m->body_parent[bb]) path[len++] = bb;
    if (row >= 4 || len - 1 > maxd) return false;
    for (int d = 0; d < len; d++) {
      const int bb = path[len - 1 - d], l = 16 * row + d;
      t->body[l] = (uint8_t)bb;
      t->flags[l] |= ROWS_BODY;
      if (owner[bb] < 0) { owner[bb] = l; t->flags[l] |= ROWS_OWNER; }
      else if (d >= 1) {   // a body on two chains
        if (d != 1 || *merge_src >= 0) return false;
        *merge_src = l; *merge_dst = owner[bb];
      }
      if (d >= 1 && m->body_jntnum[bb] == 1) {
        t->flags[l] |= ROWS_JOINT;
        t->dof[l] = (uint8_t)m->body_dofadr[bb];
        if (t->flags[l] & ROWS_OWNER) t->flags[l] |= ROWS_DOF;
      }
    }
    row++;
  }
  for (int k = 0; k < 6; k++) { t->flags[8 + k] = ROWS_TDOF; t->dof[8 + k] = (uint8_t)k; }
  for (int g = 0; g < m->ngeom && !extra_trees; g++) {
    const int b = m->geom_bodyid[g], l = b == 0 ? 14 : owner[b];   // lane 14: the world's geoms (identity pose)
    if (l < 0) return false;
    if (t->geom[l][0] == 255) t->geom[l][0] = (uint8_t)g;
    else if (t->geom[l][1] == 255) t->geom[l][1] = (uint8_t)g;
    else return false;
  }
  for (int si = 0; si < m->nsite; si++) {
    const int b = m->site_bodyid[si];
    if (b == 0 || owner[b] < 0 || t->site[owner[b]] != 255) return false;
    t->site[owner[b]] = (uint8_t)si;
  }
  for (int c = 0; c < m->ncon && !extra_trees; c++) {
    const int k = m->con_kind[c];
    if (k != DIAL_CON_PLANE_SPHERE && k != DIAL_CON_PLANE_CAPSULE_P && k != DIAL_CON_PLANE_CAPSULE_N) return false;
  }
  return true;
}
// Host: is the model the quadruped this layout assumes (smooth_quad.h; beyond the dof tree dims_match checks)?  World, a free trunk
// (body 1), four legs of three one-hinge bodies in depth-first order, the floor plane as geom 0, one sphere and one site per calf,
// the trunk's site first, one plane-sphere contact per foot in leg order.
// the body / joint / dof tree alone: world, a free trunk (body 1), four legs of three one-hinge bodies in depth-first order
static inline bool quad_tree_fits(const dial_model* m) {
  if (m->nbody < 14 || m->njnt != 13 || m->nv != 18 || m->nq != 19) return false;
  if (m->body_parent[1] != 0 || m->body_jntnum[1] != 1 || m->body_jntadr[1] != 0 || m->jnt_type[0] != DIAL_JNT_FREE ||
      m->jnt_qposadr[0] != 0 || m->jnt_dofadr[0] != 0 || m->body_dofadr[1] != 0 || m->body_rootid[1] != 1)
    return false;
  for (int r = 0; r < 4; r++)
    for (int k = 0; k < 3; k++) {
      const int b = 2 + 3 * r + k, j = b - 1;
      if (m->body_parent[b] != (k == 0 ? 1 : b - 1) || m->body_jntnum[b] != 1 || m->body_jntadr[b] != j ||
          m->jnt_type[j] != DIAL_JNT_HINGE || m->jnt_qposadr[j] != b + 5 || m->jnt_dofadr[j] != b + 4 || m->body_dofadr[b] != b + 4)
        return false;
    }
  return true;
}
// the generic feature set on that tree (Dims::quad_gen): the trunk's site first, one site per calf in leg order, every further
// body welded to the world (the crate: a constant pose, written once per kernel)
static inline bool quad_gen_fits(const dial_model* m) {
  if (m->nsite != 5 || m->site_bodyid[0] != 1) return false;
  for (int r = 0; r < 4; r++)
    if (m->site_bodyid[1 + r] != 4 + 3 * r) return false;
  for (int b = 14; b < m->nbody; b++)
    if (m->body_parent[b] != 0 || m->body_jntnum[b] != 0) return false;
  return true;
}
static inline bool quad_fits(const dial_model* m) {
  if (m->nbody != 14 || m->ngeom != 5 || m->nsite != 5 || m->ncon != 4 || !quad_tree_fits(m)) return false;
  if (m->geom_bodyid[0] != 0 || m->site_bodyid[0] != 1 || (m->nlim & 3) != 0) return false;   // (limit rows fill whole 16-byte groups)
  for (int r = 0; r < 4; r++) {
    const int calf = 4 + 3 * r;
    if (m->geom_bodyid[1 + r] != calf || m->site_bodyid[1 + r] != calf || m->con_kind[r] != DIAL_CON_PLANE_SPHERE ||
        m->con_geom1[r] != 0 || m->con_geom2[r] != 1 + r || m->con_body1[r] != 0 || m->con_body2[r] != calf)
      return false;
  }
  return true;
}

// ---- host: does a model fit a static instantiation exactly?
template <class D>
static inline bool dims_match(const dial_model* m) {
  bool ok = m->nq == D::NQ && m->nv == D::NV && m->nu == D::NU && m->nbody == D::NB && m->njnt == D::NJ &&
            m->ngeom == D::NG && m->nsite == D::NS && m->ncon == D::NC && m->nlim == D::NL &&
            (D::NFRI < 0 || m->nfri == D::NFRI);   // (dry friction rows exist in the generic feature set only)
  if constexpr (!D::Topo::dense) {   // the sparse factorisations are specialised to the dof tree as well
    for (int i = 0; ok && i < D::NV; i++) ok = m->dof_parentid[i] == D::Topo::T.p[i];
  }
  if constexpr (std::is_same<D, DimsGo2>::value) ok = ok && quad_fits(m);   // its position / velocity stage is laid out for this tree
  if constexpr (D::quad_gen) ok = ok && quad_tree_fits(m) && quad_gen_fits(m);
  if constexpr (D::gen && D::is_static && !D::h_dense) {
    // H = M + J^T D J is factorised in the dof TREE's elimination order: valid only while every contact has one static side
    // (world or a body welded to it) -- a contact between two moving bodies fills H across branches.  Such a model falls
    // through to the capacity-dimension kernel (dense order).
    const auto is_static_body = [&](int b) {
      while (b > 0 && m->body_dofnum[b] == 0) b = m->body_parent[b];
      return b == 0;
    };
    for (int c = 0; ok && c < m->ncon; c++) ok = is_static_body(m->con_body1[c]) || is_static_body(m->con_body2[c]);
  }
  if constexpr ((!D::gen && D::square && RowsOf<typename D::Topo>::maxd > 0) || D::rows_gen) {   // smooth_rows.h: the layout must come out as compiled
    using RT = RowsOf<typename D::Topo>;
    RowTab t;
    int ms, md;
    ok = ok && rows_build(m, &t, RT::maxd, &ms, &md, RT::static_root, D::rows_gen) && ms == RT::merge_src && md == RT::merge_dst;
    if constexpr (D::rows_gen) ok = ok && m->nbody == D::NB && m->body_parent[D::NB - 1] == 0 && m->body_rootid[D::NB - 1] == D::NB - 1;   // exactly one solo body, the last
  }
  return ok;
}

struct dial_derived;  // derived.h

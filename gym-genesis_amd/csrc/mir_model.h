// mir_model.h — compiled (float32) device model shared by host compile code and the HIP kernels.
//
// A MirSceneSpec (include/mirigid.h) is compiled once per mir_create() into this POD,
// uploaded to HBM and read by every workgroup (it is a few KB and L2/L1 resident).
// Lane ownership in the kernels is "lane i of a 16-lane env group = dof i = body i",
// so everything here is laid out as 16-wide per-dof / per-body columns that a lane
// pulls into registers once per launch.
#pragma once
#include <stdint.h>

#include "../../include/mirigid.h"

#define MIR_G 16 /* lanes per env group; requires nbody, nv <= 16 */

/* capacity of the 16-lanes-per-env kernel (its LDS arenas and lane ownership); larger scenes take the
 * wave-per-env kernel (mir_model64.h) */
#define K16_MAX_BODY 16
#define K16_MAX_DOF 15 /* one lane of a 16-lane env group stays free for the solver */
#define K16_MAX_Q 18
#define K16_MAX_GEOM 24
#define K16_MAX_PAIR 64
#define K16_MAX_CONTACT 16
#define K16_MAX_VERT 40 /* hull vertices the 16-lane kernel keeps in LDS beside the model table (what is left of its 40 KB per workgroup) */

static_assert(K16_MAX_BODY <= MIR_G && K16_MAX_DOF <= MIR_G, "lane ownership needs nbody, nv <= group width");
static_assert(K16_MAX_GEOM <= 2 * MIR_G, "two geoms per lane");
static_assert(K16_MAX_PAIR <= 4 * MIR_G, "four pairs per lane");

// Tables that the step kernel looks up with a *dynamic* index (geom, pair, contact body, limit
// constants).  Packed so a workgroup copies them into LDS with 16-byte loads once per launch:
// dependent lookups then cost an LDS round trip (~64 cycles) instead of an L2 one (~500+).
struct ModelTab {
  float g_pos[K16_MAX_GEOM][4];   // xyz in body frame, w = friction
  float g_quat[K16_MAX_GEOM][4];  // wxyz in body frame
  float g_size[K16_MAX_GEOM][4];  // box: half extents; sphere: radius, -, -; capsule: radius, half length, -; w = bounding radius
  int32_t g_info[K16_MAX_GEOM][4];  // body, type, 0, 0
  float g_sol[K16_MAX_GEOM][8];   // solref[2], solimp[5], 0
  int32_t pair[K16_MAX_PAIR];     // g1 | g2 << 8
  uint32_t g_allow[K16_MAX_GEOM]; // bit h: geoms g and h pass the static pair filter (sweep-and-prune broadphase)
  int32_t b_info[MIR_G][4];       // dofmask, root, qadr, dofadr
  float b_invw[MIR_G];            // body_invweight0
  float d_lim[MIR_G][12];         // lo, hi, invweight0, k, b, solimp[5], 0, 0
};
static_assert(sizeof(ModelTab) % 16 == 0, "ModelTab is copied with 16-byte accesses");

// Everything one lane (= body index = dof index) needs in registers for the whole launch, packed so the prologue fetches it
// with twelve independent 16-byte loads (no load depends on another load's result, e.g. the joint axis of the dof's body
// is resolved here, not by a second trip through b_axis[d_body]).  Filled at the end of mir_compile_model.
struct LaneK16 {
  int32_t b_jtype, b_qadr, b_root, d_body;
  float b_pos[3]; uint32_t b_dofmask;
  float b_quat[4];
  float b_axis[3]; uint32_t b_submask;
  float b_ipos[3], b_mass;
  float b_inertia[6]; int32_t d_kind, d_qadr;
  int32_t d_axis_k, d_root, d_ctrl, d_uadr;
  float d_axis[3]; uint32_t d_submask;
  uint32_t d_premask, d_ancmask; int32_t d_limited /* limited && enable_joint_limit && dof exists */; float d_damping;
  float d_kp, d_kv, d_frclo, d_frchi;
  float d_mdiag; int32_t obs_qadr /* qpos address behind agent_pos column `lane` (gripper columns), else 0 */;
  int32_t scan /* bytes: d_par, d_bef, b_last (signed, -1 = none), b_next (lane behind the body's subtree, 16 = none) */;
  float d_gw /* 1 / (a lower bound of the regularised mass matrix along this dof x the object's mass): weight of g_i^2 in the early-mask bound, see DevModel::term_bound_ok */;
};
static_assert(sizeof(LaneK16) == 12 * 16, "LaneK16 is read as twelve 16-byte quantities");

struct DevModel {
  ModelTab tab;  // first member: 16-byte aligned with the allocation
  // the LaneK16 records TRANSPOSED, [quad][lane]: load k of the prologue then reads 16 lanes x 16 B = 256 contiguous bytes (the
  // same for the four env groups of a wave) instead of 64 pieces at a stride of 192 B
  float lanek_t[12][MIR_G][4];  // (16-byte aligned: sizeof(ModelTab) is a multiple of 16)
  uint64_t parents;      // the 16 parent indices, 4 bits each (pointer-jumping FK)
  uint64_t fk_free_leaf;  // 1: every free-joint body hangs off the world and carries no children (its pose is its qpos row)
  // sizes / options
  int32_t nbody, nv, nq, ngeom, npair, nu, qstride, max_contacts;
  int32_t iterations, ls_iterations, enable_collision, enable_joint_limit;
  float dt, gx, gy, gz, tolerance, ls_tolerance, meaninertia, solver_scale;
  // task extraction
  int32_t eef_body, obj_body, n_grip, obj_qadr;
  int32_t grip_qadr[MIR_MAX_GRIP];
  float reward_z;
  int32_t n_arm_q; /* number of scalar joints (arm_qpos width) */
  // ---- per body (index = lane) ----
  int32_t b_parent[MIR_G], b_jtype[MIR_G], b_dofadr[MIR_G], b_qadr[MIR_G], b_root[MIR_G], b_static[MIR_G];
  uint32_t b_dofmask[MIR_G];  /* dofs that move body b (ancestors + own) */
  uint32_t b_submask[MIR_G];  /* bodies in the subtree of b, including b (same tree only) */
  float b_pos[MIR_G][3], b_quat[MIR_G][4], b_axis[MIR_G][3], b_ipos[MIR_G][3], b_inertia[MIR_G][6];
  float b_mass[MIR_G], b_invweight0[MIR_G];
  // ---- per dof (index = lane) ----
  int32_t d_body[MIR_G], d_limited[MIR_G], d_ctrl[MIR_G], d_uadr[MIR_G], d_qadr[MIR_G], d_kind[MIR_G];
  /* d_kind: 0 revolute, 1 prismatic, 2 free-translation k, 3 free-rotation k; d_axis_k = k for free dofs */
  int32_t d_axis_k[MIR_G], d_armidx[MIR_G]; /* index into arm_qpos for scalar joints, else -1 */
  uint32_t d_premask[MIR_G]; /* dofs whose velocity is "before" dof i (for cdof_dot) */
  uint32_t d_ancmask[MIR_G]; /* ancestor-or-self dofs j <= i (non-zero pattern of M row i) */
  float d_lo[MIR_G], d_hi[MIR_G], d_damping[MIR_G], d_kp[MIR_G], d_kv[MIR_G], d_frclo[MIR_G], d_frchi[MIR_G];
  float d_mdiag[MIR_G];   /* armature + dt (damping + kv) if implicit */
  float d_armature[MIR_G];
  float d_invweight0[MIR_G];
  float d_solimp[MIR_G][5], d_k[MIR_G], d_b[MIR_G]; /* limit-row spring/damper from solref (tc clamped to 2 dt) */
  // ---- per geom ----
  int32_t g_body[K16_MAX_GEOM], g_type[K16_MAX_GEOM];
  float g_size[K16_MAX_GEOM][3], g_pos[K16_MAX_GEOM][3], g_quat[K16_MAX_GEOM][4], g_friction[K16_MAX_GEOM];
  float g_solref[K16_MAX_GEOM][2], g_solimp[K16_MAX_GEOM][5];
  // ---- candidate pairs (static filter applied) ----
  int32_t p_g1[K16_MAX_PAIR], p_g2[K16_MAX_PAIR];
  // ---- free bodies in body order (reset / re-spawn poses) ----
  int32_t nfree, free_qadr[MIR_MAX_FREE];
  int32_t has_convex;  // any sphere / capsule geom
  int32_t gj_split;    // first dof of the second kinematic tree when the model has exactly two trees with dofs and that dof is 6 or 9
                       // (the block-diagonal eliminations instantiated in mir_dev.h), else 0 = dense
  int32_t use_sap;     // candidate pairs from the sweep-and-prune over AABBs (static list too long, or MIR_BROADPHASE=sap)
  int32_t nvert;       // hull vertices in use (MIR_GEOM_HULL geoms: g_size = first vertex, count, -)
  float hverts[K16_MAX_VERT][4];  // the scene's hull vertex pool, geom frames (16-byte rows: copied into LDS by the convex instantiations)
  float g_bbox[K16_MAX_GEOM][3];  // hull geoms: half extents of the vertices' bounding box (what the rasteriser draws)
  // ---- early `terminated` bytes (mir_step.hip: the mask leaves before the solver has converged when it provably cannot change) ----
  // The solver minimises f(a) = 1/2 (a - a_s)^T Mt (a - a_s) + convex row penalties, 1-strongly convex in the Mt norm, so an iterate
  // with gradient g is within |g|_{Mt^-1} of the minimiser, and Mt >= blockdiag(diag(d_mdiag) over the jointed dofs, the free bodies'
  // own inertias): |g|^2_{Mt^-1} <= sum_i d_gw_i g_i^2.  The object's height after the step moves by dt^2 times its vertical
  // acceleration, whose distance from the minimiser's is at most term_zscale = 1 / sqrt(mass) times that norm.
  int32_t term_bound_ok;  // the bound exists: the object is a free body with its centre of mass at its origin, a childless child of the world, and every dof has a weight
  int32_t term_zlane;     // dof (= lane) of the object's vertical translation
  float term_zscale;      // 1 / sqrt(object mass)
  float pad_term;
};

struct HostConsts {
  double dof_invweight0[MIR_MAX_DOF];
  double body_invweight0[MIR_MAX_BODY];
  double meaninertia;
};

// Compile a scene spec for the 16-lane kernel.  Returns MIR_OK, MIR_E_CAPACITY if the scene does not fit this
// kernel's limits (the caller then tries the wave-per-env model), or another MIR_E_* code; fills err (<=255 chars).
int mir_compile_model(const MirSceneSpec* spec, DevModel* out, HostConsts* hc, char* err);
// Device-side episode bookkeeping + re-spawn inside a rollout launch (mir_rollout_autoreset): same rules as k_autoreset
// in mir_api.hip.  episode_len == nullptr switches it off.
struct AutoResetArgs {
  int32_t* episode_len;     // (B) in/out
  int32_t* cursor;          // (B) in/out
  const float* spawn_pool;  // (pool_len, B, nfree, 3)
  const float* obj_quat;    // (B, nfree, 4)
  const float* arm_qpos;    // (B, n_arm)
  int32_t pool_len, max_len;
};

// shared host-side helpers (mir_compile.cpp)
int mir_host_consts(const MirSceneSpec* spec, HostConsts* out, char* err);
void mir_round_spec(MirSceneSpec* spec);

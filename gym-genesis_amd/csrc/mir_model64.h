// mir_model64.h — compiled (float32) device model of the WAVE-PER-ENV step kernel (mir_step64.hip).
//
// The 16-lanes-per-env kernel (mir_model.h) owns scenes with nv <= 15; the stack tasks (arm + five free cubes,
// 36-39 dofs: /root/reference/gym_genesis/tasks/franka/cube_stack_kitchen_batch.py, so101/cube_stack_batch.py,
// scene built at tasks/utils.py:239-426,593-794) need more.  Here one wave64 serves one env and lane = "dof slot":
// the 64 lanes are four BLOCKS of 16 (= the four DPP rows); every kinematic tree lives inside one block (arm: 9 or 6
// lanes of block 0; two cubes per later block), so the joint-space inertia is block-diagonal by construction and every
// contact touches at most two blocks.  All masks below are in LANE space (uint64); dofs keep their compact order
// only at the ABI boundary (d_dof / PlumbTab).
#pragma once
#include <stdint.h>

#include "../../include/mirigid.h"
#include "mir_model.h"

#define W64 64            /* lanes per env */
#define K64_MAX_BODY 32   /* lane b < 32 owns body b */
#define K64_BLOCK_DOF 15  /* dofs per block (one lane of a DPP row stays free for the block solver) */
#define K64_QSTRIDE 64    /* floats per env in the qpos row */

struct DevModel64 {
  int32_t nbody, nv, nq, ngeom, npair, nu, max_contacts, nfree;
  int32_t iterations, ls_iterations, enable_collision, enable_joint_limit;
  float dt, gx, gy, gz, tolerance, ls_tolerance, meaninertia, solver_scale;
  // task extraction
  int32_t eef_body, obj_body, obj2_body, n_grip, reward_mode, agent_mode, agent_dim, env_dim;
  int32_t grip_qadr[MIR_MAX_GRIP];
  float reward_z, reward_xy, reward_dz;
  int32_t n_arm_q;
  int32_t arm_qadr[MIR_MAX_DOF]; /* qpos address of scalar joint k (MIR_AGENT_QPOS) */
  uint64_t lanemask;             /* lanes that carry a dof */
  int32_t d_armidx[W64];         /* index into arm_qpos for scalar-joint lanes, else -1 */
  int32_t free_qadr[MIR_MAX_FREE]; /* qpos address of free body k (body order) */
  /* qpos address of the task's object(s) when they are free bodies hanging off the world -- their world positions ARE those qpos
   * entries, so the reward is known as soon as the step is integrated -- else -1 (obj2: -1 also when the task has none) */
  int32_t obj_qadr, obj2_qadr, term_early, pad_te;
  // ---- per body (index = lane < 32) ----
  int32_t b_parent[K64_MAX_BODY], b_jtype[K64_MAX_BODY], b_dofadr[K64_MAX_BODY] /* first LANE */, b_qadr[K64_MAX_BODY];
  int32_t b_root[K64_MAX_BODY], b_static[K64_MAX_BODY], b_block[K64_MAX_BODY] /* block of the body's tree, -1 = static */;
  uint64_t b_dofmask[K64_MAX_BODY]; /* lanes of the dofs that move body b */
  uint32_t b_submask[K64_MAX_BODY]; /* bodies in the subtree of b, including b */
  float b_pos[K64_MAX_BODY][3], b_quat[K64_MAX_BODY][4], b_axis[K64_MAX_BODY][3], b_ipos[K64_MAX_BODY][3], b_inertia[K64_MAX_BODY][6];
  float b_mass[K64_MAX_BODY], b_invweight0[K64_MAX_BODY];
  // ---- per lane (dof slot) ----
  int32_t d_dof[W64]; /* compact dof index, -1 = padding lane */
  int32_t d_body[W64], d_limited[W64], d_ctrl[W64], d_uadr[W64], d_qadr[W64], d_kind[W64], d_axis_k[W64];
  uint64_t d_premask[W64], d_ancmask[W64];
  float d_lo[W64], d_hi[W64], d_damping[W64], d_kp[W64], d_kv[W64], d_frclo[W64], d_frchi[W64], d_mdiag[W64], d_armature[W64];
  float d_invweight0[W64], d_solimp[W64][5], d_k[W64], d_b[W64];
  // ---- per geom ----
  int32_t g_body[MIR_MAX_GEOM], g_type[MIR_MAX_GEOM];
  float g_size[MIR_MAX_GEOM][4], g_pos[MIR_MAX_GEOM][4] /* w = friction */, g_quat[MIR_MAX_GEOM][4], g_sol[MIR_MAX_GEOM][8] /* solref2 solimp5 */;
  // ---- candidate pairs (static filter applied) ----
  int32_t pair[MIR_MAX_PAIR]; /* g1 | g2 << 8 */
  // ---- derived tables (end of mir_compile_model64): no prologue load of the kernel depends on another load ----
  int32_t d_root[W64], d_qbase[W64], d_lbase[W64] /* b_root / b_qadr / b_dofadr of the lane's body */;
  int32_t obs_qadr[W64];   /* qpos address behind agent_pos column `lane` (joint / gripper columns), else 0 */
  uint32_t d_bsubmask[W64];
  float d_axis[W64][4];    /* joint axis of the lane's body */
  float b_tab[K64_MAX_BODY][8]; /* contact finish, staged in LDS: invweight0, dofmask lo, dofmask hi, block, root (int bits) */
  /* tree-scan links of the dynamics (bytes, -1 = none): [0] scan parent of dof lane l (the dof before it in its chain), [1] the dof
   * whose inclusive chain sum is the velocity in front of dof l, [2] (lane = body) the body's last moving dof, [3] (lane = body)
   * the lane behind the body's subtree, -1 when the subtree ends with its 16-lane row */
  int32_t scanw[W64];
  int32_t has_convex; /* the scene has sphere / capsule / hull geoms: the launcher picks the instantiation with the convex narrowphase */
  int32_t nvert;      /* hull vertices in use (MIR_GEOM_HULL: g_size = first vertex, count, bounding radius) */
  float hverts[MIR_MAX_VERT][4]; /* the scene's hull vertex pool, geom frames (16-byte rows: copied into LDS by the collision wave) */
  float g_bbox[MIR_MAX_GEOM][3]; /* hull geoms: half extents of the vertices' bounding box (what the rasteriser draws) */
  int32_t fk_free_leaf; /* every free-joint body hangs off the world and carries no children: its pose is its qpos row (split closing FK) */
};

// Dof-order <-> storage maps used by the plumbing kernels of mir_api.hip for BOTH step kernels
// (16-lane kernel: lane == dof, rows of 16; wave kernel: lane map, rows of 64).
struct PlumbTab {
  int32_t nv, nq, nu, nfree, narm, qst /* qpos row stride */, vst /* qvel / target / warm-start row stride */, pst /* bodies per env in the pose cache */;
  int32_t d_lane[MIR_MAX_DOF], d_uadr[MIR_MAX_DOF], d_armidx[MIR_MAX_DOF], d_qadr[MIR_MAX_DOF];
  int32_t free_qadr[MIR_MAX_FREE];
};

// Geometry table read by the rasteriser (mir_render.hip) for both kernels
struct GeomTab {
  int32_t ngeom;
  int32_t g_body[MIR_MAX_GEOM], g_type[MIR_MAX_GEOM];
  float g_size[MIR_MAX_GEOM][3], g_pos[MIR_MAX_GEOM][3], g_quat[MIR_MAX_GEOM][4];
};

// Compile a scene spec for the wave-per-env kernel.  Returns MIR_OK or a MIR_E_* code and fills err (<=255 chars).
int mir_compile_model64(const MirSceneSpec* spec, DevModel64* out, HostConsts* hc, char* err);

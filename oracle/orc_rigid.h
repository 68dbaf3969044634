/*
 * orc_rigid.h — CPU ORACLE for the gym-genesis env.step() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported, linked or
 * executed by the product (gym-genesis_amd/); only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg use it, as the checker / reported baseline.
 *
 * PARITY UNPINNED.  The arithmetic of the reference's hot path lives in the
 * third-party package `genesis-world`, required unpinned from git
 * (/root/reference/pyproject.toml:11) and absent from /root/reference and from
 * this image; the reference has no tests, golden vectors or fixtures for the
 * path.  This file restates the published MuJoCo-style
 * formulation that Genesis's rigid solver implements,
 * anchored on the reference's own call sites:
 *     scene.step()                 gym_genesis/tasks/franka/cube_pick.py:107,125
 *     control_dofs_position        gym_genesis/tasks/franka/cube_pick.py:104-105,123-124
 *     set_pos/set_quat/set_qpos    gym_genesis/tasks/franka/cube_pick.py:96-102
 *     get_pos/get_quat/get_dofs_position  gym_genesis/tasks/franka/cube_pick.py:140-146
 *     reward / terminated          gym_genesis/tasks/franka/cube_pick.py:130-135, gym_genesis/env.py:63-65
 * and is pinned by first-principles known-answer tests (tests/test_oracle_physics.py).
 *
 * Build: float64 by default (the correctness oracle); -DORC_F32 gives the
 * float32 "port" used only as bench.py's timed CPU baseline.
 */
#ifndef ORC_RIGID_H
#define ORC_RIGID_H

#include "../include/mirigid.h"

#ifdef ORC_COUNT_FLOPS
#include "orc_flops.h" /* C++ build: `real` counts its own arithmetic (liborc_flops.so, bench.py's F_step) */
#elif defined(ORC_F32)
typedef float real;
#else
typedef double real;
#endif
#ifdef __cplusplus
#define ORC_TLS thread_local
#else
#define ORC_TLS _Thread_local
#endif

#ifdef ORC_SMALL /* pick-task capacities: keeps the per-env block of the timed float32 port small and cache-resident */
#define ORC_NB 16
#define ORC_NV 15
#define ORC_NQ 18
#define ORC_NG 24
#define ORC_NP 64
#define ORC_NC 16
#else
#define ORC_NB MIR_MAX_BODY
#define ORC_NV MIR_MAX_DOF
#define ORC_NQ MIR_MAX_Q
#define ORC_NG MIR_MAX_GEOM
#define ORC_NP MIR_MAX_PAIR
#define ORC_NC MIR_MAX_CONTACT
#endif
#define ORC_NEFC (4 * ORC_NC + ORC_NV)

typedef struct OrcModel {
  int nbody, nv, nq, ngeom, npair, nu;
  MirOptions opt;
  MirTaskSpec task;
  /* bodies */
  int parent[ORC_NB], jtype[ORC_NB], dofadr[ORC_NB], qadr[ORC_NB], ndof[ORC_NB], root[ORC_NB], is_static[ORC_NB];
  real pos[ORC_NB][3], quat[ORC_NB][4], axis[ORC_NB][3], mass[ORC_NB], ipos[ORC_NB][3], inertia[ORC_NB][6];
  /* dofs */
  int dof_body[ORC_NV], dof_parent[ORC_NV], dof_limited[ORC_NV], dof_ctrl[ORC_NV], dof_uadr[ORC_NV], dof_qadr[ORC_NV];
  real range[ORC_NV][2], armature[ORC_NV], damping[ORC_NV], kp[ORC_NV], kv[ORC_NV], frc[ORC_NV][2];
  real dsolref[ORC_NV][2], dsolimp[ORC_NV][5], dof_invweight0[ORC_NV];
  /* geoms */
  int gbody[ORC_NG], gtype[ORC_NG];
  real gsize[ORC_NG][3], gpos[ORC_NG][3], gquat[ORC_NG][4], gfriction[ORC_NG], gsolref[ORC_NG][2], gsolimp[ORC_NG][5];
  int pair_g1[ORC_NP], pair_g2[ORC_NP];
  int nvert;
  real vert[MIR_MAX_VERT][3]; /* vertex pool of the MIR_GEOM_HULL geoms (gsize = first vertex, count) */
  real body_invweight0[ORC_NB];
  real meaninertia;
  real qpos0[ORC_NQ];
} OrcModel;

typedef struct OrcData {
  /* state */
  real qpos[ORC_NQ], qvel[ORC_NV], target[ORC_NV], qacc_ws[ORC_NV];
  /* kinematics */
  real xpos[ORC_NB][3], xquat[ORC_NB][4], xmat[ORC_NB][9], xipos[ORC_NB][3], cref[ORC_NB][3];
  real cdof[ORC_NV][6];
  real cinert[ORC_NB][10], crb[ORC_NB][10];
  real cvel[ORC_NB][6], cacc[ORC_NB][6], cfrc[ORC_NB][6];
  /* dynamics */
  real M[ORC_NV][ORC_NV], Mt[ORC_NV][ORC_NV];
  real qfrc_bias[ORC_NV], qfrc_passive[ORC_NV], qfrc_act[ORC_NV], qfrc_smooth[ORC_NV];
  real qacc_smooth[ORC_NV], qacc[ORC_NV];
  /* contacts */
  int ncon;
  real cpos[ORC_NC][3], cframe[ORC_NC][9], cdist[ORC_NC], cmu[ORC_NC], csolref[ORC_NC][2], csolimp[ORC_NC][5];
  int cb1[ORC_NC], cb2[ORC_NC], cg1[ORC_NC], cg2[ORC_NC];
  /* constraint rows */
  int nefc;
  real J[ORC_NEFC][ORC_NV], aref[ORC_NEFC], efcD[ORC_NEFC], efcpos[ORC_NEFC], efcforce[ORC_NEFC];
  int niter;
  int ncand; /* candidate contact points of the last collision pass BEFORE the capacity was applied (> max_contacts: thinning fired) */
  real dbg_imp[64], dbg_gn[64], dbg_alpha[64], dbg_ls[64]; /* per-iteration solver trace (dbg_ls: phi' evaluations of the line search) */
} OrcData;

/* field ids for orc_read */
enum {
  ORC_F_QPOS = 0, ORC_F_QVEL, ORC_F_TARGET, ORC_F_QACC_WS, ORC_F_XPOS, ORC_F_XQUAT, ORC_F_XIPOS, ORC_F_M,
  ORC_F_MT, ORC_F_QFRC_BIAS, ORC_F_QFRC_SMOOTH, ORC_F_QACC_SMOOTH, ORC_F_QACC, ORC_F_CPOS, ORC_F_CDIST,
  ORC_F_CFRAME, ORC_F_J, ORC_F_AREF, ORC_F_EFCD, ORC_F_EFCFORCE, ORC_F_DOF_INVWEIGHT0, ORC_F_BODY_INVWEIGHT0,
  ORC_F_MEANINERTIA, ORC_F_QFRC_ACT, ORC_F_QFRC_PASSIVE, ORC_F_EFCPOS, ORC_F_DBG_IMP, ORC_F_DBG_GN, ORC_F_DBG_ALPHA, ORC_F_DBG_LS
};

#ifdef __cplusplus
extern "C" {
#endif

int orc_sizeof_model(void);
int orc_sizeof_data(void);
int orc_sizeof_real(void);
int orc_compile(const MirSceneSpec* spec, OrcModel* m);
void orc_init_data(const OrcModel* m, OrcData* d);
void orc_fk(const OrcModel* m, OrcData* d);
void orc_forward(const OrcModel* m, OrcData* d);  /* fk .. qacc, no integration */
void orc_step(const OrcModel* m, OrcData* d);     /* forward + integrate + fk */
void orc_reset(const OrcModel* m, OrcData* d, const double* obj_pos, const double* obj_quat, const double* arm_qpos);
void orc_set_targets(const OrcModel* m, OrcData* d, const double* tgt);
void orc_get_obs(const OrcModel* m, const OrcData* d, double* agent_pos, double* env_state, double* reward, unsigned char* terminated);
int orc_read(const OrcModel* m, const OrcData* d, int field, double* out);
int orc_write(const OrcModel* m, OrcData* d, int field, const double* in);
int orc_counts(const OrcData* d, int* ncon, int* nefc, int* niter);
int orc_read_batch(const OrcModel* m, const OrcData* d, int B, int field, double* out, int stride);
int orc_write_batch(const OrcModel* m, OrcData* d, int B, int field, const double* in, int stride);
void orc_counts_batch(const OrcData* d, int B, int* ncon, int* nefc, int* niter);
void orc_ncand_batch(const OrcData* d, int B, int* ncand);
void orc_get_obs_batch(const OrcModel* m, const OrcData* d, int B, double* agent_pos, int agent_dim, double* env_state, int env_dim, double* reward,
                       unsigned char* terminated);
/* independent articulated-body (Featherstone ABA) unconstrained forward dynamics, for cross-checks */
void orc_aba(const OrcModel* m, OrcData* d, double* qacc_out);
/* batch driver (OpenMP over envs) for the timed CPU baseline: action (T?) — random targets supplied by caller */
void orc_step_batch(const OrcModel* m, OrcData* d, int B, const float* action /* (B,nu) or NULL */, int nthreads);
/* -DORC_COUNT_FLOPS builds only (0 / no-op otherwise): floating-point operations executed by THIS thread since the last reset,
 * out[6] = add+sub, mul, div, sqrt, sin/cos/atan2/pow, comparisons */
int orc_flops_read(unsigned long long* out);
void orc_flops_reset(void);

#ifdef __cplusplus
}
#endif
#endif

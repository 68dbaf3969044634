/*
 * mirigid.h — C ABI of libmirigid.so, the MI355X-native batched rigid-body
 * step backend that sits behind gym-genesis's env.step() hot path.
 *
 * The reference (huggingface/gym-genesis) has NO FFI for this path: every
 * FLOP of env.step() runs inside the external `genesis-world` package through
 * its Python object API.  The entry points below are therefore the C-level
 * restatement of exactly those Genesis calls the reference makes on the path
 * (file:line cited per function, paths relative to /root/reference):
 *
 *   gs.Scene(...) + add_entity + scene.build(n_envs=B)
 *        gym_genesis/tasks/franka/cube_pick.py:34-68      -> mir_create
 *   cube.set_pos / cube.set_quat / franka.set_qpos(zero_velocity=True)
 *        gym_genesis/tasks/franka/cube_pick.py:96-102     -> mir_reset
 *   franka.control_dofs_position(target, dofs_idx)
 *        gym_genesis/tasks/franka/cube_pick.py:104-105,123-124 -> mir_set_pd_targets
 *   scene.step()
 *        gym_genesis/tasks/franka/cube_pick.py:107,125 ; gym_genesis/env.py:60 -> mir_step
 *   step() = control + scene.step() + compute_reward() + get_obs() + (reward==1)
 *        gym_genesis/tasks/franka/cube_pick.py:122-181 ; gym_genesis/env.py:61-69 -> mir_step_fused
 *   link.get_pos / link.get_quat / entity.get_dofs_position / get_qpos
 *        gym_genesis/tasks/franka/cube_pick.py:140-146    -> mir_get_links / mir_get_state
 *
 * Conventions
 *   - All device buffers are caller-owned raw device pointers (torch
 *     tensor.data_ptr()), row-major (B, D), float32 unless stated.  Nothing
 *     is retained past the call.  Internal state is env-major
 *     rows (B, D) in HBM (one 64-byte-aligned row group per env).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  No
 *     entry point synchronises the device.
 *   - Quaternions are wxyz (Genesis convention, cube_pick.py:94).
 *   - Return 0 on success, negative MIR_E_* on error; text in mir_last_error().
 *   - The scene description (MirSceneSpec) is plain host data in doubles.
 */
#ifndef MIRIGID_H
#define MIRIGID_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIR_VERSION 3

/* capacity limits of the spec.  Two step kernels sit behind the ABI: scenes with nv <= 15, nbody <= 16,
 * ngeom <= 24, <= 64 candidate pairs and max_contacts <= 16 (the pick tasks) run on the 16-lanes-per-env kernel
 * (4 envs per wave); anything larger (the 5-cube stack tasks) on the wave-per-env kernel, whose limits these are.
 * Every kinematic tree must have <= 16 dofs. */
#define MIR_MAX_BODY 32 /* including world = body 0 */
#define MIR_MAX_DOF 48  /* nv */
#define MIR_MAX_Q 56    /* nq */
#define MIR_MAX_GEOM 40
#define MIR_MAX_PAIR 256   /* candidate geom pairs after static filtering */
#define MIR_MAX_CONTACT 48 /* contacts kept per env per step (plane-box <= 4, box-box <= 8 per pair) */
#define MIR_MAX_GRIP 4
#define MIR_MAX_FREE 8     /* free bodies (cubes) */
#define MIR_MAX_HULL_VERT 32 /* vertices of one MIR_GEOM_HULL geom */
#define MIR_MAX_VERT 96      /* the scene's vertex pool */

/* error codes */
#define MIR_OK 0
#define MIR_E_INVALID (-1) /* bad argument / spec */
#define MIR_E_CAPACITY (-2) /* spec exceeds MIR_MAX_* */
#define MIR_E_HIP (-3)      /* HIP runtime error */
#define MIR_E_NODEVICE (-4) /* no gfx950 device visible */
#define MIR_E_MASK (-5)     /* early terminated bytes (mir_step_begin) turned out wrong; see mir_step_begin */

/* joint types (one joint per body, anchored at the body origin) */
#define MIR_JNT_FIXED 0
#define MIR_JNT_REVOLUTE 1
#define MIR_JNT_PRISMATIC 2
#define MIR_JNT_FREE 3 /* qpos = pos3 + quat4(wxyz); qvel = world lin3 + world ang3 */

/* geom types */
#define MIR_GEOM_PLANE 0 /* z = 0 plane of its body frame, normal +z */
#define MIR_GEOM_BOX 1   /* size = half extents */
#define MIR_GEOM_SPHERE 2  /* size[0] = radius */
#define MIR_GEOM_CAPSULE 3 /* size[0] = radius, size[1] = half length of the axis segment, axis = z of the geom frame */
#define MIR_GEOM_HULL 4    /* convex hull of size[1] vertices of the scene's pool starting at vertex size[0] (MirSceneSpec.vert, geom
                            * frame; the frame's origin must lie inside the hull: MPR starts from it); size[2] is ignored.  The
                            * stand-in for the reference's mesh collision geometry (convex hulls of link meshes).  Support mapping
                            * = the vertex of largest projection, lowest index on ties.  Plane - hull: the penetrating vertices,
                            * reduced to four like plane - box; every other pair through GJK on the cores (the hull is its own core,
                            * radius 0) and MPR.  Both step kernels: up to 40 vertices per scene in the 16-lane kernel, up to
                            * MIR_MAX_VERT in the wave kernel (a scene with more than 40 goes there). */
/* Narrowphase by pair type: plane-box / plane-sphere / plane-capsule in closed form (<= 4 / 1 / 2 points), box-box by
 * separating axes and face clipping (<= 8 points), every other convex pair by Minkowski Portal Refinement on the shapes'
 * support mappings (one point: deepest penetration) -- the default convex-convex path of Genesis.
 * Every geom type is supported by both step kernels (the wave-per-env kernel since round 3 for spheres and capsules, since
 * round 4 for hulls). */

/* dof control modes */
#define MIR_CTRL_NONE 0
#define MIR_CTRL_POSITION 1 /* tau = kp (target - q) - kv qd, clamped (control_dofs_position) */

typedef struct MirBodySpec {
  int32_t parent; /* index < own index; 0 = world */
  int32_t jtype;  /* MIR_JNT_* */
  double pos[3];  /* frame in parent (FREE: initial world pose) */
  double quat[4]; /* wxyz */
  double axis[3]; /* joint axis in body frame (REVOLUTE / PRISMATIC) */
  double mass;
  double ipos[3];    /* COM in body frame */
  double inertia[6]; /* about COM, body axes: xx yy zz xy xz yz */
} MirBodySpec;

typedef struct MirDofSpec {
  int32_t limited;   /* joint range active */
  int32_t ctrl_mode; /* MIR_CTRL_* */
  double range[2];
  double armature;
  double damping;
  double kp, kv;
  double frc_range[2];
  double solref[2]; /* timeconst, dampratio (limit constraint) */
  double solimp[5]; /* dmin dmax width mid power */
} MirDofSpec;

typedef struct MirGeomSpec {
  int32_t body;
  int32_t type; /* MIR_GEOM_* */
  int32_t contype;
  int32_t conaffinity;
  double size[3];
  double pos[3];  /* in body frame */
  double quat[4]; /* wxyz */
  double friction;
  double solref[2];
  double solimp[5];
} MirGeomSpec;

typedef struct MirOptions {
  double dt;         /* cube_pick.py:45 -> 0.01 */
  double gravity[3]; /* (0,0,-9.81) */
  double tolerance;  /* Newton scaled-improvement / gradient tolerance */
  double ls_tolerance;
  int32_t iterations; /* max Newton iterations */
  int32_t ls_iterations;
  int32_t enable_collision;
  int32_t enable_joint_limit;
  int32_t enable_self_collision;     /* geoms of one articulated entity may collide */
  int32_t enable_adjacent_collision; /* parent/child link geoms may collide */
  int32_t max_contacts;              /* <= MIR_MAX_CONTACT */
  int32_t implicit_damping;          /* M += dt (damping + kv) on the diagonal */
} MirOptions;

/* reward rules */
#define MIR_REWARD_LIFT 0  /* obj_z > reward_z                                   (cube_pick.py:130-135) */
#define MIR_REWARD_STACK 1 /* |obj.xy - obj2.xy| < reward_xy && obj.z - obj2.z > reward_dz
                              (cube_stack_batch.py:143-153, cube_stack_kitchen_batch.py:138-146) */
/* agent_pos layouts */
#define MIR_AGENT_EEF 0  /* [eef pos3, eef quat4, grip q...]                    (cube_pick.py:140-151) */
#define MIR_AGENT_QPOS 1 /* qpos of every scalar joint, body order               (cube_stack_batch.py:169) */

/* what the fused step extracts (reference get_obs / compute_reward) */
typedef struct MirTaskSpec {
  int32_t eef_body;               /* franka.get_link("hand"), cube_pick.py:68 */
  int32_t obj_body;               /* cube / cube_1 */
  int32_t n_grip;                 /* gripper dofs appended to agent_pos (MIR_AGENT_EEF) */
  int32_t grip_dof[MIR_MAX_GRIP]; /* dof indices (cube_pick.py:142 -> 7,8) */
  double reward_z;                /* MIR_REWARD_LIFT threshold (cube_pick.py:134 -> 0.1) */
  int32_t obj2_body;              /* cube_2, or -1: env_state gets obj2 pos3 appended (11 -> 14 columns) */
  int32_t reward_mode;            /* MIR_REWARD_* */
  int32_t agent_mode;             /* MIR_AGENT_* */
  int32_t _pad;
  double reward_xy, reward_dz;    /* MIR_REWARD_STACK thresholds (0.05, 0.03) */
} MirTaskSpec;

typedef struct MirSceneSpec {
  int32_t struct_size; /* = sizeof(MirSceneSpec), ABI check */
  int32_t version;     /* = MIR_VERSION */
  int32_t nbody, ndof, ngeom, _pad;
  MirOptions opt;
  MirTaskSpec task;
  MirBodySpec body[MIR_MAX_BODY];
  MirDofSpec dof[MIR_MAX_DOF]; /* in body order: 1 per REVOLUTE/PRISMATIC, 6 per FREE */
  MirGeomSpec geom[MIR_MAX_GEOM];
  int32_t nvert, _pad2;
  double vert[MIR_MAX_VERT][3]; /* vertex pool of the MIR_GEOM_HULL geoms, geom frame */
} MirSceneSpec;

typedef struct MirScene* MirHandle;

/* sizes of the batched state vectors for a created scene */
typedef struct MirDims {
  int32_t num_envs, nbody, nq, nv, ngeom, npair, agent_dim, env_dim;
  int32_t nfree, kernel; /* free bodies; 16 = 16-lanes-per-env kernel, 64 = wave-per-env kernel */
} MirDims;

int mir_version(void);
int mir_spec_sizeof(void);
const char* mir_last_error(void);

/* gs.Scene + add_entity + scene.build(n_envs): compile spec, allocate env-major state
 * for num_envs on device_id, set state to the spec's initial pose. */
int mir_create(const MirSceneSpec* spec, int32_t num_envs, int32_t device_id, MirHandle* out);
int mir_destroy(MirHandle h);
int mir_get_dims(MirHandle h, MirDims* out);

/* derived model constants, for parity tests (host doubles; arrays sized nv / nbody) */
int mir_get_model_consts(MirHandle h, double* dof_invweight0, double* body_invweight0, double* meaninertia);

/* cube.set_pos/set_quat + franka.set_qpos(zero_velocity=True) (+ PD targets = qpos).
 * obj_pos (B,nfree,3) / obj_quat (B,nfree,4): world poses of ALL free bodies in body order (the pick scenes
 * have one, task.obj_body; the stack scenes five: cube_stack_kitchen_batch.py:70-96); arm_qpos (B,n_arm):
 * values for every non-FREE joint in body order.  All velocities and the solver
 * warm start are zeroed.  env_mask (B) u8 nullable: reset only envs with mask!=0. */
int mir_reset(MirHandle h, const float* obj_pos, const float* obj_quat, const float* arm_qpos,
              const uint8_t* env_mask, void* stream);

/* Device-side episode bookkeeping and re-spawn.  The reference can only reset the
 * whole batch from the host after a D->H read of `terminated` (README.md:41-43, env.py:64,
 * cube_pick.py:86-112); this call keeps the loop on the device.  Per env:
 *   episode_len += 1;  truncated = !terminated && max_len > 0 && episode_len >= max_len;
 *   done = terminated || truncated.
 * Done envs are reset exactly as mir_reset does (arm_qpos, zero velocity, PD targets = arm_qpos,
 * free bodies at spawn_pool[cursor % pool_len][env] with obj_quat), episode_len = 0, cursor += 1.
 * spawn_pool (pool_len,B,nfree,3) f32, obj_quat (B,nfree,4): spawn poses drawn ahead by the caller (the task's host
 * RandomState stays the source of randomness).  terminated (B) u8 nullable; episode_len, cursor (B)
 * i32 in/out; truncated_out, done_out (B) u8 nullable.  No physics step is consumed. */
int mir_autoreset(MirHandle h, const uint8_t* terminated, int32_t* episode_len, int32_t max_len,
                  const float* spawn_pool, int32_t pool_len, int32_t* cursor, const float* obj_quat,
                  const float* arm_qpos, uint8_t* truncated_out, uint8_t* done_out, void* stream);

/* control_dofs_position over all position-controlled dofs, in dof order: tgt (B,nu) */
int mir_set_pd_targets(MirHandle h, const float* tgt, void* stream);

/* scene.step() x n_steps */
int mir_step(MirHandle h, int32_t n_steps, void* stream);

/* FrankaCubePickBatch.step + GenesisEnv.step in one launch:
 * targets <- action (B,nu); one physics step; agent_pos (B,agent_dim) = [eef_pos3, eef_quat4, grip_q]
 * (MIR_AGENT_EEF) or the scalar-joint qpos (MIR_AGENT_QPOS); env_state (B,env_dim) = [obj_pos3, obj_quat4,
 * eef-obj 3, |eef-obj| 1] (+ obj2_pos3 when task.obj2_body >= 0); reward (B) f32 by task.reward_mode;
 * terminated (B) u8 = reward == 1.  action may be NULL (keep current targets).  action is read once, by the launch, with its first
 * loads: it may also point into pinned host memory (hipHostMalloc / a pinned torch tensor), which the kernel then reads in place
 * over PCIe -- the buffer must stay untouched until the launch has started (mir_step_end of the same step has returned). */
int mir_step_fused(MirHandle h, const float* action, float* agent_pos, float* env_state, float* reward,
                   uint8_t* terminated, void* stream);

/* GenesisEnv.step's host-visible tail, in two halves (gym_genesis/env.py:61-69):
 *     is_success = (reward == 1); terminated = is_success.detach().cpu().numpy().astype(bool)        (env.py:63-64)
 * mir_step_begin = mir_step_fused (same arguments, `terminated` = the device copy, nullable) that also stores the
 * terminated bytes into a pinned host buffer owned by the handle, straight from the kernel (no copy command);
 * mir_step_end blocks until THAT launch has delivered its terminated bytes and copies the B bytes into terminated_host (plain
 * host memory, nullable).  In mode 3 the kernel stores those bytes before its device-side outputs, so the other outputs of the
 * launch are ordered by the stream as after any launch (a later launch, copy or synchronize on it sees them), not by this call.  Between the two calls the host is free (the Python side prepares its return values there).
 * One mir_step_end per mir_step_begin; a step left open (the caller raised between the two calls) is closed by the next
 * mir_step_begin / mir_step_go / mir_reset, which wait for its bytes and drop them.  mir_step_end is otherwise the only entry point
 * of the library that waits for the device.
 * How the wait is done (mir_get_sync_mode; environment variable MIR_SYNC_MODE overrides at mir_create):
 *   3  (default) every terminated byte carries a 5-bit tag (1..31 in bits 1 .. 5, a counter only mir_step_begin advances; bits 6 and 7: exact contacts) that changes from
 *      launch to launch; the host spins until all B bytes show the tag of this launch -- no fence, no flag, nothing in the kernel
 *      waits for the PCIe acknowledgement
 *   2  every wave waits for its host store, the kernel's last workgroup then writes a sequence word into pinned host memory
 *      and the host spins on it (16-lane kernel only)
 *   1  hipStreamWriteValue32 behind the launch writes that word, the host spins on it
 *   0  hipStreamSynchronize
 * EARLY BYTES (mode 3, 16-lane kernel, scenes whose task object is a free body under the world: DevModel::term_bound_ok).  The
 * bytes mir_step_end returns may PRECEDE the end of the solve they belong to: a workgroup stores them from inside its Newton loop
 * once  |z_pred - reward_z| > 2 dt^2 ((1 + sqrt 2) |g|_w + 1 m/s^2) + 1e-5 m  holds for its four envs, where z_pred is the object's
 * height integrated from the current iterate and |g|_w^2 = sum_i w_i g_i^2 >= |g|^2_{Mt^-1} / mass over the dofs of the object's
 * block of the problem (all dofs while a contact joins it to the arm).  The solver's cost is 1-strongly convex in the Mt norm and
 * no accepted step raises it -- per tree, steps being accepted tree by tree, when the problem separates -- so the final iterate is within
 * 2 |g|_{Mt^-1} of the current one whatever makes the solver stop; the bound is that with a 2.4 x margin for float32 evaluation.
 * The kernel still compares the bytes it sent with the integrated state.  A difference (none in any test or soak run) is counted
 * (mir_debug_early_mask_stats), raises a sticky word in pinned memory, and the NEXT mir_step_begin / mir_step_go / mir_step_end /
 * mir_reset on the handle fails ONCE with MIR_E_MASK after switching the handle to late bytes for good (the step that failed the
 * check has already returned its mask: the caller learns that one of the masks since the last successful call was wrong).
 * MIR_NO_EARLY_MASK=1 at mir_create: the bytes always wait for the integrator. */
int mir_step_begin(MirHandle h, const float* action, float* agent_pos, float* env_state, float* reward,
                   uint8_t* terminated, void* stream);
int mir_step_end(MirHandle h, uint8_t* terminated_host);
int mir_get_sync_mode(MirHandle h);
/* How mir_step_begin launches (16-lane kernel; environment variable MIR_SPLIT_STEP overrides at mir_create):
 *   1  (default) ROTATED: a step's launch runs the action-dependent half of THIS step (smooth force with the new targets, solves,
 *      integration, terminated bytes, observations) and then the action-independent half of the NEXT step (poses, dynamics, collision
 *      detection, contact Jacobians) into a scratch buffer of the handle -- work that so runs while the host is between two calls.
 *      Bit-identical to the fused launch.  Whatever touches the state in between (reset, state write, any other step entry point)
 *      makes the next mir_step_begin start over with a fused launch.
 *   2  the same split as two launches (also the fallback where a scene's closing forward kinematics cannot be shared by the waves)
 *   0  one fused launch per step (the wave kernel always) */
int mir_get_split_step(MirHandle h);
/* mir_step_begin with the four output pointers registered ahead of time (mir_step_prepare touches no device state and is meant to
 * be called while the previous step's kernel is still running): the GPU idles in front of mir_step_go, which then takes three
 * arguments instead of seven.  One mir_step_prepare per mir_step_go (the registration is consumed when a launch is queued; the
 * latest registration wins). */
int mir_step_prepare(MirHandle h, float* agent_pos, float* env_state, float* reward, uint8_t* terminated);
int mir_step_go(MirHandle h, const float* action, void* stream);

/* EXACT CONTACTS for scenes on the 16-lane kernel (no reference counterpart: Genesis keeps every contact point of its 100+ candidate
 * pairs -- RigidOptions at gym_genesis/tasks/franka/cube_pick.py:46 -- while the 16-lane kernel keeps 16 per env and thins the
 * manifolds beyond that; the reference's own expert, examples/franka/pick_cube_state.py:33-41,86-88, drives 29 % of its env-steps
 * there).  With the switch on, a step launched by mir_step_begin / mir_step_go DEFERS every env whose narrowphase found more than 16
 * candidate points: the launch stores nothing for it and says so in bit 7 of its terminated byte; mir_step_end then steps exactly those
 * envs with MIR_MAX_CONTACT points, never thinned below that, from their untouched state rows, with the step's action and into the
 * step's output pointers, recomputes their part of the split step's hand-over, and returns when their terminated bytes have arrived
 * too.  Since round 6 they take ONE launch of the 16-lane kernel's list instantiation (three contacts per lane, four envs per
 * workgroup, the next step's hand-over included); an env beyond THAT capacity too -- more than 48 points (thinned to 48 as before) or
 * more than 16 candidate geom pairs -- goes to the wave-per-env kernel (the same scene compiled for it; 64 candidate pairs) as every
 * deferred env did in round 5 (MIR_EXACT_WAVE=1 in the environment at the time of the call: all of them still do).
 * `action` of the pending mir_step_begin / mir_step_go is read AGAIN by those launches: with the switch on it must stay valid, and
 * unchanged, until mir_step_end has returned; and the next mir_step_begin may name another stream (it is made to wait for them).  An env that is not deferred is computed exactly as without the switch; a
 * step without deferred envs launches nothing extra.  The state rows, outputs and link poses of a deferred env are those of the step
 * only once mir_step_end has returned: anything queued between mir_step_begin and mir_step_end -- a mir_render of the observation's
 * images, say -- sees its OLD rows (the task classes close the step before they draw).  `spec`: the spec the scene was created from (compiled once more, for the wave
 * kernel; may be NULL when switching back on).  While the switch is on, mir_step and mir_step_fused run as begin + end (they wait for the
 * step), and mir_step_packed / mir_rollout / mir_rollout_autoreset return MIR_E_INVALID (their steps are never closed on the host).
 * MIR_E_INVALID for scenes of the wave kernel (nothing to do) and for sync modes other than 3.
 * mir_get_exact_stats: out4 = {steps closed by mir_step_end, steps that had deferred envs, deferred env-steps, most deferred envs in one
 * step} since the last reset of the counters (reset != 0 clears them).  mir_get_exact_route: out4 = {deferred env-steps handed to the
 * list instantiation, env-steps stepped by the wave-per-env kernel, HEAVY steps, steps of overflow runs as two launches} over the same period.  A heavy step: while at least
 * 1 / 16 of the envs are above 16 points (and until fewer than 1 / 32 are; MIR_EXACT_HEAVY="enter,leave" in envs, enter <= 0: never)
 * the whole batch is stepped by ONE launch of the three-contacts-per-lane instantiation instead of a launch that defers most envs and
 * a list launch behind it (an env with at most 16 points is computed there bit for bit as by the one-contact-per-lane kernel); the
 * statistics count its envs above 16 points as deferred.
 * OVERFLOW RUNS in a loop that leaves room between two steps (second session of round 6): when the caller spent at least 40 us
 * (MIR_EXACT_BIG_GAP) between mir_step_end's return and this mir_step_begin -- a policy, its IK -- a step of a run (from the step after
 * one that deferred envs until a step in which no env is above 16 points) is TWO launches of the three-contacts-per-lane instantiation for
 * the whole batch: the second half of the step from the scratch rows up to the outputs and the terminated bytes, on `stream`; and the first
 * half of the next step on the library's side stream, beside whatever the caller queues on `stream` next (the next mir_step_begin is made
 * to wait for it) -- the second half as two list launches once the first half has said which envs are above 16 points: those on the
 * three-contacts-per-lane instantiation, the others in one round of the one-contact-per-lane kernel's workgroups on the side stream.
 * Same results bit for bit; less time inside the two calls, a third more GPU time per step.  MIR_EXACT_BIG=0: never
 * (heavy phase / list launches), 2: whenever the rows are there.  out4[3] of mir_get_exact_route counts such steps.
 * on = 2 (tests): every env of every step is deferred, i.e. the whole batch is stepped by the launches that otherwise serve the
 * deferred envs only -- the twin the parity tests compare a deferred env with, bit for bit. */
int mir_set_exact_contacts(MirHandle h, const MirSceneSpec* spec, int32_t on);
int mir_get_exact_contacts(MirHandle h);
int mir_get_exact_stats(MirHandle h, uint64_t* out4, int32_t reset);
int mir_get_exact_route(MirHandle h, uint64_t* out4);

/* Same step, but every output of an env lands in ONE packed float32 row
 * rows[e*row_stride + ...] = [agent_pos (agent_dim) | env_state (env_dim) | reward | terminated(0/1)]
 * so a sharded run gathers all per-step outputs with a single collective
 * (row_stride >= agent_dim + env_dim + 2, in floats). */
int mir_step_packed(MirHandle h, const float* action, float* rows, int32_t row_stride, void* stream);

/* K consecutive env.step() calls in ONE launch (the "multi-step variant for rollouts" of SURVEY.md 7-7; the loop shape of
 * README.md:30-43 with the actions already on the device): for k < n_steps: targets <- actions[k] (B,nu); one physics step;
 * rows[k] (B,row_stride) <- packed outputs as in mir_step_packed.  actions (n_steps,B,nu), rows (n_steps,B,row_stride).
 * Bit-identical to n_steps calls of mir_step_packed; the state never leaves the chip between the steps. */
int mir_rollout(MirHandle h, const float* actions, int32_t n_steps, float* rows, int32_t row_stride, void* stream);

/* mir_rollout with the episode loop of mir_autoreset inside the launch: after every step the packed row (the terminal
 * observation of envs that just ended) is written, then episode_len / truncation / re-spawn are applied exactly as
 * mir_autoreset does, on chip.  rows[k][e][agent_dim + env_dim + 2] = truncated (row_stride >= agent_dim + env_dim + 3).
 * Bit-identical to n_steps x (mir_step_packed; mir_autoreset). */
int mir_rollout_autoreset(MirHandle h, const float* actions, int32_t n_steps, float* rows, int32_t row_stride, int32_t* episode_len,
                          int32_t max_len, const float* spawn_pool, int32_t pool_len, int32_t* cursor, const float* obj_quat,
                          const float* arm_qpos, void* stream);

/* get_obs() without stepping */
int mir_get_obs(MirHandle h, float* agent_pos, float* env_state, float* reward, uint8_t* terminated,
                void* stream);

/* full simulation state; any pointer may be NULL.  qpos (B,nq) qvel (B,nv)
 * target (B,nu) warmstart (B,nv) */
int mir_get_state(MirHandle h, float* qpos, float* qvel, float* target, float* warmstart, void* stream);
int mir_set_state(MirHandle h, const float* qpos, const float* qvel, const float* target,
                  const float* warmstart, void* stream);

/* link.get_pos / get_quat for all bodies: pos (B,nbody,3) quat (B,nbody,4) */
int mir_get_links(MirHandle h, float* pos, float* quat, void* stream);

/* per-env diagnostics of the last step: ncon (B) i32, nefc (B) i32, niter (B) i32; nullable.  The step kernels write them
 * (16 B per env-step) while they are switched on: mir_set_diag(h, 0) drops that traffic (the task classes do), after which
 * mir_get_diag is an error until they are switched on again.  On by default. */
int mir_set_diag(MirHandle h, int32_t on);
int mir_get_diag(MirHandle h, int32_t* ncon, int32_t* nefc, int32_t* niter, void* stream);
/* mir_get_diag + ncand_points (B) i32, nullable: the contact points the narrowphase found BEFORE the scene's contact capacity was
 * applied (MirOptions.max_contacts; 16 at most in the 16-lane kernel, 48 in the wave kernel).  ncand_points > capacity = the
 * manifolds of that env-step were thinned (oracle/orc_rigid.c: thin_manifolds); the fraction of such env-steps is what bench.py and
 * the reference-expert tests report as cap_hit_frac.  Saturates at 255. */
int mir_get_diag4(MirHandle h, int32_t* ncon, int32_t* nefc, int32_t* niter, int32_t* ncand_points, void* stream);

/* Divergence guard (the reference's users get Genesis's own error on a NaN state, scene.step() at
 * gym_genesis/tasks/franka/cube_pick.py:107,125).  While diagnostics are on, a step kernel flags every env whose integrated qpos / qvel
 * holds a NaN or an Inf: bad (B) u8 device, nullable <- 1 for the envs flagged by the LAST step launch (sticky over the steps of a
 * rollout launch); *env_steps (host, nullable) <- env-steps flagged since the counter was last reset (synchronises the stream);
 * reset != 0 clears the counter.  Off the hot path: nothing is computed or stored while diagnostics are off (mir_set_diag(h, 0)).
 * A flagged env keeps stepping (its state stays non-finite until it is reset); its reward is 0 and its `terminated` False -- the
 * threshold tests hold for finite heights only -- and no other env is affected (bit-identical to a run without it). */
int mir_get_bad(MirHandle h, uint8_t* bad, uint32_t* env_steps, int32_t reset, void* stream);

/* stage outputs of one forward-dynamics evaluation at the current state (no
 * integration), for per-stage parity tests: M (B,nv,nv), qfrc_bias (B,nv),
 * qacc_smooth (B,nv), qacc (B,nv); nullable */
int mir_forward(MirHandle h, float* M, float* qfrc_bias, float* qacc_smooth, float* qacc, void* stream);

/* ---- cameras / pixels (BASELINE.json configs[4]) ---------------------------------
 * scene.add_camera(res=(W,H), pos, lookat, fov)   gym_genesis/tasks/franka/cube_pick.py:56-63
 * cam.set_pose(pos, lookat) + cam.render()[0]      gym_genesis/tasks/franka/cube_pick.py:166-176,
 *                                                  gym_genesis/env.py:97-98
 * The reference renders through Genesis's OpenGL rasteriser over mesh assets that are not in the
 * reference tree; here the image is formed from the scene's own collision primitives (boxes, planes)
 * by a tiled HIP kernel: pinhole camera (vertical fov, +z up, pixel-centre sampling, row 0 = top),
 * nearest primitive per pixel, Lambert shading from one directional light, a checker on planes. */
typedef struct MirCameraSpec {
  int32_t width, height; /* res=(W,H) */
  double pos[3], lookat[3], up[3];
  double fov_deg; /* vertical field of view, degrees */
} MirCameraSpec;

typedef struct MirVisualSpec {
  int32_t struct_size; /* = sizeof(MirVisualSpec) */
  int32_t _pad;
  double geom_rgb[MIR_MAX_GEOM][3]; /* albedo per geom, 0..1 */
  double light_dir[3];              /* direction TOWARDS the light (normalised by the library) */
  double ambient, diffuse;          /* colour = albedo * (ambient + diffuse * max(0, n.l)) */
  double sky_rgb[3];                /* rays that hit nothing */
  double checker_rgb[2][3];         /* plane geoms: albedo of the two checker colours */
  double checker_size;              /* checker cell edge, metres */
} MirVisualSpec;

#define MIR_RENDER_PER_ENV 0 /* one image per env, camera pose relative to the env (cube_pick.py:166-171) */
#define MIR_RENDER_GLOBAL 1  /* one image of all envs placed at env_offset (cube_pick.py:174-176) */

int mir_visual_sizeof(void);

/* Render RGB8 images of the current state.  mode PER_ENV: pixels (B,H,W,3) u8, env e drawn alone
 * as seen from `cam` in its own frame (env_offset ignored).  mode GLOBAL: pixels (H,W,3) u8, every env
 * drawn displaced by env_offset[e] (B,3 f32 device, NULL = all zero); plane geoms are drawn once.
 * cam / vis are host structs (copied into the launch).  pixels is a device pointer.  B x H x W x 3 may exceed 2^32 bytes (64-bit
 * image bases); MIR_E_CAPACITY when a single image reaches 2^32 bytes or a per-env call has more than 65535 images.
 * Link poses: from the first render on, every call that advances the state also leaves the link poses of its final state for the
 * rasteriser, and a render queued behind it (same stream order as the calls that touched the state, as for every call on a handle)
 * launches no forward kinematics of its own; behind mir_reset / mir_autoreset / mir_set_state it does. */
/* A counter that every call which changes the positions of the bodies advances (steps, resets, state writes): two reads that return
 * the same value bracket calls that left the scene as it was -- an image rendered in between is still the image of the scene.
 * (>= 0: the counter modulo 2^31.) */
int mir_get_state_version(MirHandle h);

int mir_render(MirHandle h, const MirCameraSpec* cam, const MirVisualSpec* vis, int32_t mode, const float* env_offset,
               uint8_t* pixels, void* stream);

/* Per-env images from PER-ENV cameras (the wrist cameras that follow a link:
 * gym_genesis/tasks/franka/cube_stack_kitchen_batch.py:185-192, gym_genesis/tasks/so101/cube_stack_batch.py:199-213).
 * cam gives res / fov / default up; cam_pos, cam_lookat (B,3) f32 device, in the env's own frame; cam_up (B,3) nullable.
 * A view direction parallel to the up hint falls back to up = +y, then +x (also in mir_render). */
int mir_render_cams(MirHandle h, const MirCameraSpec* cam, const MirVisualSpec* vis, const float* cam_pos, const float* cam_lookat,
                    const float* cam_up, uint8_t* pixels, void* stream);

/* ---- batched inverse kinematics ---------------------------------------------------
 * robot.inverse_kinematics(link=eef, pos=(B,3), quat=(B,4), init_qpos=..., envs_idx=...) -> (B, n_dofs)
 *      examples/franka/pick_cube_state.py:46-51, examples/franka/stack_cube_state.py:78-83
 * Genesis's IK lives in the external package; this is a damped-least-squares solver on the scene's own kinematics,
 * defined here and restated independently by the oracle (oracle/orc_rigid.c: orc_ik):
 *   q_acc = the seed, lambda^2 = damping^2;  repeat up to max_iters, with q the candidate (the seed at first):
 *     e = [p* - p(q) ; rotvec(q* q(q)^-1)]   (rotation part dropped when target_quat == NULL), m = |e_pos|/pos_tol + |e_rot|/rot_tol;
 *     the candidate is ACCEPTED when m fell below the accepted iterate's (always at first): q_acc = q, its e and its 6 x n geometric
 *     Jacobian J of the chain are kept, lambda^2 <- max(lambda^2 / 4, damping^2 / 256), and the step is STALLED when m fell by less than
 *     1 %; otherwise it is REJECTED: lambda^2 <- min(8 lambda^2, 64 damping^2), also stalled (Levenberg - Marquardt, round 6: fixed damping
 *     crept for 32 iterations towards targets whose Jacobian has a small singular value, and a handful of such envs set the time of
 *     every launch).  Stop the env when |e_pos| < pos_tol and |e_rot| < rot_tol at the accepted iterate, or after three stalled
 *     iterations in a row (a target beyond the joint limits or the reach);
 *     dq = J^T (J J^T + lambda^2 I)^-1 e of the accepted iterate, scaled down so that max |dq_i| <= max_step;  q = q_acc + dq,
 *     clamped to the joint ranges (respect_joint_limit).  The result is q_acc.
 * Only the scalar joints on the chain world -> link move; every other entry of the result is the seed. */
typedef struct MirIkOptions {
  int32_t max_iters;           /* default 20 (Genesis's max_solver_iters, recalled: the package is not in /root/reference; 32 until round 6) */
  int32_t respect_joint_limit; /* default 1 */
  double damping;              /* default 0.05 */
  double pos_tol, rot_tol;     /* defaults 5e-4 m, 5e-3 rad */
  double max_step;             /* default 0.5 rad (or m) per iteration */
} MirIkOptions;

/* target_pos (B,3), target_quat (B,4 wxyz) nullable, init_qpos (B,n_arm) nullable (NULL = the scene's current joint
 * positions), opt nullable (defaults); qpos_out (B,n_arm) = every scalar joint in body order; err_out (B,2) nullable =
 * final |e_pos|, |e_rot|.  The scene state is not modified. */
int mir_inverse_kinematics(MirHandle h, int32_t link_body, const float* target_pos, const float* target_quat, const float* init_qpos,
                           const MirIkOptions* opt, float* qpos_out, float* err_out, void* stream);

/* The same solver for a LIST OF ROWS: robot.inverse_kinematics(link, pos, quat, init_qpos=..., envs_idx=idx) as the reference's experts
 * call it on every step of their loops (examples/franka/pick_cube_state.py:46-51 with envs_idx = arange(B) and ONE quaternion expanded
 * to the batch; examples/so_101/collect_task_stack_cube_batch.py:90-95 with init_qpos of the arm's columns) in ONE launch -- no
 * scatter of the targets to batch rows in front of it, no gather of the result rows behind it.  Row k of qpos_out (n_rows, n_arm) /
 * err_out (n_rows, 2) belongs to env env_idx[k] (int64, device; NULL: n_rows = num_envs, row k = env k; an index outside the batch is
 * clamped).  flags say how the inputs are addressed: by row k -- (n_rows, .) arrays -- unless MIR_IK_POS_BY_ENV / MIR_IK_QUAT_BY_ENV /
 * MIR_IK_INIT_BY_ENV ((num_envs, .) arrays, row of the env); MIR_IK_QUAT_ONE: target_quat is one quaternion for every row.
 * init_qpos holds init_ncols columns of the joint row starting at init_col0 (0, 0: all n_arm columns); the other joints are seeded from
 * the scene state.  rows == NULL: mir_inverse_kinematics. */
#define MIR_IK_POS_BY_ENV 1u
#define MIR_IK_QUAT_BY_ENV 2u
#define MIR_IK_QUAT_ONE 4u
#define MIR_IK_INIT_BY_ENV 8u
typedef struct MirIkRows {
  const int64_t* env_idx;  /* device, n_rows entries, or NULL */
  int32_t n_rows;
  uint32_t flags;
  int32_t init_col0, init_ncols;
} MirIkRows;
int mir_inverse_kinematics_rows(MirHandle h, int32_t link_body, const MirIkRows* rows, const float* target_pos, const float* target_quat,
                                const float* init_qpos, const MirIkOptions* opt, float* qpos_out, float* err_out, void* stream);

/* ---- multi-GPU observation gather on the copy path (no reference counterpart: README.md:41-48 is single-device) ----
 * One process per GPU; every rank owns a receive buffer that the other ranks have mapped through HIP IPC (the host side does the
 * handle exchange: gym_genesis/sharding.py: CopyPathGather).  mir_p2p_push enqueues, on `stream` of the CURRENT device, one
 * device-to-device copy of `nbytes` from src to each of the n destinations (addresses inside the peers' buffers; a local address
 * is a plain device copy) and THEN one 4-byte copy of *flag_src to each flag_dst[i]: copies of one stream complete in order, so a
 * receiver that sees the sequence word has the whole block.  No kernel is launched (the copies run on the SDMA engines), which
 * is the point: a collective kernel cannot share a CU with the step kernel's four resident workgroups.
 * mir_p2p_enable makes `peer`'s memory addressable from `device` (idempotent). */
int mir_p2p_enable(int32_t device, int32_t peer);
int mir_p2p_push(void* const* dst, int32_t n, const void* src, uint64_t nbytes, void* const* flag_dst, const void* flag_src, void* stream);
/* mir_p2p_push with one stream per destination (streams[i] carries destination i's block and then its word): the copies to different
 * peers overlap.  nbytes = 0 sends the words only (the acknowledgements of the gather's flow control). */
int mir_p2p_push_streams(void* const* dst, int32_t n, const void* src, uint64_t nbytes, void* const* flag_dst, const void* flag_src,
                         void* const* streams);

/* ---- debug aids (exported for the tests and tools/; not part of the drop-in surface) ---------------------------
 * mir_debug_profile_step: one step with phase timestamps (shader clock) of workgroup 0 into prof (32 x u64, device).
 * mir_debug_poison_lds: overwrite the LDS of every CU with signalling-NaN patterns, so that a kernel reading an LDS slot
 * before writing it yields NaNs instead of plausible stale values (the GPU tests call it before every scene).
 * mir_debug_null_roundtrip: microseconds per (launch of an empty kernel + host-visible completion word), averaged over iters:
 * the floor under one synchronous env.step() on this machine, with none of the physics in it.
 * mir_debug_copy_rows: dst[i] = src[i] for n_floats floats with the step kernel's access shape (64-thread workgroups, 4 B per
 * lane): a known byte count against which rocprofv3's FETCH_SIZE / WRITE_SIZE are calibrated for this access width. */
int mir_debug_profile_step(MirHandle h, unsigned long long* prof, void* stream);
/* profiling builds (-DMIR_PROFILE_SINGLE): the next mir_step_begin / mir_step_go launch -- whichever kernel the split-step protocol
 * picks -- leaves its stamps in prof (160 x u64, device; slot 63 = workgroup to watch) */
int mir_debug_profile_next_step(MirHandle h, unsigned long long* prof);
/* ... and the next launch of the LIST INSTANTIATION of exact contacts (mir_step_end of a step with deferred envs): prof 160 x u64; the
 * first pass's stamps as above, slots 140 / 141 / 142 = end of the second pass on the main / the collision wave, its start */
int mir_debug_profile_next_list_step(MirHandle h, unsigned long long* prof);
/* every following mir_inverse_kinematics call also writes the iterations each env took into iters (B x i32, device; NULL: off) */
int mir_debug_ik_iters(MirHandle h, int32_t* iters);
int mir_debug_null_roundtrip(MirHandle h, int32_t iters, void* stream, double* out_us);
/* n back-to-back launches of the rotated step kernel (split mode 1), cycling through n_actions (B, nu) action blocks, without
 * observation outputs -- or, with outputs = {agent_pos, env_state, reward, terminated} (device pointers, shapes as in
 * mir_step_fused), with the four device outputs of a mir_step_go launch (not its host-visible bytes): lets two events time that kernel the way the fused one is timed (bench.py's roofline).  Advances the state
 * by n steps. */
int mir_debug_rotated_launches(MirHandle h, const float* actions, int32_t n_actions, int32_t n, void* const* outputs, void* stream);
/* n steps in ONE launch, every workgroup through its steps at its own pace, actions (n, B, nu) resident, no outputs: the floor under
 * a launch that would stay resident over several env.step() calls (tools/probes/resident_steps.py).  Advances the state by n steps. */
int mir_debug_resident_steps(MirHandle h, const float* actions, int32_t n, void* stream);
/* the 16-lane kernel's row all-reduce on n_rows rows of 16 floats (device arrays): out[r][l] = lane l's result for row r.  Every lane
 * of a row must hold the same bits (the solver's per-lane decisions rest on it: tests/test_gpu_api.py). */
int mir_debug_row_sum(const float* in, float* out, int32_t n_rows, int device_id, void* stream);
int mir_debug_poison_lds(int device_id, void* stream);
int mir_debug_copy_rows(const float* src, float* dst, int64_t n_floats, int device_id, void* stream);
/* the kernels' convex narrowphase on n pairs given directly (device arrays): in (n,22) = type1, size1[3], pos1[3], quat1[4] wxyz,
 * type2, size2[3], pos2[3], quat2[4]; out (n,8) = hit, pos[3], dist, normal[3] (from geom 1 to geom 2) */
int mir_debug_convex_pairs(const float* in, float* out, int32_t n, int device_id, void* stream);
/* which pixel kernel the following mir_render / mir_render_cams calls of this handle use: generic != 0 forces the kernel that
 * culls per workgroup (the one long primitive lists and odd widths always take), 0 restores the default (per-strip lists built
 * once per render whenever an image has fewer than 64 primitives); strip_rows > 0 (a multiple of 32) overrides the height of a
 * workgroup's strip, 0 restores the default.  The two kernels must agree bit for bit (tests/test_gpu_render.py). */
int mir_debug_render_path(MirHandle h, int32_t generic, int32_t strip_rows);
/* host only (no GPU): the sizes and options of the compiled scene as the text of a C++ struct of literals, `struct <name>` with
 * a `matches(const DevModel&)` member -- the constants of a scene-specialised instantiation of the 16-lane step kernel
 * (csrc/mir_spec_pick.h is this text for the CubePick scene, written by tools/gen_scene_spec.py).  Returns the text length, or
 * MIR_E_* (MIR_E_CAPACITY: the scene belongs to the wave kernel, or cap is too small).  A build-time tool: not thread-safe. */
int mir_debug_emit_spec(const MirSceneSpec* spec, const char* name, char* out, int32_t cap);
/* 1 if this handle's launches use the scene-specialised instantiation (its compiled model matched SpecPick and the environment
 * variable MIR_NO_SPEC was unset at mir_create), else 0 */
int mir_debug_spec_active(MirHandle h);
/* counters of the early terminated bytes (see mir_step_begin): out2[0] = workgroup-launches that sent their bytes from inside the
 * solver loop, out2[1] = workgroup-launches whose early bytes differed from the integrated state (anything but 0 also makes the next
 * API call fail with MIR_E_MASK).  Both always counted (a private counter per workgroup, summed here).  Synchronises the stream;
 * reset != 0 clears the counters. */
int mir_debug_early_mask_stats(MirHandle h, uint32_t* out2, int32_t reset, void* stream);
/* test aid: raises the sticky word by hand, as a kernel that found its early bytes wrong would (the next call fails with MIR_E_MASK) */
int mir_debug_raise_mask_flag(MirHandle h);
/* 1 while mir_step_begin launches may send early bytes on this handle, 0 after MIR_NO_EARLY_MASK=1 or a MIR_E_MASK */
int mir_get_early_mask(MirHandle h);

#ifdef __cplusplus
}
#endif
#endif /* MIRIGID_H */

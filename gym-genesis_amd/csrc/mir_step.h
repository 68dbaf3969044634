// mir_step.h — launch arguments of the fused step kernel (shared by mir_step.hip and mir_api.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mir_model.h"

struct StepArgs {
  const DevModel* model;
  float* qpos;    // (B, qstride)
  float* qvel;    // (B, 16)
  float* target;  // (B, 16) indexed by dof
  float* qacc_ws; // (B, 16)
  float* poses;   // (B, 2, 16, 4) or null: link positions then quaternions written for the rasteriser (mode 2 only)
  const float* action;  // (B, nu) or null
  float* agent_pos;     // (B, 7+n_grip) or null
  float* env_state;     // (B, 11) or null
  float* reward;        // (B) or null
  uint8_t* terminated;  // (B) or null
  // host-visible tail of GenesisEnv.step (mir_step_begin / mir_step_end): the same bytes as `terminated`, stored straight into
  // pinned host memory (system-scope stores), and the launch's completion published by its LAST workgroup
  uint8_t* term_host;     // (B padded to 4) device address of pinned host memory, or null; byte = terminated | term_tag << 1
  uint32_t term_tag;      // 0..127: stamped into every byte so that the host can tell this launch's bytes from older ones
  uint32_t* done_ticket;  // device counter, 0 between launches; or null (completion is then signalled by the stream)
  uint32_t* done_flag;    // device address of a pinned host word <- done_seq once every workgroup has stored its bytes
  uint32_t done_seq;
  int32_t* diag;        // (B, 4): ncon, nefc, niter, flags; or null
  // debug / per-stage parity outputs (mir_forward), all nullable
  float* out_M;         // (B, nv, nv)
  float* out_bias;      // (B, nv)
  float* out_qas;       // (B, nv)
  float* out_qacc;      // (B, nv)
  float* out_xpos;      // (B, nbody, 3)
  float* out_xquat;     // (B, nbody, 4)
  unsigned long long* prof; // debug: 16 phase timestamps (shader clock) from block 0, or null
  float* rows;          // (B, row_stride) packed [agent | env_state | reward | terminated] or null
  int row_stride;
  long act_step;   // floats between the action blocks of consecutive steps (rollout mode), 0 = one action for all steps
  long rows_step;  // floats between the row blocks of consecutive steps (rollout mode), 0 = only the final row
  AutoResetArgs ar;  // per-step episode bookkeeping + re-spawn inside the launch (rollout mode only)
  int B;
  int qst, nu;  // qpos row stride and action width (copies of the model's, so the state loads do not wait for the model)
  int mode;     // 0: full steps; 1: forward dynamics only (mir_forward); 2: kinematics + outputs only
  int n_steps;  // mode 0 only
  int features; // bit 0: sphere / capsule geoms (GJK / MPR narrowphase); bit 1: sweep-and-prune broadphase
  // split step (GenesisEnv.step path): `phase` 0 = whole step; 1 = the ACTION-INDEPENDENT half of the coming step only (poses,
  // dynamics, collision, contact arrays, Jacobians) written to `pre`; 2 = the rest of the step, read from
  // `pre`.  3 = 2 followed by 1 (of the next step) in one launch.  `pre`: K16_PRE_STRIDE floats per env.
  // 4 = the LIST instantiation of exact contacts (three contacts per lane, capacity 48): the whole step for the envs of `env_list`,
  // then 1 for them; 5 = its first pass alone for the whole batch (mir_step.hip: VARIANT 6 / 7); 6 / 7 = 2 / 1 with three contacts per lane
  // (VARIANT 9 / 10: the two launches of a step of an overflow run -- rows of up to 48 contacts through `pre_big`); 8 = 3's first pass
  // alone for a list of envs (VARIANT 11: the envs of such a step that are at most at 16 points).
  int phase;
  float* pre;
  // [0] env-steps that ended with a non-finite state (divergence guard, counted while diag is set; mir_get_bad);
  // early terminated bytes (mir_step.hip, mir_model.h: term_bound_ok): [1] workgroups whose early bytes differed from the integrated
  // state (must be 0: a non-zero count also raises the sticky word `term_bad`), [2 + w] launches in which workgroup w sent its bytes
  // from inside the solver loop (a contention-free counter per workgroup, always counted; summed by mir_debug_early_mask_stats)
  uint32_t* early_stats;
  uint32_t* term_bad;  // device address of a pinned host word <- 1 when early bytes turned out wrong (checked by the next API call: MIR_E_MASK)
  int no_early_mask;  // MIR_NO_EARLY_MASK=1: the bytes always wait for the integrator
  int term_wstride;   // 32-bit words between the term_host words of consecutive workgroups (mir_scene.h)
  // EXACT CONTACTS (mir_set_exact_contacts; single-step launches of the mir_step_begin path only).  An env whose narrowphase found
  // more candidate points than this kernel's contact capacity -- the count travels in bits 20 .. 27 of the `coupled` word, through the
  // scratch row of a split step too -- is DEFERRED: the launch stores nothing for it (state, targets, observations, diagnostics, the
  // scratch row of the next step) and sets bit 7 of its host-visible terminated byte.  mir_step_end then steps exactly those envs on the
  // wave-per-env kernel (48 points, no thinning) from the untouched state rows and recomputes their scratch rows (`env_list`).
  int exact;
  int over_cap;  // phase 4 / 5 (three contacts per lane): bit 6 of an env's terminated byte says that it had more candidate points than this (0: never set)
  // phase 1 only: the launch serves the envs env_list[0 .. B) (B = the list's length) instead of envs 0 .. B; may point into pinned host memory
  const int32_t* env_list;
  // EXACT CONTACTS, phases 4 / 6 (three contacts per lane): K48_STRIDE floats per env -- the scratch row of an env whose NEXT step has 17 .. 48
  // contacts (head, row constants and unpacked Jacobian rows of all of them; the mass-matrix rows and the bias force stay in `pre`).  The
  // head of the env's row in `pre` then holds ncon = 0, the count in the `coupled` word and K48_MAGIC in its third word: a launch of the
  // one-contact-per-lane kernel defers the env on the count, phase 6 picks the big row up.  Null: such rows are not written.
  float* pre_big;
  // phase 7: device address of pinned host words, one per workgroup -- byte k = term_tag << 1 | (the NEXT step finds the workgroup's env k
  // with more candidate points than over_cap) -- or null
  uint32_t* next_host;
};
#define K48_HEAD 0    /* ncon, second-tree flags of contacts 16 .. 31, of 32 .. 47 (int bits), pad */
#define K48_CMETA 4   /* MIR_MAX_CONTACT x 4 */
#define K48_JB (K48_CMETA + 4 * MIR_MAX_CONTACT) /* MIR_MAX_CONTACT rows of 48 floats (n, t1, t2 x 16 dofs) */
#define K48_STRIDE (((K48_JB + 48 * MIR_MAX_CONTACT) + 15) / 16 * 16)
#define K48_MAGIC 0x42494721 /* "BIG!" */
#define K16_PRE_MROW 0     /* 16 lanes x 16: rows of the regularised mass matrix */
#define K16_PRE_BIAS 256   /* 16: qfrc_bias */
#define K16_PRE_HEAD 272   /* ncon, coupled (int bits), 2 pad */
#define K16_PRE_CMETA 276  /* K16_MAX_CONTACT x 4 */
#define K16_PRE_JB (K16_PRE_CMETA + 4 * K16_MAX_CONTACT) /* K16_MAX_CONTACT rows of 52 floats */
#define K16_PRE_STRIDE (((K16_PRE_JB + 52 * K16_MAX_CONTACT) + 15) / 16 * 16)


// enqueue the fused kernel on `stream`; returns a hipError_t as int
extern "C" __attribute__((visibility("hidden"))) int mir_launch_step(const StepArgs* args, int max_contacts_lds, hipStream_t stream);
extern "C" __attribute__((visibility("hidden"))) int mir_launch_debug_convex(const float* in, float* out, int n, hipStream_t stream);

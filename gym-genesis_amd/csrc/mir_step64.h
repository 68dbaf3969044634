// mir_step64.h — launch arguments of the wave-per-env step kernel (shared by mir_step64.hip and mir_api.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mir_model64.h"

struct StepArgs64 {
  const DevModel64* model;
  float* qpos;     // (B, 64)
  float* qvel;     // (B, 64) indexed by lane
  float* target;   // (B, 64) indexed by lane
  float* qacc_ws;  // (B, 64) indexed by lane
  float* poses;    // (B, 2, 32, 4): link positions then quaternions of the current qpos (FK cache)
  int32_t* fkvalid;  // (B)
  const float* action;  // (B, nu) or null
  float* agent_pos;     // (B, agent_dim) or null
  float* env_state;     // (B, env_dim) or null
  float* reward;        // (B) or null
  uint8_t* terminated;  // (B) or null
  uint8_t* term_host;   // (B) device address of pinned host memory (mir_step_begin), or null; byte = terminated | term_tag << 1
  uint32_t term_tag;
  int32_t* diag;        // (B, 4): ncon, nefc, niter, ncand; or null
  uint32_t* bad_count;  // device counter of env-steps that ended with a non-finite state (counted while diag is set), or null
  // per-stage parity outputs (mir_forward), all nullable; compact dof order
  float* out_M;     // (B, nv, nv)
  float* out_bias;  // (B, nv)
  float* out_qas;   // (B, nv)
  float* out_qacc;  // (B, nv)
  float* out_xpos;  // (B, nbody, 3)
  float* out_xquat; // (B, nbody, 4)
  unsigned long long* prof;  // debug: phase timestamps (shader clock) of env 0, or null
  float* rows;      // (B, row_stride) packed [agent | env_state | reward | terminated] or null
  int row_stride;
  long act_step;   // floats between the action blocks of consecutive steps (rollout mode), 0 = one action for all steps
  long rows_step;  // floats between the row blocks of consecutive steps (rollout mode), 0 = only the final row
  AutoResetArgs ar;  // per-step episode bookkeeping + re-spawn inside the launch (rollout mode only)
  int B;
  int nu;       // action width (copy of the model's: the action load does not wait for the model)
  int mode;     // 0: full steps; 1: forward dynamics only; 2: kinematics + outputs only
  int n_steps;  // mode 0 only
  int convex;   // the scene has sphere / capsule geoms (DevModel64.has_convex): instantiation with the convex narrowphase
  // Dispatch order of the single-step launch (nullable): cost_in[e] != 0 says env e was expensive in the previous step (blocks
  // coupled by a contact, or three and more Newton iterations); workgroup r then serves the r-th env of "expensive first" within
  // its chunk of 4096 envs instead of env r -- a launch ends with its slowest workgroup, and an expensive env that starts in the
  // last round of workgroups is what it then waits for.  cost_out[e]: this step's verdict, read by the NEXT launch (two buffers,
  // swapped by the host: every workgroup of a launch sees the same snapshot, so the order is a permutation).
  const uint8_t* cost_in;
  uint8_t* cost_out;
  // LIST MODE (exact contacts of a scene that lives on the 16-lane kernel: mir_set_exact_contacts, mir_step_end).  The launch serves the
  // envs env_list[0 .. B) (B = the list's length, grid = B; the list may sit in pinned host memory) of a handle whose state rows are in
  // the 16-LANE kernel's layout: qpos (B_handle, lay16_qst), qvel / target / qacc_ws (B_handle, 16) indexed by DOF (this kernel's own
  // rows are 64 wide and indexed by lane).  Outputs and diagnostics go to the env's own rows; the host-visible terminated byte of list
  // entry k goes to term_host[k].  No pose cache (poses / fkvalid null), no dispatch order (cost_in / cost_out null).
  const int32_t* env_list;  // null = the envs 0 .. B of a handle of this kernel
  int lay16_qst;            // list mode: row stride of qpos in the 16-lane layout (> 0); 0 = this kernel's own layout
};

extern "C" __attribute__((visibility("hidden"))) int mir_launch_step64(const StepArgs64* args, hipStream_t stream);

// mir_scene.h — the object behind a MirHandle (shared by mir_api.hip and mir_render.hip)
#pragma once
#include <stdint.h>

#include "mir_model.h"
#include "mir_model64.h"

struct MirScene {
  int device;
  int B;
  int kernel;       // 16: 16-lanes-per-env kernel (DevModel); 64: wave-per-env kernel (DevModel64)
  DevModel hm;      // host copy of the compiled model (kernel 16)
  DevModel64 hm64;  // host copy of the compiled model (kernel 64)
  HostConsts hc;
  PlumbTab pt;      // dof-order <-> storage maps (host copy)
  int nbody, nv, nq, nu, ngeom, npair, agent_dim, env_dim;
  DevModel* dm;     // device copies
  DevModel64* dm64;
  PlumbTab* dpt;
  GeomTab* dgeom;
  float *qpos, *qvel, *target, *qacc_ws, *poses;
  int32_t *diag, *fkvalid;
  uint8_t* cost = nullptr;  // wave kernel: two buffers of per-env cost flags (B padded to 64 each) for the dispatch order, see mir_step64.h
  int cost_par = 0;         // which of the two the next single-step launch reads
  int cost_stride = 0;
  uint32_t* early_stats = nullptr;  // 16-lane kernel: [1] mismatches of the early terminated bytes, [2 + w] launches in which workgroup w sent early (mir_step.h)
  int no_early_mask = 0;    // MIR_NO_EARLY_MASK=1, or a mismatch was seen (MIR_E_MASK)
  int spec_pick = 0;        // 16-lane kernel: the compiled model matches SpecPick (mir_spec_pick.h) and MIR_NO_SPEC is unset: specialised instantiation
  float* prims;     // render primitives (B, ngeom, 32) f32, allocated by the first mir_render
  int* bins = nullptr;      // per-strip primitive lists of the binned pixel kernel (grown on demand)
  size_t bins_cap = 0;      // ints
  int render_th = 0;        // strip height override (mir_debug_render_path; 0 = default)
  int render_generic = 0;   // force the generic pixel kernel (mir_debug_render_path)
  unsigned long long* zbuf = nullptr;  // depth / colour buffer of the global view of many envs (mir_render.hip, k_global_splat)
  size_t zbuf_cap = 0;
  unsigned* vis = nullptr;         // [0] boxes with a non-empty screen rectangle this render, [1 ..] their indices (k_render_setup -> k_global_splat)
  size_t vis_cap = 0;
  unsigned long long state_version = 0;  // bumped by every call that changes qpos (launches that integrate, resets, state writes): mir_get_state_version
  int poses_live = 0;       // a render has been asked for: step launches also leave their final link poses in `poses` (mir_api.hip, launch())
  int poses_current = 0;    // `poses` matches qpos for every env (cleared by resets and state writes): mir_render skips the pose-refresh launch
  // host-visible tail of env.step() (mir_step_begin / mir_step_end): one pinned, device-mapped allocation
  //   [terminated bytes (B, padded to 64) | completion word]
  float* scratch_row;       // one qpos row (device), used while the scene is created
  uint8_t* pin_host;        // host address
  size_t pin_flag_off = 0;  // byte offset of the completion word behind the terminated area
  int term_wstride = 0;     // 16-lane kernel: 32-bit words between the terminated words of consecutive workgroups in the pinned area (16 = one
                            // 64-byte line each, see mir_step_end); 0 = one byte per env (wave kernel)
  uint8_t* pin_dev;         // the same memory as the device sees it
  uint32_t* done_ticket;    // device counter for the kernel-side completion (sync mode 2)
  uint32_t seq;             // sequence number of the completion word (sync modes 1 / 2, mir_debug_null_roundtrip)
  uint32_t tag;             // tag of the last mir_step_begin's terminated bytes, 1..127; advances in mir_step_begin ONLY
  int sync_mode;            // 0 hipStreamSynchronize, 1 stream write-value + host spin, 2 kernel-side ticket + host spin
  int diag_on;              // step kernels write the per-env diagnostics (mir_set_diag)
  int pending;              // a mir_step_begin is waiting for its mir_step_end
  void* pending_stream;
  void* prep[4];            // output pointers registered by mir_step_prepare for the next mir_step_go
  int prepared;
  unsigned long long* dbg_prof = nullptr;  // (mir_debug_profile_next_step: shader-clock stamps of the next mir_step_begin launch)
  int32_t* dbg_ik_iters = nullptr;  // (mir_debug_ik_iters)
  unsigned long long* dbg_prof_list = nullptr;  // (mir_debug_profile_next_list_step: of the next launch of the list instantiation, exact contacts)
  // split step of the GenesisEnv.step path (16-lane kernel; MIR_SPLIT_STEP=0 switches it off): mir_step_begin launches the
  // action-independent half of the NEXT step right behind the current one; `pre_valid` says that `pre` holds that half for the
  // state as it is now (any other launch or state write clears it), `pre_stream` the stream it was launched on
  float* pre;
  int split_step, pre_valid;
  void* pre_stream;
  // EXACT CONTACTS (mir_set_exact_contacts; 16-lane scenes): the launches of mir_step_begin defer every env whose candidate contact
  // points exceed the 16-lane kernel's capacity (bit 7 of its terminated byte), and mir_step_end steps those envs on the wave kernel
  // (hm64 / dm64 = the same scene compiled for it with 48 points) from the untouched state rows, then recomputes their scratch rows.
  int exact;                // 0 off, 1 on, 2 (tests) every env is deferred: the whole batch takes the list instantiation
  int exact_big;            // the deferred envs take the list instantiation of the 16-lane kernel (three contacts per lane); 0: the wave-per-env kernel (MIR_EXACT_WAVE=1)
  int heavy;                // the coming mir_step_begin steps the WHOLE batch with three contacts per lane (one launch: mir_step.hip VARIANT 7); decided by mir_step_end
  int heavy_enter, heavy_leave;  // thresholds of that decision, in envs (MIR_EXACT_HEAVY)
  int pend_heavy;           // the pending step is such a launch
  int32_t* perm_host[2];    // pinned, device-mapped: the order in which a heavy launch serves the envs (the ones above 16 points first), double-buffered
  int32_t* perm_dev[2];
  int perm_next, pend_perm; // which of the two the next heavy launch takes / the pending one took (-1: the identity)
  int heavy_sort;           // MIR_EXACT_HEAVY_SORT=0: never permute
  unsigned long long ex_heavy_steps;
  int ovf_event_live;       // ovf_event has been recorded on the side stream behind launches the NEXT step must come after
  void* ovf_waited_stream;  // the stream that has been made to wait for it
  unsigned long long ex_big_envs;  // deferred env-steps handed to the list instantiation (those it deferred again included)
  unsigned long long ex_wave_envs; // env-steps stepped by the wave-per-env kernel
  int32_t* ovf_list_host;   // pinned, device-mapped: the deferred envs of the step being closed (B x i32), read in place by the two launches
  int32_t* ovf_list_dev;
  uint8_t* ovf_term_host;   // pinned: terminated byte of list entry k, tagged like the others (behind the list in the same allocation)
  uint8_t* ovf_term_dev;
  void* ovf_stream;         // hipStream_t of the library's own: the two launches for the deferred envs run beside the launch that deferred them
  void* ovf_event;          // hipEvent_t: recorded behind them; the step's stream waits for it before anything queued after mir_step_end
  const float* pend_action; // arguments of the pending mir_step_begin (the wave launch of mir_step_end takes the same)
  void* pend_out[4];
  int pend_rotated;         // the pending step is ONE rotated launch (else: a launch followed by the first half of the next step for all envs)
  unsigned long long ex_steps, ex_ovf_steps, ex_ovf_envs, ex_ovf_max;  // steps closed / steps with deferred envs / deferred env-steps / most in one step
  // OVERFLOW RUNS as two launches per step (mir_step.hip VARIANT 9 / 10, StepArgs::phase 6 / 7): from the step after one that
  // deferred envs until a step in which no env is above 16 points, a mir_step_begin that finds the caller left room between two steps steps
  // the WHOLE batch with the three-contacts-per-lane instantiation -- the second half from the scratch rows (an env with 17 .. 48 contacts has
  // its row in pre_big) on the step's stream, then the first half of the next step with the 48-point capacity on the side stream.  An env
  // whose row the launch before could not write (the first step of a run: the one-contact-per-lane tail found it above 16 points) is
  // deferred once more and takes the fused pass of the list instantiation.
  float* pre_big;           // (B, K48_STRIDE) device
  int big_on;               // MIR_EXACT_BIG: 0 never (the heavy phase / list launches of the first session), 1 when the caller leaves room between two steps (default), 2 always
  int bigmode;              // a run is on: the coming mir_step_begin may take the two launches (decided by mir_step_end)
  int big_side;             // the first-half launch goes on the side stream (MIR_EXACT_BIG_SIDE=0, a test switch: on the step's stream)
  int pend_big;             // the pending step was launched that way
  unsigned long long ex_big_steps;
  double t_end_us, big_gap_us;  // wall clock of the last mir_step_end's return; what the caller must spend between two steps for the two-launch steps to be taken
  void* main_event;         // hipEvent_t: recorded behind the second-half launch of such a step (the side stream's first-half launch waits for it)
  // ... and its second half as TWO LISTS once the first-half launch of the step before has said which envs are above 16 points NOW
  // (StepArgs::next_host): those on the three-contacts-per-lane instantiation on the step's stream, the others -- one round of 40 KB
  // workgroups -- on the rotated launch's first pass (VARIANT 11) on the side stream
  uint32_t* next_host;      // pinned, device-mapped: (B + 3) / 4 words
  uint32_t* next_dev;
  int big_lists;            // (MIR_EXACT_BIG_LISTS=0: never)
  int rt_ok, rt_perm;       // the first-half launch of the step before wrote them, serving the envs through perm_host[rt_perm] (-1: in order)
  uint32_t rt_tag;          // ... with this tag
  void* light_event;        // hipEvent_t behind the list launch on the side stream: the step's stream waits for it
};

// library-internal helpers implemented in mir_api.hip
int mir_set_error(int code, const char* msg);
int mir_refresh_poses(MirScene* h, void* stream);  // make h->poses match qpos for every env (mode-2 launch)

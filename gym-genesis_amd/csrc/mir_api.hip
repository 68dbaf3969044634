// mir_api.hip — C ABI of libmirigid.so (declared in include/mirigid.h) and the small
// state-plumbing kernels around the fused step kernel (mir_step.hip).
//
// Everything is enqueued on the caller's HIP stream and nothing synchronises the device.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <new>

#include "mir_model.h"
#include "mir_scene.h"
#include "mir_spec_pick.h"
#include "mir_step.h"
#include "mir_step64.h"

namespace {

thread_local char g_err[512] = "";

int set_err(int code, const char* fmt, const char* detail = "") {
  snprintf(g_err, sizeof g_err, fmt, detail);
  return code;
}
int hip_fail(hipError_t e, const char* what) {
  snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(e));
  return MIR_E_HIP;
}
#define HIPCHK(call)                                   \
  do {                                                 \
    hipError_t _e = (call);                            \
    if (_e != hipSuccess) return hip_fail(_e, #call);  \
  } while (0)

struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

constexpr int TPB = 256;
inline int nblk(long n) { return (int)((n + TPB - 1) / TPB); }

// ---- plumbing kernels: 64 threads per env, thread c = compact dof / qpos column c.  Both step kernels are
// served through the PlumbTab maps (16-lane kernel: lane == dof, rows of 16; wave kernel: lane map, rows of 64).
constexpr int PW = 64;
__global__ void k_fill_rows(float* dst, const float* row, int stride, int B) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long)B * stride) dst[i] = row[i % stride];
}

__device__ __forceinline__ void reset_env(const PlumbTab* __restrict__ t, int e, int c, float* qpos, float* qvel, float* target, float* ws,
                                          const float* free_pos, const float* free_quat, const float* arm_qpos, int32_t* fkvalid) {
  const int qst = t->qst, vst = t->vst;
  if (c < vst) {
    qvel[(long)e * vst + c] = 0.0f;
    ws[(long)e * vst + c] = 0.0f;
  }
  if (c == 0) fkvalid[e] = 0;  // cached link poses no longer match qpos
  if (c < t->nv) {
    int ai = t->d_armidx[c];
    if (ai >= 0 && arm_qpos) {
      float v = arm_qpos[(long)e * t->narm + ai];
      qpos[(long)e * qst + t->d_qadr[c]] = v;
      target[(long)e * vst + t->d_lane[c]] = v;
    }
  }
  if (c < 7 * t->nfree) {
    const int k = c / 7, j = c - 7 * k;
    if (j < 3 && free_pos) qpos[(long)e * qst + t->free_qadr[k] + j] = free_pos[((long)e * t->nfree + k) * 3 + j];
    if (j >= 3 && free_quat) qpos[(long)e * qst + t->free_qadr[k] + j] = free_quat[((long)e * t->nfree + k) * 4 + (j - 3)];
  }
}

__global__ void k_reset(const PlumbTab* __restrict__ t, float* qpos, float* qvel, float* target, float* ws, const float* obj_pos,
                        const float* obj_quat, const float* arm_qpos, const uint8_t* mask, int32_t* fkvalid, int B) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int e = (int)(i / PW), c = (int)(i % PW);
  if (e >= B) return;
  if (mask && !mask[e]) return;
  reset_env(t, e, c, qpos, qvel, target, ws, obj_pos, obj_quat, arm_qpos, fkvalid);
}

// Episode bookkeeping + re-spawn of finished envs, all on the device (no host round trip).
// The 64 consecutive threads of an env are one wave: every thread reads the env's counters before lane c==0 rewrites them.
__global__ void k_autoreset(const PlumbTab* __restrict__ t, float* qpos, float* qvel, float* target, float* ws, const uint8_t* terminated,
                            int32_t* episode_len, int max_len, const float* spawn_pool, int pool_len, int32_t* cursor,
                            const float* obj_quat, const float* arm_qpos, uint8_t* truncated_out, uint8_t* done_out, int32_t* fkvalid, int B) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int e = (int)(i / PW), c = (int)(i % PW);
  if (e >= B) return;
  const int len = episode_len[e] + 1;
  const bool term = terminated && terminated[e];
  const bool trunc = !term && max_len > 0 && len >= max_len;
  const bool done = term || trunc;
  const int cur = cursor[e];
  __builtin_amdgcn_wave_barrier();
  if (c == 0) {
    episode_len[e] = done ? 0 : len;
    if (done) cursor[e] = cur + 1;
    if (truncated_out) truncated_out[e] = trunc;
    if (done_out) done_out[e] = done;
  }
  if (!done) return;
  // spawn_pool is (pool_len, B, nfree, 3): the draw of this env is a contiguous (nfree, 3) slab at row offset e within its draw
  const float* sp = spawn_pool + (long)(cur % pool_len) * B * t->nfree * 3;
  reset_env(t, e, c, qpos, qvel, target, ws, sp, obj_quat, arm_qpos, fkvalid);
}

__global__ void k_set_targets(const PlumbTab* __restrict__ t, float* target, const float* tgt, int B) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int e = (int)(i / PW), c = (int)(i % PW);
  if (e >= B || c >= t->nv) return;
  int u = t->d_uadr[c];
  if (u >= 0) target[(long)e * t->vst + t->d_lane[c]] = tgt[(long)e * t->nu + u];
}

// dir 0: internal -> external (get); 1: external -> internal (set)
__global__ void k_copy_state(const PlumbTab* __restrict__ t, float* iq, float* iv, float* it, float* iw, float* eq, float* ev,
                             float* et, float* ew, int32_t* fkvalid, int B, int dir) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int e = (int)(i / PW), c = (int)(i % PW);
  if (e >= B) return;
  const int qst = t->qst, vst = t->vst, nq = t->nq, nv = t->nv, nu = t->nu;
  if (dir && eq && c == 0) fkvalid[e] = 0;
  if (eq && c < nq) {
    if (dir) iq[(long)e * qst + c] = eq[(long)e * nq + c];
    else eq[(long)e * nq + c] = iq[(long)e * qst + c];
  }
  if (c < nv) {
    const int l = t->d_lane[c];
    if (ev) {
      if (dir) iv[(long)e * vst + l] = ev[(long)e * nv + c];
      else ev[(long)e * nv + c] = iv[(long)e * vst + l];
    }
    if (ew) {
      if (dir) iw[(long)e * vst + l] = ew[(long)e * nv + c];
      else ew[(long)e * nv + c] = iw[(long)e * vst + l];
    }
    int u = t->d_uadr[c];
    if (et && u >= 0) {
      if (dir) it[(long)e * vst + l] = et[(long)e * nu + u];
      else et[(long)e * nu + u] = it[(long)e * vst + l];
    }
  }
}

__global__ void k_get_diag(const int32_t* diag, int32_t* ncon, int32_t* nefc, int32_t* niter, int B) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B) return;
  if (ncon) ncon[e] = diag[(long)e * 4 + 0];
  if (nefc) nefc[e] = diag[(long)e * 4 + 1];
  if (niter) niter[e] = diag[(long)e * 4 + 2];
}

__global__ void k_get_diag_points(const int32_t* diag, int32_t* points, int B) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < B) points[e] = (diag[(long)e * 4 + 3] >> 8) & 255;
}

__global__ void k_get_bad(const int32_t* diag, uint8_t* bad, int B) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < B) bad[e] = (uint8_t)((diag[(long)e * 4 + 3] >> 30) & 1);
}

// launch arguments common to both kernels
struct Outs {
  const float* action = nullptr;
  float *agent_pos = nullptr, *env_state = nullptr, *reward = nullptr;
  uint8_t* terminated = nullptr;
  uint8_t* term_host = nullptr;
  uint32_t *done_ticket = nullptr, *done_flag = nullptr;
  uint32_t done_seq = 0, term_tag = 0;
  float *out_M = nullptr, *out_bias = nullptr, *out_qas = nullptr, *out_qacc = nullptr, *out_xpos = nullptr, *out_xquat = nullptr;
  float* rows = nullptr;
  int row_stride = 0, mode = 0, n_steps = 1;
  long act_step = 0, rows_step = 0;
  AutoResetArgs ar = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
  bool diag = true;
  bool poses = false;  // 16-lane kernel: also write the link poses into h->poses (the rasteriser reads them)
  int phase = 0;       // 16-lane kernel: 0 whole step, 1 / 2 the two halves of a split step, 3 rotated, 4 the list instantiation of exact contacts (mir_step.h)
  int exact = 0;       // 16-lane kernel: defer the envs with more candidate points than lanes (StepArgs::exact)
  int over_cap = 0;    // 16-lane kernel, phase 4 / 5 (StepArgs::over_cap)
  const int32_t* env_list = nullptr;  // 16-lane kernel, phase 1: serve the envs env_list[0 .. nlist) (StepArgs::env_list)
  int nlist = 0;
  uint32_t* next_host = nullptr;      // 16-lane kernel, phase 7 (StepArgs::next_host)
  unsigned long long* prof = nullptr;  // 16-lane kernel only (debug)
};

int launch(MirScene* h, const Outs& o, void* stream) {
  int rc;
  // (phase 4 -- the list instantiation of exact contacts -- steps the envs the pending step's launch deferred and writes THEIR scratch
  //  rows for the state it leaves: it completes that launch, the handle's bookkeeping is the pending step's)
  if (o.phase != 1 && o.phase != 7 && o.phase != 4 && o.mode != 2) h->pre_valid = 0;  // (whatever this launch is, the state it leaves is not the one `pre` was made from)
  // Link poses for the rasteriser.  Once a render has been asked for (poses_live), every launch that integrates also leaves the
  // link poses of its final state in h->poses -- its closing forward kinematics has them -- so that a render behind a step needs
  // no pose-refresh launch (6 us per 1024 envs).  poses_current: h->poses matches qpos for every env.
  const bool integrates = o.mode == 0 && o.phase != 1 && o.phase != 7;
  if (integrates && o.phase != 4 && o.phase != 8) h->state_version++;  // (phase 8: the second list of a step whose first list has counted)
  const bool wr_poses = o.poses || (h->poses_live && integrates);
  if (o.mode == 2 && o.poses) h->poses_current = 1;
  else if (integrates && o.phase != 4) h->poses_current = (h->kernel == 64 || wr_poses) && !o.ar.episode_len;  // (an in-kernel reset moves envs after the closing FK)
  if (h->kernel == 16) {
    StepArgs a;
    memset(&a, 0, sizeof a);
    a.model = h->dm;
    a.qpos = h->qpos; a.qvel = h->qvel; a.target = h->target; a.qacc_ws = h->qacc_ws;
    a.poses = wr_poses ? h->poses : nullptr;
    a.early_stats = h->early_stats; a.no_early_mask = h->no_early_mask;
    a.term_bad = h->pin_dev ? reinterpret_cast<uint32_t*>(h->pin_dev + h->pin_flag_off + 16) : nullptr;
    a.term_wstride = h->term_wstride;
    a.diag = (o.diag && h->diag_on) ? h->diag : nullptr;
    a.B = h->B; a.qst = h->hm.qstride; a.nu = h->hm.nu; a.features = (h->hm.has_convex ? 1 : 0) | (h->hm.use_sap ? 3 : 0) | ((h->spec_pick && (!wr_poses || o.phase >= 3)) ? 4 : 0);
    a.action = o.action; a.agent_pos = o.agent_pos; a.env_state = o.env_state; a.reward = o.reward; a.terminated = o.terminated;
    a.out_M = o.out_M; a.out_bias = o.out_bias; a.out_qas = o.out_qas; a.out_qacc = o.out_qacc; a.out_xpos = o.out_xpos; a.out_xquat = o.out_xquat;
    a.rows = o.rows; a.row_stride = o.row_stride; a.mode = o.mode; a.n_steps = o.n_steps; a.prof = o.prof;
    a.act_step = o.act_step; a.rows_step = o.rows_step; a.ar = o.ar;
    a.term_host = o.term_host; a.term_tag = o.term_tag; a.done_ticket = o.done_ticket; a.done_flag = o.done_flag; a.done_seq = o.done_seq;
    a.phase = o.phase; a.pre = h->pre;
    a.exact = o.exact; a.over_cap = o.over_cap;
    a.pre_big = (o.phase == 4 || o.phase == 6 || o.phase == 7) ? h->pre_big : nullptr;
    a.next_host = o.next_host;
    if (o.env_list) { a.env_list = o.env_list; a.B = o.nlist; }
    if (o.phase == 4) a.term_wstride = 1;  // (the terminated byte of list entry k is byte k of term_host)
    rc = mir_launch_step(&a, h->hm.max_contacts, (hipStream_t)stream);
  } else {
    StepArgs64 a;
    memset(&a, 0, sizeof a);
    a.model = h->dm64;
    a.qpos = h->qpos; a.qvel = h->qvel; a.target = h->target; a.qacc_ws = h->qacc_ws;
    a.poses = h->poses; a.fkvalid = h->fkvalid;
    a.diag = (o.diag && h->diag_on) ? h->diag : nullptr;
    a.bad_count = h->early_stats;  // (word 0)
    a.B = h->B; a.nu = h->hm64.nu; a.convex = h->hm64.has_convex;
    a.action = o.action; a.agent_pos = o.agent_pos; a.env_state = o.env_state; a.reward = o.reward; a.terminated = o.terminated;
    a.out_M = o.out_M; a.out_bias = o.out_bias; a.out_qas = o.out_qas; a.out_qacc = o.out_qacc; a.out_xpos = o.out_xpos; a.out_xquat = o.out_xquat;
    a.rows = o.rows; a.row_stride = o.row_stride; a.mode = o.mode; a.n_steps = o.n_steps; a.prof = o.prof;
    a.act_step = o.act_step; a.rows_step = o.rows_step; a.ar = o.ar;
    a.term_host = o.term_host; a.term_tag = o.term_tag;
    auto order = [&]() {  // the single-step launches read one cost buffer and write the other
      if (!h->cost) return;
      a.cost_in = h->cost + (size_t)h->cost_par * h->cost_stride;
      a.cost_out = h->cost + (size_t)(h->cost_par ^ 1) * h->cost_stride;
      h->cost_par ^= 1;
    };
    if (a.mode == 0 && a.n_steps > 1 && a.act_step && a.rows_step && !a.ar.episode_len && !a.prof) {
      // a plain K-step rollout of the wave kernel is K launches of its two-wave single-step instantiation: at ~140 us a step the
      // launch gap is nothing, and the single-step code is faster than the step loop (which carries the loop's register spills)
      const int K = a.n_steps;
      const float* act0 = a.action;
      float* rows0 = a.rows;
      const long as = a.act_step, rs = a.rows_step;
      a.n_steps = 1; a.act_step = 0; a.rows_step = 0;
      rc = 0;
      for (int k = 0; k < K && rc == 0; k++) {
        a.action = act0 + (size_t)k * as;
        a.rows = rows0 + (size_t)k * rs;
        order();
        rc = mir_launch_step64(&a, (hipStream_t)stream);
      }
    } else {
      if (a.mode == 0 && a.n_steps == 1) order();
      rc = mir_launch_step64(&a, (hipStream_t)stream);
    }
  }
  if (rc != 0) return hip_fail((hipError_t)rc, "step kernel launch");
  return MIR_OK;
}

int check(MirHandle h) {
  if (!h) return set_err(MIR_E_INVALID, "null MirHandle");
  return MIR_OK;
}

// words of the early-mask counters: mismatches + one `sent` counter per workgroup of the 16-lane kernel (mir_step.h)
size_t early_words(const MirScene* h) { return 2 + ((size_t)h->B + 3) / 4; }

// The sticky word a step kernel raises when terminated bytes it sent early differ from the integrated state (mir_step.hip).  Checked
// by every entry point of the env.step path: fails ONCE with MIR_E_MASK and switches the handle to late bytes.
int check_mask(MirScene* h) {
  if (!h->pin_host) return MIR_OK;
  volatile uint32_t* bad = reinterpret_cast<volatile uint32_t*>(h->pin_host + h->pin_flag_off + 16);
  if (*bad == 0u) return MIR_OK;
  *bad = 0u;
  h->no_early_mask = 1;
  return set_err(MIR_E_MASK, "terminated bytes sent before the solve had finished differ from the integrated state: a mask returned since the last "
                             "successful call was wrong; early bytes are now off for this handle (MIR_NO_EARLY_MASK=1 avoids them from the start)");
}

// device half of mir_create: every allocation lands in the handle at once, so the caller can release a partially built
// scene with mir_destroy whichever call failed
int create_device_state(MirScene* h, const GeomTab& gt, const float* row, size_t row_bytes) {
  DeviceGuard guard(h->device);
  const PlumbTab& t = h->pt;
  const size_t B = (size_t)h->B;
  const size_t qst = t.qst, vst = t.vst, pst = t.pst;
  HIPCHK(hipMalloc((void**)&h->dm, sizeof(DevModel)));
  HIPCHK(hipMalloc((void**)&h->dm64, sizeof(DevModel64)));
  HIPCHK(hipMalloc((void**)&h->dpt, sizeof(PlumbTab)));
  HIPCHK(hipMalloc((void**)&h->dgeom, sizeof(GeomTab)));
  HIPCHK(hipMalloc((void**)&h->qpos, B * qst * sizeof(float)));
  HIPCHK(hipMalloc((void**)&h->qvel, B * vst * sizeof(float)));
  HIPCHK(hipMalloc((void**)&h->target, B * vst * sizeof(float)));
  HIPCHK(hipMalloc((void**)&h->qacc_ws, B * vst * sizeof(float)));
  HIPCHK(hipMalloc((void**)&h->diag, B * 4 * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&h->poses, B * 2 * pst * 4 * sizeof(float)));
  HIPCHK(hipMalloc((void**)&h->fkvalid, B * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&h->done_ticket, 64));
  HIPCHK(hipMalloc((void**)&h->scratch_row, row_bytes));
  if (h->kernel == 16) HIPCHK(hipMalloc((void**)&h->pre, B * K16_PRE_STRIDE * sizeof(float)));
  h->spec_pick = h->kernel == 16 && SpecPick::matches(h->hm) && !getenv("MIR_NO_SPEC");
  h->no_early_mask = getenv("MIR_NO_EARLY_MASK") ? 1 : 0;
  HIPCHK(hipMalloc((void**)&h->early_stats, early_words(h) * sizeof(uint32_t)));
  HIPCHK(hipMemset(h->early_stats, 0, early_words(h) * sizeof(uint32_t)));
  if (h->kernel == 64) {  // dispatch-order flags of the wave kernel (mir_step64.h): two buffers, padded to whole 64-byte reads
    h->cost_stride = (int)(((B + 63) / 64) * 64);
    HIPCHK(hipMalloc((void**)&h->cost, 2 * (size_t)h->cost_stride));
    HIPCHK(hipMemset(h->cost, 0, 2 * (size_t)h->cost_stride));
  }
  // pinned, device-mapped, coherent host memory for the API's host-visible outputs: terminated bytes + completion word
  // 16-lane kernel: every workgroup (4 envs) owns a whole 64-byte line of the area.  Sixteen workgroups storing their 4 bytes into one
  // line -- sixteen partial writes over PCIe into a line the polling CPU holds -- cost the launch that follows 1.6 us and now and then
  // several (tools/probes/launch/late_enqueue3.hip: idle gap between two kernels 4.7 us with 16, 8 or 4 writers per line, 3.9 with two,
  // 3.05 with one = as with no host stores at all)
  h->term_wstride = h->kernel == 16 ? 16 : 0;
  h->pin_flag_off = h->term_wstride ? ((B + 3) / 4) * 4 * (size_t)h->term_wstride : ((B + 63) / 64) * 64;
  h->pin_flag_off = (h->pin_flag_off + 63) / 64 * 64;
  const size_t pin_bytes = h->pin_flag_off + 64;
  HIPCHK(hipHostMalloc((void**)&h->pin_host, pin_bytes, hipHostMallocMapped | hipHostMallocCoherent));
  memset(h->pin_host, 0, pin_bytes);
  HIPCHK(hipHostGetDevicePointer((void**)&h->pin_dev, h->pin_host, 0));
  HIPCHK(hipMemcpy(h->dm, &h->hm, sizeof(DevModel), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->dm64, &h->hm64, sizeof(DevModel64), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->dpt, &h->pt, sizeof(PlumbTab), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->dgeom, &gt, sizeof(GeomTab), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(h->scratch_row, row, row_bytes, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_fill_rows, dim3(nblk((long)B * qst)), dim3(TPB), 0, 0, h->qpos, h->scratch_row, (int)qst, h->B);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemset(h->qvel, 0, B * vst * sizeof(float)));
  HIPCHK(hipMemset(h->target, 0, B * vst * sizeof(float)));
  HIPCHK(hipMemset(h->qacc_ws, 0, B * vst * sizeof(float)));
  HIPCHK(hipMemset(h->diag, 0, B * 4 * sizeof(int32_t)));
  HIPCHK(hipMemset(h->poses, 0, B * 2 * pst * 4 * sizeof(float)));
  HIPCHK(hipMemset(h->fkvalid, 0, B * sizeof(int32_t)));
  HIPCHK(hipMemset(h->done_ticket, 0, 64));
  HIPCHK(hipDeviceSynchronize());
  // how mir_step_end learns that a step has finished (MIR_SYNC_MODE overrides; see include/mirigid.h)
  h->diag_on = 1;
  h->sync_mode = 3;
  if (const char* e = getenv("MIR_SYNC_MODE")) h->sync_mode = atoi(e);
  if (h->sync_mode < 0 || h->sync_mode > 3 || (h->sync_mode == 2 && h->kernel != 16)) h->sync_mode = 3;
  h->split_step = h->kernel == 16 ? 1 : 0;
  if (const char* e = getenv("MIR_SPLIT_STEP")) h->split_step = h->kernel == 16 ? (atoi(e) == 2 ? 2 : (atoi(e) != 0 ? 1 : 0)) : 0;
  h->pre_valid = 0;
  return MIR_OK;
}

}  // namespace

int mir_set_error(int code, const char* msg) { return set_err(code, "%s", msg); }

int mir_refresh_poses(MirScene* h, void* stream) {
  Outs o;
  o.mode = 2; o.diag = false; o.poses = true;
  return launch(h, o, stream);
}

extern "C" {

int mir_version(void) { return MIR_VERSION; }
int mir_spec_sizeof(void) { return (int)sizeof(MirSceneSpec); }
const char* mir_last_error(void) { return g_err; }

int mir_create(const MirSceneSpec* spec, int32_t num_envs, int32_t device_id, MirHandle* out) {
  if (!out) return set_err(MIR_E_INVALID, "mir_create: out is null");
  *out = nullptr;
  if (num_envs <= 0) return set_err(MIR_E_INVALID, "mir_create: num_envs must be > 0");
  MirScene* h = new (std::nothrow) MirScene();
  if (!h) return set_err(MIR_E_INVALID, "out of host memory");
  memset(h, 0, sizeof *h);
  char err[256] = "";
  // the 16-lanes-per-env kernel when the scene fits it, else the wave-per-env kernel
  int rc = mir_compile_model(spec, &h->hm, &h->hc, err);
  h->kernel = 16;
  if (rc == MIR_E_CAPACITY) {
    rc = mir_compile_model64(spec, &h->hm64, &h->hc, err);
    h->kernel = 64;
  }
  if (rc != MIR_OK) {
    delete h;
    return set_err(rc, "mir_create: %s", err);
  }
  // storage maps + geometry table shared by the plumbing kernels and the rasteriser
  PlumbTab& t = h->pt;
  GeomTab gt;
  memset(&gt, 0, sizeof gt);
  float row[K64_QSTRIDE] = {0};  // qpos0: free bodies at their spec pose, scalar joints at 0
  if (h->kernel == 16) {
    const DevModel& m = h->hm;
    h->nbody = m.nbody; h->nv = m.nv; h->nq = m.nq; h->nu = m.nu; h->ngeom = m.ngeom; h->npair = m.npair;
    h->agent_dim = 7 + m.n_grip; h->env_dim = 11;
    t.nv = m.nv; t.nq = m.nq; t.nu = m.nu; t.nfree = m.nfree; t.narm = m.n_arm_q; t.qst = m.qstride; t.vst = MIR_G; t.pst = MIR_G;
    for (int i = 0; i < m.nv; i++) { t.d_lane[i] = i; t.d_uadr[i] = m.d_uadr[i]; t.d_armidx[i] = m.d_armidx[i]; t.d_qadr[i] = m.d_qadr[i]; }
    for (int k = 0; k < m.nfree; k++) t.free_qadr[k] = m.free_qadr[k];
    gt.ngeom = m.ngeom;
    for (int g = 0; g < m.ngeom; g++) {
      gt.g_body[g] = m.g_body[g]; gt.g_type[g] = m.g_type[g];
      for (int k = 0; k < 3; k++) { gt.g_size[g][k] = m.g_size[g][k]; gt.g_pos[g][k] = m.g_pos[g][k]; }
      if (m.g_type[g] == MIR_GEOM_HULL) {  // the rasteriser draws a hull as the bounding box of its vertices
        gt.g_type[g] = MIR_GEOM_BOX;
        for (int k = 0; k < 3; k++) gt.g_size[g][k] = m.g_bbox[g][k];
      }
      for (int k = 0; k < 4; k++) gt.g_quat[g][k] = m.g_quat[g][k];
    }
    for (int b = 1; b < m.nbody; b++)
      if (m.b_jtype[b] == MIR_JNT_FREE) {
        for (int k = 0; k < 3; k++) row[m.b_qadr[b] + k] = m.b_pos[b][k];
        for (int k = 0; k < 4; k++) row[m.b_qadr[b] + 3 + k] = m.b_quat[b][k];
      }
  } else {
    const DevModel64& m = h->hm64;
    h->nbody = m.nbody; h->nv = m.nv; h->nq = m.nq; h->nu = m.nu; h->ngeom = m.ngeom; h->npair = m.npair;
    h->agent_dim = m.agent_dim; h->env_dim = m.env_dim;
    t.nv = m.nv; t.nq = m.nq; t.nu = m.nu; t.nfree = m.nfree; t.narm = m.n_arm_q; t.qst = K64_QSTRIDE; t.vst = W64; t.pst = K64_MAX_BODY;
    int narm = 0, nfree = 0;
    for (int l = 0; l < W64; l++) {
      const int i = m.d_dof[l];
      if (i < 0) continue;
      t.d_lane[i] = l; t.d_uadr[i] = m.d_uadr[l]; t.d_qadr[i] = m.d_qadr[l]; t.d_armidx[i] = -1;
    }
    for (int b = 1; b < m.nbody; b++) {  // arm_qpos / free-body order = body order
      if (m.b_jtype[b] == MIR_JNT_FREE) {
        t.free_qadr[nfree++] = m.b_qadr[b];
        for (int k = 0; k < 3; k++) row[m.b_qadr[b] + k] = m.b_pos[b][k];
        for (int k = 0; k < 4; k++) row[m.b_qadr[b] + 3 + k] = m.b_quat[b][k];
      } else if (m.b_jtype[b] != MIR_JNT_FIXED) {
        for (int i = 0; i < m.nv; i++)
          if (t.d_qadr[i] == m.b_qadr[b]) t.d_armidx[i] = narm;
        narm++;
      }
    }
    gt.ngeom = m.ngeom;
    for (int g = 0; g < m.ngeom; g++) {
      gt.g_body[g] = m.g_body[g]; gt.g_type[g] = m.g_type[g];
      for (int k = 0; k < 3; k++) { gt.g_size[g][k] = m.g_size[g][k]; gt.g_pos[g][k] = m.g_pos[g][k]; }
      if (m.g_type[g] == MIR_GEOM_HULL) {  // the rasteriser draws a hull as the bounding box of its vertices
        gt.g_type[g] = MIR_GEOM_BOX;
        for (int k = 0; k < 3; k++) gt.g_size[g][k] = m.g_bbox[g][k];
      }
      for (int k = 0; k < 4; k++) gt.g_quat[g][k] = m.g_quat[g][k];
    }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) {
    delete h;
    return set_err(MIR_E_NODEVICE, "mir_create: no usable HIP device (libmirigid has no CPU path)");
  }
  h->device = device_id;
  h->B = num_envs;
  rc = create_device_state(h, gt, row, sizeof row);
  if (rc != MIR_OK) {
    mir_destroy(h);  // frees whatever was allocated before the failure (the error text is already set)
    return rc;
  }
  *out = h;
  return MIR_OK;
}

int mir_destroy(MirHandle h) {
  if (!h) return MIR_OK;
  DeviceGuard guard(h->device);
  if (h->dm) (void)hipFree(h->dm);
  if (h->dm64) (void)hipFree(h->dm64);
  if (h->dpt) (void)hipFree(h->dpt);
  if (h->dgeom) (void)hipFree(h->dgeom);
  if (h->qpos) (void)hipFree(h->qpos);
  if (h->qvel) (void)hipFree(h->qvel);
  if (h->target) (void)hipFree(h->target);
  if (h->qacc_ws) (void)hipFree(h->qacc_ws);
  if (h->diag) (void)hipFree(h->diag);
  if (h->poses) (void)hipFree(h->poses);
  if (h->fkvalid) (void)hipFree(h->fkvalid);
  if (h->early_stats) (void)hipFree(h->early_stats);
  if (h->prims) (void)hipFree(h->prims);
  if (h->cost) (void)hipFree(h->cost);
  if (h->bins) (void)hipFree(h->bins);
  if (h->zbuf) (void)hipFree(h->zbuf);
  if (h->vis) (void)hipFree(h->vis);
  if (h->done_ticket) (void)hipFree(h->done_ticket);
  if (h->scratch_row) (void)hipFree(h->scratch_row);
  if (h->pre) (void)hipFree(h->pre);
  if (h->pre_big) (void)hipFree(h->pre_big);
  if (h->main_event) (void)hipEventDestroy((hipEvent_t)h->main_event);
  if (h->light_event) (void)hipEventDestroy((hipEvent_t)h->light_event);
  if (h->next_host) (void)hipHostFree(h->next_host);
  if (h->pin_host) (void)hipHostFree(h->pin_host);
  if (h->ovf_list_host) (void)hipHostFree(h->ovf_list_host);
  if (h->ovf_event) (void)hipEventDestroy((hipEvent_t)h->ovf_event);
  if (h->ovf_stream) (void)hipStreamDestroy((hipStream_t)h->ovf_stream);
  delete h;
  return MIR_OK;
}

int mir_get_dims(MirHandle h, MirDims* out) {
  if (check(h) || !out) return set_err(MIR_E_INVALID, "mir_get_dims: null argument");
  out->num_envs = h->B; out->nbody = h->nbody; out->nq = h->nq; out->nv = h->nv;
  out->ngeom = h->ngeom; out->npair = h->npair; out->agent_dim = h->agent_dim; out->env_dim = h->env_dim;
  out->nfree = h->pt.nfree; out->kernel = h->kernel;
  return MIR_OK;
}

int mir_get_model_consts(MirHandle h, double* dof_invweight0, double* body_invweight0, double* meaninertia) {
  if (check(h)) return MIR_E_INVALID;
  if (dof_invweight0) memcpy(dof_invweight0, h->hc.dof_invweight0, sizeof(double) * h->nv);
  if (body_invweight0) memcpy(body_invweight0, h->hc.body_invweight0, sizeof(double) * h->nbody);
  if (meaninertia) *meaninertia = h->hc.meaninertia;
  return MIR_OK;
}

int mir_reset(MirHandle h, const float* obj_pos, const float* obj_quat, const float* arm_qpos, const uint8_t* env_mask, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  if (h->pending) {  // (a step left open by an exception on the caller's side: see mir_step_begin)
    int rc = mir_step_end(h, nullptr);
    if (rc != MIR_OK) return rc;
  }
  if (int rc = check_mask(h)) return rc;
  h->pre_valid = 0;
  if (!env_mask) { h->heavy = 0; h->perm_next = -1; h->bigmode = 0; }  // (exact contacts: a full reset ends a heavy phase -- every env is back at its start)
  h->poses_current = 0;
  h->state_version++;
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_reset, dim3(nblk((long)h->B * PW)), dim3(TPB), 0, (hipStream_t)stream, h->dpt, h->qpos, h->qvel, h->target,
                     h->qacc_ws, obj_pos, obj_quat, arm_qpos, env_mask, h->fkvalid, h->B);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_autoreset(MirHandle h, const uint8_t* terminated, int32_t* episode_len, int32_t max_len, const float* spawn_pool, int32_t pool_len,
                  int32_t* cursor, const float* obj_quat, const float* arm_qpos, uint8_t* truncated_out, uint8_t* done_out, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  if (!episode_len || !spawn_pool || !cursor || !obj_quat || !arm_qpos || pool_len <= 0) return set_err(MIR_E_INVALID, "mir_autoreset: null argument");
  h->pre_valid = 0;
  h->poses_current = 0;
  h->state_version++;
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_autoreset, dim3(nblk((long)h->B * PW)), dim3(TPB), 0, (hipStream_t)stream, h->dpt, h->qpos, h->qvel, h->target,
                     h->qacc_ws, terminated, episode_len, max_len, spawn_pool, pool_len, cursor, obj_quat, arm_qpos, truncated_out, done_out,
                     h->fkvalid, h->B);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_set_pd_targets(MirHandle h, const float* tgt, void* stream) {
  if (check(h) || !tgt) return set_err(MIR_E_INVALID, "mir_set_pd_targets: null argument");
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_set_targets, dim3(nblk((long)h->B * PW)), dim3(TPB), 0, (hipStream_t)stream, h->dpt, h->target, tgt, h->B);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_step(MirHandle h, int32_t n_steps, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  if (n_steps <= 0) return MIR_OK;
  if (h->exact) {  // (exact contacts: every step is closed on the host, where the deferred envs are handed to the wave kernel)
    for (int i = 0; i < n_steps; i++) {
      int rc = mir_step_begin(h, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
      if (rc == MIR_OK) rc = mir_step_end(h, nullptr);
      if (rc != MIR_OK) return rc;
    }
    return MIR_OK;
  }
  DeviceGuard guard(h->device);
  Outs o;
  o.n_steps = n_steps;
  return launch(h, o, stream);
}

int mir_step_fused(MirHandle h, const float* action, float* agent_pos, float* env_state, float* reward, uint8_t* terminated, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  if (h->exact) {  // (exact contacts: begin + end -- the call then waits for the step's terminated bytes)
    const int rc = mir_step_begin(h, action, agent_pos, env_state, reward, terminated, stream);
    return rc != MIR_OK ? rc : mir_step_end(h, nullptr);
  }
  DeviceGuard guard(h->device);
  Outs o;
  o.action = action; o.agent_pos = agent_pos; o.env_state = env_state; o.reward = reward; o.terminated = terminated;
  return launch(h, o, stream);
}

/* GenesisEnv.step in two halves (reference env.py:61-69).  mir_step_begin = mir_step_fused that ALSO stores the terminated bytes
 * into the handle's pinned host buffer and arranges for a completion word; mir_step_end blocks until that launch has finished and
 * copies the B bytes to the caller's plain host array -- `terminated = is_success.detach().cpu().numpy()` (env.py:64) without a
 * separate copy command.  The host is free between the two calls (the Python side allocates the next outputs there). */
static double wall_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
// (developer aid, MIR_EXACT_TIMING=1: host time stamps of the steps of an overflow run, microseconds since mir_step_begin's entry, to stderr)
static int g_timing = -1;
static double g_t0 = 0.0, g_ts[6];
#define TSTAMP(i) do { if (g_timing > 0) g_ts[i] = wall_us() - g_t0; } while (0)

/* A step of an overflow run whose predecessor's first-half launch has said which envs are above the one-contact-per-lane capacity NOW
 * (h->next_host, tagged h->rt_tag, in the order of perm_host[h->rt_perm]): the order of this step's launches -- those envs first, padded to
 * whole workgroups with others, then the rest -- into perm_host[*buf]; *nh = how many go to the three-contacts-per-lane list. */
static int split_lists(MirScene* h, int* buf, int* nh_out) {
  const size_t B = (size_t)h->B, nwg = (B + 3) / 4;
  const uint32_t want4 = 0x01010101u * (uint8_t)h->rt_tag, tagm4 = 0x1f1f1f1fu;
  const int32_t* const pp = h->rt_perm >= 0 ? h->perm_host[h->rt_perm] : nullptr;
  const volatile uint32_t* w = h->next_host;
  // (straight into the pinned order: the envs above the capacity from the front, the others from the back)
  const int nxt = h->rt_perm == 0 ? 1 : 0;
  int32_t* const out = h->perm_host[nxt];
  size_t nh = 0, lo = B;
  unsigned long polls = 0;
  for (size_t g = 0; g < nwg;) {
    const uint32_t v = w[g];
    if (((v >> 1) & tagm4) == want4) {
      for (size_t k = 0; k < 4 && 4 * g + k < B; k++) {
        const int32_t e = pp ? pp[4 * g + k] : (int32_t)(4 * g + k);
        if (v >> (8 * k) & 1u) out[nh++] = e; else out[--lo] = e;
      }
      g++;
      continue;
    }
    __builtin_ia32_pause();
    if ((++polls & 0xfffffu) == 0) {
      hipError_t e = hipStreamQuery((hipStream_t)h->ovf_stream);
      if (e != hipSuccess && e != hipErrorNotReady) return hip_fail(e, "mir_step_begin: side stream (exact contacts)");
      if (e == hipSuccess && polls > 0x4000000u) return set_err(MIR_E_HIP, "mir_step_begin: the first-half launch finished without saying which envs are above 16 points");
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  // (whole workgroups: the first of the others join the list of the bigger instantiation, which steps an env with few contacts to the same
  //  bits -- they sit right behind it in the array already)
  while ((nh & 3) && nh < B) nh++;
  __atomic_thread_fence(__ATOMIC_RELEASE);
  *buf = nxt;
  *nh_out = (int)nh;
  return MIR_OK;
}
int mir_step_begin(MirHandle h, const float* action, float* agent_pos, float* env_state, float* reward, uint8_t* terminated, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  // a step left open (an exception between the two calls on the Python side) is closed here: its bytes are waited for and dropped
  if (h->pending) {
    int rc = mir_step_end(h, nullptr);
    if (rc != MIR_OK) return rc;
  }
  if (int rc = check_mask(h)) return rc;
  DeviceGuard guard(h->device);
  uint32_t* flag_dev = reinterpret_cast<uint32_t*>(h->pin_dev + h->pin_flag_off);
  Outs o;
  o.action = action; o.agent_pos = agent_pos; o.env_state = env_state; o.reward = reward; o.terminated = terminated;
  o.term_host = h->pin_dev;
  const uint32_t seq = h->seq + 1u;
  // The tag has a counter of its own that nothing else advances (h->seq is shared with mir_debug_null_roundtrip and wraps): 1 .. 63,
  // never 0 (fresh memory), and a launch's tag differs from those of the 30 launches before it -- each of which overwrote every byte.
  // (Five bits: bit 7 of a byte says "deferred" -- exact contacts, StepArgs::exact -- and bit 6 "more points than the one-contact-per-lane
  //  kernel holds", from the launches that step the whole batch with three contacts per lane, StepArgs::over_cap.)
  const uint32_t tag = h->tag % 31u + 1u;
  o.term_tag = tag;
  if (h->sync_mode == 2) { o.done_ticket = h->done_ticket; o.done_flag = flag_dev; o.done_seq = seq; }
  // split step: if the previous mir_step_begin left the action-independent half of THIS step in `pre` (same stream, nothing
  // touched the state since), only the other half is launched now
  // exact contacts, HEAVY phase (mir_step_end decides): most envs of the last step had more points than the one-contact-per-lane kernel
  // holds, so the whole batch takes the three-contacts-per-lane instantiation's first pass in ONE launch (mir_step.hip: VARIANT 7) -- no
  // launch that defers, no list launch behind it, no scratch rows (the first light step afterwards is launched like the one behind a reset)
  // exact contacts, an OVERFLOW RUN in a loop that leaves room between two steps (mir_scene.h): the step as TWO launches of the
  // three-contacts-per-lane instantiation for the whole batch -- second half (-> terminated bytes), then, on the side stream, the first
  // half of the next step, which runs beside whatever the caller queues between two steps (the policy, its IK).  Costs GPU time (rows
  // through HBM, two rounds of workgroups twice) and saves time to the bytes; otherwise the heavy phase / the list launch behind the
  // bytes, which cost less GPU time.
  if (g_timing < 0) g_timing = (getenv("MIR_EXACT_TIMING") && atoi(getenv("MIR_EXACT_TIMING")) != 0) ? 1 : 0;
  if (g_timing > 0) { g_t0 = wall_us(); for (int i = 0; i < 6; i++) g_ts[i] = 0.0; }
  bool bigrot = false;
  if (h->exact == 1 && h->exact_big && h->bigmode && h->pre_big != nullptr && h->sync_mode == 3 && h->ovf_stream && h->split_step && h->pre_valid &&
      h->pre_stream == stream && h->hm.fk_free_leaf != 0) {
    // (does the caller leave the first-half launch room to hide?  The time it spent between mir_step_end's return and this call: the
    //  policy and its IK in the reference's expert loop, ~90 us; a loop that does nothing between two steps, 2 - 5 us)
    if (h->big_on == 2) bigrot = true;  // (MIR_EXACT_BIG=2: whenever the rows are there)
    else bigrot = h->t_end_us > 0.0 && wall_us() - h->t_end_us >= h->big_gap_us;
  }
  const bool heavy = h->exact && h->exact_big && h->heavy && h->sync_mode == 3 && !bigrot;
  const bool split = h->split_step && h->sync_mode != 2 && !heavy;
  const bool have_pre = split && h->pre_valid && h->pre_stream == stream;
  // (one rotated launch -- this step's second half, then the next step's first half -- where the closing FK can be shared between
  //  the waves; otherwise two launches)
  const bool rotated = have_pre && h->hm.fk_free_leaf != 0 && h->split_step != 2 && !bigrot;
  o.phase = heavy ? 5 : (bigrot ? 6 : (rotated ? 3 : (have_pre ? 2 : 0)));
  o.exact = h->exact;
  if (bigrot) {
    o.over_cap = h->hm.max_contacts < K16_MAX_CONTACT ? h->hm.max_contacts : K16_MAX_CONTACT;
    h->ex_big_steps++;
  }
  h->pend_big = bigrot ? 1 : 0;
  h->pend_perm = -1;
  if (heavy) {
    o.over_cap = h->hm.max_contacts < K16_MAX_CONTACT ? h->hm.max_contacts : K16_MAX_CONTACT;
    h->ex_heavy_steps++;
    // (the envs in the order mir_step_end left: the ones above 16 points first -- workgroups of like cost, the expensive ones early)
    if (h->perm_next >= 0) { o.env_list = h->perm_dev[h->perm_next]; o.nlist = h->B; h->pend_perm = h->perm_next; }
  }
  // (a step of an overflow run serves the envs in the order mir_step_end left too: the ones above 16 points first -- their workgroups are
  //  the long ones, and the step's terminated bytes wait for the last of them)
  int nh_split = -1;  // (>= 0: the step's second half goes out as two lists, perm[0 .. nh) and perm[nh .. B))
  const int rt_ok_prev = h->rt_ok;
  h->rt_ok = 0;
  if (bigrot && rt_ok_prev && h->big_side && h->next_host && h->big_lists) {
    int buf = -1;
    if (int rc = split_lists(h, &buf, &nh_split)) return rc;
    h->pend_perm = buf;
    if (nh_split == 0 || nh_split >= h->B) {  // (one list after all)
      o.phase = nh_split == 0 ? 8 : 6;
      o.env_list = h->perm_dev[buf]; o.nlist = h->B;
      nh_split = -1;
    }
  } else if (bigrot && h->perm_next >= 0) { o.env_list = h->perm_dev[h->perm_next]; o.nlist = h->B; h->pend_perm = h->perm_next; }
  h->pend_heavy = heavy ? 1 : 0;
  // (exact contacts, ADVICE r5: the launches for the deferred envs of an earlier step ran on the library's side stream, and only the stream
  //  of THAT step was made to wait for them; a step on another stream waits for them here -- state rows, scratch rows and the pinned
  //  list are theirs until then)
  if (h->ovf_event_live && h->ovf_waited_stream != stream) {
    HIPCHK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)h->ovf_event, 0));
    h->ovf_waited_stream = stream;
  }
  h->pend_action = action;
  h->pend_out[0] = agent_pos; h->pend_out[1] = env_state; h->pend_out[2] = reward; h->pend_out[3] = terminated;
  h->pend_rotated = rotated ? 1 : 0;
  o.prof = h->dbg_prof;
  h->dbg_prof = nullptr;
  int rc;
  if (nh_split >= 0) {
    // the envs above 16 points on the three-contacts-per-lane instantiation (80 KB workgroups, the long ones: first, on the step's stream),
    // the others in ONE round of the one-contact-per-lane kernel's 40 KB workgroups on the side stream; the step's stream waits for those
    Outs oh = o, ol = o;
    oh.env_list = h->perm_dev[h->pend_perm]; oh.nlist = nh_split;
    ol.phase = 8; ol.over_cap = 0; ol.prof = nullptr;
    ol.env_list = h->perm_dev[h->pend_perm] + nh_split; ol.nlist = h->B - nh_split;
    ol.term_host = h->pin_dev + (size_t)(nh_split / 4) * h->term_wstride * sizeof(uint32_t);
    // (the side stream's launch reads the caller's action too: behind whatever produced it on the step's stream -- an event recorded
    //  there IN FRONT of the first list's launch, or the second list would wait for the first)
    HIPCHK(hipEventRecord((hipEvent_t)h->main_event, (hipStream_t)stream));
    rc = launch(h, oh, stream);
    if (rc != MIR_OK) return rc;
    HIPCHK(hipStreamWaitEvent((hipStream_t)h->ovf_stream, (hipEvent_t)h->main_event, 0));
    rc = launch(h, ol, h->ovf_stream);
    if (rc == MIR_OK) {
      HIPCHK(hipEventRecord((hipEvent_t)h->light_event, (hipStream_t)h->ovf_stream));
      HIPCHK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)h->light_event, 0));
    }
  } else {
    rc = launch(h, o, stream);
  }
  if (rc != MIR_OK) return rc;
  TSTAMP(0);
  // the launch that carries the tag is queued: from here on the step is pending whatever happens to the calls behind it
  h->seq = seq;
  h->tag = tag;
  h->pending = 1;
  h->pending_stream = stream;
  if (rotated) h->pre_valid = 1;  // (launch() cleared it; the same launch has refilled `pre` for the state it leaves)
  if (h->sync_mode == 1) {
    hipError_t e = hipStreamWriteValue32((hipStream_t)stream, flag_dev, seq, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); h->sync_mode = 0; }  // not supported on this stack: wait on the stream instead
  }
  if (split && !rotated) {
    // ... and the action-independent half of the NEXT step goes out right behind it: it runs while the host is between two calls
    Outs p;
    p.phase = bigrot ? 7 : 1; p.diag = false;
    if (bigrot && h->big_side) {
      // (... on the side stream, behind the launch above: beside the caller's work between two steps; the next mir_step_begin -- and the
      //  launches for envs this step defers -- come behind it through ovf_event)
      if (h->pend_perm >= 0) { p.env_list = h->perm_dev[h->pend_perm]; p.nlist = h->B; }
      if (h->next_host) {  // (... and says which envs the next step finds above the one-contact-per-lane capacity: split_lists)
        memset(h->next_host, 0, ((size_t)h->B + 3) / 4 * sizeof(uint32_t));  // (tags come round every 31 steps)
        __atomic_thread_fence(__ATOMIC_RELEASE);
        p.next_host = h->next_dev; p.term_tag = tag;
        p.over_cap = h->hm.max_contacts < K16_MAX_CONTACT ? h->hm.max_contacts : K16_MAX_CONTACT;
      }
      HIPCHK(hipEventRecord((hipEvent_t)h->main_event, (hipStream_t)stream));
      HIPCHK(hipStreamWaitEvent((hipStream_t)h->ovf_stream, (hipEvent_t)h->main_event, 0));
      rc = launch(h, p, h->ovf_stream);
      if (rc != MIR_OK) return rc;
      if (h->next_host) { h->rt_ok = 1; h->rt_perm = h->pend_perm; h->rt_tag = tag; }
      HIPCHK(hipEventRecord((hipEvent_t)h->ovf_event, (hipStream_t)h->ovf_stream));
      h->ovf_event_live = 1;
      h->ovf_waited_stream = reinterpret_cast<void*>(~(uintptr_t)0);  // (no stream has been made to wait yet -- the null stream is a stream)
    } else {
      if (bigrot && h->pend_perm >= 0) { p.env_list = h->perm_dev[h->pend_perm]; p.nlist = h->B; }
      rc = launch(h, p, stream);
    }
    if (rc != MIR_OK) return rc;  // (the step itself is queued and pending: the caller may still close it, or the next begin does)
    h->pre_valid = 1;
    h->pre_stream = stream;
  }
  TSTAMP(1);
  return MIR_OK;
}

/* The same launch with its output pointers registered ahead of time: mir_step_prepare costs nothing on the device and is called
 * while the PREVIOUS step's kernel is still running, so that the call in front of which the GPU idles -- mir_step_go -- carries
 * three arguments instead of seven. */
int mir_step_prepare(MirHandle h, float* agent_pos, float* env_state, float* reward, uint8_t* terminated) {
  if (check(h)) return MIR_E_INVALID;
  h->prep[0] = agent_pos; h->prep[1] = env_state; h->prep[2] = reward; h->prep[3] = terminated;
  h->prepared = 1;
  return MIR_OK;
}

int mir_step_go(MirHandle h, const float* action, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  if (!h->prepared) return set_err(MIR_E_INVALID, "mir_step_go without mir_step_prepare");
  const int rc = mir_step_begin(h, action, (float*)h->prep[0], (float*)h->prep[1], (float*)h->prep[2], (uint8_t*)h->prep[3], stream);
  if (rc == MIR_OK || h->pending) h->prepared = 0;  // (consumed once a launch is queued; a call that failed before that may be repeated)
  return rc;
}

/* exact contacts: the n envs of h->ovf_list_host were deferred by the launch(es) of the pending mir_step_begin (their state rows are
 * those of the step's start).  They are stepped here by the LIST INSTANTIATION of the 16-lane kernel (mir_step.hip, VARIANT 6: three
 * contacts per lane, 48 points, four envs per workgroup) -- one launch that takes the action and the output pointers of the pending
 * step, stores state, observations and the terminated byte of list entry k into ovf_term_host[k], and then writes the envs' scratch
 * rows for the NEXT step (which also say whether they are deferred again).  An env beyond THAT kernel's capacity (more than 48 points,
 * or more than 16 candidate pairs: its byte comes back with bit 7 set, nothing stored) goes on a second list and takes the route every
 * deferred env took in round 5: the wave-per-env kernel in list mode (the same scene compiled for it, 64 candidates, reading and
 * writing the 16-lane kernel's rows) followed by the action-independent half of the 16-lane kernel over that list.  MIR_EXACT_WAVE=1
 * (read by mir_set_exact_contacts), or a scene without the split closing FK, sends every deferred env that way.
 * The launches go on a stream of the library's own, BESIDE the launch that deferred these envs (which is still running its second
 * half: the host is here because that launch's terminated bytes -- early bytes -- have arrived): that launch stores nothing for
 * them, and everything queued before it on the step's stream has finished, or it would not be running.  The step's stream then
 * waits for an event recorded behind them, so that whatever the caller queues after mir_step_end -- the next step, a
 * policy network reading the observations -- comes after them. */
static int exact_wait(MirScene* h, const uint8_t* term, const int32_t* list, int n, uint8_t* terminated_host, int32_t* again, int* n_again) {
  const uint8_t want = (uint8_t)h->tag;
  unsigned long polls = 0;
  for (int k = 0; k < n;) {
    const uint8_t b = __atomic_load_n(term + k, __ATOMIC_RELAXED);
    if ((uint8_t)((b >> 1) & 0x1fu) == want) {
      if ((b & 0x80u) && again) again[(*n_again)++] = list[k];
      else if (terminated_host) terminated_host[list[k]] = b & 1u;
      k++;
      continue;
    }
    __builtin_ia32_pause();
    if ((++polls & 0xfffffu) == 0) {
      hipError_t e = hipStreamQuery((hipStream_t)h->pending_stream);
      if (e != hipSuccess && e != hipErrorNotReady) return hip_fail(e, "mir_step_end: stream (exact contacts)");
      if (e == hipSuccess && polls > 0x4000000u) return set_err(MIR_E_HIP, "mir_step_end: the launch for the deferred envs finished without delivering its terminated bytes");
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  return MIR_OK;
}

static int exact_finish(MirScene* h, int n, uint8_t* terminated_host) {
  DeviceGuard guard(h->device);
  h->ex_ovf_steps++;
  h->ex_ovf_envs += (unsigned long long)n;
  if ((unsigned long long)n > h->ex_ovf_max) h->ex_ovf_max = (unsigned long long)n;
  // (not beside a step that was launched as two kernels -- a fused launch, or the second half alone, followed by the first half of the
  //  next step for ALL envs: that second kernel writes the scratch rows of the deferred envs too, from their old state, and must come
  //  BEFORE the one below that writes them from the new state: stream order does that)
  void* const side = (h->ovf_stream && (h->pend_rotated || h->pend_big)) ? h->ovf_stream : h->pending_stream;  // (pend_big: behind the first-half launch, which is there)
  const size_t B = (size_t)h->B;
  int32_t* const list2_host = reinterpret_cast<int32_t*>(h->ovf_term_host + (B + 63) / 64 * 64);
  int32_t* const list2_dev = reinterpret_cast<int32_t*>(h->ovf_term_dev + (B + 63) / 64 * 64);
  uint8_t* const term2_host = reinterpret_cast<uint8_t*>(list2_host + B);
  uint8_t* const term2_dev = reinterpret_cast<uint8_t*>(list2_dev + B);
  const int32_t* wlist_host = h->ovf_list_host;
  const int32_t* wlist_dev = h->ovf_list_dev;
  const uint8_t* wterm_host = h->ovf_term_host;
  uint8_t* wterm_dev = h->ovf_term_dev;
  int nw = n;  // envs for the wave-per-env kernel
  if (h->pend_heavy) {
    // (a heavy step: these envs were beyond the three-contacts-per-lane capacity already -- straight to the wave-per-env kernel; the
    //  statistics of a heavy step count the envs above the one-contact-per-lane capacity, mir_step_end)
    h->ex_ovf_steps--; h->ex_ovf_envs -= (unsigned long long)n;
  } else if (h->exact_big) {
    // (a step of an overflow run: envs whose big row the launch before could not write -- the fused pass of the list instantiation, which
    //  hands on what is beyond its capacity too; the step's statistics come from bit 6 of the bytes, mir_step_end)
    if (h->pend_big) { h->ex_ovf_steps--; h->ex_ovf_envs -= (unsigned long long)n; }
    memset(h->ovf_term_host, 0, (size_t)n);  // (tags come round every 63 steps: a byte of an older step must not pass for this one's)
    __atomic_thread_fence(__ATOMIC_RELEASE);
    Outs o;
    o.action = h->pend_action;
    o.agent_pos = (float*)h->pend_out[0]; o.env_state = (float*)h->pend_out[1]; o.reward = (float*)h->pend_out[2]; o.terminated = (uint8_t*)h->pend_out[3];
    o.term_host = h->ovf_term_dev; o.term_tag = h->tag;
    o.phase = 4; o.env_list = h->ovf_list_dev; o.nlist = n;
    o.prof = h->dbg_prof_list;
    h->dbg_prof_list = nullptr;
    int rc = launch(h, o, side);
    if (rc != MIR_OK) return rc;
    h->ex_big_envs += (unsigned long long)n;
    nw = 0;
    rc = exact_wait(h, h->ovf_term_host, h->ovf_list_host, n, terminated_host, list2_host, &nw);
    if (rc != MIR_OK) return rc;
    wlist_host = list2_host; wlist_dev = list2_dev; wterm_host = term2_host; wterm_dev = term2_dev;
  }
  if (nw) {
    memset(const_cast<uint8_t*>(wterm_host), 0, (size_t)nw);
    __atomic_thread_fence(__ATOMIC_RELEASE);
    StepArgs64 a;
    memset(&a, 0, sizeof a);
    a.model = h->dm64;
    a.qpos = h->qpos; a.qvel = h->qvel; a.target = h->target; a.qacc_ws = h->qacc_ws;
    a.diag = h->diag_on ? h->diag : nullptr;
    a.bad_count = h->early_stats;
    a.B = nw; a.nu = h->hm64.nu; a.convex = h->hm64.has_convex;
    a.action = h->pend_action;
    a.agent_pos = (float*)h->pend_out[0]; a.env_state = (float*)h->pend_out[1]; a.reward = (float*)h->pend_out[2]; a.terminated = (uint8_t*)h->pend_out[3];
    a.term_host = wterm_dev; a.term_tag = h->tag;
    a.mode = 0; a.n_steps = 1;
    a.env_list = wlist_dev; a.lay16_qst = h->hm.qstride;
    h->ex_wave_envs += (unsigned long long)nw;
    int rc = mir_launch_step64(&a, (hipStream_t)side);
    if (rc != 0) return hip_fail((hipError_t)rc, "wave kernel launch (exact contacts)");
    h->poses_current = 0;  // (these envs' link poses were not written)
    if (h->pre_valid) {  // (split step: the scratch rows of the coming step, for the envs that have only now reached its starting state)
      Outs p;
      p.phase = 1; p.diag = false; p.env_list = wlist_dev; p.nlist = nw;
      rc = launch(h, p, side);
      if (rc != MIR_OK) return rc;
    }
  }
  if (side != h->pending_stream) {
    HIPCHK(hipEventRecord((hipEvent_t)h->ovf_event, (hipStream_t)side));
    HIPCHK(hipStreamWaitEvent((hipStream_t)h->pending_stream, (hipEvent_t)h->ovf_event, 0));
    h->ovf_event_live = 1;
    h->ovf_waited_stream = h->pending_stream;
  }
  if (nw) return exact_wait(h, wterm_host, wlist_host, nw, terminated_host, nullptr, nullptr);
  return MIR_OK;
}

int mir_step_end(MirHandle h, uint8_t* terminated_host) {
  if (check(h)) return MIR_E_INVALID;
  if (!h->pending) return set_err(MIR_E_INVALID, "mir_step_end without mir_step_begin");
  // (early bytes of an EARLIER launch that turned out wrong: reported here, before this step's bytes are handed over; the step stays
  //  pending and is closed by the next call)
  if (int rc = check_mask(h)) return rc;
  h->pending = 0;
  volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(h->pin_host + h->pin_flag_off);
  const uint8_t* bytes = h->pin_host;
  const size_t B = (size_t)h->B;
  if (h->sync_mode == 0) {
    DeviceGuard guard(h->device);
    HIPCHK(hipStreamSynchronize((hipStream_t)h->pending_stream));
  } else if (h->sync_mode == 3 && h->term_wstride) {
    // 16-lane kernel: one 32-bit word (4 envs) per workgroup, `term_wstride` words apart; every byte carries this launch's tag.  One
    // pass: wait for a word, take its bits, go on to the next (the pointer stands still at the first workgroup that has not delivered)
    // (exact contacts: bit 7 of a byte = the env was deferred by the launch -- collected here, stepped by exact_finish)
    const uint32_t want4 = 0x01010101u * (uint8_t)h->tag, tagm4 = 0x1f1f1f1fu;
    const volatile uint32_t* w = reinterpret_cast<const volatile uint32_t*>(bytes);
    const size_t nwg = (B + 3) / 4, ws = (size_t)h->term_wstride;
    unsigned long polls = 0;
    int ndefer = 0, nover = 0;
    TSTAMP(2);
    h->ex_steps++;
    // (a launch of a heavy phase may have served the envs in a permuted order: byte k of workgroup g is env pp[4 g + k])
    const int32_t* const pp = (h->exact && h->pend_perm >= 0) ? h->perm_host[h->pend_perm] : nullptr;
    for (size_t g = 0; g < nwg;) {
      uint32_t v = w[g * ws];
      if (((v >> 1) & tagm4) == want4) {
        if ((v & 0x80808080u) && h->exact) {
          for (size_t k = 0; k < 4 && 4 * g + k < B; k++)
            if (v >> (8 * k + 7) & 1u) h->ovf_list_host[ndefer++] = pp ? pp[4 * g + k] : (int32_t)(4 * g + k);
        }
        nover += __builtin_popcount(v & 0x40404040u);
        if (terminated_host) {
          if (pp) {
            for (size_t k = 0; k < 4 && 4 * g + k < B; k++) terminated_host[pp[4 * g + k]] = (uint8_t)(v >> (8 * k) & 1u);
          } else {
            v &= 0x01010101u;
            memcpy(terminated_host + 4 * g, &v, 4 * g + 4 <= B ? 4 : B - 4 * g);
          }
        }
        g++;
        continue;
      }
      __builtin_ia32_pause();
      if ((++polls & 0xfffffu) == 0) {
        DeviceGuard guard(h->device);
        hipError_t e = hipStreamQuery((hipStream_t)h->pending_stream);
        if (e != hipSuccess && e != hipErrorNotReady) return hip_fail(e, "mir_step_end: stream");
        if (e == hipSuccess && polls > 0x4000000u)
          return set_err(MIR_E_HIP, "mir_step_end: the launch finished without delivering its terminated bytes");
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    TSTAMP(3);
    if (h->exact && h->exact_big) {
      // light -> heavy when this step deferred at least heavy_enter envs; heavy -> light when fewer than heavy_leave had more points than
      // the one-contact-per-lane kernel holds (MIR_EXACT_HEAVY="enter,leave"; enter <= 0: never heavy).  The cost model behind the
      // defaults: a light step with a deferred list costs launch + list launch, a heavy one two rounds of the bigger kernel.
      if (h->pend_heavy || h->pend_big) { h->ex_ovf_envs += (unsigned long long)nover; if (nover) h->ex_ovf_steps++; if ((unsigned long long)nover > h->ex_ovf_max) h->ex_ovf_max = nover; }
      // an overflow run starts behind the first step that deferred an env and ends with the first of its steps in which no env is above 16 points
      if (h->big_on) {
        if (!h->bigmode && ndefer > 0) h->bigmode = 1;
        else if (h->pend_big && nover == 0 && ndefer == 0) h->bigmode = 0;
      }
      const int cnt = h->pend_heavy ? nover : ndefer;
      if (!h->heavy && h->heavy_enter > 0 && cnt >= h->heavy_enter) h->heavy = 1;
      else if (h->heavy && cnt < h->heavy_leave) h->heavy = 0;
      // the order of the NEXT heavy launch: the envs that were above 16 points in this step first (bit 6 of a heavy launch's bytes, bit
      // 7 -- deferred -- of a light one's), the others behind them.  A workgroup serves four consecutive entries and lasts as long as its
      // slowest env: with 70 % of the envs at 20 - 38 points and the rest at 4 - 12, unsorted 99 % of the workgroups hold a slow env;
      // sorted, 30 % of them are done in half the time, and the expensive ones are dispatched first (the words are in this core's
      // cache: the loop above has just read them)
      h->perm_next = -1;
      if ((h->heavy || h->bigmode) && h->heavy_sort && !(h->pend_big && h->rt_ok && !h->heavy)) {  // (rt_ok: the next step of the run takes its order from the first-half launch's words)
        const int nxt = h->pend_perm == 0 ? 1 : 0;
        int32_t* const out = h->perm_host[nxt];
        const uint32_t bit = (h->pend_heavy || h->pend_big) ? 0x40u : 0x80u;
        size_t nh = 0, nl = B;
        for (size_t g = 0; g < nwg; g++) {
          const uint32_t v = w[g * ws];
          for (size_t k = 0; k < 4 && 4 * g + k < B; k++) {
            const int32_t e = pp ? pp[4 * g + k] : (int32_t)(4 * g + k);
            if (v >> (8 * k) & bit) out[nh++] = e; else out[--nl] = e;
          }
        }
        h->perm_next = nxt;
      }
    }
    if (h->big_on) {
      const int rc = ndefer ? exact_finish(h, ndefer, terminated_host) : MIR_OK;
      h->t_end_us = wall_us();
      if (g_timing > 0 && (h->pend_big || ndefer))
        fprintf(stderr, "[exact timing] two launches %d, above 16 points %d, deferred %d | main launch queued %.1f, mir_step_begin returns %.1f, mir_step_end called %.1f, "
                "last byte %.1f, mir_step_end returns %.1f us\n", h->pend_big, nover, ndefer, g_ts[0], g_ts[1], g_ts[2], g_ts[3], h->t_end_us - g_t0);
      return rc;
    }
    if (ndefer) return exact_finish(h, ndefer, terminated_host);
    return MIR_OK;
  } else if (h->sync_mode == 3) {
    // the bytes announce themselves: wait until every one of them carries this launch's tag
    // (a pointer that walks the buffer once, eight bytes at a time: it stands still at the first workgroup that has not delivered
    // yet and touches every cache line once after the device's last write to it; polling the whole buffer instead was measured
    // slower -- the host keeps pulling lines the device is still writing)
    const uint8_t want = (uint8_t)h->tag;
    const uint64_t want8 = 0x0101010101010101ull * want, tagm = 0x1f1f1f1f1f1f1f1full;
    const volatile uint64_t* w8 = reinterpret_cast<const volatile uint64_t*>(bytes);
    const size_t nw = B / 8;
    size_t i = 0;  // in 8-byte words, then the tail bytes
    unsigned long polls = 0;
    while (i < nw + (B - nw * 8)) {
      const bool ok = i < nw ? (((w8[i] >> 1) & tagm) == want8)
                             : ((uint8_t)((__atomic_load_n(bytes + nw * 8 + (i - nw), __ATOMIC_RELAXED) >> 1) & 0x1fu) == want);
      if (ok) { i++; continue; }
      __builtin_ia32_pause();
      if ((++polls & 0xfffffu) == 0) {
        DeviceGuard guard(h->device);
        hipError_t e = hipStreamQuery((hipStream_t)h->pending_stream);
        if (e != hipSuccess && e != hipErrorNotReady) return hip_fail(e, "mir_step_end: stream");
        if (e == hipSuccess && polls > 0x4000000u)
          return set_err(MIR_E_HIP, "mir_step_end: the launch finished without delivering its terminated bytes");
      }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  } else {
    // spin on the pinned completion word; every 2^20 polls make sure the stream has not died under us
    unsigned long polls = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != h->seq) {
      __builtin_ia32_pause();
      if ((++polls & 0xfffffu) == 0) {
        DeviceGuard guard(h->device);
        hipError_t e = hipStreamQuery((hipStream_t)h->pending_stream);
        if (e != hipSuccess && e != hipErrorNotReady) return hip_fail(e, "mir_step_end: stream");
        if (e == hipSuccess && __atomic_load_n(flag, __ATOMIC_ACQUIRE) != h->seq)
          return set_err(MIR_E_HIP, "mir_step_end: the launch finished without publishing its completion word");
      }
    }
  }
  if (terminated_host && h->term_wstride) {  // (sync modes 0 - 2 on the 16-lane kernel: one word per workgroup, term_wstride words apart)
    const uint32_t* w = reinterpret_cast<const uint32_t*>(bytes);
    for (size_t g = 0; 4 * g < B; g++) {
      const uint32_t v = w[g * (size_t)h->term_wstride] & 0x01010101u;
      memcpy(terminated_host + 4 * g, &v, 4 * g + 4 <= B ? 4 : B - 4 * g);
    }
  } else if (terminated_host) {
    size_t i = 0;
    for (; i + 8 <= B; i += 8) {
      uint64_t v;
      memcpy(&v, bytes + i, 8);
      v &= 0x0101010101010101ull;
      memcpy(terminated_host + i, &v, 8);
    }
    for (; i < B; i++) terminated_host[i] = bytes[i] & 1u;
  }
  return MIR_OK;
}

int mir_get_sync_mode(MirHandle h) { return check(h) ? MIR_E_INVALID : h->sync_mode; }

int mir_set_exact_contacts(MirHandle h, const MirSceneSpec* spec, int32_t on) {
  if (check(h)) return MIR_E_INVALID;
  if (h->pending) return set_err(MIR_E_INVALID, "mir_set_exact_contacts: a step is pending");
  // (a scratch row whose head says "above 16 points, the contacts are elsewhere" means something to the launches of this mode only: the
  //  next step starts from the state, not from the rows)
  h->pre_valid = 0;
  h->bigmode = 0;
  if (!on) { h->exact = 0; return MIR_OK; }
  if (h->kernel != 16) return set_err(MIR_E_INVALID, "mir_set_exact_contacts: the scene already runs on the wave-per-env kernel (48 contact points, never thinned below that)");
  if (h->sync_mode != 3 || !h->term_wstride) return set_err(MIR_E_INVALID, "mir_set_exact_contacts: needs the tagged terminated bytes (sync mode 3)");
  if (!spec) return set_err(MIR_E_INVALID, "mir_set_exact_contacts: the scene's spec is needed once (it is compiled for the wave kernel)");
  if (!h->ovf_list_host) {
    // the same scene for the wave-per-env kernel, with that kernel's contact capacity
    MirSceneSpec* s2 = new (std::nothrow) MirSceneSpec(*spec);
    if (!s2) return set_err(MIR_E_INVALID, "out of host memory");
    s2->opt.max_contacts = MIR_MAX_CONTACT;
    HostConsts hc;
    char err[256] = "";
    const int rc = mir_compile_model64(s2, &h->hm64, &hc, err);
    delete s2;
    if (rc != MIR_OK) return set_err(rc, "mir_set_exact_contacts: %s", err);
    if (h->hm64.nv != h->hm.nv || h->hm64.nq != h->hm.nq || h->hm64.nu != h->hm.nu || h->hm64.agent_dim != h->agent_dim || h->hm64.env_dim != h->env_dim)
      return set_err(MIR_E_INVALID, "mir_set_exact_contacts: the spec is not the one this scene was created from");
    DeviceGuard guard(h->device);
    HIPCHK(hipMemcpy(h->dm64, &h->hm64, sizeof(DevModel64), hipMemcpyHostToDevice));
    // [list of the deferred envs (B x i32) | their terminated bytes (B, padded to 64)] twice: the envs the list instantiation could not
    // hold either go on the second list (exact_finish)
    // ... and two permutations of the envs for the launches of a heavy phase (B x i32 each, see mir_step_end)
    const size_t B = (size_t)h->B, bytes = 2 * (B * sizeof(int32_t) + ((B + 63) / 64) * 64) + 2 * ((B + 15) / 16 * 16) * sizeof(int32_t);
    HIPCHK(hipHostMalloc((void**)&h->ovf_list_host, bytes, hipHostMallocMapped | hipHostMallocCoherent));
    memset(h->ovf_list_host, 0, bytes);
    HIPCHK(hipHostGetDevicePointer((void**)&h->ovf_list_dev, h->ovf_list_host, 0));
    h->ovf_term_host = reinterpret_cast<uint8_t*>(h->ovf_list_host + B);
    h->ovf_term_dev = reinterpret_cast<uint8_t*>(h->ovf_list_dev + B);
    {
      const size_t half = B * sizeof(int32_t) + ((B + 63) / 64) * 64, pn = (B + 15) / 16 * 16;
      for (int i = 0; i < 2; i++) {
        h->perm_host[i] = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(h->ovf_list_host) + 2 * half) + i * pn;
        h->perm_dev[i] = reinterpret_cast<int32_t*>(reinterpret_cast<uint8_t*>(h->ovf_list_dev) + 2 * half) + i * pn;
      }
      h->perm_next = -1; h->pend_perm = -1;
    }
    if (!getenv("MIR_EXACT_ONE_STREAM")) {  // (the side stream of exact_finish; MIR_EXACT_ONE_STREAM=1: everything on the step's stream)
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      hipStream_t st = nullptr;
      hipEvent_t ev = nullptr;
      HIPCHK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
      HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      h->ovf_stream = st; h->ovf_event = ev;
    }
  }
  // the list instantiation of the 16-lane kernel takes the deferred envs where the scene has the split closing forward kinematics (every
  // free body a childless child of the world: the reference's scenes); MIR_EXACT_WAVE=1: the wave-per-env kernel takes them all (round 5)
  h->exact_big = (h->hm.fk_free_leaf != 0 && !(getenv("MIR_EXACT_WAVE") && atoi(getenv("MIR_EXACT_WAVE")) != 0)) ? 1 : 0;
  // (1, the default: when the caller spent at least MIR_EXACT_BIG_GAP microseconds -- 40 -- between the last mir_step_end and this
  //  mir_step_begin: the two launches of such a step take a third more GPU time than the heavy phase's one -- rows through HBM, less
  //  overlap of the two waves -- which a loop with nothing between its steps pays in full; 2: whenever the rows are there; 0: never)
  h->big_on = (h->exact_big && on != 2 && h->ovf_stream) ? (getenv("MIR_EXACT_BIG") ? atoi(getenv("MIR_EXACT_BIG")) : 1) : 0;
  h->big_gap_us = getenv("MIR_EXACT_BIG_GAP") ? atof(getenv("MIR_EXACT_BIG_GAP")) : 40.0;
  h->big_lists = !(getenv("MIR_EXACT_BIG_LISTS") && atoi(getenv("MIR_EXACT_BIG_LISTS")) == 0);  // (0: the second half always as one launch for the whole batch)
  h->big_side = !(getenv("MIR_EXACT_BIG_SIDE") && atoi(getenv("MIR_EXACT_BIG_SIDE")) == 0);  // (0, a test switch: the first-half launch on the step's stream)
  h->t_end_us = 0.0;
  h->bigmode = 0;
  if (h->big_on && !h->main_event) {
    DeviceGuard guard(h->device);
    hipEvent_t ev = nullptr, ev2 = nullptr;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ev2, hipEventDisableTiming));
    h->main_event = ev; h->light_event = ev2;
    const size_t nb = (((size_t)h->B + 3) / 4 * sizeof(uint32_t) + 63) / 64 * 64;
    HIPCHK(hipHostMalloc((void**)&h->next_host, nb, hipHostMallocMapped | hipHostMallocCoherent));
    memset(h->next_host, 0, nb);
    HIPCHK(hipHostGetDevicePointer((void**)&h->next_dev, h->next_host, 0));
  }
  h->rt_ok = 0;
  if (h->big_on && !h->pre_big) {
    DeviceGuard guard(h->device);
    HIPCHK(hipMalloc((void**)&h->pre_big, (size_t)h->B * K48_STRIDE * sizeof(float)));
    HIPCHK(hipMemset(h->pre_big, 0, (size_t)h->B * K48_STRIDE * sizeof(float)));
  }
  h->exact = on == 2 ? 2 : 1;
  h->heavy = 0;
  h->perm_next = -1; h->pend_perm = -1;
  h->heavy_sort = !(getenv("MIR_EXACT_HEAVY_SORT") && atoi(getenv("MIR_EXACT_HEAVY_SORT")) == 0);
  h->heavy_enter = (h->B + 15) / 16; h->heavy_leave = (h->B + 31) / 32;  // (6 % / 3 % of the batch)
  if (const char* e = getenv("MIR_EXACT_HEAVY")) {
    int a = 0, b = 0;
    const int k = sscanf(e, "%d,%d", &a, &b);
    if (k >= 1) { h->heavy_enter = a; h->heavy_leave = k >= 2 ? b : a / 2; }
  }
  if (on == 2) h->heavy_enter = 0;  // (the twin of the tests: every env on the list instantiation, every step)
  return MIR_OK;
}

int mir_get_exact_contacts(MirHandle h) { return check(h) ? MIR_E_INVALID : h->exact; }

int mir_get_exact_stats(MirHandle h, uint64_t* out4, int32_t reset) {
  if (check(h) || !out4) return set_err(MIR_E_INVALID, "mir_get_exact_stats: null argument");
  out4[0] = h->ex_steps; out4[1] = h->ex_ovf_steps; out4[2] = h->ex_ovf_envs; out4[3] = h->ex_ovf_max;
  if (reset) h->ex_steps = h->ex_ovf_steps = h->ex_ovf_envs = h->ex_ovf_max = h->ex_big_envs = h->ex_wave_envs = h->ex_heavy_steps = h->ex_big_steps = 0;
  return MIR_OK;
}

int mir_get_exact_route(MirHandle h, uint64_t* out2) {
  if (check(h) || !out2) return set_err(MIR_E_INVALID, "mir_get_exact_route: null argument");
  out2[0] = h->ex_big_envs; out2[1] = h->ex_wave_envs; out2[2] = h->ex_heavy_steps; out2[3] = h->ex_big_steps;
  return MIR_OK;
}
/* debug aid (bench.py's roofline): n back-to-back launches of the rotated step kernel (what mir_step_begin launches in split mode 1)
 * cycling through n_actions action blocks of (B, nu) and without observation outputs, so that two events around the call time that kernel the way the fused one is
 * timed.  Advances the state by n steps. */
extern "C" int mir_debug_rotated_launches(MirHandle h, const float* actions, int32_t n_actions, int32_t n, void* const* outputs, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  if (h->kernel != 16 || h->split_step != 1 || !h->hm.fk_free_leaf || h->exact) return set_err(MIR_E_INVALID, "mir_debug_rotated_launches: the scene does not use rotated launches (or exact contacts are on)");
  DeviceGuard guard(h->device);
  if (!(h->pre_valid && h->pre_stream == stream)) {
    Outs f; f.action = actions; f.diag = false;
    int rc = launch(h, f, stream);
    if (rc != MIR_OK) return rc;
    Outs p; p.phase = 1; p.diag = false;
    rc = launch(h, p, stream);
    if (rc != MIR_OK) return rc;
  }
  // (with `outputs` the launches write the four outputs of a mir_step_go launch.  Its host-visible terminated bytes are left out:
  //  1000 launches storing into the same 4 KB of pinned host memory back to back, with no host reading them, time the PCIe write
  //  path -- regions of 21.5, 32 and 85 us per launch were measured in one process -- not the kernel; inside the API loop, where
  //  the host consumes them, rocprofv3 has the kernel at 22.3 us)
  const bool outs = outputs != nullptr;
  for (int i = 0; i < n; i++) {
    Outs o; o.action = actions + (size_t)(i % (n_actions > 0 ? n_actions : 1)) * h->B * h->nu; o.diag = false; o.phase = 3;
    if (outs) {
      o.agent_pos = (float*)outputs[0]; o.env_state = (float*)outputs[1]; o.reward = (float*)outputs[2]; o.terminated = (uint8_t*)outputs[3];
    }
    int rc = launch(h, o, stream);
    if (rc != MIR_OK) return rc;
  }
  h->pre_valid = 1;
  h->pre_stream = stream;
  return MIR_OK;
}

/* debug aid (tools/probes/resident_steps.py): n steps in ONE launch on the real step body, every workgroup going
 * through its steps at its own pace (no launch boundary, no grid-wide wait for the slowest env of a step) -- what a launch that stays
 * resident over several env.step() calls would cost per step at best (raw-launch semantics: the actions of all n steps are resident,
 * nothing is handed to the host, no outputs).  It is the rollout launch without rows.  actions (n, B, nu). */
extern "C" int mir_debug_resident_steps(MirHandle h, const float* actions, int32_t n, void* stream) {
  if (check(h) || !actions || n <= 0) return set_err(MIR_E_INVALID, "mir_debug_resident_steps: bad argument");
  if (h->exact) return set_err(MIR_E_INVALID, "mir_debug_resident_steps: not available with exact contacts");
  DeviceGuard guard(h->device);
  Outs o;
  o.action = actions; o.n_steps = n; o.act_step = (long)h->B * h->nu; o.diag = false;
  return launch(h, o, stream);
}

int mir_get_split_step(MirHandle h) { return check(h) ? MIR_E_INVALID : (h->sync_mode == 2 ? 0 : h->split_step); }
int mir_debug_spec_active(MirHandle h) { return check(h) ? MIR_E_INVALID : h->spec_pick; }
int mir_debug_early_mask_stats(MirHandle h, uint32_t* out2, int32_t reset, void* stream) {
  if (check(h) || !out2) return set_err(MIR_E_INVALID, "mir_debug_early_mask_stats: null argument");
  DeviceGuard guard(h->device);
  const size_t n = early_words(h);
  uint32_t* tmp = new (std::nothrow) uint32_t[n];
  if (!tmp) return set_err(MIR_E_INVALID, "out of host memory");
  hipError_t e = hipMemcpyAsync(tmp, h->early_stats, n * sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e == hipSuccess && reset) e = hipMemsetAsync(h->early_stats + 1, 0, (n - 1) * sizeof(uint32_t), (hipStream_t)stream);  // (word 0 is mir_get_bad's)
  uint64_t sent = 0;
  for (size_t i = 2; i < n; i++) sent += tmp[i];
  out2[0] = sent > 0xffffffffull ? 0xffffffffu : (uint32_t)sent;
  out2[1] = tmp[1];
  delete[] tmp;
  if (e != hipSuccess) return hip_fail(e, "mir_debug_early_mask_stats");
  return MIR_OK;
}

int mir_get_diag4(MirHandle h, int32_t* ncon, int32_t* nefc, int32_t* niter, int32_t* ncand_points, void* stream) {
  int rc = mir_get_diag(h, ncon, nefc, niter, stream);
  if (rc != MIR_OK || !ncand_points) return rc;
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_get_diag_points, dim3(nblk(h->B)), dim3(TPB), 0, (hipStream_t)stream, h->diag, ncand_points, h->B);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_get_bad(MirHandle h, uint8_t* bad, uint32_t* env_steps, int32_t reset, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  if (!h->diag_on) return set_err(MIR_E_INVALID, "mir_get_bad: diagnostics are switched off (mir_set_diag)");
  DeviceGuard guard(h->device);
  if (bad) {
    hipLaunchKernelGGL(k_get_bad, dim3(nblk(h->B)), dim3(TPB), 0, (hipStream_t)stream, h->diag, bad, h->B);
    HIPCHK(hipGetLastError());
  }
  if (env_steps) {
    HIPCHK(hipMemcpyAsync(env_steps, h->early_stats, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  }
  if (reset) HIPCHK(hipMemsetAsync(h->early_stats, 0, sizeof(uint32_t), (hipStream_t)stream));
  return MIR_OK;
}

int mir_debug_raise_mask_flag(MirHandle h) {
  if (check(h) || !h->pin_host) return set_err(MIR_E_INVALID, "mir_debug_raise_mask_flag: no pinned area");
  *reinterpret_cast<volatile uint32_t*>(h->pin_host + h->pin_flag_off + 16) = 1u;
  return MIR_OK;
}
int mir_get_state_version(MirHandle h) { return h ? (int)(h->state_version & 0x7fffffffull) : MIR_E_INVALID; }

int mir_get_early_mask(MirHandle h) { return check(h) ? MIR_E_INVALID : (h->no_early_mask ? 0 : 1); }

int mir_step_packed(MirHandle h, const float* action, float* rows, int32_t row_stride, void* stream) {
  if (check(h) || !rows) return set_err(MIR_E_INVALID, "mir_step_packed: null argument");
  if (row_stride < h->agent_dim + h->env_dim + 2) return set_err(MIR_E_INVALID, "mir_step_packed: row_stride too small");
  if (h->exact) return set_err(MIR_E_INVALID, "mir_step_packed: not available with exact contacts (the step has to be closed on the host: mir_step_begin / mir_step_end)");
  DeviceGuard guard(h->device);
  Outs o;
  o.action = action; o.rows = rows; o.row_stride = row_stride;
  return launch(h, o, stream);
}

int mir_rollout(MirHandle h, const float* actions, int32_t n_steps, float* rows, int32_t row_stride, void* stream) {
  if (check(h) || !actions || !rows) return set_err(MIR_E_INVALID, "mir_rollout: null argument");
  if (n_steps <= 0) return MIR_OK;
  if (row_stride < h->agent_dim + h->env_dim + 2) return set_err(MIR_E_INVALID, "mir_rollout: row_stride too small");
  if (h->exact) return set_err(MIR_E_INVALID, "mir_rollout: not available with exact contacts (every step has to be closed on the host)");
  DeviceGuard guard(h->device);
  Outs o;
  o.action = actions; o.rows = rows; o.row_stride = row_stride; o.n_steps = n_steps;
  o.act_step = (long)h->B * h->nu; o.rows_step = (long)h->B * row_stride;
  return launch(h, o, stream);
}

int mir_rollout_autoreset(MirHandle h, const float* actions, int32_t n_steps, float* rows, int32_t row_stride, int32_t* episode_len,
                          int32_t max_len, const float* spawn_pool, int32_t pool_len, int32_t* cursor, const float* obj_quat,
                          const float* arm_qpos, void* stream) {
  if (check(h) || !actions || !rows) return set_err(MIR_E_INVALID, "mir_rollout_autoreset: null argument");
  if (!episode_len || !spawn_pool || !cursor || !obj_quat || !arm_qpos || pool_len <= 0) return set_err(MIR_E_INVALID, "mir_rollout_autoreset: null argument");
  if (n_steps <= 0) return MIR_OK;
  if (row_stride < h->agent_dim + h->env_dim + 3) return set_err(MIR_E_INVALID, "mir_rollout_autoreset: row_stride too small (needs the truncated column)");
  if (h->exact) return set_err(MIR_E_INVALID, "mir_rollout_autoreset: not available with exact contacts (every step has to be closed on the host)");
  DeviceGuard guard(h->device);
  Outs o;
  o.action = actions; o.rows = rows; o.row_stride = row_stride; o.n_steps = n_steps;
  o.act_step = (long)h->B * h->nu; o.rows_step = (long)h->B * row_stride;
  o.ar = AutoResetArgs{episode_len, cursor, spawn_pool, obj_quat, arm_qpos, pool_len, max_len};
  return launch(h, o, stream);
}

/* debug aid (not part of the drop-in surface): one step with phase timestamps (shader clock) from block 0; 32 slots */
int mir_debug_profile_step(MirHandle h, unsigned long long* prof16, void* stream) {
  if (check(h) || !prof16) return set_err(MIR_E_INVALID, "mir_debug_profile_step: null argument");
  DeviceGuard guard(h->device);
  Outs o;
  o.prof = prof16;
  return launch(h, o, stream);
}

/* debug aid for -DMIR_PROFILE_SINGLE builds (tools/probes/rot_timeline.py): the next mir_step_begin launch -- whichever kernel the
 * split-step protocol picks for it -- leaves its shader-clock stamps in prof (device memory, 160 x u64, slot 29 = workgroup). */
/* (the same for the next launch of the list instantiation of exact contacts: tools/probes/list_timeline.py) */
extern "C" int mir_debug_profile_next_list_step(MirHandle h, unsigned long long* prof) {
  if (check(h)) return MIR_E_INVALID;
  h->dbg_prof_list = prof;
  return MIR_OK;
}
extern "C" int mir_debug_profile_next_step(MirHandle h, unsigned long long* prof) {
  if (check(h)) return MIR_E_INVALID;
  h->dbg_prof = prof;
  return MIR_OK;
}

/* debug aid (not part of the drop-in surface): the floor under one synchronous env.step() on this machine -- launch an (almost)
 * empty kernel that writes a sequence word into pinned host memory, spin on that word, repeat; *out_us = microseconds per round
 * trip (launch call + dispatch latency + host-visible completion), nothing of the physics in it. */
namespace {
__global__ void k_null_flag(uint32_t* flag, uint32_t seq) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
extern "C" int mir_debug_null_roundtrip(MirHandle h, int32_t iters, void* stream, double* out_us) {
  if (check(h) || !out_us) return set_err(MIR_E_INVALID, "mir_debug_null_roundtrip: bad argument");
  // (experiment, iters < 0: the same round trip with other launch mechanisms -- -1: hipExtLaunchKernelGGL any-order, -2: a
  // one-node graph relaunched with updated kernel parameters; |iters| is then 2000)
  const int variant = iters < 0 ? -iters : 0;
  if (iters < 0) iters = 2000;
  if (iters <= 0) return set_err(MIR_E_INVALID, "mir_debug_null_roundtrip: bad argument");
  if (h->pending) return set_err(MIR_E_INVALID, "mir_debug_null_roundtrip: a step is pending");
  DeviceGuard guard(h->device);
  const size_t off = h->pin_flag_off;
  uint32_t* flag_dev = reinterpret_cast<uint32_t*>(h->pin_dev + off);
  volatile uint32_t* flag = reinterpret_cast<volatile uint32_t*>(h->pin_host + off);
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  hipGraph_t graph = nullptr;
  hipGraphExec_t gexec = nullptr;
  hipGraphNode_t node = nullptr;
  hipKernelNodeParams kp;
  uint32_t gseq = 0;
  void* kargs[2] = {&flag_dev, &gseq};
  if (variant == 2) {
    memset(&kp, 0, sizeof kp);
    kp.func = (void*)k_null_flag; kp.gridDim = dim3(1); kp.blockDim = dim3(64); kp.kernelParams = kargs;
    HIPCHK(hipGraphCreate(&graph, 0));
    HIPCHK(hipGraphAddKernelNode(&node, graph, nullptr, 0, &kp));
    HIPCHK(hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0));
  }
  for (int i = 0; i < iters; i++) {
    const uint32_t seq = ++h->seq;
    if (variant == 1) hipExtLaunchKernelGGL(k_null_flag, dim3(1), dim3(64), 0, (hipStream_t)stream, nullptr, nullptr, 1 /* hipExtAnyOrderLaunch */, flag_dev, seq);
    else if (variant == 2) { gseq = seq; (void)hipGraphExecKernelNodeSetParams(gexec, node, &kp); (void)hipGraphLaunch(gexec, (hipStream_t)stream); }
    else hipLaunchKernelGGL(k_null_flag, dim3(1), dim3(64), 0, (hipStream_t)stream, flag_dev, seq);
    unsigned long polls = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
      __builtin_ia32_pause();
      if (++polls > (1ul << 28)) return set_err(MIR_E_HIP, "mir_debug_null_roundtrip: no completion word");
    }
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (gexec) (void)hipGraphExecDestroy(gexec);
  if (graph) (void)hipGraphDestroy(graph);
  HIPCHK(hipGetLastError());
  *out_us = ((t1.tv_sec - t0.tv_sec) * 1e9 + (t1.tv_nsec - t0.tv_nsec)) * 1e-3 / iters;
  return MIR_OK;
}

/* debug aid (not part of the drop-in surface): the kernel's lane-private convex narrowphase (GJK on the cores, MPR when they
 * overlap) on n pairs given directly; device arrays in (n,22) / out (n,8), layout in mir_step.hip: k_debug_convex */
extern "C" int mir_debug_convex_pairs(const float* in, float* out, int32_t n, int device_id, void* stream) {
  if (!in || !out || n <= 0) return set_err(MIR_E_INVALID, "mir_debug_convex_pairs: bad argument");
  DeviceGuard guard(device_id);
  int rc = mir_launch_debug_convex(in, out, n, (hipStream_t)stream);
  if (rc != 0) return hip_fail((hipError_t)rc, "k_debug_convex");
  return MIR_OK;
}

/* debug aid (not part of the drop-in surface): a copy with the step kernel's own access shape -- 64-thread workgroups, 4 bytes per
 * lane, a wave touching 256 contiguous bytes -- over a byte count the caller knows: the calibration of rocprofv3's FETCH_SIZE /
 * WRITE_SIZE for this access width (MI355X_MICROARCH.md, HBM: only 16 B/lane streams are calibrated there). */
namespace {
__global__ __launch_bounds__(64) void k_copy_rows(const float* __restrict__ src, float* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * 64 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
}  // namespace
extern "C" int mir_debug_copy_rows(const float* src, float* dst, int64_t n_floats, int device_id, void* stream) {
  if (!src || !dst || n_floats <= 0) return set_err(MIR_E_INVALID, "mir_debug_copy_rows: bad argument");
  DeviceGuard guard(device_id);
  hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)((n_floats + 63) / 64)), dim3(64), 0, (hipStream_t)stream, src, dst, (long)n_floats);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "k_copy_rows");
  return MIR_OK;
}

/* debug aid (not part of the drop-in surface): overwrite the LDS of every CU with signalling-NaN bit patterns.  LDS keeps
 * what the previous kernel left in it, so a step kernel that reads a slot before writing it usually finds plausible stale
 * values there; the GPU tests call this first, which turns such a read into a NaN in the results. */
namespace {
__global__ void k_poison_lds(int words) {
  extern __shared__ unsigned int lds_words[];
  for (int i = threadIdx.x; i < words; i += blockDim.x) lds_words[i] = 0x7fa00000u + (unsigned)i;
  __syncthreads();
  if (lds_words[(threadIdx.x * 97) % words] == 1u) lds_words[0] = 2u;  // (keeps the stores observable)
}
}  // namespace
extern "C" int mir_debug_poison_lds(int device_id, void* stream) {
  DeviceGuard guard(device_id);
  const int bytes = 160 * 1024;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_poison_lds), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return hip_fail(e, "hipFuncSetAttribute");
  // one 160 KB workgroup owns a whole CU; several waves of workgroups so that every CU is visited
  hipLaunchKernelGGL(k_poison_lds, dim3(4096), dim3(256), bytes, (hipStream_t)stream, bytes / 4);
  e = hipGetLastError();
  if (e != hipSuccess) return hip_fail(e, "k_poison_lds");
  return MIR_OK;
}

int mir_p2p_enable(int32_t device, int32_t peer) {
  // (dry-run switch of the multi-GPU tests: MIR_P2P_FORCE_FAIL = "all", or the RANK -- torch.distributed.run's variable -- whose calls
  //  are to fail; the caller, sharding.CopyPathGather, then agrees with the other ranks on the collective instead)
  if (const char* f = getenv("MIR_P2P_FORCE_FAIL")) {
    const char* r = getenv("RANK");
    if (!strcmp(f, "all") || (r && !strcmp(f, r))) return set_err(MIR_E_HIP, "mir_p2p_enable: forced to fail (MIR_P2P_FORCE_FAIL)");
  }
  if (device == peer) return MIR_OK;
  DeviceGuard guard(device);
  int can = 0;
  HIPCHK(hipDeviceCanAccessPeer(&can, device, peer));
  if (!can) return set_err(MIR_E_HIP, "mir_p2p_enable: the device cannot access the peer's memory");
  hipError_t e = hipDeviceEnablePeerAccess(peer, 0);
  if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return hip_fail(e, "hipDeviceEnablePeerAccess");
  (void)hipGetLastError();
  return MIR_OK;
}

int mir_p2p_push(void* const* dst, int32_t n, const void* src, uint64_t nbytes, void* const* flag_dst, const void* flag_src, void* stream) {
  if (!dst || !src || n <= 0 || (flag_dst && !flag_src)) return set_err(MIR_E_INVALID, "mir_p2p_push: bad argument");
  for (int i = 0; i < n; i++) HIPCHK(hipMemcpyAsync(dst[i], src, (size_t)nbytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  if (flag_dst)
    for (int i = 0; i < n; i++) HIPCHK(hipMemcpyAsync(flag_dst[i], flag_src, 4, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return MIR_OK;
}

/* the same push with ONE STREAM PER DESTINATION: destination i's block and, behind it, its sequence word travel on streams[i], so
 * the copies to different peers run side by side on different copy engines / xGMI links (on one stream they complete in order: seven
 * 11 MB copies in series fell behind a 0.7 ms chunk period).  A word never overtakes its own block. */
int mir_p2p_push_streams(void* const* dst, int32_t n, const void* src, uint64_t nbytes, void* const* flag_dst, const void* flag_src, void* const* streams) {
  if (!dst || !src || !streams || n <= 0 || (flag_dst && !flag_src)) return set_err(MIR_E_INVALID, "mir_p2p_push_streams: bad argument");
  for (int i = 0; i < n; i++) {
    if (nbytes) HIPCHK(hipMemcpyAsync(dst[i], src, (size_t)nbytes, hipMemcpyDeviceToDevice, (hipStream_t)streams[i]));
    if (flag_dst) HIPCHK(hipMemcpyAsync(flag_dst[i], flag_src, 4, hipMemcpyDeviceToDevice, (hipStream_t)streams[i]));
  }
  return MIR_OK;
}

int mir_get_obs(MirHandle h, float* agent_pos, float* env_state, float* reward, uint8_t* terminated, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  DeviceGuard guard(h->device);
  Outs o;
  o.mode = 2; o.diag = false;
  o.agent_pos = agent_pos; o.env_state = env_state; o.reward = reward; o.terminated = terminated;
  return launch(h, o, stream);
}

int mir_get_state(MirHandle h, float* qpos, float* qvel, float* target, float* warmstart, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_copy_state, dim3(nblk((long)h->B * PW)), dim3(TPB), 0, (hipStream_t)stream, h->dpt, h->qpos, h->qvel, h->target,
                     h->qacc_ws, qpos, qvel, target, warmstart, h->fkvalid, h->B, 0);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_set_state(MirHandle h, const float* qpos, const float* qvel, const float* target, const float* warmstart, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  h->pre_valid = 0;
  h->poses_current = 0;
  h->state_version++;
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_copy_state, dim3(nblk((long)h->B * PW)), dim3(TPB), 0, (hipStream_t)stream, h->dpt, h->qpos, h->qvel, h->target,
                     h->qacc_ws, (float*)qpos, (float*)qvel, (float*)target, (float*)warmstart, h->fkvalid, h->B, 1);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_get_links(MirHandle h, float* pos, float* quat, void* stream) {
  if (check(h) || !pos || !quat) return set_err(MIR_E_INVALID, "mir_get_links: null argument");
  DeviceGuard guard(h->device);
  Outs o;
  o.mode = 2; o.diag = false;
  o.out_xpos = pos; o.out_xquat = quat;
  return launch(h, o, stream);
}

int mir_set_diag(MirHandle h, int32_t on) {
  if (check(h)) return MIR_E_INVALID;
  h->diag_on = on ? 1 : 0;
  return MIR_OK;
}

int mir_get_diag(MirHandle h, int32_t* ncon, int32_t* nefc, int32_t* niter, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  if (!h->diag_on) return set_err(MIR_E_INVALID, "mir_get_diag: per-env diagnostics are switched off (mir_set_diag)");
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_get_diag, dim3(nblk(h->B)), dim3(TPB), 0, (hipStream_t)stream, h->diag, ncon, nefc, niter, h->B);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_forward(MirHandle h, float* M, float* qfrc_bias, float* qacc_smooth, float* qacc, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  DeviceGuard guard(h->device);
  Outs o;
  o.mode = 1;
  o.out_M = M; o.out_bias = qfrc_bias; o.out_qas = qacc_smooth; o.out_qacc = qacc;
  return launch(h, o, stream);
}

}  // extern "C"

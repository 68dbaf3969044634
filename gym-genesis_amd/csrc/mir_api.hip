// mir_api.hip — C ABI of libmirigid.so (declared in include/mirigid.h) and the small
// state-plumbing kernels around the fused step kernel (mir_step.hip).
//
// Everything is enqueued on the caller's HIP stream and nothing synchronises the device.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "mir_model.h"
#include "mir_scene.h"
#include "mir_step.h"

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

// ---- plumbing kernels (one thread per (env, column); rows are contiguous -> coalesced) -----------
__global__ void k_fill_rows(float* dst, const float* row, int stride, int B) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (long)B * stride) dst[i] = row[i % stride];
}

__global__ void k_reset(const DevModel* __restrict__ m, float* qpos, float* qvel, float* target, float* ws,
                        const float* obj_pos, const float* obj_quat, const float* arm_qpos, const uint8_t* mask, int32_t* fkvalid, int B) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int e = (int)(i / MIR_G), c = (int)(i % MIR_G);
  if (e >= B) return;
  if (mask && !mask[e]) return;
  const int qst = m->qstride;
  qvel[(long)e * MIR_G + c] = 0.0f;
  ws[(long)e * MIR_G + c] = 0.0f;
  if (c == 0) fkvalid[e] = 0;  // cached link poses no longer match qpos
  if (c < m->nv) {
    int ai = m->d_armidx[c];
    if (ai >= 0 && arm_qpos) {
      float v = arm_qpos[(long)e * m->n_arm_q + ai];
      qpos[(long)e * qst + m->d_qadr[c]] = v;
      target[(long)e * MIR_G + c] = v;
    }
  }
  if (m->obj_qadr >= 0) {
    if (c < 3 && obj_pos) qpos[(long)e * qst + m->obj_qadr + c] = obj_pos[(long)e * 3 + c];
    if (c >= 3 && c < 7 && obj_quat) qpos[(long)e * qst + m->obj_qadr + c] = obj_quat[(long)e * 4 + (c - 3)];
  }
}

// Episode bookkeeping + re-spawn of finished envs, all on the device (no host round trip).
// 16 consecutive threads serve one env (same wave): every thread reads the env's counters before lane c==0 rewrites them.
__global__ void k_autoreset(const DevModel* __restrict__ m, float* qpos, float* qvel, float* target, float* ws, const uint8_t* terminated,
                            int32_t* episode_len, int max_len, const float* spawn_pool, int pool_len, int32_t* cursor,
                            const float* obj_quat, const float* arm_qpos, uint8_t* truncated_out, uint8_t* done_out, int32_t* fkvalid, int B) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int e = (int)(i / MIR_G), c = (int)(i % MIR_G);
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
  const int qst = m->qstride;
  qvel[(long)e * MIR_G + c] = 0.0f;
  ws[(long)e * MIR_G + c] = 0.0f;
  if (c == 0) fkvalid[e] = 0;
  if (c < m->nv) {
    int ai = m->d_armidx[c];
    if (ai >= 0) {
      float v = arm_qpos[(long)e * m->n_arm_q + ai];
      qpos[(long)e * qst + m->d_qadr[c]] = v;
      target[(long)e * MIR_G + c] = v;
    }
  }
  if (m->obj_qadr >= 0) {
    const float* sp = spawn_pool + ((long)(cur % pool_len) * B + e) * 3;
    if (c < 3) qpos[(long)e * qst + m->obj_qadr + c] = sp[c];
    if (c >= 3 && c < 7) qpos[(long)e * qst + m->obj_qadr + c] = obj_quat[(long)e * 4 + (c - 3)];
  }
}

__global__ void k_set_targets(const DevModel* __restrict__ m, float* target, const float* tgt, int B) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int e = (int)(i / MIR_G), c = (int)(i % MIR_G);
  if (e >= B || c >= m->nv) return;
  int u = m->d_uadr[c];
  if (u >= 0) target[(long)e * MIR_G + c] = tgt[(long)e * m->nu + u];
}

// dir 0: internal -> external (get); 1: external -> internal (set)
__global__ void k_copy_state(const DevModel* __restrict__ m, float* iq, float* iv, float* it, float* iw, float* eq, float* ev,
                             float* et, float* ew, int32_t* fkvalid, int B, int dir) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int e = (int)(i / 32), c = (int)(i % 32);
  if (e >= B) return;
  const int qst = m->qstride, nq = m->nq, nv = m->nv, nu = m->nu;
  if (dir && eq && c == 0) fkvalid[e] = 0;
  if (eq && c < nq) {
    if (dir) iq[(long)e * qst + c] = eq[(long)e * nq + c];
    else eq[(long)e * nq + c] = iq[(long)e * qst + c];
  }
  if (c < nv) {
    if (ev) {
      if (dir) iv[(long)e * MIR_G + c] = ev[(long)e * nv + c];
      else ev[(long)e * nv + c] = iv[(long)e * MIR_G + c];
    }
    if (ew) {
      if (dir) iw[(long)e * MIR_G + c] = ew[(long)e * nv + c];
      else ew[(long)e * nv + c] = iw[(long)e * MIR_G + c];
    }
    int u = m->d_uadr[c];
    if (et && u >= 0) {
      if (dir) it[(long)e * MIR_G + c] = et[(long)e * nu + u];
      else et[(long)e * nu + u] = it[(long)e * MIR_G + c];
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

StepArgs base_args(MirScene* h) {
  StepArgs a;
  memset(&a, 0, sizeof a);
  a.model = h->dm;
  a.qpos = h->qpos; a.qvel = h->qvel; a.target = h->target; a.qacc_ws = h->qacc_ws;
  a.poses = h->poses; a.fkvalid = h->fkvalid;
  a.diag = h->diag;
  a.B = h->B;
  a.n_steps = 1;
  return a;
}

int launch(MirScene* h, const StepArgs& a, void* stream) {
  int rc = mir_launch_step(&a, h->hm.max_contacts, (hipStream_t)stream);
  if (rc != 0) return hip_fail((hipError_t)rc, "mir_step_kernel launch");
  return MIR_OK;
}

int check(MirHandle h) {
  if (!h) return set_err(MIR_E_INVALID, "null MirHandle");
  return MIR_OK;
}

}  // namespace

int mir_set_error(int code, const char* msg) { return set_err(code, "%s", msg); }

int mir_refresh_poses(MirScene* h, void* stream) {
  StepArgs a = base_args(h);
  a.mode = 2; a.diag = nullptr;
  return launch(h, a, stream);
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
  int rc = mir_compile_model(spec, &h->hm, &h->hc, err);
  if (rc != MIR_OK) {
    delete h;
    return set_err(rc, "mir_create: %s", err);
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) {
    delete h;
    return set_err(MIR_E_NODEVICE, "mir_create: no usable HIP device (libmirigid has no CPU path)");
  }
  h->device = device_id;
  h->B = num_envs;
  DeviceGuard guard(device_id);
  const size_t B = (size_t)num_envs;
  const int qst = h->hm.qstride;
  hipError_t e;
  if ((e = hipMalloc((void**)&h->dm, sizeof(DevModel))) != hipSuccess || (e = hipMalloc((void**)&h->qpos, B * qst * sizeof(float))) != hipSuccess ||
      (e = hipMalloc((void**)&h->qvel, B * MIR_G * sizeof(float))) != hipSuccess || (e = hipMalloc((void**)&h->target, B * MIR_G * sizeof(float))) != hipSuccess ||
      (e = hipMalloc((void**)&h->qacc_ws, B * MIR_G * sizeof(float))) != hipSuccess || (e = hipMalloc((void**)&h->diag, B * 4 * sizeof(int32_t))) != hipSuccess ||
      (e = hipMalloc((void**)&h->poses, B * 2 * MIR_G * 4 * sizeof(float))) != hipSuccess || (e = hipMalloc((void**)&h->fkvalid, B * sizeof(int32_t))) != hipSuccess) {
    mir_destroy(h);
    return hip_fail(e, "hipMalloc");
  }
  HIPCHK(hipMemcpy(h->dm, &h->hm, sizeof(DevModel), hipMemcpyHostToDevice));
  // initial state: qpos0 (free bodies at their spec pose, scalar joints at 0), everything else 0
  float row[32] = {0};
  for (int b = 1; b < h->hm.nbody; b++)
    if (h->hm.b_jtype[b] == MIR_JNT_FREE) {
      for (int k = 0; k < 3; k++) row[h->hm.b_qadr[b] + k] = h->hm.b_pos[b][k];
      for (int k = 0; k < 4; k++) row[h->hm.b_qadr[b] + 3 + k] = h->hm.b_quat[b][k];
    }
  float* drow = nullptr;
  HIPCHK(hipMalloc((void**)&drow, sizeof row));
  HIPCHK(hipMemcpy(drow, row, sizeof row, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_fill_rows, dim3(nblk((long)B * qst)), dim3(TPB), 0, 0, h->qpos, drow, qst, num_envs);
  HIPCHK(hipMemset(h->qvel, 0, B * MIR_G * sizeof(float)));
  HIPCHK(hipMemset(h->target, 0, B * MIR_G * sizeof(float)));
  HIPCHK(hipMemset(h->qacc_ws, 0, B * MIR_G * sizeof(float)));
  HIPCHK(hipMemset(h->diag, 0, B * 4 * sizeof(int32_t)));
  HIPCHK(hipMemset(h->poses, 0, B * 2 * MIR_G * 4 * sizeof(float)));
  HIPCHK(hipMemset(h->fkvalid, 0, B * sizeof(int32_t)));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipFree(drow));
  *out = h;
  return MIR_OK;
}

int mir_destroy(MirHandle h) {
  if (!h) return MIR_OK;
  DeviceGuard guard(h->device);
  if (h->dm) (void)hipFree(h->dm);
  if (h->qpos) (void)hipFree(h->qpos);
  if (h->qvel) (void)hipFree(h->qvel);
  if (h->target) (void)hipFree(h->target);
  if (h->qacc_ws) (void)hipFree(h->qacc_ws);
  if (h->diag) (void)hipFree(h->diag);
  if (h->poses) (void)hipFree(h->poses);
  if (h->fkvalid) (void)hipFree(h->fkvalid);
  if (h->prims) (void)hipFree(h->prims);
  delete h;
  return MIR_OK;
}

int mir_get_dims(MirHandle h, MirDims* out) {
  if (check(h) || !out) return set_err(MIR_E_INVALID, "mir_get_dims: null argument");
  out->num_envs = h->B; out->nbody = h->hm.nbody; out->nq = h->hm.nq; out->nv = h->hm.nv;
  out->ngeom = h->hm.ngeom; out->npair = h->hm.npair; out->agent_dim = 7 + h->hm.n_grip; out->env_dim = 11;
  return MIR_OK;
}

int mir_get_model_consts(MirHandle h, double* dof_invweight0, double* body_invweight0, double* meaninertia) {
  if (check(h)) return MIR_E_INVALID;
  if (dof_invweight0) memcpy(dof_invweight0, h->hc.dof_invweight0, sizeof(double) * h->hm.nv);
  if (body_invweight0) memcpy(body_invweight0, h->hc.body_invweight0, sizeof(double) * h->hm.nbody);
  if (meaninertia) *meaninertia = h->hc.meaninertia;
  return MIR_OK;
}

int mir_reset(MirHandle h, const float* obj_pos, const float* obj_quat, const float* arm_qpos, const uint8_t* env_mask, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_reset, dim3(nblk((long)h->B * MIR_G)), dim3(TPB), 0, (hipStream_t)stream, h->dm, h->qpos, h->qvel, h->target,
                     h->qacc_ws, obj_pos, obj_quat, arm_qpos, env_mask, h->fkvalid, h->B);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_autoreset(MirHandle h, const uint8_t* terminated, int32_t* episode_len, int32_t max_len, const float* spawn_pool, int32_t pool_len,
                  int32_t* cursor, const float* obj_quat, const float* arm_qpos, uint8_t* truncated_out, uint8_t* done_out, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  if (!episode_len || !spawn_pool || !cursor || !obj_quat || !arm_qpos || pool_len <= 0) return set_err(MIR_E_INVALID, "mir_autoreset: null argument");
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_autoreset, dim3(nblk((long)h->B * MIR_G)), dim3(TPB), 0, (hipStream_t)stream, h->dm, h->qpos, h->qvel, h->target,
                     h->qacc_ws, terminated, episode_len, max_len, spawn_pool, pool_len, cursor, obj_quat, arm_qpos, truncated_out, done_out,
                     h->fkvalid, h->B);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_set_pd_targets(MirHandle h, const float* tgt, void* stream) {
  if (check(h) || !tgt) return set_err(MIR_E_INVALID, "mir_set_pd_targets: null argument");
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_set_targets, dim3(nblk((long)h->B * MIR_G)), dim3(TPB), 0, (hipStream_t)stream, h->dm, h->target, tgt, h->B);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_step(MirHandle h, int32_t n_steps, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  if (n_steps <= 0) return MIR_OK;
  DeviceGuard guard(h->device);
  StepArgs a = base_args(h);
  a.n_steps = n_steps;
  return launch(h, a, stream);
}

int mir_step_fused(MirHandle h, const float* action, float* agent_pos, float* env_state, float* reward, uint8_t* terminated, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  DeviceGuard guard(h->device);
  StepArgs a = base_args(h);
  a.action = action; a.agent_pos = agent_pos; a.env_state = env_state; a.reward = reward; a.terminated = terminated;
  return launch(h, a, stream);
}

int mir_step_packed(MirHandle h, const float* action, float* rows, int32_t row_stride, void* stream) {
  if (check(h) || !rows) return set_err(MIR_E_INVALID, "mir_step_packed: null argument");
  if (row_stride < 7 + h->hm.n_grip + 13) return set_err(MIR_E_INVALID, "mir_step_packed: row_stride too small");
  DeviceGuard guard(h->device);
  StepArgs a = base_args(h);
  a.action = action; a.rows = rows; a.row_stride = row_stride;
  return launch(h, a, stream);
}

/* debug aid (not part of the drop-in surface): one step with phase timestamps from block 0 */
int mir_debug_profile_step(MirHandle h, unsigned long long* prof16, void* stream) {
  if (check(h) || !prof16) return set_err(MIR_E_INVALID, "mir_debug_profile_step: null argument");
  DeviceGuard guard(h->device);
  StepArgs a = base_args(h);
  a.prof = prof16;
  return launch(h, a, stream);
}

int mir_get_obs(MirHandle h, float* agent_pos, float* env_state, float* reward, uint8_t* terminated, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  DeviceGuard guard(h->device);
  StepArgs a = base_args(h);
  a.mode = 2; a.diag = nullptr;
  a.agent_pos = agent_pos; a.env_state = env_state; a.reward = reward; a.terminated = terminated;
  return launch(h, a, stream);
}

int mir_get_state(MirHandle h, float* qpos, float* qvel, float* target, float* warmstart, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_copy_state, dim3(nblk((long)h->B * 32)), dim3(TPB), 0, (hipStream_t)stream, h->dm, h->qpos, h->qvel, h->target,
                     h->qacc_ws, qpos, qvel, target, warmstart, h->fkvalid, h->B, 0);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_set_state(MirHandle h, const float* qpos, const float* qvel, const float* target, const float* warmstart, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_copy_state, dim3(nblk((long)h->B * 32)), dim3(TPB), 0, (hipStream_t)stream, h->dm, h->qpos, h->qvel, h->target,
                     h->qacc_ws, (float*)qpos, (float*)qvel, (float*)target, (float*)warmstart, h->fkvalid, h->B, 1);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_get_links(MirHandle h, float* pos, float* quat, void* stream) {
  if (check(h) || !pos || !quat) return set_err(MIR_E_INVALID, "mir_get_links: null argument");
  DeviceGuard guard(h->device);
  StepArgs a = base_args(h);
  a.mode = 2; a.diag = nullptr;
  a.out_xpos = pos; a.out_xquat = quat;
  return launch(h, a, stream);
}

int mir_get_diag(MirHandle h, int32_t* ncon, int32_t* nefc, int32_t* niter, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  DeviceGuard guard(h->device);
  hipLaunchKernelGGL(k_get_diag, dim3(nblk(h->B)), dim3(TPB), 0, (hipStream_t)stream, h->diag, ncon, nefc, niter, h->B);
  HIPCHK(hipGetLastError());
  return MIR_OK;
}

int mir_forward(MirHandle h, float* M, float* qfrc_bias, float* qacc_smooth, float* qacc, void* stream) {
  if (check(h)) return MIR_E_INVALID;
  DeviceGuard guard(h->device);
  StepArgs a = base_args(h);
  a.mode = 1;
  a.out_M = M; a.out_bias = qfrc_bias; a.out_qas = qacc_smooth; a.out_qacc = qacc;
  return launch(h, a, stream);
}

}  // extern "C"

// mir_ik.hip — batched damped-least-squares inverse kinematics.
//
// What it replaces: robot.inverse_kinematics(link, pos, quat, ...) of the reference's expert policies
// (/root/reference/examples/franka/pick_cube_state.py:46-51, stack_cube_state.py:78-83), called once per env.step()
// in those loops, i.e. on the callers' side of the hot path.  The algorithm is defined in include/mirigid.h.
//
// Mapping: like the pick kernel, 16 lanes per env (4 envs per wave64, one DPP row each); lane j owns element j of the
// kinematic chain world -> link (<= 16 bodies): its local joint transform, its world pose (prefix of the chain by a log-step DPP
// scan: no depth-serial barriers, no LDS at all since round 6), its Jacobian column.  J J^T + lambda^2 I (6 x 6, symmetric: 21 DPP row
// reductions) ends up in every lane's registers and each lane solves it redundantly (Cholesky, fully unrolled), so the
// update dq_j = J_j . y needs no further communication.  The chain description travels in the kernel arguments
// (built on the host per call: the link is a run-time argument).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstring>

#include "mir_model.h"
#include "mir_scene.h"

#define G 16
#include "mir_dev.h"

namespace {

struct IkChain {
  int n;                       // bodies on the chain (root first)
  int jtype[G], qcol[G];       // joint type; column of the joint in the (B, n_arm) arrays, -1 for fixed links
  float pos[G][3], quat[G][4], axis[G][3], lo[G], hi[G];
  int limited[G];
};

struct IkArgs {
  IkChain ch;
  const float* target_pos;   // (B,3)
  const float* target_quat;  // (B,4) or null
  const float* init_qpos;    // (B,n_arm) or null
  const float* scene_qpos;   // scene state row (B, qst) used when init_qpos is null
  int qst, n_arm;
  int arm_qadr[MIR_MAX_DOF]; // qpos address of scalar joint k in the scene row
  float* qpos_out;           // (B,n_arm)
  float* err_out;            // (B,2) or null
  int32_t* iters_out;        // (rows) or null: iterations the row took (debug: mir_debug_ik_iters)
  // rows (mir_inverse_kinematics_rows): row k of the outputs belongs to env env_idx[k] (null: env k); the inputs are addressed by row,
  // or by env (the *_by_env flags), the quaternion possibly ONE for all rows; init_qpos holds init_ncols columns from init_col0 on
  const long long* env_idx;
  int n_rows, pos_by_env, quat_by_env, quat_one, init_by_env, init_col0, init_ncols;
  int B, max_iters, respect_limits;
  float inv_pos_tol, inv_rot_tol;
  float damping2, pos_tol, rot_tol, max_step;
};

__device__ __forceinline__ Q4 qconj(Q4 q) { return {q.w, -q.x, -q.y, -q.z}; }

__global__ __launch_bounds__(64) void mir_ik_kernel(IkArgs a) {
  const int tid = threadIdx.x, lane = tid & 15, grp = tid >> 4;
  const int row_raw = blockIdx.x * 4 + grp;
  const bool valid = row_raw < a.n_rows;
  const int row = valid ? row_raw : a.n_rows - 1;
  int env = a.env_idx ? (int)a.env_idx[row] : row;
  env = env < 0 ? 0 : (env >= a.B ? a.B - 1 : env);  // (an index outside the batch is clamped, not followed: the caller's side checks it)
  const int prow = a.pos_by_env ? env : row, qrow = a.quat_one ? 0 : (a.quat_by_env ? env : row), irow = a.init_by_env ? env : row;
  const int n = a.ch.n;
  const int eef4 = ((tid & ~15) + n - 1) << 2;  // (lane_gather address of the chain's last element in this env's row)
  const bool onchain = lane < n;
  const int jt = onchain ? a.ch.jtype[lane] : MIR_JNT_FIXED, qc = onchain ? a.ch.qcol[lane] : -1;
  const V3 bpos = ld3(a.ch.pos[lane]), baxis = ld3(a.ch.axis[lane]);
  const Q4 bquat = ld4(a.ch.quat[lane]);
  const bool moving = onchain && qc >= 0 && (jt == MIR_JNT_REVOLUTE || jt == MIR_JNT_PRISMATIC);
  const float lo = a.ch.lo[lane], hi = a.ch.hi[lane];
  const bool lim = moving && a.ch.limited[lane] && a.respect_limits;
  // seed: every scalar joint (the result keeps the seed outside the chain)
  auto seed = [&](int k) -> float {
    const bool from_init = a.init_qpos && k >= a.init_col0 && k < a.init_col0 + a.init_ncols;
    return from_init ? a.init_qpos[(size_t)irow * a.init_ncols + (k - a.init_col0)] : a.scene_qpos[(size_t)env * a.qst + a.arm_qadr[k]];
  };
  for (int k = lane; k < a.n_arm; k += G) {
    const float v = seed(k);
    if (valid) a.qpos_out[(size_t)row * a.n_arm + k] = v;
  }
  float q = 0.0f;
  if (moving) q = seed(qc);
  const V3 tp = ld3(a.target_pos + (size_t)prow * 3);
  const bool userot = a.target_quat != nullptr;
  const Q4 tq = userot ? qnormalize(ld4(a.target_quat + (size_t)qrow * 4)) : Q4{1, 0, 0, 0};
  bool done = false;
  int stall = 0, my_iters = 0;
  // the ACCEPTED iterate (Levenberg - Marquardt acceptance, include/mirigid.h): this lane's joint angle, Jacobian column; the env's
  // task-space error and scaled error (identical in all of its lanes: they are computed from the same gathered values)
  float q_acc = q, lam2 = a.damping2, m_acc = 0.0f, epn = 0.0f, ern = 0.0f;
  const float lam2_min = a.damping2 * (1.0f / 256.0f), lam2_max = a.damping2 * 64.0f;
  float J[6] = {0, 0, 0, 0, 0, 0}, e[6] = {0, 0, 0, 0, 0, 0};
  for (int it = 0; it <= a.max_iters; it++) {
    // ---- local transform of my chain element (identity off the chain) at the CANDIDATE q ...
    V3 P = v3(0, 0, 0);
    Q4 Qx = Q4{1, 0, 0, 0};
    if (onchain) {
      Qx = bquat;
      P = bpos;
      if (jt == MIR_JNT_REVOLUTE) {
        float sn, cs;
        sincos_pi2(0.5f * q, &sn, &cs);
        Qx = qmul(bquat, Q4{cs, baxis.x * sn, baxis.y * sn, baxis.z * sn});
      } else if (jt == MIR_JNT_PRISMATIC) {
        P = bpos + qrot(bquat, q * baxis);
      }
    }
    // ... then my own prefix of the chain by a log-step scan over the DPP row (composition (P,Q) o (p,q) = (P + Q p, Q q)
    // is associative): 4 shifted exchanges instead of a walk of up to 15 links, no LDS round trips
    {
#define IK_SCAN_STEP(D)                                                                                          \
      {                                                                                                          \
        const V3 pp = v3(row_shr<D>(P.x), row_shr<D>(P.y), row_shr<D>(P.z));                                     \
        const Q4 pq = Q4{row_shr<D>(Qx.w), row_shr<D>(Qx.x), row_shr<D>(Qx.y), row_shr<D>(Qx.z)};                \
        if (lane >= D) {                                                                                         \
          P = pp + qrot(pq, P);                                                                                  \
          Qx = qmul(pq, Qx);                                                                                     \
        }                                                                                                        \
      }
      IK_SCAN_STEP(1)
      if (n > 2) IK_SCAN_STEP(2)
      if (n > 4) IK_SCAN_STEP(4)
      if (n > 8) IK_SCAN_STEP(8)  // (the Panda's chain to the hand is eight elements once its fixed base link is folded into joint 1: host side)
#undef IK_SCAN_STEP
    }
    // ---- task-space error of the candidate (every lane, redundantly): the pose of the chain's last element comes over the crossbar
    // (seven lane gathers, one trip; round 5 went through LDS: a store, a fence and a load per iteration of a chain that is all latency)
    const V3 pe = v3(lane_gather(eef4, P.x), lane_gather(eef4, P.y), lane_gather(eef4, P.z));
    const Q4 qe = Q4{lane_gather(eef4, Qx.w), lane_gather(eef4, Qx.x), lane_gather(eef4, Qx.y), lane_gather(eef4, Qx.z)};
    const V3 ep = tp - pe;
    V3 er = v3(0, 0, 0);
    if (userot) {
      Q4 d = qmul(tq, qconj(qe));  // rotation taking the current frame to the target, world axes
      if (d.w < 0.0f) d = Q4{-d.w, -d.x, -d.y, -d.z};
      const float sn = sqrtf(d.x * d.x + d.y * d.y + d.z * d.z);
      const float ang = 2.0f * atan2f(sn, d.w);
      const float k = sn > 1e-9f ? ang / sn : 2.0f;
      er = v3(k * d.x, k * d.y, k * d.z);
    }
    const float epn_c = sqrtf(dot(ep, ep)), ern_c = sqrtf(dot(er, er));
    const float metric = epn_c * a.inv_pos_tol + ern_c * a.inv_rot_tol;
    if (!done) {
      if (it == 0 || metric < m_acc) {
        // accepted: the damping relaxes; stagnation = an accepted step that gained less than 1 % (a target beyond the joint limits
        // or the reach: without the rule the few unreachable targets of a batch set the time of the whole launch)
        if (it > 0) {
          stall = metric > 0.99f * m_acc ? stall + 1 : 0;
          lam2 = fmaxf(lam2 * 0.25f, lam2_min);
        }
        m_acc = metric; epn = epn_c; ern = ern_c;
        e[0] = ep.x; e[1] = ep.y; e[2] = ep.z; e[3] = er.x; e[4] = er.y; e[5] = er.z;
        q_acc = q;
        // my Jacobian column at the accepted iterate (joint frame = my world pose)
        V3 jv = v3(0, 0, 0), jw = v3(0, 0, 0);
        if (moving) {
          const V3 axw = qrot(Qx, baxis);
          if (jt == MIR_JNT_REVOLUTE) { jw = axw; jv = cross(axw, pe - P); }
          else jv = axw;
        }
        J[0] = jv.x; J[1] = jv.y; J[2] = jv.z; J[3] = userot ? jw.x : 0.0f; J[4] = userot ? jw.y : 0.0f; J[5] = userot ? jw.z : 0.0f;
      } else {
        // rejected (the scaled error did not fall): back to the accepted iterate with eight times the damping; a stalled iteration
        stall++;
        lam2 = fminf(lam2 * 8.0f, lam2_max);
      }
      if ((epn < a.pos_tol && ern < a.rot_tol) || stall >= 3) done = true;
    }
    if (it == a.max_iters) break;
    if (!__any(!done)) break;
    // ---- A = J J^T + lambda^2 I in every lane (J, e: the accepted iterate's)
    float A[6][6];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c <= r; c++) {
        const float v = gsum(J[r] * J[c]) + (r == c ? lam2 : 0.0f);
        A[r][c] = v;
        A[c][r] = v;
      }
    // ---- y = A^-1 e (Cholesky, SPD by the damping)
    float L[6][6], y[6], il[6];  // il = 1 / L[c][c]: one reciprocal per column instead of a division per use
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
      for (int c = 0; c <= r; c++) {
        float sacc = A[r][c];
#pragma unroll
        for (int k = 0; k < c; k++) sacc -= L[r][k] * L[c][k];
        if (r == c) {
          const float dd = sqrtf(fmaxf(sacc, 1e-30f));
          float ir = __builtin_amdgcn_rcpf(dd);
          L[r][c] = dd;
          il[r] = ir;
        } else {
          L[r][c] = sacc * il[c];
        }
      }
#pragma unroll
    for (int r = 0; r < 6; r++) {
      float sacc = e[r];
#pragma unroll
      for (int k = 0; k < r; k++) sacc -= L[r][k] * y[k];
      y[r] = sacc * il[r];
    }
#pragma unroll
    for (int r = 5; r >= 0; r--) {
      float sacc = y[r];
#pragma unroll
      for (int k = r + 1; k < 6; k++) sacc -= L[k][r] * y[k];
      y[r] = sacc * il[r];
    }
    float dq = 0.0f;
#pragma unroll
    for (int r = 0; r < 6; r++) dq += J[r] * y[r];
    const float big = gmaxf(fabsf(dq));
    const float sc = big > a.max_step ? a.max_step / big : 1.0f;
    if (!done) my_iters = it + 1;
    q = q_acc;
    if (moving && !done) {
      q = q_acc + sc * dq;
      if (lim) q = fminf(fmaxf(q, lo), hi);
    }
  }
  q = q_acc;
  if (valid && moving) a.qpos_out[(size_t)row * a.n_arm + qc] = q;
  if (valid && a.iters_out && lane == 0) a.iters_out[row] = my_iters;
  if (valid && a.err_out && lane == 0) {
    a.err_out[(size_t)row * 2] = epn;
    a.err_out[(size_t)row * 2 + 1] = ern;
  }
}

}  // namespace

extern "C" int mir_inverse_kinematics_rows(MirHandle h, int32_t link_body, const MirIkRows* rows, const float* target_pos, const float* target_quat,
                                           const float* init_qpos, const MirIkOptions* opt, float* qpos_out, float* err_out, void* stream) {
  if (!h || !target_pos || !qpos_out) return mir_set_error(MIR_E_INVALID, "mir_inverse_kinematics: null argument");
  if (rows && (rows->n_rows < 0 || (rows->flags & ~(uint32_t)(MIR_IK_POS_BY_ENV | MIR_IK_QUAT_BY_ENV | MIR_IK_QUAT_ONE | MIR_IK_INIT_BY_ENV))))
    return mir_set_error(MIR_E_INVALID, "mir_inverse_kinematics_rows: bad row description");
  if (rows && rows->n_rows == 0) return MIR_OK;
  if (link_body <= 0 || link_body >= h->nbody) return mir_set_error(MIR_E_INVALID, "mir_inverse_kinematics: link out of range");
  IkArgs a;
  memset(&a, 0, sizeof a);
  // chain world -> link from whichever model serves the scene
  int chain[MIR_MAX_BODY], n = 0;
  auto parent = [&](int b) { return h->kernel == 16 ? h->hm.b_parent[b] : h->hm64.b_parent[b]; };
  for (int b = link_body; b > 0; b = parent(b)) {
    if (n >= G) return mir_set_error(MIR_E_CAPACITY, "mir_inverse_kinematics: chain longer than 16 bodies");
    chain[n++] = b;
  }
  a.ch.n = n;
  // column of each scalar joint in the (B, n_arm) arrays = its rank among the scalar joints in body order
  int col_of_body[MIR_MAX_BODY];
  int narm = 0;
  for (int b = 1; b < h->nbody; b++) {
    const int jt = h->kernel == 16 ? h->hm.b_jtype[b] : h->hm64.b_jtype[b];
    col_of_body[b] = (jt == MIR_JNT_REVOLUTE || jt == MIR_JNT_PRISMATIC) ? narm++ : -1;
    if (col_of_body[b] >= 0) a.arm_qadr[col_of_body[b]] = h->kernel == 16 ? h->hm.b_qadr[b] : h->hm64.b_qadr[b];
  }
  for (int i = 0; i < n; i++) {
    const int b = chain[n - 1 - i];
    int jt;
    if (h->kernel == 16) {
      const DevModel& m = h->hm;
      jt = m.b_jtype[b];
      for (int k = 0; k < 3; k++) { a.ch.pos[i][k] = m.b_pos[b][k]; a.ch.axis[i][k] = m.b_axis[b][k]; }
      for (int k = 0; k < 4; k++) a.ch.quat[i][k] = m.b_quat[b][k];
      if (col_of_body[b] >= 0) { const int d = m.b_dofadr[b]; a.ch.lo[i] = m.d_lo[d]; a.ch.hi[i] = m.d_hi[d]; a.ch.limited[i] = m.d_limited[d]; }
    } else {
      const DevModel64& m = h->hm64;
      jt = m.b_jtype[b];
      for (int k = 0; k < 3; k++) { a.ch.pos[i][k] = m.b_pos[b][k]; a.ch.axis[i][k] = m.b_axis[b][k]; }
      for (int k = 0; k < 4; k++) a.ch.quat[i][k] = m.b_quat[b][k];
      if (col_of_body[b] >= 0) { const int d = m.b_dofadr[b]; a.ch.lo[i] = m.d_lo[d]; a.ch.hi[i] = m.d_hi[d]; a.ch.limited[i] = m.d_limited[d]; }
    }
    if (jt == MIR_JNT_FREE) return mir_set_error(MIR_E_INVALID, "mir_inverse_kinematics: the link hangs off a free body");
    a.ch.jtype[i] = jt;
    a.ch.qcol[i] = col_of_body[b];
  }
  // A FIXED element in front of another one is a constant: folded into that element's base transform (pos' = p_f + R_f pos, quat' = q_f quat)
  // the chain is one element shorter -- for the Panda's hand nine become eight, and the kernel's prefix scan three steps instead of four.
  {
    auto qmul_h = [](const double* p, const double* q, double* r) {
      r[0] = p[0] * q[0] - p[1] * q[1] - p[2] * q[2] - p[3] * q[3]; r[1] = p[0] * q[1] + p[1] * q[0] + p[2] * q[3] - p[3] * q[2];
      r[2] = p[0] * q[2] - p[1] * q[3] + p[2] * q[0] + p[3] * q[1]; r[3] = p[0] * q[3] + p[1] * q[2] - p[2] * q[1] + p[3] * q[0];
    };
    int m_ = 0;
    bool carry = false;
    double cp[3] = {0, 0, 0}, cq[4] = {1, 0, 0, 0};  // the fixed elements folded so far, waiting for the next element
    for (int i = 0; i < n; i++) {
      double p[3] = {a.ch.pos[i][0], a.ch.pos[i][1], a.ch.pos[i][2]}, q[4] = {a.ch.quat[i][0], a.ch.quat[i][1], a.ch.quat[i][2], a.ch.quat[i][3]};
      if (carry) {  // this element's base transform behind the carried one
        const double v[4] = {0, p[0], p[1], p[2]};
        double t[4], cqc[4] = {cq[0], -cq[1], -cq[2], -cq[3]}, rv[4], nq[4];
        qmul_h(cq, v, t); qmul_h(t, cqc, rv);
        p[0] = cp[0] + rv[1]; p[1] = cp[1] + rv[2]; p[2] = cp[2] + rv[3];
        qmul_h(cq, q, nq);
        for (int k = 0; k < 4; k++) q[k] = nq[k];
        carry = false;
      }
      const bool fixed_inner = a.ch.jtype[i] == MIR_JNT_FIXED && i < n - 1;
      if (fixed_inner) {
        for (int k = 0; k < 3; k++) cp[k] = p[k];
        for (int k = 0; k < 4; k++) cq[k] = q[k];
        carry = true;
        continue;
      }
      a.ch.jtype[m_] = a.ch.jtype[i]; a.ch.qcol[m_] = a.ch.qcol[i];
      a.ch.lo[m_] = a.ch.lo[i]; a.ch.hi[m_] = a.ch.hi[i]; a.ch.limited[m_] = a.ch.limited[i];
      for (int k = 0; k < 3; k++) { a.ch.pos[m_][k] = (float)p[k]; a.ch.axis[m_][k] = a.ch.axis[i][k]; }
      for (int k = 0; k < 4; k++) a.ch.quat[m_][k] = (float)q[k];
      m_++;
    }
    for (int i = m_; i < n; i++) { a.ch.jtype[i] = MIR_JNT_FIXED; a.ch.qcol[i] = -1; a.ch.limited[i] = 0; }
    a.ch.n = m_;
  }
  MirIkOptions o = {20, 1, 0.05, 5e-4, 5e-3, 0.5};
  if (opt) {
    o = *opt;
    if (o.max_iters <= 0 || !(o.damping > 0.0) || !(o.max_step > 0.0)) return mir_set_error(MIR_E_INVALID, "mir_inverse_kinematics: bad options");
  }
  a.target_pos = target_pos; a.target_quat = target_quat; a.init_qpos = init_qpos; a.scene_qpos = h->qpos;
  a.qst = h->pt.qst; a.n_arm = narm; a.qpos_out = qpos_out; a.err_out = err_out; a.B = h->B;
  a.n_rows = h->B; a.init_col0 = 0; a.init_ncols = narm;
  if (rows) {
    a.env_idx = reinterpret_cast<const long long*>(rows->env_idx);
    a.n_rows = rows->env_idx ? rows->n_rows : h->B;
    a.pos_by_env = (rows->flags & MIR_IK_POS_BY_ENV) ? 1 : 0; a.quat_by_env = (rows->flags & MIR_IK_QUAT_BY_ENV) ? 1 : 0;
    a.quat_one = (rows->flags & MIR_IK_QUAT_ONE) ? 1 : 0; a.init_by_env = (rows->flags & MIR_IK_INIT_BY_ENV) ? 1 : 0;
    if (rows->init_ncols > 0) {
      if (rows->init_col0 < 0 || rows->init_col0 + rows->init_ncols > narm) return mir_set_error(MIR_E_INVALID, "mir_inverse_kinematics_rows: init columns outside the joint row");
      a.init_col0 = rows->init_col0; a.init_ncols = rows->init_ncols;
    }
  }
  a.iters_out = h->dbg_ik_iters;
  a.max_iters = o.max_iters; a.respect_limits = o.respect_joint_limit;
  a.damping2 = (float)(o.damping * o.damping); a.pos_tol = (float)o.pos_tol; a.rot_tol = (float)o.rot_tol; a.inv_pos_tol = (float)(1.0 / o.pos_tol); a.inv_rot_tol = (float)(1.0 / o.rot_tol); a.max_step = (float)o.max_step;
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != h->device) (void)hipSetDevice(h->device);
  hipLaunchKernelGGL(mir_ik_kernel, dim3((a.n_rows + 3) / 4), dim3(64), 0, (hipStream_t)stream, a);
  hipError_t e = hipGetLastError();
  if (prev != h->device && prev >= 0) (void)hipSetDevice(prev);
  if (e != hipSuccess) return mir_set_error(MIR_E_HIP, hipGetErrorString(e));
  return MIR_OK;
}

extern "C" int mir_inverse_kinematics(MirHandle h, int32_t link_body, const float* target_pos, const float* target_quat, const float* init_qpos,
                                      const MirIkOptions* opt, float* qpos_out, float* err_out, void* stream) {
  return mir_inverse_kinematics_rows(h, link_body, nullptr, target_pos, target_quat, init_qpos, opt, qpos_out, err_out, stream);
}

/* debug aid (tools/probes/ik_iters.py): every following mir_inverse_kinematics call also writes the iterations each env took into
 * iters (B x i32, device; NULL switches it off) */
extern "C" int mir_debug_ik_iters(MirHandle h, int32_t* iters) {
  if (!h) return mir_set_error(MIR_E_INVALID, "null MirHandle");
  h->dbg_ik_iters = iters;
  return MIR_OK;
}

/* debug aid (tests/test_gpu_api.py): the row all-reduce of mir_dev.h on n_rows rows of 16 floats -- out[r][l] = what lane l of row r
 * holds after gsum.  Every lane of a row must hold the same bits: decisions of the 16-lane kernel's solver are taken per lane. */
namespace {
__global__ __launch_bounds__(64) void k_debug_row_sum(const float* __restrict__ in, float* __restrict__ out, int n_rows) {
  const int i = blockIdx.x * 64 + threadIdx.x;       // (whole waves: the DPP reads need every lane of the row present)
  const float v = i < n_rows * 16 ? in[i] : 0.0f;
  const float s = gsum(v);
  if (i < n_rows * 16) out[i] = s;
}
}  // namespace
extern "C" int mir_debug_row_sum(const float* in, float* out, int32_t n_rows, int device_id, void* stream) {
  if (!in || !out || n_rows <= 0) return mir_set_error(MIR_E_INVALID, "mir_debug_row_sum: bad argument");
  int prev = -1;
  const bool sw = hipGetDevice(&prev) == hipSuccess && prev != device_id && hipSetDevice(device_id) == hipSuccess;
  hipLaunchKernelGGL(k_debug_row_sum, dim3((n_rows * 16 + 63) / 64), dim3(64), 0, (hipStream_t)stream, in, out, n_rows);
  const hipError_t e = hipGetLastError();
  if (sw) (void)hipSetDevice(prev);
  return e == hipSuccess ? MIR_OK : mir_set_error(MIR_E_HIP, hipGetErrorString(e));
}


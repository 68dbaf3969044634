// mir_step.hip — the env.step() hot path as one fused HIP kernel for gfx950 (MI355X).
//
// What it replaces: scene.step() + get_obs() + compute_reward() of the reference
// (/root/reference/gym_genesis/tasks/franka/cube_pick.py:122-181, env.py:61-69), i.e. the
// Genesis rigid solver pipeline restated in SURVEY.md App. A.
//
// Mapping to CDNA4
//   * workgroup = ONE wave64 = 4 envs x 16 lanes.  Inside an env group, lane i is
//     "dof i", "body i", "contact i" or "geom i / i+16" depending on the phase, so every
//     per-dof / per-body model constant lives in that lane's registers.
//   * all per-env working data (poses, motion subspaces, mass matrix, contact rows,
//     Newton Hessian) lives in LDS; a step reads ~0.25 KB and writes ~0.3 KB of HBM per env.
//   * kinematic-tree recursions (velocities, accelerations, composite inertias, subtree
//     forces) are evaluated as mask-driven sums over ancestors / descendants, so they are
//     single phases with no depth-serial chain of barriers.
//   * single-wave workgroups: __syncthreads() is a wave-level LDS fence (no s_barrier
//     traffic between waves), cross-lane reductions are 16-wide xor shuffles.
//   * state is stored env-major (B, D): with 16 lanes per env a wave touches 4 contiguous
//     64-byte rows, the coalesced pattern for this lane mapping.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mir_model.h"
#include "mir_step.h"

#define G MIR_G
#define EPB 4 /* envs per block */
#define MST 17 /* padded row stride of 16x16 matrices in LDS (bank-conflict free column walks) */

namespace {

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 v3(float x, float y, float z) { return {x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 ld3(const float* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ void st3(float* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }

struct Q4 {
  float w, x, y, z;
};
__device__ __forceinline__ Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ V3 qrot(Q4 q, V3 v) {
  V3 u = {q.x, q.y, q.z};
  V3 t = 2.0f * cross(u, v);
  return v + q.w * t + cross(u, t);
}
__device__ __forceinline__ Q4 ld4(const float* p) { return {p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ void st4(float* p, Q4 q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
  float n = sqrtf(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < 1e-15f) return {1, 0, 0, 0};
  float s = 1.0f / n;
  return {q.w * s, q.x * s, q.y * s, q.z * s};
}
// row-major rotation matrix as three rows
struct M3 {
  V3 r0, r1, r2;
};
__device__ __forceinline__ M3 q2m(Q4 q) {
  float w = q.w, x = q.x, y = q.y, z = q.z;
  M3 R;
  R.r0 = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)};
  R.r1 = {2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)};
  R.r2 = {2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
  return R;
}
__device__ __forceinline__ V3 mcol(const M3& R, int k) {
  return k == 0 ? v3(R.r0.x, R.r1.x, R.r2.x) : (k == 1 ? v3(R.r0.y, R.r1.y, R.r2.y) : v3(R.r0.z, R.r1.z, R.r2.z));
}
__device__ __forceinline__ V3 mmul(const M3& R, V3 v) { return {dot(R.r0, v), dot(R.r1, v), dot(R.r2, v)}; }

// sum over the 16 lanes of an env group (result in every lane)
__device__ __forceinline__ float gsum(float v) {
  v += __shfl_xor(v, 1, G);
  v += __shfl_xor(v, 2, G);
  v += __shfl_xor(v, 4, G);
  v += __shfl_xor(v, 8, G);
  return v;
}

// spatial inertia {m, h, I(xx yy zz xy xz yz)} applied to motion {w, v} -> force {t, f}
struct Inert {
  float m;
  V3 h;
  float xx, yy, zz, xy, xz, yz;
};
__device__ __forceinline__ Inert ldI(const float* p) { return {p[0], {p[1], p[2], p[3]}, p[4], p[5], p[6], p[7], p[8], p[9]}; }
__device__ __forceinline__ void imul(const Inert& I, V3 w, V3 v, V3& t, V3& f) {
  V3 Iw = {I.xx * w.x + I.xy * w.y + I.xz * w.z, I.xy * w.x + I.yy * w.y + I.yz * w.z, I.xz * w.x + I.yz * w.y + I.zz * w.z};
  t = Iw + cross(I.h, v);
  f = I.m * v - cross(I.h, w);
}

// ---------------------------------------------------------------------------------------------
// per-env LDS working set
template <int MAXCON>
struct EnvLds {
  float qpos[20], qvel[G], target[G], qacc_ws[G], qacc[G];
  float lpos[G][3], lquat[G][4];
  float xpos[G][3], xquat[G][4];
  float cdof[G][6], cddq[G][6];
  float cinert[G][10], crb[G][10];
  float cvel[G][6], cfrc[G][6];
  float M[G][MST], H[G][MST];
  float qfs[G], qas[G], grad[G], srch[G], Ma[G], Mv[G];
  // collision
  float gpos[MIR_MAX_GEOM][3], gquat[MIR_MAX_GEOM][4];
  int cand[G];                 // candidate pair ids after the broadphase (ordered)
  int ccount[G];               // contacts produced by candidate k
  float stage[G][8][4];        // narrowphase output per candidate: pos, dist
  float snorm[G][3];           // normal per candidate
  int ncon, ncand, nlim_dummy, niter;
  // contacts
  float cpos[MAXCON][3], cnrm[MAXCON][3], ct1[MAXCON][3], ct2[MAXCON][3];
  float cdist[MAXCON], cmu[MAXCON], cD[MAXCON];
  int cb1[MAXCON], cb2[MAXCON];
  float caref[MAXCON][4], cjar[MAXCON][4], cjv[MAXCON][4];
  float cfb[MAXCON][3], cW[MAXCON][6];
  float Jb[MAXCON][3][MST];
  // joint-limit rows (lane i = dof i)
  float lsign[G], lD[G], laref[G], ljar[G], ljv[G], lf[G];
};

// ---------------------------------------------------------------------------------------------
// narrowphase primitives (one lane per candidate pair)

struct BoxG {
  V3 p;
  V3 a0, a1, a2;  // world axes
  V3 h;
};
__device__ __forceinline__ V3 bax(const BoxG& b, int k) { return k == 0 ? b.a0 : (k == 1 ? b.a1 : b.a2); }
__device__ __forceinline__ float bh(const BoxG& b, int k) { return k == 0 ? b.h.x : (k == 1 ? b.h.y : b.h.z); }

// plane z=0 of frame (pp, Rp) vs box; writes up to 4 points {pos, dist}; returns count
__device__ int plane_box(V3 pp, const M3& Rp, const BoxG& bx, float (*out)[4], V3& n) {
  n = mcol(Rp, 2);
  V3 eu = mcol(Rp, 0), ev = mcol(Rp, 1);
  float d[8], u[8], v[8];
  int cnt = 0;
#pragma unroll
  for (int c = 0; c < 8; c++) {
    V3 w = bx.p + ((c & 1) ? bx.h.x : -bx.h.x) * bx.a0 + ((c & 2) ? bx.h.y : -bx.h.y) * bx.a1 + ((c & 4) ? bx.h.z : -bx.h.z) * bx.a2;
    V3 rel = w - pp;
    d[c] = dot(rel, n);
    u[c] = dot(rel, eu);
    v[c] = dot(rel, ev);
    cnt += d[c] < 0.0f;
  }
  if (cnt == 0) return 0;
  // support extremes (+u, -u, +v, -v; first index wins ties) among penetrating corners when > 4 penetrate
  int p0 = -1, p1 = -1, p2 = -1, p3 = -1;
  if (cnt > 4) {
    float uM = -3e38f, um = 3e38f, vM = -3e38f, vm = 3e38f;
#pragma unroll
    for (int c = 0; c < 8; c++) {
      if (d[c] < 0.0f) {
        if (u[c] > uM) { uM = u[c]; p0 = c; }
        if (u[c] < um) { um = u[c]; p1 = c; }
        if (v[c] > vM) { vM = v[c]; p2 = c; }
        if (v[c] < vm) { vm = v[c]; p3 = c; }
      }
    }
  }
  int k = 0;
#pragma unroll
  for (int c = 0; c < 8; c++) {
    bool keep = d[c] < 0.0f && (cnt <= 4 || c == p0 || c == p1 || c == p2 || c == p3);
    if (keep && k < 4) {
      V3 w = bx.p + ((c & 1) ? bx.h.x : -bx.h.x) * bx.a0 + ((c & 2) ? bx.h.y : -bx.h.y) * bx.a1 + ((c & 4) ? bx.h.z : -bx.h.z) * bx.a2;
      V3 pos = w - (0.5f * d[c]) * n;
      out[k][0] = pos.x; out[k][1] = pos.y; out[k][2] = pos.z; out[k][3] = d[c];
      k++;
    }
  }
  return k;
}

// box-box by separating axes + reference-face clipping; normal from A to B; up to 8 points
__device__ int box_box(const BoxG& A, const BoxG& B, float (*out)[4], V3& nout) {
  V3 t = B.p - A.p;
  float best = -1e30f;
  int code = -1;
  V3 bestL = v3(0, 0, 1);
#pragma unroll
  for (int c = 0; c < 6; c++) {
    V3 L = c < 3 ? bax(A, c) : bax(B, c - 3);
    float ra = A.h.x * fabsf(dot(A.a0, L)) + A.h.y * fabsf(dot(A.a1, L)) + A.h.z * fabsf(dot(A.a2, L));
    float rb = B.h.x * fabsf(dot(B.a0, L)) + B.h.y * fabsf(dot(B.a1, L)) + B.h.z * fabsf(dot(B.a2, L));
    float s = fabsf(dot(t, L)) - (ra + rb);
    if (s > 0.0f) return 0;
    if (s > best) { best = s; code = c; bestL = L; }
  }
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      V3 L = cross(bax(A, i), bax(B, j));
      float len = sqrtf(dot(L, L));
      if (len < 1e-3f) continue;
      L = (1.0f / len) * L;
      float ra = A.h.x * fabsf(dot(A.a0, L)) + A.h.y * fabsf(dot(A.a1, L)) + A.h.z * fabsf(dot(A.a2, L));
      float rb = B.h.x * fabsf(dot(B.a0, L)) + B.h.y * fabsf(dot(B.a1, L)) + B.h.z * fabsf(dot(B.a2, L));
      float s = fabsf(dot(t, L)) - (ra + rb);
      if (s > 0.0f) return 0;
      if (s * 1.05f > best && s > best + 1e-6f) { best = s; code = 6 + i * 3 + j; bestL = L; }
    }
  V3 n = dot(t, bestL) < 0.0f ? -1.0f * bestL : bestL;
  nout = n;
  if (code >= 6) {
    int i = (code - 6) / 3, j = (code - 6) % 3;
    V3 PA = A.p, PB = B.p;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      if (k != i) PA = PA + (dot(n, bax(A, k)) > 0.0f ? bh(A, k) : -bh(A, k)) * bax(A, k);
      if (k != j) PB = PB + (dot(n, bax(B, k)) > 0.0f ? -bh(B, k) : bh(B, k)) * bax(B, k);
    }
    V3 ua = bax(A, i), ub = bax(B, j), dd = PB - PA;
    float uaub = dot(ua, ub), q1 = dot(ua, dd), q2 = -dot(ub, dd), den = 1.0f - uaub * uaub;
    float alpha = 0.0f, beta = 0.0f;
    if (den > 1e-6f) { alpha = (q1 + uaub * q2) / den; beta = (uaub * q1 + q2) / den; }
    PA = PA + alpha * ua;
    PB = PB + beta * ub;
    V3 pos = 0.5f * (PA + PB);
    out[0][0] = pos.x; out[0][1] = pos.y; out[0][2] = pos.z; out[0][3] = best;
    return 1;
  }
  // reference / incident boxes
  const bool refA = code < 3;
  const BoxG& R = refA ? A : B;
  const BoxG& I = refA ? B : A;
  int k = refA ? code : code - 3;
  V3 nr = refA ? n : -1.0f * n;
  int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
  V3 fc = R.p + bh(R, k) * nr;
  float a0 = fabsf(dot(nr, I.a0)), a1 = fabsf(dot(nr, I.a1)), a2 = fabsf(dot(nr, I.a2));
  int jb = 0;
  float mx = a0;
  if (a1 > mx) { mx = a1; jb = 1; }
  if (a2 > mx) { mx = a2; jb = 2; }
  float sj = dot(nr, bax(I, jb)) > 0.0f ? -1.0f : 1.0f;
  int j1 = (jb + 1) % 3, j2 = (jb + 2) % 3;
  V3 ic = I.p + (sj * bh(I, jb)) * bax(I, jb);
  V3 e1 = bax(R, k1), e2 = bax(R, k2);
  float h1 = bh(R, k1), h2 = bh(R, k2);
  float px[9], py[9], pz[9], qx[9], qy[9], qz[9];
  int np = 4;
  {
    const float sx[4] = {1, -1, -1, 1}, sy[4] = {1, 1, -1, -1};
#pragma unroll
    for (int v = 0; v < 4; v++) {
      V3 w = ic + (sx[v] * bh(I, j1)) * bax(I, j1) + (sy[v] * bh(I, j2)) * bax(I, j2);
      V3 rel = w - fc;
      px[v] = dot(rel, e1); py[v] = dot(rel, e2); pz[v] = dot(rel, nr);
    }
  }
  for (int e = 0; e < 4; e++) {
    const int ax = e >> 1;
    const float sg = (e & 1) ? -1.0f : 1.0f;
    const float lim = ax == 0 ? h1 : h2;
    int nn = 0;
    for (int v = 0; v < np; v++) {
      int w = v + 1 == np ? 0 : v + 1;
      float pc = ax == 0 ? px[v] : py[v], qc = ax == 0 ? px[w] : py[w];
      float dp = sg * pc - lim, dq = sg * qc - lim;
      if (dp <= 0.0f && nn < 9) { qx[nn] = px[v]; qy[nn] = py[v]; qz[nn] = pz[v]; nn++; }
      if ((dp <= 0.0f) != (dq <= 0.0f) && nn < 9) {
        float u = dp / (dp - dq);
        qx[nn] = px[v] + u * (px[w] - px[v]); qy[nn] = py[v] + u * (py[w] - py[v]); qz[nn] = pz[v] + u * (pz[w] - pz[v]);
        nn++;
      }
    }
    np = nn;
    for (int v = 0; v < np; v++) { px[v] = qx[v]; py[v] = qy[v]; pz[v] = qz[v]; }
    if (np == 0) return 0;
  }
  int cnt = 0;
  for (int v = 0; v < np && cnt < 8; v++) {
    if (pz[v] < 0.0f) {
      V3 w = fc + px[v] * e1 + py[v] * e2 + (0.5f * pz[v]) * nr;
      out[cnt][0] = w.x; out[cnt][1] = w.y; out[cnt][2] = w.z; out[cnt][3] = pz[v];
      cnt++;
    }
  }
  return cnt;
}

// MuJoCo-style impedance from solimp at |pos|
__device__ __forceinline__ float impedance(float dmin, float dmax, float width, float mid, float power, float pos) {
  dmin = fminf(fmaxf(dmin, 1e-4f), 0.9999f);
  dmax = fminf(fmaxf(dmax, 1e-4f), 0.9999f);
  width = fmaxf(width, 1e-15f);
  mid = fminf(fmaxf(mid, 1e-4f), 0.9999f);
  power = fmaxf(power, 1.0f);
  float x = fabsf(pos) / width, y;
  if (x >= 1.0f) y = 1.0f;
  else if (x <= 0.0f) y = 0.0f;
  else if (x <= mid) y = (power == 2.0f ? (x / mid) * (x / mid) : powf(x / mid, power)) * mid;
  else { float r = (1.0f - x) / (1.0f - mid); y = 1.0f - (power == 2.0f ? r * r : powf(r, power)) * (1.0f - mid); }
  return dmin + y * (dmax - dmin);
}

// in-place Cholesky of the nv x nv SPD matrix A (LDS, row stride MST) by the 16 lanes of a group,
// lane = row.  Lower triangle holds L on exit.
__device__ __forceinline__ void group_chol(float (*A)[MST], int nv, int lane) {
  for (int k = 0; k < nv; k++) {
    float akk = A[k][k];
    float piv = sqrtf(fmaxf(akk, 1e-30f));
    float lik = 0.0f;
    if (lane > k && lane < nv) lik = A[lane][k] / piv;
    __syncthreads();
    if (lane == k) A[k][k] = piv;
    if (lane > k && lane < nv) A[lane][k] = lik;
    __syncthreads();
    if (lane > k && lane < nv) {
      for (int j = k + 1; j <= lane; j++) A[lane][j] -= lik * A[j][k];
    }
    __syncthreads();
  }
}
// solve L L^T x = b; x, b in LDS vectors (x may alias b); lane = row
__device__ __forceinline__ void group_cholsolve(float (*L)[MST], int nv, int lane, float* x) {
  for (int k = 0; k < nv; k++) {
    float yk = x[k] / L[k][k];
    __syncthreads();
    if (lane == k) x[k] = yk;
    if (lane > k && lane < nv) x[lane] -= L[lane][k] * yk;
    __syncthreads();
  }
  for (int k = nv - 1; k >= 0; k--) {
    float xk = x[k] / L[k][k];
    __syncthreads();
    if (lane == k) x[k] = xk;
    if (lane < k) x[lane] -= L[k][lane] * xk;
    __syncthreads();
  }
}


// forward kinematics of one env group: local joint transforms, then every body composes its own
// ancestor chain (leaf -> root) independently; two phases, no depth-serial barriers
template <int MAXCON>
__device__ __forceinline__ void group_fk(EnvLds<MAXCON>& S, const DevModel* __restrict__ m, int lane, int nb, int b_parent,
                                         int b_jtype, int b_qadr, V3 b_pos, Q4 b_quat, V3 b_axis) {
  if (lane > 0 && lane < nb) {
    Q4 ql = b_quat;
    V3 pl = b_pos;
    if (b_jtype == MIR_JNT_REVOLUTE) {
      float ang = S.qpos[b_qadr], sn, cs;
      sincosf(0.5f * ang, &sn, &cs);
      ql = qmul(b_quat, Q4{cs, b_axis.x * sn, b_axis.y * sn, b_axis.z * sn});
    } else if (b_jtype == MIR_JNT_PRISMATIC) {
      pl = b_pos + qrot(b_quat, S.qpos[b_qadr] * b_axis);
    } else if (b_jtype == MIR_JNT_FREE) {
      pl = ld3(&S.qpos[b_qadr]);
      ql = qnormalize(ld4(&S.qpos[b_qadr + 3]));
    }
    st3(S.lpos[lane], pl);
    st4(S.lquat[lane], ql);
  } else if (lane == 0) {
    st3(S.lpos[0], v3(0, 0, 0));
    st4(S.lquat[0], Q4{1, 0, 0, 0});
  }
  __syncthreads();
  if (lane < nb) {
    V3 P = ld3(S.lpos[lane]);
    Q4 Qx = ld4(S.lquat[lane]);
    int anc = lane > 0 ? b_parent : -1;
    while (anc > 0) {
      Q4 qa = ld4(S.lquat[anc]);
      P = ld3(S.lpos[anc]) + qrot(qa, P);
      Qx = qmul(qa, Qx);
      anc = m->b_parent[anc];
    }
    st3(S.xpos[lane], P);
    st4(S.xquat[lane], Qx);
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
template <int MAXCON>
__global__ __launch_bounds__(64) void mir_step_kernel(StepArgs a) {
  __shared__ EnvLds<MAXCON> s_env[EPB];
  const DevModel* __restrict__ m = a.model;
  const int tid = threadIdx.x;
  const int lane = tid & (G - 1);
  const int grp = tid >> 4;
  const int env_raw = blockIdx.x * EPB + grp;
  const bool valid = env_raw < a.B;
  const int env = valid ? env_raw : a.B - 1;
  EnvLds<MAXCON>& S = s_env[grp];

  const int nb = m->nbody, nv = m->nv, nq = m->nq, qst = m->qstride;
  const float dt = m->dt;

  // ---- per-lane model constants (lane = body = dof) -------------------------------------------
  const bool isbody = lane < nb && lane > 0;
  const bool isdof = lane < nv;
  const int b_parent = m->b_parent[lane], b_jtype = m->b_jtype[lane], b_qadr = m->b_qadr[lane], b_dofadr = m->b_dofadr[lane];
  const int b_root = m->b_root[lane];
  const uint32_t b_dofmask = m->b_dofmask[lane], b_submask = m->b_submask[lane];
  const V3 b_pos = ld3(m->b_pos[lane]), b_axis = ld3(m->b_axis[lane]), b_ipos = ld3(m->b_ipos[lane]);
  const Q4 b_quat = ld4(m->b_quat[lane]);
  const float b_mass = m->b_mass[lane];
  const int d_body = m->d_body[lane], d_kind = m->d_kind[lane], d_qadr = m->d_qadr[lane], d_axis_k = m->d_axis_k[lane];
  const uint32_t d_premask = m->d_premask[lane], d_ancmask = m->d_ancmask[lane];
  const int d_ctrl = m->d_ctrl[lane], d_uadr = m->d_uadr[lane], d_limited = m->d_limited[lane];
  const float d_damping = m->d_damping[lane], d_kp = m->d_kp[lane], d_kv = m->d_kv[lane];
  const float d_frclo = m->d_frclo[lane], d_frchi = m->d_frchi[lane], d_mdiag = m->d_mdiag[lane];

  // ---- load state -----------------------------------------------------------------------------
  for (int i = lane; i < qst; i += G) S.qpos[i] = a.qpos[(size_t)env * qst + i];
  S.qvel[lane] = a.qvel[(size_t)env * G + lane];
  S.qacc_ws[lane] = a.qacc_ws[(size_t)env * G + lane];
  {
    float tg = a.target[(size_t)env * G + lane];
    if (a.action && isdof && d_uadr >= 0) tg = a.action[(size_t)env * m->nu + d_uadr];
    S.target[lane] = tg;
    if (a.action && valid) a.target[(size_t)env * G + lane] = tg;
  }
  if (lane == 0) { S.ncon = 0; S.ncand = 0; S.niter = 0; }
  __syncthreads();

  const int nsteps = a.mode == 0 ? a.n_steps : (a.mode == 1 ? 1 : 0);
  for (int step = 0; step < nsteps; step++) {
    // ======================= forward kinematics =================================================
    group_fk(S, m, lane, nb, b_parent, b_jtype, b_qadr, b_pos, b_quat, b_axis);
    // motion subspaces (lane = dof) and body inertias about the tree reference point (lane = body)
    if (isdof) {
      V3 ang = v3(0, 0, 0), lin = v3(0, 0, 0);
      V3 r = ld3(S.xpos[m->b_root[d_body]]) - ld3(S.xpos[d_body]);
      if (d_kind < 2) {
        V3 ax = qrot(ld4(S.xquat[d_body]), ld3(m->b_axis[d_body]));
        if (d_kind == 0) { ang = ax; lin = cross(ax, r); }
        else lin = ax;
      } else {
        V3 e = v3(d_axis_k == 0, d_axis_k == 1, d_axis_k == 2);
        if (d_kind == 2) lin = e;
        else { ang = e; lin = cross(e, r); }
      }
      st3(&S.cdof[lane][0], ang);
      st3(&S.cdof[lane][3], lin);
    }
    if (isbody) {
      Q4 q = ld4(S.xquat[lane]);
      M3 R = q2m(q);
      const float* ib = m->b_inertia[lane];
      // W = R Ib R^T
      float Ib[3][3] = {{ib[0], ib[3], ib[4]}, {ib[3], ib[1], ib[5]}, {ib[4], ib[5], ib[2]}};
      float Rm[3][3] = {{R.r0.x, R.r0.y, R.r0.z}, {R.r1.x, R.r1.y, R.r1.z}, {R.r2.x, R.r2.y, R.r2.z}};
      float T[3][3], W[3][3];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T[i][j] = Rm[i][0] * Ib[0][j] + Rm[i][1] * Ib[1][j] + Rm[i][2] * Ib[2][j];
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) W[i][j] = T[i][0] * Rm[j][0] + T[i][1] * Rm[j][1] + T[i][2] * Rm[j][2];
      V3 xip = ld3(S.xpos[lane]) + mmul(R, b_ipos);
      V3 r = xip - ld3(S.xpos[b_root]);
      float rr = dot(r, r);
      float* c = S.cinert[lane];
      c[0] = b_mass; c[1] = b_mass * r.x; c[2] = b_mass * r.y; c[3] = b_mass * r.z;
      c[4] = W[0][0] + b_mass * (rr - r.x * r.x);
      c[5] = W[1][1] + b_mass * (rr - r.y * r.y);
      c[6] = W[2][2] + b_mass * (rr - r.z * r.z);
      c[7] = W[0][1] - b_mass * r.x * r.y;
      c[8] = W[0][2] - b_mass * r.x * r.z;
      c[9] = W[1][2] - b_mass * r.y * r.z;
    } else {
#pragma unroll
      for (int k = 0; k < 10; k++) S.cinert[lane][k] = 0.0f;
    }
    __syncthreads();

    // ======================= velocities, composite inertias =====================================
    {
      // lane = dof: cdof_dot * qvel, with "velocity before this dof" from the pre-mask
      V3 pw = v3(0, 0, 0), pv = v3(0, 0, 0);
      if (isdof) {
        uint32_t mk = d_premask;
        while (mk) {
          int j = __ffs(mk) - 1;
          mk &= mk - 1;
          float qd = S.qvel[j];
          pw = pw + qd * ld3(&S.cdof[j][0]);
          pv = pv + qd * ld3(&S.cdof[j][3]);
        }
        V3 cw = ld3(&S.cdof[lane][0]), cv = ld3(&S.cdof[lane][3]);
        float qd = S.qvel[lane];
        st3(&S.cddq[lane][0], qd * cross(pw, cw));
        st3(&S.cddq[lane][3], qd * (cross(pw, cv) + cross(pv, cw)));
      }
      // lane = body: cvel, crb
      V3 w = v3(0, 0, 0), v = v3(0, 0, 0);
      float acc[10];
#pragma unroll
      for (int k = 0; k < 10; k++) acc[k] = 0.0f;
      if (isbody) {
        uint32_t mk = b_dofmask;
        while (mk) {
          int j = __ffs(mk) - 1;
          mk &= mk - 1;
          float qd = S.qvel[j];
          w = w + qd * ld3(&S.cdof[j][0]);
          v = v + qd * ld3(&S.cdof[j][3]);
        }
        uint32_t sm = b_submask;
        while (sm) {
          int c = __ffs(sm) - 1;
          sm &= sm - 1;
#pragma unroll
          for (int k = 0; k < 10; k++) acc[k] += S.cinert[c][k];
        }
      }
      st3(&S.cvel[lane][0], w);
      st3(&S.cvel[lane][3], v);
#pragma unroll
      for (int k = 0; k < 10; k++) S.crb[lane][k] = acc[k];
    }
    __syncthreads();

    // ======================= body forces (RNE, qacc=0) and mass matrix rows =======================
    {
      V3 t = v3(0, 0, 0), f = v3(0, 0, 0);
      if (isbody) {
        V3 aw = v3(0, 0, 0), av = v3(-m->gx, -m->gy, -m->gz);
        uint32_t mk = b_dofmask;
        while (mk) {
          int j = __ffs(mk) - 1;
          mk &= mk - 1;
          aw = aw + ld3(&S.cddq[j][0]);
          av = av + ld3(&S.cddq[j][3]);
        }
        Inert I = ldI(S.cinert[lane]);
        V3 w = ld3(&S.cvel[lane][0]), v = ld3(&S.cvel[lane][3]);
        V3 ta, fa, tv, fv;
        imul(I, aw, av, ta, fa);
        imul(I, w, v, tv, fv);
        t = ta + cross(w, tv) + cross(v, fv);
        f = fa + cross(w, fv);
      }
      st3(&S.cfrc[lane][0], t);
      st3(&S.cfrc[lane][3], f);
      // mass matrix row (lane = dof i): M[i][j] = cdof_j . (crb_body(i) cdof_i), j in ancestors-or-self
#pragma unroll
      for (int j = 0; j < G; j++) S.M[lane][j] = 0.0f;
    }
    __syncthreads();
    if (isdof) {
      Inert I = ldI(S.crb[d_body]);
      V3 bt, bf;
      imul(I, ld3(&S.cdof[lane][0]), ld3(&S.cdof[lane][3]), bt, bf);
      uint32_t mk = d_ancmask;
      while (mk) {
        int j = __ffs(mk) - 1;
        mk &= mk - 1;
        float val = dot(ld3(&S.cdof[j][0]), bt) + dot(ld3(&S.cdof[j][3]), bf);
        if (j == lane) val += d_mdiag;
        S.M[lane][j] = val;
        S.M[j][lane] = val;
      }
    }
    // bias + smooth force (lane = dof)
    float qfrc_bias = 0.0f;
    if (isdof) {
      V3 t = v3(0, 0, 0), f = v3(0, 0, 0);
      uint32_t sm = m->b_submask[d_body];
      while (sm) {
        int c = __ffs(sm) - 1;
        sm &= sm - 1;
        t = t + ld3(&S.cfrc[c][0]);
        f = f + ld3(&S.cfrc[c][3]);
      }
      qfrc_bias = dot(ld3(&S.cdof[lane][0]), t) + dot(ld3(&S.cdof[lane][3]), f);
      float qd = S.qvel[lane];
      float fa = 0.0f;
      if (d_ctrl == MIR_CTRL_POSITION) {
        fa = d_kp * (S.target[lane] - S.qpos[d_qadr]) - d_kv * qd;
        fa = fminf(fmaxf(fa, d_frclo), d_frchi);
      }
      float qs = -d_damping * qd + fa - qfrc_bias;
      S.qfs[lane] = qs;
      S.qas[lane] = qs;
    } else {
      S.qfs[lane] = 0.0f;
      S.qas[lane] = 0.0f;
    }
    __syncthreads();
    if (a.out_M && valid && isdof && step == 0) {
      for (int j = 0; j < nv; j++) a.out_M[((size_t)env * nv + lane) * nv + j] = S.M[lane][j] - (j == lane ? d_mdiag - m->d_armature[lane] : 0.0f);
    }
    if (a.out_bias && valid && isdof && step == 0) a.out_bias[(size_t)env * nv + lane] = qfrc_bias;
    // qacc_smooth = Mt^-1 qfrc_smooth : factor a copy (H) and solve in place in qas
#pragma unroll
    for (int j = 0; j < G; j++) S.H[lane][j] = S.M[lane][j];
    __syncthreads();
    group_chol(S.H, nv, lane);
    group_cholsolve(S.H, nv, lane, S.qas);
    if (a.out_qas && valid && isdof && step == 0) a.out_qas[(size_t)env * nv + lane] = S.qas[lane];

    // ======================= collision detection ================================================
    if (lane == 0) { S.ncon = 0; S.ncand = 0; }
    for (int g = lane; g < m->ngeom; g += G) {
      int gb = m->g_body[g];
      Q4 qb = ld4(S.xquat[gb]);
      st3(S.gpos[g], ld3(S.xpos[gb]) + qrot(qb, ld3(m->g_pos[g])));
      st4(S.gquat[g], qmul(qb, ld4(m->g_quat[g])));
    }
    __syncthreads();
    if (m->enable_collision) {
      // broadphase: bounding test per static candidate pair, ordered compaction of survivors
      int base = 0;
      for (int p0 = 0; p0 < m->npair; p0 += G) {
        int p = p0 + lane;
        bool hit = false;
        if (p < m->npair) {
          int g1 = m->p_g1[p], g2 = m->p_g2[p];
          V3 h2 = ld3(m->g_size[g2]);
          M3 R2 = q2m(ld4(S.gquat[g2]));
          V3 c2 = ld3(S.gpos[g2]);
          if (m->g_type[g1] == MIR_GEOM_PLANE) {
            V3 n = mcol(q2m(ld4(S.gquat[g1])), 2);
            float ext = h2.x * fabsf(dot(n, mcol(R2, 0))) + h2.y * fabsf(dot(n, mcol(R2, 1))) + h2.z * fabsf(dot(n, mcol(R2, 2)));
            hit = dot(c2 - ld3(S.gpos[g1]), n) - ext < 0.0f;
          } else {
            V3 h1 = ld3(m->g_size[g1]);
            float r1 = sqrtf(dot(h1, h1)), r2 = sqrtf(dot(h2, h2));
            V3 dc = c2 - ld3(S.gpos[g1]);
            float rs = r1 + r2;
            hit = dot(dc, dc) <= rs * rs;
          }
        }
        unsigned long long bal = __ballot(hit);
        uint32_t gm = (uint32_t)(bal >> (grp * G)) & 0xffffu;
        int pos = base + __popc(gm & ((1u << lane) - 1u));
        if (hit && pos < G) S.cand[pos] = p;
        base += __popc(gm);
      }
      if (lane == 0) S.ncand = base < G ? base : G;
      __syncthreads();
      // narrowphase: lane k handles candidate k
      const int ncand = S.ncand;
      int mycount = 0;
      if (lane < ncand) {
        int p = S.cand[lane];
        int g1 = m->p_g1[p], g2 = m->p_g2[p];
        M3 R2 = q2m(ld4(S.gquat[g2]));
        BoxG B2 = {ld3(S.gpos[g2]), mcol(R2, 0), mcol(R2, 1), mcol(R2, 2), ld3(m->g_size[g2])};
        V3 n = v3(0, 0, 1);
        if (m->g_type[g1] == MIR_GEOM_PLANE) {
          M3 R1 = q2m(ld4(S.gquat[g1]));
          mycount = plane_box(ld3(S.gpos[g1]), R1, B2, S.stage[lane], n);
        } else {
          M3 R1 = q2m(ld4(S.gquat[g1]));
          BoxG B1 = {ld3(S.gpos[g1]), mcol(R1, 0), mcol(R1, 1), mcol(R1, 2), ld3(m->g_size[g1])};
          mycount = box_box(B1, B2, S.stage[lane], n);
        }
        st3(S.snorm[lane], n);
      }
      // ordered compaction of contact points (exclusive prefix over candidates, 16-wide scan)
      int incl = mycount;
#pragma unroll
      for (int o = 1; o < G; o <<= 1) {
        int up = __shfl_up(incl, o, G);
        if (lane >= o) incl += up;
      }
      int off = incl - mycount;
      int total = __shfl(incl, G - 1, G);
      const int maxc = m->max_contacts < MAXCON ? m->max_contacts : MAXCON;
      if (lane == 0) S.ncon = total < maxc ? total : maxc;
      if (lane < ncand) {
        int p = S.cand[lane];
        int g1 = m->p_g1[p], g2 = m->p_g2[p];
        V3 n = ld3(S.snorm[lane]);
        // contact frame (same construction as the oracle)
        V3 t1 = fabsf(n.y) < 0.5f ? v3(0, 1, 0) : v3(0, 0, 1);
        t1 = t1 - dot(n, t1) * n;
        t1 = (1.0f / sqrtf(dot(t1, t1))) * t1;
        V3 t2 = cross(n, t1);
        float mu = fmaxf(m->g_friction[g1], m->g_friction[g2]);
        float sr0 = 0.5f * (m->g_solref[g1][0] + m->g_solref[g2][0]), sr1 = 0.5f * (m->g_solref[g1][1] + m->g_solref[g2][1]);
        float si[5];
#pragma unroll
        for (int k = 0; k < 5; k++) si[k] = 0.5f * (m->g_solimp[g1][k] + m->g_solimp[g2][k]);
        int b1 = m->g_body[g1], b2 = m->g_body[g2];
        float wsum = m->b_invweight0[b1] + m->b_invweight0[b2];
        float dmax = fminf(fmaxf(si[1], 1e-4f), 0.9999f);
        float tc = fmaxf(sr0, 2.0f * dt);
        float kk = 1.0f / (dmax * dmax * tc * tc * sr1 * sr1), bb = 2.0f / (dmax * tc);
        for (int c = 0; c < mycount; c++) {
          int k = off + c;
          if (k >= maxc) break;
          float dist = S.stage[lane][c][3];
          st3(S.cpos[k], ld3(S.stage[lane][c]));
          st3(S.cnrm[k], n); st3(S.ct1[k], t1); st3(S.ct2[k], t2);
          S.cdist[k] = dist; S.cmu[k] = mu; S.cb1[k] = b1; S.cb2[k] = b2;
          float imp = impedance(si[0], si[1], si[2], si[3], si[4], dist);
          float Rr = fmaxf(2.0f * mu * mu * (1.0f - imp) / imp * wsum * (1.0f + mu * mu), 1e-15f);
          S.cD[k] = 1.0f / Rr;
          // aref needs the row velocities: store -k*imp*dist now, velocity term added after Jb is built
          S.caref[k][0] = -kk * imp * dist;
          S.caref[k][1] = bb;  // stash b
        }
      }
      __syncthreads();
    }
    const int ncon = S.ncon;

    // ======================= constraint rows ======================================================
    // contact base Jacobians: lane = dof; Jb[c][r][i], r = normal, t1, t2
    for (int c = 0; c < ncon; c++) {
      float jn = 0.0f, j1 = 0.0f, j2 = 0.0f;
      if (isdof) {
        int b1 = S.cb1[c], b2 = S.cb2[c];
        const bool in2 = m->b_dofmask[b2] >> lane & 1u, in1 = m->b_dofmask[b1] >> lane & 1u;
        const float sgn = (in2 ? 1.0f : 0.0f) - (in1 ? 1.0f : 0.0f);  // a dof moving both bodies cancels
        const int bref = in2 ? b2 : b1;
        if (sgn != 0.0f) {
          V3 r = ld3(S.cpos[c]) - ld3(S.xpos[m->b_root[bref]]);
          V3 vel = cross(ld3(&S.cdof[lane][0]), r) + ld3(&S.cdof[lane][3]);
          jn = sgn * dot(vel, ld3(S.cnrm[c]));
          j1 = sgn * dot(vel, ld3(S.ct1[c]));
          j2 = sgn * dot(vel, ld3(S.ct2[c]));
        }
      }
      S.Jb[c][0][lane] = jn; S.Jb[c][1][lane] = j1; S.Jb[c][2][lane] = j2;
    }
    // joint-limit rows: lane = dof
    {
      float sgn = 0.0f, D = 0.0f, aref = 0.0f;
      if (isdof && d_limited && m->enable_joint_limit) {
        float q = S.qpos[d_qadr];
        float dlo = q - m->d_lo[lane], dhi = m->d_hi[lane] - q;
        float pos = 0.0f;
        if (dlo < 0.0f) { pos = dlo; sgn = 1.0f; }
        else if (dhi < 0.0f) { pos = dhi; sgn = -1.0f; }
        if (sgn != 0.0f) {
          const float* si = m->d_solimp[lane];
          float imp = impedance(si[0], si[1], si[2], si[3], si[4], pos);
          float Rr = fmaxf((1.0f - imp) / imp * m->d_invweight0[lane], 1e-15f);
          D = 1.0f / Rr;
          aref = -m->d_b[lane] * (sgn * S.qvel[lane]) - m->d_k[lane] * imp * pos;
        }
      }
      S.lsign[lane] = sgn; S.lD[lane] = D; S.laref[lane] = aref;
    }
    __syncthreads();
    // contact reference accelerations (lane = contact): aref_r = -b (J_r qvel) - k imp dist
    if (lane < ncon) {
      float vn = 0.0f, v1 = 0.0f, v2 = 0.0f;
      for (int i = 0; i < nv; i++) {
        float qd = S.qvel[i];
        vn += S.Jb[lane][0][i] * qd; v1 += S.Jb[lane][1][i] * qd; v2 += S.Jb[lane][2][i] * qd;
      }
      float base = S.caref[lane][0], bb = S.caref[lane][1], mu = S.cmu[lane];
      S.caref[lane][0] = base - bb * (vn + mu * v1);
      S.caref[lane][1] = base - bb * (vn - mu * v1);
      S.caref[lane][2] = base - bb * (vn + mu * v2);
      S.caref[lane][3] = base - bb * (vn - mu * v2);
    }
    __syncthreads();

    // ======================= primal Newton solve ====================================================
    const float lsg = S.lsign[lane];
    const uint32_t limmask = (uint32_t)(__ballot(lsg != 0.0f) >> (grp * G)) & 0xffffu;
    const int nefc = 4 * ncon + __popc(limmask);
    bool done = nefc == 0;
    if (isdof) S.qacc[lane] = S.qas[lane];
    __syncthreads();
    if (!done) {
      // warm start: cost(ws) vs cost(qacc_smooth)
      float dq = isdof ? S.qacc_ws[lane] - S.qas[lane] : 0.0f;
      S.srch[lane] = dq;
      __syncthreads();
      float c_ws = 0.0f, c_sm = 0.0f;
      if (isdof) {
        float tsum = 0.0f;
        for (int j = 0; j < nv; j++) tsum += S.M[lane][j] * S.srch[j];
        c_ws += 0.5f * tsum * dq;
        if (lsg != 0.0f) {
          float js = lsg * S.qas[lane] - S.laref[lane], jw = lsg * S.qacc_ws[lane] - S.laref[lane];
          if (js < 0.0f) c_sm += 0.5f * S.lD[lane] * js * js;
          if (jw < 0.0f) c_ws += 0.5f * S.lD[lane] * jw * jw;
        }
      }
      if (lane < ncon) {
        float sn = 0, s1 = 0, s2 = 0, wn = 0, w1 = 0, w2 = 0;
        for (int i = 0; i < nv; i++) {
          float as = S.qas[i], aw = S.qacc_ws[i];
          float jn = S.Jb[lane][0][i], j1 = S.Jb[lane][1][i], j2 = S.Jb[lane][2][i];
          sn += jn * as; s1 += j1 * as; s2 += j2 * as;
          wn += jn * aw; w1 += j1 * aw; w2 += j2 * aw;
        }
        float mu = S.cmu[lane], D = S.cD[lane];
        float xs[4] = {sn + mu * s1, sn - mu * s1, sn + mu * s2, sn - mu * s2};
        float xw[4] = {wn + mu * w1, wn - mu * w1, wn + mu * w2, wn - mu * w2};
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float js = xs[r] - S.caref[lane][r], jw = xw[r] - S.caref[lane][r];
          if (js < 0.0f) c_sm += 0.5f * D * js * js;
          if (jw < 0.0f) c_ws += 0.5f * D * jw * jw;
        }
      }
      c_ws = gsum(c_ws);
      c_sm = gsum(c_sm);
      if (isdof) S.qacc[lane] = c_ws < c_sm ? S.qacc_ws[lane] : S.qas[lane];
    }
    __syncthreads();
    // Ma, jar at the starting point
    {
      float ma = 0.0f;
      if (isdof)
        for (int j = 0; j < nv; j++) ma += S.M[lane][j] * S.qacc[j];
      S.Ma[lane] = ma;
      S.ljar[lane] = lsg != 0.0f ? lsg * S.qacc[lane] - S.laref[lane] : 0.0f;
      if (lane < ncon) {
        float xn = 0, x1 = 0, x2 = 0;
        for (int i = 0; i < nv; i++) {
          float ai = S.qacc[i];
          xn += S.Jb[lane][0][i] * ai; x1 += S.Jb[lane][1][i] * ai; x2 += S.Jb[lane][2][i] * ai;
        }
        float mu = S.cmu[lane];
        S.cjar[lane][0] = xn + mu * x1 - S.caref[lane][0];
        S.cjar[lane][1] = xn - mu * x1 - S.caref[lane][1];
        S.cjar[lane][2] = xn + mu * x2 - S.caref[lane][2];
        S.cjar[lane][3] = xn - mu * x2 - S.caref[lane][3];
      }
    }
    __syncthreads();
    int niter = 0;
    const float tol = m->tolerance, scale = m->solver_scale;
    // float32 rounding floor of the gradient Ma - qfrc_smooth - J^T f: below it a Newton step no
    // longer changes qacc, so iterating further is noise (same rule as the oracle, with float eps)
    const float gfloor = 16.0f * 5.96e-8f * sqrtf(gsum(isdof ? S.Ma[lane] * S.Ma[lane] + S.qfs[lane] * S.qfs[lane] : 0.0f));
    for (int it = 0; it < m->iterations; it++) {
      if (!__any(!done)) break;
      // forces and per-contact 3x3 weights
      if (lane < ncon) {
        float D = S.cD[lane], mu = S.cmu[lane];
        float f[4], act[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float x = S.cjar[lane][r];
          act[r] = x < 0.0f ? D : 0.0f;
          f[r] = -act[r] * x;
        }
        S.cfb[lane][0] = f[0] + f[1] + f[2] + f[3];
        S.cfb[lane][1] = mu * (f[0] - f[1]);
        S.cfb[lane][2] = mu * (f[2] - f[3]);
        S.cW[lane][0] = act[0] + act[1] + act[2] + act[3];      // nn
        S.cW[lane][1] = mu * (act[0] - act[1]);                  // n t1
        S.cW[lane][2] = mu * (act[2] - act[3]);                  // n t2
        S.cW[lane][3] = mu * mu * (act[0] + act[1]);             // t1 t1
        S.cW[lane][4] = mu * mu * (act[2] + act[3]);             // t2 t2
        S.cW[lane][5] = 0.0f;                                    // t1 t2
      }
      float lact = 0.0f;
      {
        float x = S.ljar[lane];
        lact = (lsg != 0.0f && x < 0.0f) ? S.lD[lane] : 0.0f;
        S.lf[lane] = -lact * x;
      }
      __syncthreads();
      // gradient and Hessian row (lane = dof)
      float g = 0.0f;
      float hrow[G];
#pragma unroll
      for (int j = 0; j < G; j++) hrow[j] = S.M[lane][j];
      if (isdof) {
        g = S.Ma[lane] - S.qfs[lane] - lsg * S.lf[lane];
#pragma unroll
        for (int j = 0; j < G; j++) hrow[j] += j == lane ? lact : 0.0f;  // limit rows are +-e_i: diagonal only
        for (int c = 0; c < ncon; c++) {
          float jn = S.Jb[c][0][lane], j1 = S.Jb[c][1][lane], j2 = S.Jb[c][2][lane];
          g -= jn * S.cfb[c][0] + j1 * S.cfb[c][1] + j2 * S.cfb[c][2];
          float w0 = S.cW[c][0], w1 = S.cW[c][1], w2 = S.cW[c][2], w3 = S.cW[c][3], w4 = S.cW[c][4];
          float tn = jn * w0 + j1 * w1 + j2 * w2;
          float t1 = jn * w1 + j1 * w3;
          float t2 = jn * w2 + j2 * w4;
          if (tn != 0.0f || t1 != 0.0f || t2 != 0.0f) {
#pragma unroll
            for (int j = 0; j < G; j++) hrow[j] += tn * S.Jb[c][0][j] + t1 * S.Jb[c][1][j] + t2 * S.Jb[c][2][j];
          }
        }
      }
      float gn = gsum(g * g);
      if (!done && (scale * sqrtf(gn) < tol || sqrtf(gn) < gfloor)) done = true;
#pragma unroll
      for (int j = 0; j < G; j++) S.H[lane][j] = hrow[j];
      S.grad[lane] = g;
      S.srch[lane] = g;
      __syncthreads();
      if (!__any(!done)) break;
      group_chol(S.H, nv, lane);
      group_cholsolve(S.H, nv, lane, S.srch);
      if (isdof) S.srch[lane] = -S.srch[lane];
      __syncthreads();
      // Mv, jv
      float mv = 0.0f, sv = isdof ? S.srch[lane] : 0.0f;
      if (isdof)
        for (int j = 0; j < nv; j++) mv += S.M[lane][j] * S.srch[j];
      S.Mv[lane] = mv;
      float ljv = lsg * sv;
      float jv[4] = {0, 0, 0, 0}, jr[4] = {0, 0, 0, 0}, cD = 0.0f;
      if (lane < ncon) {
        float xn = 0, x1 = 0, x2 = 0;
        for (int i = 0; i < nv; i++) {
          float si = S.srch[i];
          xn += S.Jb[lane][0][i] * si; x1 += S.Jb[lane][1][i] * si; x2 += S.Jb[lane][2][i] * si;
        }
        float mu = S.cmu[lane];
        jv[0] = xn + mu * x1; jv[1] = xn - mu * x1; jv[2] = xn + mu * x2; jv[3] = xn - mu * x2;
#pragma unroll
        for (int r = 0; r < 4; r++) jr[r] = S.cjar[lane][r];
        cD = S.cD[lane];
      }
      const float ljar = S.ljar[lane], lD = lsg != 0.0f ? S.lD[lane] : 0.0f;
      // exact line search: safeguarded Newton on phi'(alpha)
      float A = gsum(sv * mv), Bq = gsum(sv * (S.Ma[lane] - S.qfs[lane]));
      float alpha = 0.0f, lo = 0.0f, hi = -1.0f, g0 = 0.0f;
      bool lsdone = done;
      for (int ls = 0; ls < m->ls_iterations; ls++) {
        float pg = 0.0f, ph = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          float x = jr[r] + alpha * jv[r];
          if (x < 0.0f) { pg += cD * jv[r] * x; ph += cD * jv[r] * jv[r]; }
        }
        {
          float x = ljar + alpha * ljv;
          if (x < 0.0f) { pg += lD * ljv * x; ph += lD * ljv * ljv; }
        }
        float gg = gsum(pg) + alpha * A + Bq, hh = gsum(ph) + A;
        if (!lsdone) {
          if (ls == 0) { g0 = gg; if (g0 >= 0.0f) lsdone = true; }
          if (!lsdone && fabsf(gg) <= 1e-6f * fabsf(g0)) lsdone = true;
          if (!lsdone) {
            if (gg < 0.0f) lo = alpha; else hi = alpha;
            float an = alpha - gg / hh;
            if (hi >= 0.0f && (an <= lo || an >= hi)) an = 0.5f * (lo + hi);
            if (an == alpha) lsdone = true;
            alpha = an;
          }
        }
        if (!__any(!lsdone)) break;
      }
      // improvement from the 1-D model, then the update
      float pim = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        float x0 = jr[r], x1 = jr[r] + alpha * jv[r];
        pim -= (x1 < 0.0f ? 0.5f * cD * x1 * x1 : 0.0f) - (x0 < 0.0f ? 0.5f * cD * x0 * x0 : 0.0f);
      }
      {
        float x0 = ljar, x1 = ljar + alpha * ljv;
        pim -= (x1 < 0.0f ? 0.5f * lD * x1 * x1 : 0.0f) - (x0 < 0.0f ? 0.5f * lD * x0 * x0 : 0.0f);
      }
      float improvement = gsum(pim) - (0.5f * alpha * alpha * A + alpha * Bq);
      if (!done) {
        if (isdof) { S.qacc[lane] += alpha * sv; S.Ma[lane] += alpha * mv; }
        S.ljar[lane] = ljar + alpha * ljv;
        if (lane < ncon) {
#pragma unroll
          for (int r = 0; r < 4; r++) S.cjar[lane][r] = jr[r] + alpha * jv[r];
        }
        niter = it + 1;
        if (scale * improvement < tol) done = true;
      }
      __syncthreads();
    }
    if (a.out_qacc && valid && isdof && step == 0) a.out_qacc[(size_t)env * nv + lane] = S.qacc[lane];
    if (a.diag && valid && lane == 0) {
      a.diag[(size_t)env * 4 + 0] = ncon;
      a.diag[(size_t)env * 4 + 1] = nefc;
      a.diag[(size_t)env * 4 + 2] = niter;
      a.diag[(size_t)env * 4 + 3] = S.ncand;
    }
    if (a.mode != 0) break;

    // ======================= integrate ==============================================================
    __syncthreads();
    if (isdof) {
      float acc = S.qacc[lane];
      float qd = S.qvel[lane] + dt * acc;
      S.qvel[lane] = qd;
      S.qacc_ws[lane] = acc;
    }
    __syncthreads();
    if (isdof) {
      float qd = S.qvel[lane];
      if (d_kind < 2) S.qpos[d_qadr] += dt * qd;
      else if (d_kind == 2) S.qpos[m->b_qadr[d_body] + d_axis_k] += dt * qd;
      else if (d_axis_k == 0) {
        int da = m->b_dofadr[d_body], qa = m->b_qadr[d_body];
        V3 w = v3(S.qvel[da + 3], S.qvel[da + 4], S.qvel[da + 5]);
        float wn = sqrtf(dot(w, w));
        float ang = wn * dt;
        if (ang > 1e-15f) {
          float sn, cs;
          sincosf(0.5f * ang, &sn, &cs);
          V3 ax = (1.0f / wn) * w;
          Q4 dq = {cs, ax.x * sn, ax.y * sn, ax.z * sn};
          Q4 qn = qnormalize(qmul(dq, ld4(&S.qpos[qa + 3])));
          st4(&S.qpos[qa + 3], qn);
        }
      }
    }
    __syncthreads();
  }  // steps

  // ======================= final kinematics for observations =========================================
  if (a.mode != 1) group_fk(S, m, lane, nb, b_parent, b_jtype, b_qadr, b_pos, b_quat, b_axis);
  if (!valid) return;
  // ---- store state ---------------------------------------------------------------------------------
  if (a.mode == 0) {
    for (int i = lane; i < qst; i += G) a.qpos[(size_t)env * qst + i] = S.qpos[i];
    a.qvel[(size_t)env * G + lane] = S.qvel[lane];
    a.qacc_ws[(size_t)env * G + lane] = S.qacc_ws[lane];
  }
  // ---- observations (get_obs / compute_reward / terminated) ---------------------------------------
  const int eb = m->eef_body, ob = m->obj_body;
  if (a.agent_pos) {
    const int ad = 7 + m->n_grip;
    if (lane < ad) {
      float v;
      if (lane < 3) v = S.xpos[eb][lane];
      else if (lane < 7) v = S.xquat[eb][lane - 3];
      else v = S.qpos[m->grip_qadr[lane - 7]];
      a.agent_pos[(size_t)env * ad + lane] = v;
    }
  }
  if (a.env_state && lane < 11) {
    float v;
    V3 df = ld3(S.xpos[eb]) - ld3(S.xpos[ob]);
    if (lane < 3) v = S.xpos[ob][lane];
    else if (lane < 7) v = S.xquat[ob][lane - 3];
    else if (lane < 10) v = lane == 7 ? df.x : (lane == 8 ? df.y : df.z);
    else v = sqrtf(dot(df, df));
    a.env_state[(size_t)env * 11 + lane] = v;
  }
  if (lane == 0) {
    float r = S.xpos[ob][2] > m->reward_z ? 1.0f : 0.0f;
    if (a.reward) a.reward[env] = r;
    if (a.terminated) a.terminated[env] = r == 1.0f ? 1 : 0;
  }
  // packed row per env for the sharded gather: [agent_pos | env_state | reward | terminated], float32
  if (a.rows) {
    const int ad = 7 + m->n_grip;
    float* row = a.rows + (size_t)env * a.row_stride;
    V3 df = ld3(S.xpos[eb]) - ld3(S.xpos[ob]);
    for (int c = lane; c < ad + 13; c += G) {
      float v;
      if (c < 3) v = S.xpos[eb][c];
      else if (c < 7) v = S.xquat[eb][c - 3];
      else if (c < ad) v = S.qpos[m->grip_qadr[c - 7]];
      else {
        int k = c - ad;
        if (k < 3) v = S.xpos[ob][k];
        else if (k < 7) v = S.xquat[ob][k - 3];
        else if (k < 10) v = k == 7 ? df.x : (k == 8 ? df.y : df.z);
        else if (k == 10) v = sqrtf(dot(df, df));
        else v = S.xpos[ob][2] > m->reward_z ? 1.0f : 0.0f;  // k == 11 reward, k == 12 terminated
      }
      row[c] = v;
    }
  }
  if (a.out_xpos && lane < nb) {
    st3(&a.out_xpos[((size_t)env * nb + lane) * 3], ld3(S.xpos[lane]));
    st4(&a.out_xquat[((size_t)env * nb + lane) * 4], ld4(S.xquat[lane]));
  }
}

}  // namespace

// launcher used by the C ABI (mir_api.cpp)
extern "C" int mir_launch_step(const StepArgs* args, int max_contacts_lds, hipStream_t stream) {
  StepArgs a = *args;
  int blocks = (a.B + EPB - 1) / EPB;
  (void)max_contacts_lds;
  hipLaunchKernelGGL(HIP_KERNEL_NAME(mir_step_kernel<MIR_MAX_CONTACT>), dim3(blocks), dim3(64), 0, stream, a);
  return (int)hipGetLastError();
}

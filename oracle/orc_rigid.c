/*
 * orc_rigid.c — CPU ORACLE (test infrastructure only; PARITY UNPINNED — see
 * orc_rigid.h).  One rigid-body step in the MuJoCo-style formulation that the
 * reference's external physics engine implements:
 *
 *   fk            link poses, joint motion subspaces (c-frame), body inertias
 *   crb           composite-rigid-body mass matrix            (App. A.3 1.i)
 *   rne           Coriolis/centrifugal/gravity bias forces     (App. A.3 1.iv)
 *   smooth        PD actuation + passive damping, qacc_smooth  (App. A.3 1.iii,v)
 *   collide       plane-box/sphere/capsule, box-box, convex-convex (MPR) narrowphase   (App. A.3 2)
 *   make_rows     joint-limit + pyramidal contact rows, R, aref (App. A.3 2)
 *   newton        primal Newton with exact line search          (App. A.3 2)
 *   integrate     semi-implicit Euler, quaternion integration   (App. A.3 3)
 *
 * Call sites restated: scene.step() at
 * /root/reference/gym_genesis/tasks/franka/cube_pick.py:107,125.
 *
 * Deliberately written as dense, loop-per-body scalar C: it shares no code
 * with the HIP kernels (gym-genesis_amd/csrc), which are an independent
 * implementation of the same mathematics.
 */
#include "orc_rigid.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MINVAL ((real)1e-15)
#ifdef ORC_F32
#define REAL_EPS ((real)1.1920929e-7)
#define LS_RELTOL ((real)1e-4) /* x ls_tolerance 0.01 = 1e-6 |phi'(0)|: the float32 kernels' rule */
#else
#define REAL_EPS ((real)2.220446049250313e-16)
#ifdef ORC_NO_RULES /* (see below) */
#define LS_RELTOL ((real)1e-12)
#else
#define LS_RELTOL ((real)1e-6)
#endif
#endif
/* ORC_NO_RULES (liborc64_norules.so, `make -C oracle norules`; tests/test_oracle_rules.py): the float64 oracle WITHOUT the stopping
 * rules it shares with the float32 kernels by construction -- the rounding floor of the gradient, the stagnation rule built on it,
 * the rounding floor of the line search's phi' (from the fifth evaluation on) and the looser relative tolerance of the search.  In
 * float64 none of them should ever bind before the tolerance does: the test solves the same states with and without them and finds
 * the same acceleration. */
#define MINIMP ((real)0.0001)
#define MAXIMP ((real)0.9999)
#define F32(x) ((real)(float)(x)) /* the product stores model constants, dt and gravity in float32: same inputs */

#ifdef ORC_COUNT_FLOPS
thread_local OrcFlops orc_flops_tl;
#endif
int orc_flops_read(unsigned long long* out) {
#ifdef ORC_COUNT_FLOPS
  out[0] = orc_flops_tl.add; out[1] = orc_flops_tl.mul; out[2] = orc_flops_tl.div;
  out[3] = orc_flops_tl.sqrt_; out[4] = orc_flops_tl.trans; out[5] = orc_flops_tl.cmp;
  return 1;
#else
  for (int i = 0; i < 6; i++) out[i] = 0;
  return 0;
#endif
}
void orc_flops_reset(void) {
#ifdef ORC_COUNT_FLOPS
  orc_flops_tl = OrcFlops();
#endif
}

/* ------------------------------------------------------------------ small math */
static void v3set(real* o, real a, real b, real c) { o[0] = a; o[1] = b; o[2] = c; }
static void v3copy(real* o, const real* a) { o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; }
static real v3dot(const real* a, const real* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void v3cross(real* o, const real* a, const real* b) {
  real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  o[0] = x; o[1] = y; o[2] = z;
}
static void v3addscl(real* o, const real* a, const real* b, real s) { /* o = a + s b */
  o[0] = a[0] + s * b[0]; o[1] = a[1] + s * b[1]; o[2] = a[2] + s * b[2];
}
static void v3sub(real* o, const real* a, const real* b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static real v3norm(const real* a) { return (real)sqrt((double)v3dot(a, a)); }

static void qmul(real* o, const real* a, const real* b) { /* wxyz */
  real w = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  real x = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  real y = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  real z = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
  o[0] = w; o[1] = x; o[2] = y; o[3] = z;
}
static void qnormalize(real* q) {
  real n = (real)sqrt((double)(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]));
  if (n < MINVAL) { q[0] = 1; q[1] = q[2] = q[3] = 0; return; }
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}
static void q2mat(real* R, const real* q) { /* row-major 3x3 */
  real w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z);     R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y);     R[7] = 2 * (y * z + w * x);     R[8] = 1 - 2 * (x * x + y * y);
}
static void matvec3(real* o, const real* R, const real* v) {
  real a = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  real b = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
  real c = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
  o[0] = a; o[1] = b; o[2] = c;
}
static void axisangle2quat(real* q, const real* axis, real angle) {
  real s = (real)sin((double)angle * 0.5);
  q[0] = (real)cos((double)angle * 0.5); q[1] = axis[0] * s; q[2] = axis[1] * s; q[3] = axis[2] * s;
}

/* spatial (c-frame) helpers: motion [w; v], force [t; f], inertia {m, h[3], I[6]=xx yy zz xy xz yz} */
static void inert_mul(real* out, const real* I, const real* mv) {
  const real m = I[0]; const real* h = I + 1; const real* S = I + 4;
  const real* w = mv; const real* v = mv + 3;
  real Iw[3] = {S[0] * w[0] + S[3] * w[1] + S[4] * w[2], S[3] * w[0] + S[1] * w[1] + S[5] * w[2],
                S[4] * w[0] + S[5] * w[1] + S[2] * w[2]};
  real hxv[3], hxw[3];
  v3cross(hxv, h, v); v3cross(hxw, h, w);
  out[0] = Iw[0] + hxv[0]; out[1] = Iw[1] + hxv[1]; out[2] = Iw[2] + hxv[2];
  out[3] = m * v[0] - hxw[0]; out[4] = m * v[1] - hxw[1]; out[5] = m * v[2] - hxw[2];
}
static void cross_motion(real* o, const real* a, const real* b) { /* a x_m b */
  real t1[3], t2[3], t3[3];
  v3cross(t1, a, b); v3cross(t2, a, b + 3); v3cross(t3, a + 3, b);
  o[0] = t1[0]; o[1] = t1[1]; o[2] = t1[2];
  o[3] = t2[0] + t3[0]; o[4] = t2[1] + t3[1]; o[5] = t2[2] + t3[2];
}
static void cross_force(real* o, const real* a, const real* f) { /* a x_f f */
  real t1[3], t2[3], t3[3];
  v3cross(t1, a, f); v3cross(t2, a + 3, f + 3); v3cross(t3, a, f + 3);
  o[0] = t1[0] + t2[0]; o[1] = t1[1] + t2[1]; o[2] = t1[2] + t2[2];
  o[3] = t3[0]; o[4] = t3[1]; o[5] = t3[2];
}
static real dot6(const real* a, const real* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

/* dense Cholesky A = L L^T (lower, in place in L); returns 0 on success */
static int chol(int n, const real A[ORC_NV][ORC_NV], real L[ORC_NV][ORC_NV]) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j <= i; j++) {
      real s = A[i][j];
      for (int k = 0; k < j; k++) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (s < MINVAL) s = MINVAL;
        L[i][i] = (real)sqrt((double)s);
      } else
        L[i][j] = s / L[j][j];
    }
  return 0;
}
static void chol_solve(int n, const real L[ORC_NV][ORC_NV], const real* b, real* x) {
  real y[ORC_NV];
  for (int i = 0; i < n; i++) {
    real s = b[i];
    for (int k = 0; k < i; k++) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
  for (int i = n - 1; i >= 0; i--) {
    real s = y[i];
    for (int k = i + 1; k < n; k++) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
}

/* ------------------------------------------------------------------ sizes */
int orc_sizeof_model(void) { return (int)sizeof(OrcModel); }
int orc_sizeof_data(void) { return (int)sizeof(OrcData); }
int orc_sizeof_real(void) { return (int)sizeof(real); }

/* ------------------------------------------------------------------ kinematics */
void orc_fk(const OrcModel* m, OrcData* d) {
  v3set(d->xpos[0], 0, 0, 0);
  d->xquat[0][0] = 1; d->xquat[0][1] = d->xquat[0][2] = d->xquat[0][3] = 0;
  q2mat(d->xmat[0], d->xquat[0]);
  v3set(d->xipos[0], 0, 0, 0);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->parent[b];
    if (m->jtype[b] == MIR_JNT_FREE) {
      const real* q = d->qpos + m->qadr[b];
      v3copy(d->xpos[b], q);
      d->xquat[b][0] = q[3]; d->xquat[b][1] = q[4]; d->xquat[b][2] = q[5]; d->xquat[b][3] = q[6];
      qnormalize(d->xquat[b]);
    } else {
      real off[3];
      matvec3(off, d->xmat[p], m->pos[b]);
      v3addscl(d->xpos[b], d->xpos[p], off, 1);
      qmul(d->xquat[b], d->xquat[p], m->quat[b]);
      if (m->jtype[b] == MIR_JNT_REVOLUTE) {
        real qj[4], t[4];
        axisangle2quat(qj, m->axis[b], d->qpos[m->qadr[b]]);
        qmul(t, d->xquat[b], qj);
        memcpy(d->xquat[b], t, sizeof t);
      } else if (m->jtype[b] == MIR_JNT_PRISMATIC) {
        real R[9], a[3];
        q2mat(R, d->xquat[b]);
        matvec3(a, R, m->axis[b]);
        v3addscl(d->xpos[b], d->xpos[b], a, d->qpos[m->qadr[b]]);
      }
    }
    q2mat(d->xmat[b], d->xquat[b]);
    real io[3];
    matvec3(io, d->xmat[b], m->ipos[b]);
    v3addscl(d->xipos[b], d->xpos[b], io, 1);
  }
  /* tree reference point: origin of the tree's root body */
  for (int b = 0; b < m->nbody; b++) v3copy(d->cref[b], d->xpos[m->root[b]]);
  /* motion subspaces */
  for (int b = 1; b < m->nbody; b++) {
    int da = m->dofadr[b];
    real r[3];
    v3sub(r, d->cref[b], d->xpos[b]);
    if (m->jtype[b] == MIR_JNT_REVOLUTE) {
      real a[3];
      matvec3(a, d->xmat[b], m->axis[b]);
      v3copy(d->cdof[da], a);
      v3cross(d->cdof[da] + 3, a, r);
    } else if (m->jtype[b] == MIR_JNT_PRISMATIC) {
      real a[3];
      matvec3(a, d->xmat[b], m->axis[b]);
      v3set(d->cdof[da], 0, 0, 0);
      v3copy(d->cdof[da] + 3, a);
    } else if (m->jtype[b] == MIR_JNT_FREE) {
      for (int k = 0; k < 3; k++) {
        real e[3] = {0, 0, 0};
        e[k] = 1;
        v3set(d->cdof[da + k], 0, 0, 0);
        v3copy(d->cdof[da + k] + 3, e);
        v3copy(d->cdof[da + 3 + k], e);
        v3cross(d->cdof[da + 3 + k] + 3, e, r);
      }
    }
  }
  /* body inertias about the tree reference point, world axes */
  for (int b = 1; b < m->nbody; b++) {
    const real* R = d->xmat[b];
    const real* ib = m->inertia[b];
    real Ib[9] = {ib[0], ib[3], ib[4], ib[3], ib[1], ib[5], ib[4], ib[5], ib[2]};
    real T[9], W[9];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) T[i * 3 + j] = R[i * 3 + 0] * Ib[0 * 3 + j] + R[i * 3 + 1] * Ib[1 * 3 + j] + R[i * 3 + 2] * Ib[2 * 3 + j];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) W[i * 3 + j] = T[i * 3 + 0] * R[j * 3 + 0] + T[i * 3 + 1] * R[j * 3 + 1] + T[i * 3 + 2] * R[j * 3 + 2];
    real r[3];
    v3sub(r, d->xipos[b], d->cref[b]);
    real ms = m->mass[b], rr = v3dot(r, r);
    real* c = d->cinert[b];
    c[0] = ms; c[1] = ms * r[0]; c[2] = ms * r[1]; c[3] = ms * r[2];
    c[4] = W[0] + ms * (rr - r[0] * r[0]);
    c[5] = W[4] + ms * (rr - r[1] * r[1]);
    c[6] = W[8] + ms * (rr - r[2] * r[2]);
    c[7] = W[1] - ms * r[0] * r[1];
    c[8] = W[2] - ms * r[0] * r[2];
    c[9] = W[5] - ms * r[1] * r[2];
  }
  memset(d->cinert[0], 0, sizeof d->cinert[0]);
}

/* ------------------------------------------------------------------ mass matrix (CRB) */
static void orc_crb(const OrcModel* m, OrcData* d) {
  memcpy(d->crb, d->cinert, sizeof d->crb);
  for (int b = m->nbody - 1; b >= 1; b--) {
    int p = m->parent[b];
    if (p > 0)
      for (int k = 0; k < 10; k++) d->crb[p][k] += d->crb[b][k];
  }
  memset(d->M, 0, sizeof d->M);
  for (int i = 0; i < m->nv; i++) {
    real buf[6];
    inert_mul(buf, d->crb[m->dof_body[i]], d->cdof[i]);
    for (int j = i; j >= 0; j = m->dof_parent[j]) {
      real v = dot6(d->cdof[j], buf);
      d->M[i][j] = v;
      d->M[j][i] = v;
    }
  }
  for (int i = 0; i < m->nv; i++) d->M[i][i] += m->armature[i];
}

/* ------------------------------------------------------------------ bias forces (RNE, qacc = 0) */
static void orc_rne(const OrcModel* m, OrcData* d) {
  memset(d->cvel[0], 0, sizeof d->cvel[0]);
  d->cacc[0][0] = d->cacc[0][1] = d->cacc[0][2] = 0;
  d->cacc[0][3] = -F32(m->opt.gravity[0]);
  d->cacc[0][4] = -F32(m->opt.gravity[1]);
  d->cacc[0][5] = -F32(m->opt.gravity[2]);
  for (int b = 1; b < m->nbody; b++) {
    int p = m->parent[b], da = m->dofadr[b];
    real* cv = d->cvel[b];
    real* ca = d->cacc[b];
    memcpy(cv, d->cvel[p], 6 * sizeof(real));
    memcpy(ca, d->cacc[p], 6 * sizeof(real));
    if (m->jtype[b] == MIR_JNT_FREE) {
      for (int k = 0; k < 3; k++)
        for (int c = 0; c < 6; c++) cv[c] += d->cdof[da + k][c] * d->qvel[da + k];
      real dot[3][6];
      for (int k = 0; k < 3; k++) cross_motion(dot[k], cv, d->cdof[da + 3 + k]);
      for (int k = 0; k < 3; k++)
        for (int c = 0; c < 6; c++) {
          ca[c] += dot[k][c] * d->qvel[da + 3 + k];
          cv[c] += d->cdof[da + 3 + k][c] * d->qvel[da + 3 + k];
        }
    } else if (m->ndof[b] == 1) {
      real dot[6];
      cross_motion(dot, cv, d->cdof[da]);
      for (int c = 0; c < 6; c++) {
        ca[c] += dot[c] * d->qvel[da];
        cv[c] += d->cdof[da][c] * d->qvel[da];
      }
    }
  }
  for (int b = 1; b < m->nbody; b++) {
    real Ia[6], Iv[6], x[6];
    inert_mul(Ia, d->cinert[b], d->cacc[b]);
    inert_mul(Iv, d->cinert[b], d->cvel[b]);
    cross_force(x, d->cvel[b], Iv);
    for (int c = 0; c < 6; c++) d->cfrc[b][c] = Ia[c] + x[c];
  }
  memset(d->cfrc[0], 0, sizeof d->cfrc[0]);
  for (int b = m->nbody - 1; b >= 1; b--) {
    int p = m->parent[b];
    if (p > 0)
      for (int c = 0; c < 6; c++) d->cfrc[p][c] += d->cfrc[b][c];
  }
  for (int i = 0; i < m->nv; i++) d->qfrc_bias[i] = dot6(d->cdof[i], d->cfrc[m->dof_body[i]]);
}

/* ------------------------------------------------------------------ smooth dynamics */
static void orc_smooth(const OrcModel* m, OrcData* d) {
  const real dt = F32(m->opt.dt);
  for (int i = 0; i < m->nv; i++) {
    d->qfrc_passive[i] = -m->damping[i] * d->qvel[i];
    real f = 0;
    if (m->dof_ctrl[i] == MIR_CTRL_POSITION) {
      f = m->kp[i] * (d->target[i] - d->qpos[m->dof_qadr[i]]) - m->kv[i] * d->qvel[i];
      if (f < m->frc[i][0]) f = m->frc[i][0];
      if (f > m->frc[i][1]) f = m->frc[i][1];
    }
    d->qfrc_act[i] = f;
    d->qfrc_smooth[i] = d->qfrc_passive[i] + f - d->qfrc_bias[i];
  }
  memcpy(d->Mt, d->M, sizeof d->Mt);
  if (m->opt.implicit_damping)
    for (int i = 0; i < m->nv; i++)
      d->Mt[i][i] += dt * (m->damping[i] + (m->dof_ctrl[i] == MIR_CTRL_POSITION ? m->kv[i] : 0));
  real L[ORC_NV][ORC_NV];
  chol(m->nv, d->Mt, L);
  chol_solve(m->nv, L, d->qfrc_smooth, d->qacc_smooth);
}

/* ------------------------------------------------------------------ narrowphase */
typedef struct { real pos[3]; real dist; real u, v; } CPoint;

/* manifold reduction: keep the support extremes (+u, -u, +v, -v in reference-face
 * coordinates) of the candidate points, first index wins ties, duplicates merged;
 * output keeps candidate order. */
static int reduce4(CPoint* pts, int n) {
  if (n <= 4) return n;
  int pick[4] = {0, 0, 0, 0};
  for (int i = 1; i < n; i++) {
    if (pts[i].u > pts[pick[0]].u) pick[0] = i;
    if (pts[i].u < pts[pick[1]].u) pick[1] = i;
    if (pts[i].v > pts[pick[2]].v) pick[2] = i;
    if (pts[i].v < pts[pick[3]].v) pick[3] = i;
  }
  CPoint out[4];
  int k = 0;
  for (int i = 0; i < n; i++)
    if (i == pick[0] || i == pick[1] || i == pick[2] || i == pick[3]) out[k++] = pts[i];
  memcpy(pts, out, k * sizeof(CPoint));
  return k;
}

/* plane (point pp, normal n = plane z axis) vs box: penetrating corners, contact at half depth */
static int plane_box(const real* pp, const real* Rp, const real* pb, const real* Rb, const real* hb, CPoint* pts, real* n) {
  n[0] = Rp[2]; n[1] = Rp[5]; n[2] = Rp[8];
  int cnt = 0;
  CPoint all[8];
  for (int c = 0; c < 8; c++) {
    real l[3] = {(c & 1) ? hb[0] : -hb[0], (c & 2) ? hb[1] : -hb[1], (c & 4) ? hb[2] : -hb[2]};
    real w[3], rel[3];
    matvec3(w, Rb, l);
    v3addscl(w, w, pb, 1);
    v3sub(rel, w, pp);
    real dist = v3dot(rel, n);
    if (dist < 0) {
      v3addscl(all[cnt].pos, w, n, -dist * (real)0.5);
      all[cnt].dist = dist;
      all[cnt].u = rel[0] * Rp[0] + rel[1] * Rp[3] + rel[2] * Rp[6];
      all[cnt].v = rel[0] * Rp[1] + rel[1] * Rp[4] + rel[2] * Rp[7];
      cnt++;
    }
  }
  cnt = reduce4(all, cnt);
  memcpy(pts, all, cnt * sizeof(CPoint));
  return cnt;
}

/* plane - hull: the penetrating vertices in index order, reduced to four like plane - box (support extremes) */
static int plane_hull(const real* pp, const real* Rp, const real* pb, const real* Rb, const real (*verts)[3], int nvert, CPoint* pts, real* n) {
  n[0] = Rp[2]; n[1] = Rp[5]; n[2] = Rp[8];
  int cnt = 0;
  CPoint all[MIR_MAX_HULL_VERT];
  for (int c = 0; c < nvert; c++) {
    real w[3], rel[3];
    matvec3(w, Rb, verts[c]);
    v3addscl(w, w, pb, 1);
    v3sub(rel, w, pp);
    real dist = v3dot(rel, n);
    if (dist < 0) {
      v3addscl(all[cnt].pos, w, n, -dist * (real)0.5);
      all[cnt].dist = dist;
      all[cnt].u = rel[0] * Rp[0] + rel[1] * Rp[3] + rel[2] * Rp[6];
      all[cnt].v = rel[0] * Rp[1] + rel[1] * Rp[4] + rel[2] * Rp[7];
      cnt++;
    }
  }
  cnt = reduce4(all, cnt);
  memcpy(pts, all, cnt * sizeof(CPoint));
  return cnt;
}

static void col(real* o, const real* R, int k) { o[0] = R[k]; o[1] = R[3 + k]; o[2] = R[6 + k]; }

/* box-box by separating axes + reference-face clipping; normal from A to B */
static int box_box(const real* pa, const real* Ra, const real* ha, const real* pb, const real* Rb, const real* hb, CPoint* pts, real* nout) {
  real A[3][3], Bx[3][3], t[3];
  for (int k = 0; k < 3; k++) { col(A[k], Ra, k); col(Bx[k], Rb, k); }
  v3sub(t, pb, pa);
  real best = -(real)1e30; int code = -1; real bestL[3] = {0, 0, 0};
  /* face axes of A (0-2) and B (3-5) */
  for (int c = 0; c < 6; c++) {
    const real* L = c < 3 ? A[c] : Bx[c - 3];
    real ra = 0, rb = 0;
    for (int k = 0; k < 3; k++) { ra += ha[k] * (real)fabs((double)v3dot(A[k], L)); rb += hb[k] * (real)fabs((double)v3dot(Bx[k], L)); }
    real s = (real)fabs((double)v3dot(t, L)) - (ra + rb);
    if (s > 0) return 0;
    if (s > best) { best = s; code = c; v3copy(bestL, L); }
  }
  /* edge axes */
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      real L[3];
      v3cross(L, A[i], Bx[j]);
      real len = v3norm(L);
      if (len < (real)1e-3) continue;
      L[0] /= len; L[1] /= len; L[2] /= len;
      real ra = 0, rb = 0;
      for (int k = 0; k < 3; k++) { ra += ha[k] * (real)fabs((double)v3dot(A[k], L)); rb += hb[k] * (real)fabs((double)v3dot(Bx[k], L)); }
      real s = (real)fabs((double)v3dot(t, L)) - (ra + rb);
      if (s > 0) return 0;
      if (s * (real)1.05 > best && s > best + (real)1e-6) { best = s; code = 6 + i * 3 + j; v3copy(bestL, L); }
    }
  real n[3];
  v3copy(n, bestL);
  if (v3dot(t, n) < 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
  v3copy(nout, n);
  if (code >= 6) {
    int i = (code - 6) / 3, j = (code - 6) % 3;
    real PA[3], PB[3];
    v3copy(PA, pa); v3copy(PB, pb);
    for (int k = 0; k < 3; k++) {
      if (k != i) v3addscl(PA, PA, A[k], (v3dot(n, A[k]) > 0 ? ha[k] : -ha[k]));
      if (k != j) v3addscl(PB, PB, Bx[k], (v3dot(n, Bx[k]) > 0 ? -hb[k] : hb[k]));
    }
    real dd[3];
    v3sub(dd, PB, PA);
    real uaub = v3dot(A[i], Bx[j]), q1 = v3dot(A[i], dd), q2 = -v3dot(Bx[j], dd);
    real den = 1 - uaub * uaub;
    real alpha = 0, beta = 0;
    if (den > (real)1e-6) { alpha = (q1 + uaub * q2) / den; beta = (uaub * q1 + q2) / den; }
    v3addscl(PA, PA, A[i], alpha);
    v3addscl(PB, PB, Bx[j], beta);
    for (int k = 0; k < 3; k++) pts[0].pos[k] = (real)0.5 * (PA[k] + PB[k]);
    pts[0].dist = best;
    return 1;
  }
  /* face contact: reference box owns the axis */
  const real *pr, *hr, *pi, *hi;
  real (*Ar)[3], (*Ai)[3];
  real nr[3];
  int k;
  if (code < 3) { pr = pa; hr = ha; Ar = A; pi = pb; hi = hb; Ai = Bx; k = code; v3copy(nr, n); }
  else { pr = pb; hr = hb; Ar = Bx; pi = pa; hi = ha; Ai = A; k = code - 3; v3set(nr, -n[0], -n[1], -n[2]); }
  int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
  real fc[3];
  v3addscl(fc, pr, nr, hr[k]); /* nr = +-axis_k exactly */
  /* incident face */
  int jb = 0; real mx = -1;
  for (int j = 0; j < 3; j++) { real a = (real)fabs((double)v3dot(nr, Ai[j])); if (a > mx) { mx = a; jb = j; } }
  real sj = v3dot(nr, Ai[jb]) > 0 ? -(real)1 : (real)1;
  int j1 = (jb + 1) % 3, j2 = (jb + 2) % 3;
  real ic[3];
  v3addscl(ic, pi, Ai[jb], sj * hi[jb]);
  /* polygon in reference-face coords (x along k1, y along k2, z along nr) */
  real poly[16][3], tmp[16][3];
  int np = 4;
  static const real sx[4] = {1, -1, -1, 1}, sy[4] = {1, 1, -1, -1};
  for (int v = 0; v < 4; v++) {
    real w[3], rel[3];
    v3addscl(w, ic, Ai[j1], sx[v] * hi[j1]);
    v3addscl(w, w, Ai[j2], sy[v] * hi[j2]);
    v3sub(rel, w, fc);
    poly[v][0] = v3dot(rel, Ar[k1]); poly[v][1] = v3dot(rel, Ar[k2]); poly[v][2] = v3dot(rel, nr);
  }
  for (int e = 0; e < 4; e++) { /* clip against x<=h1, -x<=h1, y<=h2, -y<=h2 */
    int ax = e >> 1; real sg = (e & 1) ? -(real)1 : (real)1; real lim = ax == 0 ? hr[k1] : hr[k2];
    int nn = 0;
    for (int v = 0; v < np; v++) {
      const real* P = poly[v]; const real* Q = poly[(v + 1) % np];
      real dp = sg * P[ax] - lim, dq = sg * Q[ax] - lim;
      if (dp <= 0) { memcpy(tmp[nn++], P, 3 * sizeof(real)); }
      if ((dp <= 0) != (dq <= 0)) {
        real u = dp / (dp - dq);
        for (int c = 0; c < 3; c++) tmp[nn][c] = P[c] + u * (Q[c] - P[c]);
        nn++;
      }
      if (nn >= 15) break;
    }
    np = nn;
    memcpy(poly, tmp, sizeof(real) * 3 * np);
    if (np == 0) return 0;
  }
  CPoint all[16];
  int cnt = 0;
  for (int v = 0; v < np; v++) {
    if (poly[v][2] < 0) {
      real w[3];
      v3addscl(w, fc, Ar[k1], poly[v][0]);
      v3addscl(w, w, Ar[k2], poly[v][1]);
      v3addscl(w, w, nr, poly[v][2] * (real)0.5); /* midway between the surfaces */
      v3copy(all[cnt].pos, w);
      all[cnt].dist = poly[v][2];
      all[cnt].u = poly[v][0];
      all[cnt].v = poly[v][1];
      cnt++;
    }
  }
  if (cnt > 8) cnt = 8; /* a quad clipped by 4 half-planes has <= 8 vertices */
  memcpy(pts, all, cnt * sizeof(CPoint));
  return cnt;
}

/* ------------------------------------------------------------------ plane vs sphere / capsule (closed form) */
/* plane (point pp, normal = plane z axis) vs sphere (centre ps, radius r): one point at half depth */
static int plane_sphere(const real* pp, const real* Rp, const real* ps, real r, CPoint* pts, real* n) {
  n[0] = Rp[2]; n[1] = Rp[5]; n[2] = Rp[8];
  real rel[3];
  v3sub(rel, ps, pp);
  real dist = v3dot(rel, n) - r;
  if (!(dist < 0)) return 0;
  v3addscl(pts[0].pos, ps, n, -(r + dist * (real)0.5));
  pts[0].dist = dist; pts[0].u = pts[0].v = 0;
  return 1;
}

/* plane vs capsule (centre pc, axis = z of Rc, radius r, half length h): the two end spheres, axis - then axis + */
static int plane_capsule(const real* pp, const real* Rp, const real* pc, const real* Rc, real r, real h, CPoint* pts, real* n) {
  real ax[3] = {Rc[2], Rc[5], Rc[8]};
  int cnt = 0;
  for (int s = -1; s <= 1; s += 2) {
    real end[3], nn[3];
    v3addscl(end, pc, ax, (real)s * h);
    cnt += plane_sphere(pp, Rp, end, r, pts + cnt, nn);
  }
  n[0] = Rp[2]; n[1] = Rp[5]; n[2] = Rp[8];
  return cnt;
}

/* ------------------------------------------------------------------ convex-convex: Minkowski Portal Refinement
 * Restated from the published XenoCollide / libccd formulation (ccd/mpr.c: discover portal, refine portal, find
 * penetration, barycentric contact position), which is the algorithm behind Genesis's default convex-convex path
 * (UPSTREAM-RECALL).  Shapes enter only through their support mappings, so any convex geom type can
 * be added by giving it one.  Result: ONE contact (deepest penetration): normal from shape A to shape B, depth > 0,
 * position between the two witness points.  Same decisions, in the same order, as the kernel's lane-private version
 * (mir_dev.h: mpr_pair). */
#define MPR_TOL ((real)1e-6)
#define MPR_MAXIT 64
typedef struct { int type; real size[3], pos[3], R[9]; const real (*verts)[3]; int nvert; } Shape;
/* vertex pool the MIR_GEOM_HULL shapes of the current call index into (set by orc_collide from the model, by orc_set_hull_pool for
 * the narrowphase test hook) */
static ORC_TLS const real (*cur_verts)[3] = 0;

/* hull: the vertex of largest projection on d (geom frame: d_l = R^T d), lowest index on ties; out in world coordinates */
static void hull_support(const Shape* s, const real* d, real* out) {
  real dl[3] = {s->R[0] * d[0] + s->R[3] * d[1] + s->R[6] * d[2], s->R[1] * d[0] + s->R[4] * d[1] + s->R[7] * d[2],
                s->R[2] * d[0] + s->R[5] * d[1] + s->R[8] * d[2]};
  int best = 0;
  real bv = v3dot(s->verts[0], dl);
  for (int i = 1; i < s->nvert; i++) {
    real v = v3dot(s->verts[i], dl);
    if (v > bv) { bv = v; best = i; }
  }
  real w[3];
  matvec3(w, s->R, s->verts[best]);
  v3addscl(out, s->pos, w, 1);
}

/* farthest point of the shape along the UNIT direction d (world frame) */
static void shape_support(const Shape* s, const real* d, real* out) {
  if (s->type == MIR_GEOM_HULL) { hull_support(s, d, out); return; }
  v3copy(out, s->pos);
  if (s->type == MIR_GEOM_SPHERE) {
    v3addscl(out, out, d, s->size[0]);
  } else if (s->type == MIR_GEOM_CAPSULE) {
    real ax[3] = {s->R[2], s->R[5], s->R[8]};
    v3addscl(out, out, ax, v3dot(d, ax) >= 0 ? s->size[1] : -s->size[1]);
    v3addscl(out, out, d, s->size[0]);
  } else { /* box */
    for (int k = 0; k < 3; k++) {
      real ax[3] = {s->R[k], s->R[3 + k], s->R[6 + k]};
      v3addscl(out, out, ax, v3dot(d, ax) >= 0 ? s->size[k] : -s->size[k]);
    }
  }
}

typedef struct { real v[3], a[3], b[3]; } MprPt; /* point of A - B with its witnesses on A and on B */

static void mpr_support(const Shape* A, const Shape* B, const real* d, MprPt* o) {
  real nd[3] = {-d[0], -d[1], -d[2]};
  shape_support(A, d, o->a);
  shape_support(B, nd, o->b);
  v3sub(o->v, o->a, o->b);
}

static int v3unit(real* d) { /* normalise in place; 0 if (numerically) zero */
  real l2 = v3dot(d, d);
  if (!(l2 > (real)1e-30)) return 0;
  real il = (real)1 / (real)sqrt((double)l2);
  d[0] *= il; d[1] *= il; d[2] *= il;
  return 1;
}

static void portal_dir(const MprPt* p, real* d) { /* normal of the triangle (v1, v2, v3) */
  real e1[3], e2[3];
  v3sub(e1, p[2].v, p[1].v);
  v3sub(e2, p[3].v, p[1].v);
  v3cross(d, e1, e2);
  v3unit(d);
}

static int portal_reach_tolerance(const MprPt* p, const MprPt* v4, const real* d) {
  real dv4 = v3dot(v4->v, d);
  real m = dv4 - v3dot(p[1].v, d), t = dv4 - v3dot(p[2].v, d), u = dv4 - v3dot(p[3].v, d);
  if (t < m) m = t;
  if (u < m) m = u;
  return m < MPR_TOL;
}

static void portal_expand(MprPt* p, const MprPt* v4) {
  real c[3];
  v3cross(c, v4->v, p[0].v);
  if (v3dot(p[1].v, c) > 0) {
    if (v3dot(p[2].v, c) > 0) p[1] = *v4; else p[3] = *v4;
  } else {
    if (v3dot(p[3].v, c) > 0) p[2] = *v4; else p[1] = *v4;
  }
}

/* closest point of the triangle (a, b, c) to the origin (Ericson, Real-Time Collision Detection 5.1.5) */
static void tri_closest_to_origin(const real* a, const real* b, const real* c, real* out) {
  real ab[3], ac[3];
  v3sub(ab, b, a); v3sub(ac, c, a);
  real d1 = -v3dot(ab, a), d2 = -v3dot(ac, a);
  if (d1 <= 0 && d2 <= 0) { v3copy(out, a); return; }
  real d3 = -v3dot(ab, b), d4 = -v3dot(ac, b);
  if (d3 >= 0 && d4 <= d3) { v3copy(out, b); return; }
  real vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { v3addscl(out, a, ab, d1 / (d1 - d3)); return; }
  real d5 = -v3dot(ab, c), d6 = -v3dot(ac, c);
  if (d6 >= 0 && d5 <= d6) { v3copy(out, c); return; }
  real vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { v3addscl(out, a, ac, d2 / (d2 - d6)); return; }
  real va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
    real bc[3];
    v3sub(bc, c, b);
    v3addscl(out, b, bc, (d4 - d3) / ((d4 - d3) + (d5 - d6)));
    return;
  }
  real den = (real)1 / (va + vb + vc);
  v3addscl(out, a, ab, vb * den);
  v3addscl(out, out, ac, vc * den);
}

static void mpr_find_pos(const MprPt* p, real* pos) {
  real d[3], c[3], b[4];
  portal_dir(p, d);
  v3cross(c, p[1].v, p[2].v); b[0] = v3dot(c, p[3].v);
  v3cross(c, p[3].v, p[2].v); b[1] = v3dot(c, p[0].v);
  v3cross(c, p[0].v, p[1].v); b[2] = v3dot(c, p[3].v);
  v3cross(c, p[2].v, p[1].v); b[3] = v3dot(c, p[0].v);
  real sum = b[0] + b[1] + b[2] + b[3];
  if (!(sum > 0)) {
    b[0] = 0;
    v3cross(c, p[2].v, p[3].v); b[1] = v3dot(c, d);
    v3cross(c, p[3].v, p[1].v); b[2] = v3dot(c, d);
    v3cross(c, p[1].v, p[2].v); b[3] = v3dot(c, d);
    sum = b[1] + b[2] + b[3];
  }
  real inv = (real)1 / sum, pa[3] = {0, 0, 0}, pb[3] = {0, 0, 0};
  for (int i = 0; i < 4; i++) {
    v3addscl(pa, pa, p[i].a, b[i]);
    v3addscl(pb, pb, p[i].b, b[i]);
  }
  for (int k = 0; k < 3; k++) pos[k] = (real)0.5 * inv * (pa[k] + pb[k]);
}

/* 1 = penetrating (depth, normal A->B, pos filled), 0 = separated */
static int mpr_pair(const Shape* A, const Shape* B, real* depth, real* normal, real* pos) {
  MprPt p[4], v4;
  real d[3], c[3];
  /* interior point of A - B: difference of the centres */
  v3copy(p[0].a, A->pos); v3copy(p[0].b, B->pos);
  v3sub(p[0].v, p[0].a, p[0].b);
  if (!(v3dot(p[0].v, p[0].v) > (real)1e-20)) { p[0].v[0] = (real)1e-5; p[0].v[1] = p[0].v[2] = 0; }
  v3set(d, -p[0].v[0], -p[0].v[1], -p[0].v[2]);
  v3unit(d);
  mpr_support(A, B, d, &p[1]);
  if (!(v3dot(p[1].v, d) > 0)) return 0;
  v3cross(d, p[0].v, p[1].v);
  if (!v3unit(d)) {
    /* the origin lies on the ray v0 -> v1: v1 itself is the penetration */
    real l = v3norm(p[1].v);
    *depth = l;
    if (l > (real)1e-15) { for (int k = 0; k < 3; k++) normal[k] = p[1].v[k] / l; }
    else { real l0 = v3norm(p[0].v); for (int k = 0; k < 3; k++) normal[k] = -p[0].v[k] / l0; }
    for (int k = 0; k < 3; k++) pos[k] = (real)0.5 * (p[1].a[k] + p[1].b[k]);
    return 1;
  }
  mpr_support(A, B, d, &p[2]);
  if (!(v3dot(p[2].v, d) > 0)) return 0;
  {
    real e1[3], e2[3];
    v3sub(e1, p[1].v, p[0].v); v3sub(e2, p[2].v, p[0].v);
    v3cross(d, e1, e2);
    v3unit(d);
    if (v3dot(d, p[0].v) > 0) { MprPt t = p[1]; p[1] = p[2]; p[2] = t; d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
  }
  for (int it = 0;; it++) { /* discover the portal */
    if (it > MPR_MAXIT) return 0;
    mpr_support(A, B, d, &p[3]);
    if (!(v3dot(p[3].v, d) > 0)) return 0;
    int again = 0;
    v3cross(c, p[1].v, p[3].v);
    if (v3dot(c, p[0].v) < 0) { p[2] = p[3]; again = 1; }
    if (!again) {
      v3cross(c, p[3].v, p[2].v);
      if (v3dot(c, p[0].v) < 0) { p[1] = p[3]; again = 1; }
    }
    if (!again) break;
    real e1[3], e2[3];
    v3sub(e1, p[1].v, p[0].v); v3sub(e2, p[2].v, p[0].v);
    v3cross(d, e1, e2);
    v3unit(d);
  }
  for (int it = 0;; it++) { /* refine until the portal is beyond the origin */
    portal_dir(p, d);
    if (v3dot(d, p[1].v) >= 0) break;
    mpr_support(A, B, d, &v4);
    if (!(v3dot(v4.v, d) >= 0) || portal_reach_tolerance(p, &v4, d) || it > MPR_MAXIT) return 0;
    portal_expand(p, &v4);
  }
  for (int it = 0;; it++) { /* push the portal to the surface of A - B */
    portal_dir(p, d);
    mpr_support(A, B, d, &v4);
    if (portal_reach_tolerance(p, &v4, d) || it > MPR_MAXIT) {
      real w[3];
      tri_closest_to_origin(p[1].v, p[2].v, p[3].v, w);
      real l = v3norm(w);
      *depth = l;
      if (l > (real)1e-15) { for (int k = 0; k < 3; k++) normal[k] = w[k] / l; }
      else v3copy(normal, d);
      mpr_find_pos(p, pos);
      return 1;
    }
    portal_expand(p, &v4);
  }
}

/* ------------------------------------------------------------------ convex-convex: GJK distance between the CORE shapes
 * Spheres and capsules are a point / a segment swept by a radius, so their contact follows from the DISTANCE between the cores
 * (point, segment, box: polytopes, for which GJK terminates on an exact feature pair): depth = r1 + r2 - distance, normal along
 * the connecting line, position midway between the two surfaces.  That is exact to rounding -- a support-mapping method run on
 * the round surfaces themselves resolves the normal only to sqrt(2 tol / R) -- and is the usual treatment of rounded convexes
 * (a collision margin around a core).  When the cores themselves overlap (penetration deeper than the radii) the pair goes to
 * MPR on the full shapes, above.  GJK as in Gilbert-Johnson-Keerthi 1988 / van den Bergen 1999: the simplex is reduced to the
 * smallest face that contains the closest point (Ericson, Real-Time Collision Detection 5.1), duplicates and lack of progress
 * end the iteration. */
#define GJK_MAXIT 32
static void core_support(const Shape* s, const real* d, real* out) { /* farthest point of the CORE along d (any length) */
  if (s->type == MIR_GEOM_HULL) { hull_support(s, d, out); return; } /* (a hull is its own core, radius 0) */
  v3copy(out, s->pos);
  if (s->type == MIR_GEOM_CAPSULE) {
    real ax[3] = {s->R[2], s->R[5], s->R[8]};
    v3addscl(out, out, ax, v3dot(d, ax) >= 0 ? s->size[1] : -s->size[1]);
  } else if (s->type == MIR_GEOM_BOX) {
    for (int k = 0; k < 3; k++) {
      real ax[3] = {s->R[k], s->R[3 + k], s->R[6 + k]};
      v3addscl(out, out, ax, v3dot(d, ax) >= 0 ? s->size[k] : -s->size[k]);
    }
  }
}
static real core_radius(const Shape* s) { return (s->type == MIR_GEOM_BOX || s->type == MIR_GEOM_HULL) ? (real)0 : s->size[0]; }

/* closest point of the segment / triangle to the origin as barycentric weights; vertices with weight 0 are dropped by the caller */
static void seg_bary(const real* a, const real* b, real* l) {
  real ab[3];
  v3sub(ab, b, a);
  real t = -v3dot(a, ab), den = v3dot(ab, ab);
  if (t <= 0 || !(den > 0)) { l[0] = 1; l[1] = 0; }
  else if (t >= den) { l[0] = 0; l[1] = 1; }
  else { l[1] = t / den; l[0] = 1 - l[1]; }
}
static void tri_bary(const real* a, const real* b, const real* c, real* l) {
  real ab[3], ac[3];
  v3sub(ab, b, a); v3sub(ac, c, a);
  real d1 = -v3dot(ab, a), d2 = -v3dot(ac, a);
  l[0] = l[1] = l[2] = 0;
  if (d1 <= 0 && d2 <= 0) { l[0] = 1; return; }
  real d3 = -v3dot(ab, b), d4 = -v3dot(ac, b);
  if (d3 >= 0 && d4 <= d3) { l[1] = 1; return; }
  real vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) { l[1] = d1 / (d1 - d3); l[0] = 1 - l[1]; return; }
  real d5 = -v3dot(ab, c), d6 = -v3dot(ac, c);
  if (d6 >= 0 && d5 <= d6) { l[2] = 1; return; }
  real vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) { l[2] = d2 / (d2 - d6); l[0] = 1 - l[2]; return; }
  real va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) { l[2] = (d4 - d3) / ((d4 - d3) + (d5 - d6)); l[1] = 1 - l[2]; return; }
  real den = (real)1 / (va + vb + vc);
  l[1] = vb * den; l[2] = vc * den; l[0] = 1 - l[1] - l[2];
}
static real bary_norm2(int n, real P[4][3], const int* idx, const real* l) {
  real v[3] = {0, 0, 0};
  for (int i = 0; i < n; i++) v3addscl(v, v, P[idx[i]], l[i]);
  return v3dot(v, v);
}

/* 0: cores apart, *dist > 0, pa / pb = closest points on core A / core B;  1: cores touch or overlap */
static int gjk_core_distance(const Shape* A, const Shape* B, real* dist, real* pa, real* pb) {
  real P[4][3], PA[4][3], PB[4][3], lam[4] = {1, 0, 0, 0}, v[3];
  int n = 0;
  v3sub(v, A->pos, B->pos);
  if (!(v3dot(v, v) > (real)1e-24)) v3set(v, 1, 0, 0);
  for (int it = 0; it < GJK_MAXIT; it++) {
    real d[3] = {-v[0], -v[1], -v[2]}, wa[3], wb[3], w[3];
    core_support(A, d, wa);
    core_support(B, v, wb);
    v3sub(w, wa, wb);
    real vv = v3dot(v, v);
    if (n > 0 && vv - v3dot(v, w) <= (real)1e-12 * vv) break; /* no point of A - B is closer along v: v is the closest point */
    int dup = 0;
    for (int i = 0; i < n; i++) { real e[3]; v3sub(e, w, P[i]); if (!(v3dot(e, e) > (real)1e-24)) dup = 1; }
    if (dup) break;
    v3copy(P[n], w); v3copy(PA[n], wa); v3copy(PB[n], wb); n++;
    /* closest point of the simplex to the origin; keep the smallest face that carries it */
    real l[4] = {1, 0, 0, 0};
    if (n == 2) seg_bary(P[0], P[1], l);
    else if (n == 3) tri_bary(P[0], P[1], P[2], l);
    else if (n == 4) {
      /* the origin is inside the tetrahedron iff it is on the inner side of all four faces; otherwise the closest of the
       * faces it is outside of (faces in the fixed order 012, 013, 023, 123) */
      static const int F[4][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 3, 1}, {1, 2, 3, 0}};
      real best = -1, bl[4] = {0, 0, 0, 0};
      int inside = 1;
      for (int f = 0; f < 4; f++) {
        real e1[3], e2[3], nn[3], eo[3];
        v3sub(e1, P[F[f][1]], P[F[f][0]]); v3sub(e2, P[F[f][2]], P[F[f][0]]);
        v3cross(nn, e1, e2);
        v3sub(eo, P[F[f][3]], P[F[f][0]]);
        real so = v3dot(nn, eo), s0 = -v3dot(nn, P[F[f][0]]); /* sides of the opposite vertex and of the origin */
        /* origin outside this face, or the tetrahedron is flatter than 1e-5 rad (four vertices of one box face): no inside */
        if (so * s0 < 0 || so * so <= (real)1e-10 * v3dot(nn, nn) * v3dot(eo, eo)) {
          inside = 0;
          real tl[3];
          tri_bary(P[F[f][0]], P[F[f][1]], P[F[f][2]], tl);
          real d2 = bary_norm2(3, P, F[f], tl);
          if (best < 0 || d2 < best) {
            best = d2;
            bl[0] = bl[1] = bl[2] = bl[3] = 0;
            for (int i = 0; i < 3; i++) bl[F[f][i]] = tl[i];
          }
        }
      }
      if (inside) return 1;
      for (int i = 0; i < 4; i++) l[i] = bl[i];
    }
    /* compact: drop the vertices with zero weight, recompute v */
    int m = 0;
    v3set(v, 0, 0, 0);
    for (int i = 0; i < n; i++)
      if (l[i] > 0) {
        if (m != i) { v3copy(P[m], P[i]); v3copy(PA[m], PA[i]); v3copy(PB[m], PB[i]); }
        lam[m] = l[i];
        v3addscl(v, v, P[m], l[i]);
        m++;
      }
    n = m;
    if (!(v3dot(v, v) > (real)1e-12)) return 1; /* the cores touch (closer than 1e-6 m: the connecting line carries no direction
                                                     * in float32, portal refinement on the full shapes takes over) */
  }
  /* (left through the no-progress / repeated-vertex exits with the origin numerically ON the simplex -- a core centre on a symmetry
   *  plane of the other core, e.g. a sphere dropped onto the middle of a box and sunk past its radius: that is an overlap too) */
  if (!(v3dot(v, v) > (real)1e-12)) return 1;
  v3set(pa, 0, 0, 0); v3set(pb, 0, 0, 0);
  for (int i = 0; i < n; i++) { v3addscl(pa, pa, PA[i], lam[i]); v3addscl(pb, pb, PB[i], lam[i]); }
  *dist = v3norm(v);
  return 0;
}

static int convex_pair(int t1, const real* s1, const real* p1, const real* R1, int t2, const real* s2, const real* p2, const real* R2,
                       CPoint* pts, real* n) {
  Shape A, B;
  A.type = t1; B.type = t2;
  for (int k = 0; k < 3; k++) { A.size[k] = s1[k]; B.size[k] = s2[k]; A.pos[k] = p1[k]; B.pos[k] = p2[k]; }
  memcpy(A.R, R1, sizeof A.R); memcpy(B.R, R2, sizeof B.R);
  A.verts = B.verts = 0; A.nvert = B.nvert = 0;
  if (t1 == MIR_GEOM_HULL) { A.verts = cur_verts + (int)s1[0]; A.nvert = (int)s1[1]; }
  if (t2 == MIR_GEOM_HULL) { B.verts = cur_verts + (int)s2[0]; B.nvert = (int)s2[1]; }
  real dist, pa[3], pb[3], ra = core_radius(&A), rb = core_radius(&B);
  if (!gjk_core_distance(&A, &B, &dist, pa, pb)) {
    if (!(dist < ra + rb)) return 0;
    for (int k = 0; k < 3; k++) n[k] = (pb[k] - pa[k]) / dist;
    /* midway between the surface points pa + ra n and pb - rb n */
    for (int k = 0; k < 3; k++) pts[0].pos[k] = (real)0.5 * (pa[k] + pb[k] + (ra - rb) * n[k]);
    pts[0].dist = dist - ra - rb; pts[0].u = pts[0].v = 0;
    return 1;
  }
  real depth; /* cores overlap: deep penetration, portal refinement on the full shapes */
  if (!mpr_pair(&A, &B, &depth, n, pts[0].pos)) return 0;
  pts[0].dist = -depth; pts[0].u = pts[0].v = 0;
  return 1;
}

static void make_frame(real* F, const real* n) { /* rows: normal, t1, t2 */
  real t1[3] = {0, 0, 0};
  if (fabs((double)n[1]) < 0.5) t1[1] = 1; else t1[2] = 1;
  real dp = v3dot(n, t1);
  v3addscl(t1, t1, n, -dp);
  real l = v3norm(t1);
  t1[0] /= l; t1[1] /= l; t1[2] /= l;
  real t2[3];
  v3cross(t2, n, t1);
  v3copy(F, n); v3copy(F + 3, t1); v3copy(F + 6, t2);
}

/* More candidate points than the scene's contact capacity: manifolds are THINNED before any pair loses all of its points
 * (the capacity is the 16 lanes of the pick kernel; Genesis itself keeps 100+ pairs, SURVEY.md App. A.1, so whatever is done
 * here is this backend's own definition).  While the total exceeds the capacity: among the pairs that hold the most points
 * (at least two), in pair order, the LAST TWO points of a pair are merged into their mean (position and depth) -- neighbours
 * on the contact polygon, so a four-corner face patch becomes a triangle that still surrounds the patch centre, then an edge --
 * one merge per pair and round, until the total fits.  No depth comparison is involved (a box lying flat has four equal
 * depths up to rounding).  Only when every pair is down to one point does the capacity cut the list, in pair order. */
static void thin_manifolds(int* cnt, CPoint (*pts)[16], int npairs, int maxc) {
  int total = 0;
  for (int i = 0; i < npairs; i++) total += cnt[i];
  while (total > maxc) {
    int mx = 0;
    for (int i = 0; i < npairs; i++) mx = cnt[i] > mx ? cnt[i] : mx;
    if (mx <= 1) break;
    int need = total - maxc;
    for (int i = 0; i < npairs && need > 0; i++) {
      if (cnt[i] != mx) continue;
      CPoint *a = &pts[i][mx - 2], *b = &pts[i][mx - 1];
      for (int k = 0; k < 3; k++) a->pos[k] = (real)0.5 * (a->pos[k] + b->pos[k]);
      a->dist = (real)0.5 * (a->dist + b->dist);
      cnt[i]--; total--; need--;
    }
  }
}

static void orc_collide(const OrcModel* m, OrcData* d) {
  d->ncon = 0;
  d->ncand = 0;
  if (!m->opt.enable_collision) return;
  int maxc = m->opt.max_contacts < ORC_NC ? m->opt.max_contacts : ORC_NC;
  /* pass 1: the narrowphase of every pair, in pair order */
  static ORC_TLS CPoint ppts[ORC_NP][16];
  static ORC_TLS real pn[ORC_NP][3];
  static ORC_TLS int pcnt[ORC_NP], ppair[ORC_NP];
  int np = 0;
  cur_verts = m->vert;
  for (int pidx = 0; pidx < m->npair; pidx++) {
    int g1 = m->pair_g1[pidx], g2 = m->pair_g2[pidx];
    int b1 = m->gbody[g1], b2 = m->gbody[g2];
    real p1[3], p2[3], R1[9], R2[9], q[4], off[3];
    matvec3(off, d->xmat[b1], m->gpos[g1]); v3addscl(p1, d->xpos[b1], off, 1);
    qmul(q, d->xquat[b1], m->gquat[g1]); q2mat(R1, q);
    matvec3(off, d->xmat[b2], m->gpos[g2]); v3addscl(p2, d->xpos[b2], off, 1);
    qmul(q, d->xquat[b2], m->gquat[g2]); q2mat(R2, q);
    CPoint* pts = ppts[np]; real* n = pn[np]; int cnt = 0;
    if (m->gtype[g1] == MIR_GEOM_PLANE && m->gtype[g2] == MIR_GEOM_BOX) cnt = plane_box(p1, R1, p2, R2, m->gsize[g2], pts, n);
    else if (m->gtype[g1] == MIR_GEOM_PLANE && m->gtype[g2] == MIR_GEOM_SPHERE) cnt = plane_sphere(p1, R1, p2, m->gsize[g2][0], pts, n);
    else if (m->gtype[g1] == MIR_GEOM_PLANE && m->gtype[g2] == MIR_GEOM_CAPSULE) cnt = plane_capsule(p1, R1, p2, R2, m->gsize[g2][0], m->gsize[g2][1], pts, n);
    else if (m->gtype[g1] == MIR_GEOM_PLANE && m->gtype[g2] == MIR_GEOM_HULL) cnt = plane_hull(p1, R1, p2, R2, m->vert + (int)m->gsize[g2][0], (int)m->gsize[g2][1], pts, n);
    else if (m->gtype[g1] == MIR_GEOM_BOX && m->gtype[g2] == MIR_GEOM_BOX) cnt = box_box(p1, R1, m->gsize[g1], p2, R2, m->gsize[g2], pts, n);
    else if (m->gtype[g1] != MIR_GEOM_PLANE) cnt = convex_pair(m->gtype[g1], m->gsize[g1], p1, R1, m->gtype[g2], m->gsize[g2], p2, R2, pts, n);
    if (cnt > 0) { pcnt[np] = cnt; ppair[np] = pidx; np++; }
  }
  /* pass 2: fit the capacity, then the contact arrays in pair order */
  d->ncand = 0;
  for (int i = 0; i < np; i++) d->ncand += pcnt[i];
  thin_manifolds(pcnt, ppts, np, maxc);
  for (int i = 0; i < np; i++) {
    int g1 = m->pair_g1[ppair[i]], g2 = m->pair_g2[ppair[i]];
    int b1 = m->gbody[g1], b2 = m->gbody[g2];
    for (int c = 0; c < pcnt[i] && d->ncon < maxc; c++) {
      int k = d->ncon++;
      v3copy(d->cpos[k], ppts[i][c].pos);
      d->cdist[k] = ppts[i][c].dist;
      make_frame(d->cframe[k], pn[i]);
      d->cmu[k] = m->gfriction[g1] > m->gfriction[g2] ? m->gfriction[g1] : m->gfriction[g2];
      for (int s = 0; s < 2; s++) d->csolref[k][s] = (real)0.5 * (m->gsolref[g1][s] + m->gsolref[g2][s]);
      for (int s = 0; s < 5; s++) d->csolimp[k][s] = (real)0.5 * (m->gsolimp[g1][s] + m->gsolimp[g2][s]);
      d->cb1[k] = b1; d->cb2[k] = b2; d->cg1[k] = g1; d->cg2[k] = g2;
    }
  }
}

/* test hook: the vertex pool that MIR_GEOM_HULL geoms handed to orc_narrowphase index into (size = first vertex, count) */
static ORC_TLS real hook_pool[MIR_MAX_VERT][3];
void orc_set_hull_pool(const double* verts, int n) {
  if (n > MIR_MAX_VERT) n = MIR_MAX_VERT;
  for (int i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) hook_pool[i][k] = (real)(float)verts[3 * i + k];
  cur_verts = hook_pool;
}

/* test hook: narrowphase of ONE pair of geoms given directly (types, sizes, world poses as pos3 + quat4 wxyz);
 * out = up to 8 x (pos3, dist), normal3; returns the number of points */
int orc_narrowphase(int t1, const double* size1, const double* pos1, const double* quat1, int t2, const double* size2, const double* pos2,
                    const double* quat2, double* out_pts, double* out_normal) {
  real s1[3], s2[3], p1[3], p2[3], q1[4], q2[4], R1[9], R2[9], n[3] = {0, 0, 0};
  for (int k = 0; k < 3; k++) { s1[k] = (real)size1[k]; s2[k] = (real)size2[k]; p1[k] = (real)pos1[k]; p2[k] = (real)pos2[k]; }
  for (int k = 0; k < 4; k++) { q1[k] = (real)quat1[k]; q2[k] = (real)quat2[k]; }
  qnormalize(q1); qnormalize(q2);
  q2mat(R1, q1); q2mat(R2, q2);
  CPoint pts[16];
  int cnt = 0;
  if (t1 == MIR_GEOM_PLANE && t2 == MIR_GEOM_BOX) cnt = plane_box(p1, R1, p2, R2, s2, pts, n);
  else if (t1 == MIR_GEOM_PLANE && t2 == MIR_GEOM_SPHERE) cnt = plane_sphere(p1, R1, p2, s2[0], pts, n);
  else if (t1 == MIR_GEOM_PLANE && t2 == MIR_GEOM_CAPSULE) cnt = plane_capsule(p1, R1, p2, R2, s2[0], s2[1], pts, n);
  else if (t1 == MIR_GEOM_PLANE && t2 == MIR_GEOM_HULL) cnt = plane_hull(p1, R1, p2, R2, cur_verts + (int)s2[0], (int)s2[1], pts, n);
  else if (t1 == MIR_GEOM_BOX && t2 == MIR_GEOM_BOX) cnt = box_box(p1, R1, s1, p2, R2, s2, pts, n);
  else if (t1 != MIR_GEOM_PLANE && t2 != MIR_GEOM_PLANE) cnt = convex_pair(t1, s1, p1, R1, t2, s2, p2, R2, pts, n);
  for (int c = 0; c < cnt; c++) {
    for (int k = 0; k < 3; k++) out_pts[4 * c + k] = (double)pts[c].pos[k];
    out_pts[4 * c + 3] = (double)pts[c].dist;
  }
  for (int k = 0; k < 3; k++) out_normal[k] = (double)n[k];
  return cnt;
}

/* ------------------------------------------------------------------ constraint rows */
static void imp_kb(const OrcModel* m, const real* solref, const real* solimp, real pos, real* imp, real* k, real* b) {
  real dmin = solimp[0], dmax = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  dmin = dmin < MINIMP ? MINIMP : (dmin > MAXIMP ? MAXIMP : dmin);
  dmax = dmax < MINIMP ? MINIMP : (dmax > MAXIMP ? MAXIMP : dmax);
  if (width < MINVAL) width = MINVAL;
  mid = mid < MINIMP ? MINIMP : (mid > MAXIMP ? MAXIMP : mid);
  if (power < 1) power = 1;
  real x = (real)fabs((double)pos) / width, y;
  if (x >= 1) y = 1;
  else if (x <= 0) y = 0;
  else if (x <= mid) y = (real)pow((double)(x / mid), (double)power) * mid;
  else y = 1 - (real)pow((double)((1 - x) / (1 - mid)), (double)power) * (1 - mid);
  *imp = dmin + y * (dmax - dmin);
  real tc = solref[0], dr = solref[1];
  if (tc < 2 * F32(m->opt.dt)) tc = 2 * F32(m->opt.dt);
  *k = 1 / (dmax * dmax * tc * tc * dr * dr);
  *b = 2 / (dmax * tc);
}

/* translational Jacobian row of point p on body b, projected on direction dir, added with weight w */
static void add_jac(const OrcModel* m, const OrcData* d, int b, const real* p, const real* dir, real w, real* row) {
  if (b == 0 || m->is_static[b]) return;
  real r[3];
  v3sub(r, p, d->cref[b]);
  while (b > 0) {
    for (int k = 0; k < m->ndof[b]; k++) {
      int i = m->dofadr[b] + k;
      real v[3];
      v3cross(v, d->cdof[i], r);
      v[0] += d->cdof[i][3]; v[1] += d->cdof[i][4]; v[2] += d->cdof[i][5];
      row[i] += w * v3dot(v, dir);
    }
    b = m->parent[b];
  }
}

static void orc_make_rows(const OrcModel* m, OrcData* d) {
  int n = 0;
  const int nv = m->nv;
  if (m->opt.enable_joint_limit)
    for (int i = 0; i < nv; i++) {
      if (!m->dof_limited[i]) continue;
      real q = d->qpos[m->dof_qadr[i]];
      real dlo = q - m->range[i][0], dhi = m->range[i][1] - q;
      real pos, sgn;
      if (dlo < 0) { pos = dlo; sgn = 1; }
      else if (dhi < 0) { pos = dhi; sgn = -1; }
      else continue;
      memset(d->J[n], 0, sizeof d->J[n]);
      d->J[n][i] = sgn;
      real imp, k, b;
      imp_kb(m, m->dsolref[i], m->dsolimp[i], pos, &imp, &k, &b);
      real R = (1 - imp) / imp * m->dof_invweight0[i];
      if (R < MINVAL) R = MINVAL;
      d->efcD[n] = 1 / R;
      d->aref[n] = -b * (sgn * d->qvel[i]) - k * imp * pos;
      d->efcpos[n] = pos;
      n++;
    }
  for (int c = 0; c < d->ncon; c++) {
    real mu = d->cmu[c];
    real imp, k, b;
    imp_kb(m, d->csolref[c], d->csolimp[c], d->cdist[c], &imp, &k, &b);
    real w = m->body_invweight0[d->cb1[c]] + m->body_invweight0[d->cb2[c]];
    real R = 2 * mu * mu * (1 - imp) / imp * w * (1 + mu * mu);
    if (R < MINVAL) R = MINVAL;
    const real* F = d->cframe[c];
    for (int r = 0; r < 4; r++) {
      real dir[3];
      const real* t = F + 3 * (1 + (r >> 1));
      real s = (r & 1) ? -mu : mu;
      for (int a = 0; a < 3; a++) dir[a] = F[a] + s * t[a];
      memset(d->J[n], 0, sizeof d->J[n]);
      add_jac(m, d, d->cb2[c], d->cpos[c], dir, 1, d->J[n]);
      add_jac(m, d, d->cb1[c], d->cpos[c], dir, -1, d->J[n]);
      real vel = 0;
      for (int i = 0; i < nv; i++) vel += d->J[n][i] * d->qvel[i];
      d->efcD[n] = 1 / R;
      d->aref[n] = -b * vel - k * imp * d->cdist[c];
      d->efcpos[n] = d->cdist[c];
      n++;
    }
  }
  d->nefc = n;
}

/* ------------------------------------------------------------------ primal Newton solver */
static real row_cost(const OrcData* d, int n, const real* jar) {
  real c = 0;
  for (int r = 0; r < n; r++)
    if (jar[r] < 0) c += (real)0.5 * d->efcD[r] * jar[r] * jar[r];
  return c;
}

static void orc_solve(const OrcModel* m, OrcData* d) {
  const int nv = m->nv, n = d->nefc;
  d->niter = 0;
  if (n == 0) {
    memcpy(d->qacc, d->qacc_smooth, sizeof d->qacc);
    return;
  }
  real jar[ORC_NEFC], jv[ORC_NEFC], Ma[ORC_NV], Mv[ORC_NV], grad[ORC_NV], s[ORC_NV];
  /* warm start selection */
  {
    real c_sm, c_ws = 0;
    real jw[ORC_NEFC];
    for (int r = 0; r < n; r++) {
      real a = -d->aref[r], b = -d->aref[r];
      for (int i = 0; i < nv; i++) { a += d->J[r][i] * d->qacc_smooth[i]; b += d->J[r][i] * d->qacc_ws[i]; }
      jar[r] = a; jw[r] = b;
    }
    c_sm = row_cost(d, n, jar);
    for (int i = 0; i < nv; i++) {
      real t = 0;
      for (int j = 0; j < nv; j++) t += d->Mt[i][j] * (d->qacc_ws[j] - d->qacc_smooth[j]);
      c_ws += (real)0.5 * t * (d->qacc_ws[i] - d->qacc_smooth[i]);
    }
    c_ws += row_cost(d, n, jw);
    if (c_ws < c_sm) { memcpy(d->qacc, d->qacc_ws, sizeof d->qacc); memcpy(jar, jw, sizeof jar); }
    else memcpy(d->qacc, d->qacc_smooth, sizeof d->qacc);
  }
  for (int i = 0; i < nv; i++) {
    real t = 0;
    for (int j = 0; j < nv; j++) t += d->Mt[i][j] * d->qacc[j];
    Ma[i] = t;
  }
  const real scale = 1 / (m->meaninertia * (real)(nv > 1 ? nv : 1));
  const real tol = (real)m->opt.tolerance;
  /* rounding floor of the gradient Ma - qfrc_smooth - J^T f in this precision: below it Newton
   * steps no longer change qacc, so stop (never binding in float64 before tol is) */
  real gfloor = 0;
  for (int i = 0; i < nv; i++) gfloor += Ma[i] * Ma[i] + d->qfrc_smooth[i] * d->qfrc_smooth[i];
  gfloor = 16 * (sizeof(real) == 4 ? (real)5.96e-8 : (real)1.11e-16) * (real)sqrt((double)gfloor);
#ifdef ORC_NO_RULES
  gfloor = 0;
#endif
  real gprev = 0;
  for (int it = 0; it < m->opt.iterations; it++) {
    /* gradient, forces, Hessian */
    real H[ORC_NV][ORC_NV], L[ORC_NV][ORC_NV];
    memcpy(H, d->Mt, sizeof H);
    for (int i = 0; i < nv; i++) grad[i] = Ma[i] - d->qfrc_smooth[i];
    for (int r = 0; r < n; r++) {
      if (jar[r] < 0) {
        real f = -d->efcD[r] * jar[r];
        d->efcforce[r] = f;
        for (int i = 0; i < nv; i++) {
          grad[i] -= d->J[r][i] * f;
          real di = d->efcD[r] * d->J[r][i];
          if (di != 0)
            for (int j = 0; j < nv; j++) H[i][j] += di * d->J[r][j];
        }
      } else
        d->efcforce[r] = 0;
    }
    real gn = 0;
    for (int i = 0; i < nv; i++) gn += grad[i] * grad[i];
    if (it < 64) d->dbg_gn[it] = scale * (real)sqrt((double)gn);
    if (scale * (real)sqrt((double)gn) < tol || (real)sqrt((double)gn) < gfloor) break;
    chol(nv, H, L);
    chol_solve(nv, L, grad, s);
    for (int i = 0; i < nv; i++) s[i] = -s[i];
    for (int i = 0; i < nv; i++) {
      real t = 0;
      for (int j = 0; j < nv; j++) t += d->Mt[i][j] * s[j];
      Mv[i] = t;
    }
    for (int r = 0; r < n; r++) {
      real t = 0;
      for (int i = 0; i < nv; i++) t += d->J[r][i] * s[i];
      jv[r] = t;
    }
    /* exact line search on the piecewise-quadratic phi(alpha): safeguarded Newton on phi' */
    real A = 0, Bq = 0;
    for (int i = 0; i < nv; i++) { A += s[i] * Mv[i]; Bq += s[i] * (Ma[i] - d->qfrc_smooth[i]); }
    /* phi'(0) = grad . s, and with the exact Hessian the first Newton iterate on phi' is alpha = 1: the search starts there,
     * bracket [0, ?), with phi'(0) as the scale of its stopping rule (ls counts evaluations of phi', the one at 0 included) */
    real g0 = 0;
    for (int i = 0; i < nv; i++) g0 += s[i] * grad[i];
    real alpha = g0 >= 0 ? 0 : 1, lo = 0, hi = -1;
    int nls = 1;
    for (int ls = 1; ls < m->opt.ls_iterations && g0 < 0; ls++) {
      nls = ls + 1;
      /* phi'(alpha), phi''(alpha), and the magnitude of the terms phi' is summed from: at the root they cancel, and
       * what is left is rounding noise of about an epsilon of that magnitude -- no evaluation can resolve phi' below it */
      real g = alpha * A + Bq, h = A, gabs = (real)fabs((double)(alpha * A)) + (real)fabs((double)Bq);
      for (int r = 0; r < n; r++) {
        real x = jar[r] + alpha * jv[r];
        if (x < 0) {
          g += d->efcD[r] * jv[r] * x; h += d->efcD[r] * jv[r] * jv[r];
          gabs += d->efcD[r] * (real)fabs((double)jv[r]) * ((real)fabs((double)jar[r]) + (real)fabs((double)(alpha * jv[r])));
        }
      }
      {
        real tolg = (real)m->opt.ls_tolerance * (real)fabs((double)g0) * LS_RELTOL, floorg = ls >= 4 ? 4 * REAL_EPS * gabs : 0; /* (from the fifth evaluation on, as the kernels) */
#ifdef ORC_NO_RULES
        floorg = 0;
#endif
        if ((real)fabs((double)g) <= (tolg > floorg ? tolg : floorg) + MINVAL) break;
      }
      if (g < 0) lo = alpha; else hi = alpha;
      real an = alpha - g / h;
      if (hi >= 0 && (an <= lo || an >= hi)) an = (real)0.5 * (lo + hi);
      if (an == alpha) break;
      alpha = an;
    }
    /* improvement from the 1-D model; row-cost differences as 1/2 D dx (2 x0 + dx), never as a
     * difference of squares (a step below the resolution of jar must give a tiny improvement) */
    real imp = -((real)0.5 * alpha * alpha * A + alpha * Bq);
    for (int r = 0; r < n; r++) {
      /* with a = min(x, 0) the cost of a row is 1/2 D a^2 and its change 1/2 D (a1 - a0)(a1 + a0); a1 - a0 = dx while it stays active */
      real x0 = jar[r], dx = alpha * jv[r], x1 = x0 + dx;
      real a0 = x0 < 0 ? x0 : 0, a1 = x1 < 0 ? x1 : 0;
      imp -= (real)0.5 * d->efcD[r] * ((x0 < 0 && x1 < 0) ? dx : a1 - a0) * (a1 + a0);
    }
    /* resolution of this precision: no dof moves, or the gradient stopped shrinking near its floor */
    {
      int moved = 0;
      for (int i = 0; i < nv; i++) moved |= (d->qacc[i] + alpha * s[i] != d->qacc[i]);
      real gnorm = (real)sqrt((double)gn);
      int stagnant = it > 0 && gnorm > (real)0.5 * gprev && gnorm < 4 * gfloor;
      gprev = gnorm;
      if (!moved || stagnant) { d->niter = it + 1; break; }
    }
    for (int i = 0; i < nv; i++) { d->qacc[i] += alpha * s[i]; Ma[i] += alpha * Mv[i]; }
    for (int r = 0; r < n; r++) jar[r] += alpha * jv[r];
    d->niter = it + 1;
    if (it < 64) { d->dbg_imp[it] = scale * imp; d->dbg_alpha[it] = alpha; d->dbg_ls[it] = (real)nls; }
    if (scale * imp < tol) break;
  }
  for (int r = 0; r < n; r++) d->efcforce[r] = jar[r] < 0 ? -d->efcD[r] * jar[r] : 0;
}

/* ------------------------------------------------------------------ pipeline */
void orc_forward(const OrcModel* m, OrcData* d) {
  orc_fk(m, d);
  orc_crb(m, d);
  orc_rne(m, d);
  orc_smooth(m, d);
  orc_collide(m, d);
  orc_make_rows(m, d);
  orc_solve(m, d);
}

static void orc_integrate(const OrcModel* m, OrcData* d) {
  const real dt = F32(m->opt.dt);
  for (int i = 0; i < m->nv; i++) d->qvel[i] += dt * d->qacc[i];
  for (int b = 1; b < m->nbody; b++) {
    int da = m->dofadr[b], qa = m->qadr[b];
    if (m->jtype[b] == MIR_JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] += dt * d->qvel[da + k];
      real w[3] = {d->qvel[da + 3], d->qvel[da + 4], d->qvel[da + 5]};
      real ang = v3norm(w) * dt;
      if (ang > MINVAL) {
        real ax[3] = {w[0] * dt / ang, w[1] * dt / ang, w[2] * dt / ang}, dq[4], t[4];
        axisangle2quat(dq, ax, ang);
        qmul(t, dq, d->qpos + qa + 3);
        qnormalize(t);
        memcpy(d->qpos + qa + 3, t, sizeof t);
      }
    } else if (m->ndof[b] == 1)
      d->qpos[qa] += dt * d->qvel[da];
  }
  memcpy(d->qacc_ws, d->qacc, sizeof d->qacc_ws);
}

void orc_step(const OrcModel* m, OrcData* d) {
  orc_forward(m, d);
  orc_integrate(m, d);
  orc_fk(m, d);
}

/* ------------------------------------------------------------------ model compile */
int orc_compile(const MirSceneSpec* sp, OrcModel* m) {
  memset(m, 0, sizeof *m);
  if (sp->struct_size != (int)sizeof(MirSceneSpec) || sp->version != MIR_VERSION) return MIR_E_INVALID;
  if (sp->nbody > ORC_NB || sp->ndof > ORC_NV || sp->ngeom > ORC_NG) return MIR_E_CAPACITY;
  m->nbody = sp->nbody; m->ngeom = sp->ngeom; m->opt = sp->opt; m->task = sp->task;
  int nv = 0, nq = 0;
  for (int b = 0; b < sp->nbody; b++) {
    const MirBodySpec* s = &sp->body[b];
    m->parent[b] = b == 0 ? -1 : s->parent;
    m->jtype[b] = b == 0 ? MIR_JNT_FIXED : s->jtype;
    for (int k = 0; k < 3; k++) { m->pos[b][k] = F32(s->pos[k]); m->axis[b][k] = F32(s->axis[k]); m->ipos[b][k] = F32(s->ipos[k]); }
    for (int k = 0; k < 4; k++) m->quat[b][k] = F32(s->quat[k]);
    for (int k = 0; k < 6; k++) m->inertia[b][k] = F32(s->inertia[k]);
    m->mass[b] = F32(s->mass);
    m->dofadr[b] = nv; m->qadr[b] = nq;
    int nd = 0, nqq = 0;
    if (m->jtype[b] == MIR_JNT_REVOLUTE || m->jtype[b] == MIR_JNT_PRISMATIC) { nd = 1; nqq = 1; }
    else if (m->jtype[b] == MIR_JNT_FREE) { nd = 6; nqq = 7; }
    m->ndof[b] = nd;
    if (b == 0) { m->root[b] = 0; m->is_static[b] = 1; }
    else {
      m->root[b] = s->parent == 0 ? b : m->root[s->parent];
      m->is_static[b] = (nd == 0) && m->is_static[s->parent];
    }
    for (int k = 0; k < nd; k++) {
      int i = nv + k;
      m->dof_body[i] = b;
      if (k > 0) m->dof_parent[i] = i - 1;
      else {
        int a = s->parent;
        while (a > 0 && m->ndof[a] == 0) a = m->parent[a];
        m->dof_parent[i] = a > 0 ? m->dofadr[a] + m->ndof[a] - 1 : -1;
      }
      m->dof_qadr[i] = nq + k; /* meaningful for scalar joints */
    }
    if (m->jtype[b] == MIR_JNT_FREE) {
      for (int k = 0; k < 3; k++) m->qpos0[nq + k] = F32(s->pos[k]);
      for (int k = 0; k < 4; k++) m->qpos0[nq + 3 + k] = F32(s->quat[k]);
    }
    nv += nd; nq += nqq;
  }
  if (nv != sp->ndof || nq > ORC_NQ) return MIR_E_INVALID;
  m->nv = nv; m->nq = nq;
  int nu = 0;
  for (int i = 0; i < nv; i++) {
    const MirDofSpec* s = &sp->dof[i];
    m->dof_limited[i] = s->limited; m->dof_ctrl[i] = s->ctrl_mode;
    m->dof_uadr[i] = s->ctrl_mode == MIR_CTRL_POSITION ? nu++ : -1;
    m->range[i][0] = F32(s->range[0]); m->range[i][1] = F32(s->range[1]);
    m->armature[i] = F32(s->armature); m->damping[i] = F32(s->damping);
    m->kp[i] = F32(s->kp); m->kv[i] = F32(s->kv);
    m->frc[i][0] = F32(s->frc_range[0]); m->frc[i][1] = F32(s->frc_range[1]);
    for (int k = 0; k < 2; k++) m->dsolref[i][k] = F32(s->solref[k]);
    for (int k = 0; k < 5; k++) m->dsolimp[i][k] = F32(s->solimp[k]);
  }
  m->nu = nu;
  m->nvert = sp->nvert < 0 ? 0 : (sp->nvert > MIR_MAX_VERT ? MIR_MAX_VERT : sp->nvert);
  for (int i = 0; i < m->nvert; i++)
    for (int k = 0; k < 3; k++) m->vert[i][k] = F32(sp->vert[i][k]);
  for (int g = 0; g < sp->ngeom; g++) {
    const MirGeomSpec* s = &sp->geom[g];
    m->gbody[g] = s->body; m->gtype[g] = s->type;
    for (int k = 0; k < 3; k++) { m->gsize[g][k] = F32(s->size[k]); m->gpos[g][k] = F32(s->pos[k]); }
    if (s->type == MIR_GEOM_HULL && ((int)s->size[0] < 0 || (int)s->size[1] < 4 || (int)s->size[1] > MIR_MAX_HULL_VERT || (int)s->size[0] + (int)s->size[1] > sp->nvert))
      return -1;
    for (int k = 0; k < 4; k++) m->gquat[g][k] = F32(s->quat[k]);
    m->gfriction[g] = F32(s->friction);
    for (int k = 0; k < 2; k++) m->gsolref[g][k] = F32(s->solref[k]);
    for (int k = 0; k < 5; k++) m->gsolimp[g][k] = F32(s->solimp[k]);
  }
  /* candidate pairs: ordered (g1<g2), planes first in a pair */
  int np = 0;
  for (int g1 = 0; g1 < sp->ngeom; g1++)
    for (int g2 = g1 + 1; g2 < sp->ngeom; g2++) {
      int a = g1, b = g2;
      if (m->gtype[b] == MIR_GEOM_PLANE) { int t = a; a = b; b = t; }
      if (m->gtype[b] == MIR_GEOM_PLANE) continue; /* plane-plane */
      int ba = m->gbody[a], bb = m->gbody[b];
      if (ba == bb) continue;
      if (m->is_static[ba] && m->is_static[bb]) continue;
      if (!((sp->geom[a].contype & sp->geom[b].conaffinity) || (sp->geom[b].contype & sp->geom[a].conaffinity))) continue;
      if (!m->opt.enable_adjacent_collision) {
        /* parent/child, treating welded (jointless) chains as one link; world excluded */
        int la = ba, lb = bb;
        while (la > 0 && m->ndof[la] == 0) la = m->parent[la];
        while (lb > 0 && m->ndof[lb] == 0) lb = m->parent[lb];
        if (la == lb) continue;
        int pa2 = la > 0 ? m->parent[la] : -1, pb2 = lb > 0 ? m->parent[lb] : -1;
        while (pa2 > 0 && m->ndof[pa2] == 0) pa2 = m->parent[pa2];
        while (pb2 > 0 && m->ndof[pb2] == 0) pb2 = m->parent[pb2];
        if ((pa2 == lb && lb > 0) || (pb2 == la && la > 0)) continue;
      }
      if (!m->opt.enable_self_collision && ba > 0 && bb > 0 && m->root[ba] == m->root[bb]) continue;
      if (np >= ORC_NP) return MIR_E_CAPACITY;
      m->pair_g1[np] = a; m->pair_g2[np] = b; np++;
    }
  m->npair = np;
  /* constants at the reference configuration qpos0 */
  OrcData* d = (OrcData*)__builtin_alloca(sizeof(OrcData));
  memset(d, 0, sizeof *d);
  memcpy(d->qpos, m->qpos0, sizeof d->qpos);
  orc_fk(m, d);
  orc_crb(m, d);
  real L[ORC_NV][ORC_NV], Minv[ORC_NV][ORC_NV];
  chol(nv, d->M, L);
  for (int i = 0; i < nv; i++) {
    real e[ORC_NV] = {0}, x[ORC_NV];
    e[i] = 1;
    chol_solve(nv, L, e, x);
    for (int j = 0; j < nv; j++) Minv[j][i] = x[j];
  }
  real tr = 0;
  for (int i = 0; i < nv; i++) tr += d->M[i][i];
  m->meaninertia = nv ? tr / nv : 1;
  for (int b = 0; b < m->nbody; b++) {
    int da = m->dofadr[b];
    if (m->jtype[b] == MIR_JNT_FREE) {
      real t = (Minv[da][da] + Minv[da + 1][da + 1] + Minv[da + 2][da + 2]) / 3;
      real r = (Minv[da + 3][da + 3] + Minv[da + 4][da + 4] + Minv[da + 5][da + 5]) / 3;
      for (int k = 0; k < 3; k++) { m->dof_invweight0[da + k] = t; m->dof_invweight0[da + 3 + k] = r; }
    } else if (m->ndof[b] == 1)
      m->dof_invweight0[da] = Minv[da][da];
  }
  for (int b = 1; b < m->nbody; b++) {
    if (m->is_static[b]) { m->body_invweight0[b] = 0; continue; }
    /* translational inverse weight at the body COM: trace(Jp Minv Jp^T)/3 */
    real Jp[3][ORC_NV];
    memset(Jp, 0, sizeof Jp);
    for (int a = 0; a < 3; a++) {
      real dir[3] = {0, 0, 0};
      dir[a] = 1;
      add_jac(m, d, b, d->xipos[b], dir, 1, Jp[a]);
    }
    real t = 0;
    for (int a = 0; a < 3; a++)
      for (int i = 0; i < nv; i++)
        for (int j = 0; j < nv; j++) t += Jp[a][i] * Minv[i][j] * Jp[a][j];
    m->body_invweight0[b] = t / 3 > MINVAL ? t / 3 : MINVAL;
  }
  return MIR_OK;
}

void orc_init_data(const OrcModel* m, OrcData* d) {
  memset(d, 0, sizeof *d);
  memcpy(d->qpos, m->qpos0, sizeof d->qpos);
  orc_fk(m, d);
}

/* obj_pos (nfree,3) / obj_quat (nfree,4): poses of ALL free bodies in body order (mir_reset) */
void orc_reset(const OrcModel* m, OrcData* d, const double* obj_pos, const double* obj_quat, const double* arm_qpos) {
  int ia = 0, nf = 0;
  for (int b = 1; b < m->nbody; b++) {
    int qa = m->qadr[b];
    if (m->jtype[b] == MIR_JNT_FREE) {
      for (int k = 0; k < 3; k++) d->qpos[qa + k] = (real)(float)obj_pos[nf * 3 + k];
      for (int k = 0; k < 4; k++) d->qpos[qa + 3 + k] = (real)(float)obj_quat[nf * 4 + k];
      nf++;
    } else if (m->ndof[b] == 1) {
      d->qpos[qa] = (real)(float)arm_qpos[ia];
      d->target[m->dofadr[b]] = (real)(float)arm_qpos[ia];
      ia++;
    }
  }
  memset(d->qvel, 0, sizeof d->qvel);
  memset(d->qacc_ws, 0, sizeof d->qacc_ws);
  orc_fk(m, d);
}

void orc_set_targets(const OrcModel* m, OrcData* d, const double* tgt) {
  for (int i = 0; i < m->nv; i++)
    if (m->dof_uadr[i] >= 0) d->target[i] = (real)tgt[m->dof_uadr[i]];
}

/* agent_pos: [eef pos3, eef quat4, grip q] (MIR_AGENT_EEF) or the scalar-joint qpos in body order (MIR_AGENT_QPOS);
 * env_state: [obj pos3, obj quat4, eef-obj 3, |eef-obj|] (+ obj2 pos3 when task.obj2_body >= 0) */
void orc_get_obs(const OrcModel* m, const OrcData* d, double* agent_pos, double* env_state, double* reward, unsigned char* terminated) {
  int e = m->task.eef_body, o = m->task.obj_body, o2 = m->task.obj2_body;
  if (m->task.agent_mode == MIR_AGENT_QPOS) {
    int ia = 0;
    for (int b = 1; b < m->nbody; b++)
      if (m->jtype[b] != MIR_JNT_FREE && m->ndof[b] == 1) agent_pos[ia++] = (double)d->qpos[m->qadr[b]];
  } else {
    for (int k = 0; k < 3; k++) agent_pos[k] = (double)d->xpos[e][k];
    for (int k = 0; k < 4; k++) agent_pos[3 + k] = (double)d->xquat[e][k];
    for (int k = 0; k < m->task.n_grip; k++) agent_pos[7 + k] = (double)d->qpos[m->dof_qadr[m->task.grip_dof[k]]];
  }
  double nn = 0;
  for (int k = 0; k < 3; k++) env_state[k] = (double)d->xpos[o][k];
  for (int k = 0; k < 4; k++) env_state[3 + k] = (double)d->xquat[o][k];
  for (int k = 0; k < 3; k++) {
    double df = (double)d->xpos[e][k] - (double)d->xpos[o][k];
    env_state[7 + k] = df;
    nn += df * df;
  }
  env_state[10] = sqrt(nn);
  if (o2 >= 0)
    for (int k = 0; k < 3; k++) env_state[11 + k] = (double)d->xpos[o2][k];
  if (m->task.reward_mode == MIR_REWARD_STACK) {
    /* compared in float32 as the reference does on its float32 tensors (cube_stack_batch.py:143-153) */
    float dx = (float)d->xpos[o][0] - (float)d->xpos[o2][0], dy = (float)d->xpos[o][1] - (float)d->xpos[o2][1];
    float dz = (float)d->xpos[o][2] - (float)d->xpos[o2][2];
    *reward = (sqrtf(dx * dx + dy * dy) < (float)m->task.reward_xy && dz > (float)m->task.reward_dz) ? 1.0 : 0.0;
  } else {
    /* reward compared in float32 as the reference does (cube_pick.py:132-134) */
    float z = (float)d->xpos[o][2];
    *reward = z > (float)m->task.reward_z ? 1.0 : 0.0;
  }
  *terminated = *reward == 1.0;
}

/* ------------------------------------------------------------------ inverse kinematics (checker of mir_inverse_kinematics)
 * Restates the algorithm DEFINED in include/mirigid.h (the reference's IK is inside the external Genesis package:
 * examples/franka/pick_cube_state.py:46-51): damped least squares on the geometric Jacobian of the chain world -> link,
 * step clamp, joint-range clamp.  Written against the oracle's own dense kinematics (orc_fk on a scratch copy), no
 * chain compaction, Gaussian elimination with partial pivoting for the 6x6 system.
 * q_io (n_arm): scalar joints in body order, seed in / solution out.  err2: |e_pos|, |e_rot| at exit. */
int orc_ik(const OrcModel* m, int link, const double* target_pos, const double* target_quat /* nullable */, double* q_io,
           int max_iters, double damping, double pos_tol, double rot_tol, double max_step, int respect_limits, double* err2) {
  /* include/mirigid.h states the iteration (damped least squares with the Levenberg - Marquardt acceptance rule of round 6);
   * mir_ik.hip applies the same.  Returns the iterations taken (>= 0), -1 when out of memory. */
  OrcData* d = (OrcData*)malloc(sizeof(OrcData));
  if (!d) return -1;
  orc_init_data(m, d);
  /* scalar joints in body order <-> bodies */
  int jb[ORC_NB], nj = 0;
  for (int b = 1; b < m->nbody; b++)
    if (m->jtype[b] != MIR_JNT_FREE && m->ndof[b] == 1) jb[nj++] = b;
  int onchain[ORC_NB] = {0};
  for (int b = link; b > 0; b = m->parent[b]) onchain[b] = 1;
  double tq[4] = {1, 0, 0, 0};
  if (target_quat) {
    double nn = sqrt(target_quat[0] * target_quat[0] + target_quat[1] * target_quat[1] + target_quat[2] * target_quat[2] + target_quat[3] * target_quat[3]);
    for (int k = 0; k < 4; k++) tq[k] = target_quat[k] / nn;
  }
  /* the ACCEPTED iterate: joint angles, its task-space error, its Jacobian, its scaled error */
  double q_acc[ORC_NB], e_acc[6] = {0}, J[6][ORC_NB], m_acc = 0, epn = 0, ern = 0;
  double lam2 = damping * damping;
  const double lam2_min = lam2 / 256.0, lam2_max = lam2 * 64.0;
  int stall = 0, iters = 0;
  for (int j = 0; j < nj; j++) q_acc[j] = q_io[j];
  for (int it = 0; it <= max_iters; it++) {
    /* the candidate q_io: forward kinematics, task-space error */
    for (int j = 0; j < nj; j++) d->qpos[m->qadr[jb[j]]] = (real)q_io[j];
    orc_fk(m, d);
    double ep[3], er[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) ep[k] = target_pos[k] - (double)d->xpos[link][k];
    if (target_quat) {
      /* dq = tq * conj(q_link) */
      double qc[4] = {(double)d->xquat[link][0], -(double)d->xquat[link][1], -(double)d->xquat[link][2], -(double)d->xquat[link][3]};
      double dq[4] = {tq[0] * qc[0] - tq[1] * qc[1] - tq[2] * qc[2] - tq[3] * qc[3], tq[0] * qc[1] + tq[1] * qc[0] + tq[2] * qc[3] - tq[3] * qc[2],
                      tq[0] * qc[2] - tq[1] * qc[3] + tq[2] * qc[0] + tq[3] * qc[1], tq[0] * qc[3] + tq[1] * qc[2] - tq[2] * qc[1] + tq[3] * qc[0]};
      if (dq[0] < 0) for (int k = 0; k < 4; k++) dq[k] = -dq[k];
      double sn = sqrt(dq[1] * dq[1] + dq[2] * dq[2] + dq[3] * dq[3]);
      double kk = sn > 1e-9 ? 2.0 * atan2(sn, dq[0]) / sn : 2.0;
      for (int k = 0; k < 3; k++) er[k] = kk * dq[1 + k];
    }
    const double epn_c = sqrt(ep[0] * ep[0] + ep[1] * ep[1] + ep[2] * ep[2]), ern_c = sqrt(er[0] * er[0] + er[1] * er[1] + er[2] * er[2]);
    const double metric = epn_c / pos_tol + ern_c / rot_tol;
    if (it == 0 || metric < m_acc) {
      /* accepted: the damping relaxes; stagnation = an accepted step that gained less than 1 % */
      if (it > 0) {
        stall = metric > 0.99 * m_acc ? stall + 1 : 0;
        lam2 = lam2 * 0.25 > lam2_min ? lam2 * 0.25 : lam2_min;
      }
      m_acc = metric; epn = epn_c; ern = ern_c;
      for (int k = 0; k < 3; k++) { e_acc[k] = ep[k]; e_acc[3 + k] = er[k]; }
      for (int j = 0; j < nj; j++) q_acc[j] = q_io[j];
      /* Jacobian columns of the chain joints at the accepted iterate */
      for (int j = 0; j < nj; j++) {
        int b = jb[j];
        for (int r = 0; r < 6; r++) J[r][j] = 0;
        if (!onchain[b]) continue;
        double ax[3];
        for (int r = 0; r < 3; r++) ax[r] = (double)d->xmat[b][3 * r] * (double)m->axis[b][0] + (double)d->xmat[b][3 * r + 1] * (double)m->axis[b][1] + (double)d->xmat[b][3 * r + 2] * (double)m->axis[b][2];
        if (m->jtype[b] == MIR_JNT_REVOLUTE) {
          double rr[3] = {(double)d->xpos[link][0] - (double)d->xpos[b][0], (double)d->xpos[link][1] - (double)d->xpos[b][1], (double)d->xpos[link][2] - (double)d->xpos[b][2]};
          J[0][j] = ax[1] * rr[2] - ax[2] * rr[1]; J[1][j] = ax[2] * rr[0] - ax[0] * rr[2]; J[2][j] = ax[0] * rr[1] - ax[1] * rr[0];
          if (target_quat) { J[3][j] = ax[0]; J[4][j] = ax[1]; J[5][j] = ax[2]; }
        } else {
          J[0][j] = ax[0]; J[1][j] = ax[1]; J[2][j] = ax[2];
        }
      }
    } else {
      /* rejected (the scaled error did not fall): back to the accepted iterate with eight times the damping; counts as a stalled iteration */
      stall++;
      lam2 = lam2 * 8.0 < lam2_max ? lam2 * 8.0 : lam2_max;
    }
    if ((epn < pos_tol && ern < rot_tol) || it == max_iters || stall >= 3) break;
    iters = it + 1;
    double A[6][7];
    for (int r = 0; r < 6; r++) {
      for (int c = 0; c < 6; c++) {
        double s2 = r == c ? lam2 : 0.0;
        for (int j = 0; j < nj; j++) s2 += J[r][j] * J[c][j];
        A[r][c] = s2;
      }
      A[r][6] = e_acc[r];
    }
    for (int c = 0; c < 6; c++) { /* Gaussian elimination, partial pivoting */
      int p = c;
      for (int r = c + 1; r < 6; r++) if (fabs(A[r][c]) > fabs(A[p][c])) p = r;
      if (p != c) for (int k = 0; k < 7; k++) { double t = A[c][k]; A[c][k] = A[p][k]; A[p][k] = t; }
      for (int r = 0; r < 6; r++) {
        if (r == c) continue;
        double f = A[r][c] / A[c][c];
        for (int k = c; k < 7; k++) A[r][k] -= f * A[c][k];
      }
    }
    double y[6], dqv[ORC_NB], big = 0;
    for (int r = 0; r < 6; r++) y[r] = A[r][6] / A[r][r];
    for (int j = 0; j < nj; j++) {
      dqv[j] = 0;
      for (int r = 0; r < 6; r++) dqv[j] += J[r][j] * y[r];
      if (fabs(dqv[j]) > big) big = fabs(dqv[j]);
    }
    double sc = big > max_step ? max_step / big : 1.0;
    for (int j = 0; j < nj; j++) {
      int b = jb[j];
      q_io[j] = q_acc[j];
      if (!onchain[b]) continue;
      q_io[j] = q_acc[j] + sc * dqv[j];
      int dofi = m->dofadr[b];
      if (respect_limits && m->dof_limited[dofi]) {
        if (q_io[j] < (double)m->range[dofi][0]) q_io[j] = (double)m->range[dofi][0];
        if (q_io[j] > (double)m->range[dofi][1]) q_io[j] = (double)m->range[dofi][1];
      }
    }
  }
  for (int j = 0; j < nj; j++) q_io[j] = q_acc[j];
  if (err2) { err2[0] = epn; err2[1] = ern; }
  free(d);
  return iters;
}

/* ------------------------------------------------------------------ accessors */
static int cp(double* out, const real* in, int n) { for (int i = 0; i < n; i++) out[i] = (double)in[i]; return n; }

int orc_read(const OrcModel* m, const OrcData* d, int field, double* out) {
  int nv = m->nv, k = 0;
  switch (field) {
    case ORC_F_QPOS: return cp(out, d->qpos, m->nq);
    case ORC_F_QVEL: return cp(out, d->qvel, nv);
    case ORC_F_TARGET: return cp(out, d->target, nv);
    case ORC_F_QACC_WS: return cp(out, d->qacc_ws, nv);
    case ORC_F_XPOS: for (int b = 0; b < m->nbody; b++) k += cp(out + k, d->xpos[b], 3); return k;
    case ORC_F_XQUAT: for (int b = 0; b < m->nbody; b++) k += cp(out + k, d->xquat[b], 4); return k;
    case ORC_F_XIPOS: for (int b = 0; b < m->nbody; b++) k += cp(out + k, d->xipos[b], 3); return k;
    case ORC_F_M: for (int i = 0; i < nv; i++) k += cp(out + k, d->M[i], nv); return k;
    case ORC_F_MT: for (int i = 0; i < nv; i++) k += cp(out + k, d->Mt[i], nv); return k;
    case ORC_F_QFRC_BIAS: return cp(out, d->qfrc_bias, nv);
    case ORC_F_QFRC_SMOOTH: return cp(out, d->qfrc_smooth, nv);
    case ORC_F_QFRC_ACT: return cp(out, d->qfrc_act, nv);
    case ORC_F_QFRC_PASSIVE: return cp(out, d->qfrc_passive, nv);
    case ORC_F_QACC_SMOOTH: return cp(out, d->qacc_smooth, nv);
    case ORC_F_QACC: return cp(out, d->qacc, nv);
    case ORC_F_CPOS: for (int c = 0; c < d->ncon; c++) k += cp(out + k, d->cpos[c], 3); return k;
    case ORC_F_CDIST: return cp(out, d->cdist, d->ncon);
    case ORC_F_CFRAME: for (int c = 0; c < d->ncon; c++) k += cp(out + k, d->cframe[c], 9); return k;
    case ORC_F_J: for (int r = 0; r < d->nefc; r++) k += cp(out + k, d->J[r], nv); return k;
    case ORC_F_AREF: return cp(out, d->aref, d->nefc);
    case ORC_F_EFCD: return cp(out, d->efcD, d->nefc);
    case ORC_F_EFCPOS: return cp(out, d->efcpos, d->nefc);
    case ORC_F_EFCFORCE: return cp(out, d->efcforce, d->nefc);
    case ORC_F_DOF_INVWEIGHT0: return cp(out, m->dof_invweight0, nv);
    case ORC_F_BODY_INVWEIGHT0: return cp(out, m->body_invweight0, m->nbody);
    case ORC_F_MEANINERTIA: out[0] = (double)m->meaninertia; return 1;
    case ORC_F_DBG_IMP: return cp(out, d->dbg_imp, d->niter < 64 ? d->niter : 64);
    case ORC_F_DBG_GN: return cp(out, d->dbg_gn, d->niter < 64 ? d->niter : 64);
    case ORC_F_DBG_ALPHA: return cp(out, d->dbg_alpha, d->niter < 64 ? d->niter : 64);
    case ORC_F_DBG_LS: return cp(out, d->dbg_ls, d->niter < 64 ? d->niter : 64);
  }
  return -1;
}

int orc_write(const OrcModel* m, OrcData* d, int field, const double* in) {
  real* dst; int n;
  switch (field) {
    case ORC_F_QPOS: dst = d->qpos; n = m->nq; break;
    case ORC_F_QVEL: dst = d->qvel; n = m->nv; break;
    case ORC_F_TARGET: dst = d->target; n = m->nv; break;
    case ORC_F_QACC_WS: dst = d->qacc_ws; n = m->nv; break;
    default: return -1;
  }
  for (int i = 0; i < n; i++) dst[i] = (real)in[i];
  return n;
}

int orc_counts(const OrcData* d, int* ncon, int* nefc, int* niter) {
  if (ncon) *ncon = d->ncon;
  if (nefc) *nefc = d->nefc;
  if (niter) *niter = d->niter;
  return 0;
}

/* batched accessors (tests at 4096 envs): field `field` of B consecutive data blocks into out (B, stride) / from in (B, stride);
 * contact counts into ncon (B); observations as orc_get_obs, rows of agent_dim / env_dim doubles */
int orc_read_batch(const OrcModel* m, const OrcData* d, int B, int field, double* out, int stride) {
  int n = 0;
  for (int e = 0; e < B; e++) {
    n = orc_read(m, d + e, field, out + (size_t)e * stride);
    if (n < 0 || n > stride) return -1;
  }
  return n;
}
int orc_write_batch(const OrcModel* m, OrcData* d, int B, int field, const double* in, int stride) {
  int n = 0;
  for (int e = 0; e < B; e++) {
    n = orc_write(m, d + e, field, in + (size_t)e * stride);
    if (n < 0) return -1;
  }
  return n;
}
void orc_counts_batch(const OrcData* d, int B, int* ncon, int* nefc, int* niter) {
  for (int e = 0; e < B; e++) {
    if (ncon) ncon[e] = d[e].ncon;
    if (nefc) nefc[e] = d[e].nefc;
    if (niter) niter[e] = d[e].niter;
  }
}
void orc_ncand_batch(const OrcData* d, int B, int* ncand) {
  for (int e = 0; e < B; e++) ncand[e] = d[e].ncand;
}
void orc_get_obs_batch(const OrcModel* m, const OrcData* d, int B, double* agent_pos, int agent_dim, double* env_state, int env_dim, double* reward,
                       unsigned char* terminated) {
  for (int e = 0; e < B; e++) orc_get_obs(m, d + e, agent_pos + (size_t)e * agent_dim, env_state + (size_t)e * env_dim, reward + e, terminated + e);
}

/* ------------------------------------------------------------------ independent ABA (unconstrained) */
/* Featherstone articulated-body algorithm in the same c-frame, used only to
 * cross-check CRB + Cholesky: qacc_aba == M^-1 (tau - bias) with the plain
 * (armature-included, no implicit damping) mass matrix. */
void orc_aba(const OrcModel* m, OrcData* d, double* qacc_out) {
  const int nb = m->nbody, nv = m->nv;
  orc_fk(m, d);
  orc_rne(m, d); /* gives cvel per body */
  /* articulated inertias as dense 6x6 (motion->force), bias forces pA */
  real IA[ORC_NB][6][6], pA[ORC_NB][6], cb[ORC_NB][6];
  real U[ORC_NV][6], Dd[ORC_NV], u[ORC_NV];
  real tau[ORC_NV];
  for (int i = 0; i < nv; i++) tau[i] = d->qfrc_passive[i] + d->qfrc_act[i];
  for (int b = 1; b < nb; b++) {
    for (int c = 0; c < 6; c++) {
      real e[6] = {0, 0, 0, 0, 0, 0}, colv[6];
      e[c] = 1;
      inert_mul(colv, d->cinert[b], e);
      for (int r = 0; r < 6; r++) IA[b][r][c] = colv[r];
    }
    real Iv[6];
    inert_mul(Iv, d->cinert[b], d->cvel[b]);
    cross_force(pA[b], d->cvel[b], Iv);
    /* velocity-product acceleration c_b = sum cdof_dot qvel (recomputed like orc_rne) */
    memset(cb[b], 0, sizeof cb[b]);
    int p = m->parent[b], da = m->dofadr[b];
    real cv[6];
    memcpy(cv, d->cvel[p], sizeof cv);
    if (m->jtype[b] == MIR_JNT_FREE) {
      for (int k = 0; k < 3; k++)
        for (int c = 0; c < 6; c++) cv[c] += d->cdof[da + k][c] * d->qvel[da + k];
      for (int k = 0; k < 3; k++) {
        real dot[6];
        cross_motion(dot, cv, d->cdof[da + 3 + k]);
        for (int c = 0; c < 6; c++) cb[b][c] += dot[c] * d->qvel[da + 3 + k];
      }
    } else if (m->ndof[b] == 1) {
      real dot[6];
      cross_motion(dot, cv, d->cdof[da]);
      for (int c = 0; c < 6; c++) cb[b][c] = dot[c] * d->qvel[da];
    }
  }
  /* backward pass; multi-dof joints handled one dof at a time as a chain of
   * massless intermediate bodies (valid because dofs of one body are processed
   * last-to-first with rank-1 updates) */
  for (int b = nb - 1; b >= 1; b--) {
    int da = m->dofadr[b], nd = m->ndof[b];
    /* working copies */
    real Ia[6][6], pa[6];
    memcpy(Ia, IA[b], sizeof Ia);
    memcpy(pa, pA[b], sizeof pa);
    /* add Ia*cb to bias */
    for (int r = 0; r < 6; r++) { real t = 0; for (int c = 0; c < 6; c++) t += Ia[r][c] * cb[b][c]; pa[r] += t; }
    for (int k = nd - 1; k >= 0; k--) {
      int i = da + k;
      const real* S = d->cdof[i];
      for (int r = 0; r < 6; r++) { real t = 0; for (int c = 0; c < 6; c++) t += Ia[r][c] * S[c]; U[i][r] = t; }
      Dd[i] = dot6(S, U[i]) + m->armature[i];
      u[i] = tau[i] - dot6(S, pa);
      for (int r = 0; r < 6; r++) {
        for (int c = 0; c < 6; c++) Ia[r][c] -= U[i][r] * U[i][c] / Dd[i];
        pa[r] += U[i][r] * u[i] / Dd[i];
      }
    }
    int p = m->parent[b];
    if (p > 0) {
      for (int r = 0; r < 6; r++) { for (int c = 0; c < 6; c++) IA[p][r][c] += Ia[r][c]; pA[p][r] += pa[r]; }
    }
    /* store reduced quantities needed in the forward pass */
  }
  /* forward pass needs, per dof, the inertia "below" that dof; recompute by
   * repeating the reduction (cheap, clarity over speed) */
  real acc[ORC_NB][6];
  memset(acc, 0, sizeof acc);
  acc[0][3] = -F32(m->opt.gravity[0]); acc[0][4] = -F32(m->opt.gravity[1]); acc[0][5] = -F32(m->opt.gravity[2]);
  for (int b = 1; b < nb; b++) {
    int p = m->parent[b], da = m->dofadr[b], nd = m->ndof[b];
    real a[6];
    for (int c = 0; c < 6; c++) a[c] = acc[p][c] + cb[b][c];
    /* note: Ia*cb was folded into pa before the per-dof reduction, so here a starts from the parent only */
    for (int c = 0; c < 6; c++) a[c] = acc[p][c];
    for (int k = 0; k < nd; k++) {
      int i = da + k;
      real qdd = (u[i] - dot6(U[i], a)) / Dd[i];
      qacc_out[i] = (double)qdd;
      for (int c = 0; c < 6; c++) a[c] += d->cdof[i][c] * qdd;
    }
    for (int c = 0; c < 6; c++) acc[b][c] = a[c] + cb[b][c];
  }
}

/* ------------------------------------------------------------------ batch driver (timed CPU baseline) */
void orc_step_batch(const OrcModel* m, OrcData* d, int B, const float* action, int nthreads) {
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
  for (int e = 0; e < B; e++) {
    if (action)
      for (int i = 0; i < m->nv; i++)
        if (m->dof_uadr[i] >= 0) d[e].target[i] = (real)action[(size_t)e * m->nu + m->dof_uadr[i]];
    orc_step(m, &d[e]);
  }
}

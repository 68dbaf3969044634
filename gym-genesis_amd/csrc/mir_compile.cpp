// mir_compile.cpp — host-side compile of a MirSceneSpec into the float32 DevModel.
//
// Restates what the reference obtains from scene.build()
// (/root/reference/gym_genesis/tasks/franka/cube_pick.py:65): topology tables, the
// static collision-pair filter and the constraint-regularisation constants
// (inverse weights at the reference pose, mean inertia) that the soft-constraint
// model needs.  Host double precision, run once.
//
// The joint-space inertia at qpos0 is assembled from body Jacobians
// (M = sum_b m Jv^T Jv + Jw^T I Jw), not by the CRB recursion the kernels use.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <cstring>
#include <vector>

#include "mir_model.h"

namespace {

struct V3 {
  double x, y, z;
};
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(double s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Q4 {
  double w, x, y, z;
};
inline Q4 qmul(Q4 a, Q4 b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
inline V3 qrot(Q4 q, V3 v) {
  V3 u{q.x, q.y, q.z};
  V3 t = 2.0 * cross(u, v);
  return v + q.w * t + cross(u, t);
}
struct M3 {
  double m[3][3];
};
inline M3 q2m(Q4 q) {
  M3 R;
  V3 c0 = qrot(q, {1, 0, 0}), c1 = qrot(q, {0, 1, 0}), c2 = qrot(q, {0, 0, 1});
  R.m[0][0] = c0.x; R.m[1][0] = c0.y; R.m[2][0] = c0.z;
  R.m[0][1] = c1.x; R.m[1][1] = c1.y; R.m[2][1] = c1.z;
  R.m[0][2] = c2.x; R.m[1][2] = c2.y; R.m[2][2] = c2.z;
  return R;
}

int fail(char* err, int code, const char* msg) {
  if (err) snprintf(err, 255, "%s", msg);
  return code;
}

// nearest ancestor-or-self that carries a joint (welded chains collapse onto it); 0 = world-welded
int moving_link(const DevModel& m, int b) {
  while (b > 0 && m.b_jtype[b] == MIR_JNT_FIXED) b = m.b_parent[b];
  return b;
}

}  // namespace

// Constants of the soft-constraint model at the reference pose qpos0 (free bodies at their spec pose, scalar
// joints at 0): joint-space inertia from body Jacobians, its inverse, per-dof and per-body inverse weights, mean
// inertia.  Works on the spec alone (no kernel-specific layout), so both device models share it.
int mir_host_consts(const MirSceneSpec* sp, HostConsts* out, char* err) {
  const int nb = sp->nbody;
  std::vector<int> dofadr(nb, 0), ndof(nb, 0), parent(nb, -1), is_static(nb, 0);
  int nv = 0;
  for (int b = 0; b < nb; b++) {
    int jt = b == 0 ? MIR_JNT_FIXED : sp->body[b].jtype;
    parent[b] = b == 0 ? -1 : sp->body[b].parent;
    dofadr[b] = nv;
    ndof[b] = jt == MIR_JNT_FREE ? 6 : (jt == MIR_JNT_FIXED ? 0 : 1);
    is_static[b] = b == 0 ? 1 : (ndof[b] == 0 && is_static[parent[b]]);
    nv += ndof[b];
  }
  std::vector<int> dbody(nv), dkind(nv), daxk(nv);
  for (int b = 0; b < nb; b++)
    for (int k = 0; k < ndof[b]; k++) {
      int i = dofadr[b] + k;
      dbody[i] = b;
      if (sp->body[b].jtype == MIR_JNT_FREE) { dkind[i] = k < 3 ? 2 : 3; daxk[i] = k % 3; }
      else { dkind[i] = sp->body[b].jtype == MIR_JNT_REVOLUTE ? 0 : 1; daxk[i] = 0; }
    }
  auto moves = [&](int i, int b) {  // does dof i move body b (is dof i's body an ancestor-or-self of b)?
    int db = dbody[i];
    while (b > 0) {
      if (b == db) return true;
      b = parent[b];
    }
    return false;
  };
  std::vector<Q4> xq(nb);
  std::vector<V3> xp(nb), xc(nb);
  xq[0] = {1, 0, 0, 0}; xp[0] = {0, 0, 0}; xc[0] = {0, 0, 0};
  for (int b = 1; b < nb; b++) {
    const MirBodySpec& s = sp->body[b];
    int p = s.parent;
    Q4 ql{s.quat[0], s.quat[1], s.quat[2], s.quat[3]};
    V3 pl{s.pos[0], s.pos[1], s.pos[2]};
    if (s.jtype == MIR_JNT_FREE) { xq[b] = ql; xp[b] = pl; }  // qpos0 = spec pose, joint value 0 elsewhere
    else { xq[b] = qmul(xq[p], ql); xp[b] = xp[p] + qrot(xq[p], pl); }
    xc[b] = xp[b] + qrot(xq[b], {s.ipos[0], s.ipos[1], s.ipos[2]});
  }
  // per-dof world axis / anchor
  std::vector<V3> ax(nv), an(nv);
  for (int i = 0; i < nv; i++) {
    int b = dbody[i];
    const MirBodySpec& s = sp->body[b];
    if (dkind[i] < 2) ax[i] = qrot(xq[b], {s.axis[0], s.axis[1], s.axis[2]});
    else { V3 e{0, 0, 0}; (&e.x)[daxk[i]] = 1; ax[i] = e; }
    an[i] = xp[b];
  }
  auto jac = [&](int b, V3 pt, int i, V3& jv, V3& jw) {  // column i of the Jacobian of point pt on body b
    jv = {0, 0, 0}; jw = {0, 0, 0};
    if (!moves(i, b)) return;
    if (dkind[i] == 0 || dkind[i] == 3) { jw = ax[i]; jv = cross(ax[i], pt - an[i]); }
    else jv = ax[i];
  };
  std::vector<double> M(nv * nv, 0.0);
  for (int b = 1; b < nb; b++) {
    const MirBodySpec& s = sp->body[b];
    M3 R = q2m(xq[b]);
    double Ib[3][3] = {{s.inertia[0], s.inertia[3], s.inertia[4]}, {s.inertia[3], s.inertia[1], s.inertia[5]}, {s.inertia[4], s.inertia[5], s.inertia[2]}};
    double Iw[3][3];
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) {
        double t = 0;
        for (int k = 0; k < 3; k++)
          for (int l = 0; l < 3; l++) t += R.m[r][k] * Ib[k][l] * R.m[c][l];
        Iw[r][c] = t;
      }
    for (int i = 0; i < nv; i++) {
      V3 vi, wi;
      jac(b, xc[b], i, vi, wi);
      V3 Iwi{Iw[0][0] * wi.x + Iw[0][1] * wi.y + Iw[0][2] * wi.z, Iw[1][0] * wi.x + Iw[1][1] * wi.y + Iw[1][2] * wi.z,
             Iw[2][0] * wi.x + Iw[2][1] * wi.y + Iw[2][2] * wi.z};
      for (int j = 0; j < nv; j++) {
        V3 vj, wj;
        jac(b, xc[b], j, vj, wj);
        M[i * nv + j] += s.mass * dot(vi, vj) + dot(Iwi, wj);
      }
    }
  }
  for (int i = 0; i < nv; i++) M[i * nv + i] += sp->dof[i].armature;
  // invert by Gauss-Jordan (SPD, tiny)
  std::vector<double> A(M), Inv(nv * nv, 0.0);
  for (int i = 0; i < nv; i++) Inv[i * nv + i] = 1;
  for (int c = 0; c < nv; c++) {
    double p = A[c * nv + c];
    if (!(p > 1e-12)) return fail(err, MIR_E_INVALID, "joint-space inertia at qpos0 is not positive definite");
    for (int j = 0; j < nv; j++) { A[c * nv + j] /= p; Inv[c * nv + j] /= p; }
    for (int r = 0; r < nv; r++)
      if (r != c) {
        double f = A[r * nv + c];
        if (f != 0)
          for (int j = 0; j < nv; j++) { A[r * nv + j] -= f * A[c * nv + j]; Inv[r * nv + j] -= f * Inv[c * nv + j]; }
      }
  }
  double tr = 0;
  for (int i = 0; i < nv; i++) tr += M[i * nv + i];
  HostConsts& H = *out;
  memset(&H, 0, sizeof H);
  H.meaninertia = nv ? tr / nv : 1.0;
  for (int b = 1; b < nb; b++) {
    int da = dofadr[b];
    if (sp->body[b].jtype == MIR_JNT_FREE) {
      double t = (Inv[da * nv + da] + Inv[(da + 1) * nv + da + 1] + Inv[(da + 2) * nv + da + 2]) / 3;
      double r = (Inv[(da + 3) * nv + da + 3] + Inv[(da + 4) * nv + da + 4] + Inv[(da + 5) * nv + da + 5]) / 3;
      for (int k = 0; k < 3; k++) { H.dof_invweight0[da + k] = t; H.dof_invweight0[da + 3 + k] = r; }
    } else if (sp->body[b].jtype != MIR_JNT_FIXED)
      H.dof_invweight0[da] = Inv[da * nv + da];
    if (is_static[b]) continue;
    double t = 0;
    for (int i = 0; i < nv; i++) {
      V3 vi, wi;
      jac(b, xc[b], i, vi, wi);
      for (int j = 0; j < nv; j++) {
        V3 vj, wj;
        jac(b, xc[b], j, vj, wj);
        t += dot(vi, vj) * Inv[i * nv + j];
      }
    }
    H.body_invweight0[b] = fmax(t / 3, 1e-15);
  }
  return MIR_OK;
}

// Float32-round every model constant of a spec in place: the device models are float32, so the scene that is
// actually simulated is the rounded one and every derived constant must come from the same rounded values.
void mir_round_spec(MirSceneSpec* rounded) {
  auto r32 = [](double& v) { v = (double)(float)v; };
  for (int b = 0; b < rounded->nbody; b++) {
    MirBodySpec& s = rounded->body[b];
    for (double& v : s.pos) r32(v);
    for (double& v : s.quat) r32(v);
    for (double& v : s.axis) r32(v);
    for (double& v : s.ipos) r32(v);
    for (double& v : s.inertia) r32(v);
    r32(s.mass);
  }
  for (int i = 0; i < rounded->ndof; i++) {
    MirDofSpec& s = rounded->dof[i];
    r32(s.armature); r32(s.damping); r32(s.kp); r32(s.kv);
    for (double& v : s.range) r32(v);
    for (double& v : s.solref) r32(v);
    for (double& v : s.solimp) r32(v);
  }
  r32(rounded->opt.dt);
  for (double& v : rounded->opt.gravity) r32(v);
}

int mir_compile_model(const MirSceneSpec* sp, DevModel* out, HostConsts* hc, char* err) {
  if (!sp || !out) return fail(err, MIR_E_INVALID, "null spec");
  if (sp->struct_size != (int)sizeof(MirSceneSpec) || sp->version != MIR_VERSION)
    return fail(err, MIR_E_INVALID, "MirSceneSpec size/version mismatch (ABI)");
  if (sp->nbody < 1 || sp->nbody > K16_MAX_BODY || sp->ndof > K16_MAX_DOF || sp->ngeom > K16_MAX_GEOM || sp->ndof < 0 ||
      sp->ngeom < 0)
    return fail(err, MIR_E_CAPACITY, "scene exceeds K16_MAX_BODY/DOF/GEOM");
  if (sp->task.obj2_body >= 0 || sp->task.reward_mode != MIR_REWARD_LIFT || sp->task.agent_mode != MIR_AGENT_EEF)
    return fail(err, MIR_E_CAPACITY, "the 16-lane kernel extracts only the pick-task observation layout");
  // The device model is float32, so the scene that is actually simulated is the float32-rounded
  // one: derive every constant (inverse weights, mean inertia) from those same rounded values.
  MirSceneSpec rounded = *sp;
  mir_round_spec(&rounded);
  sp = &rounded;
  DevModel& m = *out;
  memset(&m, 0, sizeof m);
  const int nb = sp->nbody;
  m.nbody = nb;
  m.ngeom = sp->ngeom;
  m.dt = (float)sp->opt.dt;
  m.gx = (float)sp->opt.gravity[0]; m.gy = (float)sp->opt.gravity[1]; m.gz = (float)sp->opt.gravity[2];
  m.tolerance = (float)sp->opt.tolerance;
  m.ls_tolerance = (float)sp->opt.ls_tolerance;
  m.iterations = sp->opt.iterations;
  m.ls_iterations = sp->opt.ls_iterations;
  m.enable_collision = sp->opt.enable_collision;
  m.enable_joint_limit = sp->opt.enable_joint_limit;
  m.max_contacts = sp->opt.max_contacts;
  if (m.max_contacts > K16_MAX_CONTACT || m.max_contacts < 0) return fail(err, MIR_E_CAPACITY, "max_contacts > K16_MAX_CONTACT");

  // ---- topology --------------------------------------------------------------------------
  int nv = 0, nq = 0, narm = 0;
  for (int b = 0; b < nb; b++) {
    const MirBodySpec& s = sp->body[b];
    int jt = b == 0 ? MIR_JNT_FIXED : s.jtype;
    if (b > 0 && (s.parent < 0 || s.parent >= b)) return fail(err, MIR_E_INVALID, "body parent must precede the body");
    if (jt == MIR_JNT_FREE && s.parent != 0) return fail(err, MIR_E_INVALID, "free joint must hang off the world");
    m.b_parent[b] = b == 0 ? -1 : s.parent;
    m.b_jtype[b] = jt;
    m.b_dofadr[b] = nv;
    m.b_qadr[b] = nq;
    m.b_root[b] = b == 0 ? 0 : (s.parent == 0 ? b : m.b_root[s.parent]);
    int nd = jt == MIR_JNT_FREE ? 6 : (jt == MIR_JNT_FIXED ? 0 : 1);
    m.b_static[b] = b == 0 ? 1 : (nd == 0 && m.b_static[s.parent]);
    for (int k = 0; k < 3; k++) { m.b_pos[b][k] = (float)s.pos[k]; m.b_axis[b][k] = (float)s.axis[k]; m.b_ipos[b][k] = (float)s.ipos[k]; }
    for (int k = 0; k < 4; k++) m.b_quat[b][k] = (float)s.quat[k];
    for (int k = 0; k < 6; k++) m.b_inertia[b][k] = (float)s.inertia[k];
    m.b_mass[b] = (float)s.mass;
    uint32_t inherited = (b > 0 && s.parent > 0) ? m.b_dofmask[s.parent] : 0u;
    if (nv + nd > K16_MAX_DOF) return fail(err, MIR_E_CAPACITY, "too many dofs");
    for (int k = 0; k < nd; k++) {
      int i = nv + k;
      m.d_body[i] = b;
      m.d_qadr[i] = nq + k;
      m.d_armidx[i] = -1;
      if (jt == MIR_JNT_FREE) {
        m.d_kind[i] = k < 3 ? 2 : 3;
        m.d_axis_k[i] = k % 3;
        // velocity "before" dof i: ancestors + (for rotations) the three translations only
        m.d_premask[i] = inherited | (k < 3 ? ((1u << k) - 1u) << nv : 7u << nv);
      } else {
        m.d_kind[i] = jt == MIR_JNT_REVOLUTE ? 0 : 1;
        m.d_premask[i] = inherited;
        m.d_armidx[i] = narm++;
      }
      m.d_ancmask[i] = inherited | (((1u << (k + 1)) - 1u) << nv);
    }
    m.b_dofmask[b] = inherited | (nd ? (((1u << nd) - 1u) << nv) : 0u);
    if (jt == MIR_JNT_FREE) {
      if (m.nfree >= MIR_MAX_FREE) return fail(err, MIR_E_CAPACITY, "more than MIR_MAX_FREE free bodies");
      m.free_qadr[m.nfree++] = nq;
    }
    nv += nd;
    nq += jt == MIR_JNT_FREE ? 7 : nd;
  }
  if (nv != sp->ndof) return fail(err, MIR_E_INVALID, "ndof does not match the joints");
  if (nq > K16_MAX_Q) return fail(err, MIR_E_CAPACITY, "nq > K16_MAX_Q");
  m.nv = nv; m.nq = nq; m.n_arm_q = narm;
  m.qstride = (m.nq + 3) & ~3;  // (the Franka pick scene: 16 floats = one 64-byte row)
  for (int b = 0; b < nb; b++) {
    uint32_t sub = 1u << b;
    for (int c = b + 1; c < nb; c++) {
      int a = m.b_parent[c];
      while (a > b) a = m.b_parent[a];
      if (a == b && b > 0) sub |= 1u << c;
    }
    m.b_submask[b] = sub;
  }

  // The tree scans of the step kernel need (a) bodies numbered in depth-first preorder: the subtree of body b is the lane
  // range [b, b + size), and (b) every ancestor set to be a prefix of a dof chain (true by construction above).  A scene
  // whose bodies are not in preorder goes to the wave-per-env kernel, which makes no such assumption.
  for (int b = 1; b < nb; b++) {
    int size = 0;
    for (int c = 0; c < nb; c++) size += (m.b_submask[b] >> c & 1u) ? 1 : 0;
    const uint32_t range = (size >= 32 ? 0xffffffffu : ((1u << size) - 1u)) << b;
    if (m.b_submask[b] != range) return fail(err, MIR_E_CAPACITY, "bodies are not numbered in depth-first preorder");
  }

  {  // block split of the mass matrix: trees own contiguous dof ranges (bodies are in preorder)
    int ntree = 0, second = 0, last_root = -1;
    for (int i = 0; i < nv; i++) {
      const int r = m.b_root[m.d_body[i]];
      if (r != last_root) { ntree++; if (ntree == 2) second = i; last_root = r; }
    }
    m.gj_split = (ntree == 2 && (second == 6 || second == 9)) ? second : 0;
  }

  // ---- dof parameters -------------------------------------------------------------------
  int nu = 0;
  for (int i = 0; i < nv; i++) {
    const MirDofSpec& s = sp->dof[i];
    m.d_limited[i] = s.limited && m.d_kind[i] < 2;
    m.d_ctrl[i] = s.ctrl_mode;
    m.d_uadr[i] = s.ctrl_mode == MIR_CTRL_POSITION ? nu++ : -1;
    m.d_lo[i] = (float)s.range[0]; m.d_hi[i] = (float)s.range[1];
    m.d_damping[i] = (float)s.damping; m.d_kp[i] = (float)s.kp; m.d_kv[i] = (float)s.kv;
    m.d_frclo[i] = (float)fmax(s.frc_range[0], -3.0e38); m.d_frchi[i] = (float)fmin(s.frc_range[1], 3.0e38);
    m.d_armature[i] = (float)s.armature;
    double add = s.armature;
    if (sp->opt.implicit_damping) add += sp->opt.dt * (s.damping + (s.ctrl_mode == MIR_CTRL_POSITION ? s.kv : 0.0));
    m.d_mdiag[i] = (float)add;
    double dmax = fmin(fmax(s.solimp[1], 1e-4), 0.9999);
    double tc = fmax(s.solref[0], 2 * sp->opt.dt), dr = s.solref[1];
    m.d_k[i] = (float)(1.0 / (dmax * dmax * tc * tc * dr * dr));
    m.d_b[i] = (float)(2.0 / (dmax * tc));
    for (int k = 0; k < 5; k++) m.d_solimp[i][k] = (float)s.solimp[k];
  }
  m.nu = nu;

  // ---- task -----------------------------------------------------------------------------
  m.eef_body = sp->task.eef_body; m.obj_body = sp->task.obj_body; m.n_grip = sp->task.n_grip;
  if (m.eef_body < 0 || m.eef_body >= nb || m.obj_body < 0 || m.obj_body >= nb || m.n_grip < 0 || m.n_grip > MIR_MAX_GRIP)
    return fail(err, MIR_E_INVALID, "task body / gripper indices out of range");
  m.obj_qadr = m.b_jtype[m.obj_body] == MIR_JNT_FREE ? m.b_qadr[m.obj_body] : -1;
  for (int k = 0; k < m.n_grip; k++) {
    int d = sp->task.grip_dof[k];
    if (d < 0 || d >= nv) return fail(err, MIR_E_INVALID, "grip dof out of range");
    m.grip_qadr[k] = m.d_qadr[d];
  }
  m.reward_z = (float)sp->task.reward_z;

  // ---- geoms + static pair filter ---------------------------------------------------------
  for (int g = 0; g < sp->ngeom; g++) {
    const MirGeomSpec& s = sp->geom[g];
    if (s.body < 0 || s.body >= nb) return fail(err, MIR_E_INVALID, "geom body out of range");
    if (s.type != MIR_GEOM_PLANE && s.type != MIR_GEOM_BOX && s.type != MIR_GEOM_SPHERE && s.type != MIR_GEOM_CAPSULE && s.type != MIR_GEOM_HULL)
      return fail(err, MIR_E_INVALID, "unsupported geom type");
    if (s.type == MIR_GEOM_HULL) {
      const int v0 = (int)s.size[0], nvg = (int)s.size[1];
      if (v0 < 0 || nvg < 4 || nvg > MIR_MAX_HULL_VERT || v0 + nvg > sp->nvert || sp->nvert > MIR_MAX_VERT)
        return fail(err, MIR_E_INVALID, "hull geom: vertex range outside the scene's vertex pool (4 .. MIR_MAX_HULL_VERT vertices)");
      if (sp->nvert > K16_MAX_VERT) return fail(err, MIR_E_CAPACITY, "more hull vertices than the 16-lane kernel keeps in LDS (K16_MAX_VERT)");
      m.has_convex = 1;
    }
    if ((s.type == MIR_GEOM_SPHERE || s.type == MIR_GEOM_CAPSULE) && !(s.size[0] > 0)) return fail(err, MIR_E_INVALID, "sphere / capsule radius must be > 0");
    if (s.type == MIR_GEOM_CAPSULE && !(s.size[1] >= 0)) return fail(err, MIR_E_INVALID, "capsule half length must be >= 0");
    if (s.type == MIR_GEOM_SPHERE || s.type == MIR_GEOM_CAPSULE) m.has_convex = 1;
    m.g_body[g] = s.body; m.g_type[g] = s.type;
    for (int k = 0; k < 3; k++) { m.g_size[g][k] = (float)s.size[k]; m.g_pos[g][k] = (float)s.pos[k]; }
    for (int k = 0; k < 4; k++) m.g_quat[g][k] = (float)s.quat[k];
    m.g_friction[g] = (float)s.friction;
    for (int k = 0; k < 2; k++) m.g_solref[g][k] = (float)s.solref[k];
    for (int k = 0; k < 5; k++) m.g_solimp[g][k] = (float)s.solimp[k];
  }
  m.nvert = sp->nvert > 0 && sp->nvert <= K16_MAX_VERT ? sp->nvert : 0;
  for (int i = 0; i < m.nvert; i++) {
    for (int k = 0; k < 3; k++) m.hverts[i][k] = (float)sp->vert[i][k];
    m.hverts[i][3] = 0.0f;
  }
  int np = 0;
  uint32_t allow[K16_MAX_GEOM] = {0};
  for (int ga = 0; ga < sp->ngeom; ga++)
    for (int gb = ga + 1; gb < sp->ngeom; gb++) {
      int a = ga, b = gb;
      if (m.g_type[b] == MIR_GEOM_PLANE) { a = gb; b = ga; }
      if (m.g_type[b] == MIR_GEOM_PLANE) continue;
      int ba = m.g_body[a], bb = m.g_body[b];
      if (ba == bb || (m.b_static[ba] && m.b_static[bb])) continue;
      const MirGeomSpec &sa = sp->geom[a], &sb = sp->geom[b];
      if (!((sa.contype & sb.conaffinity) || (sb.contype & sa.conaffinity))) continue;
      if (!sp->opt.enable_adjacent_collision) {
        int la = moving_link(m, ba), lb = moving_link(m, bb);
        if (la == lb) continue;
        int pa = la > 0 ? moving_link(m, m.b_parent[la]) : -1;
        int pb = lb > 0 ? moving_link(m, m.b_parent[lb]) : -1;
        if ((pa == lb && lb > 0) || (pb == la && la > 0)) continue;
      }
      if (!sp->opt.enable_self_collision && ba > 0 && bb > 0 && m.b_root[ba] == m.b_root[bb]) continue;
      allow[ga] |= 1u << gb; allow[gb] |= 1u << ga;
      if (np < K16_MAX_PAIR) { m.p_g1[np] = a; m.p_g2[np] = b; }
      np++;
    }
  // Broadphase: a short static list is tested pair by pair; a list longer than K16_MAX_PAIR (self-collision enabled, many
  // geoms) is replaced by the sweep-and-prune over world AABBs, which needs only the symmetric "may collide" masks.
  // MIR_BROADPHASE=sap / static forces one or the other (static fails with MIR_E_CAPACITY when the list does not fit).
  const char* bp = getenv("MIR_BROADPHASE");
  m.use_sap = (np > K16_MAX_PAIR || (bp && !strcmp(bp, "sap"))) ? 1 : 0;
  if (bp && !strcmp(bp, "static")) m.use_sap = 0;
  if (!m.use_sap && np > K16_MAX_PAIR) return fail(err, MIR_E_CAPACITY, "too many candidate collision pairs");
  m.npair = np;

  // ---- constants at qpos0: M from body Jacobians, inverse weights ----------------------------
  HostConsts local;
  HostConsts& H = hc ? *hc : local;
  {
    int rc = mir_host_consts(sp, &H, err);
    if (rc != MIR_OK) return rc;
  }
  for (int i = 0; i < nv; i++) m.d_invweight0[i] = (float)H.dof_invweight0[i];
  for (int b = 0; b < nb; b++) m.b_invweight0[b] = (float)H.body_invweight0[b];
  m.meaninertia = (float)H.meaninertia;
  m.solver_scale = (float)(1.0 / (H.meaninertia * (nv > 1 ? nv : 1)));

  // ---- packed lookup tables for the kernel's LDS copy ---------------------------------------
  ModelTab& t = m.tab;
  for (int g = 0; g < m.ngeom; g++) {
    for (int k = 0; k < 3; k++) { t.g_pos[g][k] = m.g_pos[g][k]; t.g_size[g][k] = m.g_size[g][k]; }
    t.g_pos[g][3] = m.g_friction[g];
    {  // bounding radius about the geom centre
      const float* z = m.g_size[g];
      t.g_size[g][3] = m.g_type[g] == MIR_GEOM_SPHERE ? z[0] : (m.g_type[g] == MIR_GEOM_CAPSULE ? z[0] + z[1] : sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]));
      if (m.g_type[g] == MIR_GEOM_HULL) {  // farthest vertex from the geom frame's origin; bounding box for the rasteriser
        float r2 = 0.0f;
        for (int i = (int)z[0]; i < (int)z[0] + (int)z[1]; i++) {
          const float* v = m.hverts[i];
          r2 = fmaxf(r2, v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
          for (int k = 0; k < 3; k++) m.g_bbox[g][k] = fmaxf(m.g_bbox[g][k], fabsf(v[k]));
        }
        t.g_size[g][3] = sqrtf(r2);
      }
    }
    for (int k = 0; k < 4; k++) t.g_quat[g][k] = m.g_quat[g][k];
    t.g_info[g][0] = m.g_body[g]; t.g_info[g][1] = m.g_type[g];
    t.g_sol[g][0] = m.g_solref[g][0]; t.g_sol[g][1] = m.g_solref[g][1];
    for (int k = 0; k < 5; k++) t.g_sol[g][2 + k] = m.g_solimp[g][k];
  }
  for (int p = 0; p < m.npair && p < K16_MAX_PAIR; p++) t.pair[p] = m.p_g1[p] | (m.p_g2[p] << 8);
  for (int g = 0; g < m.ngeom; g++) t.g_allow[g] = allow[g];
  for (int b = 0; b < nb; b++) {
    t.b_info[b][0] = (int32_t)m.b_dofmask[b]; t.b_info[b][1] = m.b_root[b]; t.b_info[b][2] = m.b_qadr[b]; t.b_info[b][3] = m.b_dofadr[b];
    t.b_invw[b] = m.b_invweight0[b];
  }
  for (int i = 0; i < nv; i++) {
    t.d_lim[i][0] = m.d_lo[i]; t.d_lim[i][1] = m.d_hi[i]; t.d_lim[i][2] = m.d_invweight0[i]; t.d_lim[i][3] = m.d_k[i]; t.d_lim[i][4] = m.d_b[i];
    for (int k = 0; k < 5; k++) t.d_lim[i][5 + k] = m.d_solimp[i][k];
  }
  // ---- per-lane register constants (LaneK16) and the packed parent links ---------------------
  // FK links: the static transforms of FIXED bodies are folded into their children (double precision, once), so the pointer
  // jumping of the kernel's FK walks only over jointed ancestors -- the Panda's finger is 8 links from the world instead of 10,
  // three rounds instead of four.  (A fixed body itself keeps its parent: its pose is still needed.)
  int fk_parent[MIR_G];
  double fk_pos[MIR_G][3], fk_quat[MIR_G][4];
  m.parents = 0;
  for (int b = 1; b < nb; b++) {
    const MirBodySpec& sb0 = sp->body[b];
    int par = sb0.parent;
    double pos[3] = {sb0.pos[0], sb0.pos[1], sb0.pos[2]}, q[4] = {sb0.quat[0], sb0.quat[1], sb0.quat[2], sb0.quat[3]};
    while (par > 0 && sp->body[par].jtype == MIR_JNT_FIXED) {
      const MirBodySpec& sp_ = sp->body[par];
      const double w = sp_.quat[0], x = sp_.quat[1], y = sp_.quat[2], z = sp_.quat[3];
      // pos <- p.pos + R(p.quat) pos
      const double tx = 2 * (y * pos[2] - z * pos[1]), ty = 2 * (z * pos[0] - x * pos[2]), tz = 2 * (x * pos[1] - y * pos[0]);
      const double rx = pos[0] + w * tx + (y * tz - z * ty), ry = pos[1] + w * ty + (z * tx - x * tz), rz = pos[2] + w * tz + (x * ty - y * tx);
      pos[0] = sp_.pos[0] + rx; pos[1] = sp_.pos[1] + ry; pos[2] = sp_.pos[2] + rz;
      // quat <- p.quat * quat
      const double a = q[0], bq = q[1], c = q[2], d = q[3];
      q[0] = w * a - x * bq - y * c - z * d; q[1] = w * bq + x * a + y * d - z * c;
      q[2] = w * c - x * d + y * a + z * bq; q[3] = w * d + x * c - y * bq + z * a;
      par = sp_.parent;
    }
    fk_parent[b] = par;
    for (int k = 0; k < 3; k++) fk_pos[b][k] = pos[k];
    for (int k = 0; k < 4; k++) fk_quat[b][k] = q[k];
    m.parents |= (uint64_t)(par & 15) << (4 * b);
  }
  m.fk_free_leaf = 1;
  for (int b = 1; b < nb; b++) {
    if (sp->body[b].jtype == MIR_JNT_FREE && fk_parent[b] != 0) m.fk_free_leaf = 0;
    if (fk_parent[b] > 0 && sp->body[fk_parent[b]].jtype == MIR_JNT_FREE) m.fk_free_leaf = 0;
  }
  // ---- weights of the early-mask bound (mir_model.h: term_bound_ok): Mt >= blockdiag(diag(d_mdiag) over jointed dofs, free bodies) ----
  float gw[MIR_G];
  {
    bool ok = m.fk_free_leaf != 0 && m.obj_qadr >= 0;
    for (int l = 0; l < MIR_G; l++) gw[l] = 0.0f;
    for (int l = 0; l < nv && ok; l++) {
      const int b = m.d_body[l];
      if (m.d_kind[l] < 2) {
        if (m.d_mdiag[l] > 0.0f) gw[l] = 1.0f / m.d_mdiag[l];
        else ok = false;
      } else {
        // free body: translations and rotations decouple when the centre of mass is the body origin; lowest eigenvalue of the
        // rotational inertia bounded from below by Gershgorin
        const float* I = m.b_inertia[b];
        const float lo = fminf(fminf(I[0] - fabsf(I[3]) - fabsf(I[4]), I[1] - fabsf(I[3]) - fabsf(I[5])), I[2] - fabsf(I[4]) - fabsf(I[5]));
        if (m.b_ipos[b][0] != 0.0f || m.b_ipos[b][1] != 0.0f || m.b_ipos[b][2] != 0.0f || !(m.b_mass[b] > 0.0f) || !(lo > 0.0f)) ok = false;
        else gw[l] = m.d_kind[l] == 2 ? 1.0f / m.b_mass[b] : 1.0f / lo;
      }
    }
    // (1 / object mass folded into the weights: the kernel needs sqrt(sum_i gw_i g_i^2) = |g|_{Mt^-1} / sqrt(mass) only)
    for (int l = 0; l < MIR_G; l++) gw[l] = ok ? gw[l] / m.b_mass[m.obj_body] : 0.0f;
    m.term_bound_ok = ok ? 1 : 0;
    m.term_zlane = ok ? m.b_dofadr[m.obj_body] + 2 : 0;
    m.term_zscale = ok ? 1.0f / sqrtf(m.b_mass[m.obj_body]) : 0.0f;
    m.pad_term = 0.0f;
  }
  for (int l = 0; l < MIR_G; l++) {
    LaneK16 k;
    memset(&k, 0, sizeof k);
    k.b_jtype = m.b_jtype[l]; k.b_qadr = m.b_qadr[l]; k.b_root = m.b_root[l];
    k.b_dofmask = m.b_dofmask[l]; k.b_submask = m.b_submask[l]; k.b_mass = m.b_mass[l];
    for (int c = 0; c < 3; c++) { k.b_pos[c] = (l > 0 && l < nb) ? (float)fk_pos[l][c] : m.b_pos[l][c]; k.b_axis[c] = m.b_axis[l][c]; k.b_ipos[c] = m.b_ipos[l][c]; }
    for (int c = 0; c < 4; c++) k.b_quat[c] = (l > 0 && l < nb) ? (float)fk_quat[l][c] : m.b_quat[l][c];
    for (int c = 0; c < 6; c++) k.b_inertia[c] = m.b_inertia[l][c];
    const int db = l < nv ? m.d_body[l] : 0;
    k.d_body = db; k.d_kind = m.d_kind[l]; k.d_qadr = m.d_qadr[l]; k.d_axis_k = m.d_axis_k[l]; k.d_root = m.b_root[db];
    k.d_ctrl = m.d_ctrl[l]; k.d_uadr = m.d_uadr[l]; k.d_submask = m.b_submask[db];
    for (int c = 0; c < 3; c++) k.d_axis[c] = m.b_axis[db][c];
    k.d_premask = m.d_premask[l]; k.d_ancmask = m.d_ancmask[l];
    k.d_limited = (l < nv && m.d_limited[l] && m.enable_joint_limit) ? 1 : 0;
    k.d_damping = m.d_damping[l]; k.d_kp = m.d_kp[l]; k.d_kv = m.d_kv[l]; k.d_frclo = m.d_frclo[l]; k.d_frchi = m.d_frchi[l];
    k.d_mdiag = m.d_mdiag[l];
    k.obs_qadr = (l >= 7 && l < 7 + m.n_grip) ? m.grip_qadr[l - 7] : 0;
    {  // links of the tree scans (mir_step.hip): all derived from the masks above
      auto top = [](uint32_t mk) { int t = -1; for (int i = 0; i < 32; i++) if (mk >> i & 1u) t = i; return t; };
      int d_par = -1, d_bef = -1, b_last = -1, b_next = MIR_G;
      if (l < nv) {
        d_par = top(m.d_ancmask[l] & ~(1u << l));   // the dof in front of this one on its chain
        d_bef = top(m.d_premask[l]);                 // whose inclusive sum is the velocity "before" this dof
        // (both are prefixes of the chain: premask / ancmask are the chain up to that dof -- checked below)
      }
      if (l > 0 && l < nb) {
        b_last = top(m.b_dofmask[l]);
        int size = 0;
        for (int c = 0; c < nb; c++) size += (m.b_submask[l] >> c & 1u) ? 1 : 0;
        b_next = l + size;
      }
      k.scan = (d_par & 255) | ((d_bef & 255) << 8) | ((b_last & 255) << 16) | ((b_next & 255) << 24);
    }
    k.d_gw = gw[l];
    for (int q = 0; q < 12; q++) memcpy(m.lanek_t[q][l], reinterpret_cast<const char*>(&k) + 16 * q, 16);
  }
  return MIR_OK;
}

// ---- scene constants for a specialised instantiation of the 16-lane kernel ------------------------------------------
// The sizes and options of a compiled scene as a C++ struct of literals (`struct <name>`), plus `matches(const DevModel&)`:
// mir_create runs the specialised instantiation only for a scene whose compiled model carries exactly these values, every other
// scene runs the generic one.  tools/gen_scene_spec.py writes the headline scene's struct into mir_spec_pick.h (committed; a CPU
// test regenerates and compares it).
#define MIR_SPEC_FIELDS(X) \
  X(nbody) X(nv) X(nq) X(ngeom) X(npair) X(max_contacts) X(enable_collision) X(iterations) X(ls_iterations) X(eef_body) X(obj_body) X(n_grip) \
  X(gj_split) X(obj_qadr) X(use_sap) X(has_convex) X(term_zlane)
extern "C" int mir_debug_emit_spec(const MirSceneSpec* spec, const char* name, char* out, int32_t cap) {
  static DevModel m;  // (large: not on the stack)
  HostConsts hc;
  char err[256] = {0};
  const int rc = mir_compile_model(spec, &m, &hc, err);
  if (rc != MIR_OK) return rc;
  std::string t = std::string("struct ") + name + " {\n";
#define X(f) t += "  static constexpr int " #f " = " + std::to_string((int)m.f) + ";\n";
  MIR_SPEC_FIELDS(X)
#undef X
  t += "  static constexpr int fk_free_leaf = " + std::to_string((int)m.fk_free_leaf) + ";\n";
  t += "  static bool matches(const DevModel& m) {\n    return m.fk_free_leaf == (uint64_t)fk_free_leaf";
#define X(f) t += " && m." #f " == " #f;
  MIR_SPEC_FIELDS(X)
#undef X
  t += ";\n  }\n};\n";
  if ((int)t.size() + 1 > cap) return MIR_E_CAPACITY;
  memcpy(out, t.c_str(), t.size() + 1);
  return (int)t.size();
}

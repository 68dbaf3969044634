// mir_compile64.cpp — host-side compile of a MirSceneSpec into the lane-mapped DevModel64 of the
// wave-per-env kernel (mir_model64.h): kinematic trees are packed into the four 16-lane blocks, every mask is
// expressed in lane space, the static collision-pair filter and the soft-constraint constants are the same as
// for the 16-lane model (mir_compile.cpp).  Host double precision, run once per mir_create.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mir_model64.h"

namespace {
int fail(char* err, int code, const char* msg) {
  if (err) snprintf(err, 255, "%s", msg);
  return code;
}
}  // namespace

int mir_compile_model64(const MirSceneSpec* sp0, DevModel64* out, HostConsts* hc, char* err) {
  if (!sp0 || !out) return fail(err, MIR_E_INVALID, "null spec");
  if (sp0->struct_size != (int)sizeof(MirSceneSpec) || sp0->version != MIR_VERSION)
    return fail(err, MIR_E_INVALID, "MirSceneSpec size/version mismatch (ABI)");
  if (sp0->nbody < 1 || sp0->nbody > K64_MAX_BODY || sp0->nbody > MIR_MAX_BODY || sp0->ndof > MIR_MAX_DOF || sp0->ngeom > MIR_MAX_GEOM ||
      sp0->ndof < 0 || sp0->ngeom < 0)
    return fail(err, MIR_E_CAPACITY, "scene exceeds MIR_MAX_BODY/DOF/GEOM");
  MirSceneSpec rounded = *sp0;
  mir_round_spec(&rounded);
  const MirSceneSpec* sp = &rounded;
  DevModel64& m = *out;
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
  if (m.max_contacts > MIR_MAX_CONTACT || m.max_contacts < 0) return fail(err, MIR_E_CAPACITY, "max_contacts > MIR_MAX_CONTACT");

  // ---- topology in compact dof order ---------------------------------------------------------
  std::vector<int> ndof(nb, 0), dofadr(nb, 0), tree_ndof(nb, 0), tree_lane(nb, -1), tree_next(nb, 0);
  int nv = 0, nq = 0, narm = 0, nfree = 0;
  for (int b = 0; b < nb; b++) {
    const MirBodySpec& s = sp->body[b];
    int jt = b == 0 ? MIR_JNT_FIXED : s.jtype;
    if (b > 0 && (s.parent < 0 || s.parent >= b)) return fail(err, MIR_E_INVALID, "body parent must precede the body");
    if (jt == MIR_JNT_FREE && s.parent != 0) return fail(err, MIR_E_INVALID, "free joint must hang off the world");
    m.b_parent[b] = b == 0 ? -1 : s.parent;
    m.b_jtype[b] = jt;
    m.b_qadr[b] = nq;
    m.b_root[b] = b == 0 ? 0 : (s.parent == 0 ? b : m.b_root[s.parent]);
    ndof[b] = jt == MIR_JNT_FREE ? 6 : (jt == MIR_JNT_FIXED ? 0 : 1);
    dofadr[b] = nv;
    m.b_static[b] = b == 0 ? 1 : (ndof[b] == 0 && m.b_static[s.parent]);
    for (int k = 0; k < 3; k++) { m.b_pos[b][k] = (float)s.pos[k]; m.b_axis[b][k] = (float)s.axis[k]; m.b_ipos[b][k] = (float)s.ipos[k]; }
    for (int k = 0; k < 4; k++) m.b_quat[b][k] = (float)s.quat[k];
    for (int k = 0; k < 6; k++) m.b_inertia[b][k] = (float)s.inertia[k];
    m.b_mass[b] = (float)s.mass;
    tree_ndof[m.b_root[b]] += ndof[b];
    nv += ndof[b];
    nq += jt == MIR_JNT_FREE ? 7 : ndof[b];
    if (jt == MIR_JNT_FREE) nfree++;
  }
  if (nv != sp->ndof) return fail(err, MIR_E_INVALID, "ndof does not match the joints");
  if (nq > MIR_MAX_Q || nq > K64_QSTRIDE) return fail(err, MIR_E_CAPACITY, "nq > MIR_MAX_Q");
  if (nfree > MIR_MAX_FREE) return fail(err, MIR_E_CAPACITY, "more than MIR_MAX_FREE free bodies");
  m.nv = nv; m.nq = nq; m.nfree = nfree;

  // ---- pack trees into 16-lane blocks (first fit, in root order; a tree never straddles a block) -----
  {
    int blk = 0, fill = 0;
    for (int r = 1; r < nb; r++) {
      if (m.b_root[r] != r || tree_ndof[r] == 0) continue;
      if (tree_ndof[r] > K64_BLOCK_DOF) return fail(err, MIR_E_CAPACITY, "a kinematic tree has more than 15 dofs");
      if (fill + tree_ndof[r] > K64_BLOCK_DOF) { blk++; fill = 0; }
      if (blk >= W64 / 16) return fail(err, MIR_E_CAPACITY, "the scene's trees do not fit four 15-dof blocks");
      tree_lane[r] = blk * 16 + fill;
      tree_next[r] = tree_lane[r];
      fill += tree_ndof[r];
    }
  }
  std::vector<int> lane_of_dof(nv, -1);
  for (int l = 0; l < W64; l++) { m.d_dof[l] = -1; m.d_uadr[l] = -1; m.d_armidx[l] = -1; }
  int nfree_seen = 0;
  for (int b = 1; b < nb; b++) {
    int r = m.b_root[b];
    m.b_block[b] = tree_lane[r] >= 0 ? tree_lane[r] / 16 : -1;
    m.b_dofadr[b] = tree_next[r];
    uint64_t inherited = m.b_parent[b] > 0 ? m.b_dofmask[m.b_parent[b]] : 0ull;
    const int la = tree_next[r];
    for (int k = 0; k < ndof[b]; k++) {
      const int l = la + k, i = dofadr[b] + k;
      lane_of_dof[i] = l;
      m.d_dof[l] = i;
      m.d_body[l] = b;
      m.d_qadr[l] = m.b_qadr[b] + k;
      if (m.b_jtype[b] == MIR_JNT_FREE) {
        m.d_kind[l] = k < 3 ? 2 : 3;
        m.d_axis_k[l] = k % 3;
        m.d_premask[l] = inherited | (k < 3 ? ((1ull << k) - 1ull) << la : 7ull << la);
      } else {
        m.d_kind[l] = m.b_jtype[b] == MIR_JNT_REVOLUTE ? 0 : 1;
        m.d_premask[l] = inherited;
        m.d_armidx[l] = narm;
        m.arm_qadr[narm++] = m.b_qadr[b];
      }
      m.d_ancmask[l] = inherited | (((1ull << (k + 1)) - 1ull) << la);
      m.lanemask |= 1ull << l;
    }
    if (m.b_jtype[b] == MIR_JNT_FREE) m.free_qadr[nfree_seen++] = m.b_qadr[b];
    m.b_dofmask[b] = inherited | (ndof[b] ? (((1ull << ndof[b]) - 1ull) << la) : 0ull);
    tree_next[r] += ndof[b];
  }
  m.b_block[0] = -1;
  m.n_arm_q = narm;
  for (int b = 0; b < nb; b++) {
    uint32_t sub = 1u << b;
    for (int c = b + 1; c < nb; c++) {
      int a = m.b_parent[c];
      while (a > b) a = m.b_parent[a];
      if (a == b && b > 0) sub |= 1u << c;
    }
    m.b_submask[b] = sub;
  }

  // ---- dof parameters (lane-indexed) -------------------------------------------------------------
  int nu = 0;
  for (int i = 0; i < nv; i++) {
    const MirDofSpec& s = sp->dof[i];
    const int l = lane_of_dof[i];
    m.d_limited[l] = s.limited && m.d_kind[l] < 2;
    m.d_ctrl[l] = s.ctrl_mode;
    m.d_uadr[l] = s.ctrl_mode == MIR_CTRL_POSITION ? nu++ : -1;
    m.d_lo[l] = (float)s.range[0]; m.d_hi[l] = (float)s.range[1];
    m.d_damping[l] = (float)s.damping; m.d_kp[l] = (float)s.kp; m.d_kv[l] = (float)s.kv;
    m.d_frclo[l] = (float)fmax(s.frc_range[0], -3.0e38); m.d_frchi[l] = (float)fmin(s.frc_range[1], 3.0e38);
    m.d_armature[l] = (float)s.armature;
    double add = s.armature;
    if (sp->opt.implicit_damping) add += sp->opt.dt * (s.damping + (s.ctrl_mode == MIR_CTRL_POSITION ? s.kv : 0.0));
    m.d_mdiag[l] = (float)add;
    double dmax = fmin(fmax(s.solimp[1], 1e-4), 0.9999);
    double tc = fmax(s.solref[0], 2 * sp->opt.dt), dr = s.solref[1];
    m.d_k[l] = (float)(1.0 / (dmax * dmax * tc * tc * dr * dr));
    m.d_b[l] = (float)(2.0 / (dmax * tc));
    for (int k = 0; k < 5; k++) m.d_solimp[l][k] = (float)s.solimp[k];
  }
  m.nu = nu;

  // ---- task -----------------------------------------------------------------------------------------
  const MirTaskSpec& t = sp->task;
  m.eef_body = t.eef_body; m.obj_body = t.obj_body; m.obj2_body = t.obj2_body; m.n_grip = t.n_grip;
  m.reward_mode = t.reward_mode; m.agent_mode = t.agent_mode;
  if (m.eef_body < 0 || m.eef_body >= nb || m.obj_body < 0 || m.obj_body >= nb || m.obj2_body >= nb || m.n_grip < 0 || m.n_grip > MIR_MAX_GRIP)
    return fail(err, MIR_E_INVALID, "task body / gripper indices out of range");
  if (t.reward_mode == MIR_REWARD_STACK && t.obj2_body < 0) return fail(err, MIR_E_INVALID, "MIR_REWARD_STACK needs task.obj2_body");
  if (t.reward_mode != MIR_REWARD_LIFT && t.reward_mode != MIR_REWARD_STACK) return fail(err, MIR_E_INVALID, "unknown task.reward_mode");
  if (t.agent_mode != MIR_AGENT_EEF && t.agent_mode != MIR_AGENT_QPOS) return fail(err, MIR_E_INVALID, "unknown task.agent_mode");
  for (int k = 0; k < m.n_grip; k++) {
    int d = t.grip_dof[k];
    if (d < 0 || d >= nv) return fail(err, MIR_E_INVALID, "grip dof out of range");
    m.grip_qadr[k] = m.d_qadr[lane_of_dof[d]];
  }
  m.reward_z = (float)t.reward_z; m.reward_xy = (float)t.reward_xy; m.reward_dz = (float)t.reward_dz;
  m.agent_dim = t.agent_mode == MIR_AGENT_QPOS ? narm : 7 + m.n_grip;
  m.env_dim = t.obj2_body >= 0 ? 14 : 11;
  {
    auto free_root = [&](int b) { return b > 0 && m.b_jtype[b] == MIR_JNT_FREE && m.b_parent[b] == 0; };
    m.obj_qadr = free_root(m.obj_body) ? m.b_qadr[m.obj_body] : -1;
    m.obj2_qadr = (m.obj2_body >= 0 && free_root(m.obj2_body)) ? m.b_qadr[m.obj2_body] : -1;
    m.term_early = (m.obj_qadr >= 0 && (t.reward_mode != MIR_REWARD_STACK || m.obj2_qadr >= 0)) ? 1 : 0;
    m.pad_te = 0;
  }

  // ---- geoms + static pair filter (same rules as mir_compile.cpp) -----------------------------------
  for (int g = 0; g < sp->ngeom; g++) {
    const MirGeomSpec& s = sp->geom[g];
    if (s.body < 0 || s.body >= nb) return fail(err, MIR_E_INVALID, "geom body out of range");
    if (s.type != MIR_GEOM_PLANE && s.type != MIR_GEOM_BOX && s.type != MIR_GEOM_SPHERE && s.type != MIR_GEOM_CAPSULE && s.type != MIR_GEOM_HULL)
      return fail(err, MIR_E_INVALID, "unsupported geom type");
    if ((s.type == MIR_GEOM_SPHERE || s.type == MIR_GEOM_CAPSULE) && !(s.size[0] > 0)) return fail(err, MIR_E_INVALID, "sphere / capsule radius must be > 0");
    if (s.type == MIR_GEOM_CAPSULE && !(s.size[1] >= 0)) return fail(err, MIR_E_INVALID, "capsule half length must be >= 0");
    if (s.type == MIR_GEOM_SPHERE || s.type == MIR_GEOM_CAPSULE || s.type == MIR_GEOM_HULL) m.has_convex = 1;
    m.g_body[g] = s.body; m.g_type[g] = s.type;
    for (int k = 0; k < 3; k++) { m.g_size[g][k] = (float)s.size[k]; m.g_pos[g][k] = (float)s.pos[k]; }
    // bounding-sphere radius of the broadphase (box: half diagonal; sphere: radius; capsule: half length + radius), as in mir_compile.cpp
    m.g_size[g][3] = s.type == MIR_GEOM_SPHERE ? (float)s.size[0]
                     : (s.type == MIR_GEOM_CAPSULE ? (float)(s.size[0] + s.size[1])
                                                    : (float)std::sqrt(s.size[0] * s.size[0] + s.size[1] * s.size[1] + s.size[2] * s.size[2]));
    if (s.type == MIR_GEOM_HULL) {
      // vertex hull (the stand-in for the reference's mesh collision geometry, tasks/utils.py:372,561,732): size = (first vertex, count)
      // in the scene's pool; the kernel reads (first vertex, count, bounding radius about the geom origin) from g_size[0..2]
      const int v0 = (int)s.size[0], nvg = (int)s.size[1];
      if (v0 < 0 || nvg < 4 || nvg > MIR_MAX_HULL_VERT || v0 + nvg > sp->nvert || sp->nvert > MIR_MAX_VERT)
        return fail(err, MIR_E_INVALID, "hull geom: vertex range outside the scene's vertex pool (4 .. MIR_MAX_HULL_VERT vertices)");
      float r2 = 0.0f;
      for (int i = v0; i < v0 + nvg; i++) {
        const float v[3] = {(float)sp->vert[i][0], (float)sp->vert[i][1], (float)sp->vert[i][2]};
        r2 = fmaxf(r2, v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        for (int k = 0; k < 3; k++) m.g_bbox[g][k] = fmaxf(m.g_bbox[g][k], fabsf(v[k]));
      }
      m.g_size[g][2] = m.g_size[g][3] = sqrtf(r2);
    }
    m.g_pos[g][3] = (float)s.friction;
    for (int k = 0; k < 4; k++) m.g_quat[g][k] = (float)s.quat[k];
    m.g_sol[g][0] = (float)s.solref[0]; m.g_sol[g][1] = (float)s.solref[1];
    for (int k = 0; k < 5; k++) m.g_sol[g][2 + k] = (float)s.solimp[k];
  }
  m.nvert = sp->nvert > 0 && sp->nvert <= MIR_MAX_VERT ? sp->nvert : 0;
  for (int i = 0; i < m.nvert; i++) {
    for (int k = 0; k < 3; k++) m.hverts[i][k] = (float)sp->vert[i][k];
    m.hverts[i][3] = 0.0f;
  }
  auto moving_link = [&](int b) {
    while (b > 0 && m.b_jtype[b] == MIR_JNT_FIXED) b = m.b_parent[b];
    return b;
  };
  int np = 0;
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
        int la = moving_link(ba), lb = moving_link(bb);
        if (la == lb) continue;
        int pa = la > 0 ? moving_link(m.b_parent[la]) : -1;
        int pb = lb > 0 ? moving_link(m.b_parent[lb]) : -1;
        if ((pa == lb && lb > 0) || (pb == la && la > 0)) continue;
      }
      if (!sp->opt.enable_self_collision && ba > 0 && bb > 0 && m.b_root[ba] == m.b_root[bb]) continue;
      if (np >= MIR_MAX_PAIR) return fail(err, MIR_E_CAPACITY, "too many candidate collision pairs");
      m.pair[np++] = a | (b << 8);
    }
  m.npair = np;

  // ---- constants at qpos0 (shared helper, compact order) -> lanes ------------------------------------
  HostConsts local;
  HostConsts& H = hc ? *hc : local;
  {
    int rc = mir_host_consts(sp, &H, err);
    if (rc != MIR_OK) return rc;
  }
  for (int i = 0; i < nv; i++) m.d_invweight0[lane_of_dof[i]] = (float)H.dof_invweight0[i];
  for (int b = 0; b < nb; b++) m.b_invweight0[b] = (float)H.body_invweight0[b];
  m.meaninertia = (float)H.meaninertia;
  m.solver_scale = (float)(1.0 / (H.meaninertia * (nv > 1 ? nv : 1)));

  // ---- derived tables --------------------------------------------------------------------------------
  auto bits = [](uint32_t u) { float f; memcpy(&f, &u, 4); return f; };
  for (int l = 0; l < W64; l++) {
    const int db = ((m.lanemask >> l) & 1ull) ? m.d_body[l] : 0;
    m.d_root[l] = m.b_root[db]; m.d_qbase[l] = m.b_qadr[db]; m.d_lbase[l] = m.b_dofadr[db]; m.d_bsubmask[l] = m.b_submask[db];
    for (int c = 0; c < 3; c++) m.d_axis[l][c] = m.b_axis[db][c];
    m.d_axis[l][3] = 0.0f;
    int oq = 0;
    if (l < m.agent_dim) {
      if (m.agent_mode == MIR_AGENT_QPOS) oq = m.arm_qadr[l];
      else if (l >= 7) oq = m.grip_qadr[l - 7];
    }
    m.obs_qadr[l] = oq;
  }
  m.fk_free_leaf = 1;
  for (int b = 1; b < nb; b++) {
    if (m.b_jtype[b] == MIR_JNT_FREE && m.b_parent[b] != 0) m.fk_free_leaf = 0;
    if (m.b_parent[b] > 0 && m.b_jtype[m.b_parent[b]] == MIR_JNT_FREE) m.fk_free_leaf = 0;
  }
  // tree-scan links (the kernel's dynamics run prefix sums along dof chains and suffix sums over body lanes)
  {
    auto top = [](uint64_t mk) { int t = -1; for (int l = 0; l < 64; l++) if ((mk >> l) & 1ull) t = l; return t; };
    for (int l = 0; l < W64; l++) {
      int par = -1, bef = -1, last = -1, next = -1;
      if ((m.lanemask >> l) & 1ull) {
        par = top(m.d_ancmask[l] & ~(1ull << l));
        bef = top(m.d_premask[l]);
        if (par >= 0 && m.d_ancmask[par] != (m.d_ancmask[l] & ~(1ull << l))) return fail(err, MIR_E_INVALID, "dof chains must be laid out ancestor-first");
        if (bef >= 0 && m.d_ancmask[bef] != m.d_premask[l]) return fail(err, MIR_E_INVALID, "the dofs in front of a dof must form a chain prefix");
      }
      if (l > 0 && l < nb) {
        last = top(m.b_dofmask[l]);
        if (last >= 0 && m.d_ancmask[last] != m.b_dofmask[l]) return fail(err, MIR_E_INVALID, "the dofs that move a body must form a chain prefix");
        int e = l + 1;
        while (e < nb && ((m.b_submask[l] >> e) & 1u)) e++;
        if (m.b_submask[l] != (uint32_t)(((1ull << e) - 1ull) & ~((1ull << l) - 1ull)))
          return fail(err, MIR_E_INVALID, "bodies must be numbered depth-first (a subtree is a contiguous index range)");
        if (((e - 1) >> 4) != (l >> 4)) return fail(err, MIR_E_CAPACITY, "the bodies of a subtree must share a 16-lane row");
        next = (e >> 4) == (l >> 4) ? e : -1;
      }
      m.scanw[l] = (par & 255) | (bef & 255) << 8 | (last & 255) << 16 | (int32_t)((uint32_t)(next & 255) << 24);
    }
  }
  for (int b = 0; b < K64_MAX_BODY; b++) {
    const bool on = b < nb;
    m.b_tab[b][0] = on ? m.b_invweight0[b] : 0.0f;
    m.b_tab[b][1] = bits(on ? (uint32_t)m.b_dofmask[b] : 0u);
    m.b_tab[b][2] = bits(on ? (uint32_t)(m.b_dofmask[b] >> 32) : 0u);
    m.b_tab[b][3] = bits((uint32_t)(on ? m.b_block[b] : -1));
    m.b_tab[b][4] = bits((uint32_t)(on ? m.b_root[b] : 0));
    m.b_tab[b][5] = m.b_tab[b][6] = m.b_tab[b][7] = 0.0f;
  }
  return MIR_OK;
}

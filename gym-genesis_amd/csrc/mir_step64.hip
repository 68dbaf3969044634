// mir_step64.hip — the env.step() hot path for scenes beyond the 16-lane kernel: ONE WAVE PER ENV.
//
// What it replaces: scene.step() + get_obs() + compute_reward() of the reference's five-cube stack tasks
// (/root/reference/gym_genesis/tasks/franka/cube_stack_kitchen_batch.py:131-160,
//  /root/reference/gym_genesis/tasks/so101/cube_stack_batch.py:135-181; scene at tasks/utils.py:239-426,593-794):
// arm + five free cubes on the kitchen-island slab, 36-39 dofs, up to 48 contacts.  Same formulation as
// mir_step.hip: CRB mass matrix + RNE bias, PD torques, box/plane narrowphase, pyramidal soft
// contacts, primal Newton with exact line search, semi-implicit Euler.
//
// Mapping to CDNA4
//   * workgroup = one wave64 = one env.  The 64 lanes are four BLOCKS of 16 = the four DPP rows; a kinematic tree
//     never straddles a block (arm in block 0, two cubes per later block: mir_compile64.cpp), so the joint-space
//     inertia is block-diagonal: M rows are 16 wide, and M^-1 f is FOUR independent register-row Gauss-Jordan
//     solves running side by side on row_newbcast DPP (the 16-lane kernel's solver, unchanged).
//   * a contact touches at most two trees => at most two blocks: its Jacobian is stored as two dense 16-wide
//     segments (3 base rows n, t1, t2 each) tagged with their block ids, and every dof lane walks the contact list of
//     ITS block only (the four DPP rows side by side).
//   * the Newton Hessian H = M + J^T D J lives in registers: the 16 columns of the lane's own block always, the columns of
//     the other blocks only in steps where a contact couples two blocks (wave-uniform test).  Uncoupled: four DPP block
//     solves side by side.  Coupled: a rolled Gauss-Jordan over 64 register columns, pivot row by v_readlane, pivot column
//     by wave-uniform VGPR indexing, column blocks outside the pivot's component skipped.  In the single-step instantiation
//     the two cases are compiled separately, so the common one carries none of the coupled one's registers.
//   * lane i is also body i (< 32), geom i, candidate pair i (4 passes), contact i; box-box runs on a DPP row per pair
//     (mir_dev.h); wave-wide reductions are a DPP row reduction + 4 v_readlane.
//   * 40 KB of LDS per env (phase-aliased like the 16-lane kernel) -> 4 envs per CU; no scratch.
//   * three instantiations (single step / rollout loop / everything), as in mir_step.hip.  The single-step one runs TWO waves
//     per env (256-register budget: two waves per SIMD): wave 1 does the collision phase beside wave 0's dynamics, half of the
//     Jacobian segments, and the all-rows-active Newton Hessian beside wave 0's warm start and first gradient.  DESIGN.md section 10.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <stdint.h>

#include "mir_model64.h"
#include "mir_step64.h"

#define G 16
#include "mir_dev.h"
#define NL W64
#define NB K64_MAX_BODY
#define MAXC MIR_MAX_CONTACT
#define JSEG 52 /* floats per contact segment: 3 rows x 16 + 4 pad */
#define MSTR 20 /* row stride of the block-diagonal M rows in LDS */
#define STAMP(k) do { if (a.prof && blockIdx.x == 0 && (threadIdx.x & 63) == 0) a.prof[k] = __builtin_readcyclecounter(); } while (0)
static_assert(MAXC <= NL, "lane c owns contact c");
static_assert(MIR_MAX_GEOM <= NL && MIR_MAX_PAIR <= 4 * NL, "lane ownership of geoms / pairs");

namespace {

__device__ __forceinline__ float rl(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
__device__ __forceinline__ float wsum(float v) {  // all-reduce over the wave: row reduction, then the four row sums
  v = gsum(v);
  return (rl(v, 0) + rl(v, 16)) + (rl(v, 32) + rl(v, 48));
}
__device__ __forceinline__ float wmaxf(float v) {
  v = gmaxf(v);
  return fmaxf(fmaxf(rl(v, 0), rl(v, 16)), fmaxf(rl(v, 32), rl(v, 48)));
}

// Dense Gauss-Jordan over the 64 register rows (lane i = row i of the SPD matrix, b_i the right-hand side; on
// return b = x_i).  The pivot loop is ROLLED per block (a fully unrolled 64-pivot elimination is ~10 k instructions,
// more than the instruction cache, and with one wave per SIMD nothing hides the fetch misses): pivot K of block BLK
// only touches columns >= 16 BLK (row K is already zero to the left), so each block has a static column range; the
// own-row entry a[K] is picked by a wave-uniform switch.  One readlane + one fma per column: with
// f = (1 - 1/p) on the pivot row and a_iK / p elsewhere, a[j] -= f * pivotrow[j] both scales the pivot row and
// eliminates the others.  Padding lanes (identity rows) are skipped through the wave-uniform `act` mask.
__device__ __forceinline__ float rlv(float v, int src) {  // src wave-uniform (SGPR lane select)
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
typedef float v16f __attribute__((ext_vector_type(16)));
template <int BLK>
__device__ __forceinline__ void gj_block(v16f (&a)[4], float& b, int lane, uint64_t act, unsigned comp) {
#pragma nounroll
  for (int kk = 0; kk < 16; kk++) {
    const int K = __builtin_amdgcn_readfirstlane(16 * BLK + kk);
    if (!((act >> K) & 1ull)) continue;
    const float aK = a[BLK][__builtin_amdgcn_readfirstlane(kk)];  // column K of every row: wave-uniform register index (VGPR index mode)
    const float pk = rlv(aK, K);
    float inv = __builtin_amdgcn_rcpf(pk);

    const float f = lane == K ? 1.0f - inv : aK * inv;
    // column blocks outside this block's connected component hold zeros in the pivot row: skipped (wave-uniform);
    // pivot-row entries are fetched eight at a time so the readlane -> fma SGPR dependencies overlap
#pragma unroll
    for (int bj = BLK; bj < 4; bj++) {
      if (!((comp >> (4 * BLK + bj)) & 1u)) continue;
#pragma unroll
      for (int j0 = 0; j0 < 16; j0 += 8) {
        float r[8];
#pragma unroll
        for (int t = 0; t < 8; t++) r[t] = rlv(a[bj][j0 + t], K);
#pragma unroll
        for (int t = 0; t < 8; t++) a[bj][j0 + t] = fmaf(-f, r[t], a[bj][j0 + t]);
      }
    }
    b = fmaf(-f, rlv(b, K), b);
  }
}
// comp: bit 4 b + b' set when blocks b and b' are in one connected component of the contact coupling graph
__device__ __forceinline__ void gj_wave(v16f (&a)[4], float& b, int lane, uint64_t act, unsigned comp) {
  gj_block<0>(a, b, lane, act, comp);
  gj_block<1>(a, b, lane, act, comp);
  gj_block<2>(a, b, lane, act, comp);
  gj_block<3>(a, b, lane, act, comp);
}

struct Dyn64 {
  float cddq[NL][8];
  float cinert[NB][12], crb[NB][12];
  union {
    struct { float cvel[NB][8], cfrc[NB][8]; };
    struct { float lpos[NB][4], lquat[NB][4]; };  // forward-kinematics exchange: dead before the dynamics start
  };
};
struct Col64 {
  float gpos[MIR_MAX_GEOM][4], gquat[MIR_MAX_GEOM][4];
  int cand[NL];
  int cmap[NL];          // contact slot -> candidate * 8 + point index
  int ccount[NL];        // points found per candidate
  float stage[NL][8][4];  // narrowphase output per candidate pair: pos, dist
  float snorm[NL][4];
};
struct Con64 {
  float cpos[MAXC][4];    // pos, dist
  float cfrm[MAXC][12];   // normal, t1, t2 (4-padded)
  float cmeta[MAXC][4];   // mu, D, -k imp dist, b
  float cref[MAXC][8];    // reference points of body1 / body2 trees
  unsigned cmask[MAXC][4];  // lane masks of body1 (lo, hi), body2 (lo, hi)
  int cblk[MAXC][4];      // block of segment 0, of segment 1 (-1 = none), pad, pad
  float cfb[MAXC][4];     // per-iteration base forces (n, t1, t2), active-row flags
  int blist[4][MAXC];     // per block: the contacts that touch it, as contact * 2 + segment, ascending
  int bcount[4];
};
struct DynM64 {
  Dyn64 dyn;
  float M[NL][MSTR];  // block-diagonal: row of lane i holds the 16 columns of its own block (dead once mrow is loaded)
};
struct Env64 {
  float qpos[K64_QSTRIDE], qvel[NL], target[NL], qacc_ws[NL], qacc[NL];
  float qas[NL], srch[NL];
  float xpos[NB][4], xquat[NB][4];
  float cdof[NL][8];
  int parent[NB];
  int ncon, ncand, pad0, pad1;
  // two phase-local areas share storage: dynamics scratch + M (FK .. smooth solve) next to the collision scratch -- both
  // live at once in the single-step instantiation, where a second wave detects the contacts while this one runs the
  // dynamics -- and the contact Jacobian segments (written once both are dead)
  union {
    struct {
      DynM64 dm;
      Col64 col;
    };
    struct {
      float Jb[MAXC][2][JSEG];
      float hx3[NL - 48][G];  // (second piece of the Hessian hand-over, see hx)
    };
  };
  Con64 con;          // (before the contacts are finished: the box-box clipping exchange, 48 floats per DPP row)
  // the few model tables that are looked up with a DYNAMIC index inside the collision phases, staged once per launch:
  // an LDS round trip (~64 cycles) instead of an L2 one per dependent hop (there is no room for the whole model)
  union {
    struct {
      float gts[MIR_MAX_GEOM][4];           // type | body << 8 (as int bits), half extents
      float gfr[MIR_MAX_GEOM];              // friction
      unsigned short pairs[MIR_MAX_PAIR];   // g1 | g2 << 8
      float gsol[MIR_MAX_GEOM][8];          // solref[2], solimp[5]
      float btab[NB][8];                    // body_invweight0, dofmask lo, hi, block, root (contact finish)
    };
    // two-wave single-step instantiation only (one launch = one step, the tables are dead once the contacts are finished): the
    // Hessian rows wave 1 accumulates for wave 0, lanes 0..47 here and the rest in hx3
    float hx[48][G];
  };
};
static_assert(sizeof(float[48][G]) <= sizeof(float[MIR_MAX_GEOM][4]) + sizeof(float[MIR_MAX_GEOM]) + sizeof(unsigned short[MIR_MAX_PAIR]) +
                                          sizeof(float[MIR_MAX_GEOM][8]) + sizeof(float[NB][8]), "Hessian hand-over fits the staged tables");
static_assert(MIR_MAX_GEOM <= 256, "pair entries are 16 bits");
static_assert(sizeof(Con64) >= 4 * 48 * sizeof(float), "the box-box clipping exchange lives in the contact arrays");

struct BodyK64 {
  int jtype, qadr;
  V3 pos, axis;
  Q4 quat;
};

// forward kinematics by pointer jumping over the parent links (see group_fk in mir_step.hip): 5 rounds cover any tree
// of up to 32 bodies; the pointer travels in the w slot of the position
__device__ __forceinline__ void wave_fk(Env64& S, int lane, int nb, const BodyK64& k) {
  V3 P = v3(0, 0, 0);
  Q4 Qx = Q4{1, 0, 0, 0};
  int anc = 0;
  if (lane > 0 && lane < nb) {
    Qx = k.quat;
    P = k.pos;
    if (k.jtype == MIR_JNT_REVOLUTE) {
      float ang = S.qpos[k.qadr], sn, cs;
      sincos_pi2(0.5f * ang, &sn, &cs);
      Qx = qmul(k.quat, Q4{cs, k.axis.x * sn, k.axis.y * sn, k.axis.z * sn});
    } else if (k.jtype == MIR_JNT_PRISMATIC) {
      P = k.pos + qrot(k.quat, S.qpos[k.qadr] * k.axis);
    } else if (k.jtype == MIR_JNT_FREE) {
      P = ld3(&S.qpos[k.qadr]);
      Qx = qnormalize(ld4(&S.qpos[k.qadr + 3]));
    }
    anc = S.parent[lane];
  }
#pragma unroll 1
  for (int round = 0; round < 5; round++) {
    if (!__any(anc > 0)) break;
    if (lane < NB) {
      stv(S.dm.dyn.lpos[lane], f4{P.x, P.y, P.z, __int_as_float(anc)});
      st4v(S.dm.dyn.lquat[lane], Qx);
    }
    WSYNC();
    if (anc > 0) {
      const f4 pa = ldv(S.dm.dyn.lpos[anc]);
      const Q4 qa = ld4v(S.dm.dyn.lquat[anc]);
      P = v3(pa.x, pa.y, pa.z) + qrot(qa, P);
      Qx = qmul(qa, Qx);
      anc = __float_as_int(pa.w);
    }
    WSYNC();
  }
  if (lane < nb) {
    st3v(S.xpos[lane], P);
    st4v(S.xquat[lane], Qx);
  }
  WSYNC();
}

// dot of the three base rows of one Jacobian segment with a 16-float LDS vector
__device__ __forceinline__ void segdot3(const float* seg, const float* x, float& dn, float& d1, float& d2) {
  float an, a1, a2;
  jdot3(seg, x, an, a1, a2);
  dn += an; d1 += a1; d2 += a2;
}
// J_c x for contact c: both segments against their blocks of the 64-float LDS vector x
__device__ __forceinline__ void condot3(const Env64& S, int c, const float* x, float& dn, float& d1, float& d2) {
  dn = d1 = d2 = 0.0f;
  const int b0 = S.con.cblk[c][0], b1 = S.con.cblk[c][1];
  if (b0 >= 0) segdot3(&S.Jb[c][0][0], x + 16 * b0, dn, d1, d2);
  if (b1 >= 0) segdot3(&S.Jb[c][1][0], x + 16 * b1, dn, d1, d2);
}

// ---------------------------------------------------------------------------------------------
// SINGLE = one full step per launch without the rollout / autoreset / per-stage-output options: no step loop, hence none of
// the scalar-register spills the loop structure forces (see mir_step.hip).
// VARIANT 0 = SINGLE; 1 = the step loop of rollouts (packed rows only, no per-stage / separate outputs); 2 = everything.
// DUAL (the single-step instantiation): the workgroup is TWO waves on one env.  Wave 1 stages the model tables, then -- once
// wave 0 has the body poses -- runs the whole collision phase (geom poses, broadphase, plane-box, box-box, contact finish, per-block
// lists) in its own scratch while wave 0 runs the dynamics up to the smooth solve; they meet before the Jacobian segments are
// written, and wave 1 retires.  The register budget is held at 256 so that the four workgroups of a CU (LDS) are two waves per SIMD.
template <int VARIANT>
__global__ __launch_bounds__(VARIANT == 0 ? 128 : 64) __attribute__((amdgpu_waves_per_eu(VARIANT == 0 ? 2 : 1, VARIANT == 0 ? 2 : 1)))
void mir_step64_kernel(StepArgs64 a) {
  constexpr bool SINGLE = VARIANT == 0;
  constexpr bool DUAL = SINGLE;
  __shared__ __attribute__((aligned(16))) Env64 S;
  const DevModel64* __restrict__ m = a.model;
  const int lane = threadIdx.x & 63;
  const bool helper = DUAL && threadIdx.x >= 64;  // wave-uniform
  const int blk = lane >> 4, l16 = lane & 15;
  const int env = blockIdx.x;  // grid = B exactly

  const int nb = m->nbody, nv = m->nv, nq = m->nq;
  const int ngeom = m->ngeom, npair = m->npair, max_contacts = m->max_contacts, enable_collision = m->enable_collision;
  const float dt = m->dt;
  const uint64_t lanemask = m->lanemask;

  // ---- per-lane model constants (lane = body = dof slot) ------------------------------------------
  const bool isbody = lane < nb && lane > 0;
  const bool isdof = (lanemask >> lane) & 1ull;
  const int bl = lane < NB ? lane : 0;  // body index this lane may own
  BodyK64 bk;
  bk.jtype = m->b_jtype[bl]; bk.qadr = m->b_qadr[bl];
  bk.pos = ld3(m->b_pos[bl]); bk.axis = ld3(m->b_axis[bl]); bk.quat = ld4(m->b_quat[bl]);
  const int b_root = m->b_root[bl];
  const uint64_t b_dofmask = m->b_dofmask[bl];
  const uint32_t b_submask = m->b_submask[bl];
  const V3 b_ipos = ld3(m->b_ipos[bl]);
  const float b_mass = m->b_mass[bl];
  float ib[6];
#pragma unroll
  for (int k = 0; k < 6; k++) ib[k] = m->b_inertia[bl][k];
  const int d_body = isdof ? m->d_body[lane] : 0;  // (lanemask is a scalar; the load itself is unconditional in effect)
  const int d_kind = m->d_kind[lane], d_qadr = m->d_qadr[lane], d_axis_k = m->d_axis_k[lane];
  const int d_root = m->d_root[lane];
  const V3 d_axis = ld3(m->d_axis[lane]);
  const uint64_t d_premask = m->d_premask[lane], d_ancmask = m->d_ancmask[lane];
  const uint32_t d_submask = m->d_bsubmask[lane];
  const int d_ctrl = m->d_ctrl[lane], d_uadr = m->d_uadr[lane];
  const bool d_limited = isdof && m->d_limited[lane] && m->enable_joint_limit;
  const float d_damping = m->d_damping[lane], d_kp = m->d_kp[lane], d_kv = m->d_kv[lane];
  const float d_frclo = m->d_frclo[lane], d_frchi = m->d_frchi[lane], d_mdiag = m->d_mdiag[lane];
  const int d_qbase = m->d_qbase[lane], d_lbase = m->d_lbase[lane];
  const float d_lo = m->d_lo[lane], d_hi = m->d_hi[lane];
  // joint-limit row constants (used only while the limit is violated, but then inside the step's critical path)
  const float d_iw0 = m->d_invweight0[lane], d_lk = m->d_k[lane], d_lb = m->d_b[lane];
  const float d_si0 = m->d_solimp[lane][0], d_si1 = m->d_solimp[lane][1], d_si2 = m->d_solimp[lane][2], d_si3 = m->d_solimp[lane][3], d_si4 = m->d_solimp[lane][4];
  const int obs_qadr = m->obs_qadr[lane];
  // lane = geom: frame in its body, staged tables
  const int gl = lane < ngeom ? lane : 0;
  const int g_bodyl = m->g_body[gl];
  const V3 g_posl = ld3(m->g_pos[gl]);
  const Q4 g_quatl = ld4(m->g_quat[gl]);
  // Every global read of the launch -- staged tables, state rows, action, cached poses -- is issued before the first LDS
  // store: one L2 round trip at the start of the launch instead of one per group of stores (see mir_step.hip).
  const int gi = lane < ngeom ? lane : 0, bi = lane < NB ? lane : 0;
  const int gt_in = m->g_type[gi];
  const float gsx = m->g_size[gi][0], gsy = m->g_size[gi][1], gsz = m->g_size[gi][2], gfr_in = m->g_pos[gi][3];
  static_assert(MIR_MAX_PAIR <= 4 * NL, "pair list: at most four entries per lane");
  int pr_in[4];
#pragma unroll
  for (int u = 0; u < 4; u++) pr_in[u] = m->pair[lane + NL * u < npair ? lane + NL * u : 0];
  const f4 gs0 = reinterpret_cast<const f4*>(m->g_sol[gi])[0], gs1 = reinterpret_cast<const f4*>(m->g_sol[gi])[1];
  const f4 bt0 = reinterpret_cast<const f4*>(m->b_tab[bi])[0], bt1 = reinterpret_cast<const f4*>(m->b_tab[bi])[1];
  const int par_in = m->b_parent[bi];
  float q_in = 0.0f, qv_in = 0.0f, ws_in = 0.0f, tg = 0.0f, au = 0.0f;
  f4 cpos = {0, 0, 0, 0}, cquat = {0, 0, 0, 0};
  bool cached = false;
  if (!helper) {
    q_in = a.qpos[(size_t)env * K64_QSTRIDE + lane]; qv_in = a.qvel[(size_t)env * NL + lane]; ws_in = a.qacc_ws[(size_t)env * NL + lane];
    tg = a.target[(size_t)env * NL + lane];
    au = (a.action && lane < a.nu) ? a.action[(size_t)env * a.nu + lane] : 0.0f;
    // (the cached poses travel with their validity flag; (B, 2, 32, 4): the speculative read is in bounds)
    const float* pose_p = a.poses + ((size_t)env * 2 * NB + (lane & (NB - 1))) * 4;
    cpos = *reinterpret_cast<const f4*>(pose_p); cquat = *reinterpret_cast<const f4*>(pose_p + 4 * NB);
    cached = a.fkvalid[env] != 0;  // wave-uniform
  }
  __builtin_amdgcn_sched_barrier(0);  // (nothing below may move in front of the loads above)
  if (!DUAL || helper) {
    if (lane < ngeom) {
      stv(S.gts[lane], f4{__int_as_float(gt_in | (g_bodyl << 8)), gsx, gsy, gsz});
      S.gfr[lane] = gfr_in;
      stv(&S.gsol[lane][0], gs0); stv(&S.gsol[lane][4], gs1);
    }
#pragma unroll
    for (int u = 0; u < 4; u++)
      if (lane + NL * u < npair) S.pairs[lane + NL * u] = (unsigned short)pr_in[u];
    if (lane < NB) { stv(&S.btab[lane][0], bt0); stv(&S.btab[lane][4], bt1); }
  }
  if (!helper && lane < NB) S.parent[lane] = lane < nb ? par_in : 0;

  // ======================= collision detection ================================================
  // From the body poses to the finished contact arrays and the per-block contact lists; leaves S.ncon / S.ncand.  Touches only
  // its own scratch (S.col), the contact arrays (S.con) and read-only state, so in the single-step instantiation it runs on
  // wave 1 next to the dynamics.
  auto collide = [&]() {
    if (lane == 0) { S.ncon = 0; S.ncand = 0; }
    if (lane < ngeom) {
      Q4 qb = ld4v(S.xquat[g_bodyl]);
      st3v(S.col.gpos[lane], ld3v(S.xpos[g_bodyl]) + qrot(qb, g_posl));
      st4v(S.col.gquat[lane], qmul(qb, g_quatl));
    }
    S.col.ccount[lane] = 0;
    WSYNC();
    int mycount = 0;
    int ncand = 0;
    if (enable_collision) {
      // broadphase: bounding test per static candidate pair, ordered compaction of survivors (lane = pair)
      int base = 0;
      for (int p0 = 0; p0 < npair; p0 += NL) {
        int p = p0 + lane;
        bool hit = false;
        if (p < npair) {
          const int pr = (int)S.pairs[p];
          const int g1 = pr & 255, g2 = pr >> 8;
          V3 h2 = ld3(&S.gts[g2][1]);
          M3 R2 = q2m(ld4v(S.col.gquat[g2]));
          V3 c2 = ld3v(S.col.gpos[g2]);
          if ((__float_as_int(S.gts[g1][0]) & 255) == MIR_GEOM_PLANE) {
            V3 n = mcol(q2m(ld4v(S.col.gquat[g1])), 2);
            float ext = h2.x * fabsf(dot(n, mcol(R2, 0))) + h2.y * fabsf(dot(n, mcol(R2, 1))) + h2.z * fabsf(dot(n, mcol(R2, 2)));
            hit = dot(c2 - ld3v(S.col.gpos[g1]), n) - ext < 0.0f;
          } else {
            V3 h1 = ld3(&S.gts[g1][1]);
            float rs = sqrtf(dot(h1, h1)) + sqrtf(dot(h2, h2));
            V3 dc = c2 - ld3v(S.col.gpos[g1]);
            hit = dot(dc, dc) <= rs * rs;
            if (hit) {
              // the six face axes of the narrowphase's separating-axis test (same expressions): a pair they separate
              // would come back with zero contacts, and the narrowphase walks its candidates four at a time
              const M3 R1 = q2m(ld4v(S.col.gquat[g1]));
              const V3 A0 = mcol(R1, 0), A1 = mcol(R1, 1), A2 = mcol(R1, 2), B0 = mcol(R2, 0), B1 = mcol(R2, 1), B2 = mcol(R2, 2);
              const V3 Ls[6] = {A0, A1, A2, B0, B1, B2};
#pragma unroll
              for (int c = 0; c < 6; c++) {
                const V3 L = Ls[c];
                const float ra = h1.x * fabsf(dot(A0, L)) + h1.y * fabsf(dot(A1, L)) + h1.z * fabsf(dot(A2, L));
                const float rb = h2.x * fabsf(dot(B0, L)) + h2.y * fabsf(dot(B1, L)) + h2.z * fabsf(dot(B2, L));
                if (fabsf(dot(dc, L)) - (ra + rb) > 0.0f) hit = false;
              }
            }
          }
        }
        const unsigned long long bal = __ballot(hit);
        int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (hit && pos < NL) S.col.cand[pos] = p;
        base += __popcll(bal);
      }
      ncand = base < NL ? base : NL;
      if (lane == 0) S.ncand = ncand;
      WSYNC();
      STAMP(6);
      // narrowphase, plane-box: DPP row r takes candidates r, r + 4, ...; the 8 box corners on lanes 0..7 of the row
      for (int k0 = 0; k0 < ncand; k0 += 4) {
        const int k = k0 + blk;
        const bool act = k < ncand;
        const int pr = act ? (int)S.pairs[S.col.cand[k]] : 0;
        const int g1 = pr & 255, g2 = pr >> 8;
        const bool isplane = act && (__float_as_int(S.gts[g1][0]) & 255) == MIR_GEOM_PLANE;
        if (!__any(isplane)) continue;
        const M3 Rp = q2m(ld4v(S.col.gquat[g1]));
        const V3 n = mcol(Rp, 2), eu = mcol(Rp, 0), ev = mcol(Rp, 1);
        const M3 R2 = q2m(ld4v(S.col.gquat[g2]));
        const V3 h = ld3(&S.gts[g2][1]);
        const int c = lane & 7;
        const V3 w = ld3v(S.col.gpos[g2]) + ((c & 1) ? h.x : -h.x) * mcol(R2, 0) + ((c & 2) ? h.y : -h.y) * mcol(R2, 1) +
                     ((c & 4) ? h.z : -h.z) * mcol(R2, 2);
        const V3 rel = w - ld3v(S.col.gpos[g1]);
        const float d = dot(rel, n), u = dot(rel, eu), v = dot(rel, ev);
        const bool pen = isplane && l16 < 8 && d < 0.0f;
        const uint32_t penm = (uint32_t)(__ballot(pen) >> (blk * G)) & 0xffu;
        const int cnt = __popc(penm);
        // support extremes (+u, -u, +v, -v; lowest corner index wins ties), needed only when more than 4 corners penetrate
        // somewhere in the wave (a box lying flat has exactly 4: the reductions are skipped)
        uint32_t ext = 0u;
        if (__any(cnt > 4)) {
          const float uM = gmaxf(pen ? u : -3e38f), um = -gmaxf(pen ? -u : -3e38f);
          const float vM = gmaxf(pen ? v : -3e38f), vm = -gmaxf(pen ? -v : -3e38f);
          const uint32_t e0 = (uint32_t)(__ballot(pen && u == uM) >> (blk * G)) & 0xffu, e1 = (uint32_t)(__ballot(pen && u == um) >> (blk * G)) & 0xffu;
          const uint32_t e2 = (uint32_t)(__ballot(pen && v == vM) >> (blk * G)) & 0xffu, e3 = (uint32_t)(__ballot(pen && v == vm) >> (blk * G)) & 0xffu;
          ext = (e0 & -e0) | (e1 & -e1) | (e2 & -e2) | (e3 & -e3);
        }
        const uint32_t keepm = cnt <= 4 ? penm : ext;
        const bool keep = (keepm >> l16 & 1u) && l16 < 8;
        const int slot = __popc(keepm & ((1u << l16) - 1u));
        if (isplane && keep && slot < 4) {
          const V3 pos = w - (0.5f * d) * n;
          stv(S.col.stage[k][slot], f4{pos.x, pos.y, pos.z, d});
        }
        if (isplane && l16 == 0) {
          S.col.ccount[k] = min(__popc(keepm), 4);
          st3v(S.col.snorm[k], n);
        }
      }
      WSYNC();
      mycount = S.col.ccount[lane];
      STAMP(7);
      // narrowphase, box-box: DPP row r takes candidates r, r + 4, ... (like plane-box).  The 15 separating axes sit on
      // lanes 0..14 of the row, the incident-face vertices on lanes 0..3 (box_box_row, mir_dev.h)
      for (int k0 = 0; k0 < ncand; k0 += 4) {
        const int k = k0 + blk;
        const bool actk = k < ncand;
        const int pr = actk ? (int)S.pairs[S.col.cand[k]] : 0;
        const int g1 = pr & 255, g2 = pr >> 8;
        const bool isbox = actk && (__float_as_int(S.gts[g1][0]) & 255) != MIR_GEOM_PLANE;
        if (!__any(isbox)) continue;
        if (isbox) {  // whole rows
          const M3 R1 = q2m(ld4v(S.col.gquat[g1])), R2 = q2m(ld4v(S.col.gquat[g2]));
          const BoxG A = {ld3v(S.col.gpos[g1]), mcol(R1, 0), mcol(R1, 1), mcol(R1, 2), ld3(&S.gts[g1][1])};
          const BoxG B = {ld3v(S.col.gpos[g2]), mcol(R2, 0), mcol(R2, 1), mcol(R2, 2), ld3(&S.gts[g2][1])};
          const int cnt = box_box_row(A, B, l16, lane, blk * G, S.col.stage[k], S.col.snorm[k], reinterpret_cast<float*>(&S.con) + 48 * blk);  // (contact arrays: not written yet)
          if (l16 == 0) S.col.ccount[k] = cnt;
        }
      }
      WSYNC();
      mycount = S.col.ccount[lane];
    }
    STAMP(8);
    // ordered compaction of the contact points: exclusive prefix over candidate lanes (convergent code)
    const int maxc = max_contacts < MAXC ? max_contacts : MAXC;
    int ncon;
    {
      // more candidate points than the capacity: the largest manifolds are thinned before any pair loses all of its points (the
      // rule is defined at oracle/orc_rigid.c: thin_manifolds; same code path as in the 16-lane kernel, one env per wave here)
      int total0 = (int)wsum((float)mycount);
      if (total0 > maxc) {
        for (int round = 0; round < 8; round++) {  // (a manifold holds at most 8 points)
          const int mx = (int)wmaxf((float)mycount);
          if (!(total0 > maxc && mx > 1)) break;
          const bool is = mycount == mx;
          const unsigned long long bm = __ballot(is);
          const int rank = __popcll(bm & ((1ull << lane) - 1ull)), nis = __popcll(bm), need = total0 - maxc;
          if (is && rank < need) {
            const f4 pa = ldv(S.col.stage[lane][mx - 2]), pb = ldv(S.col.stage[lane][mx - 1]);
            stv(S.col.stage[lane][mx - 2], f4{0.5f * (pa.x + pb.x), 0.5f * (pa.y + pb.y), 0.5f * (pa.z + pb.z), 0.5f * (pa.w + pb.w)});
            mycount--;
          }
          total0 -= need < nis ? need : nis;
        }
        WSYNC();
      }
    }
    {
      float inclf = (float)mycount;
      inclf += row_shr<1>(inclf);
      inclf += row_shr<2>(inclf);
      inclf += row_shr<4>(inclf);
      inclf += row_shr<8>(inclf);
      const float r0 = rl(inclf, 15), r1 = rl(inclf, 31), r2 = rl(inclf, 47), r3 = rl(inclf, 63);
      const float rowbase = blk == 0 ? 0.0f : (blk == 1 ? r0 : (blk == 2 ? r0 + r1 : r0 + r1 + r2));
      const int incl = (int)(inclf + rowbase);
      const int off = incl - mycount;
      const int total = (int)(r0 + r1 + r2 + r3);
      ncon = total < maxc ? total : maxc;
      if (lane == 0) S.ncon = ncon;
      for (int c = 0; c < mycount; c++)
        if (off + c < maxc) S.col.cmap[off + c] = lane * 8 + c;
      WSYNC();
      // every contact is finished by its own lane (staging lives in col scratch, disjoint from the contact arrays)
      if (lane < ncon) {
        const int k = lane;
        const int mp = S.col.cmap[k];
        const int cl = mp >> 3, ci = mp & 7;
        const int pr = (int)S.pairs[S.col.cand[cl]];
        const int g1 = pr & 255, g2 = pr >> 8;
        const V3 n = ld3v(S.col.snorm[cl]);
        V3 t1 = fabsf(n.y) < 0.5f ? v3(0, 1, 0) : v3(0, 0, 1);  // same frame construction as the oracle
        t1 = t1 - dot(n, t1) * n;
        t1 = __builtin_amdgcn_rsqf(dot(t1, t1)) * t1;
        const V3 t2 = cross(n, t1);
        const float mu = fmaxf(S.gfr[g1], S.gfr[g2]);
        const float* s1 = S.gsol[g1];
        const float* s2 = S.gsol[g2];
        const float sr0 = 0.5f * (s1[0] + s2[0]), sr1 = 0.5f * (s1[1] + s2[1]);
        const float si[5] = {0.5f * (s1[2] + s2[2]), 0.5f * (s1[3] + s2[3]), 0.5f * (s1[4] + s2[4]), 0.5f * (s1[5] + s2[5]), 0.5f * (s1[6] + s2[6])};
        const int b1 = __float_as_int(S.gts[g1][0]) >> 8, b2 = __float_as_int(S.gts[g2][0]) >> 8;
        const f4 bt1 = ldv(&S.btab[b1][0]), bt2 = ldv(&S.btab[b2][0]);
        const float wsumw = bt1.x + bt2.x;
        const float dmax = fminf(fmaxf(si[1], 1e-4f), 0.9999f);
        const float tc = fmaxf(sr0, 2.0f * dt);
        const float kk = 1.0f / (dmax * dmax * tc * tc * sr1 * sr1), bb = 2.0f / (dmax * tc);
        const uint64_t dm1 = (uint64_t)__float_as_uint(bt1.y) | ((uint64_t)__float_as_uint(bt1.z) << 32);
        const uint64_t dm2 = (uint64_t)__float_as_uint(bt2.y) | ((uint64_t)__float_as_uint(bt2.z) << 32);
        const int k1 = __float_as_int(bt1.w), k2 = __float_as_int(bt2.w);
        const int sg0 = k1 >= 0 ? k1 : k2, sg1 = (k1 >= 0 && k2 >= 0 && k2 != k1) ? k2 : -1;
        const V3 ref1 = ld3v(S.xpos[__float_as_int(S.btab[b1][4])]), ref2 = ld3v(S.xpos[__float_as_int(S.btab[b2][4])]);
        const f4 pd = ldv(S.col.stage[cl][ci]);
        const float dist = pd.w;
        stv(S.con.cpos[k], pd);
        st3v(&S.con.cfrm[k][0], n); st3v(&S.con.cfrm[k][4], t1); st3v(&S.con.cfrm[k][8], t2);
        const float imp = impedance(si[0], si[1], si[2], si[3], si[4], dist);
        const float Rr = fmaxf(2.0f * mu * mu * (1.0f - imp) / imp * wsumw * (1.0f + mu * mu), 1e-15f);
        stv(S.con.cmeta[k], f4{mu, 1.0f / Rr, -kk * imp * dist, bb});
        st3v(&S.con.cref[k][0], ref1); st3v(&S.con.cref[k][4], ref2);
        S.con.cmask[k][0] = (unsigned)dm1; S.con.cmask[k][1] = (unsigned)(dm1 >> 32);
        S.con.cmask[k][2] = (unsigned)dm2; S.con.cmask[k][3] = (unsigned)(dm2 >> 32);
        S.con.cblk[k][0] = sg0; S.con.cblk[k][1] = sg1; S.con.cblk[k][2] = 0; S.con.cblk[k][3] = 0;
      }
    }
    WSYNC();  // col scratch is dead from here on
    // per-block contact lists (lane = contact; ordered ballot compaction): every later loop of a dof lane runs over
    // the contacts that touch ITS block only, and the four blocks (DPP rows) walk their lists side by side
    {
      const bool isc = lane < ncon;
      const int s0 = isc ? S.con.cblk[lane][0] : -1, s1 = isc ? S.con.cblk[lane][1] : -1;
#pragma unroll
      for (int bq = 0; bq < 4; bq++) {
        const bool touch = isc && (s0 == bq || s1 == bq);
        const unsigned long long bal = __ballot(touch);
        if (touch) S.con.blist[bq][__popcll(bal & ((1ull << lane) - 1ull))] = lane * 2 + (s0 == bq ? 0 : 1);
        if (lane == 0) S.con.bcount[bq] = __popcll(bal);
      }
    }
    WSYNC();
  };
  auto jacobians = [&](int nmine, int first, int stride) {
    // contact base Jacobians: lane = dof writes its entry of the segment its block owns (zeros included, so a
    // segment is always fully defined)
    // (cdof of the lane is loop-invariant; every read of a contact is issued in one batch ahead of the arithmetic, and a
    // dof that moves neither body simply ends with sgn = 0: no divergent branch around the reads)
    const V3 cd_ang = ld3v(&S.cdof[lane][0]), cd_lin = ld3v(&S.cdof[lane][4]);
    for (int k0 = first; k0 < nmine; k0 += stride) {  // two list entries per trip
      const int2 e2 = *reinterpret_cast<const int2*>(&S.con.blist[blk][k0]);
      const int eqA = e2.x, eqB = k0 + 1 < nmine ? e2.y : e2.x;
      const int cA = eqA >> 1, cB = eqB >> 1;
      const f4 mkA = ldv(reinterpret_cast<const float*>(S.con.cmask[cA])), mkB = ldv(reinterpret_cast<const float*>(S.con.cmask[cB]));
      const f4 cpA = ldv(S.con.cpos[cA]), r1A = ldv(&S.con.cref[cA][0]), r2A = ldv(&S.con.cref[cA][4]);
      const f4 cpB = ldv(S.con.cpos[cB]), r1B = ldv(&S.con.cref[cB][0]), r2B = ldv(&S.con.cref[cB][4]);
      const f4 fnA = ldv(&S.con.cfrm[cA][0]), f1A = ldv(&S.con.cfrm[cA][4]), f2A = ldv(&S.con.cfrm[cA][8]);
      const f4 fnB = ldv(&S.con.cfrm[cB][0]), f1B = ldv(&S.con.cfrm[cB][4]), f2B = ldv(&S.con.cfrm[cB][8]);
      __builtin_amdgcn_sched_barrier(0);
#define MIR_JCOL64(mk, cp, r1, r2, fn, f1, f2, eq)                                                                  \
      {                                                                                                             \
        const uint64_t dm1 = (uint64_t)__float_as_uint(mk.x) | ((uint64_t)__float_as_uint(mk.y) << 32);             \
        const uint64_t dm2 = (uint64_t)__float_as_uint(mk.z) | ((uint64_t)__float_as_uint(mk.w) << 32);             \
        const bool in2 = dm2 >> lane & 1ull, in1 = dm1 >> lane & 1ull;                                              \
        const float sgn = (in2 ? 1.0f : 0.0f) - (in1 ? 1.0f : 0.0f); /* a dof moving both bodies cancels */         \
        const V3 r = v3(cp.x, cp.y, cp.z) - (in2 ? v3(r2.x, r2.y, r2.z) : v3(r1.x, r1.y, r1.z));                    \
        const V3 vel = cross(cd_ang, r) + cd_lin;                                                                   \
        float* jb = &S.Jb[(eq) >> 1][(eq) & 1][0];                                                                  \
        /* (selects, not products: a lane that carries no dof holds stale LDS in cd_ang / cd_lin, and 0 x NaN is NaN) */       \
        jb[l16] = sgn != 0.0f ? sgn * dot(vel, v3(fn.x, fn.y, fn.z)) : 0.0f;                                        \
        jb[16 + l16] = sgn != 0.0f ? sgn * dot(vel, v3(f1.x, f1.y, f1.z)) : 0.0f;                                   \
        jb[32 + l16] = sgn != 0.0f ? sgn * dot(vel, v3(f2.x, f2.y, f2.z)) : 0.0f;                                   \
      }
      MIR_JCOL64(mkA, cpA, r1A, r2A, fnA, f1A, f2A, eqA)
      if (k0 + 1 < nmine) MIR_JCOL64(mkB, cpB, r1B, r2B, fnB, f1B, f2B, eqB)
#undef MIR_JCOL64
    }
  };
  // J^T D J of the lane's own block with EVERY pyramid row of every contact active (lane = dof row, 16 columns), from zero in list
  // order.  The Newton loop starts its incremental Hessian from M + this (resting contacts have all four rows active; the rows
  // that are not come off in the first incremental update), so with two waves it is accumulated by wave 1 while wave 0 evaluates
  // the constraint rows, the warm start and the first gradient.
  auto hess_full = [&](float (&hp)[G], int nmine) {
#pragma unroll
    for (int j = 0; j < G; j++) hp[j] = 0.0f;
    for (int kq = 0; kq < nmine; kq++) {
      const int eq = S.con.blist[blk][kq];
      const int c = eq >> 1;
      const float* seg = &S.Jb[c][eq & 1][0];
      const float jn = seg[l16], j1 = seg[16 + l16], j2 = seg[32 + l16];
      const f4 mt = ldv(S.con.cmeta[c]);
      f4 xn[4], x1[4], x2[4];
#pragma unroll
      for (int q = 0; q < 4; q++) { xn[q] = ldv(seg + 4 * q); x1[q] = ldv(seg + 16 + 4 * q); x2[q] = ldv(seg + 32 + 4 * q); }
      __builtin_amdgcn_sched_barrier(0);  // (the whole batch of reads ahead of the arithmetic: one LDS round trip)
      const float mu = mt.x, D = mt.y;
      const float tn = jn * (4.0f * D), t1 = j1 * (mu * mu * (2.0f * D)), t2 = j2 * (mu * mu * (2.0f * D));
#pragma unroll
      for (int q = 0; q < 4; q++) {
        hp[4 * q + 0] += tn * xn[q].x + t1 * x1[q].x + t2 * x2[q].x;
        hp[4 * q + 1] += tn * xn[q].y + t1 * x1[q].y + t2 * x2[q].y;
        hp[4 * q + 2] += tn * xn[q].z + t1 * x1[q].z + t2 * x2[q].z;
        hp[4 * q + 3] += tn * xn[q].w + t1 * x1[q].w + t2 * x2[q].w;
      }
    }
  };
  float* const hxrow = lane < 48 ? &S.hx[lane][0] : &S.hx3[lane - 48][0];
  if (helper) {
#ifndef MIR_PROFILE_SINGLE
    a.prof = nullptr;
#endif
    STAMP(24);
    __syncthreads();  // (1) staged tables (this wave), body poses (wave 0)
    STAMP(25);
    collide();
    STAMP(26);
    __syncthreads();  // (2) wave 0 is done with the dynamics scratch: the Jacobian segments may overwrite it
    STAMP(27);
    const int nmine = S.con.bcount[blk];
    jacobians(nmine, 2, 4);
    STAMP(28);
    __syncthreads();  // (3)
    float hp[G];
    hess_full(hp, nmine);
#pragma unroll
    for (int q = 0; q < 4; q++) stv(hxrow + 4 * q, f4{hp[4 * q], hp[4 * q + 1], hp[4 * q + 2], hp[4 * q + 3]});
    STAMP(29);
    __syncthreads();  // (4)
    return;
  }

  STAMP(0);
  // ---- state -------------------------------------------------------------------------------------
  S.qpos[lane] = q_in;
  S.qvel[lane] = qv_in;
  S.qacc_ws[lane] = ws_in;
  if (a.action) {  // lane u fetched action component u; the dof it drives picks it up across the wave
    const float mine = __shfl(au, d_uadr >= 0 ? d_uadr : 0);
    if (isdof && d_uadr >= 0) tg = mine;
  }
  S.target[lane] = tg;
  WSYNC();

  // ======================= forward kinematics (FK cache as in the 16-lane kernel) ====================
  {
    if (cached) {
      if (lane < nb) {
        stv(S.xpos[lane], cpos);
        stv(S.xquat[lane], cquat);
      }
      WSYNC();
    } else {
      wave_fk(S, lane, nb, bk);
    }
  }
  if (DUAL) __syncthreads();  // (1)
  STAMP(1);
  const int nsteps = SINGLE ? 1 : (a.mode == 0 ? a.n_steps : (a.mode == 1 ? 1 : 0));
  if (SINGLE) { a.mode = 0; a.act_step = 0; a.rows_step = 0; a.ar.episode_len = nullptr; a.out_M = a.out_bias = a.out_qas = a.out_qacc = a.out_xpos = a.out_xquat = nullptr;
#ifndef MIR_PROFILE_SINGLE  /* (a profiling build keeps the phase stamps in the single-step instantiation: tools/phase_profile64.py) */
    a.prof = nullptr;
#endif
  }
  if (VARIANT == 1) { a.mode = 0; a.out_M = a.out_bias = a.out_qas = a.out_qacc = a.out_xpos = a.out_xquat = nullptr; a.prof = nullptr; a.agent_pos = a.env_state = a.reward = nullptr; a.terminated = a.term_host = nullptr; }
  // packed output row [agent_pos | env_state | reward | terminated] of the current kinematic state
  const int eb = m->eef_body, ob = m->obj_body, ob2 = m->obj2_body;
  const int ad = m->agent_dim, ed = m->env_dim;
  auto reward_of = [&](const V3 po, const V3 p2) -> float {
    if (m->reward_mode == MIR_REWARD_STACK) {
      const float dx = po.x - p2.x, dy = po.y - p2.y;
      return (sqrtf(dx * dx + dy * dy) < m->reward_xy && po.z - p2.z > m->reward_dz) ? 1.0f : 0.0f;
    }
    return po.z > m->reward_z ? 1.0f : 0.0f;
  };
  auto reward_now = [&]() -> float { return reward_of(ld3v(S.xpos[ob]), ld3v(S.xpos[ob2 >= 0 ? ob2 : ob])); };
  // (free objects hanging off the world: the same positions straight from qpos, before any forward kinematics)
  const bool term_early = m->term_early != 0;
  const int obj_qadr = m->obj_qadr, obj2_qadr = m->obj2_qadr;
  auto column = [&](int c) -> float {
    if (c < ad) {
      // (every caller asks for its own lane's column: the qpos address comes from the lane constants, no model trip)
      if (m->agent_mode == MIR_AGENT_QPOS) return S.qpos[c == lane ? obs_qadr : m->arm_qadr[c]];
      if (c < 3) return S.xpos[eb][c];
      if (c < 7) return S.xquat[eb][c - 3];
      return S.qpos[c == lane ? obs_qadr : m->grip_qadr[c - 7]];
    }
    const int k = c - ad;
    if (k < 3) return S.xpos[ob][k];
    if (k < 7) return S.xquat[ob][k - 3];
    const V3 df = ld3v(S.xpos[eb]) - ld3v(S.xpos[ob]);
    if (k < 10) return k == 7 ? df.x : (k == 8 ? df.y : df.z);
    if (k == 10) return sqrtf(dot(df, df));
    if (k < ed) return S.xpos[ob2][k - 11];
    return reward_now();  // k == ed reward, k == ed + 1 terminated
  };
  int eplen = a.ar.episode_len ? a.ar.episode_len[env] : 0, epcur = a.ar.episode_len ? a.ar.cursor[env] : 0;
  for (int step = 0; step < nsteps; step++) {
    // rollout mode (mir_rollout): a fresh action block per step
    if (step > 0 && a.action && a.act_step) {
      if (isdof && d_uadr >= 0) S.target[lane] = a.action[(size_t)step * a.act_step + (size_t)env * a.nu + d_uadr];
    }
    // motion subspaces (lane = dof) and body inertias about the tree reference point (lane = body)
    if (isdof) {
      V3 ang = v3(0, 0, 0), lin = v3(0, 0, 0);
      V3 r = ld3v(S.xpos[d_root]) - ld3v(S.xpos[d_body]);
      if (d_kind < 2) {
        V3 ax = qrot(ld4v(S.xquat[d_body]), d_axis);
        if (d_kind == 0) { ang = ax; lin = cross(ax, r); }
        else lin = ax;
      } else {
        V3 e = v3(d_axis_k == 0, d_axis_k == 1, d_axis_k == 2);
        if (d_kind == 2) lin = e;
        else { ang = e; lin = cross(e, r); }
      }
      st3v(&S.cdof[lane][0], ang);
      st3v(&S.cdof[lane][4], lin);
    } else {
      st3v(&S.cdof[lane][0], v3(0, 0, 0));
      st3v(&S.cdof[lane][4], v3(0, 0, 0));
    }
    if (lane < NB) {
      float* c = S.dm.dyn.cinert[lane];
      if (isbody) {
        M3 R = q2m(ld4v(S.xquat[lane]));
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
        V3 r = ld3v(S.xpos[lane]) + mmul(R, b_ipos) - ld3v(S.xpos[b_root]);
        float rr = dot(r, r);
        stv(c, f4{b_mass, b_mass * r.x, b_mass * r.y, b_mass * r.z});
        stv(c + 4, f4{W[0][0] + b_mass * (rr - r.x * r.x), W[1][1] + b_mass * (rr - r.y * r.y), W[2][2] + b_mass * (rr - r.z * r.z),
                      W[0][1] - b_mass * r.x * r.y});
        stv(c + 8, f4{W[0][2] - b_mass * r.x * r.z, W[1][2] - b_mass * r.y * r.z, 0.0f, 0.0f});
      } else {
        stv(c, f4{0, 0, 0, 0}); stv(c + 4, f4{0, 0, 0, 0}); stv(c + 8, f4{0, 0, 0, 0});
      }
    }
    WSYNC();

    STAMP(2);
    // ======================= velocities, composite inertias =====================================
    {
      if (isdof) {  // lane = dof: cdof_dot * qvel, "velocity before this dof" from the pre-mask
        V3 pw = v3(0, 0, 0), pv = v3(0, 0, 0);
        uint64_t mk = d_premask;
        while (mk) {
          int j = __ffsll((unsigned long long)mk) - 1;
          mk &= mk - 1;
          float qd = S.qvel[j];
          pw = pw + qd * ld3v(&S.cdof[j][0]);
          pv = pv + qd * ld3v(&S.cdof[j][4]);
        }
        V3 cw = ld3v(&S.cdof[lane][0]), cv = ld3v(&S.cdof[lane][4]);
        float qd = S.qvel[lane];
        st3v(&S.dm.dyn.cddq[lane][0], qd * cross(pw, cw));
        st3v(&S.dm.dyn.cddq[lane][4], qd * (cross(pw, cv) + cross(pv, cw)));
      } else {
        st3v(&S.dm.dyn.cddq[lane][0], v3(0, 0, 0));
        st3v(&S.dm.dyn.cddq[lane][4], v3(0, 0, 0));
      }
      if (lane < NB) {
        V3 w = v3(0, 0, 0), v = v3(0, 0, 0);
        f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0;
        if (isbody) {  // lane = body: cvel, composite inertia over the subtree
          uint64_t mk = b_dofmask;
          while (mk) {
            int j = __ffsll((unsigned long long)mk) - 1;
            mk &= mk - 1;
            float qd = S.qvel[j];
            w = w + qd * ld3v(&S.cdof[j][0]);
            v = v + qd * ld3v(&S.cdof[j][4]);
          }
          uint32_t sm = b_submask;
          while (sm) {
            int c = __ffs(sm) - 1;
            sm &= sm - 1;
            const float* p = S.dm.dyn.cinert[c];
            c0 += ldv(p); c1 += ldv(p + 4); c2 += ldv(p + 8);
          }
        }
        st3v(&S.dm.dyn.cvel[lane][0], w);
        st3v(&S.dm.dyn.cvel[lane][4], v);
        float* p = S.dm.dyn.crb[lane];
        stv(p, c0); stv(p + 4, c1); stv(p + 8, c2);
      }
    }
    WSYNC();

    STAMP(3);
    // ======================= body forces (RNE, qacc=0) and mass matrix rows =======================
    {
      if (lane < NB) {
        V3 t = v3(0, 0, 0), f = v3(0, 0, 0);
        if (isbody) {
          V3 aw = v3(0, 0, 0), av = v3(-m->gx, -m->gy, -m->gz);
          uint64_t mk = b_dofmask;
          while (mk) {
            int j = __ffsll((unsigned long long)mk) - 1;
            mk &= mk - 1;
            aw = aw + ld3v(&S.dm.dyn.cddq[j][0]);
            av = av + ld3v(&S.dm.dyn.cddq[j][4]);
          }
          Inert I = ldI(S.dm.dyn.cinert[lane]);
          V3 w = ld3v(&S.dm.dyn.cvel[lane][0]), v = ld3v(&S.dm.dyn.cvel[lane][4]);
          V3 ta, fa, tv, fv;
          imul(I, aw, av, ta, fa);
          imul(I, w, v, tv, fv);
          t = ta + cross(w, tv) + cross(v, fv);
          f = fa + cross(w, fv);
        }
        st3v(&S.dm.dyn.cfrc[lane][0], t);
        st3v(&S.dm.dyn.cfrc[lane][4], f);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) stv(&S.dm.M[lane][4 * q], f4{0, 0, 0, 0});
    }
    WSYNC();
    if (isdof) {  // M[i][j] = cdof_j . (crb_body(i) cdof_i), j over ancestors-or-self (same tree => same block)
      Inert I = ldI(S.dm.dyn.crb[d_body]);
      V3 bt, bf;
      imul(I, ld3v(&S.cdof[lane][0]), ld3v(&S.cdof[lane][4]), bt, bf);
      uint64_t mk = d_ancmask;
      while (mk) {
        int j = __ffsll((unsigned long long)mk) - 1;
        mk &= mk - 1;
        float val = dot(ld3v(&S.cdof[j][0]), bt) + dot(ld3v(&S.cdof[j][4]), bf);
        if (j == lane) val += d_mdiag;
        S.dm.M[lane][j & 15] = val;
        S.dm.M[j][l16] = val;
      }
    }
    float qfrc_bias = 0.0f, qfs = 0.0f;
    if (isdof) {
      V3 t = v3(0, 0, 0), f = v3(0, 0, 0);
      uint32_t sm = d_submask;
      while (sm) {
        int c = __ffs(sm) - 1;
        sm &= sm - 1;
        t = t + ld3v(&S.dm.dyn.cfrc[c][0]);
        f = f + ld3v(&S.dm.dyn.cfrc[c][4]);
      }
      qfrc_bias = dot(ld3v(&S.cdof[lane][0]), t) + dot(ld3v(&S.cdof[lane][4]), f);
      float qd = S.qvel[lane];
      float fa = 0.0f;
      if (d_ctrl == MIR_CTRL_POSITION) {
        fa = d_kp * (S.target[lane] - S.qpos[d_qadr]) - d_kv * qd;
        fa = fminf(fmaxf(fa, d_frclo), d_frchi);
      }
      qfs = -d_damping * qd + fa - qfrc_bias;
    }
    WSYNC();
    STAMP(4);
    // qacc_smooth = Mt^-1 qfrc_smooth: four block solves side by side (Gauss-Jordan on 16-wide register rows)
    float mrow[G];
    {
      f4 r0 = ldv(&S.dm.M[lane][0]), r1 = ldv(&S.dm.M[lane][4]), r2 = ldv(&S.dm.M[lane][8]), r3 = ldv(&S.dm.M[lane][12]);
      mrow[0] = r0.x; mrow[1] = r0.y; mrow[2] = r0.z; mrow[3] = r0.w; mrow[4] = r1.x; mrow[5] = r1.y; mrow[6] = r1.z; mrow[7] = r1.w;
      mrow[8] = r2.x; mrow[9] = r2.y; mrow[10] = r2.z; mrow[11] = r2.w; mrow[12] = r3.x; mrow[13] = r3.y; mrow[14] = r3.z; mrow[15] = r3.w;
    }
    if (a.out_M && isdof && step == 0) {  // parity output in compact dof order (M is block-diagonal)
      const int di = m->d_dof[lane];
      for (int j = 0; j < nv; j++) a.out_M[((size_t)env * nv + di) * nv + j] = 0.0f;
#pragma unroll
      for (int j = 0; j < G; j++) {
        const int dj = m->d_dof[16 * blk + j];
        if (dj >= 0) a.out_M[((size_t)env * nv + di) * nv + dj] = mrow[j] - (j == l16 ? d_mdiag - m->d_armature[lane] : 0.0f);
      }
    }
    if (a.out_bias && isdof && step == 0) a.out_bias[(size_t)env * nv + m->d_dof[lane]] = qfrc_bias;
    float qas;
    {
      float arow[G];
#pragma unroll
      for (int j = 0; j < G; j++) arow[j] = isdof ? mrow[j] : (j == l16 ? 1.0f : 0.0f);
      qas = isdof ? qfs : 0.0f;
      GJ<0>::run(arow, qas, l16);
    }
    S.qas[lane] = qas;
    S.qacc[lane] = qas;
    if (a.out_qas && isdof && step == 0) a.out_qas[(size_t)env * nv + m->d_dof[lane]] = qas;
    WSYNC();  // dyn scratch is dead from here on

    STAMP(5);
    // ======================= collision detection (wave 1's work in the single-step instantiation) ==
    if (!DUAL) collide();
    else __syncthreads();  // (2) contacts finished by wave 1; this wave is done with the dynamics scratch
    const int ncon = __builtin_amdgcn_readfirstlane(S.ncon), ncand = __builtin_amdgcn_readfirstlane(S.ncand);
    const int nmine = S.con.bcount[blk];  // contacts touching this lane's block

    STAMP(9);
    // ======================= constraint rows ======================================================
    // contact base Jacobians (with two waves: wave 1 takes every other pair of list entries)
    jacobians(nmine, 0, DUAL ? 4 : 2);
    // joint-limit rows: lane = dof, lane-private
    float lsg = 0.0f, lD = 0.0f, laref = 0.0f;
    if (d_limited) {
      float q = S.qpos[d_qadr];
      float dlo = q - d_lo, dhi = d_hi - q;
      float pos = 0.0f;
      if (dlo < 0.0f) { pos = dlo; lsg = 1.0f; }
      else if (dhi < 0.0f) { pos = dhi; lsg = -1.0f; }
      if (lsg != 0.0f) {
        float imp = impedance(d_si0, d_si1, d_si2, d_si3, d_si4, pos);
        float Rr = fmaxf((1.0f - imp) / imp * d_iw0, 1e-15f);
        lD = 1.0f / Rr;
        laref = -d_lb * (lsg * S.qvel[lane]) - d_lk * imp * pos;
      }
    }
    WSYNC();
    if (DUAL) __syncthreads();  // (3) both halves of the Jacobian segments
    // contact rows, lane = contact, lane-private: aref_r = -b (J_r qvel) - k imp dist
    const bool iscon = lane < ncon;
    float cmu = 0.0f, cD = 0.0f;
    float aref[4] = {0, 0, 0, 0}, jar[4] = {0, 0, 0, 0};
    if (iscon) {
      float vn, v1, v2;
      condot3(S, lane, S.qvel, vn, v1, v2);
      f4 mt = ldv(S.con.cmeta[lane]);
      cmu = mt.x; cD = mt.y;
      const float base = mt.z, bb = mt.w;
      aref[0] = base - bb * (vn + cmu * v1);
      aref[1] = base - bb * (vn - cmu * v1);
      aref[2] = base - bb * (vn + cmu * v2);
      aref[3] = base - bb * (vn - cmu * v2);
    }

    STAMP(10);
    // ======================= primal Newton solve ====================================================
    const unsigned long long limmask = __ballot(lsg != 0.0f);
    const int nefc = 4 * ncon + __popcll(limmask);
    bool done = nefc == 0;
    float qacc = qas, Ma = 0.0f, ljar = 0.0f;
    const float* xblk_srch = &S.srch[16 * blk];
    const float* xblk_qacc = &S.qacc[16 * blk];
    {
      // warm start: cost(ws) vs cost(qacc_smooth); Gauss part 1/2 dq^T Mt dq
      const float ws = S.qacc_ws[lane];
      const float dq = isdof ? ws - qas : 0.0f;
      S.srch[lane] = dq;
      WSYNC();
      float c_ws = isdof ? 0.5f * rowdot(mrow, xblk_srch) * dq : 0.0f, c_sm = 0.0f;
      const float ljs = lsg * qas - laref, ljw = lsg * ws - laref;
      if (lsg != 0.0f) {
        if (ljs < 0.0f) c_sm += 0.5f * lD * ljs * ljs;
        if (ljw < 0.0f) c_ws += 0.5f * lD * ljw * ljw;
      }
      float js[4] = {0, 0, 0, 0}, jw[4] = {0, 0, 0, 0};
      if (iscon) {
        float sn, s1, s2, wn, w1, w2;
        condot3(S, lane, S.qas, sn, s1, s2);
        condot3(S, lane, S.qacc_ws, wn, w1, w2);
        js[0] = sn + cmu * s1 - aref[0]; js[1] = sn - cmu * s1 - aref[1]; js[2] = sn + cmu * s2 - aref[2]; js[3] = sn - cmu * s2 - aref[3];
        jw[0] = wn + cmu * w1 - aref[0]; jw[1] = wn - cmu * w1 - aref[1]; jw[2] = wn + cmu * w2 - aref[2]; jw[3] = wn - cmu * w2 - aref[3];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          if (js[r] < 0.0f) c_sm += 0.5f * cD * js[r] * js[r];
          if (jw[r] < 0.0f) c_ws += 0.5f * cD * jw[r] * jw[r];
        }
      }
      c_ws = wsum(c_ws);
      c_sm = wsum(c_sm);
      const bool usews = c_ws < c_sm;
      qacc = usews ? ws : qas;
      ljar = usews ? ljw : ljs;
#pragma unroll
      for (int r = 0; r < 4; r++) jar[r] = usews ? jw[r] : js[r];
      WSYNC();
      S.qacc[lane] = qacc;
      WSYNC();
      Ma = isdof ? rowdot(mrow, xblk_qacc) : 0.0f;
    }
    STAMP(11);
    int niter = 0;
    const float tol = m->tolerance, scale = m->solver_scale;
    const float gfloor = 16.0f * 5.96e-8f * sqrtf(wsum(Ma * Ma + qfs * qfs));
    // which blocks can ever be coupled this step: contacts spanning two blocks, closed transitively (wave-uniform)
    unsigned comp = 0x8421u;  // every block with itself
    {
      unsigned mine = 0u;
      if (iscon) {
        const int s0 = S.con.cblk[lane][0], s1 = S.con.cblk[lane][1];
        if (s0 >= 0 && s1 >= 0) mine = (1u << (4 * s0 + s1)) | (1u << (4 * s1 + s0));
      }
#pragma unroll
      for (int bit = 0; bit < 16; bit++)
        if (__ballot((mine >> bit) & 1u)) comp |= 1u << bit;
#pragma unroll
      for (int round = 0; round < 2; round++)
#pragma unroll
        for (int p = 0; p < 4; p++)
#pragma unroll
          for (int q = 0; q < 4; q++)
            if ((comp >> (4 * p + q)) & 1u) comp |= ((comp >> (4 * q)) & 15u) << (4 * p);
    }
    // Newton Hessian H = Mt + J^T D_active J, kept in REGISTERS across iterations and updated incrementally (only rows
    // whose active flag flipped contribute, as in the 16-lane kernel): lane = dof row; hd = the 16 columns of the row's own
    // block (M is block-diagonal), ho[b] = the columns of block b, non-zero only when a contact couples two blocks this
    // step (comp != identity: the arm touches a cube, or two cubes of different blocks touch).  The wave runs alone on its
    // SIMD (LDS bounds the occupancy), so the 80 registers are free and every update is FMA work without memory round trips.
    const bool coupled = comp != 0x8421u;
    STAMP(20);
    // In the single-step instantiation the solve is compiled twice: the block-diagonal case (no contact couples two blocks:
    // 73 % of the envs) carries no off-diagonal rows and no 64-wide working copy, i.e. ~130 registers less than the coupled
    // case.  (The loop instantiations keep one run-time-switched copy: there the duplication cost more than it saved.)
    bool met4 = false;
    auto newton = [&](auto mode_t) {
    constexpr int MODE = decltype(mode_t)::value;  // 0: block-diagonal, 1: coupled, 2: decided at run time (loop instantiations)
    const bool cpl = MODE == 2 ? coupled : MODE == 1;
    float hd[G], ho[MODE == 0 ? 1 : 4][G];
#pragma unroll
    for (int j = 0; j < G; j++) {
      hd[j] = isdof ? mrow[j] : (j == l16 ? 1.0f : 0.0f);
#pragma unroll
      for (int bq = 0; bq < (MODE == 0 ? 1 : 4); bq++) ho[bq][j] = 0.0f;
    }
    const unsigned long long twoblk = __ballot(iscon && S.con.cblk[lane < MAXC ? lane : 0][1] >= 0);  // contacts with two segments
    float oldlact = 0.0f;
    unsigned prevbits = 0u;
    float gprev = 0.0f;
    for (int it = 0; it < m->iterations; it++) {
      if (done) break;  // wave-uniform: one env per wave
      float lact = (lsg != 0.0f && ljar < 0.0f) ? lD : 0.0f;
      const float lf = -lact * ljar;
      bool flip_d = false, flip_o = false;
      if (iscon) {
        float f[4];
        unsigned bits = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const bool on = jar[r] < 0.0f;
          f[r] = on ? -cD * jar[r] : 0.0f;
          bits |= on ? (1u << r) : 0u;
        }
        stv(S.con.cfb[lane], f4{f[0] + f[1] + f[2] + f[3], cmu * (f[0] - f[1]), cmu * (f[2] - f[3]), (float)(bits | (prevbits << 4))});
        flip_d = bits != (it == 0 ? 15u : prevbits);  // (the diagonal blocks start from all rows active,
        flip_o = bits != prevbits;                    //  the off-diagonal ones from zero)
        prevbits = bits;
      }
      // contacts whose active rows changed since the Hessian last saw them (bit = contact): the updates walk those only
      const unsigned long long flipd = __ballot(flip_d), flipo = __ballot(flip_o);
      WSYNC();
      if (it == 0) STAMP(21);
      // ---- gradient first (cheap): convergence is decided before any Hessian work
      float g = isdof ? Ma - qfs - lsg * lf : 0.0f;
      for (int k0 = 0; k0 < nmine; k0 += 4) {  // four list entries per trip: the entries, then every read in one batch
        const int4 e4 = *reinterpret_cast<const int4*>(&S.con.blist[blk][k0]);  // (MAXC is a multiple of 4; k0 is one too)
        const int eqs[4] = {e4.x, k0 + 1 < nmine ? e4.y : e4.x, k0 + 2 < nmine ? e4.z : e4.x, k0 + 3 < nmine ? e4.w : e4.x};
        float jn[4], j1[4], j2[4];
        f4 fb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float* jb = &S.Jb[eqs[u] >> 1][eqs[u] & 1][0];
          jn[u] = jb[l16]; j1[u] = jb[16 + l16]; j2[u] = jb[32 + l16];
          fb[u] = ldv(S.con.cfb[eqs[u] >> 1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (k0 + u < nmine) g -= jn[u] * fb[u].x + j1[u] * fb[u].y + j2[u] * fb[u].z;
      }
      if (!isdof) g = 0.0f;
      if (it == 0) STAMP(22);
      const float gn = sqrtf(wsum(g * g));
      if (scale * gn < tol || gn < gfloor) { done = true; break; }
      if (it == 0) STAMP(12);
      // ---- Hessian rows (lane = dof): incremental update of H = Mt + J^T D_active J
      if (__any(lact != oldlact)) {
        const float dl = lact - oldlact;
#pragma unroll
        for (int j = 0; j < G; j++) hd[j] += j == l16 ? dl : 0.0f;
      }
      oldlact = lact;
      if (it == 0) {  // the all-rows-active Hessian: from wave 1 where there is one
        float hp[G];
        if (DUAL) {
          __syncthreads();  // (4)
          STAMP(23);
          met4 = true;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const f4 v = ldv(hxrow + 4 * q);
            hp[4 * q] = v.x; hp[4 * q + 1] = v.y; hp[4 * q + 2] = v.z; hp[4 * q + 3] = v.w;
          }
        } else {
          hess_full(hp, nmine);
        }
#pragma unroll
        for (int j = 0; j < G; j++) hd[j] += hp[j];
      }
      // diagonal blocks: the four DPP rows walk their own contact lists side by side (first iteration: relative to all-active)
      if (flipd) for (int kq = 0; kq < nmine; kq++) {
        const int eq = S.con.blist[blk][kq];
        const int c = eq >> 1, myseg = eq & 1;
        if (!((flipd >> c) & 1ull)) continue;
        const f4 fb = ldv(S.con.cfb[c]);
        const float* seg = &S.Jb[c][myseg][0];
        const float jn = seg[l16], j1 = seg[16 + l16], j2 = seg[32 + l16];
        const f4 mt = ldv(S.con.cmeta[c]);
        f4 xn[4], x1[4], x2[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { xn[q] = ldv(seg + 4 * q); x1[q] = ldv(seg + 16 + 4 * q); x2[q] = ldv(seg + 32 + 4 * q); }
        __builtin_amdgcn_sched_barrier(0);  // (keep the whole batch of reads ahead of the arithmetic: one LDS round trip)
        const unsigned both = (unsigned)fb.w;
        const unsigned bits = both & 15u, old = it == 0 ? 15u : both >> 4;
        const float mu = mt.x, D = mt.y;
        const float a0 = D * (float)((int)(bits & 1u) - (int)(old & 1u)), a1 = D * (float)((int)(bits >> 1 & 1u) - (int)(old >> 1 & 1u));
        const float a2 = D * (float)((int)(bits >> 2 & 1u) - (int)(old >> 2 & 1u)), a3 = D * (float)((int)(bits >> 3 & 1u) - (int)(old >> 3 & 1u));
        const float w0 = a0 + a1 + a2 + a3, w1 = mu * (a0 - a1), w2 = mu * (a2 - a3), w3 = mu * mu * (a0 + a1), w4 = mu * mu * (a2 + a3);
        const float tn = jn * w0 + j1 * w1 + j2 * w2, t1 = jn * w1 + j1 * w3, t2 = jn * w2 + j2 * w4;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          hd[4 * q + 0] += tn * xn[q].x + t1 * x1[q].x + t2 * x2[q].x;
          hd[4 * q + 1] += tn * xn[q].y + t1 * x1[q].y + t2 * x2[q].y;
          hd[4 * q + 2] += tn * xn[q].z + t1 * x1[q].z + t2 * x2[q].z;
          hd[4 * q + 3] += tn * xn[q].w + t1 * x1[q].w + t2 * x2[q].w;
        }
      }
      // off-diagonal blocks: wave-uniform walk over the two-block contacts; the rows of either block take the other
      // block's segment as columns
      if constexpr (MODE != 0) if (cpl) {
        for (unsigned long long tw = twoblk & flipo; tw; tw &= tw - 1ull) {
          const int c = __builtin_amdgcn_readfirstlane(__builtin_ctzll(tw));
          const f4 fb = ldv(S.con.cfb[c]);
          const unsigned both = (unsigned)fb.w;
          const unsigned bits = both & 15u, old = both >> 4;
          if (bits == old) continue;
          const int sg0 = S.con.cblk[c][0], sg1 = S.con.cblk[c][1];
          const bool in0 = blk == sg0, in1 = blk == sg1;
          if (in0 || in1) {
            const float* seg = &S.Jb[c][in1 ? 1 : 0][0];
            const float* oseg = &S.Jb[c][in1 ? 0 : 1][0];
            const int ob = in1 ? sg0 : sg1;
            const float jn = seg[l16], j1 = seg[16 + l16], j2 = seg[32 + l16];
            const f4 mt = ldv(S.con.cmeta[c]);
            f4 xn[4], x1[4], x2[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { xn[q] = ldv(oseg + 4 * q); x1[q] = ldv(oseg + 16 + 4 * q); x2[q] = ldv(oseg + 32 + 4 * q); }
            const float mu = mt.x, D = mt.y;
            const float a0 = D * (float)((int)(bits & 1u) - (int)(old & 1u)), a1 = D * (float)((int)(bits >> 1 & 1u) - (int)(old >> 1 & 1u));
            const float a2 = D * (float)((int)(bits >> 2 & 1u) - (int)(old >> 2 & 1u)), a3 = D * (float)((int)(bits >> 3 & 1u) - (int)(old >> 3 & 1u));
            const float w0 = a0 + a1 + a2 + a3, w1 = mu * (a0 - a1), w2 = mu * (a2 - a3), w3 = mu * mu * (a0 + a1), w4 = mu * mu * (a2 + a3);
            const float tn = jn * w0 + j1 * w1 + j2 * w2, t1 = jn * w1 + j1 * w3, t2 = jn * w2 + j2 * w4;
            float d[G];
#pragma unroll
            for (int q = 0; q < 4; q++) {
              d[4 * q + 0] = tn * xn[q].x + t1 * x1[q].x + t2 * x2[q].x;
              d[4 * q + 1] = tn * xn[q].y + t1 * x1[q].y + t2 * x2[q].y;
              d[4 * q + 2] = tn * xn[q].z + t1 * x1[q].z + t2 * x2[q].z;
              d[4 * q + 3] = tn * xn[q].w + t1 * x1[q].w + t2 * x2[q].w;
            }
#pragma unroll
            for (int bq = 0; bq < 4; bq++)
              if (ob == bq) {
#pragma unroll
                for (int j = 0; j < G; j++) ho[bq][j] += d[j];
              }
          }
        }
      }
      if (it == 0) STAMP(13);
      // ---- Newton direction: H s = -g: four 16-wide DPP block solves side by side, or dense over the wave
      float sv = -g;
      bool dense = false;
      if constexpr (MODE != 0) dense = cpl;
      if (!dense) {
        float hb[G];
#pragma unroll
        for (int j = 0; j < G; j++) hb[j] = hd[j];
        GJ<0>::run(hb, sv, l16);
      }
      if constexpr (MODE != 0) {
        if (dense) {
          v16f hrow[4];
#pragma unroll
          for (int bq = 0; bq < 4; bq++)
#pragma unroll
            for (int j = 0; j < G; j++) hrow[bq][j] = blk == bq ? hd[j] : ho[bq][j];
          gj_wave(hrow, sv, lane, lanemask, comp);
        }
      }
      if (!isdof) sv = 0.0f;
      if (it == 0) STAMP(14);
      S.srch[lane] = sv;
      WSYNC();
      const float mv = isdof ? rowdot(mrow, xblk_srch) : 0.0f;
      const float ljv = lsg * sv;
      float jv[4] = {0, 0, 0, 0};
      if (iscon) {
        float xn, x1, x2;
        condot3(S, lane, S.srch, xn, x1, x2);
        jv[0] = xn + cmu * x1; jv[1] = xn - cmu * x1; jv[2] = xn + cmu * x2; jv[3] = xn - cmu * x2;
      }
      // ---- exact line search on the piecewise-quadratic phi(alpha): safeguarded Newton on phi'
      // (the search starts at alpha = 1 with phi'(0) = g . s, see the 16-lane kernel; ls counts evaluations, the one at 0 included)
      const float A = wsum(sv * mv), Bq = wsum(sv * (Ma - qfs)), g0 = wsum(sv * g);
      bool lsdone = g0 >= 0.0f;
      float alpha = lsdone ? 0.0f : 1.0f, lo = 0.0f, hi = -1.0f;
      for (int ls = 1; ls < m->ls_iterations && !lsdone; ls++) {
        float pg = 0.0f, ph = 0.0f, pa = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float x = jar[r] + alpha * jv[r];
          if (x < 0.0f) { pg += cD * jv[r] * x; ph += cD * jv[r] * jv[r]; }
        }
        {
          const float x = ljar + alpha * ljv;
          if (x < 0.0f) { pg += lD * ljv * x; ph += lD * ljv * ljv; }
        }
        const float gg = wsum(pg) + alpha * A + Bq, hh = wsum(ph) + A;
        // (from the fifth evaluation on: termination at the resolution of the evaluation itself, see the 16-lane kernel)
        float floorg = 0.0f;
        if (ls >= 4) {
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (jar[r] + alpha * jv[r] < 0.0f) pa += cD * fabsf(jv[r]) * (fabsf(jar[r]) + fabsf(alpha * jv[r]));
          if (ljar + alpha * ljv < 0.0f) pa += lD * fabsf(ljv) * (fabsf(ljar) + fabsf(alpha * ljv));
          floorg = 4.0f * 1.1920929e-7f * (wsum(pa) + fabsf(alpha * A) + fabsf(Bq));
        }
        if (fabsf(gg) <= fmaxf(1e-6f * fabsf(g0), floorg)) lsdone = true;
        if (!lsdone) {
          if (gg < 0.0f) lo = alpha; else hi = alpha;
          float an = alpha - gg / hh;
          if (hi >= 0.0f && (an <= lo || an >= hi)) an = 0.5f * (lo + hi);
          if (an == alpha) lsdone = true;
          if (!lsdone) alpha = an;
        }
      }
      if (it == 0) STAMP(15);
      // ---- improvement from the 1-D model, then the update (row-cost differences as 1/2 D d (2 x0 + d))
      float pim = 0.0f;
#pragma unroll
      // (with a = min(x, 0) the cost of a row is 1/2 D a^2, and its change 1/2 D (a1 - a0)(a1 + a0); a1 - a0 is the step d
      //  itself while the row stays active)
      for (int r = 0; r < 4; r++) {
        const float x0 = jar[r], d = alpha * jv[r], x1 = x0 + d;
        const float a0 = fminf(x0, 0.0f), a1 = fminf(x1, 0.0f);
        pim -= 0.5f * cD * ((x0 < 0.0f && x1 < 0.0f) ? d : a1 - a0) * (a1 + a0);
      }
      {
        const float x0 = ljar, d = alpha * ljv, x1 = x0 + d;
        const float a0 = fminf(x0, 0.0f), a1 = fminf(x1, 0.0f);
        pim -= 0.5f * lD * ((x0 < 0.0f && x1 < 0.0f) ? d : a1 - a0) * (a1 + a0);
      }
      // (rows whose sign the step changes, from the same x0 / x1: the three reductions of this block are independent and overlap)
      float crossed = ((ljar < 0.0f) != (ljar + alpha * ljv < 0.0f)) && lsg != 0.0f ? 1.0f : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; r++) crossed += ((jar[r] < 0.0f) != (jar[r] + alpha * jv[r] < 0.0f)) ? 1.0f : 0.0f;
      const float improvement = wsum(pim) - (0.5f * alpha * alpha * A + alpha * Bq);
      const float moved = wsum((isdof && qacc + alpha * sv != qacc) ? 1.0f : 0.0f);
      const float ncross = wsum(crossed);
      const bool stagnant = it > 0 && gn > 0.5f * gprev && gn < 4.0f * gfloor;
      gprev = gn;
      niter = it + 1;
      if (moved == 0.0f || stagnant) { done = true; }
      if (!done) {
        qacc += alpha * sv;
        Ma += alpha * mv;
        ljar += alpha * ljv;
#pragma unroll
        for (int r = 0; r < 4; r++) jar[r] += alpha * jv[r];
        if (scale * improvement < tol) done = true;
      }
      if (!done) {
        // if the step crossed no row boundary the new gradient is exactly (1 - alpha) g
        const float gnew = fabsf(1.0f - alpha) * gn;
        if (ncross == 0.0f && (scale * gnew < tol || gnew < gfloor)) done = true;
      }
      WSYNC();
    }
    };
    if constexpr (!SINGLE) newton(std::integral_constant<int, 2>{});
    else if (coupled) newton(std::integral_constant<int, 1>{});
    else newton(std::integral_constant<int, 0>{});
    if (DUAL && !met4) __syncthreads();  // (4) (no Hessian was needed: wave 1 is let go)
    STAMP(16);
    if (a.out_qacc && isdof && step == 0) a.out_qacc[(size_t)env * nv + m->d_dof[lane]] = qacc;
    if (a.diag && lane == 0) {
      a.diag[(size_t)env * 4 + 0] = ncon;
      a.diag[(size_t)env * 4 + 1] = nefc;
      a.diag[(size_t)env * 4 + 2] = niter;
      a.diag[(size_t)env * 4 + 3] = ncand;
    }
    if (a.mode != 0) break;

    // ======================= integrate ==============================================================
    WSYNC();
    if (isdof) {
      S.qvel[lane] += dt * qacc;
      S.qacc_ws[lane] = qacc;
    }
    WSYNC();
    if (isdof) {
      const float qd = S.qvel[lane];
      if (d_kind < 2) S.qpos[d_qadr] += dt * qd;
      else if (d_kind == 2) S.qpos[d_qbase + d_axis_k] += dt * qd;
      else if (d_axis_k == 0) {
        V3 w = v3(S.qvel[d_lbase + 3], S.qvel[d_lbase + 4], S.qvel[d_lbase + 5]);
        float wn = sqrtf(dot(w, w));
        float ang = wn * dt;
        if (ang > 1e-15f) {
          float sn, cs;
          sincos_pi2(0.5f * ang, &sn, &cs);
          V3 ax = (1.0f / wn) * w;
          Q4 dq = {cs, ax.x * sn, ax.y * sn, ax.z * sn};
          st4(&S.qpos[d_qbase + 3], qnormalize(qmul(dq, ld4(&S.qpos[d_qbase + 3]))));
        }
      }
    }
    WSYNC();
    STAMP(17);
    if (a.term_host && term_early && lane == 0) {
      // the host-visible terminated byte leaves here, before the closing FK, the observation and the state stores: its trip over
      // PCIe runs under them (see the 16-lane kernel)
      const V3 po = ld3(&S.qpos[obj_qadr]);
      const float r0 = reward_of(po, obj2_qadr >= 0 ? ld3(&S.qpos[obj2_qadr]) : po);
      __hip_atomic_store(&a.term_host[env], (uint8_t)((r0 == 1.0f ? 1u : 0u) | a.term_tag << 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // kinematics of the new state: observations of this step, and the next step's starting poses
    wave_fk(S, lane, nb, bk);
    if (a.rows && a.rows_step && (step + 1 < nsteps || a.ar.episode_len) && lane < ad + ed + 2)  // rollout mode: one packed row per env per step
      a.rows[(size_t)step * a.rows_step + (size_t)env * a.row_stride + lane] = column(lane);
    if (a.ar.episode_len) {
      // episode bookkeeping and re-spawn on chip (the rules of k_autoreset): the row above is the terminal observation
      const bool term = reward_now() == 1.0f;
      const int len = eplen + 1;
      const bool trunc = !term && a.ar.max_len > 0 && len >= a.ar.max_len;
      const bool done = term || trunc;  // wave-uniform: one env per wave
      if (lane == 0 && a.rows && a.row_stride > ad + ed + 2)
        a.rows[(size_t)step * a.rows_step + (size_t)env * a.row_stride + ad + ed + 2] = trunc ? 1.0f : 0.0f;
      eplen = done ? 0 : len;
      WSYNC();
      if (done) {
        S.qvel[lane] = 0.0f;
        S.qacc_ws[lane] = 0.0f;
        if (isdof) {
          const int ai = m->d_armidx[lane];
          if (ai >= 0) {
            const float v = a.ar.arm_qpos[(size_t)env * m->n_arm_q + ai];
            S.qpos[d_qadr] = v;
            S.target[lane] = v;
          }
        }
        const int nfree = m->nfree;
        const float* sp = a.ar.spawn_pool + ((size_t)(epcur % a.ar.pool_len) * a.B + env) * nfree * 3;
        if (lane < 7 * nfree) {
          const int k = lane / 7, j = lane - 7 * k;
          S.qpos[m->free_qadr[k] + j] = j < 3 ? sp[k * 3 + j] : a.ar.obj_quat[((size_t)env * nfree + k) * 4 + (j - 3)];
        }
        epcur += 1;
        WSYNC();
        wave_fk(S, lane, nb, bk);
      }
    }
  }  // steps
  STAMP(18);
  // (the host-visible terminated byte goes out first: its trip over PCIe runs under the stores below)
  const float rew = reward_now();
  if (lane == 0 && a.term_host && !term_early)
    __hip_atomic_store(&a.term_host[env], (uint8_t)((rew == 1.0f ? 1u : 0u) | a.term_tag << 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (lane < nb) {
    float* p = a.poses + ((size_t)env * 2 * NB + lane) * 4;
    *reinterpret_cast<f4*>(p) = ldv(S.xpos[lane]);
    *reinterpret_cast<f4*>(p + 4 * NB) = ldv(S.xquat[lane]);
  }
  if (lane == 0) a.fkvalid[env] = 1;
  // ---- store state ---------------------------------------------------------------------------------
  if (a.mode == 0) {
    a.qpos[(size_t)env * K64_QSTRIDE + lane] = S.qpos[lane];
    a.qvel[(size_t)env * NL + lane] = S.qvel[lane];
    a.qacc_ws[(size_t)env * NL + lane] = S.qacc_ws[lane];
  }
  if (a.action) a.target[(size_t)env * NL + lane] = S.target[lane];
  (void)nq;
  // ---- observations (get_obs / compute_reward / terminated) ---------------------------------------
  if (a.agent_pos && lane < ad) a.agent_pos[(size_t)env * ad + lane] = column(lane);
  if (a.env_state && lane < ed) a.env_state[(size_t)env * ed + lane] = column(ad + lane);
  if (lane == 0) {
    if (a.reward) a.reward[env] = rew;
    if (a.terminated) a.terminated[env] = rew == 1.0f ? 1 : 0;
  }
  if (a.ar.episode_len && lane == 0) { a.ar.episode_len[env] = eplen; a.ar.cursor[env] = epcur; }
  if (a.rows && !(a.ar.episode_len && a.rows_step) && lane < ad + ed + 2)  // (in rollout mode: the last step's row; with autoreset: written in the loop)
    a.rows[(size_t)(a.rows_step ? (nsteps > 0 ? nsteps - 1 : 0) : 0) * a.rows_step + (size_t)env * a.row_stride + lane] = column(lane);
  if (a.out_xpos && lane < nb) {
    st3(&a.out_xpos[((size_t)env * nb + lane) * 3], ld3v(S.xpos[lane]));
    st4(&a.out_xquat[((size_t)env * nb + lane) * 4], ld4v(S.xquat[lane]));
  }
  STAMP(19);
}

}  // namespace

extern "C" int mir_launch_step64(const StepArgs64* args, hipStream_t stream) {
  StepArgs64 a = *args;
#ifdef MIR_PROFILE_SINGLE
  const bool prof_blocks_single = false;
#else
  const bool prof_blocks_single = a.prof != nullptr;
#endif
  const bool single = a.mode == 0 && a.n_steps == 1 && !a.act_step && !a.rows_step && !a.ar.episode_len && !prof_blocks_single && !a.out_M && !a.out_bias &&
                      !a.out_qas && !a.out_qacc && !a.out_xpos && !a.out_xquat;
  const bool plain_loop = a.mode == 0 && !a.prof && !a.out_M && !a.out_bias && !a.out_qas && !a.out_qacc && !a.out_xpos && !a.out_xquat && !a.agent_pos &&
                          !a.env_state && !a.reward && !a.terminated && !a.term_host;
  if (single) hipLaunchKernelGGL(mir_step64_kernel<0>, dim3(a.B), dim3(128), 0, stream, a);  // two waves per env
  else if (plain_loop) hipLaunchKernelGGL(mir_step64_kernel<1>, dim3(a.B), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL(mir_step64_kernel<2>, dim3(a.B), dim3(64), 0, stream, a);
  return (int)hipGetLastError();
}

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
#include "mir_convex.h"
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

// ---- the Newton direction when contacts couple blocks: block elimination on the DPP rows --------------------------------------
// Lane (block r, row i) holds row i of H as four column blocks h[0..3] (h[r] = its own diagonal block; h[q] is zero unless a
// contact couples r and q this step).  `comp` (wave-uniform) has bit 4 p + q set when blocks p and q lie in one connected
// component of the coupling graph.  Blocks are eliminated in index order; later(p) = the blocks q > p of p's component:
//   * forward, p = 0, 1, 2 with later(p) not empty: the rows of p run the 16-wide register Gauss-Jordan of the uncoupled case on
//     [H_pp | H_pq, q in later(p) | g_p] (one DPP row_newbcast fma per column and pivot, the other rows idle), which leaves
//     X_pq = H_pp^-1 H_pq in h[q] and y_p in b; X_pq goes through a 1 KB LDS tile (one column block at a time) to the rows of
//     every r in later(p), which take the Schur update H_rq -= H_rp X_pq, g_r -= H_rp y_p with their own H_rp = h[p] as
//     multipliers (broadcast reads: every lane of the wave reads the same 16 bytes);
//   * then every block that was not eliminated solves its (updated) diagonal block: the four-rows-side-by-side solve of the
//     uncoupled case;
//   * backward, p = 2, 1, 0: s_p = y_p - sum_q X_pq s_q with X_pq still in the rows' registers and s_q read from LDS.
// Work is proportional to what is coupled: one coupled pair of blocks costs one augmented 15-pivot elimination (31 columns), one
// 15 x 15 x 15 update and the common final solve -- ~1 k instructions; the rolled 64-column Gauss-Jordan this replaces (pivot row by
// v_readlane, pivot column by VGPR indexing) took 25-35 k cycles per Newton iteration (per-env timing, tools/probes/env_time_hist64.py).
// (four separately named arrays and compile-time block indices: with a float[4][16] walked by unrolled loops the matrix stayed in
//  private memory -- the unrolling comes too late for the scalar-replacement pass)
struct H4 {
  float (&c0)[G], (&c1)[G], (&c2)[G], (&c3)[G];  // (four separate 64-byte objects: one 256-byte object was left in private memory too)
  template <int Q>
  __device__ __forceinline__ float (&col())[G] {
    if constexpr (Q == 0) return c0;
    else if constexpr (Q == 1) return c1;
    else if constexpr (Q == 2) return c2;
    else return c3;
  }
};
__device__ __forceinline__ unsigned later_of(unsigned comp, int p) { return (comp >> (4 * p)) & (0xfu << (p + 1)) & 0xfu; }

// The 16-wide solve of the uncoupled case with its column updates as v_fmac_f32_dpp (mir_gj_dpp.h): bit-identical to GJ<0>::run.
template <int K>
__device__ __forceinline__ void gj16_dpp(float (&a)[G], float& b, int l16) {
  const float inv = __builtin_amdgcn_rcpf(row_bcast<K>(a[K]));
  const float nf = l16 == K ? inv - 1.0f : -(a[K] * inv);
  gj_dpp_step<K, K + 1>(a, b, nf);
  if constexpr (K + 1 < G - 1) gj16_dpp<K + 1>(a, b, l16);
}
template <int P, int K>
struct GJA {
  // (all lanes run it -- the DPP broadcasts want convergent code; the rows of the other blocks carry f = 0 and keep their values.
  //  Column updates: one v_fmac_f32_dpp per entry, mir_gj_dpp.h)
  static __device__ __forceinline__ void run(H4& h, float& b, int l16, bool mine, unsigned later) {
    float (&own)[G] = h.col<P>();
    const float pk = row_bcast<K>(own[K]);
    const float inv = __builtin_amdgcn_rcpf(pk);
    const float nf = mine ? (l16 == K ? inv - 1.0f : -(own[K] * inv)) : 0.0f;  // -f of GJ in mir_dev.h
    gj_dpp_step<K, K + 1>(own, b, nf);
    if constexpr (P < 1) { if (later & 2u) gj_dpp_cols<K>(h.col<1>(), nf); }
    if constexpr (P < 2) { if (later & 4u) gj_dpp_cols<K>(h.col<2>(), nf); }
    if constexpr (P < 3) { if (later & 8u) gj_dpp_cols<K>(h.col<3>(), nf); }
    if constexpr (K + 1 < G - 1) GJA<P, K + 1>::run(h, b, l16, mine, later);
  }
};

// dst[j] -= sum_k mul[k] * X[k][j] over the 15 x 15 tile in LDS (rows of 16 floats; broadcast reads, two rows per batch)
__device__ __forceinline__ void schur_update(float (&dst)[G], const float (&mul)[G], const float* xs) {
#pragma unroll
  for (int k0 = 0; k0 < 16; k0 += 2) {
    f4 x[2][4];
#pragma unroll
    for (int u = 0; u < 2; u++)
#pragma unroll
      for (int c = 0; c < 4; c++) x[u][c] = ldv(xs + 16 * (k0 + u) + 4 * c);
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (k0 + u >= G - 1) continue;  // (slot 15 of a block never carries a dof)
      const float mk = -mul[k0 + u];
#pragma unroll
      for (int c = 0; c < 4; c++) {
        dst[4 * c + 0] = fmaf(mk, x[u][c].x, dst[4 * c + 0]);
        dst[4 * c + 1] = fmaf(mk, x[u][c].y, dst[4 * c + 1]);
        dst[4 * c + 2] = fmaf(mk, x[u][c].z, dst[4 * c + 2]);
        dst[4 * c + 3] = fmaf(mk, x[u][c].w, dst[4 * c + 3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // (the batches stay apart: registers)
  }
}
__device__ __forceinline__ float dot16(const float (&a)[G], const float* x) {  // sum_j a[j] x[j], x in LDS
  const f4 x0 = ldv(x), x1 = ldv(x + 4), x2 = ldv(x + 8), x3 = ldv(x + 12);
  return ((a[0] * x0.x + a[1] * x0.y + a[2] * x0.z + a[3] * x0.w) + (a[4] * x1.x + a[5] * x1.y + a[6] * x1.z + a[7] * x1.w)) +
         ((a[8] * x2.x + a[9] * x2.y + a[10] * x2.z + a[11] * x2.w) + (a[12] * x3.x + a[13] * x3.y + a[14] * x3.z + a[15] * x3.w));
}

// X_pq through the LDS tile to the rows of later(p): H_rq -= H_rp X_pq
template <int P, int Q>
__device__ __forceinline__ void schur_block(H4& h, bool mine, bool target, int l16, float* xs) {
  float (&xq)[G] = h.col<Q>();
  if (mine) {
#pragma unroll
    for (int c = 0; c < 4; c++) stv(xs + 16 * l16 + 4 * c, f4{xq[4 * c], xq[4 * c + 1], xq[4 * c + 2], xq[4 * c + 3]});
  }
  WSYNC();
  if (target) schur_update(xq, h.col<P>(), xs);
  WSYNC();
}
// forward step for pivot block P (wave-uniform early-out when nothing later is coupled to it); xs = 1 KB tile, ys = 64 floats
template <int P>
__device__ __forceinline__ void coupled_forward(H4& h, float& b, int blk, int l16, unsigned comp, float* xs, float* ys) {
  const unsigned later = later_of(comp, P);
  if (!later) return;
  const bool mine = blk == P, target = (later >> blk) & 1u;
  GJA<P, 0>::run(h, b, l16, mine, later);
  if (mine) ys[16 * P + l16] = b;
  if constexpr (P < 1) { if (later & 2u) schur_block<P, 1>(h, mine, target, l16, xs); }
  if constexpr (P < 2) { if (later & 4u) schur_block<P, 2>(h, mine, target, l16, xs); }
  if constexpr (P < 3) { if (later & 8u) schur_block<P, 3>(h, mine, target, l16, xs); }
  if (target) b -= dot16(h.col<P>(), ys + 16 * P);
}
template <int P>
__device__ __forceinline__ void coupled_backward(H4& h, float& b, int blk, unsigned comp, float* ys, int lane) {
  const unsigned later = later_of(comp, P);
  if (!later) return;
  if (blk == P) {
    float acc = 0.0f;
    if constexpr (P < 1) { if (later & 2u) acc += dot16(h.col<1>(), ys + 16); }
    if constexpr (P < 2) { if (later & 4u) acc += dot16(h.col<2>(), ys + 32); }
    if constexpr (P < 3) { if (later & 8u) acc += dot16(h.col<3>(), ys + 48); }
    b -= acc;
  }
  WSYNC();
  if (blk == P) ys[lane] = b;
  WSYNC();
}
// H s = b for the whole wave; on return b = s_i.  h is consumed.
__device__ __forceinline__ void coupled_solve(H4& h, float& b, int lane, unsigned comp, float* xs, float* ys) {
  const int blk = lane >> 4, l16 = lane & 15;
  coupled_forward<0>(h, b, blk, l16, comp, xs, ys);
  coupled_forward<1>(h, b, blk, l16, comp, xs, ys);
  coupled_forward<2>(h, b, blk, l16, comp, xs, ys);
  {
    float hb[G], sf = b;
#pragma unroll
    for (int j = 0; j < G; j++) hb[j] = blk == 0 ? h.c0[j] : (blk == 1 ? h.c1[j] : (blk == 2 ? h.c2[j] : h.c3[j]));
    gj16_dpp<0>(hb, sf, l16);
    if (!later_of(comp, blk)) b = sf;  // (the eliminated blocks keep their y)
  }
  WSYNC();
  ys[lane] = b;
  WSYNC();
  coupled_backward<2>(h, b, blk, comp, ys, lane);
  coupled_backward<1>(h, b, blk, comp, ys, lane);
  coupled_backward<0>(h, b, blk, comp, ys, lane);
}

// [own | aug | b] of the rows of ONE block (`mine`); the other rows run along with nf = 0 and keep their values
template <int K>
__device__ __forceinline__ void gj16_aug(float (&own)[G], float (&aug)[G], float& b, int l16, bool mine) {
  const float inv = __builtin_amdgcn_rcpf(row_bcast<K>(own[K]));
  const float nf = mine ? (l16 == K ? inv - 1.0f : -(own[K] * inv)) : 0.0f;
  gj_dpp_step<K, K + 1>(own, b, nf);
  gj_dpp_cols<K>(aug, nf);
  if constexpr (K + 1 < G - 1) gj16_aug<K + 1>(own, aug, b, l16, mine);
}
// ONE coupled pair of blocks p < q (92 % of the coupled envs of the stack tasks: two cubes of different blocks touching, the arm
// on a cube): the block elimination above with everything static but the two row masks -- a lane needs one off-diagonal block
// only, hx = H_pq for the rows of p and H_qp for the rows of q.  ~1.1 k instructions (the general routine runs ~4 k for a
// component of three blocks, and instructions, not flops, are what a wave alone on its SIMD pays for).  hb / hx are consumed.
__device__ __forceinline__ void pair_solve(float (&hb)[G], float (&hx)[G], float& b, int blk, int l16, int lane, int p, int q, float* xs, float* ys) {
  const bool mine = blk == p, target = blk == q;
  gj16_aug<0>(hb, hx, b, l16, mine);  // rows of p: hx = X = H_pp^-1 H_pq, b = y_p
  if (mine) {
#pragma unroll
    for (int c = 0; c < 4; c++) stv(xs + 16 * l16 + 4 * c, f4{hx[4 * c], hx[4 * c + 1], hx[4 * c + 2], hx[4 * c + 3]});
    ys[lane] = b;
  }
  WSYNC();
  if (target) {  // Schur complement of the rows of q
    schur_update(hb, hx, xs);
    b -= dot16(hx, ys + 16 * p);
  }
  float sf = b;
  gj16_dpp<0>(hb, sf, l16);  // every block's own solve, side by side (the rows of p have theirs already)
  if (!mine) b = sf;
  WSYNC();
  if (target) ys[lane] = b;
  WSYNC();
  if (mine) b -= dot16(hx, ys + 16 * q);
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
static_assert(sizeof(Con64) >= 4 * 48 * sizeof(float) + MIR_MAX_VERT * 16, "the box-box clipping exchange and the hull vertex pool live in the contact arrays during the narrowphase");

struct BodyK64 {
  int jtype, qadr;
  V3 pos, axis;
  Q4 quat;
};

// forward kinematics by pointer jumping over the parent links (see group_fk in mir_step.hip): 5 rounds cover any tree
// of up to 32 bodies; the pointer travels in the w slot of the position
// JOINTED_ONLY: the free bodies are left out (their poses are their qpos rows; the caller writes them) -- requires that no jointed
// body hangs off a free body (DevModel64.fk_free_leaf).
template <bool JOINTED_ONLY = false>
__device__ __forceinline__ void wave_fk(Env64& S, int lane, int nb, const BodyK64& k) {
  V3 P = v3(0, 0, 0);
  Q4 Qx = Q4{1, 0, 0, 0};
  int anc = 0;
  const bool mine = lane < nb && !(JOINTED_ONLY && k.jtype == MIR_JNT_FREE);
  if (lane > 0 && mine) {
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
  if (mine) {
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

// J_c against THREE vectors in one pass over the contact's Jacobian segments (each product in condot3's own order of operations, so
// the results are the same bits): the segment rows are read once instead of three times -- 24 of the 32 LDS reads of a segment are
// Jacobian rows.  Used where the velocity, the smooth acceleration and the warm start meet the contact rows back to back.
__device__ __forceinline__ void condot3x3(const Env64& S, int c, const float* x, const float* y, const float* z, float (&dx)[3], float (&dy)[3], float (&dz)[3]) {
#pragma unroll
  for (int k = 0; k < 3; k++) dx[k] = dy[k] = dz[k] = 0.0f;
  const int bs[2] = {S.con.cblk[c][0], S.con.cblk[c][1]};
#pragma unroll
  for (int sgi = 0; sgi < 2; sgi++) {
    const int b = bs[sgi];
    if (b < 0) continue;
    const float* jb = &S.Jb[c][sgi][0];
    float ax[3] = {0, 0, 0}, ay[3] = {0, 0, 0}, az[3] = {0, 0, 0};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f4 ja = ldv(jb + 4 * q), jbb = ldv(jb + 16 + 4 * q), jc = ldv(jb + 32 + 4 * q);
      const f4 xv = ldv(x + 16 * b + 4 * q), yv = ldv(y + 16 * b + 4 * q), zv = ldv(z + 16 * b + 4 * q);
      ax[0] += ja.x * xv.x + ja.y * xv.y + ja.z * xv.z + ja.w * xv.w;
      ax[1] += jbb.x * xv.x + jbb.y * xv.y + jbb.z * xv.z + jbb.w * xv.w;
      ax[2] += jc.x * xv.x + jc.y * xv.y + jc.z * xv.z + jc.w * xv.w;
      ay[0] += ja.x * yv.x + ja.y * yv.y + ja.z * yv.z + ja.w * yv.w;
      ay[1] += jbb.x * yv.x + jbb.y * yv.y + jbb.z * yv.z + jbb.w * yv.w;
      ay[2] += jc.x * yv.x + jc.y * yv.y + jc.z * yv.z + jc.w * yv.w;
      az[0] += ja.x * zv.x + ja.y * zv.y + ja.z * zv.z + ja.w * zv.w;
      az[1] += jbb.x * zv.x + jbb.y * zv.y + jbb.z * zv.z + jbb.w * zv.w;
      az[2] += jc.x * zv.x + jc.y * zv.y + jc.z * zv.z + jc.w * zv.w;
    }
#pragma unroll
    for (int k = 0; k < 3; k++) { dx[k] += ax[k]; dy[k] += ay[k]; dz[k] += az[k]; }
  }
}

// ---------------------------------------------------------------------------------------------
// SINGLE = one full step per launch without the rollout / autoreset / per-stage-output options: no step loop, hence none of
// the scalar-register spills the loop structure forces (see mir_step.hip).
// VARIANT 0 = SINGLE; 1 = the step loop of rollouts (packed rows only, no per-stage / separate outputs); 2 = everything.
// DUAL (the single-step instantiation): the workgroup is TWO waves on one env.  Wave 1 stages the model tables, then -- once
// wave 0 has the body poses -- runs the whole collision phase (geom poses, broadphase, plane-box, box-box, contact finish, per-block
// lists) in its own scratch while wave 0 runs the dynamics up to the smooth solve; they meet before the Jacobian segments are
// written, and wave 1 retires.  The register budget is held at 256 so that the four workgroups of a CU (LDS) are two waves per SIMD.
// CONVEX: the scene has sphere / capsule geoms (the Panda's links 1-7 in the reference's scenes are rounded link meshes: capsules
// here, as in the 16-lane kernel): plane - sphere / capsule in closed form, every other pair that is not box - box through GJK on
// the cores and MPR (mir_convex.h), lane = candidate pair.
template <int VARIANT, bool CONVEX>
__global__ __launch_bounds__(VARIANT == 0 ? 128 : 64) __attribute__((amdgpu_waves_per_eu(VARIANT == 0 ? 2 : 1, VARIANT == 0 ? 2 : 1)))
void mir_step64_kernel(StepArgs64 a) {
  constexpr bool SINGLE = VARIANT == 0;
  constexpr bool DUAL = SINGLE;
  __shared__ __attribute__((aligned(16))) Env64 S;
  const DevModel64* __restrict__ m = a.model;
  const int lane = threadIdx.x & 63;
  const bool helper = DUAL && threadIdx.x >= 64;  // wave-uniform
  const bool fk_split = DUAL && m->fk_free_leaf != 0;  // the closing FK of the jointed bodies runs on wave 1 (wave-uniform)
#ifdef MIR_PROFILE_SINGLE
  const unsigned long long t_entry = __builtin_readcyclecounter();
#endif
  const int blk = lane >> 4, l16 = lane & 15;
  int env = blockIdx.x;  // grid = B exactly
  if (SINGLE && a.cost_in) {
    // expensive-first order inside this workgroup's chunk of 4096 envs (see mir_step64.h): lane i counts the flags of envs
    // 64 i .. 64 i + 63 (one 64-byte read, four bytes per v_sad_u8), a wave scan of the counts finds the lane that holds the
    // wanted env, and the flag bytes of that lane alone are walked on the scalar unit.  Every workgroup of the launch reads the
    // same bytes, so the map is a permutation of the chunk.
    const int base = (int)blockIdx.x & ~4095, r = (int)blockIdx.x - base;
    const int n = min(4096, a.B - base);
    const int first = 64 * lane, have = min(max(n - first, 0), 64);  // envs of this lane that exist
    const uint4* fp = reinterpret_cast<const uint4*>(a.cost_in + base + first);
    uint4 fw[4] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    if (have > 0) {  // (the buffers are padded to a multiple of 64 bytes; flags are 0 / 1)
#pragma unroll
      for (int k = 0; k < 4; k++) fw[k] = fp[k];
    }
    unsigned cnt = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      cnt = __builtin_amdgcn_sad_u8(fw[k].x, 0u, cnt); cnt = __builtin_amdgcn_sad_u8(fw[k].y, 0u, cnt);
      cnt = __builtin_amdgcn_sad_u8(fw[k].z, 0u, cnt); cnt = __builtin_amdgcn_sad_u8(fw[k].w, 0u, cnt);
    }
    // inclusive scans over the wave of (expensive, cheap) counts, packed in one float pair via two DPP row scans + row totals
    float sh = (float)cnt, sl = (float)(have - (int)cnt);
    sh += row_shr<1>(sh); sl += row_shr<1>(sl);
    sh += row_shr<2>(sh); sl += row_shr<2>(sl);
    sh += row_shr<4>(sh); sl += row_shr<4>(sl);
    sh += row_shr<8>(sh); sl += row_shr<8>(sl);
    const float h0 = rl(sh, 15), h1 = rl(sh, 31), h2 = rl(sh, 47), h3 = rl(sh, 63);
    const float l0 = rl(sl, 15), l1 = rl(sl, 31), l2 = rl(sl, 47);
    const int rowq = lane >> 4;
    const int inch = (int)(sh + (rowq == 0 ? 0.0f : (rowq == 1 ? h0 : (rowq == 2 ? h0 + h1 : h0 + h1 + h2))));
    const int incl = (int)(sl + (rowq == 0 ? 0.0f : (rowq == 1 ? l0 : (rowq == 2 ? l0 + l1 : l0 + l1 + l2))));
    const int H = (int)(h0 + h1 + h2 + h3);
    const bool wantexp = r < H;
    const int want = wantexp ? r : r - H;  // rank inside its class
    const unsigned long long reach = __ballot((wantexp ? inch : incl) > want);
    const int L = __builtin_amdgcn_readfirstlane(__builtin_ctzll(reach));  // (r < n, so some lane reaches it)
    const int before = L == 0 ? 0 : __builtin_amdgcn_readlane(wantexp ? inch : incl, L - 1);
    int k = want - before;  // the (k + 1)-th env of the class among lane L's
    int pos = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const unsigned wq[4] = {(unsigned)__builtin_amdgcn_readlane((int)fw[q].x, L), (unsigned)__builtin_amdgcn_readlane((int)fw[q].y, L),
                              (unsigned)__builtin_amdgcn_readlane((int)fw[q].z, L), (unsigned)__builtin_amdgcn_readlane((int)fw[q].w, L)};
#pragma unroll
      for (int d = 0; d < 4; d++)
#pragma unroll
        for (int b8 = 0; b8 < 4; b8++) {
          const bool isexp = ((wq[d] >> (8 * b8)) & 0xffu) != 0u;
          const bool hit = k >= 0 && isexp == wantexp;
          if (hit && k == 0) pos = 16 * q + 4 * d + b8;
          k -= hit ? 1 : 0;
        }
    }
    env = base + 64 * L + pos;
  }
  // list mode (mir_step64.h): the workgroup serves list entry `slot`; state rows in the 16-lane kernel's layout
  const int slot = blockIdx.x;
  if (a.env_list) env = a.env_list[slot];
  const bool lay16 = a.lay16_qst > 0;  // wave-uniform
  // the state rows of the env, from LDS to HBM, in the layout of the handle (called where the step is integrated)
  auto store_state = [&](Env64& Sx, int lane_, int dof16_, bool isdof_) {
    if (lay16) {
      const bool hasd = isdof_ && dof16_ >= 0;
      if (a.mode == 0) {
        if (lane_ < a.lay16_qst) a.qpos[(size_t)env * a.lay16_qst + lane_] = Sx.qpos[lane_];
        if (hasd) { a.qvel[(size_t)env * 16 + dof16_] = Sx.qvel[lane_]; a.qacc_ws[(size_t)env * 16 + dof16_] = Sx.qacc_ws[lane_]; }
      }
      if (a.action && hasd) a.target[(size_t)env * 16 + dof16_] = Sx.target[lane_];
    } else {
      if (a.mode == 0) {
        a.qpos[(size_t)env * K64_QSTRIDE + lane_] = Sx.qpos[lane_];
        a.qvel[(size_t)env * NL + lane_] = Sx.qvel[lane_];
        a.qacc_ws[(size_t)env * NL + lane_] = Sx.qacc_ws[lane_];
      }
      if (a.action) a.target[(size_t)env * NL + lane_] = Sx.target[lane_];
    }
  };

  const int nb = m->nbody, nv = m->nv, nq = m->nq;
  const int ngeom = m->ngeom, npair = m->npair, max_contacts = m->max_contacts, enable_collision = m->enable_collision;
  const int mdl_nvert = CONVEX ? m->nvert : 0;  // hull vertices of the scene (wave-uniform)
  const float dt = m->dt;
  const uint64_t lanemask = m->lanemask;
  // every scalar of the model that the step reads is fetched here, with the first batch of loads: a read through `m` further down
  // cannot move above the barriers and sits where it is used -- a scalar-cache round trip in the middle of the serial chain
  // (m->iterations was re-read in every Newton iteration, m->ls_iterations in every line-search evaluation)
  const int mdl_iterations = m->iterations, mdl_ls_iterations = m->ls_iterations;
  const float mdl_tolerance = m->tolerance, mdl_scale = m->solver_scale;

  // ---- per-lane model constants (lane = body = dof slot) ------------------------------------------
  const bool isbody = lane < nb && lane > 0;
  const bool isdof = (lanemask >> lane) & 1ull;
  const int bl = lane < NB ? lane : 0;  // body index this lane may own
  BodyK64 bk;
  bk.jtype = m->b_jtype[bl]; bk.qadr = m->b_qadr[bl];
  bk.pos = ld3(m->b_pos[bl]); bk.axis = ld3(m->b_axis[bl]); bk.quat = ld4(m->b_quat[bl]);
  const int b_root = m->b_root[bl];
  const V3 b_ipos = ld3(m->b_ipos[bl]);
  const float b_mass = m->b_mass[bl];
  float ib[6];
#pragma unroll
  for (int k = 0; k < 6; k++) ib[k] = m->b_inertia[bl][k];
  const int d_body = isdof ? m->d_body[lane] : 0;  // (lanemask is a scalar; the load itself is unconditional in effect)
  const int d_kind = m->d_kind[lane], d_qadr = m->d_qadr[lane], d_axis_k = m->d_axis_k[lane];
  const int d_root = m->d_root[lane];
  const V3 d_axis = ld3(m->d_axis[lane]);
  const uint64_t d_ancmask = m->d_ancmask[lane];
  // tree-scan links (mir_compile64.cpp): scan parent of the dof, the dof whose inclusive sum is the velocity in front of this dof,
  // the body's last moving dof, the lane behind the body's subtree (-1: none / the subtree ends with its row)
  const int scanw = m->scanw[lane];
  const int d_par = (int)(signed char)(scanw & 255), d_bef = (int)(signed char)(scanw >> 8 & 255);
  const int b_last = (int)(signed char)(scanw >> 16 & 255), b_next = (int)(signed char)(scanw >> 24 & 255);
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
  const int dof16 = m->d_dof[lane];  // (compact dof index of this lane, -1 for none: the row index of the 16-lane layout)
  // lane = geom: frame in its body, staged tables
  const int gl = lane < ngeom ? lane : 0;
  const int g_bodyl = m->g_body[gl];
  const V3 g_posl = ld3(m->g_pos[gl]);
  const Q4 g_quatl = ld4(m->g_quat[gl]);
  // Every global read of the launch -- staged tables, state rows, action, cached poses -- is issued before the first LDS
  // store: one L2 round trip at the start of the launch instead of one per group of stores (see mir_step.hip).
  const int gi = lane < ngeom ? lane : 0, bi = lane < NB ? lane : 0;
  const int gt_in = m->g_type[gi];
  const int gstatic_in = m->b_static[m->g_body[gi]];  // (the geom sits on a body that never moves: candidate for the slab fast path)
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
    if (lay16) {  // (the 16-lane kernel's rows: qpos by address, the rest by dof; lanes without a dof hold zeros, as this kernel's own rows do)
      const bool hasd = isdof && dof16 >= 0;
      q_in = lane < a.lay16_qst ? a.qpos[(size_t)env * a.lay16_qst + lane] : 0.0f;
      qv_in = hasd ? a.qvel[(size_t)env * 16 + dof16] : 0.0f; ws_in = hasd ? a.qacc_ws[(size_t)env * 16 + dof16] : 0.0f;
      tg = hasd ? a.target[(size_t)env * 16 + dof16] : 0.0f;
    } else {
      q_in = a.qpos[(size_t)env * K64_QSTRIDE + lane]; qv_in = a.qvel[(size_t)env * NL + lane]; ws_in = a.qacc_ws[(size_t)env * NL + lane];
      tg = a.target[(size_t)env * NL + lane];
    }
    au = (a.action && lane < a.nu) ? a.action[(size_t)env * a.nu + lane] : 0.0f;
    if (a.poses) {
      // (the cached poses travel with their validity flag; (B, 2, 32, 4): the speculative read is in bounds)
      const float* pose_p = a.poses + ((size_t)env * 2 * NB + (lane & (NB - 1))) * 4;
      cpos = *reinterpret_cast<const f4*>(pose_p); cquat = *reinterpret_cast<const f4*>(pose_p + 4 * NB);
      cached = a.fkvalid[env] != 0;  // wave-uniform
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // (nothing below may move in front of the loads above)
  if (!DUAL || helper) {
    if (lane < ngeom) {
      stv(S.gts[lane], f4{__int_as_float(gt_in | (g_bodyl << 8) | (gstatic_in ? 1 << 16 : 0)), gsx, gsy, gsz});
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
    if (lane == 0) { S.ncon = 0; S.ncand = 0; S.pad0 = 0; }
    if (lane < ngeom) {
      Q4 qb = ld4v(S.xquat[g_bodyl]);
      st3v(S.col.gpos[lane], ld3v(S.xpos[g_bodyl]) + qrot(qb, g_posl));
      st4v(S.col.gquat[lane], qmul(qb, g_quatl));
    }
    S.col.ccount[lane] = 0;
    // hull vertices (MIR_GEOM_HULL, convex instantiation): the scene's pool, <= 96 rows of 16 bytes, is brought into the contact
    // arrays' space for the duration of the narrowphase (they are written after it; the box-box routine's scratch is its first
    // 768 bytes) -- there is no LDS left for a resident copy (40.6 of 40 KB per env), and the pool is the same for every env: L2
    float* const hullp = reinterpret_cast<float*>(&S.con) + 48 * 4;
    const int nvert = mdl_nvert;
    if (CONVEX && nvert > 0) {
      for (int i = lane; i < nvert; i += NL) stv(hullp + 4 * i, *reinterpret_cast<const f4*>(m->hverts[i]));
    }
    WSYNC();
    int mycount = 0;
    int ncand = 0;
    if (enable_collision) {
      // broadphase: bounding test per static candidate pair, ordered compaction of survivors (lane = pair)
      int base = 0;
      bool any_plane = false, any_solid = false, any_slab = false, any_round = false;  // wave-uniform: which narrowphase loops have anything to do
      for (int p0 = 0; p0 < npair; p0 += NL) {
        int p = p0 + lane;
        bool hit = false, planepair = false, slab = false, roundpair = false;  // roundpair: the lane-private narrowphase takes it
        if (p < npair) {
          const int pr = (int)S.pairs[p];
          const int g1 = pr & 255, g2 = pr >> 8;
          planepair = (__float_as_int(S.gts[g1][0]) & 255) == MIR_GEOM_PLANE;
          V3 h2 = ld3(&S.gts[g2][1]);
          M3 R2 = q2m(ld4v(S.col.gquat[g2]));
          V3 c2 = ld3v(S.col.gpos[g2]);
          const int t1 = __float_as_int(S.gts[g1][0]) & 255, t2 = __float_as_int(S.gts[g2][0]) & 255;
          roundpair = CONVEX && (t1 == MIR_GEOM_PLANE ? t2 != MIR_GEOM_BOX : !(t1 == MIR_GEOM_BOX && t2 == MIR_GEOM_BOX));
          if (t1 == MIR_GEOM_PLANE) {
            V3 n = mcol(q2m(ld4v(S.col.gquat[g1])), 2);
            float ext = h2.x * fabsf(dot(n, mcol(R2, 0))) + h2.y * fabsf(dot(n, mcol(R2, 1))) + h2.z * fabsf(dot(n, mcol(R2, 2)));
            if (CONVEX && t2 == MIR_GEOM_SPHERE) ext = h2.x;
            if (CONVEX && t2 == MIR_GEOM_CAPSULE) ext = h2.y * fabsf(dot(n, mcol(R2, 2))) + h2.x;
            if (CONVEX && t2 == MIR_GEOM_HULL) ext = h2.z;  // (bounding sphere about the geom origin)
            hit = dot(c2 - ld3v(S.col.gpos[g1]), n) - ext < 0.0f;
          } else {
            V3 h1 = ld3(&S.gts[g1][1]);
            // bounding spheres (box: half diagonal; sphere: radius; capsule: half length + radius)
            float b1 = sqrtf(dot(h1, h1)), b2 = sqrtf(dot(h2, h2));
            if (CONVEX) {
              b1 = t1 == MIR_GEOM_SPHERE ? h1.x : (t1 == MIR_GEOM_CAPSULE ? h1.x + h1.y : (t1 == MIR_GEOM_HULL ? h1.z : b1));
              b2 = t2 == MIR_GEOM_SPHERE ? h2.x : (t2 == MIR_GEOM_CAPSULE ? h2.x + h2.y : (t2 == MIR_GEOM_HULL ? h2.z : b2));
            }
            float rs = b1 + b2;
            V3 dc = c2 - ld3v(S.col.gpos[g1]);
            hit = dot(dc, dc) <= rs * rs;
            if (CONVEX && hit && (t1 == MIR_GEOM_BOX) != (t2 == MIR_GEOM_BOX) && t1 != MIR_GEOM_HULL && t2 != MIR_GEOM_HULL) {
              // one box, one round geom: the round geom's CORE (centre, or the axis segment's bounding box in the box frame) against
              // the box itself (clamped distance <= radius), not bounding spheres -- the kitchen slab's is a metre wide, and link 1
              // stands a few centimetres above it for the whole episode: every such pair would run GJK every step
              const bool box1 = t1 == MIR_GEOM_BOX;
              const M3 Rb = box1 ? q2m(ld4v(S.col.gquat[g1])) : R2;
              const M3 Rr = box1 ? R2 : q2m(ld4v(S.col.gquat[g1]));
              const V3 hb = box1 ? h1 : h2, hr = box1 ? h2 : h1;
              const V3 dd = box1 ? dc : v3(-dc.x, -dc.y, -dc.z);  // round centre - box centre
              const bool cap = (box1 ? t2 : t1) == MIR_GEOM_CAPSULE;
              const V3 axr = mcol(Rr, 2);
              const float hl = cap ? hr.y : 0.0f;  // half length of the core segment
              const float ex = fmaxf(fabsf(dot(dd, mcol(Rb, 0))) - hb.x - hl * fabsf(dot(axr, mcol(Rb, 0))), 0.0f);
              const float ey = fmaxf(fabsf(dot(dd, mcol(Rb, 1))) - hb.y - hl * fabsf(dot(axr, mcol(Rb, 1))), 0.0f);
              const float ez = fmaxf(fabsf(dot(dd, mcol(Rb, 2))) - hb.z - hl * fabsf(dot(axr, mcol(Rb, 2))), 0.0f);
              hit = ex * ex + ey * ey + ez * ez <= hr.x * hr.x;
            }
            if (hit && (!CONVEX || (t1 == MIR_GEOM_BOX && t2 == MIR_GEOM_BOX))) {
              // the six face axes of the narrowphase's separating-axis test (same expressions): a pair they separate
              // would come back with zero contacts, and the narrowphase walks its candidates four at a time
              const M3 R1 = q2m(ld4v(S.col.gquat[g1]));
              const V3 A0 = mcol(R1, 0), A1 = mcol(R1, 1), A2 = mcol(R1, 2), B0 = mcol(R2, 0), B1 = mcol(R2, 1), B2 = mcol(R2, 2);
              // A box over the +z face of a STATIC box, well inside its footprint (a cube on the kitchen slab): with
              //   c = centre of B in A's frame,  c.z >= hA.z,  |c.x| + 2 rB <= hA.x,  |c.y| + 2 rB <= hA.y   (rB = B's half diagonal)
              // the overlap of the two boxes along ANY direction L is at least the overlap along A's z (overlap(L) - overlap(z) >=
              // 2 rB |L_xy| - rB |L - z| >= 0, support functions being rB-Lipschitz), so the separating-axis test of box_box_row would
              // pick A's +z face as the reference face -- first among ties -- and B's whole incident face lies over it: the pair goes
              // to the lane-private slab path (same vertices, same order, same expressions), not through the 15-axis test.
              if ((__float_as_int(S.gts[g1][0]) >> 16 & 1) != 0) {
                const float cx = dot(dc, A0), cy = dot(dc, A1), cz = dot(dc, A2);
                if (cz >= h1.z && fabsf(cx) + 2.0f * b2 <= h1.x && fabsf(cy) + 2.0f * b2 <= h1.y) {
                  slab = true;
                  const float extz = h2.x * fabsf(dot(B0, A2)) + h2.y * fabsf(dot(B1, A2)) + h2.z * fabsf(dot(B2, A2));
                  hit = cz - h1.z - extz <= 0.0f;  // (the test of axis A2 below, which is the only one that can separate here)
                }
              }
              const V3 Ls[6] = {A0, A1, A2, B0, B1, B2};
#pragma unroll
              for (int c = 0; c < 6; c++) {
                const V3 L = Ls[c];
                const float ra = h1.x * fabsf(dot(A0, L)) + h1.y * fabsf(dot(A1, L)) + h1.z * fabsf(dot(A2, L));
                const float rb = h2.x * fabsf(dot(B0, L)) + h2.y * fabsf(dot(B1, L)) + h2.z * fabsf(dot(B2, L));
                if (!slab && fabsf(dot(dc, L)) - (ra + rb) > 0.0f) hit = false;
              }
            }
          }
        }
        const unsigned long long bal = __ballot(hit);
        int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (hit && pos < NL) S.col.cand[pos] = p | (slab ? 1 << 16 : 0);  // (bit 16: the slab path takes this candidate)
        base += __popcll(bal);
        any_plane = any_plane || __ballot(hit && planepair) != 0ull;
        any_solid = any_solid || __ballot(hit && !planepair && !slab) != 0ull;
        any_slab = any_slab || __ballot(hit && slab) != 0ull;
        any_round = any_round || __ballot(hit && roundpair) != 0ull;
      }
      ncand = base < NL ? base : NL;
      if (lane == 0) S.ncand = ncand;
      WSYNC();
      STAMP(6);
      // narrowphase, plane-box: DPP row r takes candidates r, r + 4, ...; the 8 box corners on lanes 0..7 of the row
      // (skipped as a whole when no candidate has a plane: on the kitchen slab nothing reaches the floor)
      // (round 5: the candidates of the kind are ranked first, and trip t serves the ranks 4t .. 4t + 3 -- as many trips as a quarter of
      // their number, where positions r, r + 4, ... made one trip per group of four list positions that held any)
      unsigned long long kindm = 0ull;  // (wave-uniform: the walk below is scalar work)
      auto kth_of_row = [&]() -> int {  // takes the four lowest candidates off the mask: row r gets the r-th of them, or -1
        int k = -1;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int kr = kindm ? __ffsll(kindm) - 1 : -1;
          kindm &= kindm - 1ull;
          if (blk == r) k = kr;
        }
        return k;
      };
      if (any_plane) {
        bool mine = false;
        if (lane < ncand) {
          const int prl = (int)S.pairs[S.col.cand[lane] & 0xffff];
          mine = (__float_as_int(S.gts[prl & 255][0]) & 255) == MIR_GEOM_PLANE && (!CONVEX || (__float_as_int(S.gts[prl >> 8][0]) & 255) == MIR_GEOM_BOX);
        }
        kindm = __ballot(mine);
      }
      while (kindm) {
        const int k = kth_of_row();
        const bool isplane = k >= 0;
        const int pr = isplane ? (int)S.pairs[S.col.cand[k] & 0xffff] : 0;
        const int g1 = pr & 255, g2 = pr >> 8;
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
      // narrowphase, box on a static slab (see the broadphase): LANE-PRIVATE, lane c takes candidate c.  The face-contact branch of
      // box_box_row with reference face = A's +z and the incident face entirely over it: the penetrating vertices of B's most
      // anti-parallel face, in that routine's order (+,+), (-,+), (-,-), (+,-).  Five cubes on the slab used to be two trips of the
      // 15-axis routine, 11 k of the collision wave's 21 k cycles, with the other wave waiting for them.
      if (any_slab) {
        if (lane < ncand && (S.col.cand[lane] >> 16) != 0) {
          const int pr = (int)S.pairs[S.col.cand[lane] & 0xffff];
          const int g1 = pr & 255, g2 = pr >> 8;
          const M3 R1 = q2m(ld4v(S.col.gquat[g1])), R2 = q2m(ld4v(S.col.gquat[g2]));
          const BoxG A = {ld3v(S.col.gpos[g1]), mcol(R1, 0), mcol(R1, 1), mcol(R1, 2), ld3(&S.gts[g1][1])};
          const BoxG B = {ld3v(S.col.gpos[g2]), mcol(R2, 0), mcol(R2, 1), mcol(R2, 2), ld3(&S.gts[g2][1])};
          const V3 nr = A.a2;
          const V3 fc = A.p + A.h.z * nr;
          const float a0 = fabsf(dot(nr, B.a0)), a1 = fabsf(dot(nr, B.a1)), a2 = fabsf(dot(nr, B.a2));
          int jb = 0;
          float mx = a0;
          if (a1 > mx) { mx = a1; jb = 1; }
          if (a2 > mx) { mx = a2; jb = 2; }
          const float sj = dot(nr, bax(B, jb)) > 0.0f ? -1.0f : 1.0f;
          const int j1 = (jb + 1) % 3, j2 = (jb + 2) % 3;
          const V3 ic = B.p + (sj * bh(B, jb)) * bax(B, jb);
          const V3 e1 = A.a0, e2 = A.a1;
          int cnt = 0;
#pragma unroll
          for (int v = 0; v < 4; v++) {
            const float sxv = (v == 0 || v == 3) ? 1.0f : -1.0f, syv = v < 2 ? 1.0f : -1.0f;
            const V3 w = ic + (sxv * bh(B, j1)) * bax(B, j1) + (syv * bh(B, j2)) * bax(B, j2);
            const V3 rel = w - fc;
            const float vx = dot(rel, e1), vy = dot(rel, e2), vz = dot(rel, nr);
            if (vz < 0.0f) {
              const V3 wp = fc + vx * e1 + vy * e2 + (0.5f * vz) * nr;
              stv(S.col.stage[lane][cnt], f4{wp.x, wp.y, wp.z, vz});
              cnt++;
            }
          }
          if (cnt) st3v(S.col.snorm[lane], nr);
          S.col.ccount[lane] = cnt;
        }
        WSYNC();
      }
      // narrowphase, box-box: DPP row r takes candidates r, r + 4, ... (like plane-box).  The 15 separating axes sit on
      // lanes 0..14 of the row, the incident-face vertices on lanes 0..3 (box_box_row, mir_dev.h)
      if (any_solid) {
        bool mine = false;
        if (lane < ncand && (S.col.cand[lane] >> 16) == 0) {
          const int prl = (int)S.pairs[S.col.cand[lane] & 0xffff];
          const int t1l = __float_as_int(S.gts[prl & 255][0]) & 255, t2l = __float_as_int(S.gts[prl >> 8][0]) & 255;
          mine = t1l != MIR_GEOM_PLANE && (!CONVEX || (t1l == MIR_GEOM_BOX && t2l == MIR_GEOM_BOX));
        }
        kindm = __ballot(mine);
      }
#ifdef MIR_PROFILE_SINGLE
      STAMP(36);
      if (a.prof && blockIdx.x == 0 && lane == 0) a.prof[38] = (unsigned long long)__popcll(kindm);
#endif
      while (kindm) {
        const int k = kth_of_row();
        const bool isbox = k >= 0;
        const int pr = isbox ? (int)S.pairs[S.col.cand[k] & 0xffff] : 0;
        const int g1 = pr & 255, g2 = pr >> 8;
        if (isbox) {  // whole rows
          const M3 R1 = q2m(ld4v(S.col.gquat[g1])), R2 = q2m(ld4v(S.col.gquat[g2]));
          const BoxG A = {ld3v(S.col.gpos[g1]), mcol(R1, 0), mcol(R1, 1), mcol(R1, 2), ld3(&S.gts[g1][1])};
          const BoxG B = {ld3v(S.col.gpos[g2]), mcol(R2, 0), mcol(R2, 1), mcol(R2, 2), ld3(&S.gts[g2][1])};
          const int cnt = box_box_row(A, B, l16, lane, blk * G, S.col.stage[k], S.col.snorm[k], reinterpret_cast<float*>(&S.con) + 48 * blk);  // (contact arrays: not written yet)
          if (l16 == 0) S.col.ccount[k] = cnt;
        }
      }
#ifdef MIR_PROFILE_SINGLE
      STAMP(37);
#endif
      WSYNC();
      if constexpr (CONVEX) {
        // narrowphase of the round shapes, LANE-PRIVATE: lane c takes candidate c (see mir_step.hip).  Lanes diverge here and
        // reconverge at the end of the block.  (Skipped as a whole when no candidate has a round geom or a hull: wave-uniform.)
        if (any_round && lane < ncand) {
          const int pr = (int)S.pairs[S.col.cand[lane] & 0xffff];
          const int g1 = pr & 255, g2 = pr >> 8;
          const int t1 = __float_as_int(S.gts[g1][0]) & 255, t2 = __float_as_int(S.gts[g2][0]) & 255;
          if (t1 == MIR_GEOM_PLANE && (t2 == MIR_GEOM_SPHERE || t2 == MIR_GEOM_CAPSULE)) {
            const V3 n = mcol(q2m(ld4v(S.col.gquat[g1])), 2), pp = ld3v(S.col.gpos[g1]), pc = ld3v(S.col.gpos[g2]);
            const V3 sz = ld3(&S.gts[g2][1]);
            const float r = sz.x;
            int cnt = 0;
            if (t2 == MIR_GEOM_SPHERE) {
              const float dist = dot(pc - pp, n) - r;
              if (dist < 0.0f) { const V3 c = pc - (r + 0.5f * dist) * n; stv(S.col.stage[lane][0], f4{c.x, c.y, c.z, dist}); cnt = 1; }
            } else {  // the two end spheres, axis - then axis +
              const V3 ax = mcol(q2m(ld4v(S.col.gquat[g2])), 2);
#pragma unroll
              for (int sgn = -1; sgn <= 1; sgn += 2) {
                const V3 e = pc + ((float)sgn * sz.y) * ax;
                const float dist = dot(e - pp, n) - r;
                if (dist < 0.0f) { const V3 c = e - (r + 0.5f * dist) * n; stv(S.col.stage[lane][cnt], f4{c.x, c.y, c.z, dist}); cnt++; }
              }
            }
            if (cnt) st3v(S.col.snorm[lane], n);
            S.col.ccount[lane] = cnt;
          } else if (t1 == MIR_GEOM_PLANE && t2 == MIR_GEOM_HULL) {
            // plane - hull: the penetrating vertices in index order, reduced to four like plane - box (support extremes, first
            // index wins ties; oracle: plane_hull / reduce4; the same two lane-private passes as in the 16-lane kernel)
            const M3 Rp = q2m(ld4v(S.col.gquat[g1])), R2 = q2m(ld4v(S.col.gquat[g2]));
            const V3 n = mcol(Rp, 2), eu = mcol(Rp, 0), ev = mcol(Rp, 1), pp = ld3v(S.col.gpos[g1]), pc = ld3v(S.col.gpos[g2]);
            const int v0 = (int)S.gts[g2][1], nvg = (int)S.gts[g2][2];
            int npen = 0, pk0 = -1, pk1 = -1, pk2 = -1, pk3 = -1;
            float uM = 0.0f, um = 0.0f, vM = 0.0f, vm = 0.0f;
            for (int i = 0; i < nvg; i++) {
              const V3 l = ld3v(hullp + 4 * (v0 + i));
              const V3 rel = pc + l.x * mcol(R2, 0) + l.y * mcol(R2, 1) + l.z * mcol(R2, 2) - pp;
              if (dot(rel, n) < 0.0f) {
                const float u = dot(rel, eu), v = dot(rel, ev);
                if (npen == 0 || u > uM) { uM = u; pk0 = i; }
                if (npen == 0 || u < um) { um = u; pk1 = i; }
                if (npen == 0 || v > vM) { vM = v; pk2 = i; }
                if (npen == 0 || v < vm) { vm = v; pk3 = i; }
                npen++;
              }
            }
            int cnt = 0;
            for (int i = 0; i < nvg && cnt < 4; i++) {
              const V3 l = ld3v(hullp + 4 * (v0 + i));
              const V3 w = pc + l.x * mcol(R2, 0) + l.y * mcol(R2, 1) + l.z * mcol(R2, 2);
              const float d = dot(w - pp, n);
              if (d < 0.0f && (npen <= 4 || i == pk0 || i == pk1 || i == pk2 || i == pk3)) {
                const V3 c = w - (0.5f * d) * n;
                stv(S.col.stage[lane][cnt], f4{c.x, c.y, c.z, d});
                cnt++;
              }
            }
            if (cnt) st3v(S.col.snorm[lane], n);
            S.col.ccount[lane] = cnt;
          } else if (t1 != MIR_GEOM_PLANE && !(t1 == MIR_GEOM_BOX && t2 == MIR_GEOM_BOX)) {
            const M3 R1 = q2m(ld4v(S.col.gquat[g1])), R2 = q2m(ld4v(S.col.gquat[g2]));
            const V3 z1 = ld3(&S.gts[g1][1]), z2 = ld3(&S.gts[g2][1]);
            const ShapeD A = {t1, z1, ld3v(S.col.gpos[g1]), mcol(R1, 0), mcol(R1, 1), mcol(R1, 2), hullp + 4 * (t1 == MIR_GEOM_HULL ? (int)z1.x : 0), (int)z1.y};
            const ShapeD B = {t2, z2, ld3v(S.col.gpos[g2]), mcol(R2, 0), mcol(R2, 1), mcol(R2, 2), hullp + 4 * (t2 == MIR_GEOM_HULL ? (int)z2.x : 0), (int)z2.y};
            f4 pt;
            V3 n;
            if (convex_pair(A, B, pt, n)) {
              stv(S.col.stage[lane][0], pt);
              st3v(S.col.snorm[lane], n);
              S.col.ccount[lane] = 1;
            }
          }
        }
        WSYNC();
      }
      mycount = S.col.ccount[lane];
#ifdef MIR_PROFILE_SINGLE  /* (profiling build: candidates by narrowphase path, for tools/probes/cand_hist64.py) */
      {
        int kind = 0;
        if (lane < ncand) {
          const int cd = S.col.cand[lane];
          const int pr = (int)S.pairs[cd & 0xffff];
          const int t1 = __float_as_int(S.gts[pr & 255][0]) & 255, t2 = __float_as_int(S.gts[pr >> 8][0]) & 255;
          kind = (cd >> 16) ? 1 : (t1 == MIR_GEOM_PLANE ? (t2 == MIR_GEOM_BOX ? 2 : 3) : ((t1 == MIR_GEOM_BOX && t2 == MIR_GEOM_BOX) ? 4 : 5));
        }
        const int n1 = __popcll(__ballot(kind == 1)), n2 = __popcll(__ballot(kind == 2)), n3 = __popcll(__ballot(kind == 3)), n4 = __popcll(__ballot(kind == 4)),
                  n5 = __popcll(__ballot(kind == 5)), n5hit = __popcll(__ballot(kind == 5 && mycount > 0));
        if (lane == 0) S.pad1 = n1 | n2 << 5 | n3 << 10 | n4 << 15 | n5 << 20 | n5hit << 25;
      }
#endif
    }
    STAMP(8);
    // ordered compaction of the contact points: exclusive prefix over candidate lanes (convergent code)
    const int maxc = max_contacts < MAXC ? max_contacts : MAXC;
    int ncon;
    {
      // more candidate points than the capacity: the largest manifolds are thinned before any pair loses all of its points (the
      // rule is defined at oracle/orc_rigid.c: thin_manifolds; same code path as in the 16-lane kernel, one env per wave here)
      int total0 = (int)wsum((float)mycount);
      if (lane == 0) S.pad0 = total0;  // (candidate points before the capacity is applied, for the diagnostics record)
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
        const int pr = (int)S.pairs[S.col.cand[cl] & 0xffff];
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
        const int b1 = __float_as_int(S.gts[g1][0]) >> 8 & 255, b2 = __float_as_int(S.gts[g2][0]) >> 8 & 255;  // (bit 16: static-body flag)
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
    if (fk_split) {
      // the closing forward kinematics of the jointed bodies, while wave 0 integrates the free bodies' quaternions, writes their
      // poses and stores the state rows
      __syncthreads();  // (5) scalar joints integrated
      wave_fk<true>(S, lane, nb, bk);
      __syncthreads();  // (6) link poses of the new state
    }
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
      return (sqrtf(dx * dx + dy * dy) < m->reward_xy && above(po.z - p2.z, m->reward_dz)) ? 1.0f : 0.0f;
    }
    return above(po.z, m->reward_z) ? 1.0f : 0.0f;
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
  bool bad_acc = false;  // (diagnostics on) the env's state went non-finite in some step of this launch
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
    // ======================= velocities, composite inertias, body forces: tree SCANS (as in the 16-lane kernel) =========
    // Sums over the ancestors of a dof are inclusive prefix sums along its dof chain: POINTER JUMPING over the chain's parent
    // links (<= 4 rounds of one lane gather each) instead of a masked gather per lane and per quantity (those loops were a
    // dependent LDS round trip per ancestor: 11 k of this wave's 39 k cycles).  Sums over the subtree of a body are suffix sums
    // over the body lanes -- bodies are numbered in depth-first preorder and a tree's bodies share a DPP row (checked by
    // mir_compile64.cpp), so a subtree is the lane range [b, b_next) -- four DPP row shifts per component and one subtraction.
    float qfrc_bias = 0.0f, qfs = 0.0f;
    {
      auto ancestor_scan = [&](V3& A, V3& Bv, float* tab) {  // tab: NL rows of 8 floats, inclusive sums by dof lane
        int pj = isdof ? d_par : -1;
#pragma unroll 1
        for (int round = 0; round < 4; round++) {
          if (!__any(pj >= 0)) break;
          const int src = (pj >= 0 ? pj : lane) << 2;
          const V3 xa = v3(lane_gather(src, A.x), lane_gather(src, A.y), lane_gather(src, A.z));
          const V3 xb = v3(lane_gather(src, Bv.x), lane_gather(src, Bv.y), lane_gather(src, Bv.z));
          const int nxt = lane_gather(src, pj);
          if (pj >= 0) {
            A = A + xa;
            Bv = Bv + xb;
            pj = nxt;
          }
        }
        st3v(tab + 8 * lane, A);
        st3v(tab + 8 * lane + 4, Bv);
        WSYNC();
      };
      float* const tab = &S.dm.dyn.cddq[0][0];
      const V3 cw = ld3v(&S.cdof[lane][0]), cv = ld3v(&S.cdof[lane][4]);
      const float qd = isdof ? S.qvel[lane] : 0.0f;
      // (1) V_i = sum over the dof chain up to and including i of qvel_j cdof_j
      V3 Vw = isdof ? qd * cw : v3(0, 0, 0), Vv = isdof ? qd * cv : v3(0, 0, 0);
      ancestor_scan(Vw, Vv, tab);
      // cdof_dot * qvel from the velocity in front of the dof; body velocity = V at the last dof that moves the body
      V3 Yw = v3(0, 0, 0), Yv = v3(0, 0, 0);
      if (isdof) {
        V3 pw = v3(0, 0, 0), pv = v3(0, 0, 0);
        if (d_bef >= 0) { pw = ld3v(tab + 8 * d_bef); pv = ld3v(tab + 8 * d_bef + 4); }
        Yw = qd * cross(pw, cw);
        Yv = qd * (cross(pw, cv) + cross(pv, cw));
      }
      V3 w = v3(0, 0, 0), v = v3(0, 0, 0);
      if (isbody && b_last >= 0) { w = ld3v(tab + 8 * b_last); v = ld3v(tab + 8 * b_last + 4); }
      // (2) composite inertia: suffix sums of the body inertias over the row, minus the suffix behind the subtree
      const bool has_next = b_next >= 0;  // (the subtree ends inside the row)
      const int nsrc = (has_next ? b_next : lane) << 2;
      {
        const float* ci = S.dm.dyn.cinert[bl];
        const f4 c0 = ldv(ci), c1 = ldv(ci + 4), c2 = ldv(ci + 8);
        float comp[10] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y};
#pragma unroll
        for (int k = 0; k < 10; k++) {
          float t = lane < NB ? comp[k] : 0.0f;
          t += row_shl<1>(t); t += row_shl<2>(t); t += row_shl<4>(t); t += row_shl<8>(t);
          comp[k] = t;
        }
        float e[10];
#pragma unroll
        for (int k = 0; k < 10; k++) {
          const float gk = lane_gather(nsrc, comp[k]);
          e[k] = has_next ? gk : 0.0f;
        }
        if (lane < NB) {
          float* cs = S.dm.dyn.crb[lane];
          stv(cs, isbody ? f4{comp[0] - e[0], comp[1] - e[1], comp[2] - e[2], comp[3] - e[3]} : f4{0, 0, 0, 0});
          stv(cs + 4, isbody ? f4{comp[4] - e[4], comp[5] - e[5], comp[6] - e[6], comp[7] - e[7]} : f4{0, 0, 0, 0});
          stv(cs + 8, isbody ? f4{comp[8] - e[8], comp[9] - e[9], 0.0f, 0.0f} : f4{0, 0, 0, 0});
        }
      }
      WSYNC();  // (every read of the V table is done before the next scan overwrites it)
      STAMP(3);
      // (3) A_i = sum over the dof chain of cdof_dot_j qvel_j; body forces at zero acceleration (RNE)
      ancestor_scan(Yw, Yv, tab);
      V3 t = v3(0, 0, 0), f = v3(0, 0, 0);
      if (isbody) {
        V3 aw = v3(0, 0, 0), av = v3(-m->gx, -m->gy, -m->gz);
        if (b_last >= 0) { aw = aw + ld3v(tab + 8 * b_last); av = av + ld3v(tab + 8 * b_last + 4); }
        Inert I = ldI(S.dm.dyn.cinert[lane]);
        V3 ta, fa, tv, fv;
        imul(I, aw, av, ta, fa);
        imul(I, w, v, tv, fv);
        t = ta + cross(w, tv) + cross(v, fv);
        f = fa + cross(w, fv);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) stv(&S.dm.M[lane][4 * q], f4{0, 0, 0, 0});
      // (4) subtree forces: suffix sums again
      {
        float comp[6] = {t.x, t.y, t.z, f.x, f.y, f.z};
#pragma unroll
        for (int k = 0; k < 6; k++) {
          float u = comp[k];
          u += row_shl<1>(u); u += row_shl<2>(u); u += row_shl<4>(u); u += row_shl<8>(u);
          comp[k] = u;
        }
        float e[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const float gk = lane_gather(nsrc, comp[k]);
          e[k] = has_next ? gk : 0.0f;
        }
        if (lane < NB) {
          float* fs = S.dm.dyn.cfrc[lane];
          st3v(fs, v3(comp[0], comp[1], comp[2]) - v3(e[0], e[1], e[2]));
          st3v(fs + 4, v3(comp[3], comp[4], comp[5]) - v3(e[3], e[4], e[5]));
        }
      }
      WSYNC();
      if (isdof) {  // M[i][j] = cdof_j . (crb_body(i) cdof_i), j over ancestors-or-self (same tree => same block)
        Inert I = ldI(S.dm.dyn.crb[d_body]);
        V3 bt, bf;
        imul(I, cw, cv, bt, bf);
        uint64_t mk = d_ancmask;
        while (mk) {  // four ancestors per trip, reads batched
          int j[4];
          bool ok[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { ok[u] = mk != 0ull; j[u] = ok[u] ? __ffsll((unsigned long long)mk) - 1 : 0; mk &= mk - 1ull; }
          f4 ca[4], cl[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { ca[u] = ldv(&S.cdof[j[u]][0]); cl[u] = ldv(&S.cdof[j[u]][4]); }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (ok[u]) {
              float val = dot(v3(ca[u].x, ca[u].y, ca[u].z), bt) + dot(v3(cl[u].x, cl[u].y, cl[u].z), bf);
              if (j[u] == lane) val += d_mdiag;
              S.dm.M[lane][j[u] & 15] = val;
              S.dm.M[j[u]][l16] = val;
            }
        }
        // bias = cdof . (forces of the subtree of the dof's body); smooth force
        const V3 ft = ld3v(&S.dm.dyn.cfrc[d_body][0]), ff = ld3v(&S.dm.dyn.cfrc[d_body][4]);
        qfrc_bias = dot(cw, ft) + dot(cv, ff);
        float fa = 0.0f;
        if (d_ctrl == MIR_CTRL_POSITION) {
          fa = d_kp * (S.target[lane] - S.qpos[d_qadr]) - d_kv * qd;
          fa = fminf(fmaxf(fa, d_frclo), d_frchi);
        }
        qfs = -d_damping * qd + fa - qfrc_bias;
      }
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
      gj16_dpp<0>(arow, qas, l16);
    }
    S.qas[lane] = qas;
    S.qacc[lane] = qas;
    if (a.out_qas && isdof && step == 0) a.out_qas[(size_t)env * nv + m->d_dof[lane]] = qas;
    WSYNC();  // dyn scratch is dead from here on

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
    // Everything of the warm start that does not need the contacts, ahead of the meeting with the collision wave (this wave
    // arrives there first): the Gauss term of the warm-start candidate and Mt times either candidate.
    const float* xblk_srch = &S.srch[16 * blk];
    const float ws = S.qacc_ws[lane];
    const float dq = isdof ? ws - qas : 0.0f;
    S.srch[lane] = dq;
    WSYNC();
    const float gauss_ws = isdof ? 0.5f * rowdot(mrow, xblk_srch) * dq : 0.0f;
    const float Ma_qas = isdof ? rowdot(mrow, &S.qas[16 * blk]) : 0.0f, Ma_ws = isdof ? rowdot(mrow, &S.qacc_ws[16 * blk]) : 0.0f;
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
    WSYNC();
    if (DUAL) __syncthreads();  // (3) both halves of the Jacobian segments
    // contact rows, lane = contact, lane-private: aref_r = -b (J_r qvel) - k imp dist
    const bool iscon = lane < ncon;
    float cmu = 0.0f, cD = 0.0f;
    float aref[4] = {0, 0, 0, 0}, jar[4] = {0, 0, 0, 0};
    float cds[3] = {0, 0, 0}, cdw[3] = {0, 0, 0};  // J_c qacc_smooth, J_c warm start (used by the warm-start choice below)
    if (iscon) {
      float cdv[3];
      condot3x3(S, lane, S.qvel, S.qas, S.qacc_ws, cdv, cds, cdw);
      const float vn = cdv[0], v1 = cdv[1], v2 = cdv[2];
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
    {
      // warm start: cost(ws) vs cost(qacc_smooth); Gauss part 1/2 dq^T Mt dq
      float c_ws = gauss_ws, c_sm = 0.0f;
      const float ljs = lsg * qas - laref, ljw = lsg * ws - laref;
      if (lsg != 0.0f) {
        if (ljs < 0.0f) c_sm += 0.5f * lD * ljs * ljs;
        if (ljw < 0.0f) c_ws += 0.5f * lD * ljw * ljw;
      }
      float js[4] = {0, 0, 0, 0}, jw[4] = {0, 0, 0, 0};
      if (iscon) {
        const float sn = cds[0], s1 = cds[1], s2 = cds[2], wn = cdw[0], w1 = cdw[1], w2 = cdw[2];
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
      Ma = usews ? Ma_ws : Ma_qas;  // = rowdot(mrow, the chosen candidate)
    }
    STAMP(11);
    int niter = 0;
    const float tol = mdl_tolerance, scale = mdl_scale;
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
    // Newton Hessian H = Mt + J^T D_active J.  Its diagonal blocks are kept in REGISTERS across iterations and updated incrementally
    // (only rows whose active flag flipped contribute, as in the 16-lane kernel): lane = dof row, hd = the 16 columns of the row's
    // own block (M is block-diagonal).  The off-diagonal blocks are non-zero only when a contact couples two blocks this step
    // (comp != identity: the arm touches a cube, or two cubes of different blocks touch); they are formed inside the coupled solve.
    const bool coupled = comp != 0x8421u;
    // exactly one coupled pair of blocks (p < q)?  (wave-uniform; -1 otherwise)
    int pair_p = -1, pair_q = -1;
    {
      const unsigned off = comp & ~0x8421u;
      if (__popc(off) == 2) {
        const int bit = __ffs(off) - 1;  // lowest set bit = 4 p + q with p < q
        pair_p = bit >> 2; pair_q = bit & 3;
      }
    }
    STAMP(20);
    // In the single-step instantiation the solve is compiled twice: the block-diagonal case (no contact couples two blocks:
    // 73 % of the envs) carries no 64-wide working copy, i.e. ~65 registers less than the coupled case.  (The loop instantiations keep one run-time-switched copy: there the duplication cost more than it saved.)
    bool met4 = false;
    auto newton = [&](auto mode_t) {
    constexpr int MODE = decltype(mode_t)::value;  // 0: block-diagonal, 1: coupled, 2: decided at run time (loop instantiations)
    const bool cpl = MODE == 2 ? coupled : MODE == 1;
    float hd[G];
#pragma unroll
    for (int j = 0; j < G; j++) hd[j] = isdof ? mrow[j] : (j == l16 ? 1.0f : 0.0f);
    const unsigned long long twoblk = __ballot(iscon && S.con.cblk[lane < MAXC ? lane : 0][1] >= 0);  // contacts with two segments
    float oldlact = 0.0f;
    unsigned prevbits = 0u;
    float gprev = 0.0f;
    for (int it = 0; it < mdl_iterations; it++) {
      if (done) break;  // wave-uniform: one env per wave
      float lact = (lsg != 0.0f && ljar < 0.0f) ? lD : 0.0f;
      const float lf = -lact * ljar;
      bool flip_d = false, act_any = false;
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
        flip_d = bits != (it == 0 ? 15u : prevbits);  // (the diagonal blocks start from all rows active)
        act_any = bits != 0u;
        prevbits = bits;
      }
      // contacts whose active rows changed since the Hessian last saw them (bit = contact): the updates walk those only
      const unsigned long long flipd = __ballot(flip_d), actc = __ballot(act_any);
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
      if (it == 0) STAMP(13);
      // ---- Newton direction: H s = -g: four 16-wide DPP block solves side by side, or dense over the wave
      float sv = -g;
      bool dense = false;
      if constexpr (MODE != 0) dense = cpl;
      if (!dense) {
        float hb[G];
#pragma unroll
        for (int j = 0; j < G; j++) hb[j] = hd[j];
        gj16_dpp<0>(hb, sv, l16);
      }
      if constexpr (MODE != 0) {
        if (dense) {
          // The solve consumes its matrix: a working copy of the diagonal block, and the off-diagonal blocks built HERE, from zero, in
          // every iteration: J_r^T D_active J_q over the two-block contacts with an active row (wave-uniform walk; the rows of either
          // block take the other block's segment as columns).  They are nowhere else: persistent off-diagonal rows cost 64 registers
          // for blocks that are zero in most envs, and the walk is a few contacts long.
          if (pair_p >= 0) {
            // one coupled pair: the rows of either block carry ONE off-diagonal block, the other block's columns
            float hb[G], hx[G];
#pragma unroll
            for (int j = 0; j < G; j++) { hb[j] = hd[j]; hx[j] = 0.0f; }
            const bool inpair = blk == pair_p || blk == pair_q;
            // (segment 0 of contact c belongs to block p?  one ballot instead of a dependent LDS read per contact)
            const unsigned long long seg0p = __ballot(iscon && S.con.cblk[lane < MAXC ? lane : 0][0] == pair_p);
            for (unsigned long long tw = twoblk & actc; tw;) {  // two contacts per trip, every read of both in one batch
              const int cA = __builtin_amdgcn_readfirstlane(__builtin_ctzll(tw));
              tw &= tw - 1ull;
              const bool two = tw != 0ull;
              const int cB = two ? __builtin_amdgcn_readfirstlane(__builtin_ctzll(tw)) : cA;
              if (two) tw &= tw - 1ull;
              // my segment: segment 0 when (segment 0 is p's) == (I am in p)
              const int mA = (((seg0p >> cA) & 1ull) != 0ull) == (blk == pair_p) ? 0 : 1, mB = (((seg0p >> cB) & 1ull) != 0ull) == (blk == pair_p) ? 0 : 1;
              const float* segA = &S.Jb[cA][mA][0]; const float* osegA = &S.Jb[cA][mA ^ 1][0];
              const float* segB = &S.Jb[cB][mB][0]; const float* osegB = &S.Jb[cB][mB ^ 1][0];
              const f4 fbA = ldv(S.con.cfb[cA]), fbB = ldv(S.con.cfb[cB]);
              const f4 mtA = ldv(S.con.cmeta[cA]), mtB = ldv(S.con.cmeta[cB]);
              const float jnA = segA[l16], j1A = segA[16 + l16], j2A = segA[32 + l16];
              const float jnB = segB[l16], j1B = segB[16 + l16], j2B = segB[32 + l16];
              f4 xnA[4], x1A[4], x2A[4], xnB[4], x1B[4], x2B[4];
#pragma unroll
              for (int q = 0; q < 4; q++) {
                xnA[q] = ldv(osegA + 4 * q); x1A[q] = ldv(osegA + 16 + 4 * q); x2A[q] = ldv(osegA + 32 + 4 * q);
                xnB[q] = ldv(osegB + 4 * q); x1B[q] = ldv(osegB + 16 + 4 * q); x2B[q] = ldv(osegB + 32 + 4 * q);
              }
              __builtin_amdgcn_sched_barrier(0);
#define MIR_OFFDIAG(fb, mt, jn, j1, j2, xn, x1, x2, on)                                                                                   \
              {                                                                                                                             \
                const unsigned bits = (unsigned)fb.w & 15u;                                                                                 \
                const float mu = mt.x, D = (on) ? mt.y : 0.0f;                                                                              \
                const float a0 = (bits & 1u) ? D : 0.0f, a1 = (bits & 2u) ? D : 0.0f, a2 = (bits & 4u) ? D : 0.0f, a3 = (bits & 8u) ? D : 0.0f; \
                const float w0 = a0 + a1 + a2 + a3, w1 = mu * (a0 - a1), w2 = mu * (a2 - a3), w3 = mu * mu * (a0 + a1), w4 = mu * mu * (a2 + a3); \
                const float tn = jn * w0 + j1 * w1 + j2 * w2, t1 = jn * w1 + j1 * w3, t2 = jn * w2 + j2 * w4;                               \
                _Pragma("unroll") for (int q = 0; q < 4; q++) {                                                                             \
                  hx[4 * q + 0] += tn * xn[q].x + t1 * x1[q].x + t2 * x2[q].x;                                                              \
                  hx[4 * q + 1] += tn * xn[q].y + t1 * x1[q].y + t2 * x2[q].y;                                                              \
                  hx[4 * q + 2] += tn * xn[q].z + t1 * x1[q].z + t2 * x2[q].z;                                                              \
                  hx[4 * q + 3] += tn * xn[q].w + t1 * x1[q].w + t2 * x2[q].w;                                                              \
                }                                                                                                                           \
              }
              MIR_OFFDIAG(fbA, mtA, jnA, j1A, j2A, xnA, x1A, x2A, inpair)
              MIR_OFFDIAG(fbB, mtB, jnB, j1B, j2B, xnB, x1B, x2B, inpair && two)
#undef MIR_OFFDIAG
            }
            if (it == 0) STAMP(30);
            pair_solve(hb, hx, sv, blk, l16, lane, pair_p, pair_q, &S.hx3[0][0], S.srch);
          } else {
          float hw0[G], hw1[G], hw2[G], hw3[G];
          H4 hw{hw0, hw1, hw2, hw3};
#pragma unroll
          for (int j = 0; j < G; j++) {
            hw.c0[j] = blk == 0 ? hd[j] : 0.0f; hw.c1[j] = blk == 1 ? hd[j] : 0.0f;
            hw.c2[j] = blk == 2 ? hd[j] : 0.0f; hw.c3[j] = blk == 3 ? hd[j] : 0.0f;
          }
          for (unsigned long long tw = twoblk & actc; tw; tw &= tw - 1ull) {
            const int c = __builtin_amdgcn_readfirstlane(__builtin_ctzll(tw));
            const f4 fb = ldv(S.con.cfb[c]);
            const unsigned bits = (unsigned)fb.w & 15u;
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
              const float a0 = (bits & 1u) ? D : 0.0f, a1 = (bits & 2u) ? D : 0.0f, a2 = (bits & 4u) ? D : 0.0f, a3 = (bits & 8u) ? D : 0.0f;
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
              for (int j = 0; j < G; j++) {  // (selects, not a branch per block: `ob` differs between the two blocks of the contact)
                hw.c0[j] += ob == 0 ? d[j] : 0.0f; hw.c1[j] += ob == 1 ? d[j] : 0.0f;
                hw.c2[j] += ob == 2 ? d[j] : 0.0f; hw.c3[j] += ob == 3 ? d[j] : 0.0f;
              }
            }
          }
          coupled_solve(hw, sv, lane, comp, &S.hx3[0][0], S.srch);
          }
          if (it == 0) STAMP(31);
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
      for (int ls = 1; ls < mdl_ls_iterations && !lsdone; ls++) {
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
    if (SINGLE && a.cost_out && lane == 0) a.cost_out[env] = (coupled || niter >= 3) ? 1 : 0;
    if (a.diag && lane == 0) {
      a.diag[(size_t)env * 4 + 0] = ncon;
#ifndef MIR_PROFILE_SINGLE
      a.diag[(size_t)env * 4 + 1] = nefc;
#endif
      a.diag[(size_t)env * 4 + 2] = niter;
#ifdef MIR_PROFILE_SINGLE  /* (profiling build: cycles from this wave's entry to the end of its solve, and whether blocks coupled) */
      a.diag[(size_t)env * 4 + 1] = a.prof && a.prof[33] == 77ull ? S.pad1 : ((int)(__builtin_readcyclecounter() - t_entry) | (coupled ? 1 << 30 : 0));
#endif
      a.diag[(size_t)env * 4 + 3] = ncand | (S.pad0 & 255) << 8;
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
    }
    if (fk_split) {
      WSYNC();
      __syncthreads();  // (5) the scalar joints are integrated: wave 1 starts on the jointed bodies' kinematics
    }
    if (isdof && d_kind == 3 && d_axis_k == 0) {
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
    WSYNC();
    STAMP(17);
    if (a.diag) {
      // divergence guard (diagnostics on; as in the 16-lane kernel): a NaN or an Inf in the integrated state -> bit 30 of word 3 of
      // the env's diagnostics record, sticky over the steps of a launch, and the handle's counter; `terminated` is False for it
      const bool nf = nonfinite(S.qpos[lane]) || nonfinite(S.qvel[lane]);
      const bool bad = __any(nf);
      bad_acc = bad_acc || bad;
      if (lane == 0) {
        a.diag[(size_t)env * 4 + 3] = ncand | (S.pad0 & 255) << 8 | (bad_acc ? 1 << 30 : 0);
        if (bad && a.bad_count) atomicAdd(a.bad_count, 1u);
      }
    }
    if (a.term_host && term_early && lane == 0) {
      // the host-visible terminated byte leaves here, before the closing FK, the observation and the state stores: its trip over
      // PCIe runs under them (see the 16-lane kernel)
      const V3 po = ld3(&S.qpos[obj_qadr]);
      const float r0 = reward_of(po, obj2_qadr >= 0 ? ld3(&S.qpos[obj2_qadr]) : po);
      __hip_atomic_store(&a.term_host[a.env_list ? slot : env], (uint8_t)((r0 == 1.0f ? 1u : 0u) | a.term_tag << 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    // kinematics of the new state: observations of this step, and the next step's starting poses
    if (fk_split) {
      // (wave 1 has the jointed bodies; the free bodies' poses are their qpos rows -- the expressions of wave_fk -- and the state rows
      //  leave meanwhile: nothing in them depends on the kinematics)
      if (lane > 0 && lane < nb && bk.jtype == MIR_JNT_FREE) {
        st3v(S.xpos[lane], ld3(&S.qpos[bk.qadr]));
        st4v(S.xquat[lane], qnormalize(ld4(&S.qpos[bk.qadr + 3])));
      }
      store_state(S, lane, dof16, isdof);
      WSYNC();
      __syncthreads();  // (6)
    } else {
      wave_fk(S, lane, nb, bk);
    }
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
    __hip_atomic_store(&a.term_host[a.env_list ? slot : env], (uint8_t)((rew == 1.0f ? 1u : 0u) | a.term_tag << 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (a.poses) {  // (null in list mode: the handle's pose buffer has the 16-lane kernel's shape)
    if (lane < nb) {
      float* p = a.poses + ((size_t)env * 2 * NB + lane) * 4;
      *reinterpret_cast<f4*>(p) = ldv(S.xpos[lane]);
      *reinterpret_cast<f4*>(p + 4 * NB) = ldv(S.xquat[lane]);
    }
    if (lane == 0) a.fkvalid[env] = 1;
  }
  // ---- store state ---------------------------------------------------------------------------------
  if (!fk_split) {  // (with the split closing FK the state rows left while wave 1 ran the kinematics)
    store_state(S, lane, dof16, isdof);
  }
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
  if (a.convex) {
    if (single) hipLaunchKernelGGL((mir_step64_kernel<0, true>), dim3(a.B), dim3(128), 0, stream, a);  // two waves per env
    else if (plain_loop) hipLaunchKernelGGL((mir_step64_kernel<1, true>), dim3(a.B), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((mir_step64_kernel<2, true>), dim3(a.B), dim3(64), 0, stream, a);
  } else {
    if (single) hipLaunchKernelGGL((mir_step64_kernel<0, false>), dim3(a.B), dim3(128), 0, stream, a);  // two waves per env
    else if (plain_loop) hipLaunchKernelGGL((mir_step64_kernel<1, false>), dim3(a.B), dim3(64), 0, stream, a);
    else hipLaunchKernelGGL((mir_step64_kernel<2, false>), dim3(a.B), dim3(64), 0, stream, a);
  }
  return (int)hipGetLastError();
}

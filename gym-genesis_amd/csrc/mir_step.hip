// mir_step.hip — the env.step() hot path as one fused HIP kernel for gfx950 (MI355X).
//
// What it replaces: scene.step() + get_obs() + compute_reward() of the reference
// (/root/reference/gym_genesis/tasks/franka/cube_pick.py:122-181, env.py:61-69), i.e. the
// Genesis rigid-solver pipeline restated in SURVEY.md App. A.
//
// Mapping to CDNA4
//   * workgroup = ONE wave64 = 4 envs x 16 lanes; a 16-lane env group is exactly one DPP "row".
//     Inside a group lane i is "dof i", "body i", "contact i" or "geom i / i+16" depending on the
//     phase, so per-dof / per-body / per-contact solver state is lane-private (registers).
//   * cross-lane traffic inside a group uses DPP: row_ror adds for reductions and row_newbcast
//     for broadcasts (a few cycles each, no LDS round trip).
//   * the two dense solves per Newton iteration (M^-1 f, H^-1 g) are Gauss-Jordan eliminations
//     on register-resident matrix rows (lane i = row i), the pivot row travelling by
//     row_newbcast: no LDS, no sqrt, no forward/back substitution chains.
//   * kinematic-tree recursions are scans: pointer jumping over the parent links for sums over ancestors (forward
//     kinematics, velocities, bias accelerations), DPP suffix sums over the depth-first body lanes for sums over
//     subtrees -- no depth-serial chain; the parent table is a 64-bit register.
//   * per-env working data that other lanes must see (poses, motion subspaces, M, contact
//     Jacobians) lives in ~8 KB of LDS per env, phase-aliased, so 4 workgroups (16 envs) fit a CU
//     and all 4096 envs of the headline batch are co-resident on the 256 CUs.
//   * LDS accesses of a wave execute in order, so the phases of a wave are separated by a compiler-level fence only (WSYNC).
//     The single-step instantiations put a SECOND wave on the same four envs (collision detection, contact arrays, opening and
//     closing forward kinematics, body inertias beside the first wave's dynamics and solves); the two meet at s_barriers.
//   * HBM state is env-major (B, D): with 16 lanes per env a wave touches 4 contiguous 64-byte
//     rows, the coalesced pattern for this lane mapping.  228 B read + 341 B written per env-step (489 B algorithmic:
//     qpos, qvel, warm start, action in; qpos, qvel, warm start, targets, observations, reward, mask out).
//   * the wave runs alone on its SIMD at the headline batch (the batch bounds the occupancy), so the kernel is a serial chain
//     of LDS round trips: loops over contacts / tree masks issue the reads of two or four entries in one batch ahead of
//     the arithmetic, vectors that a whole row needs once travel by row broadcast instead of through LDS, and every global
//     read of a launch (tables, per-lane record, model scalars, state, action, cached poses) leaves before the first LDS
//     store -- one L2 round trip at the start.
//   * instantiations: <0> one step per launch (two waves, no step loop, no AGPRs), <1> K-step rollouts with packed rows, <2>
//     everything (per-stage outputs, pose refresh, profiling stamps); <3> / <4> the action-independent / action-dependent half
//     of a step through a scratch row in HBM, <5> the ROTATED launch of GenesisEnv.step -- <4> of this step followed by <3> of
//     the next in one kernel, so that the host has its `terminated` after half a step and the rest runs while it is between two
//     calls; with three contacts per lane (CPL = 3, exact contacts) <6> the whole step + the next step's <3> for a list of envs, <7> the
//     whole step for the batch (heavy phase), <9> / <10> the two halves as two launches (overflow runs), and <11> (CPL = 1) <5>'s first
//     pass alone for a list.  One step body serves all of them; built with -ffp-contract=on so that they agree bit for bit.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <stdint.h>

#include "mir_model.h"
#include "mir_step.h"
#include "mir_spec_pick.h"

#define G MIR_G
#include "mir_dev.h"
#include "mir_convex.h"
#define EPB 4       /* envs per block */
#define JST 52      /* floats per contact in Jb: 3 rows x 16 + 4 pad -> conflict-free ds_read_b128 across contact lanes */
#define MSTR 20     /* row stride of M in LDS (floats): 16-byte aligned rows, conflict-free b128 row reads */
#define JB_SKEW 8   /* see EnvLds::Jb_ */
#define JBROW(Sx, c) (&(Sx).Jb_[(c) * JST + jbs])
static_assert(K16_MAX_CONTACT == G, "lane c owns contact c (one contact per lane; the list instantiation for exact contacts keeps CPL = 3 per lane)");

// optional phase timestamps (debug): block 0, thread 0 records the shader clock at phase boundaries
#ifdef MIR_PROFILE_SINGLE
/* profiling build: the block to stamp is chosen by the host (slot 63 of the buffer), and every Newton iteration gets its own
 * eight slots from 64 on (tools/phase_profile.py) */
/* (prof_mute: the list instantiation's second pass -- the next step's action-independent half -- leaves the first pass's stamps alone and
 *  stamps its own end: slot 140 the main wave, 141 the collision wave) */
#define STAMP(k) do { if (a.prof && !prof_mute && (int)blockIdx.x == prof_blk && threadIdx.x == 0) a.prof[k] = __builtin_readcyclecounter(); } while (0)
#define ITSTAMP(it, k) do { if ((it) < 8) STAMP(64 + 8 * (it) + (k)); } while (0)
#define HSTAMP(k) do { if (a.prof && !prof_mute && (int)blockIdx.x == prof_blk && threadIdx.x == 64) a.prof[k] = __builtin_readcyclecounter(); } while (0)
#else
#define HSTAMP(k) do { } while (0)
#define STAMP(k) do { if (a.prof && blockIdx.x == 0 && threadIdx.x == 0) a.prof[k] = __builtin_readcyclecounter(); } while (0)
#define ITSTAMP(it, k) do { } while (0)
#endif

#ifdef MIR_PROFILE_SINGLE
#define TERMSTAMP() do { if (a.prof && threadIdx.x == 0 && (blockIdx.x & 7) == (unsigned)(prof_blk & 7)) atomicMax(&a.prof[130], (unsigned long long)__builtin_amdgcn_s_memrealtime()); } while (0)
#else
#define TERMSTAMP() do { } while (0)
#endif
namespace {

// ---------------------------------------------------------------------------------------------
// per-env LDS working set (~8 KB).  Three phase-local scratch areas share storage:
//   dyn  (FK .. smooth dynamics)   overlays   contact arrays + Jb
//   col  (collision detection)     overlays   Jb
struct DynScratch {
  float cddq[G][8];               // cdof_dot * qvel: ang(3) pad lin(3) pad
  float cinert[G][12], crb[G][12];
  float cvel[G][8], cfrc[G][8];
};
template <int CPL>
struct ColScratchT {
  float gpos[K16_MAX_GEOM][4], gquat[K16_MAX_GEOM][4];
  int cand[G];
  int cmap[G * CPL];              // contact slot -> candidate lane * 8 + point index
  union {
    float stage[G][8][4];         // narrowphase output per candidate pair: pos, dist
    struct {                      // sweep-and-prune scratch (dead before the narrowphase writes `stage`)
      float lo[K16_MAX_GEOM][4], hi[K16_MAX_GEOM][4];  // world AABB; lo.w = geom type as int bits
      int order[K16_MAX_GEOM];    // non-plane geoms sorted by lo.x
      unsigned hitrow[K16_MAX_GEOM];  // bit b of row a: geoms a < b overlap (AABB) and may collide
      int plist[K16_MAX_PAIR];    // overlapping pairs g1 | g2 << 8 (plane first), in the order of the static list
      int npl, nnp, pad0, pad1;
    } sap;
  };
  float snorm[G][4];
  int count[G];                   // contact points found per candidate (handed from the collision wave to the main wave)
  float clip[48];                 // polygon clipping scratch of the box-box routine (one row)
};
static_assert(sizeof(((ColScratchT<1>*)nullptr)->sap) <= sizeof(((ColScratchT<1>*)nullptr)->stage), "SAP scratch must fit under the staging area");
// CPL = contacts per lane: 1 everywhere but in the list instantiation for exact contacts (VARIANT 6), where lane c owns contacts
// c, c + 16, c + 32 (capacity 48 = MIR_MAX_CONTACT, what the wave-per-env kernel holds).
template <int CPL>
struct ContactArraysT {
  static constexpr int MAXCON = G * CPL;
  float cpos[MAXCON][4];          // pos, dist.  CPL > 1: ALSO the per-iteration base forces (cfb below) -- the positions are dead once the Jacobian rows are built
  float cfrm[MAXCON][12];         // normal, t1, t2 (4-padded)
  float cmeta[MAXCON][4];         // mu, D, k*imp*dist stash / unused, b stash
  float cref[MAXCON][8];          // reference points of body1 / body2 trees (4-padded)
  unsigned cmask[MAXCON][4];      // dof masks of body1, body2, chunk mask, pad
  float cfb[CPL == 1 ? MAXCON : 1][4];  // per-iteration base forces (n, t1, t2), active-row flags (as int bits); CPL > 1: see cpos
};
// (CPL > 1 only: bit c of conB[s] = contact 16 s + c moves dofs of the second tree only -- CPL == 1 keeps those flags in bits 1 .. 16 of
//  `coupled` and has no such words: an empty base, the env block of the one-contact-per-lane instantiations is unchanged)
//  Also CPL > 1 only: what the two waves need to SHARE the narrowphase's box - box trips -- the main wave idles at barrier (2) while the
//  collision wave takes one 5 k-cycle trip after the other, five of them where both fingertips stand on the floor around a cube held by
//  the pads: `bp_ready` (the collision wave has published the candidate list of pass bp_ready - 1), `bb_done` (the main wave has
//  stored the contact points and counts of ITS trips), a polygon-clipping scratch of the main wave's own.)
template <int CPL> struct ConBWords { int conB[CPL]; int bp_ready, bb_done; int conB_pad[(8 - (CPL + 2) % 4) % 4 + 0]; float clip2[48]; };
template <> struct ConBWords<1> {};
template <int CPL>
struct EnvLdsT : ConBWords<CPL> {
  static constexpr int MAXCON = G * CPL;
  float qpos[20], qvel[G], target[G], qacc_ws[G];
  float xpos[G][4], xquat[G][4];
  float cdof[G][8];               // ang(3) pad lin(3) pad
  float M[G][MSTR];
  int ncon, ncand, coupled /* some contact joins the two kinematic trees: the Newton Hessian is not block diagonal */;
  int cin_ready;  // two-wave instantiations: the collision wave has stored the body inertias of this step (main wave spins on it; VARIANT 6: of pass cin_ready - 1)
  // Phase-aliased working set.  `dyn` (smooth dynamics) and `col` (collision detection) are live AT THE SAME TIME in the
  // single-step instantiation, where a second wave of the workgroup detects collisions while the first one does the dynamics;
  // the contact arrays and the contact Jacobians take the place of both afterwards (CPL == 1: con never overlaps col -- the contact
  // finishing reads the staging area while it writes con; Jb does, and is written after col is dead.  CPL > 1: con reaches into col;
  // the contact finishing there computes all of a lane's contacts into registers first, and stores behind barrier (2)).
  union {
    struct {
      DynScratch dyn;
      ColScratchT<CPL> col;
    };
    struct {
      ContactArraysT<CPL> con;
      // contact Jacobian rows, row c at Jb_[c * JST + jbs]: the rows of the ODD envs of a wave start JB_SKEW floats later.  An env's block
      // is 2280 dwords = 8 mod 32 long, so the 16 consecutive dwords that the 16 lanes of two neighbouring envs read from the same
      // row with one ds_read_b32 (banks = dword mod 32, two envs per 32-lane group) would overlap in 8 banks: every such read -- three
      // per contact in the gradient loop of every Newton iteration -- took two LDS cycles instead of one.  The skew moves the odd env's
      // rows to the other 16 banks; the space comes out of the slack of this half of the union.
      float Jb_[MAXCON * JST + JB_SKEW];
    };
  };
};
typedef EnvLdsT<1> EnvLds;
static_assert(sizeof(EnvLds) == 2280 * 4, "the env block of the one-contact-per-lane instantiations (JB_SKEW above counts on 2280 dwords = 8 mod 32)");
static_assert(sizeof(ContactArraysT<1>) <= sizeof(DynScratch), "the contact arrays must not reach the collision staging area");
static_assert(sizeof(ContactArraysT<1>) + (G * JST + JB_SKEW) * sizeof(float) <= sizeof(DynScratch) + sizeof(ColScratchT<1>), "the skewed Jacobian rows must fit the union");
static_assert(EPB * sizeof(EnvLds) + sizeof(ModelTab) + K16_MAX_VERT * 16 <= 40960, "four workgroups per CU: 160 KB of LDS / 4 (hull vertices included)");
static_assert(EPB * sizeof(EnvLdsT<3>) + sizeof(ModelTab) + K16_MAX_VERT * 16 <= 81920, "the list instantiation (three contacts per lane): two workgroups per CU");

// body-lane constants needed by forward kinematics
struct BodyK {
  int jtype, qadr;
  V3 pos, axis;
  Q4 quat;
};

// forward kinematics of one env group: local joint transforms, then POINTER JUMPING over the parent links -- in every
// round a body composes the (partially composed) transform of its current ancestor pointer and inherits that body's
// pointer, so after r rounds it holds the product over 2^r ancestors: 4 rounds for any tree of up to 16 bodies instead of
// a dependent walk as long as the chain (10 links for a Panda finger).  `parents` packs the 16 parent indices, 4 bits each;
// the ancestor's transform and pointer are gathered lane to lane (ds_bpermute), so a round is one crossbar trip and no LDS fence.
// `row4` = byte offset of the group's first lane in the wave (64 x group).
// SKIPFREE: free-joint bodies are left out (neither read nor written): in the two-wave instantiation the closing FK runs on the
// collision wave while the main wave is still integrating the free bodies' quaternions, and writes their poses itself.
template <bool SKIPFREE = false, class ENV>
__device__ __forceinline__ void group_fk(ENV& S, int lane, int nb, uint64_t parents, const BodyK& k, int row4) {
  V3 P = v3(0, 0, 0);
  Q4 Qx = Q4{1, 0, 0, 0};
  int anc = 0;
  const bool mine = lane < nb && !(SKIPFREE && k.jtype == MIR_JNT_FREE);
  if (lane > 0 && lane < nb && mine) {
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
    anc = (int)((parents >> (4 * lane)) & 15u);
  }
#pragma unroll 1
  for (int round = 0; round < 4; round++) {
    if (!__any(anc > 0)) break;
    const int src = row4 + ((anc > 0 ? anc : lane) << 2);
    const V3 pa = v3(lane_gather(src, P.x), lane_gather(src, P.y), lane_gather(src, P.z));
    const Q4 qa = Q4{lane_gather(src, Qx.w), lane_gather(src, Qx.x), lane_gather(src, Qx.y), lane_gather(src, Qx.z)};
    const int nxt = lane_gather(src, anc);
    if (anc > 0) {
      P = pa + qrot(qa, P);
      Qx = qmul(qa, Qx);
      anc = nxt;
    }
  }
  if (mine) {
    st3v(S.xpos[lane], P);
    st4v(S.xquat[lane], Qx);
  }
  WSYNC();
}

// One env's rows along a Newton direction: this lane's share of the cost decrease of the step al * s in the 1-D model, and the
// number of its rows whose sign the step changes.  With a = min(x, 0) the cost of a row is 1/2 D a^2 and its change
// 1/2 D (a1 - a0)(a1 + a0), where a1 - a0 is the step d itself while the row stays active: never a difference of squares (a step
// below the resolution of jar must yield a correctly tiny improvement, not an absorbed one).
// (the contact rows' share and the limit row's share are returned apart: where the problem separates by tree, the lane's contact
//  and its dof may belong to different trees; _c = the four pyramid rows of one contact, _l = the lane's joint-limit row)
__device__ __forceinline__ void step_rows_c(float al, const float (&jr)[4], const float (&vr)[4], float cD, float& pimc, float& crossc) {
  pimc = 0.0f;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const float x0 = jr[r], d = al * vr[r], x1 = x0 + d;
    const float a0 = fminf(x0, 0.0f), a1 = fminf(x1, 0.0f);
    pimc -= 0.5f * cD * ((x0 < 0.0f && x1 < 0.0f) ? d : a1 - a0) * (a1 + a0);
  }
  crossc = 0.0f;
#pragma unroll
  for (int r = 0; r < 4; r++) crossc += ((jr[r] < 0.0f) != (jr[r] + al * vr[r] < 0.0f)) ? 1.0f : 0.0f;
}
__device__ __forceinline__ void step_rows_l(float al, float ljar, float ljv, float lD, float lsg, float& piml, float& crossl) {
  {
    const float x0 = ljar, d = al * ljv, x1 = x0 + d;
    const float a0 = fminf(x0, 0.0f), a1 = fminf(x1, 0.0f);
    piml = -(0.5f * lD * ((x0 < 0.0f && x1 < 0.0f) ? d : a1 - a0) * (a1 + a0));
  }
  crossl = ((ljar < 0.0f) != (ljar + al * ljv < 0.0f)) && lsg != 0.0f ? 1.0f : 0.0f;
}

// ---------------------------------------------------------------------------------------------
// SINGLE = one full step per launch without the rollout / autoreset / per-stage-output options (the headline launch): the
// step loop disappears at compile time, and with it the block of scalar-register spills that the loop structure forces in
// front of it (every launch-invariant value is otherwise saved before the loop and restored inside it).
// VARIANT 0 = SINGLE (above); 1 = the step loop for rollouts (no per-stage / debug outputs and no separate observation
// buffers: only packed rows), which keeps 11 pointers out of the scalar registers; 2 = everything.
// FEAT bit 0 (CONVEX) = the scene has sphere / capsule geoms: the closed-form plane cases and the lane-private GJK / MPR
// narrowphase (mir_convex.h) are compiled in.  FEAT bit 1 (SAP) = the candidate pairs come from a sweep-and-prune over the
// geoms' world AABBs instead of the static pair list (scenes whose static list would exceed K16_MAX_PAIR, e.g. with
// self-collision enabled).  Scenes of planes and boxes with a short static list run the instantiation without either, whose
// register allocation and schedule are therefore untouched by that code.
// FEAT bit 2 (SPEC) = the headline scene: its sizes and options (SpecPick, mir_spec_pick.h) are literals instead of model reads.
//
// DUAL (the single-step instantiation): the workgroup has TWO waves.  Collision detection (geom poses, broadphase, narrowphase)
// needs only the link poses, and so do the smooth dynamics (subspaces, CRB, RNE, mass matrix, smooth solve): wave 1 does the
// former while wave 0 does the latter, in disjoint LDS areas, between two workgroup barriers.  At 4096 envs there is otherwise
// ONE wave per SIMD that spends 60 % of its life waiting on LDS round trips; the second wave fills those slots and takes
// ~5 k cycles out of the ~50 k of a step.  Wave 1 also opens the launch with the forward kinematics of the stored state (its loads
// -- one qpos row, four quads of lane constants -- are back before wave 0's, which brings in the model table and everything else),
// and after the contacts it accumulates the all-rows-active Newton Hessian beside wave 0's warm start and first gradient: four
// barriers in all.  The step-loop instantiations keep one wave (they need the AGPRs a second wave per SIMD would have to give up);
// both run the same code in the same order of operations, so they agree bit for bit.
// VARIANT 3 / 4 = the two halves of a single step for GenesisEnv.step (StepArgs.phase 1 / 2): 3 is the action-independent
// half -- everything the two waves do up to the contact Jacobians -- whose results go to the `pre` buffer instead of staying in
// LDS; 4 picks them up and runs the rest on one wave (all-rows-active Hessian included).  The host launches 3 for the NEXT step
// right behind the current step, so that it runs while the host is between two env.step() calls.
// VARIANT 6 (CPL = 3) = the LIST instantiation for EXACT CONTACTS: the envs of StepArgs::env_list -- the ones a launch of the
// mir_step_begin path deferred because their narrowphase found more candidate points than lanes -- take the WHOLE step here (the
// fused launch's pass: two waves, dynamics beside collision detection) with three contacts per lane, capacity 48 = MIR_MAX_CONTACT
// (Genesis keeps every point of its candidate pairs: /root/reference/gym_genesis/tasks/franka/cube_pick.py:46), store state, targets,
// observations and their terminated bytes (byte k of StepArgs::term_host for list entry k), and then run the action-independent half
// of the NEXT step like the rotated launch's second pass, into the env's scratch row: one launch instead of the wave-per-env kernel
// on the list followed by VARIANT 3 on the list, four envs per workgroup instead of one.  80 KB of LDS per workgroup, two workgroups
// per CU, one wave per SIMD with the whole register file.  An env with more than 48 points or more than 16 candidate pairs is
// deferred AGAIN (bit 7 of its byte; nothing stored): the wave-per-env kernel (64 candidates) stays the fallback for those.
// VARIANT 7 (CPL = 3) = the same first pass ALONE, for the whole batch (no list, the regular terminated words): what mir_step_begin
// launches INSTEAD of the one-contact-per-lane kernel while most envs would be deferred anyway (the reference's expert holds 70 % of
// its envs above 16 points through its two grasp stages): one launch per step instead of a launch that computes garbage for the
// deferred majority followed by the list launch for them.  Bit 6 of a byte: the env had more points than StepArgs::over_cap (what
// the host decides on when to go back).
template <int VARIANT, int FEAT, int CPL = 1>
__global__ __launch_bounds__((VARIANT == 0 || VARIANT == 3 || VARIANT >= 5) ? 128 : 64)
__attribute__((amdgpu_waves_per_eu((VARIANT == 5 || VARIANT == 11) ? 2 : 1, (VARIANT == 5 || VARIANT == 11) ? 2 : (VARIANT >= 6 ? 1 : 10)))) void mir_step_kernel(StepArgs a) {
  // VARIANT 5 = both halves in one launch, ROTATED: first the action-dependent half of THIS step (from the pre buffer), then the
  // action-independent half of the NEXT one (into the pre buffer).  The host sees `terminated` after the first half; the second
  // runs while it is between two env.step() calls, without a second launch, a second prologue or a second forward kinematics
  // (the closing FK of this step is the opening FK of the next).  Needs the split closing FK (fk_free_leaf scenes).
  constexpr bool ROT = VARIANT == 5 || VARIANT == 8 || VARIANT == 9 || VARIANT == 11;
  // (VARIANT 11, CPL = 1 = the rotated launch's first pass alone for a LIST of envs: the second half of the step for the envs of an
  //  overflow run that are at most at 16 points -- one round of 40 KB workgroups beside VARIANT 9's list of the others)
  constexpr bool FIRSTONLY = VARIANT == 9 || VARIANT == 11;
  // VARIANT 9 / 10 (CPL = 3) = the two halves of a step with three contacts per lane, as TWO launches: what mir_step_begin launches for
  // the whole batch while some env is above 16 points (an OVERFLOW RUN).  9 = the second half of this step from the scratch rows -- an
  // env with 17 .. 48 contacts has its row in StepArgs::pre_big, written by the launch before -- up to the outputs and the terminated
  // bytes (the rotated launch's first pass, then it returns); 10 = the first half of the next step with the 48-point capacity (VARIANT 3
  // with three contacts per lane).  The bytes of EVERY env leave after a solver pass -- two rounds of short workgroups -- instead of
  // after two rounds of whole steps (the heavy phase) or a fused pass of the list instantiation behind the main launch's bytes; the first
  // halves run while the host is between two env.step calls.  (8 = both in one rotated launch: measured, no better than the heavy phase
  // -- the second round's bytes wait for the first round's first halves -- and not instantiated.)
  constexpr bool BIGV = VARIANT >= 6 && VARIANT <= 10;  // three contacts per lane: the whole step in one pass (two waves, as VARIANT 0) ...
  constexpr bool BIG2 = VARIANT == 6;                  // ... then the outputs, then the action-independent half of the next step (the list instantiation)
  constexpr bool PRE = VARIANT == 3 || VARIANT == 10, POST = VARIANT == 4;
  constexpr bool SINGLE = VARIANT == 0 || PRE || POST || ROT || BIGV;
  constexpr bool DUAL = VARIANT == 0 || PRE || ROT || BIGV;
  constexpr int MAXCON = G * CPL;
  static_assert(CPL == 1 || BIGV, "several contacts per lane: the list instantiation only");
  static_assert(JST == 52 && K16_PRE_STRIDE >= K16_PRE_JB + JST * K16_MAX_CONTACT, "pre-buffer layout");
  typedef EnvLdsT<CPL> EnvLds;
  constexpr bool CONVEX = (FEAT & 1) != 0, SAP = (FEAT & 2) != 0, SPEC = (FEAT & 4) != 0;
  __shared__ __attribute__((aligned(16))) EnvLds s_env[EPB];
#define CFB(Sx, c) (CPL == 1 ? (Sx).con.cfb[CPL == 1 ? (c) : 0] : (Sx).con.cpos[c])
  __shared__ __attribute__((aligned(16))) ModelTab T;  // dynamically indexed model tables, one copy per workgroup
  // hull vertices (MIR_GEOM_HULL), convex instantiations only: what is left of the workgroup's 40 KB
  __shared__ __attribute__((aligned(16))) float s_hull[(FEAT & 1) ? K16_MAX_VERT : 1][4];
  // three contacts per lane: the collision wave HELPS in the Newton loop (it would wait at barrier (5) meanwhile) -- the contacts of the
  // slots above the first are its share of the gradient and of the Hessian update.  [0] request (iteration + 1, -1 = the loop is over),
  // [1] its gradient share of that iteration is in LDS, [2] its Hessian share is (see helper_newton)
  __shared__ int s_help[CPL > 1 ? 4 : 1];
  const DevModel* __restrict__ m = a.model;
  const int tid = threadIdx.x & 63;   // lane within the wave
  const int wave = threadIdx.x >> 6;  // 0 = main wave; 1 = collision wave (DUAL only)
#ifdef MIR_PROFILE_SINGLE
  const int prof_blk = a.prof ? (int)a.prof[63] : -1;
#else
  const int prof_blk = 0;
#endif
  [[maybe_unused]] bool prof_mute = false;
  STAMP(24);
  if (a.prof && (int)blockIdx.x == prof_blk && threadIdx.x == 0) a.prof[26] = __builtin_amdgcn_s_memrealtime();
  // Prologue: EVERY global read of the launch -- model table, per-lane constants, state rows, action, cached poses -- is
  // issued before the first LDS store, so the launch starts with one L2 round trip (the compiler otherwise kept three: the
  // stores of one group sat in front of the loads of the next).
  constexpr int TAB_NQ = (int)(sizeof(ModelTab) / 16), TAB_NPASS = (TAB_NQ + 63) / 64;
  f4 tabtmp[TAB_NPASS];
  if (wave == 0) {  // (DUAL: the collision wave fetches the qpos row and runs the forward kinematics meanwhile)
    const f4* src = reinterpret_cast<const f4*>(&m->tab);
#pragma unroll
    for (int k = 0; k < TAB_NPASS; k++) tabtmp[k] = src[min(tid + 64 * k, TAB_NQ - 1)];
  }
  f4 hulltmp = {0, 0, 0, 0};
  if ((FEAT & 1) && wave == 0) hulltmp = reinterpret_cast<const f4*>(&m->hverts[0][0])[min(tid, K16_MAX_VERT - 1)];
  const int lane = tid & (G - 1);
  const int row4 = (tid & ~(G - 1)) << 2;  // byte offset of this env's first lane in the wave (lane_gather)
  const int grp = tid >> 4;
  const int jbs = (grp & 1) * JB_SKEW;  // (EnvLds::Jb_)
  const int env_raw = blockIdx.x * EPB + grp;
  const bool valid = env_raw < a.B;
  int env = valid ? env_raw : a.B - 1;
  // (exact contacts: the action-independent half for a LIST of envs -- the ones the wave kernel has just stepped; see StepArgs::env_list)
  if constexpr (VARIANT == 3 || VARIANT == 11 || BIGV) { if (a.env_list) env = a.env_list[env]; }
  EnvLds& S = s_env[grp];

  // (SPEC: the headline scene's sizes and options are literals -- SpecPick, emitted by mir_compile into mir_spec_pick.h -- so
  // `lane < nv`, the one-trip geom / pair loops and the solver dispatch fold at compile time)
  const int nb = SPEC ? SpecPick::nbody : m->nbody, nv = SPEC ? SpecPick::nv : m->nv, qst = a.qst;
  const int ngeom = SPEC ? SpecPick::ngeom : m->ngeom, npair = SPEC ? SpecPick::npair : m->npair;
  // (the list instantiation for exact contacts holds MIR_MAX_CONTACT points whatever capacity the scene gives the other launches)
  const int max_contacts = BIGV ? MAXCON : (SPEC ? SpecPick::max_contacts : m->max_contacts), enable_collision = SPEC ? SpecPick::enable_collision : m->enable_collision;
  const float dt = m->dt;
  // Every scalar of the model that the step reads is fetched HERE, with the first batch of loads.  A read through `m` further
  // down cannot be hoisted by the compiler above the wave fences that separate the phases, so it would sit where it is used
  // -- an L2 round trip in the middle of the serial chain (m->iterations was re-read in every Newton iteration).
  const int mdl_iterations = SPEC ? SpecPick::iterations : m->iterations, mdl_ls_iterations = SPEC ? SpecPick::ls_iterations : m->ls_iterations;
  const int mdl_eef = SPEC ? SpecPick::eef_body : m->eef_body, mdl_obj = SPEC ? SpecPick::obj_body : m->obj_body, mdl_ngrip = SPEC ? SpecPick::n_grip : m->n_grip;
  const int mdl_split = SPEC ? SpecPick::gj_split : m->gj_split;
  const float mdl_tolerance = m->tolerance, mdl_scale = m->solver_scale, mdl_reward_z = m->reward_z;
  const float mdl_gx = m->gx, mdl_gy = m->gy, mdl_gz = m->gz;
  // ---- the scratch row of the split step, PACKED (it travels through HBM twice per env.step): a mass-matrix row keeps the
  // 16-byte quads of its own tree block only (9 + 6 dofs: 3 quads for the arm's rows, 2 for the cube's, 624 B instead of 1 KB), a
  // contact's three Jacobian rows keep the quads in which one of its two bodies has a dof (a cube on the floor: 2 of 4) and no
  // bank-conflict pad.  Which quads a contact keeps is 4 bits in the row's head word.
  const int pk_split = mdl_split, pk_nvq = (nv + 3) >> 2;
  const bool pk_rowA = pk_split > 0 && lane < pk_split, pk_rowB = pk_split > 0 && lane >= pk_split;
  const int pk_qlo = pk_rowB ? pk_split >> 2 : 0, pk_qhi = pk_rowA ? (pk_split + 3) >> 2 : pk_nvq;  // quads [qlo, qhi) of this lane's row
  const int pk_moff = 4 * (pk_rowB ? pk_split * ((pk_split + 3) >> 2) + (lane - pk_split) * (pk_nvq - (pk_split >> 2)) : lane * (pk_qhi - pk_qlo));
  // Jacobian rows of nc contacts from LDS to the scratch row; returns the quad masks (4 bits per contact).  Four contacts per trip:
  // every LDS read of the trip is issued before the first global store (one round trip per trip, not per contact).
  auto jrows_store = [&](float* pre, int nc, auto& Sx) -> uint64_t {
    uint64_t qm = 0ull;
    int off = 0;
    const int r = lane >> 2, q = lane & 3;
    for (int c0 = 0; c0 < nc; c0 += 4) {
      uint32_t mk[4];
      f4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int c = c0 + u < nc ? c0 + u : c0;
        mk[u] = c0 + u < nc ? Sx.con.cmask[c][2] : 0u;  // (the contact's 4-dof chunks: the quads in which one of its bodies has a dof)
        v[u] = ldv(JBROW(Sx, c) + (lane < 12 ? 16 * r + 4 * q : 0));
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int nq = __popc(mk[u]);
        if (lane < 12 && ((mk[u] >> q) & 1u)) *reinterpret_cast<f4*>(pre + K16_PRE_JB + 4 * (off + r * nq + __popc(mk[u] & ((1u << q) - 1u)))) = v[u];
        off += 3 * nq;
        qm |= (uint64_t)mk[u] << (4 * (c0 + u));
      }
    }
    return qm;
  };
  // ... and back: the loads of four contacts in flight together (addresses follow from the head word alone), then their LDS stores;
  // a quad the contact did not keep reads the row's first quad and is zeroed by a select (no divergent branch around a load)
  auto jrows_load = [&](const float* pre, int nc, uint64_t qm, auto& Sx) {
    int off = 0;
    const int r = lane >> 2, q = lane & 3;
    for (int c0 = 0; c0 < nc; c0 += 4) {
      f4 v[4];
      bool on[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const uint32_t mk = c0 + u < nc ? (uint32_t)(qm >> (4 * (c0 + u))) & 15u : 0u;
        const int nq = __popc(mk);
        on[u] = lane < 12 && ((mk >> q) & 1u);
        v[u] = *reinterpret_cast<const f4*>(pre + K16_PRE_JB + 4 * (on[u] ? off + r * nq + __popc(mk & ((1u << q) - 1u)) : 0));
        off += 3 * nq;
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (lane < 12 && c0 + u < nc) stv(JBROW(Sx, c0 + u) + 16 * r + 4 * q, on[u] ? v[u] : f4{0, 0, 0, 0});
    }
  };
  // ... and the BIG rows (three contacts per lane): the Jacobian rows of nc <= 48 contacts unpacked, 48 floats each, twelve lanes an f4
  // each, four contacts per trip
  auto bigrows_store = [&](float* big, int nc, auto& Sx) {
    const int r = lane >> 2, q = lane & 3;
    for (int c0 = 0; c0 < nc; c0 += 4) {
      f4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = ldv(JBROW(Sx, c0 + u < nc ? c0 + u : c0) + (lane < 12 ? 16 * r + 4 * q : 0));
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (lane < 12 && c0 + u < nc) *reinterpret_cast<f4*>(big + K48_JB + 48 * (c0 + u) + 16 * r + 4 * q) = v[u];
    }
  };
  // (twelve contacts per trip: the loads of a trip are in flight together, and a trip is an HBM round trip on the critical path of the
  //  launch's second half -- the solver waits for these rows; four per trip made it twelve round trips at 48 contacts)
  auto bigrows_load = [&](const float* big, int nc, auto& Sx) {
    const int r = lane >> 2, q = lane & 3;
    for (int c0 = 0; c0 < nc; c0 += 12) {
      f4 v[12];
#pragma unroll
      for (int u = 0; u < 12; u++) v[u] = *reinterpret_cast<const f4*>(big + K48_JB + 48 * (c0 + u < nc ? c0 + u : c0) + (lane < 12 ? 16 * r + 4 * q : 0));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < 12; u++)
        if (lane < 12 && c0 + u < nc) stv(JBROW(Sx, c0 + u) + 16 * r + 4 * q, v[u]);
    }
  };
  // DUAL: the closing FK is split between the waves when every free-joint body is a childless child of the world (wave-uniform)
  const bool fk_free_leaf = SPEC ? SpecPick::fk_free_leaf != 0 : m->fk_free_leaf != 0;
  const bool fksplit = DUAL && fk_free_leaf;
  // the task's object is a free body hanging off the world: its height -- all that `terminated` needs -- is a qpos entry, final as
  // soon as the translations are integrated, so the host-visible bytes can leave before the closing FK (wave-uniform)
  const int mdl_obj_qadr = SPEC ? SpecPick::obj_qadr : m->obj_qadr;
  const bool term_early = VARIANT != 1 && fk_free_leaf && mdl_obj_qadr >= 0;
  // ... and before the solver has converged where the mask provably cannot change any more (see mir_model.h: term_bound_ok)
  // (VARIANT 9, the second-half launch of an overflow run: two rounds of workgroups, and the step's bytes wait for the second round's --
  //  from inside the solver loop they leave at the first gradient instead of behind the last Newton iteration of 17 - 48 contacts)
  const bool term_bound = (VARIANT == 0 || VARIANT == 5 || VARIANT == 9 || VARIANT == 11) && term_early && m->term_bound_ok != 0 && a.term_host != nullptr && !a.no_early_mask;
  const int term_zlane = SPEC ? SpecPick::term_zlane : m->term_zlane;
  // exact contacts (StepArgs::exact): an env whose candidate contact points exceed this is DEFERRED to the wave kernel -- the launch
  // computes on (its lanes cannot leave the wave) but stores nothing for it and flags its terminated byte (wave-uniform; never without the flag)
  // (StepArgs::exact == 2, a test switch: EVERY env is deferred -- the whole batch then takes the list instantiation; the list
  //  instantiation itself defers what exceeds ITS capacity, to the wave-per-env kernel)
  constexpr bool DEFER = VARIANT == 0 || VARIANT == 4 || VARIANT == 5 || VARIANT == 11 || BIGV;
  const int defer_above = BIGV ? MAXCON : ((DEFER && a.exact) ? (a.exact == 2 ? -1 : (max_contacts < MAXCON ? max_contacts : MAXCON)) : 0x7fffffff);
  bool ovf_env = false;  // this lane's env is deferred: set where the step reads the `coupled` word (uniform over the env's row)
  bool over_env = false; // (three contacts per lane: the env had more candidate points than StepArgs::over_cap -- bit 6 of its terminated byte)

  // ---- collision detection: geom poses, broadphase, narrowphase into the staging area; returns this lane's point count
  // (lane = candidate).  Needs the link poses and the model table in LDS, nothing else: in the DUAL instantiation the second
  // wave of the workgroup runs it while the first one does the smooth dynamics.
  // (more candidate PAIRS passed the broadphase than the row has lanes: the pairs beyond the 16th were dropped.  Never seen on the
  //  reference's scenes -- the contact counts agree with the oracle's, which has no such limit, in every parity test -- but with exact
  //  contacts such an env must go to the wave kernel (64 candidates) like one with too many points: its count is reported saturated)
  bool pair_ovf = false;
  // narrowphase, box-box: one candidate at a time on the whole row (box_box_row, mir_dev.h): the 15 separating axes on lanes 0..14, the
  // incident-face vertices on lanes 0..3.  `bm`: the env's candidates to take, one bit per list position; the point count goes to the
  // candidate's lane (the collision wave's own trips) or to S.col.count (the main wave's, three contacts per lane).
  auto box_trips = [&](uint32_t bm, float* clipbuf, int& mycount, bool to_lds) {
    while (__any(bm != 0u)) {
      const bool isbox = bm != 0u;
      const int k = isbox ? __ffs(bm) - 1 : 0;
      bm &= bm - 1u;
      const int pr = isbox ? S.col.cand[k] : 0;
      const int g1 = pr & 255, g2 = pr >> 8 & 255;
      if (isbox) {  // whole rows
        const M3 R1 = q2m(ld4v(S.col.gquat[g1])), R2 = q2m(ld4v(S.col.gquat[g2]));
        const BoxG B1 = {ld3v(S.col.gpos[g1]), mcol(R1, 0), mcol(R1, 1), mcol(R1, 2), ld3v(T.g_size[g1])};
        const BoxG B2 = {ld3v(S.col.gpos[g2]), mcol(R2, 0), mcol(R2, 1), mcol(R2, 2), ld3v(T.g_size[g2])};
        const int cnt = box_box_row(B1, B2, lane, tid, grp * G, S.col.stage[k], S.col.snorm[k], clipbuf);
        if (to_lds) { if (lane == 0) S.col.count[k] = cnt; }
        else if (lane == k) mycount = cnt;
      }
    }
  };
  // the set bits of m dealt out alternately: first, third, ... to a, second, fourth, ... to b
  auto split_alternate = [](uint32_t m, uint32_t& a_, uint32_t& b_) {
    a_ = 0u; b_ = 0u;
    bool odd = false;
    while (m) {
      const uint32_t low = m & (0u - m);
      if (odd) b_ |= low; else a_ |= low;
      m ^= low;
      odd = !odd;
    }
  };
  // which candidates of the env go through the 15-axis routine (lane = list position; the same expression in both waves)
  auto box_candidates = [&](int ncand) -> uint32_t {
    const int prl = lane < ncand ? S.col.cand[lane] : 0;
    const bool boxl = lane < ncand && (prl >> 16 & 255) != MIR_GEOM_PLANE && (!CONVEX || ((prl >> 16 & 255) == MIR_GEOM_BOX && (prl >> 24) == MIR_GEOM_BOX));
    return (uint32_t)(__ballot(boxl) >> (grp * G)) & 0xffffu;
  };
  // (three contacts per lane) the main wave's share of the box - box trips: it has finished the smooth dynamics and would wait at barrier (2)
  auto box_share = [&](int pass_no) {
    if constexpr (CPL > 1) {
      while (__hip_atomic_load(&S.bp_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < pass_no) __builtin_amdgcn_s_sleep(1);
      uint32_t otherA, mineB;
      split_alternate(box_candidates(S.ncand), otherA, mineB);
      int unused = 0;
      box_trips(mineB, S.clip2, unused, true);
      WSYNC();
      if (lane == 0) __hip_atomic_store(&S.bb_done, pass_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  auto collide_detect = [&](int pass_no) -> int {
  if (lane == 0) { S.ncon = 0; S.ncand = 0; }
  for (int g = lane; g < ngeom; g += G) {
    int gb = T.g_info[g][0];
    Q4 qb = ld4v(S.xquat[gb]);
    st3v(S.col.gpos[g], ld3v(S.xpos[gb]) + qrot(qb, ld3v(T.g_pos[g])));
    st4v(S.col.gquat[g], qmul(qb, ld4v(T.g_quat[g])));
  }
  WSYNC();
  STAMP(11);
  HSTAMP(58);
  int mycount = 0;
  if (enable_collision) {
    int npl = npair;  // pairs that reach the bounding test below
    if constexpr (SAP) {
      // ---- sweep and prune over world AABBs (two geoms per lane: g = lane, lane + 16) ----
      // (a) AABB of every geom; planes are unbounded and are tested against every geom's AABB directly in (c)
      for (int g = lane; g < ngeom; g += G) {
        const int tg = T.g_info[g][1];
        const V3 c = ld3v(S.col.gpos[g]), hz = ld3v(T.g_size[g]);
        const M3 R = q2m(ld4v(S.col.gquat[g]));
        V3 e = v3(hz.x, hz.x, hz.x);  // sphere
        if (tg == MIR_GEOM_BOX) e = v3(fabsf(R.r0.x) * hz.x + fabsf(R.r0.y) * hz.y + fabsf(R.r0.z) * hz.z, fabsf(R.r1.x) * hz.x + fabsf(R.r1.y) * hz.y + fabsf(R.r1.z) * hz.z,
                                       fabsf(R.r2.x) * hz.x + fabsf(R.r2.y) * hz.y + fabsf(R.r2.z) * hz.z);
        else if (tg == MIR_GEOM_CAPSULE) e = v3(fabsf(R.r0.z) * hz.y + hz.x, fabsf(R.r1.z) * hz.y + hz.x, fabsf(R.r2.z) * hz.y + hz.x);
        else if (tg == MIR_GEOM_HULL) e = v3(T.g_size[g][3], T.g_size[g][3], T.g_size[g][3]);  // (bounding sphere)
        stv(S.col.sap.lo[g], f4{c.x - e.x, c.y - e.y, c.z - e.z, __int_as_float(tg)});
        stv(S.col.sap.hi[g], f4{c.x + e.x, c.y + e.y, c.z + e.z, 0.0f});
        S.col.sap.hitrow[g] = 0u;
      }
      WSYNC();
      // (b) sort the bounded geoms by lo.x: the rank of a geom is the number of geoms in front of it (ties by index)
      int nnp = 0;
      for (int g = lane; g < ngeom; g += G) {
        const f4 me = ldv(S.col.sap.lo[g]);
        int rank = 0;
        for (int h = 0; h < ngeom; h++) {
          const f4 ot = ldv(S.col.sap.lo[h]);
          if (__float_as_int(ot.w) != MIR_GEOM_PLANE && (ot.x < me.x || (ot.x == me.x && h < g))) rank++;
        }
        if (__float_as_int(me.w) != MIR_GEOM_PLANE) S.col.sap.order[rank] = g;
      }
      for (int h = 0; h < ngeom; h++) nnp += __float_as_int(S.col.sap.lo[h][3]) != MIR_GEOM_PLANE ? 1 : 0;
      WSYNC();
      // (c) sweep: the geom at sorted position p meets those behind it until one starts beyond its end
      for (int p = lane; p < nnp; p += G) {
        const int g = S.col.sap.order[p];
        const f4 lg = ldv(S.col.sap.lo[g]), hg = ldv(S.col.sap.hi[g]);
        const unsigned allow = T.g_allow[g];
        for (int q = p + 1; q < nnp; q++) {
          const int h = S.col.sap.order[q];
          const f4 lh = ldv(S.col.sap.lo[h]);
          if (lh.x > hg.x) break;
          const f4 hh = ldv(S.col.sap.hi[h]);
          if ((allow >> h & 1u) && lh.y <= hg.y && lg.y <= hh.y && lh.z <= hg.z && lg.z <= hh.z)
            atomicOr(&S.col.sap.hitrow[g < h ? g : h], 1u << (g < h ? h : g));
        }
        // unbounded geoms (planes): the AABB's lowest corner along the plane normal
        for (int pl = 0; pl < ngeom; pl++)
          if (T.g_info[pl][1] == MIR_GEOM_PLANE && (allow >> pl & 1u)) {
            const V3 n = mcol(q2m(ld4v(S.col.gquat[pl])), 2), pp = ld3v(S.col.gpos[pl]);
            const V3 low = v3(n.x >= 0.0f ? lg.x : hg.x, n.y >= 0.0f ? lg.y : hg.y, n.z >= 0.0f ? lg.z : hg.z);
            if (dot(low - pp, n) < 0.0f) atomicOr(&S.col.sap.hitrow[g < pl ? g : pl], 1u << (g < pl ? pl : g));
          }
      }
      WSYNC();
      // (d) the overlapping pairs in the order of the static list (lower geom index, then higher), plane first in a pair
      int off = 0;
      for (int r0 = 0; r0 < ngeom; r0 += G) {
        const int r = r0 + lane;
        unsigned bits = r < ngeom ? S.col.sap.hitrow[r] : 0u;
        float inc = (float)__popc(bits);
        const float cnt = inc;
        inc += row_shr<1>(inc);
        inc += row_shr<2>(inc);
        inc += row_shr<4>(inc);
        inc += row_shr<8>(inc);
        int k = off + (int)(inc - cnt);
        off += (int)row_bcast<15>(inc);
        while (bits) {
          const int b = __ffs(bits) - 1;
          bits &= bits - 1u;
          if (k < K16_MAX_PAIR) S.col.sap.plist[k] = T.g_info[b][1] == MIR_GEOM_PLANE ? (b | r << 8) : (r | b << 8);
          k++;
        }
      }
      npl = off < K16_MAX_PAIR ? off : K16_MAX_PAIR;
      pair_ovf = off > K16_MAX_PAIR;  // (more overlapping AABB pairs than the list holds: as for the candidates below)
      WSYNC();
    }
    // broadphase: bounding test per candidate pair (static list, or the sweep's survivors), ordered compaction
    int base = 0;
    for (int p0 = 0; p0 < npl; p0 += G) {
      int p = p0 + lane;
      bool hit = false;
      int pr = 0, ptypes = 0;
      if (p < npl) {
        pr = SAP ? S.col.sap.plist[p] : T.pair[p];
        const int g1 = pr & 255, g2 = pr >> 8;
        V3 h2 = ld3v(T.g_size[g2]);
        M3 R2 = q2m(ld4v(S.col.gquat[g2]));
        V3 c2 = ld3v(S.col.gpos[g2]);
        const int t1 = T.g_info[g1][1], t2 = T.g_info[g2][1];
        ptypes = t1 | t2 << 8;
        if (t1 == MIR_GEOM_PLANE) {
          V3 n = mcol(q2m(ld4v(S.col.gquat[g1])), 2);
          float ext = h2.x * fabsf(dot(n, mcol(R2, 0))) + h2.y * fabsf(dot(n, mcol(R2, 1))) + h2.z * fabsf(dot(n, mcol(R2, 2)));
          if (CONVEX && t2 == MIR_GEOM_SPHERE) ext = h2.x;
          if (CONVEX && t2 == MIR_GEOM_CAPSULE) ext = h2.y * fabsf(dot(n, mcol(R2, 2))) + h2.x;
          if (CONVEX && t2 == MIR_GEOM_HULL) ext = T.g_size[g2][3];  // (bounding sphere)
          hit = dot(c2 - ld3v(S.col.gpos[g1]), n) - ext < 0.0f;
        } else {
          V3 h1 = ld3v(T.g_size[g1]);
          // bounding spheres (box: half diagonal; sphere: radius; capsule: half length + radius -- T.g_size[.][3])
          float rs = CONVEX ? T.g_size[g1][3] + T.g_size[g2][3] : sqrtf(dot(h1, h1)) + sqrtf(dot(h2, h2));
          V3 dc = c2 - ld3v(S.col.gpos[g1]);
          hit = dot(dc, dc) <= rs * rs;
          if (hit && (!CONVEX || (t1 == MIR_GEOM_BOX && t2 == MIR_GEOM_BOX))) {
            // the six face axes of the narrowphase's separating-axis test (same expressions): a pair they separate would
            // come back with zero contacts, and the narrowphase takes the candidates of an env one after the other
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
      unsigned long long bal = __ballot(hit);
      uint32_t gm = (uint32_t)(bal >> (grp * G)) & 0xffffu;
      int pos = base + __popc(gm & ((1u << lane) - 1u));
      if (hit && pos < G) S.col.cand[pos] = pr | ptypes << 16;  // (the pair itself with its geom types: g1 | g2 << 8 | t1 << 16 | t2 << 24)
      base += __popc(gm);
    }
    const int ncand = base < G ? base : G;
    pair_ovf = pair_ovf || base > G;
    if (lane == 0) S.ncand = ncand;
    WSYNC();
    STAMP(12);
    HSTAMP(59);
    // narrowphase, plane-box: one candidate at a time, its 8 box corners on lanes 0..7 of the group
    // (wave-uniform loop: the DPP/ballot selection below needs all lanes present)
    // (every env takes ITS next candidate of the kind in a trip -- the lists of the four envs of a wave hold them at different
    // positions -- so a wave makes as many trips as its busiest env has such candidates, not one per position of their union)
    const int prl = lane < ncand ? S.col.cand[lane] : 0;
    const bool planel = lane < ncand && (prl >> 16 & 255) == MIR_GEOM_PLANE && (!CONVEX || (prl >> 24) == MIR_GEOM_BOX);
    uint32_t planem = (uint32_t)(__ballot(planel) >> (grp * G)) & 0xffffu, boxm = box_candidates(ncand);
    while (__any(planem != 0u)) {
      const bool isplane = planem != 0u;
      const int k = isplane ? __ffs(planem) - 1 : 0;
      planem &= planem - 1u;
      const int pr = isplane ? S.col.cand[k] : 0;
      const int g1 = pr & 255, g2 = pr >> 8 & 255;
      const M3 Rp = q2m(ld4v(S.col.gquat[g1]));
      const V3 n = mcol(Rp, 2), eu = mcol(Rp, 0), ev = mcol(Rp, 1);
      const M3 R2 = q2m(ld4v(S.col.gquat[g2]));
      const V3 h = ld3v(T.g_size[g2]);
      const int c = lane & 7;
      const V3 w = ld3v(S.col.gpos[g2]) + ((c & 1) ? h.x : -h.x) * mcol(R2, 0) + ((c & 2) ? h.y : -h.y) * mcol(R2, 1) +
                   ((c & 4) ? h.z : -h.z) * mcol(R2, 2);
      const V3 rel = w - ld3v(S.col.gpos[g1]);
      const float d = dot(rel, n), u = dot(rel, eu), v = dot(rel, ev);
      const bool pen = isplane && lane < 8 && d < 0.0f;
      const uint32_t penm = (uint32_t)(__ballot(pen) >> (grp * G)) & 0xffu;
      const int cnt = __popc(penm);
      // support extremes (+u, -u, +v, -v; lowest corner index wins ties), needed only when more than 4 corners penetrate
      // somewhere in the wave (a box lying flat has exactly 4: the reductions are skipped)
      uint32_t ext = 0u;
      if (__any(cnt > 4)) {
        const float uM = gmaxf(pen ? u : -3e38f), um = -gmaxf(pen ? -u : -3e38f);
        const float vM = gmaxf(pen ? v : -3e38f), vm = -gmaxf(pen ? -v : -3e38f);
        const uint32_t e0 = (uint32_t)(__ballot(pen && u == uM) >> (grp * G)) & 0xffu, e1 = (uint32_t)(__ballot(pen && u == um) >> (grp * G)) & 0xffu;
        const uint32_t e2 = (uint32_t)(__ballot(pen && v == vM) >> (grp * G)) & 0xffu, e3 = (uint32_t)(__ballot(pen && v == vm) >> (grp * G)) & 0xffu;
        ext = (e0 & -e0) | (e1 & -e1) | (e2 & -e2) | (e3 & -e3);
      }
      const uint32_t keepm = cnt <= 4 ? penm : ext;
      const bool keep = (keepm >> lane & 1u) && lane < 8;
      const int slot = __popc(keepm & ((1u << lane) - 1u));
      if (keep && slot < 4) {
        const V3 pos = w - (0.5f * d) * n;
        stv(S.col.stage[k][slot], f4{pos.x, pos.y, pos.z, d});
      }
      if (lane == k && isplane) {
        mycount = min(__popc(keepm), 4);
        st3v(S.col.snorm[k], n);
      }
    }
    // narrowphase, box-box: one candidate at a time on the whole row (box_box_row, mir_dev.h): the 15 separating axes
    // on lanes 0..14, the incident-face vertices on lanes 0..3
    HSTAMP(60);
    if constexpr (BIGV) {
      // (three contacts per lane: every other trip of an env is the main wave's -- box_share, below; its counts come back through LDS)
      if (lane == 0) __hip_atomic_store(&S.bp_ready, pass_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      uint32_t mineA, otherB;
      split_alternate(boxm, mineA, otherB);
      box_trips(mineA, S.col.clip, mycount, false);
      while (__hip_atomic_load(&S.bb_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < pass_no) __builtin_amdgcn_s_sleep(1);
      if ((otherB >> lane) & 1u) mycount = S.col.count[lane];
    } else {
      box_trips(boxm, S.col.clip, mycount, false);
    }
    HSTAMP(61);
    if constexpr (CONVEX) {
      // narrowphase of the round shapes, LANE-PRIVATE: lane c takes candidate c.  Plane - sphere / capsule in closed form
      // (one / two points at half depth), every other pair that is not box - box through GJK on the cores, MPR when the
      // cores overlap (mir_convex.h).  Lanes diverge here and reconverge at the end of the block.
      if (lane < ncand) {
        const int pr = S.col.cand[lane];
        const int g1 = pr & 255, g2 = pr >> 8 & 255;
        const int t1 = pr >> 16 & 255, t2 = pr >> 24;
        if (t1 == MIR_GEOM_PLANE && (t2 == MIR_GEOM_SPHERE || t2 == MIR_GEOM_CAPSULE)) {
          const V3 n = mcol(q2m(ld4v(S.col.gquat[g1])), 2), pp = ld3v(S.col.gpos[g1]), pc = ld3v(S.col.gpos[g2]);
          const V3 sz = ld3v(T.g_size[g2]);
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
          mycount = cnt;
        } else if (t1 == MIR_GEOM_PLANE && t2 == MIR_GEOM_HULL) {
          // plane - hull: the penetrating vertices in index order, reduced to four like plane - box (support extremes, first
          // index wins ties; oracle: plane_hull / reduce4).  Two passes over the vertices, lane-private.
          const M3 Rp = q2m(ld4v(S.col.gquat[g1])), R2 = q2m(ld4v(S.col.gquat[g2]));
          const V3 n = mcol(Rp, 2), eu = mcol(Rp, 0), ev = mcol(Rp, 1), pp = ld3v(S.col.gpos[g1]), pc = ld3v(S.col.gpos[g2]);
          const int v0 = (int)T.g_size[g2][0], nvg = (int)T.g_size[g2][1];
          int npen = 0, pk0 = -1, pk1 = -1, pk2 = -1, pk3 = -1;
          float uM = 0.0f, um = 0.0f, vM = 0.0f, vm = 0.0f;
          for (int i = 0; i < nvg; i++) {
            const V3 l = ld3v(s_hull[v0 + i]);
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
            const V3 l = ld3v(s_hull[v0 + i]);
            const V3 w = pc + l.x * mcol(R2, 0) + l.y * mcol(R2, 1) + l.z * mcol(R2, 2);
            const float d = dot(w - pp, n);
            if (d < 0.0f && (npen <= 4 || i == pk0 || i == pk1 || i == pk2 || i == pk3)) {
              const V3 c = w - (0.5f * d) * n;
              stv(S.col.stage[lane][cnt], f4{c.x, c.y, c.z, d});
              cnt++;
            }
          }
          if (cnt) st3v(S.col.snorm[lane], n);
          mycount = cnt;
        } else if (t1 != MIR_GEOM_PLANE && !(t1 == MIR_GEOM_BOX && t2 == MIR_GEOM_BOX)) {
          const M3 R1 = q2m(ld4v(S.col.gquat[g1])), R2 = q2m(ld4v(S.col.gquat[g2]));
          const V3 z1 = ld3v(T.g_size[g1]), z2 = ld3v(T.g_size[g2]);
          const ShapeD A = {t1, z1, ld3v(S.col.gpos[g1]), mcol(R1, 0), mcol(R1, 1), mcol(R1, 2), &s_hull[t1 == MIR_GEOM_HULL ? (int)z1.x : 0][0], (int)z1.y};
          const ShapeD B = {t2, z2, ld3v(S.col.gpos[g2]), mcol(R2, 0), mcol(R2, 1), mcol(R2, 2), &s_hull[t2 == MIR_GEOM_HULL ? (int)z2.x : 0][0], (int)z2.y};
          f4 pt;
          V3 n;
          if (convex_pair(A, B, pt, n)) {
            stv(S.col.stage[lane][0], pt);
            st3v(S.col.snorm[lane], n);
            mycount = 1;
          }
        }
      }
    }
  }
    return mycount;
  };
  // ---- contact arrays and base Jacobians from the staging area (`mycount` = this lane's point count, lane = candidate): ordered
  // compaction, per-contact frames / impedance / references, then the contact Jacobian columns (lane = dof).  Needs the staging
  // area, the link poses, the motion subspaces and the model table; writes `con` (over the dead dynamics scratch) and `Jb`.
  // ======================= constraint rows ======================================================
  // contact base Jacobians: lane = dof; Jb[c][r*16 + i], r = normal, t1, t2
  // (two contacts per trip, every read of both issued in one batch ahead of the arithmetic; a dof that moves neither
  // body ends with sgn = 0, so there is no divergent branch around the reads)
  auto jac_build = [&](int first, int stride) {
    const int ncon = S.ncon;
    const V3 cd_ang = ld3v(&S.cdof[lane][0]), cd_lin = ld3v(&S.cdof[lane][4]);
    for (int c0 = first; c0 < ncon; c0 += stride) {
      const int cA = c0, cB = c0 + 1 < ncon ? c0 + 1 : c0;
      const f4 mkA = ldv(reinterpret_cast<const float*>(S.con.cmask[cA])), mkB = ldv(reinterpret_cast<const float*>(S.con.cmask[cB]));
      const f4 cpA = ldv(S.con.cpos[cA]), r1A = ldv(&S.con.cref[cA][0]), r2A = ldv(&S.con.cref[cA][4]);
      const f4 cpB = ldv(S.con.cpos[cB]), r1B = ldv(&S.con.cref[cB][0]), r2B = ldv(&S.con.cref[cB][4]);
      const f4 fnA = ldv(&S.con.cfrm[cA][0]), f1A = ldv(&S.con.cfrm[cA][4]), f2A = ldv(&S.con.cfrm[cA][8]);
      const f4 fnB = ldv(&S.con.cfrm[cB][0]), f1B = ldv(&S.con.cfrm[cB][4]), f2B = ldv(&S.con.cfrm[cB][8]);
      __builtin_amdgcn_sched_barrier(0);
#define MIR_JCOL(mk, cp, r1, r2, fn, f1, f2, cc)                                                              \
      {                                                                                                     \
        const uint32_t dm1 = __float_as_uint(mk.x), dm2 = __float_as_uint(mk.y);                            \
        const bool in2 = dm2 >> lane & 1u, in1 = dm1 >> lane & 1u;                                          \
        const float sgn = (in2 ? 1.0f : 0.0f) - (in1 ? 1.0f : 0.0f); /* a dof moving both bodies cancels */ \
        const V3 r = v3(cp.x, cp.y, cp.z) - (in2 ? v3(r2.x, r2.y, r2.z) : v3(r1.x, r1.y, r1.z));            \
        const V3 vel = cross(cd_ang, r) + cd_lin;                                                           \
        float* jb = JBROW(S, cc);                                                                           \
        /* (selects, not products: a lane that carries no dof holds stale LDS in cd_ang / cd_lin, and 0 x NaN is NaN) */ \
        jb[lane] = sgn != 0.0f ? sgn * dot(vel, v3(fn.x, fn.y, fn.z)) : 0.0f;                               \
        jb[16 + lane] = sgn != 0.0f ? sgn * dot(vel, v3(f1.x, f1.y, f1.z)) : 0.0f;                          \
        jb[32 + lane] = sgn != 0.0f ? sgn * dot(vel, v3(f2.x, f2.y, f2.z)) : 0.0f;                          \
      }
      MIR_JCOL(mkA, cpA, r1A, r2A, fnA, f1A, f2A, cA)
      if (c0 + 1 < ncon) MIR_JCOL(mkB, cpB, r1B, r2B, fnB, f1B, f2B, cB)
#undef MIR_JCOL
    }
  };
  // (shared_jac: the main wave takes every other pair of contacts of the Jacobian build -- barrier (2b) -- where the action-independent
  //  half is all the launch has left to do.  Returns the candidate points before the capacity was applied, 255 = saturated.)
  auto contacts_build = [&](int mycount, bool shared_jac) -> int {
  STAMP(13);
  // ordered compaction of the contact points: exclusive prefix over candidate lanes (convergent code)
  const int maxc = max_contacts < MAXCON ? max_contacts : MAXCON;
  int ptotal = 0;
  {
    // ---- more candidate points than the capacity: the largest manifolds are thinned before any pair loses all of its points
    // (the rule is defined at oracle/orc_rigid.c: thin_manifolds -- per round, the pairs holding the most points merge their last
    // two points into the mean, in pair order, until the total fits).  Rare path: both fingers AND the cube on the floor while
    // the pads hold the cube (20+ points), which is exactly where the reference's expert puts the hand
    // (examples/franka/pick_cube_state.py:38-39 of the reference: hand target 3 cm above the cube centre).
    float totf = (float)mycount;
    totf += row_shr<1>(totf);
    totf += row_shr<2>(totf);
    totf += row_shr<4>(totf);
    totf += row_shr<8>(totf);
    int total0 = (int)row_bcast<15>(totf);
    ptotal = pair_ovf ? 255 : total0;  // (candidate points before the capacity is applied: more than max_contacts says the manifolds were thinned)
    if (__any(total0 > maxc)) {
      WSYNC();  // (the staging area was written by other lanes of the row)
      for (int round = 0; round < 8; round++) {  // (a manifold holds at most 8 points)
        const int mx = (int)gmaxf((float)mycount);
        const bool live = total0 > maxc && mx > 1;
        if (!__any(live)) break;
        const bool is = live && mycount == mx;
        float r = is ? 1.0f : 0.0f;
        const float self = r;
        r += row_shr<1>(r);
        r += row_shr<2>(r);
        r += row_shr<4>(r);
        r += row_shr<8>(r);
        const int rank = (int)(r - self), nis = (int)row_bcast<15>(r), need = total0 - maxc;
        if (is && rank < need) {
          const f4 pa = ldv(S.col.stage[lane][mx - 2]), pb = ldv(S.col.stage[lane][mx - 1]);
          stv(S.col.stage[lane][mx - 2], f4{0.5f * (pa.x + pb.x), 0.5f * (pa.y + pb.y), 0.5f * (pa.z + pb.z), 0.5f * (pa.w + pb.w)});
          mycount--;
        }
        if (live) total0 -= need < nis ? need : nis;
      }
      WSYNC();
    }
  }
  {
    // inclusive prefix sum over the row by DPP shifts (zeros shifted in), total from lane 15
    float inclf = (float)mycount;
    inclf += row_shr<1>(inclf);
    inclf += row_shr<2>(inclf);
    inclf += row_shr<4>(inclf);
    inclf += row_shr<8>(inclf);
    const int incl = (int)inclf;
    const int off = incl - mycount;
    const int total = (int)row_bcast<15>(inclf);
    if (lane == 0) S.ncon = total < maxc ? total : maxc;
    const int ncon_new = total < maxc ? total : maxc;
    STAMP(22);
    // candidate lanes publish which (candidate, point) fills each contact slot ...
    for (int c = 0; c < mycount; c++)
      if (off + c < maxc) S.col.cmap[off + c] = lane * 8 + c;
    WSYNC();
    STAMP(23);
    // ... and every contact is then finished by its own lane, in parallel (staging lives in col
    // scratch, which does not overlap the contact arrays)
    // (computed first, stored second: in the DUAL instantiation the contact arrays overlay the dynamics scratch of the main
    // wave, so the collision wave does the arithmetic while that wave is still busy and stores after the barrier)
    // (CPL > 1: lane c finishes contacts c, c + 16, c + 32 one after the other, all of them into registers before the first store:
    //  the contact arrays of that instantiation reach into the staging area)
    bool mine[CPL];
    f4 pd[CPL], meta[CPL];
    V3 n[CPL], t1[CPL], t2[CPL], ref1[CPL], ref2[CPL];
    uint32_t dm1[CPL], dm2[CPL], chunks[CPL];
#pragma unroll
    for (int sl = 0; sl < CPL; sl++) {
      const int k = lane + G * sl;
      mine[sl] = k < ncon_new;
      pd[sl] = f4{0, 0, 0, 0}; meta[sl] = f4{0, 0, 0, 0};
      n[sl] = v3(0, 0, 0); t1[sl] = n[sl]; t2[sl] = n[sl]; ref1[sl] = n[sl]; ref2[sl] = n[sl];
      dm1[sl] = 0u; dm2[sl] = 0u; chunks[sl] = 0u;
      if (mine[sl]) {
        const int mp = S.col.cmap[k];
        const int cl = mp >> 3, ci = mp & 7;
        const int pr = S.col.cand[cl];
        const int g1 = pr & 255, g2 = pr >> 8 & 255;
        n[sl] = ld3v(S.col.snorm[cl]);
        V3 ta = fabsf(n[sl].y) < 0.5f ? v3(0, 1, 0) : v3(0, 0, 1);  // same frame construction as the oracle
        ta = ta - dot(n[sl], ta) * n[sl];
        ta = __builtin_amdgcn_rsqf(dot(ta, ta)) * ta;
        t1[sl] = ta;
        t2[sl] = cross(n[sl], ta);
        const float mu = fmaxf(T.g_pos[g1][3], T.g_pos[g2][3]);
        const f4 s1a = ldv(&T.g_sol[g1][0]), s1b = ldv(&T.g_sol[g1][4]), s2a = ldv(&T.g_sol[g2][0]), s2b = ldv(&T.g_sol[g2][4]);
        const float sr0 = 0.5f * (s1a.x + s2a.x), sr1 = 0.5f * (s1a.y + s2a.y);
        const float si[5] = {0.5f * (s1a.z + s2a.z), 0.5f * (s1a.w + s2a.w), 0.5f * (s1b.x + s2b.x), 0.5f * (s1b.y + s2b.y), 0.5f * (s1b.z + s2b.z)};
        const int b1 = T.g_info[g1][0], b2 = T.g_info[g2][0];
        const float wsum = T.b_invw[b1] + T.b_invw[b2];
        const float dmax = fminf(fmaxf(si[1], 1e-4f), 0.9999f);
        const float tc = fmaxf(sr0, 2.0f * dt);
        const float kk = 1.0f / (dmax * dmax * tc * tc * sr1 * sr1), bb = 2.0f / (dmax * tc);
        dm1[sl] = (uint32_t)T.b_info[b1][0]; dm2[sl] = (uint32_t)T.b_info[b2][0];
        const uint32_t inv = dm1[sl] | dm2[sl];
        chunks[sl] = ((inv & 0xfu) ? 1u : 0u) | ((inv & 0xf0u) ? 2u : 0u) | ((inv & 0xf00u) ? 4u : 0u) | ((inv & 0xf000u) ? 8u : 0u);
        ref1[sl] = ld3v(S.xpos[T.b_info[b1][1]]); ref2[sl] = ld3v(S.xpos[T.b_info[b2][1]]);
        pd[sl] = ldv(S.col.stage[cl][ci]);
        const float dist = pd[sl].w;
        const float imp = impedance(si[0], si[1], si[2], si[3], si[4], dist);
        const float Rr = fmaxf(2.0f * mu * mu * (1.0f - imp) / imp * wsum * (1.0f + mu * mu), 1e-15f);
        meta[sl] = f4{mu, 1.0f / Rr, -kk * imp * dist, bb};
      }
    }
    {  // does any contact of the env move dofs of both trees?  (bit 0; bits 1 .. 16: contact c moves dofs of the second tree only --
       // what the tree-wise parts of the solver need to know about a contact when the problem separates; bits 20 .. 27: candidate
       // points before the capacity was applied, for the diagnostics record -- the word travels with the scratch row of a split step)
       // (CPL > 1: the second-tree flags of contacts 16 s + c also in S.conB[s])
      const uint32_t low = (1u << mdl_split) - 1u;
      bool cpl_any = false;
      uint32_t tbw[CPL];
#pragma unroll
      for (int sl = 0; sl < CPL; sl++) {
        const uint32_t both = dm1[sl] | dm2[sl];
        cpl_any = cpl_any || (mine[sl] && (both & low) != 0u && (both & ~low) != 0u);
        tbw[sl] = (uint32_t)(__ballot(mine[sl] && (both & low) == 0u) >> (grp * G)) & 0xffffu;
      }
      const unsigned long long cb = __ballot(cpl_any);
      if (lane == 0) {
        S.coupled = (int)(((uint32_t)(cb >> (grp * G)) & 0xffffu ? 1u : 0u) | tbw[0] << 1 | (uint32_t)(ptotal < 255 ? ptotal : 255) << 20);
        if constexpr (CPL > 1) {
#pragma unroll
          for (int sl = 0; sl < CPL; sl++) S.conB[sl] = (int)tbw[sl];
        }
      }
    }
    HSTAMP(43);
    if (DUAL) __syncthreads();  // (2) the main wave has left the dynamics scratch: the contact arrays may be stored over it
    HSTAMP(44);
#pragma unroll
    for (int sl = 0; sl < CPL; sl++) {
      const int k = lane + G * sl;
      if (mine[sl]) {
        stv(S.con.cpos[k], pd[sl]);
        st3v(&S.con.cfrm[k][0], n[sl]); st3v(&S.con.cfrm[k][4], t1[sl]); st3v(&S.con.cfrm[k][8], t2[sl]);
        stv(S.con.cmeta[k], meta[sl]);
        st3v(&S.con.cref[k][0], ref1[sl]); st3v(&S.con.cref[k][4], ref2[sl]);
        S.con.cmask[k][0] = dm1[sl]; S.con.cmask[k][1] = dm2[sl]; S.con.cmask[k][2] = chunks[sl]; S.con.cmask[k][3] = 0u;
      }
    }
  }
  WSYNC();  // col scratch is dead from here on (Jb may overwrite it)
  STAMP(5);
  // (where this half is all the launch has left to do -- the pre half of a split step -- the main wave takes every other pair
  //  of contacts of the Jacobian build: barrier (2b) hands it the contact arrays stored just above)
  if (shared_jac) { __syncthreads(); jac_build(0, 4); }
  else jac_build(0, 2);
  return ptotal;
  };
  // J^T D J with EVERY pyramid row of every contact active (lane = dof row, 16 columns), summed from zero in contact order.  The
  // Newton loop starts its incremental Hessian from Mt + this: resting and gripping contacts have all four rows active, and the
  // rows that are not come off in the first incremental update.  In the DUAL instantiation the collision wave accumulates it
  // while the main wave evaluates the constraint rows, the warm start and the first gradient (hand-over through S.M, which is
  // dead once the mass-matrix rows are in registers).
  auto hess_full = [&](float (&hp)[G], int ncon) {
#pragma unroll
    for (int j = 0; j < G; j++) hp[j] = 0.0f;
    for (int c = 0; c < ncon; c++) {
      const float* jb = JBROW(S, c);
      const float jn = jb[lane], j1 = jb[16 + lane], j2 = jb[32 + lane];
      const f4 mt = ldv(S.con.cmeta[c]);
      f4 xn[4], x1[4], x2[4];
#pragma unroll
      for (int q = 0; q < 4; q++) { xn[q] = ldv(jb + 4 * q); x1[q] = ldv(jb + 16 + 4 * q); x2[q] = ldv(jb + 32 + 4 * q); }
      __builtin_amdgcn_sched_barrier(0);
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
  // one contact's signed contribution to the Hessian row of this lane (lane = dof): the change of its pyramid's 3 x 3 weight between the
  // row flags `old` and `bits` (the expressions of the incremental update in the Newton loop, which the helper wave runs for the slots
  // above the first)
  auto hess_flip = [&](float (&hk)[G], int c, bool first_it) __attribute__((always_inline)) {
    const float* jb = JBROW(S, c);
    const f4 fb = ldv(CFB(S, c));
    const float jn = jb[lane], j1 = jb[16 + lane], j2 = jb[32 + lane];
    const f4 mt = ldv(S.con.cmeta[c]);
    f4 xn[4], x1[4], x2[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { xn[q] = ldv(jb + 4 * q); x1[q] = ldv(jb + 16 + 4 * q); x2[q] = ldv(jb + 32 + 4 * q); }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned both = (unsigned)fb.w;
    const unsigned bits = both & 15u, old = first_it ? 15u : both >> 4;
    const float mu = mt.x, D = mt.y;
    const float a0 = D * (float)((int)(bits & 1u) - (int)(old & 1u)), a1 = D * (float)((int)(bits >> 1 & 1u) - (int)(old >> 1 & 1u));
    const float a2 = D * (float)((int)(bits >> 2 & 1u) - (int)(old >> 2 & 1u)), a3 = D * (float)((int)(bits >> 3 & 1u) - (int)(old >> 3 & 1u));
    const float w0 = a0 + a1 + a2 + a3, w1 = mu * (a0 - a1), w2 = mu * (a2 - a3), w3 = mu * mu * (a0 + a1), w4 = mu * mu * (a2 + a3);
    const float tn = jn * w0 + j1 * w1 + j2 * w2, t1 = jn * w1 + j1 * w3, t2 = jn * w2 + j2 * w4;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      hk[4 * q + 0] += tn * xn[q].x + t1 * x1[q].x + t2 * x2[q].x;
      hk[4 * q + 1] += tn * xn[q].y + t1 * x1[q].y + t2 * x2[q].y;
      hk[4 * q + 2] += tn * xn[q].z + t1 * x1[q].z + t2 * x2[q].z;
      hk[4 * q + 3] += tn * xn[q].w + t1 * x1[q].w + t2 * x2[q].w;
    }
  };
  // (three contacts per lane) where the helper wave leaves its shares: the Hessian rows in the 1 KB of xpos | xquat | cdof (dead between
  // the Jacobian build and the closing FK), the gradient entry in the spare floats of this lane's M row
  auto help_hrow = [&]() -> float* { return &S.xpos[0][0] + 16 * lane; };
  static_assert(sizeof(((EnvLds*)nullptr)->xpos) + sizeof(((EnvLds*)nullptr)->xquat) + sizeof(((EnvLds*)nullptr)->cdof) == G * G * sizeof(float), "xpos | xquat | cdof (declared in this order): 256 floats");
  // The helper wave's side of the Newton loop (called behind barrier (4)): for request k = iteration + 1 -- from the second iteration on
  // the gradient share first (contacts 16 .. ncon in contact order), then the Hessian share (the flipped contacts among them).  An env
  // with at most 16 contacts gets exact zeros from here: it is computed as by the one-contact-per-lane kernel.
  auto helper_newton = [&]() {
    if constexpr (CPL > 1) {
      if (!__any(S.ncon > G)) return;  // (no env of the workgroup has a contact above the first slot: the main wave does not ask -- the same test there)
      for (int k = 1;; k++) {
        int r;
        while ((r = __hip_atomic_load(&s_help[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) >= 0 && r < k) __builtin_amdgcn_s_sleep(1);
        if (r < 0) break;
        const int ncon = S.ncon;
        if (k > 1) {
          float gp = 0.0f;
          for (int c0 = G; c0 < ncon; c0 += 4) {
            float jn[4], j1[4], j2[4];
            f4 fb[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int c = c0 + u < ncon ? c0 + u : c0;
              const float* jb = JBROW(S, c);
              jn[u] = jb[lane]; j1[u] = jb[16 + lane]; j2[u] = jb[32 + lane];
              fb[u] = ldv(CFB(S, c));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; u++)
              if (c0 + u < ncon) gp -= jn[u] * fb[u].x + j1[u] * fb[u].y + j2[u] * fb[u].z;
          }
          S.M[lane][G] = gp;
          WSYNC();
          if (tid == 0) __hip_atomic_store(&s_help[1], k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        float hd[G];
#pragma unroll
        for (int j = 0; j < G; j++) hd[j] = 0.0f;
#pragma unroll
        for (int sl = 1; sl < CPL; sl++) {
          bool flipped = false;
          if (lane + G * sl < ncon) {
            const unsigned both = (unsigned)CFB(S, lane + G * sl)[3];
            flipped = (both & 15u) != (k == 1 ? 15u : both >> 4);
          }
          for (unsigned fm = (unsigned)(__ballot(flipped) >> (tid & 48)) & 0xffffu; fm; fm &= fm - 1u) hess_flip(hd, __ffs(fm) - 1 + G * sl, k == 1);
        }
        float* hr = help_hrow();
#pragma unroll
        for (int q = 0; q < 4; q++) stv(hr + 4 * q, f4{hd[4 * q], hd[4 * q + 1], hd[4 * q + 2], hd[4 * q + 3]});
        WSYNC();
        if (tid == 0) __hip_atomic_store(&s_help[2], k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  };
  // body inertia about the tree reference point, in the world frame (lane = body): needs the link poses only.  In the two-wave
  // instantiations the collision wave computes it in front of its detection (it has ~1.8 k cycles of slack before the second
  // barrier, the main wave none) and raises S.cin_ready; the main wave picks it up where the composite inertias start.
  auto cinert_store = [&](bool isb, const float (&ibv)[6], V3 ipos, float mass, int root) {
    float* c = S.dyn.cinert[lane];
    if (isb) {
      M3 R = q2m(ld4v(S.xquat[lane]));
      float Ib[3][3] = {{ibv[0], ibv[3], ibv[4]}, {ibv[3], ibv[1], ibv[5]}, {ibv[4], ibv[5], ibv[2]}};
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
      V3 r = ld3v(S.xpos[lane]) + mmul(R, ipos) - ld3v(S.xpos[root]);
      float rr = dot(r, r);
      stv(c, f4{mass, mass * r.x, mass * r.y, mass * r.z});
      stv(c + 4, f4{W[0][0] + mass * (rr - r.x * r.x), W[1][1] + mass * (rr - r.y * r.y), W[2][2] + mass * (rr - r.z * r.z),
                    W[0][1] - mass * r.x * r.y});
      stv(c + 8, f4{W[0][2] - mass * r.x * r.z, W[1][2] - mass * r.y * r.z, 0.0f, 0.0f});
    } else {
      stv(c, f4{0, 0, 0, 0}); stv(c + 4, f4{0, 0, 0, 0}); stv(c + 8, f4{0, 0, 0, 0});
    }
  };
  if (DUAL && wave == 1) {
    // (lane constants of the body inertia: quads 0, 4, 5, 6 of LaneK16)
    auto helper_cinert = [&](int pass_no) {
      const f4 h0 = *reinterpret_cast<const f4*>(m->lanek_t[0][lane]), h4 = *reinterpret_cast<const f4*>(m->lanek_t[4][lane]);
      const f4 h5 = *reinterpret_cast<const f4*>(m->lanek_t[5][lane]), h6 = *reinterpret_cast<const f4*>(m->lanek_t[6][lane]);
      const float hib[6] = {h5.x, h5.y, h5.z, h5.w, h6.x, h6.y};
      cinert_store(lane < nb && lane > 0, hib, v3(h4.x, h4.y, h4.z), h4.w, __float_as_int(h0.z));
      WSYNC();
      if (lane == 0) __hip_atomic_store(&S.cin_ready, pass_no, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // the four quads of lane constants the forward kinematics needs (fetched again for the closing FK: nothing is kept live
    // through the collision phase)
    auto fk_consts = [&](BodyK& hk) {
      const f4 h0 = *reinterpret_cast<const f4*>(m->lanek_t[0][lane]), h1 = *reinterpret_cast<const f4*>(m->lanek_t[1][lane]);
      const f4 h2 = *reinterpret_cast<const f4*>(m->lanek_t[2][lane]), h3 = *reinterpret_cast<const f4*>(m->lanek_t[3][lane]);
      hk.jtype = __float_as_int(h0.x); hk.qadr = __float_as_int(h0.y);
      hk.pos = v3(h1.x, h1.y, h1.z);
      hk.quat = Q4{h2.x, h2.y, h2.z, h2.w};
      hk.axis = v3(h3.x, h3.y, h3.z);
    };
    const uint64_t hparents = m->parents;
    bool ovf_h = false;  // (rotated launch, exact contacts: this env is deferred -- no scratch row of the next step is stored for it)
    if (ROT) {
      // rotated launch.  First, as in the fused launch, this wave hands the main wave the contact rows of the step it is solving
      // (here they come from the scratch row the previous launch left) and the all-rows-active Hessian accumulated from them
      {
        const float* pre = a.pre + (size_t)env * K16_PRE_STRIDE;
        const f4 head = *reinterpret_cast<const f4*>(pre + K16_PRE_HEAD);
        int nc = __float_as_int(head.x);
        int cplw = __float_as_int(head.y);
        const int pts_h = (cplw >> 20) & 255;
        // (three contacts per lane: an env above 16 points has its row in pre_big -- if the launch before wrote one: the tail of a
        //  one-contact-per-lane launch cannot, and an env beyond 48 points has none either; such an env is deferred -- its count
        //  saturated in the word the main wave reads -- and takes the fused pass of the list instantiation behind this launch)
        bool big_h = false;
        if constexpr (CPL > 1) {
          big_h = pts_h > K16_MAX_CONTACT && pts_h <= MAXCON && __float_as_int(head.z) == K48_MAGIC && a.pre_big != nullptr;
          // (a row in `pre` is complete when it holds as many contacts as the narrowphase found points: the tail of a one-contact-per-lane
          //  launch thins beyond ITS capacity -- the scene's, possibly below 16 -- and the env is then deferred here too)
          if (!big_h && nc != pts_h) { cplw |= 255 << 20; nc = 0; }
        }
        ovf_h = ((cplw >> 20) & 255) > defer_above;
        if constexpr (CPL > 1) {
          const float* big = a.pre_big + (size_t)env * K48_STRIDE;
          f4 h2 = {0, 0, 0, 0};
          if (big_h) { h2 = *reinterpret_cast<const f4*>(big + K48_HEAD); nc = __float_as_int(h2.x); }
          if (lane == 0) { S.ncon = nc; S.coupled = cplw; S.ncand = 0; S.conB[0] = (cplw >> 1) & 0xffff; S.conB[1] = __float_as_int(h2.y); S.conB[2] = __float_as_int(h2.z); }
          if (big_h) {
#pragma unroll
            for (int sl = 0; sl < CPL; sl++)
              if (lane + G * sl < nc) stv(S.con.cmeta[lane + G * sl], *reinterpret_cast<const f4*>(big + K48_CMETA + 4 * (lane + G * sl)));
            bigrows_load(big, nc, S);
          } else {
            if (lane < nc) stv(S.con.cmeta[lane], *reinterpret_cast<const f4*>(pre + K16_PRE_CMETA + 4 * lane));
            jrows_load(pre, nc, (uint64_t)__float_as_uint(head.z) | ((uint64_t)__float_as_uint(head.w) << 32), S);
          }
        } else {
          if (lane == 0) { S.ncon = nc; S.coupled = cplw; S.ncand = 0; }
          if (lane < nc) stv(S.con.cmeta[lane], *reinterpret_cast<const f4*>(pre + K16_PRE_CMETA + 4 * lane));
          jrows_load(pre, nc, (uint64_t)__float_as_uint(head.z) | ((uint64_t)__float_as_uint(head.w) << 32), S);
        }
        WSYNC();
        HSTAMP(54);
        __syncthreads();  // (3) contact rows of this step are in LDS
        float hp[G];
        hess_full(hp, nc);
#pragma unroll
        for (int q = 0; q < 4; q++) stv(&S.M[lane][4 * q], f4{hp[4 * q], hp[4 * q + 1], hp[4 * q + 2], hp[4 * q + 3]});
        HSTAMP(55);
        __syncthreads();  // (4) all-rows-active Hessian handed to the main wave
        helper_newton();
      }
      // then the closing FK of the step the main wave is finishing -- which is the opening FK of the step whose
      // action-independent half follows
      BodyK hk;
      fk_consts(hk);
      __syncthreads();  // (5) the main wave has integrated the jointed dofs
      HSTAMP(56);
      group_fk<true>(S, lane, nb, hparents, hk, row4);
      HSTAMP(57);
      __syncthreads();  // (6) link poses of the new state handed to the main wave
      if (FIRSTONLY) return;  // (the second half alone: the first half of the next step is the next launch)
    } else {
      // The collision wave opens the launch with the FORWARD KINEMATICS of the stored state: it needs one row of qpos and four of
      // the twelve quads of lane constants, so its loads are back sooner than the main wave's (which also brings in the model
      // table, the other rows and the action), and the main wave finds the link poses ready when it reaches the first barrier.
      const float hq_lo = lane < a.qst ? a.qpos[(size_t)env * a.qst + lane] : 0.0f;
      const float hq_hi = lane + G < a.qst ? a.qpos[(size_t)env * a.qst + lane + G] : 0.0f;
      BodyK hk;
      fk_consts(hk);
      if (lane < a.qst) S.qpos[lane] = hq_lo;
      if (lane + G < a.qst) S.qpos[lane + G] = hq_hi;
      WSYNC();
      group_fk(S, lane, nb, hparents, hk, row4);
    }
    HSTAMP(40);
    if (!ROT) __syncthreads();  // (1) link poses (this wave) and model table, velocities, targets (main wave) are in LDS
    HSTAMP(41);
    // the action-independent half ends with the contact data and the Jacobian rows in the pre buffer (the all-active Hessian is
    // accumulated from them by this wave at the start of the launch that consumes them, while the main wave starts on the action)
    auto pre_store = [&]() {
      if (valid && !ovf_h) {
        float* pre = a.pre + (size_t)env * K16_PRE_STRIDE;
        const int nc = S.ncon;
        if (CPL == 1 || nc <= K16_MAX_CONTACT) {
          const uint64_t qm = jrows_store(pre, nc, S);
          if (lane == 0) *reinterpret_cast<f4*>(pre + K16_PRE_HEAD) = f4{__int_as_float(nc), __int_as_float(S.coupled), __uint_as_float((uint32_t)qm), __uint_as_float((uint32_t)(qm >> 32))};
          if (lane < nc) *reinterpret_cast<f4*>(pre + K16_PRE_CMETA + 4 * lane) = ldv(S.con.cmeta[lane]);
        } else {
          // (three contacts per lane: more points than a scratch row holds -- a launch of the one-contact-per-lane kernel that reads the
          //  row defers the env on the count in the head word, bits 20 .. 27, and never looks at the rest; the contacts themselves go
          //  to the env's BIG row where the count is within this instantiation's capacity: the rotated launch with three contacts per
          //  lane picks them up, K48_MAGIC in the head says they are there)
          const bool bigok = CPL > 1 && a.pre_big != nullptr && ((S.coupled >> 20) & 255) <= MAXCON;
          if constexpr (CPL > 1) {
            if (bigok) {
              float* big = a.pre_big + (size_t)env * K48_STRIDE;
              bigrows_store(big, nc, S);
#pragma unroll
              for (int sl = 0; sl < CPL; sl++)
                if (lane + G * sl < nc) *reinterpret_cast<f4*>(big + K48_CMETA + 4 * (lane + G * sl)) = ldv(S.con.cmeta[lane + G * sl]);
              if (lane == 0) *reinterpret_cast<f4*>(big + K48_HEAD) = f4{__int_as_float(nc), __int_as_float(S.conB[1]), __int_as_float(S.conB[2]), 0.0f};
            }
          }
          if (lane == 0) *reinterpret_cast<f4*>(pre + K16_PRE_HEAD) = f4{__int_as_float(0), __int_as_float(S.coupled), __int_as_float(bigok ? K48_MAGIC : 0), 0.0f};
        }
      }
#ifdef MIR_PROFILE_SINGLE
      // (debug: wall clock of the last exit among the workgroups on the watched one's XCD = the end of the launch)
      if (a.prof && threadIdx.x == 64 && (blockIdx.x & 7) == (unsigned)(prof_blk & 7)) atomicMax(&a.prof[31], (unsigned long long)__builtin_amdgcn_s_memrealtime());
#endif
    };
    helper_cinert(1);
    const int cnt = collide_detect(1);
    HSTAMP(42);
    const int pts0 = contacts_build(cnt, PRE || ROT || BIGV);  // (three contacts per lane: the main wave shares the Jacobian build in every pass)  // (barrier (2) sits inside, between the arithmetic and the stores of the contact arrays)
    if (BIGV && !ROT && !PRE) ovf_h = pts0 > defer_above;  // (beyond this instantiation's capacity too: nothing is stored for the env, the wave-per-env kernel takes it)
    HSTAMP(45);
    __syncthreads();  // (3) contact arrays and base Jacobians handed to the main wave
    HSTAMP(46);
    {
      if (PRE || ROT) {
        // (the first-half launch of an overflow run tells the host which envs the next step finds above the one-contact-per-lane capacity
        //  -- a tagged byte per env, a word per workgroup, in pinned memory -- so that its second half can go out as two lists: those on
        //  the three-contacts-per-lane instantiation, the others in one round of the one-contact-per-lane kernel's workgroups)
        if (VARIANT == 10 && a.next_host) {
          const unsigned long long fb = __ballot(valid && pts0 > a.over_cap && lane == 0);
          if (tid == 0) {
            const uint32_t bits = (uint32_t)(fb & 1u) | (uint32_t)(fb >> 16 & 1u) << 8 | (uint32_t)(fb >> 32 & 1u) << 16 | (uint32_t)(fb >> 48 & 1u) << 24;
            __hip_atomic_store(a.next_host + blockIdx.x, bits | (a.term_tag << 1) * 0x01010101u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        }
        pre_store();
        return;
      }
      float hp[G];
      hess_full(hp, S.ncon);
#pragma unroll
      for (int q = 0; q < 4; q++) stv(&S.M[lane][4 * q], f4{hp[4 * q], hp[4 * q + 1], hp[4 * q + 2], hp[4 * q + 3]});
    }
    HSTAMP(47);
    __syncthreads();  // (4) all-rows-active Hessian handed to the main wave
    helper_newton();
    if (fksplit) {
      // the closing forward kinematics of the jointed bodies, beside the main wave's quaternion integration of the free bodies
      // and its state stores (the four quads of lane constants are fetched again: nothing of the opening FK was kept in registers)
      BodyK hk;
      fk_consts(hk);
      __syncthreads();  // (5) the main wave has integrated the jointed dofs
      group_fk<true>(S, lane, nb, hparents, hk, row4);
      __syncthreads();  // (6) link poses of the new state handed to the main wave
    }
    if constexpr (BIG2) {
      // the list instantiation goes on like the rotated launch: the closing FK above is the opening FK of the next step, whose
      // action-independent half follows (host side: only scenes with the split closing FK take this instantiation)
      HSTAMP(142);
      prof_mute = true;
      helper_cinert(2);
      const int cnt2 = collide_detect(2);
      contacts_build(cnt2, true);
      __syncthreads();  // (3) of the second pass
      pre_store();
#ifdef MIR_PROFILE_SINGLE
      if (a.prof && (int)blockIdx.x == prof_blk && threadIdx.x == 64) a.prof[141] = __builtin_readcyclecounter();
#endif
    }
    return;
  }

  // ---- per-lane model constants (lane = body = dof): twelve independent 16-byte loads (LaneK16) -------------
  const bool isbody = lane < nb && lane > 0;
  const bool isdof = lane < nv;
  const uint64_t parents = m->parents;
  f4 lk[12];
  {
#pragma unroll
    for (int k = 0; k < 12; k++) lk[k] = *reinterpret_cast<const f4*>(m->lanek_t[k][lane]);
  }
  BodyK bk;
  bk.jtype = __float_as_int(lk[0].x); bk.qadr = __float_as_int(lk[0].y);
  const int b_root = __float_as_int(lk[0].z), d_body = __float_as_int(lk[0].w);
  bk.pos = v3(lk[1].x, lk[1].y, lk[1].z);
  bk.quat = Q4{lk[2].x, lk[2].y, lk[2].z, lk[2].w};
  bk.axis = v3(lk[3].x, lk[3].y, lk[3].z);
  const V3 b_ipos = v3(lk[4].x, lk[4].y, lk[4].z);
  const float b_mass = lk[4].w;
  const float ib[6] = {lk[5].x, lk[5].y, lk[5].z, lk[5].w, lk[6].x, lk[6].y};
  const int d_kind = __float_as_int(lk[6].z), d_qadr = __float_as_int(lk[6].w);
  const int d_axis_k = __float_as_int(lk[7].x), d_root = __float_as_int(lk[7].y), d_ctrl = __float_as_int(lk[7].z), d_uadr = __float_as_int(lk[7].w);
  const V3 d_axis = v3(lk[8].x, lk[8].y, lk[8].z);
  const uint32_t d_ancmask = __float_as_uint(lk[9].y);
  const bool d_limited = __float_as_int(lk[9].z) != 0;
  const float d_damping = lk[9].w, d_kp = lk[10].x, d_kv = lk[10].y, d_frclo = lk[10].z, d_frchi = lk[10].w, d_mdiag = lk[11].x;
  const int obs_qadr = __float_as_int(lk[11].y);
  // tree-scan links (mir_compile.cpp): scan parent of the dof, the dof whose inclusive sum is the velocity in front of this
  // dof, the last dof that moves this body, the lane behind this body's subtree (bytes of one word; 255 = none)
  const int scanw = __float_as_int(lk[11].z);
  const int d_par = (int)(signed char)(scanw & 255), d_bef = (int)(signed char)(scanw >> 8 & 255);
  const int b_last = (int)(signed char)(scanw >> 16 & 255), b_next = scanw >> 24 & 255;

  // ---- load state -----------------------------------------------------------------------------
  // (addresses from the launch arguments only: these loads leave together with the model loads above)
  static_assert(sizeof(((EnvLds*)nullptr)->qpos) / sizeof(float) <= 2 * G, "qpos row: at most two entries per lane");
  // (DUAL: the qpos row is fetched and stored by the collision wave, with the forward kinematics)
  const float q_lo = ((!DUAL || ROT) && lane < a.qst) ? a.qpos[(size_t)env * a.qst + lane] : 0.0f;
  const float q_hi = ((!DUAL || ROT) && lane + G < a.qst) ? a.qpos[(size_t)env * a.qst + lane + G] : 0.0f;
  const float qv_in = a.qvel[(size_t)env * G + lane], ws_in = a.qacc_ws[(size_t)env * G + lane];
  // (with an action every controlled dof takes its target from it and nothing else reads a target: the stored row is not fetched)
  float tg = a.action ? 0.0f : a.target[(size_t)env * G + lane];
  const float au = (a.action && lane < a.nu) ? a.action[(size_t)env * a.nu + lane] : 0.0f;
  __builtin_amdgcn_sched_barrier(0);  // (nothing below may move in front of the loads above)
  {
    f4* dst = reinterpret_cast<f4*>(&T);
#pragma unroll
    for (int k = 0; k < TAB_NPASS; k++)
      if (tid + 64 * k < TAB_NQ) dst[tid + 64 * k] = tabtmp[k];
    if ((FEAT & 1) && tid < K16_MAX_VERT) stv(s_hull[tid], hulltmp);
  }
  if (!DUAL || ROT) {
    if (lane < a.qst) S.qpos[lane] = q_lo;
    if (lane + G < a.qst) S.qpos[lane + G] = q_hi;
  }
  S.qvel[lane] = qv_in;
  S.qacc_ws[lane] = ws_in;
  if (a.action) {  // lane u fetched action component u; the dof that it drives picks it up across the row
    const float mine = __shfl(au, (tid & ~(G - 1)) + (d_uadr >= 0 ? d_uadr : 0));
    if (isdof && d_uadr >= 0) tg = mine;
  }
  S.target[lane] = tg;
  if (lane == 0) {
    if (!ROT) { S.ncon = 0; S.ncand = 0; }  // (rotated launch: the collision wave is writing this step's contact count meanwhile)
    S.cin_ready = 0;
    if constexpr (CPL > 1) { S.bp_ready = 0; S.bb_done = 0; }
  }
  if constexpr (CPL > 1) { if (tid == 0) { s_help[0] = 0; s_help[1] = 0; s_help[2] = 0; } }
  WSYNC();

  // ======================= forward kinematics =================================================
  // Link poses are a function of qpos and are recomputed at the start of every launch: four pointer-jumping rounds (~1 us)
  // instead of 16 x 32 B per env written by one launch and read back by the next (round 1 cached them in HBM: 2.8x the
  // algorithmic traffic).
  STAMP(0);
  if (!DUAL && !POST) group_fk(S, lane, nb, parents, bk, row4);
  STAMP(1);
  STAMP(48);
  if (DUAL && !ROT) __syncthreads();  // (1) link poses from the collision wave; model table, velocities and targets from this one
  // (ROT: the loop below runs twice -- pass 0 is the second half of this step, pass 1 the first half of the next one)
  const int nsteps = (ROT || BIG2) ? 2 : (SINGLE ? 1 : (a.mode == 0 ? a.n_steps : (a.mode == 1 ? 1 : 0)));
  // (the action-independent half alone integrates nothing.  Of the scene-specialised instantiations the ROTATED launch stores poses too
  //  since round 5 -- the pointer test costs it nothing measurable, and the steps of the pixel modes keep the faster instantiation; the
  //  fused launch lost 2 % to the same code, so a fused launch that wants poses takes the generic-scene instantiation: launch() in mir_api.hip)
  if (PRE || (SPEC && !ROT && !BIGV)) a.poses = nullptr;
  if (SINGLE) { a.mode = 0; a.act_step = 0; a.rows_step = 0; a.ar.episode_len = nullptr; a.out_M = a.out_bias = a.out_qas = a.out_qacc = a.out_xpos = a.out_xquat = nullptr;
#ifndef MIR_PROFILE_SINGLE  /* (a profiling build keeps the phase stamps in the single-step instantiation: tools/phase_profile.py) */
    a.prof = nullptr;
#endif
  }
  if (VARIANT == 1) { a.mode = 0; a.out_M = a.out_bias = a.out_qas = a.out_qacc = a.out_xpos = a.out_xquat = nullptr; a.prof = nullptr; a.agent_pos = a.env_state = a.reward = nullptr; a.terminated = a.term_host = nullptr; a.done_ticket = nullptr; }
  // packed output row [agent_pos | env_state | reward | terminated] of the current kinematic state
  const int eb = mdl_eef, ob = mdl_obj;
  const int ad = 7 + mdl_ngrip;
  auto column = [&](int c) -> float {
    const V3 pe = ld3v(S.xpos[eb]), po = ld3v(S.xpos[ob]);
    const V3 df = pe - po;
    if (c < 3) return S.xpos[eb][c];
    if (c < 7) return S.xquat[eb][c - 3];
    if (c < ad) return S.qpos[c == lane ? obs_qadr : m->grip_qadr[c - 7]];  // (every caller asks for its own lane: no model trip)
    const int k = c - ad;
    if (k < 3) return S.xpos[ob][k];
    if (k < 7) return S.xquat[ob][k - 3];
    if (k < 10) return k == 7 ? df.x : (k == 8 ? df.y : df.z);
    if (k == 10) return sqrtf(dot(df, df));
    return above(po.z, mdl_reward_z) ? 1.0f : 0.0f;  // k == 11 reward, k == 12 terminated
  };
  int eplen = a.ar.episode_len ? a.ar.episode_len[env] : 0, epcur = a.ar.episode_len ? a.ar.cursor[env] : 0;
  // POST (the second half of a split step): everything the previous launch left in the pre buffer is fetched here, in one batch --
  // the mass-matrix row and the bias force into registers; contact count, coupling flag, row constants and Jacobian rows into the
  // LDS arrays the rest of the step reads (by the collision wave in the rotated launch, which then accumulates the all-rows-active
  // Hessian from them while this wave starts on the action)
  f4 pre_m[4] = {};
  float pre_bias = 0.0f;
  if (POST || ROT) {
    const float* pre = a.pre + (size_t)env * K16_PRE_STRIDE;
#pragma unroll
    for (int q = 0; q < 4; q++) {  // (a quad outside the row's block reads the row's first quad and is zeroed: no branch around a load)
      const bool in = lane < nv && q >= pk_qlo && q < pk_qhi;
      const f4 v = *reinterpret_cast<const f4*>(pre + K16_PRE_MROW + (lane < nv ? pk_moff : 0) + (in ? 4 * (q - pk_qlo) : 0));
      pre_m[q] = in ? v : f4{0, 0, 0, 0};
    }
    pre_bias = pre[K16_PRE_BIAS + lane];
    if (POST) {  // (one wave: it fetches the contact rows itself; in the rotated launch the collision wave does)
      const f4 head = *reinterpret_cast<const f4*>(pre + K16_PRE_HEAD);
      const int nc = __float_as_int(head.x);
      if (lane == 0) { S.ncon = nc; S.coupled = __float_as_int(head.y); S.ncand = 0; }
      if (lane < nc) stv(S.con.cmeta[lane], *reinterpret_cast<const f4*>(pre + K16_PRE_CMETA + 4 * lane));
      jrows_load(pre, nc, (uint64_t)__float_as_uint(head.z) | ((uint64_t)__float_as_uint(head.w) << 32), S);
      WSYNC();
    }
  }
  // ---- what a step hands back: host-visible terminated bytes, state rows, observations (after the step loop; in the rotated
  // launch after its first pass) -------------------------------------------------------------------------------------------
  auto emit_outputs = [&]() {
    STAMP(10);
    // GenesisEnv.step's D->H copy of `terminated`, done by the kernel and issued FIRST: the four masks of the wave as ONE 32-bit store
    // straight into pinned host memory (write-through, system scope), each byte = term | tag << 1; its trip over PCIe runs under the
    // state and observation stores below.  The tag changes from launch to launch, so the host recognises the bytes of THIS launch by
    // themselves (sync mode 3: no fence, no ticket, nothing waits).
    const bool term_now = valid && !ovf_env && above(S.xpos[ob][2], mdl_reward_z);
    if (VARIANT != 1 && a.term_host && !term_early) {
      const unsigned long long tb = __ballot(term_now && lane == 0), db = __ballot(ovf_env && valid && lane == 0), ob = BIGV ? __ballot(over_env && valid && lane == 0) : 0ull;
      if (tid == 0) {
        const uint32_t bits = (uint32_t)(tb & 1u) | (uint32_t)(tb >> 16 & 1u) << 8 | (uint32_t)(tb >> 32 & 1u) << 16 | (uint32_t)(tb >> 48 & 1u) << 24 |
                              (uint32_t)(db & 1u) << 7 | (uint32_t)(db >> 16 & 1u) << 15 | (uint32_t)(db >> 32 & 1u) << 23 | (uint32_t)(db >> 48 & 1u) << 31 |
                              (uint32_t)(ob & 1u) << 6 | (uint32_t)(ob >> 16 & 1u) << 14 | (uint32_t)(ob >> 32 & 1u) << 22 | (uint32_t)(ob >> 48 & 1u) << 30;
        __hip_atomic_store(reinterpret_cast<uint32_t*>(a.term_host) + (size_t)blockIdx.x * a.term_wstride, bits | (a.term_tag << 1) * 0x01010101u, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    if (valid && !ovf_env) {  // (a deferred env -- exact contacts -- keeps the state it had: the wave kernel steps it from there)
      if (a.poses && lane < nb) {  // (link poses for the rasteriser: the pose-refresh launch, and every integrating launch once a render has been asked for)
        float* p = a.poses + ((size_t)env * 2 * G + lane) * 4;
        *reinterpret_cast<f4*>(p) = ldv(S.xpos[lane]);
        *reinterpret_cast<f4*>(p + 4 * G) = ldv(S.xquat[lane]);
      }
      // ---- store state ---------------------------------------------------------------------------------
      if (a.mode == 0) {
        for (int i = lane; i < qst; i += G) a.qpos[(size_t)env * qst + i] = S.qpos[i];
        a.qvel[(size_t)env * G + lane] = S.qvel[lane];
        a.qacc_ws[(size_t)env * G + lane] = S.qacc_ws[lane];
        // (the wave kernel keeps the link poses of the stored state in HBM and opens with them; this kernel did too in round 3 --
        //  +2.7 % on bare fused launches for 832 B per env-step, 3.7 x the algorithmic traffic instead of 1.7 x -- and does not any more)
      }
      if (a.action) a.target[(size_t)env * G + lane] = S.target[lane];
      // ---- observations (get_obs / compute_reward / terminated) ---------------------------------------
      const float rew = term_now ? 1.0f : 0.0f;
      if (a.agent_pos && lane < ad) a.agent_pos[(size_t)env * ad + lane] = column(lane);
      if (a.env_state && lane < 11) a.env_state[(size_t)env * 11 + lane] = column(ad + lane);
      if (lane == 0) {
        if (a.reward) a.reward[env] = rew;
        if (a.terminated) a.terminated[env] = rew == 1.0f ? 1 : 0;
      }
      if (a.ar.episode_len && lane == 0) { a.ar.episode_len[env] = eplen; a.ar.cursor[env] = epcur; }
      if (a.rows && !(a.ar.episode_len && a.rows_step)) {  // (in rollout mode: the last step's row; with autoreset it was written in the loop)
        float* row = a.rows + (size_t)(a.rows_step ? (nsteps > 0 ? nsteps - 1 : 0) : 0) * a.rows_step + (size_t)env * a.row_stride;
        for (int c = lane; c < ad + 13; c += G) row[c] = column(c);
      }
      if (a.out_xpos && lane < nb) {
        st3(&a.out_xpos[((size_t)env * nb + lane) * 3], ld3v(S.xpos[lane]));
        st4(&a.out_xquat[((size_t)env * nb + lane) * 4], ld4v(S.xquat[lane]));
      }
    }  // valid
    if (VARIANT != 1 && a.done_ticket) {
      // Completion published by the kernel itself (mir_step_begin, sync mode 2): every wave waits for its host store to be
      // acknowledged (~3 us over PCIe), then takes a ticket; the wave that takes the last one knows that every terminated byte of the launch is in
      // host memory and writes the sequence number the host is spinning on.  (The host-side stores above are system-scope
      // write-through atomics, so no cache write-back is needed to order them: s_waitcnt is the release.)
      __builtin_amdgcn_s_waitcnt(0);
      unsigned old = 0;
      if (tid == 0) old = __hip_atomic_fetch_add(a.done_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = __builtin_amdgcn_readfirstlane(old);
      if (old == gridDim.x - 1 && tid == 0) {
        __hip_atomic_store(a.done_ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // the next launch starts from zero
        __hip_atomic_store(a.done_flag, a.done_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    STAMP(25);
    if (a.prof && threadIdx.x == 0) {  // debug: wall-clock (100 MHz) exit time of block 0 and of the last block
      const unsigned long long tnow = __builtin_amdgcn_s_memrealtime();
      if ((int)blockIdx.x == prof_blk) a.prof[27] = tnow;
      if ((blockIdx.x & 7) == (unsigned)(prof_blk & 7)) {  // same XCD as block 0: the realtime counters of different XCDs are not aligned
        atomicMax(&a.prof[28], tnow);
        atomicMin(&a.prof[29], tnow);
      }
    }
  };
  bool bad_acc = false;  // (diagnostics on) the env's state went non-finite in some step of this launch
  // One step of the loop as a function of the step index: an int in the step-loop instantiations, a compile-time constant in the
  // rotated launch (pass 0 = second half of this step, pass 1 = first half of the next), whose two calls therefore compile to
  // straight-line code like the single-step kernel.  Returns 0 to go on, 1 to leave the loop, 2 to leave the kernel.
  auto step_body = [&](auto stepv) __attribute__((always_inline)) -> int {
    const int step = stepv;
    // rollout mode (mir_rollout): a fresh action block per step
    if (step > 0 && a.action && a.act_step) {
      if (isdof && d_uadr >= 0) S.target[lane] = a.action[(size_t)step * a.act_step + (size_t)env * a.nu + d_uadr];
    }
    const bool post_now = POST || (ROT && step == 0), pre_now = PRE || ((ROT || BIG2) && step == 1);
    if (BIG2 && step == 1) prof_mute = true;
    float qfrc_bias = 0.0f, qfs = 0.0f;
    if (!post_now) {
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
    } else {  // (a lane without a dof still owns a row: zeros, not stale LDS)
      st3v(&S.cdof[lane][0], v3(0, 0, 0));
      st3v(&S.cdof[lane][4], v3(0, 0, 0));
    }
    if (!DUAL) cinert_store(isbody, ib, b_ipos, b_mass, b_root);
    WSYNC();
    STAMP(32);
    // ======================= velocities, composite inertias, body forces: tree SCANS =============
    // Sums over the ancestors of a dof are inclusive prefix sums along its dof chain: POINTER JUMPING over the chain's parent
    // links (<= 4 rounds of one lane gather each, as in the FK) instead of a masked gather per lane and per quantity.  Sums over
    // the subtree of a body are suffix sums over the body lanes -- bodies are numbered in depth-first preorder, so a subtree is
    // the lane range [b, b_next) -- formed by four DPP row shifts per component and one subtraction (distal bodies sit at the
    // end of the row, so the subtraction never takes a small subtree out of a large total).
    {
      // ancestor scan of a 6-vector held by the dof lanes; the result table (inclusive sums, by dof lane) is left in `tab`
      auto ancestor_scan = [&](V3& A, V3& Bv, float* tab) {
        int p = isdof ? d_par : -1;
#pragma unroll 1
        for (int round = 0; round < 4; round++) {
          if (!__any(p >= 0)) break;
          const int src = row4 + ((p >= 0 ? p : lane) << 2);
          const V3 xa = v3(lane_gather(src, A.x), lane_gather(src, A.y), lane_gather(src, A.z));
          const V3 xb = v3(lane_gather(src, Bv.x), lane_gather(src, Bv.y), lane_gather(src, Bv.z));
          const int nxt = lane_gather(src, p);
          if (p >= 0) {
            A = A + xa;
            Bv = Bv + xb;
            p = nxt;
          }
        }
        st3v(tab + 8 * lane, A);
        st3v(tab + 8 * lane + 4, Bv);
        WSYNC();
      };
      const V3 cw = ld3v(&S.cdof[lane][0]), cv = ld3v(&S.cdof[lane][4]);
      const float qd = isdof ? S.qvel[lane] : 0.0f;
      // (1) V_i = sum over the dof chain up to and including i of qvel_j cdof_j
      V3 Vw = isdof ? qd * cw : v3(0, 0, 0), Vv = isdof ? qd * cv : v3(0, 0, 0);
      ancestor_scan(Vw, Vv, &S.dyn.cvel[0][0]);
      STAMP(33);
      // cdof_dot * qvel from the velocity in front of the dof; body velocity = V at the last dof that moves the body
      V3 Yw = v3(0, 0, 0), Yv = v3(0, 0, 0);
      if (isdof) {
        V3 pw = v3(0, 0, 0), pv = v3(0, 0, 0);
        if (d_bef >= 0) { pw = ld3v(&S.dyn.cvel[d_bef][0]); pv = ld3v(&S.dyn.cvel[d_bef][4]); }
        Yw = qd * cross(pw, cw);
        Yv = qd * (cross(pw, cv) + cross(pv, cw));
      }
      V3 w = v3(0, 0, 0), v = v3(0, 0, 0);
      if (isbody && b_last >= 0) { w = ld3v(&S.dyn.cvel[b_last][0]); v = ld3v(&S.dyn.cvel[b_last][4]); }
      // (2) composite inertia: suffix sums of the body inertias over the row, minus the suffix behind the subtree
      {
        if (DUAL) {  // (the body inertias come from the collision wave: long since there, as a rule)
          while (__hip_atomic_load(&S.cin_ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < ((BIGV && !ROT) ? step + 1 : 1)) __builtin_amdgcn_s_sleep(1);
        }
        float* ci = S.dyn.cinert[lane];
        f4 c0 = ldv(ci), c1 = ldv(ci + 4), c2 = ldv(ci + 8);
        float comp[10] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w, c2.x, c2.y};
#pragma unroll
        for (int k = 0; k < 10; k++) {
          float t = comp[k];
          t += row_shl<1>(t); t += row_shl<2>(t); t += row_shl<4>(t); t += row_shl<8>(t);
          comp[k] = t;
        }
        float* cs = S.dyn.crb[lane];
        // (the suffix behind the subtree is the suffix sum of lane b_next: a lane gather, no LDS round trip)
        const int src = row4 + ((b_next < G ? b_next : lane) << 2);
        float e[10];
#pragma unroll
        for (int k = 0; k < 10; k++) {
          const float gk = lane_gather(src, comp[k]);
          e[k] = b_next < G ? gk : 0.0f;
        }
        const bool own = isbody;
        stv(cs, own ? f4{comp[0] - e[0], comp[1] - e[1], comp[2] - e[2], comp[3] - e[3]} : f4{0, 0, 0, 0});
        stv(cs + 4, own ? f4{comp[4] - e[4], comp[5] - e[5], comp[6] - e[6], comp[7] - e[7]} : f4{0, 0, 0, 0});
        stv(cs + 8, own ? f4{comp[8] - e[8], comp[9] - e[9], 0.0f, 0.0f} : f4{0, 0, 0, 0});
      }
      STAMP(2);
      // (3) A_i = sum over the dof chain of cdof_dot_j qvel_j; body forces at zero acceleration (RNE)
      ancestor_scan(Yw, Yv, &S.dyn.cddq[0][0]);
      STAMP(34);
      V3 t = v3(0, 0, 0), f = v3(0, 0, 0);
      if (isbody) {
        V3 aw = v3(0, 0, 0), av = v3(-mdl_gx, -mdl_gy, -mdl_gz);
        if (b_last >= 0) { aw = aw + ld3v(&S.dyn.cddq[b_last][0]); av = av + ld3v(&S.dyn.cddq[b_last][4]); }
        Inert I = ldI(S.dyn.cinert[lane]);
        V3 ta, fa, tv, fv;
        imul(I, aw, av, ta, fa);
        imul(I, w, v, tv, fv);
        t = ta + cross(w, tv) + cross(v, fv);
        f = fa + cross(w, fv);
      }
      // zero this lane's row of M, then (after the fence) fill the tree-sparse entries
#pragma unroll
      for (int q = 0; q < 4; q++) stv(&S.M[lane][4 * q], f4{0, 0, 0, 0});
      // (4) subtree forces: suffix sums again
      {
        float comp[6] = {t.x, t.y, t.z, f.x, f.y, f.z};
#pragma unroll
        for (int k = 0; k < 6; k++) {
          float u = comp[k];
          u += row_shl<1>(u); u += row_shl<2>(u); u += row_shl<4>(u); u += row_shl<8>(u);
          comp[k] = u;
        }
        float* fs = S.dyn.cfrc[lane];
        const int src = row4 + ((b_next < G ? b_next : lane) << 2);
        float e[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
          const float gk = lane_gather(src, comp[k]);
          e[k] = b_next < G ? gk : 0.0f;
        }
        st3v(fs, v3(comp[0], comp[1], comp[2]) - v3(e[0], e[1], e[2]));
        st3v(fs + 4, v3(comp[3], comp[4], comp[5]) - v3(e[3], e[4], e[5]));
      }
      WSYNC();
      STAMP(35);
      if (isdof) {  // M[i][j] = cdof_j . (crb_body(i) cdof_i), j over ancestors-or-self
        Inert I = ldI(S.dyn.crb[d_body]);
        V3 bt, bf;
        imul(I, cw, cv, bt, bf);
        uint32_t mk = d_ancmask;
        while (mk) {  // four ancestors per trip, reads batched
          int j[4];
          bool ok[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { ok[u] = mk != 0u; j[u] = ok[u] ? __ffs(mk) - 1 : 0; mk &= mk - 1u; }
          f4 ca[4], cl[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { ca[u] = ldv(&S.cdof[j[u]][0]); cl[u] = ldv(&S.cdof[j[u]][4]); }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int u = 0; u < 4; u++)
            if (ok[u]) {
              float val = dot(v3(ca[u].x, ca[u].y, ca[u].z), bt) + dot(v3(cl[u].x, cl[u].y, cl[u].z), bf);
              if (j[u] == lane) val += d_mdiag;
              S.M[lane][j[u]] = val;
              S.M[j[u]][lane] = val;
            }
        }
        // bias = cdof . (forces of the subtree of the dof's body); smooth force
        const V3 ft = ld3v(&S.dyn.cfrc[d_body][0]), ff = ld3v(&S.dyn.cfrc[d_body][4]);
        qfrc_bias = dot(cw, ft) + dot(cv, ff);
        float fa = 0.0f;
        if (d_ctrl == MIR_CTRL_POSITION) {
          fa = d_kp * (S.target[lane] - S.qpos[d_qadr]) - d_kv * qd;
          fa = fminf(fmaxf(fa, d_frclo), d_frchi);
        }
        qfs = -d_damping * qd + fa - qfrc_bias;
      }
    }
    }  // !post_now
    WSYNC();
    STAMP(3);
    if (BIGV && !post_now) box_share(ROT ? 1 : step + 1);
    STAMP(49);
    if (DUAL && !post_now) __syncthreads();  // (2) this wave is done with the dynamics scratch (M is in its own area, the rest in registers)
    // qacc_smooth = Mt^-1 qfrc_smooth: Gauss-Jordan on register rows
    float mrow[G];
    {
      f4 r0, r1, r2, r3;
      if (post_now) {
        // the action-independent half of this step was computed by the previous launch (VARIANT 3): mass-matrix row and bias force
        // from the pre buffer, then the smooth force with THIS launch's targets (the expressions of the fused kernel)
        r0 = pre_m[0]; r1 = pre_m[1]; r2 = pre_m[2]; r3 = pre_m[3];
        qfrc_bias = pre_bias;
        if (isdof) {
          const float qd = S.qvel[lane];
          float fa = 0.0f;
          if (d_ctrl == MIR_CTRL_POSITION) {
            fa = d_kp * (S.target[lane] - S.qpos[d_qadr]) - d_kv * qd;
            fa = fminf(fmaxf(fa, d_frclo), d_frchi);
          }
          qfs = -d_damping * qd + fa - qfrc_bias;
        } else {
          qfrc_bias = 0.0f;
        }
      } else {
        r0 = ldv(&S.M[lane][0]); r1 = ldv(&S.M[lane][4]); r2 = ldv(&S.M[lane][8]); r3 = ldv(&S.M[lane][12]);
      }
      if (pre_now) {
        if (valid && !ovf_env) {
          float* pre = a.pre + (size_t)env * K16_PRE_STRIDE;
          const f4 rq[4] = {r0, r1, r2, r3};
#pragma unroll
          for (int q = 0; q < 4; q++)
            if (lane < nv && q >= pk_qlo && q < pk_qhi) *reinterpret_cast<f4*>(pre + K16_PRE_MROW + pk_moff + 4 * (q - pk_qlo)) = rq[q];
          pre[K16_PRE_BIAS + lane] = qfrc_bias;
        }
        __syncthreads();  // (2b) the collision wave has stored the contact arrays: this wave builds every other pair of Jacobian rows
        jac_build(2, 4);
        __syncthreads();  // (3) every Jacobian row of the coming step is in LDS: the collision wave stores them
#ifdef MIR_PROFILE_SINGLE
        if (BIG2 && a.prof && (int)blockIdx.x == prof_blk && threadIdx.x == 0) a.prof[140] = __builtin_readcyclecounter();
#endif
        return 2;
      }
      mrow[0] = r0.x; mrow[1] = r0.y; mrow[2] = r0.z; mrow[3] = r0.w; mrow[4] = r1.x; mrow[5] = r1.y; mrow[6] = r1.z; mrow[7] = r1.w;
      mrow[8] = r2.x; mrow[9] = r2.y; mrow[10] = r2.z; mrow[11] = r2.w; mrow[12] = r3.x; mrow[13] = r3.y; mrow[14] = r3.z; mrow[15] = r3.w;
    }
    if (a.out_M && valid && isdof && step == 0) {
#pragma unroll
      for (int j = 0; j < G; j++)
        if (j < nv) a.out_M[((size_t)env * nv + lane) * nv + j] = mrow[j] - (j == lane ? d_mdiag - m->d_armature[lane] : 0.0f);
    }
    if (a.out_bias && valid && isdof && step == 0) a.out_bias[(size_t)env * nv + lane] = qfrc_bias;
    float qas;
    {
      float arow[G];
#pragma unroll
      for (int j = 0; j < G; j++) arow[j] = isdof ? mrow[j] : (j == lane ? 1.0f : 0.0f);
      qas = isdof ? qfs : 0.0f;
      gj_solve(arow, qas, lane, mdl_split, nv);  // (the mass matrix is block diagonal by tree)
    }
    if (a.out_qas && valid && isdof && step == 0) a.out_qas[(size_t)env * nv + lane] = qas;
    WSYNC();  // dyn scratch is dead from here on
    STAMP(4);

    // ======================= collision detection, contact arrays, contact Jacobians ==============
    // (DUAL: the collision wave does all of it, detection since the first barrier, the rest since the second)
    if (!DUAL && !POST) {
      const int mc = collide_detect(1);
      contacts_build(mc, false);
    }

    // joint-limit rows: lane = dof, lane-private
    float lsg = 0.0f, lD = 0.0f, laref = 0.0f;
    if (d_limited) {
      float q = S.qpos[d_qadr];
      const f4 l0 = ldv(&T.d_lim[lane][0]);  // lo, hi, invweight0, k
      float dlo = q - l0.x, dhi = l0.y - q;
      float pos = 0.0f;
      if (dlo < 0.0f) { pos = dlo; lsg = 1.0f; }
      else if (dhi < 0.0f) { pos = dhi; lsg = -1.0f; }
      if (lsg != 0.0f) {
        const f4 l1 = ldv(&T.d_lim[lane][4]), l2 = ldv(&T.d_lim[lane][8]);  // b, solimp[0..2] | solimp[3..4]
        float imp = impedance(l1.y, l1.z, l1.w, l2.x, l2.y, pos);
        float Rr = fmaxf((1.0f - imp) / imp * l0.z, 1e-15f);
        lD = 1.0f / Rr;
        laref = -l1.x * (lsg * S.qvel[lane]) - l0.w * imp * pos;
      }
    }
    WSYNC();
    if (BIGV && !post_now) {  // (three contacts per lane: every other pair of contacts of the Jacobian build, as in the action-independent half)
      __syncthreads();  // (2b)
      jac_build(2, 4);
    }
    STAMP(50);
    if (DUAL && (!post_now || ROT)) __syncthreads();  // (3) contact arrays and base Jacobians are in LDS
    STAMP(52);
    const int ncon = S.ncon;
    // contact rows, lane = contact, lane-private: aref_r = -b (J_r qvel) - k imp dist
    // (CPL > 1, the list instantiation: lane c owns contacts c + 16 s, s < CPL -- every contact quantity below once per slot, the sums
    //  over a lane's slots in slot order; a slot that holds no contact anywhere in the wave is skipped, and an empty slot adds exact
    //  zeros: with at most 16 contacts the arithmetic is that of the one-contact-per-lane instantiations)
    bool iscon[CPL];
    float cmu[CPL], cD[CPL];
    float aref[CPL][4], jar[CPL][4];
#pragma unroll
    for (int sl = 0; sl < CPL; sl++) {
      iscon[sl] = lane + G * sl < ncon;
      cmu[sl] = 0.0f; cD[sl] = 0.0f;
#pragma unroll
      for (int r = 0; r < 4; r++) { aref[sl][r] = 0.0f; jar[sl][r] = 0.0f; }
    }
    // the contact's Jacobian rows stay in the registers of its lane for the whole solve; J x products take x_j from the dof
    // lanes by DPP row broadcast (every lane of the row takes part)
    // (CPL > 1: slot 0's rows only -- the other slots' are read from LDS where they are used, twelve 16-byte reads per product)
    const JRow jrow = jrow_load(JBROW(S, iscon[0] ? lane : 0));
    auto jd3 = [&](int sl, float x, float& dn, float& d1, float& d2) __attribute__((always_inline)) {
      if (sl == 0) jdot3_bc(jrow, x, dn, d1, d2);
      else jdot3_bc(jrow_load(JBROW(S, iscon[sl] ? lane + G * sl : 0)), x, dn, d1, d2);
    };
#pragma unroll
    for (int sl = 0; sl < CPL; sl++) {
      if (sl > 0 && !__any(iscon[sl])) continue;  // (wave-uniform)
      float vn, v1, v2;
      jd3(sl, S.qvel[lane], vn, v1, v2);
      if (iscon[sl]) {
        f4 mt = ldv(S.con.cmeta[lane + G * sl]);
        cmu[sl] = mt.x; cD[sl] = mt.y;
        const float base = mt.z, bb = mt.w;
        aref[sl][0] = base - bb * (vn + cmu[sl] * v1);
        aref[sl][1] = base - bb * (vn - cmu[sl] * v1);
        aref[sl][2] = base - bb * (vn + cmu[sl] * v2);
        aref[sl][3] = base - bb * (vn - cmu[sl] * v2);
      }
    }
    STAMP(6);

    // ======================= primal Newton solve ====================================================
    const uint32_t limmask = (uint32_t)(__ballot(lsg != 0.0f) >> (grp * G)) & 0xffffu;
    const int nefc = 4 * ncon + __popc(limmask);
    // the Hessian is block diagonal by tree unless a contact joins the arm and the cube somewhere in this wave
    const int cpl = S.coupled;  // bit 0: some contact joins the trees; bits 1 .. 16: contact c belongs to the second tree; 20 .. 27: see contacts_build
    if (DEFER && (!ROT || step == 0)) ovf_env = ((cpl >> 20) & 255) > defer_above;  // (exact contacts: more candidate points than lanes)
    if (BIGV) over_env = a.over_cap > 0 && ((cpl >> 20) & 255) > a.over_cap;
    const int hsplit = __any((cpl & 1) != 0) ? 0 : mdl_split;
    // Where no contact joins the two trees the problem SEPARATES -- f = f_A(a_A) + f_B(a_B), block-diagonal Hessian.  The line search
    // stays one per env, but a step is ACCEPTED tree by tree (below), so that each tree's own cost decreases monotonically: what the
    // early `terminated` bytes rely on.  `sep` is uniform over the env's row; an env that does not separate has every row in tree A.
    const bool sep = mdl_split > 0 && (cpl & 1) == 0;
    const bool dofB = sep && lane >= mdl_split;               // this lane's dof, and its joint-limit row
    bool conB[CPL];                                           // this lane's contact(s)
    conB[0] = sep && ((cpl >> (1 + lane)) & 1) != 0;
    if constexpr (CPL > 1) {
#pragma unroll
      for (int sl = 1; sl < CPL; sl++) conB[sl] = sep && ((S.conB[sl] >> lane) & 1) != 0;
    }
    bool done = nefc == 0;
    float qacc = qas, Ma = 0.0f, ljar = 0.0f;
    {
      // warm start: cost(ws) vs cost(qacc_smooth); Gauss part 1/2 dq^T Mt dq.  Where the problem separates the choice is made TREE BY
      // TREE (the cost is a sum over the trees, and so is every term below: Mt is block diagonal, a limit row belongs to its dof's
      // tree, a contact's rows to the one tree it touches): the cube keeps yesterday's solution -- at rest, the minimiser itself --
      // whatever a jump of the arm's targets does to the arm's, and the first gradient already bounds its height (the early bytes).
      const float ws = S.qacc_ws[lane];
      const float dq = isdof ? ws - qas : 0.0f;
      float d_ws = 0.5f * rowdot_bc(mrow, dq) * dq, d_sm = 0.0f;  // this lane's dof: Gauss term + limit row
      const float ljs = lsg * qas - laref, ljw = lsg * ws - laref;
      if (lsg != 0.0f) {
        if (ljs < 0.0f) d_sm += 0.5f * lD * ljs * ljs;
        if (ljw < 0.0f) d_ws += 0.5f * lD * ljw * ljw;
      }
      float js[CPL][4], jw[CPL][4], k_df[CPL];  // this lane's contact(s): the four rows at either start, the cost difference
#pragma unroll
      for (int sl = 0; sl < CPL; sl++) {
        float k_ws = 0.0f, k_sm = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; r++) { js[sl][r] = 0.0f; jw[sl][r] = 0.0f; }
        if (sl == 0 || __any(iscon[sl])) {  // (wave-uniform)
          float sn, s1, s2, wn, w1, w2;
          jd3(sl, qas, sn, s1, s2);
          jd3(sl, ws, wn, w1, w2);
          if (iscon[sl]) {
            const float mu = cmu[sl];
            js[sl][0] = sn + mu * s1 - aref[sl][0]; js[sl][1] = sn - mu * s1 - aref[sl][1]; js[sl][2] = sn + mu * s2 - aref[sl][2]; js[sl][3] = sn - mu * s2 - aref[sl][3];
            jw[sl][0] = wn + mu * w1 - aref[sl][0]; jw[sl][1] = wn - mu * w1 - aref[sl][1]; jw[sl][2] = wn + mu * w2 - aref[sl][2]; jw[sl][3] = wn - mu * w2 - aref[sl][3];
#pragma unroll
            for (int r = 0; r < 4; r++) {
              if (js[sl][r] < 0.0f) k_sm += 0.5f * cD[sl] * js[sl][r] * js[sl][r];
              if (jw[sl][r] < 0.0f) k_ws += 0.5f * cD[sl] * jw[sl][r] * jw[sl][r];
            }
          }
        }
        k_df[sl] = k_ws - k_sm;
      }
      // (one reduction per tree: the sign of the summed cost DIFFERENCES decides)
      const float d_df = d_ws - d_sm;
      float dfA = (dofB ? 0.0f : d_df) + (conB[0] ? 0.0f : k_df[0]), dfB = (dofB ? d_df : 0.0f) + (conB[0] ? k_df[0] : 0.0f);
#pragma unroll
      for (int sl = 1; sl < CPL; sl++) { dfA += conB[sl] ? 0.0f : k_df[sl]; dfB += conB[sl] ? k_df[sl] : 0.0f; }
      // (gsum_u: one value per env -- every lane must take the same decision, see mir_dev.h)
      bool usewsA = gsum_u(dfA) < 0.0f, usewsB = usewsA;
      if (__any(sep)) usewsB = gsum_u(dfB) < 0.0f;  // (wave-uniform)
      const bool usewsd = dofB ? usewsB : usewsA;
      qacc = usewsd ? ws : qas;
      ljar = usewsd ? ljw : ljs;
#pragma unroll
      for (int sl = 0; sl < CPL; sl++) {
        const bool usewsc = conB[sl] ? usewsB : usewsA;
#pragma unroll
        for (int r = 0; r < 4; r++) jar[sl][r] = usewsc ? jw[sl][r] : js[sl][r];
      }
      const float Mab = rowdot_bc(mrow, qacc);
      Ma = isdof ? Mab : 0.0f;
    }
    STAMP(7);
    int niter = 0;
    const float tol = mdl_tolerance, scale = mdl_scale;
    // float32 rounding floor of the gradient Ma - qfrc_smooth - J^T f: below it a Newton step no
    // longer changes qacc, so iterating further is noise (same rule as the oracle, with float eps).  The floor is that of the
    // forces the gradient is summed from, TREE BY TREE where the problem separates: a cube of 64 g is not converged because its
    // gradient has sunk below the rounding noise of the ARM's torques (with the per-tree warm start it may never have been far
    // above it).  One weighted norm does: each dof's gradient entry is measured in units of its own tree's force scale.
    constexpr float gfl = 16.0f * 5.96e-8f;
    float gfw;  // 1 / (force scale of this lane's tree)^2
    {
      const float f2 = Ma * Ma + qfs * qfs;
      const float FA2 = gsum(dofB ? 0.0f : f2);
      float FB2 = FA2;
      if (__any(sep)) FB2 = gsum(dofB ? f2 : 0.0f);  // (wave-uniform)
      gfw = 1.0f / fmaxf(dofB ? FB2 : FA2, 1e-30f);
    }
    // Hessian row kept across iterations: H = Mt + J^T D_active J is updated incrementally, only rows
    // whose active flag flipped since the previous iteration contribute a (signed) delta
    float hkeep[G];
#pragma unroll
    for (int j = 0; j < G; j++) hkeep[j] = isdof ? mrow[j] : (j == lane ? 1.0f : 0.0f);
    float oldlact = 0.0f;
    bool term_sent = false;    // (wave-uniform) the terminated bytes of this step have left, from inside the solver loop
    uint32_t term_bits = 0u;
    // (the weight of this lane's gradient entry in the bound, object mass folded in: fetched here, used after the first gradient)
    float d_gw = 0.0f;
    if (term_bound) d_gw = m->lanek_t[11][lane][3];
    unsigned prevbits[CPL];  // contact lane: flags written in the previous iteration
#pragma unroll
    for (int sl = 0; sl < CPL; sl++) prevbits[sl] = 0u;
    float gprev = 0.0f;
    bool met4 = false;
    // (three contacts per lane: the helper wave takes part in the Newton loop when some env of the workgroup has contacts above the first slot)
    [[maybe_unused]] const bool use_help = CPL > 1 && __any(ncon > G);
    for (int it = 0; it < mdl_iterations; it++) {
      if (!__any(!done)) break;
      // ---- forces of the active rows; base-force triple and active flags to LDS for the dof lanes
      float lact = (lsg != 0.0f && ljar < 0.0f) ? lD : 0.0f;
      const float lf = -lact * ljar;
      // the env's flipped contacts -- some pyramid row changed sides since the Hessian last saw the contact -- as a bit mask in every
      // one of its lanes: the Hessian update walks those only
      unsigned flipmask[CPL];
#pragma unroll
      for (int sl = 0; sl < CPL; sl++) {
        bool flipped = false;
        if (iscon[sl]) {
          float f[4];
          unsigned bits = 0;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const bool on = jar[sl][r] < 0.0f;
            f[r] = on ? -cD[sl] * jar[sl][r] : 0.0f;
            bits |= on ? (1u << r) : 0u;
          }
          // w = new flags | previous flags << 4, as an exactly representable small float
          stv(CFB(S, lane + G * sl), f4{f[0] + f[1] + f[2] + f[3], cmu[sl] * (f[0] - f[1]), cmu[sl] * (f[2] - f[3]), (float)(bits | (prevbits[sl] << 4))});
          flipped = bits != (it == 0 ? 15u : prevbits[sl]);  // (the Hessian starts from all rows active)
          prevbits[sl] = bits;
        }
        flipmask[sl] = (unsigned)(__ballot(flipped) >> (tid & 48)) & 0xffffu;
      }
      WSYNC();
      if constexpr (CPL > 1) {  // (the helper wave starts on its shares of this iteration; of the first one behind barrier (4))
        if (use_help && it > 0 && tid == 0) __hip_atomic_store(&s_help[0], it + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      if (it == 0) STAMP(16);
      ITSTAMP(it, 0);
      // ---- gradient first (cheap): convergence is decided before any Hessian work
      float g = isdof ? Ma - qfs - lsg * lf : 0.0f;
      const int ncon_g = (CPL > 1 && use_help && it > 0 && ncon > G) ? G : ncon;  // (three contacts per lane, from the second iteration on: the contacts above 16 are the helper wave's)
      for (int c0 = 0; c0 < ncon_g; c0 += 4) {  // four contacts per trip: one batch of reads, then the sums in contact order
        float jn[4], j1[4], j2[4];
        f4 fb[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int c = c0 + u < ncon_g ? c0 + u : c0;
          const float* jb = JBROW(S, c);
          jn[u] = jb[lane]; j1[u] = jb[16 + lane]; j2[u] = jb[32 + lane];
          fb[u] = ldv(CFB(S, c));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; u++)
          if (c0 + u < ncon_g) g -= jn[u] * fb[u].x + j1[u] * fb[u].y + j2[u] * fb[u].z;
      }
      if constexpr (CPL > 1) {
        if (use_help && it > 0) {  // (the helper wave's share: exact zero for an env with at most 16 contacts)
          while (__hip_atomic_load(&s_help[1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < it + 1) __builtin_amdgcn_s_sleep(1);
          g += S.M[lane][G];
        }
      }
      if (!isdof) g = 0.0f;
      if (it == 0) STAMP(17);
      ITSTAMP(it, 1);
      if (term_bound && !term_sent) {
        // ---- the host-visible terminated bytes, as soon as they cannot change.  The cost f is 1-strongly convex in the Mt norm, so the
        // iterate is within |g|_{Mt^-1} of the minimiser, and so is the FINAL iterate, however the solver stops -- gradient rule,
        // rounding floor or iteration cap -- because every accepted step lowers the cost (enforced where the step is taken, further
        // down in this loop: a step whose exact 1-D model does not, is not taken): f(a_F) - f* <= f(a_k) - f* <= 1/2 |g_k|^2.  The object's vertical acceleration is therefore
        // within 2 / sqrt(mass) |g|_{Mt^-1} of its final value, and |g|^2_{Mt^-1} / mass <= sum_i d_gw_i g_i^2.  Where no contact joins
        // the arm and the object the problem separates, a step is accepted tree by tree and each tree's own cost is monotone: the same
        // argument holds for the object's block with the object's share of the gradient alone (the arm's limit rows are what keeps
        // a straggler iterating; with the warm start chosen per tree 99.5 % of the workgroups qualify at the first gradient this way).
        // The height the current iterate predicts (same two fused multiply-adds as the integrator below) must be farther from the
        // threshold than dt^2 times that bound -- with (1 + sqrt 2) for 2, doubled again, plus 1 m/s^2, plus 1e-5 m: the margin that
        // float32 evaluation of g, of the 1-D model and of the integrator could consume -- for all four envs of the wave; otherwise the
        // bytes wait for the next iteration, or for the integrator.  A cube at rest on the floor qualifies at the first gradient.
        // The bytes are checked against the integrated state all the same (below): a difference raises the sticky word `term_bad`.
        const float wg = d_gw * g * g;
        const float sall = gsum(wg), sobj = gsum(lane >= mdl_split ? wg : 0.0f);
        const float gm = sqrtf(sep ? sobj : sall);
        const float az = lane_gather(row4 + (term_zlane << 2), qacc);
        const float zp = S.qpos[mdl_obj_qadr + 2] + dt * (S.qvel[term_zlane] + dt * az);
        const float slack = 2.0f * dt * dt * (2.4142137f * gm + 1.0f) + 1e-5f;
        const bool decided = !valid || ovf_env || fabsf(zp - mdl_reward_z) > slack;  // (a deferred env's byte says so, whatever its height)
        if (!__any(!decided)) {
          const unsigned long long tb = __ballot(valid && !ovf_env && above(zp, mdl_reward_z) && lane == 0), db = __ballot(ovf_env && valid && lane == 0);
          const unsigned long long ob = BIGV ? __ballot(over_env && valid && lane == 0) : 0ull;  // (three contacts per lane: bit 6, as in the bytes behind the integrator)
          term_bits = (uint32_t)(tb & 1u) | (uint32_t)(tb >> 16 & 1u) << 8 | (uint32_t)(tb >> 32 & 1u) << 16 | (uint32_t)(tb >> 48 & 1u) << 24 |
                      (uint32_t)(db & 1u) << 7 | (uint32_t)(db >> 16 & 1u) << 15 | (uint32_t)(db >> 32 & 1u) << 23 | (uint32_t)(db >> 48 & 1u) << 31 |
                      (uint32_t)(ob & 1u) << 6 | (uint32_t)(ob >> 16 & 1u) << 14 | (uint32_t)(ob >> 32 & 1u) << 22 | (uint32_t)(ob >> 48 & 1u) << 30;
          if (tid == 0)
            __hip_atomic_store(reinterpret_cast<uint32_t*>(a.term_host) + (size_t)blockIdx.x * a.term_wstride, term_bits | (a.term_tag << 1) * 0x01010101u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
          term_sent = true;
          TERMSTAMP();
          if (a.early_stats && tid == 0) atomicAdd(a.early_stats + 2 + blockIdx.x, 1u);  // (this workgroup's own counter: no contention, no return value)
          STAMP(30);
        }
      }
      const float gn = sqrtf(gsum_u(g * g)), gw = sqrtf(gsum_u(gfw * g * g));
      if (!done && (scale * gn < tol || gw < gfl)) done = true;
      if (it == 0) STAMP(14);
      ITSTAMP(it, 2);
      if (!__any(!done)) break;
      if (it == 0) {  // start from Mt + the all-rows-active J^T D J (from the collision wave where there is one)
        float hp[G];
        if (DUAL && (!post_now || ROT)) {
          STAMP(51);
          __syncthreads();  // (4)
          STAMP(53);
          met4 = true;
          if constexpr (CPL > 1) { if (use_help && tid == 0) __hip_atomic_store(&s_help[0], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const f4 v = ldv(&S.M[lane][4 * q]);
            hp[4 * q] = v.x; hp[4 * q + 1] = v.y; hp[4 * q + 2] = v.z; hp[4 * q + 3] = v.w;
          }
        } else {
          hess_full(hp, ncon);
        }
#pragma unroll
        for (int j = 0; j < G; j++) hkeep[j] += hp[j];
      }
      // ---- Hessian row (lane = dof): incremental update of H = Mt + J^T D_active J; per contact the
      // change enters through the 3x3 weight of its pyramid in (n, t1, t2) coordinates, which is
      // linear in the per-row activity, so flipped rows contribute +-D and unchanged contacts nothing
      if (__any(lact != oldlact)) {  // (a joint-limit row switched somewhere in the wave: rare)
#pragma unroll
        for (int j = 0; j < G; j++) hkeep[j] += j == lane ? lact - oldlact : 0.0f;
      }
      oldlact = lact;
      if constexpr (CPL == 1) {
#pragma unroll
      for (int sl = 0; sl < CPL; sl++)
      for (unsigned fm = flipmask[sl]; fm; fm &= fm - 1u) {  // (group-uniform trip count; contact order: slot by slot)
        const int c = __ffs(fm) - 1 + G * sl;
        const float* jb = JBROW(S, c);
        // every read of this contact in one batch, before any arithmetic (one LDS round trip)
        const f4 fb = ldv(CFB(S, c));
        const float jn = jb[lane], j1 = jb[16 + lane], j2 = jb[32 + lane];
        const f4 mt = ldv(S.con.cmeta[c]);
        f4 xn[4], x1[4], x2[4];
#pragma unroll
        for (int q = 0; q < 4; q++) { xn[q] = ldv(jb + 4 * q); x1[q] = ldv(jb + 16 + 4 * q); x2[q] = ldv(jb + 32 + 4 * q); }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned both = (unsigned)fb.w;
        const unsigned bits = both & 15u, old = it == 0 ? 15u : both >> 4;  // (first iteration: relative to all rows active)
        const float mu = mt.x, D = mt.y;
        const float a0 = D * (float)((int)(bits & 1u) - (int)(old & 1u)), a1 = D * (float)((int)(bits >> 1 & 1u) - (int)(old >> 1 & 1u));
        const float a2 = D * (float)((int)(bits >> 2 & 1u) - (int)(old >> 2 & 1u)), a3 = D * (float)((int)(bits >> 3 & 1u) - (int)(old >> 3 & 1u));
        const float w0 = a0 + a1 + a2 + a3, w1 = mu * (a0 - a1), w2 = mu * (a2 - a3), w3 = mu * mu * (a0 + a1), w4 = mu * mu * (a2 + a3);
        const float tn = jn * w0 + j1 * w1 + j2 * w2, t1 = jn * w1 + j1 * w3, t2 = jn * w2 + j2 * w4;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          hkeep[4 * q + 0] += tn * xn[q].x + t1 * x1[q].x + t2 * x2[q].x;
          hkeep[4 * q + 1] += tn * xn[q].y + t1 * x1[q].y + t2 * x2[q].y;
          hkeep[4 * q + 2] += tn * xn[q].z + t1 * x1[q].z + t2 * x2[q].z;
          hkeep[4 * q + 3] += tn * xn[q].w + t1 * x1[q].w + t2 * x2[q].w;
        }
      }
      } else {
        // (three contacts per lane: this wave takes the flipped contacts of the first slot, the helper wave those of the others --
        //  its rows are added behind this wave's own, exact zeros for an env with at most 16 contacts)
        for (unsigned fm = flipmask[0]; fm; fm &= fm - 1u) hess_flip(hkeep, __ffs(fm) - 1, it == 0);
        if (use_help) {
          while (__hip_atomic_load(&s_help[2], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < it + 1) __builtin_amdgcn_s_sleep(1);
          const float* hr = help_hrow();
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const f4 v = ldv(hr + 4 * q);
            hkeep[4 * q] += v.x; hkeep[4 * q + 1] += v.y; hkeep[4 * q + 2] += v.z; hkeep[4 * q + 3] += v.w;
          }
        }
      }
      float hrow[G];
#pragma unroll
      for (int j = 0; j < G; j++) hrow[j] = hkeep[j];
      if (it == 0) STAMP(18);
      ITSTAMP(it, 3);
      // ---- Newton direction: H s = -g
      float sv = -g;
      gj_solve(hrow, sv, lane, hsplit, nv);
      if (!isdof) sv = 0.0f;
      if (it == 0) STAMP(15);
      ITSTAMP(it, 4);
      // (the direction stays in the lanes: M s and J s take s_j by row broadcast, no LDS round trip)
      const float mvb = rowdot_bc(mrow, sv);
      const float mv = isdof ? mvb : 0.0f;
      const float ljv = lsg * sv;
      float jv[CPL][4];
#pragma unroll
      for (int sl = 0; sl < CPL; sl++) {
#pragma unroll
        for (int r = 0; r < 4; r++) jv[sl][r] = 0.0f;
        if (sl > 0 && !__any(iscon[sl])) continue;  // (wave-uniform)
        float xn, x1, x2;
        jd3(sl, sv, xn, x1, x2);  // (every lane of the row takes part in the broadcasts)
        if (iscon[sl]) { jv[sl][0] = xn + cmu[sl] * x1; jv[sl][1] = xn - cmu[sl] * x1; jv[sl][2] = xn + cmu[sl] * x2; jv[sl][3] = xn - cmu[sl] * x2; }
      }
      if (it == 0) STAMP(19);
      ITSTAMP(it, 5);
      // ---- exact line search on the piecewise-quadratic phi(alpha): safeguarded Newton on phi'
      // phi'(0) is the gradient along the direction, g . s, and with the exact Hessian the first Newton iterate on phi' is
      // alpha = 1: the evaluation at alpha = 0 (a full pass over the rows and two reductions) is not spent; the search starts
      // at 1 with the bracket [0, ?) and phi'(0) = g . s as the scale of its stopping rule
      const float svmv = sv * mv, svb = sv * (Ma - qfs);
      const float A = gsum_u(svmv), Bq = gsum_u(svb), g0 = gsum_u(sv * g);
      bool lsdone = done || g0 >= 0.0f;
      float alpha = lsdone ? 0.0f : 1.0f, lo = 0.0f, hi = -1.0f;
      // improvement of a step alpha s from the 1-D model (exact: phi is piecewise quadratic) and the number of rows whose sign it
      // changes.  Row-cost differences are formed as 1/2 D d (2 x0 + d) with d = alpha jv, never as a difference of squares: a
      // step below the resolution of jar must yield a (correctly) tiny improvement, not an absorbed one
      // (with a = min(x, 0) the cost of a row is 1/2 D a^2, and its change 1/2 D (a1 - a0)(a1 + a0); a1 - a0 is the step d
      //  itself while the row stays active)
      float pimc[CPL], piml = 0.0f, crsc[CPL], crsl = 0.0f;  // this lane's share at the last evaluation: contact rows (per slot) / limit row
#pragma unroll
      for (int sl = 0; sl < CPL; sl++) { pimc[sl] = 0.0f; crsc[sl] = 0.0f; }
      auto step_gain = [&](float al, float& gain, float& ncr) __attribute__((always_inline)) {
#pragma unroll
        for (int sl = 0; sl < CPL; sl++) step_rows_c(al, jar[sl], jv[sl], cD[sl], pimc[sl], crsc[sl]);
        float pc = pimc[0], cc = crsc[0];
#pragma unroll
        for (int sl = 1; sl < CPL; sl++) { pc += pimc[sl]; cc += crsc[sl]; }
        step_rows_l(al, ljar, ljv, lD, lsg, piml, crsl);
        gain = gsum_u(pc + piml) - (0.5f * al * al * A + al * Bq);
        ncr = gsum(cc + crsl);  // (the two reductions are independent and overlap)
      };
      // The full Newton step first.  Where it crosses no row boundary it IS the minimiser along s; where it does, it is taken as it
      // is when it realises at least a quarter of the decrease the quadratic piece at alpha = 0 predicts for it (-g0 / 2): the
      // search below then has nothing to do for that env, and a wave whose envs all accept skips it.  (The fixed point is the
      // same minimiser; the oracle keeps the exact search, and the parity tests hold the two together.)
      float improvement, ncross;
      step_gain(alpha, improvement, ncross);
      const float alpha0 = alpha;
      lsdone = lsdone || improvement > -0.125f * g0;
      for (int ls = 1; ls < mdl_ls_iterations && __any(!lsdone); ls++) {  // (ls counts evaluations of phi', the one at 0 included)
        float pg = 0.0f, ph = 0.0f, pa = 0.0f;
#pragma unroll
        for (int sl = 0; sl < CPL; sl++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const float x = jar[sl][r] + alpha * jv[sl][r];
          if (x < 0.0f) { pg += cD[sl] * jv[sl][r] * x; ph += cD[sl] * jv[sl][r] * jv[sl][r]; }
        }
        {
          const float x = ljar + alpha * ljv;
          if (x < 0.0f) { pg += lD * ljv * x; ph += lD * ljv * ljv; }
        }
        const float gg = gsum_u(pg) + alpha * A + Bq, hh = gsum_u(ph) + A;
        // From the fifth evaluation on (rare) also the magnitude of the
        // terms phi' is summed from: at the root they cancel and what is left is rounding noise of about an epsilon of
        // that magnitude, which no evaluation can resolve (same rule as the oracle)
        float floorg = 0.0f;
        if (ls >= 4) {  // wave-uniform
#pragma unroll
          for (int sl = 0; sl < CPL; sl++)
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (jar[sl][r] + alpha * jv[sl][r] < 0.0f) pa += cD[sl] * fabsf(jv[sl][r]) * (fabsf(jar[sl][r]) + fabsf(alpha * jv[sl][r]));
          if (ljar + alpha * ljv < 0.0f) pa += lD * fabsf(ljv) * (fabsf(ljar) + fabsf(alpha * ljv));
          floorg = 4.0f * 1.1920929e-7f * (gsum_u(pa) + fabsf(alpha * A) + fabsf(Bq));
        }
        if (!lsdone) {
          if (fabsf(gg) <= fmaxf(1e-6f * fabsf(g0), floorg)) lsdone = true;
          if (!lsdone) {
            if (gg < 0.0f) lo = alpha; else hi = alpha;
            float an = alpha - gg / hh;
            if (hi >= 0.0f && (an <= lo || an >= hi)) an = 0.5f * (lo + hi);
            if (an == alpha) lsdone = true;
            if (!lsdone) alpha = an;
          }
        }
      }
      if (it == 0) STAMP(20);
      ITSTAMP(it, 6);
      // ---- improvement of the step the search settled on (envs that took the full step have theirs already), then the update
      if (__any(alpha != alpha0)) {
        float gi, gc;
        step_gain(alpha, gi, gc);
        if (alpha != alpha0) { improvement = gi; ncross = gc; }
      }
      // ---- acceptance, TREE BY TREE.  A tree moves only if the step lowers ITS OWN cost: where no contact joins the two trees the
      // cost is a sum f_A(a_A) + f_B(a_B), the common step length minimises the sum along s and may raise one of the terms; that tree
      // then stays where it is for this iteration (the other one's gain is all the larger).  The cost of every tree -- of the whole
      // problem where it does not separate -- is so non-increasing over the iterations BY CONSTRUCTION, whatever made the search
      // stop: the property the early `terminated` bytes are derived from (above).
      //   * no row changes sides along the step: every tree's cost is ONE quadratic along its own Newton direction, s_T . H_T s_T =
      //     -g_T . s_T, and falls by -g_T . s_T (alpha - alpha^2 / 2) > 0 at the alpha = 1 the search then returns: nothing to check;
      //   * some row does: tree B's share of the exact 1-D model -- three masked reductions over the lane shares of the last
      //     evaluation, still in registers -- and tree A's = the rest, each accepted on its own sign.
      float ald = alpha, alc[CPL];  // the step length this lane's dof / its contact(s) take
#pragma unroll
      for (int sl = 0; sl < CPL; sl++) alc[sl] = alpha;
      const bool need = sep && ncross != 0.0f && alpha != 0.0f;
      bool partial = false;  // one tree moves, the other was held back: the env is not finished whatever the moving tree's gain says
      if (__any(need)) {  // (wave-uniform; an env's result does not depend on its neighbours: only `need` envs use the sums)
        const float AB = gsum_u(dofB ? svmv : 0.0f), BB = gsum_u(dofB ? svb : 0.0f);
        float pB = conB[0] ? pimc[0] : 0.0f;
#pragma unroll
        for (int sl = 1; sl < CPL; sl++) pB += conB[sl] ? pimc[sl] : 0.0f;
        const float gainB = gsum_u(pB + (dofB ? piml : 0.0f)) - (0.5f * alpha * alpha * AB + alpha * BB);
        const float gainA = improvement - gainB;
        const bool okA = gainA > 0.0f, okB = gainB > 0.0f;
        if (need && !(okA && okB)) {  // (rare: an exact search on the sum usually lowers both terms)
          ald = (dofB ? okB : okA) ? alpha : 0.0f;
#pragma unroll
          for (int sl = 0; sl < CPL; sl++) alc[sl] = (conB[sl] ? okB : okA) ? alpha : 0.0f;
          improvement = (okA ? gainA : 0.0f) + (okB ? gainB : 0.0f);
          partial = okA != okB;
        }
      }
      if (!need && !(improvement > 0.0f)) {  // (one tree, or a crossing-free step at the rounding floor of the model)
        ald = 0.0f;
#pragma unroll
        for (int sl = 0; sl < CPL; sl++) alc[sl] = 0.0f;
        improvement = 0.0f;
      }
      // float32 resolution: if no dof's acceleration changes, or the gradient has stopped shrinking
      // within a few floors of its rounding level, further iterations are noise
      const float moved = gsum((isdof && qacc + ald * sv != qacc) ? 1.0f : 0.0f);
      const bool stagnant = it > 0 && gw > 0.5f * gprev && gw < 4.0f * gfl;
      gprev = gw;
      // (an env with a tree held back goes on: with the other tree out of the way -- converged, or closer -- the common step length
      //  becomes the held-back tree's own)
      if (!done && !partial && (moved == 0.0f || stagnant)) { done = true; niter = it + 1; }
      if (!done) {
        qacc += ald * sv;
        Ma += ald * mv;
        ljar += ald * ljv;
#pragma unroll
        for (int sl = 0; sl < CPL; sl++)
#pragma unroll
        for (int r = 0; r < 4; r++) jar[sl][r] += alc[sl] * jv[sl][r];
        niter = it + 1;
        if (!partial && scale * improvement < tol) done = true;
      }
      {
        // if the step crossed no row boundary, phi is one quadratic along it and the new gradient is
        // exactly (1 - alpha) g: decide convergence now instead of paying another gradient pass
        const float gnew = fabsf(1.0f - alpha) * gn, gwnew = fabsf(1.0f - alpha) * gw;
        if (!done && ncross == 0.0f && (scale * gnew < tol || gwnew < gfl)) done = true;
      }
#ifdef MIR_DEBUG_TRACE
      /* developer aid (make EXTRA=-DMIR_DEBUG_TRACE; tools/solver_trace.py): the solver's per-lane state at the end of every Newton
       * iteration of env prof[255], as floats behind the 256 stamp slots of the buffer given to mir_debug_profile_step */
      if (a.prof && valid && env == (int)a.prof[255] && it < 8) {
        float* tr = reinterpret_cast<float*>(a.prof + 256) + (it * G + lane) * 16;
        tr[0] = qacc; tr[1] = jar[0][0]; tr[2] = jar[0][1]; tr[3] = jar[0][2]; tr[4] = jar[0][3]; tr[5] = (float)prevbits[0]; tr[6] = g; tr[7] = sv;
        tr[8] = alpha; tr[9] = ald; tr[10] = alc[0]; tr[11] = improvement; tr[12] = ncross; tr[13] = done ? 1.0f : 0.0f; tr[14] = (float)flipmask[0]; tr[15] = partial ? 1.0f : 0.0f;
      }
#endif
      WSYNC();
      if (it == 0) STAMP(21);
      ITSTAMP(it, 7);
    }
    if (DUAL && (!post_now || ROT) && !met4) __syncthreads();  // (4) (no Hessian was needed: the collision wave is let go)
    if constexpr (CPL > 1) { if (tid == 0) __hip_atomic_store(&s_help[0], -1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }  // (the helper wave leaves the Newton loop)
    if (a.out_qacc && valid && isdof && step == 0) a.out_qacc[(size_t)env * nv + lane] = qacc;
    if (a.diag && valid && !ovf_env && lane == 0) {
      a.diag[(size_t)env * 4 + 0] = ncon;
      a.diag[(size_t)env * 4 + 1] = nefc;
      a.diag[(size_t)env * 4 + 2] = niter;
      a.diag[(size_t)env * 4 + 3] = S.ncand | (cpl >> 20 & 255) << 8;
    }
    STAMP(8);
    if (a.mode != 0) return 1;

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
      else if (d_kind == 2) S.qpos[T.b_info[d_body][2] + d_axis_k] += dt * qd;
    }
    if (VARIANT != 1 && a.term_host && term_early) {
      // GenesisEnv.step's D->H copy of `terminated`, done by the kernel: see the epilogue; here ~1 us earlier, so that the trip
      // over PCIe is over when the launch ends
      WSYNC();
      const bool tn = valid && !ovf_env && above(S.qpos[mdl_obj_qadr + 2], mdl_reward_z);
      const unsigned long long tb = __ballot(tn && lane == 0), db = __ballot(ovf_env && valid && lane == 0), ob = BIGV ? __ballot(over_env && valid && lane == 0) : 0ull;
      if (tid == 0) {
        const uint32_t bits = (uint32_t)(tb & 1u) | (uint32_t)(tb >> 16 & 1u) << 8 | (uint32_t)(tb >> 32 & 1u) << 16 | (uint32_t)(tb >> 48 & 1u) << 24 |
                              (uint32_t)(db & 1u) << 7 | (uint32_t)(db >> 16 & 1u) << 15 | (uint32_t)(db >> 32 & 1u) << 23 | (uint32_t)(db >> 48 & 1u) << 31 |
                              (uint32_t)(ob & 1u) << 6 | (uint32_t)(ob >> 16 & 1u) << 14 | (uint32_t)(ob >> 32 & 1u) << 22 | (uint32_t)(ob >> 48 & 1u) << 30;
        // (bytes that left from inside the solver loop are checked against the integrated state: a difference would mean the bound
        //  was violated -- it is counted, mir_debug_early_mask_stats, and the right bytes are stored over the wrong ones)
        if (!term_sent || bits != term_bits)
          __hip_atomic_store(reinterpret_cast<uint32_t*>(a.term_host) + (size_t)blockIdx.x * a.term_wstride, bits | (a.term_tag << 1) * 0x01010101u, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
        if (term_sent && bits != term_bits) {
          if (a.early_stats) atomicAdd(a.early_stats + 1, 1u);
          if (a.term_bad) __hip_atomic_store(a.term_bad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // sticky: the next API call fails with MIR_E_MASK
        }
      }
      if (!term_sent) { STAMP(30); TERMSTAMP(); }
    }
    if (fksplit) {
      WSYNC();
      __syncthreads();  // (5) jointed dofs integrated: the collision wave starts the closing FK of the jointed bodies
    }
    if (isdof) {
      if (d_kind == 3 && d_axis_k == 0) {
        const int da = T.b_info[d_body][3], qa = T.b_info[d_body][2];
        V3 w = v3(S.qvel[da + 3], S.qvel[da + 4], S.qvel[da + 5]);
        float wn = sqrtf(dot(w, w));
        float ang = wn * dt;
        if (ang > 1e-15f) {
          float sn, cs;
          sincos_pi2(0.5f * ang, &sn, &cs);
          V3 ax = (1.0f / wn) * w;
          Q4 dq = {cs, ax.x * sn, ax.y * sn, ax.z * sn};
          st4(&S.qpos[qa + 3], qnormalize(qmul(dq, ld4(&S.qpos[qa + 3]))));
        }
      }
    }
    WSYNC();
    STAMP(9);
    if (a.diag) {
      // divergence guard (diagnostics on): an env whose integrated state holds a NaN or an Inf is flagged in its
      // diagnostics record -- bit 30 of word 3, sticky over the steps of a rollout launch -- and counted; its `terminated` is False
      // (mir_dev.h: above()).  The other envs of the wave are not touched by it: every reduction and gather stays inside an env's row.
      // (only the words of the rows that are ever written: a scene with nq <= 12 keeps stale LDS in S.qpos[qst .. 15] -- ADVICE r4)
      const bool nf = (lane < qst && nonfinite(S.qpos[lane])) || (lane + G < qst && nonfinite(S.qpos[lane + G])) || nonfinite(S.qvel[lane]);
      const bool bad = ((uint32_t)(__ballot(nf) >> (grp * G)) & 0xffffu) != 0u;
      bad_acc = bad_acc || bad;
      if (valid && !ovf_env && lane == 0) {
        a.diag[(size_t)env * 4 + 3] = S.ncand | (cpl >> 20 & 255) << 8 | (bad_acc ? 1 << 30 : 0);
        if (bad && a.early_stats) atomicAdd(a.early_stats, 1u);  // (word 0: env-steps that ended non-finite, since the last reset of the counters)
      }
    }
    // kinematics of the new state: observations of this step, and the next step's starting poses
    if (fksplit) {
      // (the free bodies' poses are their qpos rows -- the expressions of group_fk for a childless child of the world)
      if (isbody && bk.jtype == MIR_JNT_FREE) {
        st3v(S.xpos[lane], ld3(&S.qpos[bk.qadr]));
        st4v(S.xquat[lane], qnormalize(ld4(&S.qpos[bk.qadr + 3])));
      }
      WSYNC();
      __syncthreads();  // (6) the jointed bodies' poses from the collision wave
    } else {
      group_fk(S, lane, nb, parents, bk, row4);
    }
    if (a.rows && a.rows_step && (step + 1 < nsteps || a.ar.episode_len) && valid) {  // rollout mode: one packed row per env per step
      float* row = a.rows + (size_t)step * a.rows_step + (size_t)env * a.row_stride;
      for (int c = lane; c < ad + 13; c += G) row[c] = column(c);
    }
    if (a.ar.episode_len) {
      // episode bookkeeping and re-spawn on chip (the rules of k_autoreset): the row above is the terminal observation
      const bool term = above(S.xpos[ob][2], mdl_reward_z);
      const int len = eplen + 1;
      const bool trunc = !term && a.ar.max_len > 0 && len >= a.ar.max_len;
      const bool done = term || trunc;
      if (valid && lane == 0 && a.rows && a.row_stride > ad + 13)
        a.rows[(size_t)step * a.rows_step + (size_t)env * a.row_stride + ad + 13] = trunc ? 1.0f : 0.0f;
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
        for (int c = lane; c < 7 * nfree; c += G) {
          const int k = c / 7, j = c - 7 * k;
          S.qpos[m->free_qadr[k] + j] = j < 3 ? sp[k * 3 + j] : a.ar.obj_quat[((size_t)env * nfree + k) * 4 + (j - 3)];
        }
        epcur += 1;
      }
      WSYNC();
      if (__any(done)) group_fk(S, lane, nb, parents, bk, row4);
    }
    if ((ROT || BIG2) && step == 0) emit_outputs();
    return 0;
  };  // step_body
  if constexpr (FIRSTONLY) {
    step_body(std::integral_constant<int, 0>{});  // (the outputs leave at its end)
  } else if constexpr (ROT || BIG2) {
    if (step_body(std::integral_constant<int, 0>{}) == 0) step_body(std::integral_constant<int, 1>{});
  } else if constexpr (SINGLE) {
    if (step_body(std::integral_constant<int, 0>{}) == 2) return;
    emit_outputs();
  } else {
    for (int step = 0; step < nsteps; step++) {
      const int r = step_body(step);
      if (r == 2) return;
      if (r == 1) break;
    }
    emit_outputs();
  }
}

}  // namespace

#ifdef MIR_STEP_CONVEX_TU
// debug aid: the lane-private convex narrowphase on n pairs given directly, one thread per pair.
//   in  (n, 22): type1, size1[3], pos1[3], quat1[4] (wxyz), type2, size2[3], pos2[3], quat2[4]
//   out (n, 8):  hit (0/1), pos[3], dist, normal[3]
namespace {
__global__ void k_debug_convex(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = in + (size_t)i * 22;
  const M3 R1 = q2m(qnormalize(Q4{r[7], r[8], r[9], r[10]})), R2 = q2m(qnormalize(Q4{r[18], r[19], r[20], r[21]}));
  const ShapeD A = {(int)r[0], v3(r[1], r[2], r[3]), v3(r[4], r[5], r[6]), mcol(R1, 0), mcol(R1, 1), mcol(R1, 2), nullptr, 0};
  const ShapeD B = {(int)r[11], v3(r[12], r[13], r[14]), v3(r[15], r[16], r[17]), mcol(R2, 0), mcol(R2, 1), mcol(R2, 2), nullptr, 0};
  f4 pt = {0, 0, 0, 0};
  V3 nrm = v3(0, 0, 0);
  const bool hit = convex_pair(A, B, pt, nrm);
  float* o = out + (size_t)i * 8;
  o[0] = hit ? 1.0f : 0.0f; o[1] = pt.x; o[2] = pt.y; o[3] = pt.z; o[4] = pt.w; o[5] = nrm.x; o[6] = nrm.y; o[7] = nrm.z;
}
}  // namespace
extern "C" int mir_launch_debug_convex(const float* in, float* out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(k_debug_convex, dim3((n + 63) / 64), dim3(64), 0, stream, in, out, n);
  return (int)hipGetLastError();
}

#endif

// launcher used by the C ABI (mir_api.hip).  The instantiations with the convex narrowphase live in their own translation unit
// (mir_step_convex.hip = this file compiled with MIR_STEP_CONVEX_TU and WITHOUT -fno-signed-zeros: together with
// -ffp-contract=on that flag miscompiles the support-mapping selects of mir_convex.h -- box pairs lose contacts -- while the
// planes-and-boxes kernels gain 1 % from it).
#ifdef MIR_STEP_CONVEX_TU
template <int FEAT>
static void launch_feat(const StepArgs& a, int blocks, int single, int plain_loop, hipStream_t stream) {
  if (a.phase == 1) hipLaunchKernelGGL((mir_step_kernel<3, FEAT>), dim3(blocks), dim3(128), 0, stream, a);
  else if (a.phase == 3) hipLaunchKernelGGL((mir_step_kernel<5, FEAT>), dim3(blocks), dim3(128), 0, stream, a);
  else if (a.phase == 4) hipLaunchKernelGGL((mir_step_kernel<6, FEAT, 3>), dim3(blocks), dim3(128), 0, stream, a);
  else if (a.phase == 5) hipLaunchKernelGGL((mir_step_kernel<7, FEAT, 3>), dim3(blocks), dim3(128), 0, stream, a);
  else if (a.phase == 6) hipLaunchKernelGGL((mir_step_kernel<9, FEAT, 3>), dim3(blocks), dim3(128), 0, stream, a);
  else if (a.phase == 7) hipLaunchKernelGGL((mir_step_kernel<10, FEAT, 3>), dim3(blocks), dim3(128), 0, stream, a);
  else if (a.phase == 8) hipLaunchKernelGGL((mir_step_kernel<11, FEAT>), dim3(blocks), dim3(128), 0, stream, a);
  else if (single) hipLaunchKernelGGL((mir_step_kernel<0, FEAT>), dim3(blocks), dim3(128), 0, stream, a);
  else if (plain_loop) hipLaunchKernelGGL((mir_step_kernel<1, FEAT>), dim3(blocks), dim3(64), 0, stream, a);
  else if constexpr ((FEAT & 4) == 0) hipLaunchKernelGGL((mir_step_kernel<2, FEAT>), dim3(blocks), dim3(64), 0, stream, a);
}
extern "C" __attribute__((visibility("hidden"))) int mir_launch_step_convex(const StepArgs* args, int single, int plain_loop, hipStream_t stream) {
  StepArgs a = *args;
  const int blocks = (a.B + EPB - 1) / EPB;
  // the headline scene's instantiation (features bit 2: mir_create found SpecPick::matches); the everything-variant stays generic
  if ((a.features & 4) && (a.phase == 1 || a.phase == 3 || a.phase == 4 || a.phase == 5 || a.phase == 6 || a.phase == 7 || a.phase == 8 || single || plain_loop)) launch_feat<5>(a, blocks, single, plain_loop, stream);
  else if (a.features & 2) launch_feat<3>(a, blocks, single, plain_loop, stream);  // sweep-and-prune scenes carry the convex code too
  else launch_feat<1>(a, blocks, single, plain_loop, stream);
  return (int)hipGetLastError();
}
#else
extern "C" __attribute__((visibility("hidden"))) int mir_launch_step_convex(const StepArgs* args, int single, int plain_loop, hipStream_t stream);
extern "C" int mir_launch_step(const StepArgs* args, int max_contacts_lds, hipStream_t stream) {
  StepArgs a = *args;
  int blocks = (a.B + EPB - 1) / EPB;
  (void)max_contacts_lds;
#ifdef MIR_PROFILE_SINGLE
  const bool prof_blocks_single = false;
#else
  const bool prof_blocks_single = a.prof != nullptr;
#endif
  const bool single = a.mode == 0 && a.n_steps == 1 && !a.act_step && !a.rows_step && !a.ar.episode_len && !prof_blocks_single && !a.out_M && !a.out_bias &&
                      !a.out_qas && !a.out_qacc && !a.out_xpos && !a.out_xquat;
  const bool plain_loop = a.mode == 0 && !a.prof && !a.out_M && !a.out_bias && !a.out_qas && !a.out_qacc && !a.out_xpos && !a.out_xquat && !a.agent_pos &&
                          !a.env_state && !a.reward && !a.terminated && !a.term_host && !a.done_ticket;
  if (a.phase == 2) {  // (the second half of a split step has no collision code in it: one instantiation serves every scene)
    hipLaunchKernelGGL((mir_step_kernel<4, 0>), dim3(blocks), dim3(64), 0, stream, a);
    return (int)hipGetLastError();
  }
  if (a.features) return mir_launch_step_convex(&a, single, plain_loop, stream);
  if (a.phase == 1) hipLaunchKernelGGL((mir_step_kernel<3, 0>), dim3(blocks), dim3(128), 0, stream, a);
  else if (a.phase == 3) hipLaunchKernelGGL((mir_step_kernel<5, 0>), dim3(blocks), dim3(128), 0, stream, a);
  else if (a.phase == 4) hipLaunchKernelGGL((mir_step_kernel<6, 0, 3>), dim3(blocks), dim3(128), 0, stream, a);
  else if (a.phase == 5) hipLaunchKernelGGL((mir_step_kernel<7, 0, 3>), dim3(blocks), dim3(128), 0, stream, a);
  else if (a.phase == 6) hipLaunchKernelGGL((mir_step_kernel<9, 0, 3>), dim3(blocks), dim3(128), 0, stream, a);
  else if (a.phase == 7) hipLaunchKernelGGL((mir_step_kernel<10, 0, 3>), dim3(blocks), dim3(128), 0, stream, a);
  else if (a.phase == 8) hipLaunchKernelGGL((mir_step_kernel<11, 0>), dim3(blocks), dim3(128), 0, stream, a);
  else if (single) hipLaunchKernelGGL((mir_step_kernel<0, 0>), dim3(blocks), dim3(128), 0, stream, a);
  else if (plain_loop) hipLaunchKernelGGL((mir_step_kernel<1, 0>), dim3(blocks), dim3(64), 0, stream, a);
  else hipLaunchKernelGGL((mir_step_kernel<2, 0>), dim3(blocks), dim3(64), 0, stream, a);
  return (int)hipGetLastError();
}
#endif

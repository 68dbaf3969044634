// mir_dev.h — device-side helpers shared by the two step kernels (mir_step.hip: 16 lanes per env;
// mir_step64.hip: one wave per env): small vector / quaternion algebra, DPP primitives over one 16-lane row,
// the register-row Gauss-Jordan solve of a 16-wide block, spatial inertia products, the box-box narrowphase and
// the soft-constraint impedance.  Everything is in an anonymous namespace (one copy per translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mirigid.h"

#ifndef G
#define G 16 /* lanes per DPP row = width of a dof block */
#endif

// wave-level phase separator: LDS operations of one wave execute in order, so only the compiler
// must be kept from moving LDS accesses across the phase boundary
#define WSYNC()                                              \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

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
// 16-byte LDS accesses (arrays below are padded to 4 floats per element for these)
__device__ __forceinline__ f4 ldv(const float* p) { return *reinterpret_cast<const f4*>(p); }
__device__ __forceinline__ void stv(float* p, f4 v) { *reinterpret_cast<f4*>(p) = v; }
__device__ __forceinline__ V3 ld3v(const float* p) { f4 v = ldv(p); return {v.x, v.y, v.z}; }
__device__ __forceinline__ void st3v(float* p, V3 a, float w = 0.0f) { stv(p, f4{a.x, a.y, a.z, w}); }

// sin and cos of one argument without the library routine's large-argument path: Cody-Waite reduction by pi/2 (exact for the
// |x| < 1e3 rad that joint angles and one-step rotation angles can reach), Cephes single-precision kernels on
// [-pi/4, pi/4] (< 1 ulp there), quadrant fix-up by selects.  ~25 straight-line instructions in the FK chain of every step.
__device__ __forceinline__ void sincos_pi2(float x, float* sn, float* cs) {
  const float k = rintf(x * 0.63661977236758134308f);
  float r = fmaf(-k, 1.57079625129699707031f, x);
  r = fmaf(-k, 7.54978941586159635335e-08f, r);
  r = fmaf(-k, 5.39030252995776476554e-15f, r);
  const float z = r * r;
  const float s = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float c = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z, fmaf(-0.5f, z, 1.0f));
  const int q = (int)k & 3;
  const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
  *sn = (q & 2) ? -ss : ss;
  *cs = ((q + 1) & 2) ? -cc : cc;
}

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
__device__ __forceinline__ Q4 ld4v(const float* p) { f4 v = ldv(p); return {v.x, v.y, v.z, v.w}; }
__device__ __forceinline__ void st4(float* p, Q4 q) { p[0] = q.w; p[1] = q.x; p[2] = q.y; p[3] = q.z; }
__device__ __forceinline__ void st4v(float* p, Q4 q) { stv(p, f4{q.w, q.x, q.y, q.z}); }
__device__ __forceinline__ Q4 qnormalize(Q4 q) {
  const float n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  if (n2 < 1e-30f) return {1, 0, 0, 0};
  const float s = __builtin_amdgcn_rsqf(n2);  // one v_rsq_f32 (1 ulp) instead of sqrt + reciprocal
  return {q.w * s, q.x * s, q.y * s, q.z * s};
}
struct M3 {
  V3 r0, r1, r2;  // rows
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

// ---- DPP cross-lane primitives over one 16-lane row (= one env group).  Must be executed with all
// lanes of the wave active (convergent code): an inactive source lane would feed garbage.
template <int N>
__device__ __forceinline__ float row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
template <int K>
__device__ __forceinline__ float row_bcast(float v) {  // value of lane K of the row, in every lane of the row
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + K, 0xf, 0xf, false));
}
template <int N>
__device__ __forceinline__ float row_shr(float v) {  // lane i reads lane i-N of its row, 0 shifted in
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x110 + N, 0xf, 0xf, true));
}
template <int N>
__device__ __forceinline__ float row_shl(float v) {  // lane i reads lane i+N of its row, 0 shifted in
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + N, 0xf, 0xf, true));
}
// the task's threshold test on a height (reward / terminated): true for a FINITE z above the threshold.  A state that has gone
// non-finite (a NaN target from the caller, a diverged env) never terminates an episode: NaN fails the first comparison, +Inf the second.
__device__ __forceinline__ bool above(float z, float thr) { return z > thr && z <= 3.0e38f; }
__device__ __forceinline__ bool nonfinite(float x) { return (__float_as_uint(x) & 0x7f800000u) == 0x7f800000u; }
// lane gather through the LDS crossbar (ds_bpermute_b32: no LDS memory, no write-then-read fence): every lane names the wave
// lane it reads from as a BYTE offset (4 x lane)
__device__ __forceinline__ float lane_gather(int src4, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src4, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int lane_gather(int src4, int v) { return __builtin_amdgcn_ds_bpermute(src4, v); }
// All-reduce sum over the row, THE SAME VALUE IN EVERY LANE, bit for bit.  Every stage pairs the lanes by an involution -- the
// row's mirror image, the mirror image of each half, then the neighbours inside a quad and the pairs of a quad -- so that the two
// lanes of a pair add the same two numbers (a + b and b + a: floating-point addition commutes), and by induction all sixteen lanes
// end with the same sum.  The rotation butterfly this replaces (four `row_ror` adds) gave every lane the sum in an association of
// its own: two lanes of a row could hold sums that differ in the last bit -- harmless in arithmetic, fatal in a DECISION that every
// lane of the env takes for itself.  The line search of an env stopped one evaluation earlier on lanes 1 and 9 than on the others
// (|phi'| one ulp on either side of its tolerance), two dofs and one contact took a different step length, and the next gradient
// sent a resting cube spinning: 3e-4 rad in one step, 2 envs of 1024 on the reference expert's second step (found by the
// state-parity check of tests/test_gpu_exact_contacts.py; tools/solver_trace.py shows it lane by lane).  Same four DPP adds as before.
template <int CTRL>
__device__ __forceinline__ float row_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float gsum(float v) {
  v += row_dpp<0x140>(v);  // row_mirror: lane i <-> 15 - i
  v += row_dpp<0x141>(v);  // row_half_mirror: lane i <-> 7 - i inside each half
  v += row_dpp<0xB1>(v);   // quad_perm [1, 0, 3, 2]: lane i <-> i ^ 1
  v += row_dpp<0x4E>(v);   // quad_perm [2, 3, 0, 1]: lane i <-> i ^ 2
  return v;
}
__device__ __forceinline__ float gsum_u(float v) { return gsum(v); }  // (the name the decision-feeding sums were given when gsum was not uniform)
__device__ __forceinline__ float gmaxf(float v) {  // all-reduce max over the row
  v = fmaxf(v, row_ror<1>(v));
  v = fmaxf(v, row_ror<2>(v));
  v = fmaxf(v, row_ror<4>(v));
  v = fmaxf(v, row_ror<8>(v));
  return v;
}
__device__ __forceinline__ int gsumi(int v) { return (int)(gsum((float)v) + 0.5f); }

#include "mir_gj_dpp.h"

// ---- Gauss-Jordan solve A x = b on register rows: lane i holds row i of the SPD matrix A in
// a[0..15] and b_i in b; on return b = x_i.  Rows >= nv must be identity rows.  15 pivots, each:
// one reciprocal, (16-k) row_newbcast + fma pairs.  No pivoting needed (SPD).
template <int K>
struct GJ {
  static __device__ __forceinline__ void run(float (&a)[G], float& b, int lane) {
    const float pk = row_bcast<K>(a[K]);
    float inv = __builtin_amdgcn_rcpf(pk);
    // (v_rcp_f32 is good to 1 ulp; a Newton step here sat in the dependent chain of every pivot)
    // one fma per column for every row: with f = 1 - 1/p on the pivot row (whose broadcast entry is its own) and
    // a_iK / p elsewhere, a[j] - f * pivotrow[j] scales the pivot row and eliminates the others
    const float f = lane == K ? 1.0f - inv : a[K] * inv;
#pragma unroll
    for (int j = K + 1; j < G - 1; j++) a[j] = fmaf(-f, row_bcast<K>(a[j]), a[j]);
    b = fmaf(-f, row_bcast<K>(b), b);
    GJ<K + 1>::run(a, b, lane);
  }
};
template <>
struct GJ<G - 1> {
  static __device__ __forceinline__ void run(float (&)[G], float&, int) {}
};

// The same elimination for a matrix that is block diagonal with blocks [0, S) and [S, 15): a pivot of the first block only
// touches the columns of the first block (the rest of its row is zero), so those fmas are left out -- for every row, since
// f * 0 changes nothing.  Bit-identical to GJ<0> on such a matrix.  (The mass matrix is block diagonal by kinematic tree --
// arm and cube -- and so is the Newton Hessian as long as no contact joins the two trees.)
template <int S, int K>
struct GJ2 {
  static __device__ __forceinline__ void run(float (&a)[G], float& b, int lane) {
    const float pk = row_bcast<K>(a[K]);
    const float inv = __builtin_amdgcn_rcpf(pk);
    const float f = lane == K ? 1.0f - inv : a[K] * inv;
    constexpr int END = K < S ? S : G - 1;
#pragma unroll
    for (int j = K + 1; j < END; j++) a[j] = fmaf(-f, row_bcast<K>(a[j]), a[j]);
    b = fmaf(-f, row_bcast<K>(b), b);
    GJ2<S, K + 1>::run(a, b, lane);
  }
};
template <int S>
struct GJ2<S, G - 1> {
  static __device__ __forceinline__ void run(float (&)[G], float&, int) {}
};
// The same two blocks eliminated SIDE BY SIDE: step K pivots on row K of the first block (columns < S) and on row S + K of the
// second (columns S .. E-1) at once.  A row of one block holds zeros in the other block's columns, so its factor for the other
// block's pivot is zero and every entry sees exactly the operations GJ2 applies to it, in the same order: bit-identical to
// GJ2<S> (and to GJ<0>) on such a matrix, in max(S, E - S) dependent steps instead of E.  Rows >= E are identity rows.
template <int S, int E, int K>
struct GJP {
  static __device__ __forceinline__ void run(float (&a)[G], float& b, int lane) {
    constexpr bool HA = K < S, HB = S + K < E;
    constexpr int KA = HA ? K : 0, KB = HB ? S + K : 0;
    float fA = 0.0f, fB = 0.0f;
    if (HA) {
      const float inv = __builtin_amdgcn_rcpf(row_bcast<KA>(a[KA]));
      fA = lane == KA ? 1.0f - inv : a[KA] * inv;
    }
    if (HB) {
      const float inv = __builtin_amdgcn_rcpf(row_bcast<KB>(a[KB]));
      fB = lane == KB ? 1.0f - inv : a[KB] * inv;
    }
    if (HA) {
#pragma unroll
      for (int j = KA + 1; j < S; j++) a[j] = fmaf(-fA, row_bcast<KA>(a[j]), a[j]);
    }
    if (HB) {
#pragma unroll
      for (int j = KB + 1; j < E; j++) a[j] = fmaf(-fB, row_bcast<KB>(a[j]), a[j]);
    }
    if (HA) b = fmaf(-fA, row_bcast<KA>(b), b);
    if (HB) b = fmaf(-fB, row_bcast<KB>(b), b);
    if constexpr (K + 1 < (S > E - S ? S : E - S)) GJP<S, E, K + 1>::run(a, b, lane);
  }
};
// GJP<9, 15> / GJP<6, 12> with the column updates as v_fmac_f32_dpp blocks (mir_gj_dpp.h): the same operations on every entry in
// the same order, one instruction per entry instead of a v_mov_b32_dpp + v_fma_f32 pair -- bit-identical results.
#define MIR_GJPD(S, E)                                                                                              \
  template <int K>                                                                                                  \
  __device__ __forceinline__ void gjpd_##S##_##E(float (&a)[G], float& b, int lane) {                               \
    constexpr bool HA = K < S, HB = S + K < E;                                                                      \
    constexpr int KA = HA ? K : 0, KB = HB ? S + K : 0;                                                             \
    float nfa = 0.0f, nfb = 0.0f;                                                                                   \
    if (HA) {                                                                                                       \
      const float inv = __builtin_amdgcn_rcpf(row_bcast<KA>(a[KA]));                                                \
      nfa = lane == KA ? inv - 1.0f : -(a[KA] * inv);                                                               \
    }                                                                                                               \
    if (HB) {                                                                                                       \
      const float inv = __builtin_amdgcn_rcpf(row_bcast<KB>(a[KB]));                                                \
      nfb = lane == KB ? inv - 1.0f : -(a[KB] * inv);                                                               \
    }                                                                                                               \
    gjp_dpp_step_##S##_##E<K>(a, b, nfa, nfb);                                                                      \
    if constexpr (K + 1 < (S > E - S ? S : E - S)) gjpd_##S##_##E<K + 1>(a, b, lane);                               \
  }
MIR_GJPD(9, 15)
MIR_GJPD(6, 12)
#undef MIR_GJPD
// dispatch on the model's block split (0 = dense) and dof count; both must be wave-uniform
__device__ __forceinline__ void gj_solve(float (&a)[G], float& b, int lane, int split, int nv) {
  if (split == 9 && nv == 15) { gjpd_9_15<0>(a, b, lane); return; }
  if (split == 6 && nv == 12) { gjpd_6_12<0>(a, b, lane); return; }
  if (split == 9) GJ2<9, 0>::run(a, b, lane);
  else if (split == 6) GJ2<6, 0>::run(a, b, lane);
  else GJ<0>::run(a, b, lane);
}

// spatial inertia {m, h, I(xx yy zz xy xz yz)} applied to motion {w, v} -> force {t, f}
struct Inert {
  float m;
  V3 h;
  float xx, yy, zz, xy, xz, yz;
};
__device__ __forceinline__ Inert ldI(const float* p) {
  f4 a = ldv(p), b = ldv(p + 4), c = ldv(p + 8);
  return {a.x, {a.y, a.z, a.w}, b.x, b.y, b.z, b.w, c.x, c.y};
}
__device__ __forceinline__ void imul(const Inert& I, V3 w, V3 v, V3& t, V3& f) {
  V3 Iw = {I.xx * w.x + I.xy * w.y + I.xz * w.z, I.xy * w.x + I.yy * w.y + I.yz * w.z, I.xz * w.x + I.yz * w.y + I.zz * w.z};
  t = Iw + cross(I.h, v);
  f = I.m * v - cross(I.h, w);
}

// ---------------------------------------------------------------------------------------------
// narrowphase primitives (one DPP row per candidate pair)
struct BoxG {
  V3 p;
  V3 a0, a1, a2;  // world axes
  V3 h;
};
// (every field is read into a value first and the selects work on those: a conditional whose arms read the fields is
// turned into a select between ADDRESSES, which forces the box into private memory)
__device__ __forceinline__ V3 bax(const BoxG& b, int k) {
  const float x0 = b.a0.x, y0 = b.a0.y, z0 = b.a0.z, x1 = b.a1.x, y1 = b.a1.y, z1 = b.a1.z, x2 = b.a2.x, y2 = b.a2.y, z2 = b.a2.z;
  const bool k0 = k == 0, k1 = k == 1;
  return v3(k0 ? x0 : (k1 ? x1 : x2), k0 ? y0 : (k1 ? y1 : y2), k0 ? z0 : (k1 ? z1 : z2));
}
__device__ __forceinline__ float bh(const BoxG& b, int k) {
  const float hx = b.h.x, hy = b.h.y, hz = b.h.z;
  return k == 0 ? hx : (k == 1 ? hy : hz);
}

// Sutherland-Hodgman clipping of the incident face against the four edges of the reference face, on a DPP row: lane v
// holds vertex v of the polygon (reference-face coordinates x, y and height z; 4 vertices on entry, at most 8 later).
// Per edge every lane emits its vertex if it is inside and the edge crossing towards its successor, at the slots an
// exclusive prefix sum over the row assigns (the order of the sequential algorithm); the new polygon travels through
// ex (3 x 16 floats of LDS owned by this row).  Returns the vertex count (row-uniform); vertices in x, y, z again.
__device__ __forceinline__ int clip_row(float& x, float& y, float& z, float h1, float h2, int l16, int wlane, float* ex) {
  int np = 4;
  const int rowbase = wlane & ~15;
#pragma unroll 1
  for (int e = 0; e < 4; e++) {
    const int ax = e >> 1;
    const float sg = (e & 1) ? -1.0f : 1.0f;
    const float lim = ax == 0 ? h1 : h2;
    // an edge that cuts nothing leaves the polygon as it is (same vertices, same order): skipped without the exchange
    if ((((unsigned)(__ballot(l16 < np && sg * (ax == 0 ? x : y) - lim > 0.0f) >> (rowbase & 63)) & 0xffffu)) == 0u) continue;
    const int succ = rowbase + (l16 + 1 >= np ? 0 : l16 + 1);
    const float xs = __shfl(x, succ), ys = __shfl(y, succ), zs = __shfl(z, succ);
    const float pc = ax == 0 ? x : y, qc = ax == 0 ? xs : ys;
    const float dp = sg * pc - lim, dq = sg * qc - lim;
    const bool act = l16 < np;
    const bool keep = act && dp <= 0.0f, crs = act && ((dp <= 0.0f) != (dq <= 0.0f));
    const float mine = (keep ? 1.0f : 0.0f) + (crs ? 1.0f : 0.0f);
    float incl = mine;
    incl += row_shr<1>(incl);
    incl += row_shr<2>(incl);
    incl += row_shr<4>(incl);
    incl += row_shr<8>(incl);
    const int off = (int)(incl - mine);
    np = (int)row_bcast<15>(incl);
    if (np == 0) return 0;  // row-uniform
    if (keep) { ex[off] = x; ex[16 + off] = y; ex[32 + off] = z; }
    if (crs) {
      const float u = dp / (dp - dq);
      const int o2 = off + (keep ? 1 : 0);
      ex[o2] = x + u * (xs - x); ex[16 + o2] = y + u * (ys - y); ex[32 + o2] = z + u * (zs - z);
    }
    WSYNC();
    x = ex[l16]; y = ex[16 + l16]; z = ex[32 + l16];
    WSYNC();
  }
  return np;
}

// Box-box on a whole DPP row: every lane of the row calls this with the same pair.  The 15 separating axes sit on
// lanes 0..14; the axis choice replays the sequential rule of box_box on row broadcasts, so both routines pick the same
// feature; the four incident-face vertices sit on lanes 0..3, and a partially overlapping face pair is clipped on the
// row as well (clip_row; ex = 48 floats of LDS owned by this row).  Returns the contact count in every lane of the row;
// points go to out[0..cnt), the normal (A to B) to snorm.  l16 = lane in the row, wlane = lane in the wave,
// rowshift = bit position of the row in a wave ballot.  Same algorithm as oracle/orc_rigid.c (sequential there).
__device__ __forceinline__ V3 selv(bool c, V3 a, V3 b) { return v3(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z); }
__device__ __forceinline__ int box_box_row(const BoxG& A, const BoxG& B, int l16, int wlane, int rowshift, float (*out)[4], float* snorm, float* ex) {
  const V3 t = B.p - A.p;
  // ---- lane = axis: faces of A (0..2), faces of B (3..5), edge pairs i x j (6 + 3 i + j)
  const int ax = l16;
  const int ei = ax < 9 ? 0 : (ax < 12 ? 1 : 2), ej = ax - 6 - 3 * ei;
  const V3 Lf = selv(ax < 3, bax(A, ax), bax(B, ax - 3));
  V3 Le = cross(bax(A, ei), bax(B, ej));
  const float len = sqrtf(dot(Le, Le));
  Le = (1.0f / len) * Le;
  const bool isedge = ax >= 6;
  const bool deg = ax >= 15 || (isedge && len < 1e-3f);
  const V3 L = selv(isedge, Le, Lf);
  const float ra = A.h.x * fabsf(dot(A.a0, L)) + A.h.y * fabsf(dot(A.a1, L)) + A.h.z * fabsf(dot(A.a2, L));
  const float rb = B.h.x * fabsf(dot(B.a0, L)) + B.h.y * fabsf(dot(B.a1, L)) + B.h.z * fabsf(dot(B.a2, L));
  const float s = deg ? -3e38f : fabsf(dot(t, L)) - (ra + rb);
  const unsigned sepm = (unsigned)(__ballot(s > 0.0f) >> rowshift) & 0xffffu;
  int cnt = 0;
  if (sepm == 0u) {  // row-uniform
    float best = -1e30f;
    int code = -1;
#define MIR_FACE(c) { const float sc = row_bcast<c>(s); if (sc > best) { best = sc; code = c; } }
    MIR_FACE(0) MIR_FACE(1) MIR_FACE(2) MIR_FACE(3) MIR_FACE(4) MIR_FACE(5)
#undef MIR_FACE
#define MIR_EDGE(c) { const float sc = row_bcast<c>(s); if (sc > -1e38f && sc * 1.05f > best && sc > best + 1e-6f) { best = sc; code = c; } }
    MIR_EDGE(6) MIR_EDGE(7) MIR_EDGE(8) MIR_EDGE(9) MIR_EDGE(10) MIR_EDGE(11) MIR_EDGE(12) MIR_EDGE(13) MIR_EDGE(14)
#undef MIR_EDGE
    const int src = (wlane & ~15) + code;
    const V3 bestL = v3(__shfl(L.x, src), __shfl(L.y, src), __shfl(L.z, src));
    const V3 n = selv(dot(t, bestL) < 0.0f, -1.0f * bestL, bestL);
    if (l16 == 0) st3v(snorm, n);
    if (code >= 6) {
      // edge-edge: closest points of the two supporting edges (every lane of the row computes, lane 0 writes)
      const int i = (code - 6) / 3, j = (code - 6) % 3;
      V3 PA = A.p, PB = B.p;
  #pragma unroll
      for (int q = 0; q < 3; q++) {
        if (q != i) PA = PA + (dot(n, bax(A, q)) > 0.0f ? bh(A, q) : -bh(A, q)) * bax(A, q);
        if (q != j) PB = PB + (dot(n, bax(B, q)) > 0.0f ? -bh(B, q) : bh(B, q)) * bax(B, q);
      }
      const V3 ua = bax(A, i), ub = bax(B, j), dd = PB - PA;
      const float uaub = dot(ua, ub), q1 = dot(ua, dd), q2 = -dot(ub, dd), den = 1.0f - uaub * uaub;
      float alpha = 0.0f, beta = 0.0f;
      if (den > 1e-6f) { alpha = (q1 + uaub * q2) / den; beta = (uaub * q1 + q2) / den; }
      PA = PA + alpha * ua;
      PB = PB + beta * ub;
      const V3 pos = 0.5f * (PA + PB);
      if (l16 == 0) stv(out[0], f4{pos.x, pos.y, pos.z, best});
      cnt = 1;
    } else {
      // face contact: reference face = the chosen axis, incident face = the most anti-parallel face of the other box
      const bool refA = code < 3;
      const BoxG R = {selv(refA, A.p, B.p), selv(refA, A.a0, B.a0), selv(refA, A.a1, B.a1), selv(refA, A.a2, B.a2), selv(refA, A.h, B.h)};
      const BoxG I = {selv(refA, B.p, A.p), selv(refA, B.a0, A.a0), selv(refA, B.a1, A.a1), selv(refA, B.a2, A.a2), selv(refA, B.h, A.h)};
      const int kf = refA ? code : code - 3;
      const V3 nr = selv(refA, n, -1.0f * n);
      const int k1 = (kf + 1) % 3, k2 = (kf + 2) % 3;
      const V3 fc = R.p + bh(R, kf) * nr;
      const float a0 = fabsf(dot(nr, I.a0)), a1 = fabsf(dot(nr, I.a1)), a2 = fabsf(dot(nr, I.a2));
      int jb = 0;
      float mx = a0;
      if (a1 > mx) { mx = a1; jb = 1; }
      if (a2 > mx) { mx = a2; jb = 2; }
      const float sj = dot(nr, bax(I, jb)) > 0.0f ? -1.0f : 1.0f;
      const int j1 = (jb + 1) % 3, j2 = (jb + 2) % 3;
      const V3 ic = I.p + (sj * bh(I, jb)) * bax(I, jb);
      const V3 e1 = bax(R, k1), e2 = bax(R, k2);
      const float h1 = bh(R, k1), h2 = bh(R, k2);
      // lane = incident-face vertex (0..3, same order as box_box: (+,+), (-,+), (-,-), (+,-))
      const int v = l16 & 3;
      const float sxv = (v == 0 || v == 3) ? 1.0f : -1.0f, syv = v < 2 ? 1.0f : -1.0f;
      const V3 w = ic + (sxv * bh(I, j1)) * bax(I, j1) + (syv * bh(I, j2)) * bax(I, j2);
      const V3 rel = w - fc;
      const float vx = dot(rel, e1), vy = dot(rel, e2), vz = dot(rel, nr);
      const bool in_v = fabsf(vx) <= h1 && fabsf(vy) <= h2;
      const unsigned insm = (unsigned)(__ballot(in_v) >> rowshift) & 0xfu;
      if (insm == 0xfu) {
        // the whole incident face lies over the reference face (a cube resting on the slab): the clipped polygon
        // is the face itself, so its penetrating vertices are emitted in order
        const bool pen = l16 < 4 && vz < 0.0f;
        const unsigned penm = (unsigned)(__ballot(pen) >> rowshift) & 0xfu;
        cnt = __popc(penm);
        if (pen) {
          const V3 wp = fc + vx * e1 + vy * e2 + (0.5f * vz) * nr;
          stv(out[__popc(penm & ((1u << l16) - 1u))], f4{wp.x, wp.y, wp.z, vz});
        }
      } else {
        // partial overlap: clip the incident face on the row, then emit the penetrating vertices (at most 8) in order
        float px = vx, py = vy, pz = vz;
        const int np = clip_row(px, py, pz, h1, h2, l16, wlane, ex);
        const bool pen = l16 < np && pz < 0.0f;
        const unsigned penm = (unsigned)(__ballot(pen) >> rowshift) & 0xffffu;
        const int slot = __popc(penm & ((1u << l16) - 1u));
        cnt = min(__popc(penm), 8);
        if (pen && slot < 8) {
          const V3 wp = fc + px * e1 + py * e2 + (0.5f * pz) * nr;
          stv(out[slot], f4{wp.x, wp.y, wp.z, pz});
        }
      }
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

// dot of a register row with a 16-float LDS vector (4 broadcast b128 reads)
__device__ __forceinline__ float rowdot(const float (&r)[G], const float* x) {
  f4 x0 = ldv(x), x1 = ldv(x + 4), x2 = ldv(x + 8), x3 = ldv(x + 12);
  return r[0] * x0.x + r[1] * x0.y + r[2] * x0.z + r[3] * x0.w + r[4] * x1.x + r[5] * x1.y + r[6] * x1.z + r[7] * x1.w +
         r[8] * x2.x + r[9] * x2.y + r[10] * x2.z + r[11] * x2.w + r[12] * x3.x + r[13] * x3.y + r[14] * x3.z + r[15] * x3.w;
}
// The same two products with the vector spread over the lanes of the DPP row (lane j holds x_j) instead of in LDS: every
// term takes its x_j by row broadcast inside the fma (v_fmac_f32_dpp), so the vector needs no LDS round trip.  Four
// independent partial sums keep the dependent chain at four fmas.  Convergent code only (all lanes of the row active).
__device__ __forceinline__ float rowdot_bc(const float (&r)[G], float x) {
  float a0 = r[0] * row_bcast<0>(x), a1 = r[1] * row_bcast<1>(x), a2 = r[2] * row_bcast<2>(x), a3 = r[3] * row_bcast<3>(x);
  a0 = fmaf(r[4], row_bcast<4>(x), a0); a1 = fmaf(r[5], row_bcast<5>(x), a1); a2 = fmaf(r[6], row_bcast<6>(x), a2); a3 = fmaf(r[7], row_bcast<7>(x), a3);
  a0 = fmaf(r[8], row_bcast<8>(x), a0); a1 = fmaf(r[9], row_bcast<9>(x), a1); a2 = fmaf(r[10], row_bcast<10>(x), a2); a3 = fmaf(r[11], row_bcast<11>(x), a3);
  a0 = fmaf(r[12], row_bcast<12>(x), a0); a1 = fmaf(r[13], row_bcast<13>(x), a1); a2 = fmaf(r[14], row_bcast<14>(x), a2); a3 = fmaf(r[15], row_bcast<15>(x), a3);
  return (a0 + a1) + (a2 + a3);
}
struct JRow {  // the three base rows (normal, t1, t2) of one contact, kept in registers by the contact's lane
  f4 a[4], b[4], c[4];
};
__device__ __forceinline__ JRow jrow_load(const float* jb) {
  JRow r;
#pragma unroll
  for (int q = 0; q < 4; q++) { r.a[q] = ldv(jb + 4 * q); r.b[q] = ldv(jb + 16 + 4 * q); r.c[q] = ldv(jb + 32 + 4 * q); }
  return r;
}
__device__ __forceinline__ void jdot3_bc(const JRow& jr, float x, float& dn, float& d1, float& d2) {
  const f4 (&a)[4] = jr.a;
  const f4 (&b)[4] = jr.b;
  const f4 (&c)[4] = jr.c;
  const float x0 = row_bcast<0>(x), x1 = row_bcast<1>(x), x2 = row_bcast<2>(x), x3 = row_bcast<3>(x), x4 = row_bcast<4>(x), x5 = row_bcast<5>(x),
              x6 = row_bcast<6>(x), x7 = row_bcast<7>(x), x8 = row_bcast<8>(x), x9 = row_bcast<9>(x), x10 = row_bcast<10>(x), x11 = row_bcast<11>(x),
              x12 = row_bcast<12>(x), x13 = row_bcast<13>(x), x14 = row_bcast<14>(x), x15 = row_bcast<15>(x);
  dn = ((a[0].x * x0 + a[0].y * x1) + (a[0].z * x2 + a[0].w * x3)) + ((a[1].x * x4 + a[1].y * x5) + (a[1].z * x6 + a[1].w * x7)) +
       (((a[2].x * x8 + a[2].y * x9) + (a[2].z * x10 + a[2].w * x11)) + ((a[3].x * x12 + a[3].y * x13) + (a[3].z * x14 + a[3].w * x15)));
  d1 = ((b[0].x * x0 + b[0].y * x1) + (b[0].z * x2 + b[0].w * x3)) + ((b[1].x * x4 + b[1].y * x5) + (b[1].z * x6 + b[1].w * x7)) +
       (((b[2].x * x8 + b[2].y * x9) + (b[2].z * x10 + b[2].w * x11)) + ((b[3].x * x12 + b[3].y * x13) + (b[3].z * x14 + b[3].w * x15)));
  d2 = ((c[0].x * x0 + c[0].y * x1) + (c[0].z * x2 + c[0].w * x3)) + ((c[1].x * x4 + c[1].y * x5) + (c[1].z * x6 + c[1].w * x7)) +
       (((c[2].x * x8 + c[2].y * x9) + (c[2].z * x10 + c[2].w * x11)) + ((c[3].x * x12 + c[3].y * x13) + (c[3].z * x14 + c[3].w * x15)));
}
// the three base-row dots (normal, t1, t2) of contact Jacobian block `jb` with a 16-float LDS vector
__device__ __forceinline__ void jdot3(const float* jb, const float* x, float& dn, float& d1, float& d2) {
  dn = d1 = d2 = 0.0f;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    f4 xv = ldv(x + 4 * q);
    f4 a = ldv(jb + 4 * q), b = ldv(jb + 16 + 4 * q), c = ldv(jb + 32 + 4 * q);
    dn += a.x * xv.x + a.y * xv.y + a.z * xv.z + a.w * xv.w;
    d1 += b.x * xv.x + b.y * xv.y + b.z * xv.z + b.w * xv.w;
    d2 += c.x * xv.x + c.y * xv.y + c.z * xv.z + c.w * xv.w;
  }
}

}  // namespace

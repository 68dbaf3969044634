// mir_convex.h — convex-convex narrowphase of the 16-lane step kernel, LANE-PRIVATE: lane c of an env group runs the whole
// algorithm for candidate pair c in its own registers (up to 16 pairs of an env side by side, 4 envs per wave; lanes diverge on
// the iteration counts and reconverge at the end).  No LDS, no cross-lane traffic.
//
//   * spheres and capsules are a core (point / segment) swept by a radius: GJK distance between the CORES (point, segment,
//     box: polytopes, so GJK ends on an exact feature pair) gives depth = r1 + r2 - distance, the normal along the connecting
//     line and the position midway between the two surfaces -- exact to rounding;
//   * when the cores themselves overlap (penetration deeper than the radii): Minkowski Portal Refinement on the full shapes
//     (the XenoCollide / libccd formulation; Genesis's default convex-convex path).
//
// Same decisions in the same order as the oracle's double-precision version (oracle/orc_rigid.c: gjk_core_distance, mpr_pair,
// convex_pair), which the tests hold it against.  Register economy: a simplex / portal vertex is kept as (point of A - B,
// support direction); the witness points on A and B are recomputed from the direction at the end (supports are deterministic).
#pragma once
#ifndef MIR_CONVEX_STANDALONE /* (tests/convex_host.cpp compiles this header for the host with its own vector helpers) */
#include "mir_dev.h"
#endif

namespace {

struct ShapeD {
  int type;
  V3 size, pos, a0, a1, a2;  // a* = columns of the rotation (the geom's axes in the world)
  const float* verts;        // MIR_GEOM_HULL: the hull's vertices (rows of 4 floats, geom frame; LDS in the kernel), else unused
  int nvert;
};

// hull: the vertex of largest projection on d, lowest index on ties (the oracle's hull_support), in world coordinates
__device__ __forceinline__ V3 hull_support(const ShapeD& s, V3 d) {
  const V3 dl = v3(dot(d, s.a0), dot(d, s.a1), dot(d, s.a2));
  V3 best = v3(s.verts[0], s.verts[1], s.verts[2]);
  float bv = dot(best, dl);
  for (int i = 1; i < s.nvert; i++) {
    const V3 v = v3(s.verts[4 * i], s.verts[4 * i + 1], s.verts[4 * i + 2]);
    const float pv = dot(v, dl);
    if (pv > bv) { bv = pv; best = v; }
  }
  return s.pos + best.x * s.a0 + best.y * s.a1 + best.z * s.a2;
}

__device__ __forceinline__ V3 core_support(const ShapeD& s, V3 d) {  // farthest point of the CORE along d (any length)
  if (s.type == MIR_GEOM_HULL) return hull_support(s, d);  // (a hull is its own core, radius 0)
  V3 o = s.pos;
  if (s.type == MIR_GEOM_CAPSULE) {
    o = o + (dot(d, s.a2) >= 0.0f ? s.size.y : -s.size.y) * s.a2;
  } else if (s.type == MIR_GEOM_BOX) {
    o = o + (dot(d, s.a0) >= 0.0f ? s.size.x : -s.size.x) * s.a0;
    o = o + (dot(d, s.a1) >= 0.0f ? s.size.y : -s.size.y) * s.a1;
    o = o + (dot(d, s.a2) >= 0.0f ? s.size.z : -s.size.z) * s.a2;
  }
  return o;
}
__device__ __forceinline__ V3 shape_support(const ShapeD& s, V3 d) {  // farthest point of the shape along the UNIT direction d
  if (s.type == MIR_GEOM_HULL) return hull_support(s, d);
  V3 o = s.pos;
  if (s.type == MIR_GEOM_SPHERE) {
    o = o + s.size.x * d;
  } else if (s.type == MIR_GEOM_CAPSULE) {
    o = o + (dot(d, s.a2) >= 0.0f ? s.size.y : -s.size.y) * s.a2;
    o = o + s.size.x * d;
  } else {
    o = o + (dot(d, s.a0) >= 0.0f ? s.size.x : -s.size.x) * s.a0;
    o = o + (dot(d, s.a1) >= 0.0f ? s.size.y : -s.size.y) * s.a1;
    o = o + (dot(d, s.a2) >= 0.0f ? s.size.z : -s.size.z) * s.a2;
  }
  return o;
}
__device__ __forceinline__ float core_radius(const ShapeD& s) { return (s.type == MIR_GEOM_BOX || s.type == MIR_GEOM_HULL) ? 0.0f : s.size.x; }
__device__ __forceinline__ V3 neg(V3 a) { return {-a.x, -a.y, -a.z}; }
__device__ __forceinline__ bool unit(V3& d) {  // normalise in place; false if (numerically) zero
  const float l2 = dot(d, d);
  if (!(l2 > 1e-30f)) return false;
  d = (1.0f / sqrtf(l2)) * d;
  return true;
}

// closest point of the segment / triangle to the origin as barycentric weights (Ericson, Real-Time Collision Detection 5.1)
__device__ __forceinline__ void seg_bary(V3 a, V3 b, float* l) {
  const V3 ab = b - a;
  const float t = -dot(a, ab), den = dot(ab, ab);
  if (t <= 0.0f || !(den > 0.0f)) { l[0] = 1.0f; l[1] = 0.0f; }
  else if (t >= den) { l[0] = 0.0f; l[1] = 1.0f; }
  else { l[1] = t / den; l[0] = 1.0f - l[1]; }
}
__device__ __forceinline__ void tri_bary(V3 a, V3 b, V3 c, float* l) {
  const V3 ab = b - a, ac = c - a;
  const float d1 = -dot(ab, a), d2 = -dot(ac, a);
  l[0] = l[1] = l[2] = 0.0f;
  if (d1 <= 0.0f && d2 <= 0.0f) { l[0] = 1.0f; return; }
  const float d3 = -dot(ab, b), d4 = -dot(ac, b);
  if (d3 >= 0.0f && d4 <= d3) { l[1] = 1.0f; return; }
  const float vc = d1 * d4 - d3 * d2;
  if (vc <= 0.0f && d1 >= 0.0f && d3 <= 0.0f) { l[1] = d1 / (d1 - d3); l[0] = 1.0f - l[1]; return; }
  const float d5 = -dot(ab, c), d6 = -dot(ac, c);
  if (d6 >= 0.0f && d5 <= d6) { l[2] = 1.0f; return; }
  const float vb = d5 * d2 - d1 * d6;
  if (vb <= 0.0f && d2 >= 0.0f && d6 <= 0.0f) { l[2] = d2 / (d2 - d6); l[0] = 1.0f - l[2]; return; }
  const float va = d3 * d6 - d5 * d4;
  if (va <= 0.0f && (d4 - d3) >= 0.0f && (d5 - d6) >= 0.0f) { l[2] = (d4 - d3) / ((d4 - d3) + (d5 - d6)); l[1] = 1.0f - l[2]; return; }
  const float den = 1.0f / (va + vb + vc);
  l[1] = vb * den; l[2] = vc * den; l[0] = 1.0f - l[1] - l[2];
}

// GJK distance between the cores.  false: cores apart, dist > 0, pa / pb = closest points on core A / core B; true: they touch
// or overlap.
__device__ __forceinline__ bool gjk_core_distance(const ShapeD& A, const ShapeD& B, float& dist, V3& pa, V3& pb) {
  V3 P[4], D[4];  // simplex vertices of A - B and the direction each was obtained with (w = supA(-D) - supB(D))
  float lam[4] = {1.0f, 0.0f, 0.0f, 0.0f};
  int n = 0;
  V3 v = A.pos - B.pos;
  if (!(dot(v, v) > 1e-24f)) v = v3(1.0f, 0.0f, 0.0f);
  for (int it = 0; it < 32; it++) {
    const V3 w = core_support(A, neg(v)) - core_support(B, v);
    const float vv = dot(v, v);
    if (n > 0 && vv - dot(v, w) <= 1e-12f * vv) break;  // no point of A - B is closer along v
    bool dup = false;
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (i < n) { const V3 e = w - P[i]; if (!(dot(e, e) > 1e-24f)) dup = true; }
    if (dup) break;
    // (static indexing keeps the simplex in registers)
    if (n == 0) { P[0] = w; D[0] = v; } else if (n == 1) { P[1] = w; D[1] = v; } else if (n == 2) { P[2] = w; D[2] = v; } else { P[3] = w; D[3] = v; }
    n++;
    float l[4] = {1.0f, 0.0f, 0.0f, 0.0f};
    if (n == 2) seg_bary(P[0], P[1], l);
    else if (n == 3) tri_bary(P[0], P[1], P[2], l);
    else if (n == 4) {
      // inside iff on the inner side of all four faces; otherwise the closest of the faces the origin is outside of
      // (faces in the fixed order 012, 013, 023, 123)
      float best = -1.0f, bl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
      bool inside = true;
#pragma unroll
      for (int f = 0; f < 4; f++) {
        const int i0 = f == 3 ? 1 : 0, i1 = f < 2 ? 1 : 2, i2 = f == 0 ? 2 : 3, io = f == 0 ? 3 : (f == 1 ? 2 : (f == 2 ? 1 : 0));
        const V3 nn = cross(P[i1] - P[i0], P[i2] - P[i0]);
        const V3 eo = P[io] - P[i0];
        const float so = dot(nn, eo), s0 = -dot(nn, P[i0]);
        // (a tetrahedron flatter than 1e-5 rad -- four vertices of one box face -- has no inside: all its faces are candidates)
        if (so * s0 < 0.0f || so * so <= 1e-10f * dot(nn, nn) * dot(eo, eo)) {
          inside = false;
          float tl[3];
          tri_bary(P[i0], P[i1], P[i2], tl);
          const V3 c = tl[0] * P[i0] + tl[1] * P[i1] + tl[2] * P[i2];
          const float d2 = dot(c, c);
          if (best < 0.0f || d2 < best) {
            best = d2;
            bl[0] = bl[1] = bl[2] = bl[3] = 0.0f;
            bl[i0] = tl[0]; bl[i1] = tl[1]; bl[i2] = tl[2];
          }
        }
      }
      if (inside) return true;
#pragma unroll
      for (int i = 0; i < 4; i++) l[i] = bl[i];
    }
    // compact: drop the vertices with zero weight, recompute v
    int m = 0;
    V3 nv = v3(0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int i = 0; i < 4; i++)
      if (i < n && l[i] > 0.0f) {
        const V3 pi = P[i], di = D[i];
        const float li = l[i];
        if (m == 0) { P[0] = pi; D[0] = di; lam[0] = li; } else if (m == 1) { P[1] = pi; D[1] = di; lam[1] = li; }
        else if (m == 2) { P[2] = pi; D[2] = di; lam[2] = li; } else { P[3] = pi; D[3] = di; lam[3] = li; }
        nv = nv + li * pi;
        m++;
      }
    n = m;
    v = nv;
    if (!(dot(v, v) > 1e-12f)) return true;  // the cores touch (closer than 1e-6 m: the connecting line carries no direction in float32)
  }
  // (left through the no-progress / repeated-vertex exits with the origin numerically ON the simplex -- a core centre on a symmetry
  //  plane of the other core, e.g. a sphere sunk into the middle of a box face past its radius: an overlap too.  Without this the pair
  //  came back as a "shallow" contact of depth r with an unnormalised normal: tests/test_convex_host.py, analytic cases)
  if (!(dot(v, v) > 1e-12f)) return true;
  pa = v3(0.0f, 0.0f, 0.0f); pb = pa;
#pragma unroll
  for (int i = 0; i < 4; i++)
    if (i < n) {
      pa = pa + lam[i] * core_support(A, neg(D[i]));
      pb = pb + lam[i] * core_support(B, D[i]);
    }
  dist = sqrtf(dot(v, v));
  return false;
}

// ---- Minkowski Portal Refinement on the full shapes (deep penetration) ------------------------------------------------
struct Portal {
  V3 v[4], d[4];  // v[0] = interior point (difference of the centres); v[i] = supA(d[i]) - supB(-d[i]) for i >= 1
};
__device__ __forceinline__ V3 mink(const ShapeD& A, const ShapeD& B, V3 d) { return shape_support(A, d) - shape_support(B, neg(d)); }
__device__ __forceinline__ V3 portal_dir(const Portal& p) {
  V3 d = cross(p.v[2] - p.v[1], p.v[3] - p.v[1]);
  unit(d);
  return d;
}
__device__ __forceinline__ bool portal_reach_tolerance(const Portal& p, V3 v4, V3 d) {
  const float dv4 = dot(v4, d);
  float m = dv4 - dot(p.v[1], d);
  m = fminf(m, dv4 - dot(p.v[2], d));
  m = fminf(m, dv4 - dot(p.v[3], d));
  return m < 1e-6f;
}
__device__ __forceinline__ void portal_expand(Portal& p, V3 v4, V3 d4) {
  const V3 c = cross(v4, p.v[0]);
  if (dot(p.v[1], c) > 0.0f) {
    if (dot(p.v[2], c) > 0.0f) { p.v[1] = v4; p.d[1] = d4; } else { p.v[3] = v4; p.d[3] = d4; }
  } else {
    if (dot(p.v[3], c) > 0.0f) { p.v[2] = v4; p.d[2] = d4; } else { p.v[1] = v4; p.d[1] = d4; }
  }
}

// true = penetrating: depth, normal (A -> B), position
__device__ __forceinline__ bool mpr_pair(const ShapeD& A, const ShapeD& B, float& depth, V3& normal, V3& pos) {
  Portal p;
  p.v[0] = A.pos - B.pos;
  if (!(dot(p.v[0], p.v[0]) > 1e-20f)) p.v[0] = v3(1e-5f, 0.0f, 0.0f);
  V3 d = neg(p.v[0]);
  unit(d);
  p.d[1] = d; p.v[1] = mink(A, B, d);
  if (!(dot(p.v[1], d) > 0.0f)) return false;
  d = cross(p.v[0], p.v[1]);
  if (!unit(d)) {  // the origin lies on the ray v0 -> v1: v1 itself is the penetration
    const float l = sqrtf(dot(p.v[1], p.v[1]));
    depth = l;
    if (l > 1e-15f) normal = (1.0f / l) * p.v[1];
    else normal = (-1.0f / sqrtf(dot(p.v[0], p.v[0]))) * p.v[0];
    pos = 0.5f * (shape_support(A, p.d[1]) + shape_support(B, neg(p.d[1])));
    return true;
  }
  p.d[2] = d; p.v[2] = mink(A, B, d);
  if (!(dot(p.v[2], d) > 0.0f)) return false;
  d = cross(p.v[1] - p.v[0], p.v[2] - p.v[0]);
  unit(d);
  if (dot(d, p.v[0]) > 0.0f) {
    const V3 tv = p.v[1], td = p.d[1];
    p.v[1] = p.v[2]; p.d[1] = p.d[2]; p.v[2] = tv; p.d[2] = td;
    d = neg(d);
  }
  for (int it = 0;; it++) {  // discover the portal
    if (it > 64) return false;
    p.d[3] = d; p.v[3] = mink(A, B, d);
    if (!(dot(p.v[3], d) > 0.0f)) return false;
    bool again = false;
    if (dot(cross(p.v[1], p.v[3]), p.v[0]) < 0.0f) { p.v[2] = p.v[3]; p.d[2] = p.d[3]; again = true; }
    if (!again && dot(cross(p.v[3], p.v[2]), p.v[0]) < 0.0f) { p.v[1] = p.v[3]; p.d[1] = p.d[3]; again = true; }
    if (!again) break;
    d = cross(p.v[1] - p.v[0], p.v[2] - p.v[0]);
    unit(d);
  }
  for (int it = 0;; it++) {  // refine until the portal is beyond the origin
    d = portal_dir(p);
    if (dot(d, p.v[1]) >= 0.0f) break;
    const V3 v4 = mink(A, B, d);
    if (!(dot(v4, d) >= 0.0f) || portal_reach_tolerance(p, v4, d) || it > 64) return false;
    portal_expand(p, v4, d);
  }
  for (int it = 0;; it++) {  // push the portal to the surface of A - B
    d = portal_dir(p);
    const V3 v4 = mink(A, B, d);
    if (portal_reach_tolerance(p, v4, d) || it > 64) {
      float tl[3];
      tri_bary(p.v[1], p.v[2], p.v[3], tl);
      // (the oracle's tri_closest_to_origin builds the point as a + s ab + t ac; same regions, same weights)
      const V3 ab = p.v[2] - p.v[1], ac = p.v[3] - p.v[1];
      const V3 w = tl[0] == 1.0f ? p.v[1] : (tl[1] == 1.0f ? p.v[2] : (tl[2] == 1.0f ? p.v[3] : p.v[1] + tl[1] * ab + tl[2] * ac));
      const float l = sqrtf(dot(w, w));
      depth = l;
      normal = l > 1e-15f ? (1.0f / l) * w : d;
      // position: the origin's barycentric coordinates in the tetrahedron (v0, v1, v2, v3) applied to the witness points
      float b0 = dot(cross(p.v[1], p.v[2]), p.v[3]), b1 = dot(cross(p.v[3], p.v[2]), p.v[0]);
      float b2 = dot(cross(p.v[0], p.v[1]), p.v[3]), b3 = dot(cross(p.v[2], p.v[1]), p.v[0]);
      float sum = b0 + b1 + b2 + b3;
      if (!(sum > 0.0f)) {
        b0 = 0.0f;
        b1 = dot(cross(p.v[2], p.v[3]), d); b2 = dot(cross(p.v[3], p.v[1]), d); b3 = dot(cross(p.v[1], p.v[2]), d);
        sum = b1 + b2 + b3;
      }
      const float inv = 1.0f / sum;
      V3 qa = b0 * A.pos, qb = b0 * B.pos;
      qa = qa + b1 * shape_support(A, p.d[1]); qb = qb + b1 * shape_support(B, neg(p.d[1]));
      qa = qa + b2 * shape_support(A, p.d[2]); qb = qb + b2 * shape_support(B, neg(p.d[2]));
      qa = qa + b3 * shape_support(A, p.d[3]); qb = qb + b3 * shape_support(B, neg(p.d[3]));
      pos = (0.5f * inv) * (qa + qb);
      return true;
    }
    portal_expand(p, v4, d);
  }
}

// one contact of a convex pair (neither a plane, not both boxes): point (pos, dist < 0) and normal from A to B
__device__ __attribute__((noinline)) bool convex_pair(const ShapeD& A, const ShapeD& B, f4& point, V3& n) {
  float dist;
  V3 pa, pb;
  const float ra = core_radius(A), rb = core_radius(B);
  if (!gjk_core_distance(A, B, dist, pa, pb)) {
    if (!(dist < ra + rb)) return false;
    n = (1.0f / dist) * (pb - pa);
    const V3 c = 0.5f * (pa + pb + (ra - rb) * n);  // midway between the surface points pa + ra n and pb - rb n
    point = f4{c.x, c.y, c.z, dist - ra - rb};
    return true;
  }
  float depth;
  V3 pos;
  if (!mpr_pair(A, B, depth, n, pos)) return false;
  point = f4{pos.x, pos.y, pos.z, -depth};
  return true;
}

}  // namespace

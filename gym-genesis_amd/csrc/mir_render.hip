// mir_render.hip — batched tiled rasteriser for the `pixels` observation and render() (SURVEY.md 8f-2,
// BASELINE.json configs[4]).
//
// What it replaces: the reference's per-env Python loop of cam.set_pose + cam.render()
// (/root/reference/gym_genesis/tasks/franka/cube_pick.py:159-180) and GenesisEnv.render
// (/root/reference/gym_genesis/env.py:97-98), which drive Genesis's OpenGL rasteriser B times per step.
//
// MI355X mapping.  At B=1024, 480x640 the output is 943 MB per step, so the kernel is HBM-WRITE-bound
// by construction; everything else is arranged to stay below that:
//   * k_render_setup: world pose of every primitive from the step kernel's FK cache and the camera expressed in the
//     primitive's frame (origin o', and the images F', R', U' of the camera basis, so a pixel's ray in that frame is
//     d' = F' + x R' + y U').  A box becomes SIX AFFINE FUNCTIONS of the pixel that bound the reciprocal depth w = 1 / t of
//     the ray inside it (upper bounds: the faces towards the camera, where it may enter; lower bounds: the others) with the
//     FINAL packed RGB8 colours of the entry faces (albedo x shade); a plane keeps o', F', R', U', its two checker colours and
//     the affine numerators of the perspective-correct checker coordinate.  Plus a conservative screen rectangle; 128 B per
//     primitive.  Per-env cameras (mir_render_cams) are resolved here too.  For per-env images one WAVE handles an env
//     (lane = geom) and also writes the per-strip primitive lists.
//   * mir_render_binned (per-env images, the pixels observation): workgroup = 4 waves stacked on a 128-pixel-wide strip
//     of 256 rows, walked as 128 x 32 sub-tiles, wave = a band of 128 x 8 pixels.  The strip's list arrives with one
//     vector load (lane k = entry k); depth-tested primitives are drawn on 32 x 8-pixel regions (lane = 4 pixels of one
//     row), only where the entry's row range and column mask say so, records by wave-uniform SCALAR loads; their colours
//     move through a wave-private LDS tile into the store layout (regions of 128 x 2: lane = 4 consecutive pixels, one
//     global_store_dwordx3 per lane and region = whole 128-byte lines, not waited for); the floor is drawn last, in
//     the store layout, without a depth buffer.  The strips are walked XCD by XCD: the workgroups of one XCD stream one
//     contiguous eighth of the output.  1024 x 480 x 640: 180 - 215 us = 4.4 - 5.2 TB/s written, depending on where the
//     driver placed the output buffer (round 3: 245 - 257 us; round 2: 335 us).
//   * mir_render_kernel (the global view of all envs, odd widths): the same tiling with the list culled per workgroup
//     (ordered ballot compaction into LDS) and everything drawn in the store layout.
//   * ray / box is min(upper bounds) >= max(lower bounds) on those six functions: packed FMAs, min3 / max3, no reciprocal
//     per pixel; depth order is resolved per pixel on w (strict >, list in ascending primitive order => deterministic);
//     the colour of a hit is selected at hit time from the record, so there is no shading pass and no per-pixel lookup.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <type_traits>
#include <cstdlib>
#include <cstring>

#include "mir_model.h"
#include "mir_scene.h"

#define TW 128   /* tile width, pixels  */
#ifndef TH
#define TH 96   /* rows per workgroup strip (multiple of 32): walked as 128 x 32 sub-tiles */
#endif
#ifndef TH_BINNED
#define TH_BINNED 256 /* strip height of the binned kernel (tools/probes/render_sweep.py, 1024 x 480 x 640 in the bench's state, XCD-contiguous walk, two boxes: 64 -> 220, 96 -> 195, 128 -> 187, 160 -> 185 / 195, 224 -> 200, 256 -> 180 / 196, 320 -> 207, 480 -> 209 us) */
#endif
#define PREC 32  /* floats per primitive record */
#ifndef MIR_RENDER_NT
#define MIR_RENDER_NT 0 /* 1: nontemporal pixel stores */
#endif

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef f4 __attribute__((address_space(4))) cf4;  // constant address space: uniform reads become scalar loads
typedef unsigned u3 __attribute__((ext_vector_type(3)));

struct V3 {
  float x, y, z;
};
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

struct Cam {
  float pos[3], f[3], r[3], u[3];  // orthonormal camera basis in world axes
  float tanx, tany;                // half-extent of the image plane at unit depth
  int W, H;
};

#define VIS_HDR 3u /* words in front of the work items: [0] their count, [1] [2] bit g set = geom g is a plane (env 0's planes are drawn by k_global_resolve) */
#define GLOBAL_ITEMS_MAX 32 /* work items per box of the global splat: an entry = box index (22 bits) | item << 22 | (items - 1) << 27 */
struct SetupArgs {
  const GeomTab* geom;
  const float* poses;       // (B, 2, pst, 4) FK cache
  int pst;                  // bodies per env in the pose cache
  const float* env_offset;  // (B, 3) or null
  const float* cam_pos;     // (B, 3) per-env camera position, or null (use cam)
  const float* cam_look;    // (B, 3) per-env look-at point
  const float* cam_up;      // (B, 3) per-env up hint, or null (use up0)
  float up0[3];
  float* prims;             // (B, ngeom, PREC)
  Cam cam;
  float light[3];
  float rgb[MIR_MAX_GEOM][3];
  float chk[2][3];
  float amb, dif, inv_chk;
  int B, ngeom, global_mode;
  unsigned* vis;            // global view of many envs: the boxes that can be on screen are LISTED here ([0] = count), the others skipped
};

struct PixArgs {
  const float* prims;
  uint8_t* pixels;
  int W, H, nprim;       // primitives per image
  float x0, dx, y0, dy;  // image-plane coordinates of pixel (i, j): x0 + i dx, y0 + j dy
  unsigned sky;          // packed RGB8
  int th;                // rows per workgroup strip (multiple of 32)
  const int* bins;       // binned kernel: (image, strip row, strip column, BINW) lists written by k_render_bin
  int nsx, nsy, nwg;     // binned kernel: strips per image row / column, workgroups = images x nsx x nsy (1-D grid, see the kernel)
};

#define BINW 64  /* ints per strip list: header (count | floor flag << 8), then one packed entry per listed primitive */
struct BinArgs {
  const float* prims;
  int* bins;
  int W, H, nprim, th, nsx, nsy, nimg;
};

__device__ __forceinline__ f2 rcp2(f2 v) { return f2{__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)}; }
__device__ __forceinline__ unsigned to_u8(float c) {
  return (unsigned)(fminf(fmaxf(c, 0.0f), 1.0f) * 255.0f + 0.5f);
}
__device__ __forceinline__ float pack_rgb(const float* alb, float shade) {
  return __uint_as_float(to_u8(alb[0] * shade) | to_u8(alb[1] * shade) << 8 | to_u8(alb[2] * shade) << 16);
}

// ---- per-(env, geom) primitive record (8 quads; every colour is final: albedo x shade, packed RGB8) ----
//  plane: q0 o' | type      q1 F' | xmin      q2 R' | xmax      q3 U' | ymin      q4 half extents | ymax
//  box  : six affine bounds A_i(x, y) = A0_i + x AX_i + y AY_i on the reciprocal depth (see the box branch below):
//         q0 A0_0..2 | type + (nup << 8)   q1 AX_0..2 | xmin   q2 AY_0..2 | xmax   q3 A0_3..5 | ymin   q4 AX_3..5 | ymax   q7 AY_3..5
//         q5 = colours of the faces behind the upper bounds 0 .. nup-1
//  plane: q5 = Nu0 NuX NuY colour(even cell), q6 = Nv0 NvX NvY colour(odd cell), with the hit point's checker
//         coordinate u/2 = (Nu0 + x NuX + y NuY) / d'z  (perspective-correct ratio of two affine functions)
// BIN: one WAVE per env (lane = geom, ngeom <= 63) and the per-strip lists of the binned pixel kernel are written here as well
// (see below); otherwise one thread per (env, geom).
template <bool BIN>
__global__ void k_render_setup(SetupArgs a, BinArgs bn) {
  const int gi = blockIdx.x * blockDim.x + threadIdx.x;
  const int e = BIN ? gi >> 6 : gi / a.ngeom;
  const bool valid = BIN ? (e < a.B && (gi & 63) < a.ngeom) : gi < a.B * a.ngeom;
  if (BIN ? e >= a.B : !valid) return;  // (BIN: whole waves leave together)
  const int g = valid ? (BIN ? gi & 63 : gi % a.ngeom) : 0;
  const int i = e * a.ngeom + g;
  const GeomTab* __restrict__ m = a.geom;
  const int b = m->g_body[g], gtype = m->g_type[g];
  // round geoms are drawn as their bounding boxes (sphere: r r r; capsule: r r half+r): the rasteriser knows boxes and planes
  const int type = (gtype == MIR_GEOM_SPHERE || gtype == MIR_GEOM_CAPSULE) ? MIR_GEOM_BOX : gtype;
  const float* pp = a.poses + ((size_t)e * 2 * a.pst + b) * 4;
  const V3 xp = {pp[0], pp[1], pp[2]};
  const float* qq = pp + 4 * a.pst;
  const float bw = qq[0], bx = qq[1], by = qq[2], bz = qq[3];
  // geom frame in the world: c = xpos + R(xquat) g_pos (+ env offset), q = xquat * g_quat
  const V3 gp = {m->g_pos[g][0], m->g_pos[g][1], m->g_pos[g][2]};
  const V3 uq = {bx, by, bz};
  const V3 t = 2.0f * cross(uq, gp);
  V3 c = xp + gp + bw * t + cross(uq, t);
  if (a.global_mode && a.env_offset) c = c + V3{a.env_offset[e * 3 + 0], a.env_offset[e * 3 + 1], a.env_offset[e * 3 + 2]};
  const float gw = m->g_quat[g][0], gx = m->g_quat[g][1], gy = m->g_quat[g][2], gz = m->g_quat[g][3];
  const float w = bw * gw - bx * gx - by * gy - bz * gz, x = bw * gx + bx * gw + by * gz - bz * gy;
  const float y = bw * gy - bx * gz + by * gw + bz * gx, z = bw * gz + bx * gy - by * gx + bz * gw;
  const V3 ax[3] = {{1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)},
                    {2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)},
                    {2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)}};  // world axes (columns of R)
  V3 cp = {a.cam.pos[0], a.cam.pos[1], a.cam.pos[2]}, cf = {a.cam.f[0], a.cam.f[1], a.cam.f[2]};
  V3 cr = {a.cam.r[0], a.cam.r[1], a.cam.r[2]}, cu = {a.cam.u[0], a.cam.u[1], a.cam.u[2]};
  if (a.cam_pos) {  // per-env camera (wrist cameras): the same look-at construction as the host does for `cam`
    cp = V3{a.cam_pos[e * 3], a.cam_pos[e * 3 + 1], a.cam_pos[e * 3 + 2]};
    V3 d = V3{a.cam_look[e * 3], a.cam_look[e * 3 + 1], a.cam_look[e * 3 + 2]} - cp;
    cf = (1.0f / sqrtf(fmaxf(dot(d, d), 1e-30f))) * d;
    V3 up = a.cam_up ? V3{a.cam_up[e * 3], a.cam_up[e * 3 + 1], a.cam_up[e * 3 + 2]} : V3{a.up0[0], a.up0[1], a.up0[2]};
    V3 rr = cross(cf, up);
    if (dot(rr, rr) < 1e-12f) rr = cross(cf, V3{0.0f, 1.0f, 0.0f});  // view parallel to up: fall back to +y, then +x
    if (dot(rr, rr) < 1e-12f) rr = cross(cf, V3{1.0f, 0.0f, 0.0f});
    cr = (1.0f / sqrtf(dot(rr, rr))) * rr;
    cu = cross(cr, cf);
  }
  const V3 L = {a.light[0], a.light[1], a.light[2]};
  const V3 rel = cp - c;
  const float* gs = m->g_size[g];
  const V3 h = gtype == MIR_GEOM_SPHERE ? V3{gs[0], gs[0], gs[0]} : (gtype == MIR_GEOM_CAPSULE ? V3{gs[0], gs[0], gs[0] + gs[1]} : V3{gs[0], gs[1], gs[2]});
  if constexpr (!BIN) {
    // The global view of thousands of envs shows a few dozen of them (the camera stands inside the grid: 99 % of the boxes are beside or
    // behind it).  A box whose bounding sphere lies outside the view pyramid is dropped HERE, before its corners and bounds are worked
    // out and without a record: the splat kernel walks the list of the others (a.vis), nobody reads the records of these.  Conservative:
    // every point p of the sphere has |p_x| >= |x_c| - r and p_z <= z_c + r.
    if (a.vis && type == MIR_GEOM_BOX) {
      const float r = sqrtf(dot(h, h));
      const float zc = -dot(rel, cf), xc = fabsf(dot(rel, cr)), yc = fabsf(dot(rel, cu));
      if (zc + r <= 1e-3f || xc - r > (zc + r) * a.cam.tanx || yc - r > (zc + r) * a.cam.tany) return;
    }
  }
  const V3 o = {dot(ax[0], rel), dot(ax[1], rel), dot(ax[2], rel)};
  const V3 F = {dot(ax[0], cf), dot(ax[1], cf), dot(ax[2], cf)}, R = {dot(ax[0], cr), dot(ax[1], cr), dot(ax[2], cr)};
  const V3 U = {dot(ax[0], cu), dot(ax[1], cu), dot(ax[2], cu)};
  const float lk[3] = {dot(ax[0], L), dot(ax[1], L), dot(ax[2], L)};
  int xmin = 0, xmax = a.cam.W - 1, ymin = 0, ymax = a.cam.H - 1;
  f4 q5, q6;
  float bA0[6], bAX[6], bAY[6], ucol[3] = {0, 0, 0};  // boxes: the six bounds, see below
  int nup = 0;
  if (type == MIR_GEOM_BOX) {
    float pxmin = 3e38f, pxmax = -3e38f, pymin = 3e38f, pymax = -3e38f;
    int behind = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const V3 v = c + ((k & 1) ? h.x : -h.x) * ax[0] + ((k & 2) ? h.y : -h.y) * ax[1] + ((k & 4) ? h.z : -h.z) * ax[2] - cp;
      const float zc = dot(v, cf);
      if (zc <= 1e-3f) { behind++; continue; }
      const float px = (dot(v, cr) / (zc * a.cam.tanx) + 1.0f) * 0.5f * (float)a.cam.W;
      const float py = (1.0f - dot(v, cu) / (zc * a.cam.tany)) * 0.5f * (float)a.cam.H;
      pxmin = fminf(pxmin, px); pxmax = fmaxf(pxmax, px);
      pymin = fminf(pymin, py); pymax = fmaxf(pymax, py);
    }
    if (behind != 0 && behind != 8) {
      // straddling the camera plane: the part in front of it is the convex hull of the corners in front and of the points where
      // the box's edges cross the plane z = 1e-3 -- their projections bound its image (such a box stands beside or under the
      // camera: most of them end up off screen, the others with a band along one edge instead of the whole image).  In the global
      // view of thousands of envs the camera stands INSIDE the grid, and a few hundred boxes straddle.
#pragma unroll
      for (int ed = 0; ed < 12; ed++) {
        const int ax_ = ed >> 2, lo2 = ed & 3;                     // edge along axis ax_, the other two coordinates from lo2
        const int bit = 1 << ax_;
        const int base = ((lo2 & 1) << (ax_ == 0 ? 1 : 0)) | ((lo2 >> 1) << (ax_ == 2 ? 1 : 2));
        const int ka = base, kb = base | bit;
        const V3 va = c + ((ka & 1) ? h.x : -h.x) * ax[0] + ((ka & 2) ? h.y : -h.y) * ax[1] + ((ka & 4) ? h.z : -h.z) * ax[2] - cp;
        const V3 vb = c + ((kb & 1) ? h.x : -h.x) * ax[0] + ((kb & 2) ? h.y : -h.y) * ax[1] + ((kb & 4) ? h.z : -h.z) * ax[2] - cp;
        const float za = dot(va, cf), zb = dot(vb, cf);
        if ((za > 1e-3f) == (zb > 1e-3f)) continue;
        const float t = (1e-3f - za) / (zb - za);
        const V3 vi = va + t * (vb - va);
        const float px = (dot(vi, cr) / (1e-3f * a.cam.tanx) + 1.0f) * 0.5f * (float)a.cam.W;
        const float py = (1.0f - dot(vi, cu) / (1e-3f * a.cam.tany)) * 0.5f * (float)a.cam.H;
        pxmin = fminf(pxmin, px); pxmax = fmaxf(pxmax, px);
        pymin = fminf(pymin, py); pymax = fmaxf(pymax, py);
      }
    }
    if (behind == 8) { xmin = 1; xmax = 0; ymin = 1; ymax = 0; }  // wholly behind the camera: never drawn
    else {
      xmin = max(0, (int)fmaxf(floorf(fmaxf(pxmin, -1e6f)) - 1.0f, -1.0f)); xmax = min(a.cam.W - 1, (int)fminf(ceilf(fminf(pxmax, 1e6f)) + 1.0f, 1e9f));
      ymin = max(0, (int)fmaxf(floorf(fmaxf(pymin, -1e6f)) - 1.0f, -1.0f)); ymax = min(a.cam.H - 1, (int)fminf(ceilf(fminf(pymax, 1e6f)) + 1.0f, 1e9f));
    }
    // The box as six bounds on w = 1 / t along the pixel's ray d' = F' + x R' + y U' (t = depth: F' is the unit view axis):
    //   |o'_k + t d'_k| <= h_k   <=>   (h_k - o'_k) w >= d'_k   and   (h_k + o'_k) w >= -d'_k        (k = x, y, z)
    // Dividing by the left factors, whose signs are properties of the BOX (which side of slab k the camera is on), every
    // constraint is  w <= A(x, y)  (the slab face towards a camera outside slab k: where the ray may enter) or  w >= A(x, y)
    // (every other face), with A affine in the pixel -- no reciprocal per pixel.  The ray hits iff max(lower) <= min(upper); it
    // enters at w = min(upper), through the face that supplies the minimum.  Functions 0 .. nup-1 are the upper bounds (with their
    // faces' colours in q5), nup .. 5 the lower ones; nup = 0 (camera inside the box) is never drawn.
    const float* alb = a.rgb[g];
    const float oo[3] = {o.x, o.y, o.z}, hh[3] = {h.x, h.y, h.z};
    const float d0[3] = {F.x, F.y, F.z}, dX[3] = {R.x, R.y, R.z}, dY[3] = {U.x, U.y, U.z};
    // per axis k: f1 = the bound that is the upper one when the camera is outside slab k (else the slab's second lower bound),
    // f2 = the slab's (first) lower bound; as functions of the pixel: f * (F'_k, R'_k, U'_k)
    float a1[3][3], a2[3][3], acol[3];
    unsigned outm = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
      const float dm = hh[k] - oo[k], dp = hh[k] + oo[k];  // dm < 0: beyond the +k face; dp < 0: beyond the -k face
      const float c1 = 1.0f / (dm < 0.0f ? fminf(dm, -1e-20f) : fmaxf(dm, 1e-20f)), c2 = -1.0f / (dp < 0.0f ? fminf(dp, -1e-20f) : fmaxf(dp, 1e-20f));
      const bool hi = dm < 0.0f;
      const float f1 = hi ? c1 : c2, f2 = hi ? c2 : c1;
      a1[k][0] = f1 * d0[k]; a1[k][1] = f1 * dX[k]; a1[k][2] = f1 * dY[k];
      a2[k][0] = f2 * d0[k]; a2[k][1] = f2 * dX[k]; a2[k][2] = f2 * dY[k];
      acol[k] = pack_rgb(alb, a.amb + a.dif * fmaxf(hi ? lk[k] : -lk[k], 0.0f));
      outm |= (hi || dp < 0.0f) ? 1u << k : 0u;
    }
    // the axes with the camera outside their slab first (a stable partition of x, y, z; six bits per case, no indexed memory)
    const unsigned perm = (unsigned)((0x909612921924ull >> (6 * outm)) & 63ull);
    nup = __popc(outm);
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const unsigned k = (perm >> (2 * j)) & 3u;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float v1 = k == 0 ? a1[0][c] : (k == 1 ? a1[1][c] : a1[2][c]), v2 = k == 0 ? a2[0][c] : (k == 1 ? a2[1][c] : a2[2][c]);
        (c == 0 ? bA0 : (c == 1 ? bAX : bAY))[j] = v1;
        (c == 0 ? bA0 : (c == 1 ? bAX : bAY))[3 + j] = v2;
      }
      ucol[j] = k == 0 ? acol[0] : (k == 1 ? acol[1] : acol[2]);
    }
    if (nup == 0) { xmin = 1; xmax = 0; ymin = 1; ymax = 0; }
    q5 = f4{ucol[0], ucol[1], ucol[2], 0.0f};
    q6 = f4{0.0f, 0.0f, 0.0f, 0.0f};
  } else {
    if (a.global_mode && e > 0) { xmin = 1; xmax = 0; ymin = 1; ymax = 0; }  // an unbounded plane is drawn once (env 0's)
    if constexpr (!BIN) { if (a.vis && e == 0) atomicOr(a.vis + 1 + (g >> 5), 1u << (g & 31)); }
    const float nl = o.z < 0.0f ? -lk[2] : lk[2];  // the side of the plane the camera is on
    const float sh = a.amb + a.dif * fmaxf(nl, 0.0f);
    const float s = 0.5f * a.inv_chk;
    q5 = f4{s * (o.x * F.z - o.z * F.x), s * (o.x * R.z - o.z * R.x), s * (o.x * U.z - o.z * U.x), pack_rgb(a.chk[0], sh)};
    q6 = f4{s * (o.y * F.z - o.z * F.y), s * (o.y * R.z - o.z * R.y), s * (o.y * U.z - o.z * U.y), pack_rgb(a.chk[1], sh)};
  }
  if (valid) {
    f4* o4 = reinterpret_cast<f4*>(a.prims + (size_t)i * PREC);
    if (type == MIR_GEOM_BOX) {
      o4[0] = f4{bA0[0], bA0[1], bA0[2], __int_as_float(type | nup << 8)};
      o4[1] = f4{bAX[0], bAX[1], bAX[2], __int_as_float(xmin)};
      o4[2] = f4{bAY[0], bAY[1], bAY[2], __int_as_float(xmax)};
      o4[3] = f4{bA0[3], bA0[4], bA0[5], __int_as_float(ymin)};
      o4[4] = f4{bAX[3], bAX[4], bAX[5], __int_as_float(ymax)};
      o4[7] = f4{bAY[3], bAY[4], bAY[5], 0.0f};
    } else {
      o4[0] = f4{o.x, o.y, o.z, __int_as_float(type)};
      o4[1] = f4{F.x, F.y, F.z, __int_as_float(xmin)};
      o4[2] = f4{R.x, R.y, R.z, __int_as_float(xmax)};
      o4[3] = f4{U.x, U.y, U.z, __int_as_float(ymin)};
      o4[4] = f4{h.x, h.y, h.z, __int_as_float(ymax)};
      o4[7] = f4{0, 0, 0, 0};
    }
    o4[5] = q5;
    o4[6] = q6;
  }
  if constexpr (!BIN) {
    if (a.vis && type == MIR_GEOM_BOX && xmin <= xmax && ymin <= ymax) {
      // work items of k_global_splat: the 32 x 8 blocks of the rectangle in up to GLOBAL_ITEMS_MAX interleaved shares of about four
      // (a slab close to the camera covers hundreds of blocks and must not keep one workgroup busy after the others have left)
      const int nb = ((xmax >> 5) - (xmin >> 5) + 1) * ((ymax >> 3) - (ymin >> 3) + 1);
      const unsigned n = (unsigned)min(GLOBAL_ITEMS_MAX, (nb + 3) >> 2);
      const unsigned at = atomicAdd(a.vis, n);
      for (unsigned k = 0; k < n; k++) a.vis[VIS_HDR + at + k] = (unsigned)i | k << 22 | (n - 1u) << 27;  // (i < 2^22: checked by mir_render)
    }
  }
  if (!BIN) return;
  // ---- per-strip primitive lists of the binned pixel kernel (per-env images with <= 63 primitives), built while the rectangles
  // are still in registers.  For every strip of the image the primitives whose rectangle meets it are listed in ascending order
  // (ballot + prefix count), each as  id | ymin << 6 | ymax << 17 | columns << 28  (rows clamped to 11 bits; `columns` = which of
  // the strip's four 32-pixel columns the rectangle meets), so that the pixel kernel can cull a primitive against a wave's 8-row
  // band and its 32 x 8 regions WITHOUT fetching its record.  Header = count | floor << 8: `floor` says that primitive 0 is a
  // plane seen by a roll-free camera (the floor of every scene of the reference); it is then left out of the list and drawn by
  // the pixel kernel's floor pass.  This replaces the per-workgroup culling prologue of the generic kernel (one global load per
  // thread, two ballots and three barriers per strip: 40 us of the 335 us a 1024 x 480 x 640 render took).
  const int lane = gi & 63;
  const bool is_floor = valid && g == 0 && type != MIR_GEOM_BOX && R.z == 0.0f && xmin <= 0 && xmax >= bn.W - 1 && ymin <= 0 && ymax >= bn.H - 1;
  const int floor = __builtin_amdgcn_readfirstlane(is_floor ? 1 : 0);  // lane 0 = geom 0
  for (int st = 0; st < bn.nsx * bn.nsy; st++) {
    const int sx = st % bn.nsx, sy = st / bn.nsx;
    const int tx0 = sx * TW, txmax = min(tx0 + TW, bn.W) - 1, sy0 = sy * bn.th, symax = min(sy0 + bn.th, bn.H) - 1;
    const bool hit = valid && !is_floor && xmin <= txmax && xmax >= tx0 && ymin <= symax && ymax >= sy0;
    const unsigned long long bal = __ballot(hit);
    int* out = bn.bins + ((size_t)e * bn.nsx * bn.nsy + st) * BINW;
    if (hit) {
      int rmask = 0;
#pragma unroll
      for (int r = 0; r < 4; r++) rmask |= (xmax < tx0 + 32 * r || xmin > tx0 + 32 * r + 31) ? 0 : 1 << r;
      out[1 + __popcll(bal & ((1ull << lane) - 1ull))] = g | max(ymin, 0) << 6 | min(ymax, 2047) << 17 | rmask << 28;
    }
    if (lane == 0) out[0] = __popcll(bal) | floor << 8;
  }
}

// A box on ONE region of a wave: the lane's 4 consecutive pixels (two packed pairs, image-plane x in xs) of the row with image-plane
// y `ys`.  Six packed FMAs per pixel pair for the bounds (the record's affine functions of the pixel), min3 / max3, two compares,
// the face select -- ~15 VALU instructions per pixel and no reciprocal (the slab test in t this replaces: ~35 plus three
// quarter-rate reciprocals).  `best` holds RECIPROCAL depths (larger = nearer; 0 = nothing).  Shared by both pixel kernels,
// which therefore agree bit for bit.
template <int NUP>
__device__ __forceinline__ void box_bounds(const f4 a0, const f4 ax, const f4 ay, const f4 b0, const f4 bx, const f4 by, const f4 q5, float ys,
                                           const f2 (&xs)[2], f2 (&best)[2], unsigned (&col)[4]) {
  const float e0 = fmaf(ys, ay.x, a0.x), e1 = fmaf(ys, ay.y, a0.y), e2 = fmaf(ys, ay.z, a0.z);
  const float e3 = fmaf(ys, by.x, b0.x), e4 = fmaf(ys, by.y, b0.y), e5 = fmaf(ys, by.z, b0.z);
  const unsigned c0 = __float_as_uint(q5.x), c1 = __float_as_uint(q5.y), c2 = __float_as_uint(q5.z);
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const f2 v0 = xs[h] * ax.x + e0, v1 = xs[h] * ax.y + e1, v2 = xs[h] * ax.z + e2;
    const f2 v3 = xs[h] * bx.x + e3, v4 = xs[h] * bx.y + e4, v5 = xs[h] * bx.z + e5;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      float hi, lo = fmaxf(fmaxf(v3[q], v4[q]), v5[q]);
      unsigned c;
      if constexpr (NUP == 3) {
        hi = fminf(fminf(v0[q], v1[q]), v2[q]);
        c = hi == v0[q] ? c0 : (hi == v1[q] ? c1 : c2);
      } else if constexpr (NUP == 2) {
        hi = fminf(v0[q], v1[q]);
        lo = fmaxf(lo, v2[q]);
        c = hi == v0[q] ? c0 : c1;
      } else {
        hi = v0[q];
        lo = fmaxf(lo, fmaxf(v1[q], v2[q]));
        c = c0;
      }
      const bool upd = lo <= hi && hi > best[h][q];
      best[h][q] = upd ? hi : best[h][q];
      col[2 * h + q] = upd ? c : col[2 * h + q];
    }
  }
}
// (q0 .. q4, q7, q5 of a box record; the number of upper bounds is wave-uniform: a scalar branch)
__device__ __forceinline__ void box_region(const f4 q0, const f4 q1, const f4 q2, const f4 q3, const f4 q4, const f4 q7, const f4 q5, float ys,
                                           const f2 (&xs)[2], f2 (&best)[2], unsigned (&col)[4]) {
  const int nup = __float_as_int(q0.w) >> 8;
  if (nup == 3) box_bounds<3>(q0, q1, q2, q3, q4, q7, q5, ys, xs, best, col);
  else if (nup == 2) box_bounds<2>(q0, q1, q2, q3, q4, q7, q5, ys, xs, best, col);
  else box_bounds<1>(q0, q1, q2, q3, q4, q7, q5, ys, xs, best, col);
}

// a plane in the general position (depth-tested, any camera) on one region
__device__ __forceinline__ void plane_region(const f4 ro, const f4 rf, const f4 rr, const f4 ru, const f4 q5, const f4 q6, float ys,
                                             const f2 (&xs)[2], f2 (&best)[2], unsigned (&col)[4]) {
  const unsigned ceven = __float_as_uint(q5.w), codd = __float_as_uint(q6.w);
  const float ez = fmaf(ys, ru.z, rf.z), eu = fmaf(ys, q5.z, q5.x), ev = fmaf(ys, q6.z, q6.x);
  const float nio = __builtin_amdgcn_rcpf(-ro.z);  // w = 1 / t = d'z / -o'z
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const f2 dz = xs[h] * rr.z + ez;
    const f2 iz = rcp2(dz);
    const f2 w = dz * nio;
    const f2 u = (xs[h] * q5.y + eu) * iz, v = (xs[h] * q6.y + ev) * iz;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const bool odd = (__builtin_amdgcn_fractf(u[q]) >= 0.5f) != (__builtin_amdgcn_fractf(v[q]) >= 0.5f);
      const bool upd = w[q] < 1e6f && w[q] > best[h][q];
      best[h][q] = upd ? w[q] : best[h][q];
      col[2 * h + q] = upd ? (odd ? codd : ceven) : col[2 * h + q];
    }
  }
}

// ---- pixels --------------------------------------------------------------------------------------
// 256 threads = 4 waves stacked; a workgroup walks a 128-pixel-wide strip of `th` rows in 128 x 32 sub-tiles.
// Per sub-tile: the image's primitives are culled against it (ordered ballot compaction into an LDS id list), then
// every listed primitive is tested on the wave's 128 x 8 band (4 regions of 128 x 2; lane = 4 consecutive pixels of
// one row; wave-uniform band and region cull), then the 4 regions are stored as packed RGB8, one global_store_dwordx3 per
// lane and region: a store instruction writes two whole rows of 384 B, i.e. six full 128-byte lines (round 1 gave each wave a
// 32-pixel column, 96-byte pieces that straddle lines shared with the neighbouring wave: 348 -> 335 us per 1024 x 480 x 640).  The wave does NOT wait for those stores: it goes on to the next sub-tile, so the write traffic
// of one sub-tile drains under the arithmetic of the next (a workgroup per sub-tile serialises the two: measured
// 240 us of arithmetic + 150 us of stores = 390 us).  Primitive records are read with wave-uniform addresses
// straight from global memory (scalar loads, amortised over 16 pixels per lane); LDS holds only the id list.
__global__ __launch_bounds__(256) void mir_render_kernel(PixArgs a) {
  __shared__ int s_ids[256];
  __shared__ int s_wcnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tx0 = blockIdx.x * TW, sy0 = blockIdx.y * a.th, img = blockIdx.z;
  // lane -> pixels: a wave covers the full 128-pixel width of the strip, 2 rows per region (lane & 31 = 4 consecutive pixels,
  // lane >> 5 = row), four regions = 8 rows; the four waves are stacked.  One store instruction of a wave thus writes whole
  // rows of 384 B = three full 128-byte lines (32-pixel columns per wave wrote 96-byte pieces that straddle lines shared with
  // the neighbouring wave).
  const int px = tx0 + 4 * (lane & 31);
  const float* __restrict__ prims = a.prims + (size_t)img * a.nprim * PREC;
  const int txmax = min(tx0 + TW, a.W) - 1, symax = min(sy0 + a.th, a.H) - 1;
  const bool fast = (a.W & 3) == 0;
  const bool onechunk = a.nprim <= 256;
  int total = 0;
  f2 xs[2];  // pixel pairs feed the packed-fp32 pipe (v_pk_fma_f32 / v_pk_mul_f32)
  xs[0] = f2{a.x0 + (float)px * a.dx, a.x0 + (float)(px + 1) * a.dx};
  xs[1] = f2{a.x0 + (float)(px + 2) * a.dx, a.x0 + (float)(px + 3) * a.dx};

  for (int ty0 = sy0; ty0 <= symax; ty0 += 32) {
    const int tymax = min(ty0 + 31, symax);
    const int wy0 = ty0 + 8 * wv;            // this wave's band of 8 rows
    const int prow = wy0 + (lane >> 5);      // rows prow, prow + 2, prow + 4, prow + 6
    f2 best[4][2];
    unsigned col[4][4];
    float ysr[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      best[r][0] = best[r][1] = f2{0.0f, 0.0f};  // reciprocal depths: 0 = nothing drawn
      ysr[r] = a.y0 + (float)(prow + 2 * r) * a.dy;
#pragma unroll
      for (int p = 0; p < 4; p++) col[r][p] = a.sky;
    }
    for (int base = 0; base < a.nprim; base += 256) {
      // ---- cull this chunk of primitives; ordered compaction of the survivors.  With a single chunk (every per-env
      // image) the list is built ONCE per strip, against the strip: no vector load is issued after the first pixel
      // stores, so no s_waitcnt vmcnt ever has to drain them
      if (!onechunk || ty0 == sy0) {
        const int cy0 = onechunk ? sy0 : ty0, cy1 = onechunk ? symax : tymax;
        const int pi = base + tid;
        bool hit = false;
        if (pi < a.nprim) {
          const float* rec = prims + (size_t)pi * PREC;
          const int xmin = __float_as_int(rec[7]), xmax = __float_as_int(rec[11]), ymin = __float_as_int(rec[15]), ymax = __float_as_int(rec[19]);
          hit = xmin <= txmax && xmax >= tx0 && ymin <= cy1 && ymax >= cy0;
        }
        const unsigned long long bal = __ballot(hit);
        __syncthreads();  // the previous list is no longer being read
        if (lane == 0) s_wcnt[wv] = __popcll(bal);
        __syncthreads();
        int off = 0;
        total = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const int c = s_wcnt[k];
          off += k < wv ? c : 0;
          total += c;
        }
        if (hit) s_ids[off + __popcll(bal & ((1ull << lane) - 1ull))] = pi;
        __syncthreads();
      }
      for (int k = 0; k < total; k++) {
        const int id = __builtin_amdgcn_readfirstlane(s_ids[k]);
        // constant address space => s_load_dwordx4 into SGPRs (the records were written by the previous kernel); a
        // plain global_load here would also tie the record fetch to the outstanding pixel stores through vmcnt
        const cf4* rec = (const cf4*)(uintptr_t)(prims + (size_t)id * PREC);
        const f4 ro = rec[0], rf = rec[1], rr = rec[2], ru = rec[3], rh = rec[4], q5 = rec[5], q6 = rec[6];
        const int ymin = __float_as_int(ru.w), ymax = __float_as_int(rh.w);
        if (ymax < wy0 || ymin > wy0 + 7) continue;  // wave-uniform band cull (the strip cull covered x)
        if ((__float_as_int(ro.w) & 255) == MIR_GEOM_BOX) {
          const f4 q7 = rec[7];
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int wy = wy0 + 2 * r;  // wave-uniform region cull: rows wy, wy + 1
            if (ymax < wy || ymin > wy + 1) continue;
            box_region(ro, rf, rr, ru, rh, q7, q5, ysr[r], xs, best[r], col[r]);
          }
        } else {
          const unsigned ceven = __float_as_uint(q5.w), codd = __float_as_uint(q6.w);
          // (the floor is the first primitive of every scene: nothing has been drawn into the sub-tile when it comes up first, so
          //  its depth test is the validity test alone and the pixel is ASSIGNED: one select per pixel instead of a compare and
          //  two selects.  Wave-uniform.)
          const bool first = k == 0 && base == 0;
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const float ez = fmaf(ysr[r], ru.z, rf.z), eu = fmaf(ysr[r], q5.z, q5.x), ev = fmaf(ysr[r], q6.z, q6.x);
            if (rr.z == 0.0f) {
              // the camera's right vector lies in the plane (no roll over a horizontal floor: every camera of the
              // reference): the ray's normal component, hence 1/d'z and the depth, are constant along an image row --
              // one reciprocal per lane instead of one per pixel (x * 0 + ez == ez: bit-identical to the general path)
              const float iz1 = __builtin_amdgcn_rcpf(ez);
              const float t1 = iz1 * (-ro.z);
              const bool vld = t1 > 1e-6f;
              const float w1 = __builtin_amdgcn_rcpf(t1);
              if (first) {
                const float tb = vld ? w1 : 0.0f;
                const unsigned ca = vld ? ceven : a.sky, cb = vld ? codd : a.sky;
#pragma unroll
                for (int h = 0; h < 2; h++) {
                  const f2 u = (xs[h] * q5.y + eu) * iz1, v = (xs[h] * q6.y + ev) * iz1;
#pragma unroll
                  for (int q = 0; q < 2; q++) {
                    const bool odd = (__builtin_amdgcn_fractf(u[q]) >= 0.5f) != (__builtin_amdgcn_fractf(v[q]) >= 0.5f);
                    best[r][h][q] = tb;
                    col[r][2 * h + q] = odd ? cb : ca;
                  }
                }
                continue;
              }
#pragma unroll
              for (int h = 0; h < 2; h++) {
                const f2 u = (xs[h] * q5.y + eu) * iz1, v = (xs[h] * q6.y + ev) * iz1;
#pragma unroll
                for (int q = 0; q < 2; q++) {
                  const bool odd = (__builtin_amdgcn_fractf(u[q]) >= 0.5f) != (__builtin_amdgcn_fractf(v[q]) >= 0.5f);
                  const bool upd = vld && w1 > best[r][h][q];
                  best[r][h][q] = upd ? w1 : best[r][h][q];
                  col[r][2 * h + q] = upd ? (odd ? codd : ceven) : col[r][2 * h + q];
                }
              }
              continue;
            }
            plane_region(ro, rf, rr, ru, q5, q6, ysr[r], xs, best[r], col[r]);
          }
        }
      }
    }
    // ---- packed RGB8 store: 4 pixels = 3 dwords per lane and region; not waited for.  The address is a uniform
    // 64-bit image base plus a 32-bit per-lane byte offset (saddr + voffset form, no 64-bit multiplies per store)
    uint8_t* __restrict__ ibase = a.pixels + (size_t)img * a.H * a.W * 3;
    unsigned boff = ((unsigned)prow * (unsigned)a.W + (unsigned)px) * 3u;
#pragma unroll
    for (int r = 0; r < 4; r++, boff += 6u * (unsigned)a.W) {
      const int y = prow + 2 * r;
      if (y > symax || px >= a.W) continue;
      uint8_t* dst = ibase + boff;
      const unsigned c0 = col[r][0], c1 = col[r][1], c2 = col[r][2], c3 = col[r][3];
      if (fast) {
        u3 v;
        v.x = c0 | c1 << 24;
        v.y = c1 >> 8 | c2 << 16;
        v.z = c2 >> 16 | c3 << 8;
#if MIR_RENDER_NT
        __builtin_nontemporal_store(v, reinterpret_cast<u3*>(dst));
#else
        *reinterpret_cast<u3*>(dst) = v;
#endif
      } else {
        const unsigned cc[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int p = 0; p < 4; p++)
          if (px + p < a.W) { dst[3 * p] = (uint8_t)cc[p]; dst[3 * p + 1] = (uint8_t)(cc[p] >> 8); dst[3 * p + 2] = (uint8_t)(cc[p] >> 16); }
      }
    }
  }
}


// The binned pixel kernel.  Same tiling (workgroup = 4 waves stacked on a 128-pixel-wide strip, walked in 128 x 32 sub-tiles, a
// wave = a band of 128 x 8 pixels), same per-pixel arithmetic and the same stores as mir_render_kernel -- the two agree bit for
// bit -- but
//   * nothing in it waits for memory once the strip has started: the strip's list arrives with ONE vector load per wave (lane k
//     holds entry k, the loop reads it with v_readlane), the floor's scalars are fetched once per workgroup, a record is fetched
//     only for the bands its rectangle meets, and there is no workgroup barrier;
//   * depth-tested primitives are drawn in a second lane layout, four regions of 32 x 8 pixels SIDE BY SIDE (lane = 4 pixels of
//     row lane / 8), so a primitive costs arithmetic only in the 32-pixel columns its rectangle meets -- the robot's boxes are
//     20 to 60 pixels wide, and in the store layout (regions of 128 x 2 stacked) each of them ran on all 128 columns.  Their
//     colours go through a wave-private 4 KB LDS tile into the store layout (NOHIT where nothing was hit);
//   * the floor is drawn last, in the store layout, without a depth buffer: its depth and colours are row constants (one
//     reciprocal per lane and row), it fills the pixels the boxes left (all of them in the bands no rectangle meets), and the
//     boxes were depth-tested against its row depth.
#define NOHIT 0xffffffffu
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void mir_render_binned(PixArgs a) {
  __shared__ unsigned s_tile[4][8 * 128];  // per wave: 8 rows x 128 pixels of packed colours
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-contiguous walk.  Workgroups go to the 8 XCDs round robin by linear id; the workgroups of XCD j take the j-th eighth
  // of the (image, strip row, strip column) sequence in order, so each XCD's L2 streams ONE contiguous range of the output
  // (a dozen images at a time) instead of every eighth strip of all of it: tools/probes/render/store_probe.hip measures
  // 173 -> 160 us for the bare store pattern of 1024 x 480 x 640 (a linear fill: 137 us in either order, 166 - 175 us when
  // an XCD's pages are scattered).
  const unsigned wid = blockIdx.x, wseq = (wid & 7u) * (gridDim.x >> 3) + (wid >> 3);
  if (wseq >= (unsigned)a.nwg) return;
  // (within an image the strips are taken in an order rotated by 7 per image: the strips that hold the robot would otherwise
  //  come up with a fixed period in the XCD's dispatch sequence and pile onto the same few CUs)
  const unsigned nst = (unsigned)(a.nsx * a.nsy), uimg = wseq / nst, st = (wseq % nst + 7u * uimg) % nst;
  const unsigned ssx = st % (unsigned)a.nsx, ssy = st / (unsigned)a.nsx;
  const int tx0 = ssx * TW, sy0 = ssy * a.th, img = uimg;
  const int* bin = a.bins + ((size_t)uimg * nst + st) * BINW;
  const int ent = bin[lane];
  const int px = tx0 + 4 * (lane & 31);         // store layout: 4 pixels of rows prow + 2r
  const int pxa = tx0 + 4 * (lane & 7);         // box layout: 4 pixels of row wy0 + lane / 8, in the columns pxa + 32r
  const float* __restrict__ prims = a.prims + (size_t)img * a.nprim * PREC;
  const int symax = min(sy0 + a.th, a.H) - 1;
  f2 xs[2];
  xs[0] = f2{a.x0 + (float)px * a.dx, a.x0 + (float)(px + 1) * a.dx};
  xs[1] = f2{a.x0 + (float)(px + 2) * a.dx, a.x0 + (float)(px + 3) * a.dx};
  const int hdr = __builtin_amdgcn_readfirstlane(ent);
  const int cnt = hdr & 255;
  const bool floor = (hdr >> 8) & 1;
  // the floor record (primitive 0), read once per workgroup: o'z, F'z, U'z, the checker numerators and the two colours
  const cf4* frec = (const cf4*)(uintptr_t)prims;
  // (fetched whether or not the header says `floor`: record 0 always exists, and waiting for the header first would put a second
  //  memory round trip in front of every strip)
  const float f_oz = frec[0].z, f_fz = frec[1].z, f_uz = frec[3].z;
  const f4 fq5 = frec[5], fq6 = frec[6];
  const unsigned ceven = __float_as_uint(fq5.w), codd = __float_as_uint(fq6.w);
  uint8_t* __restrict__ ibase = a.pixels + (size_t)img * a.H * a.W * 3;
  unsigned* tile = s_tile[wv];
  // (a VALU instruction takes ONE scalar operand: the addends of the row expressions below live in vector registers for the whole
  //  strip instead of being moved there in every region -- the kernel is VALU-bound: 23 instructions per pixel x 4 cycles)
  float v_y0 = a.y0, v_fz = f_fz, v_5x = fq5.x, v_6x = fq6.x;
  unsigned v_sky = a.sky;
  asm volatile("" : "+v"(v_y0), "+v"(v_fz), "+v"(v_5x), "+v"(v_6x), "+v"(v_sky));

  for (int ty0 = sy0; ty0 <= symax; ty0 += 32) {
    const int wy0 = ty0 + 8 * wv;            // this wave's band of 8 rows
    if (wy0 > symax) break;
    const int prow = wy0 + (lane >> 5);      // store layout: rows prow, prow + 2, prow + 4, prow + 6
    unsigned col[4][4];
    // ---- depth-tested primitives, box layout.  Lane k tests entry k against the band; one ballot per 32-pixel column gives
    // the entries to draw there (ascending = list order), so untouched bands and columns cost a handful of instructions
    unsigned long long cm[4];
    {
      const int ymin = (ent >> 6) & 0x7ff, ymax = (ent >> 17) & 0x7ff;
      const bool on = lane >= 1 && lane <= cnt && !(ymax < wy0 || ymin > wy0 + 7);
#pragma unroll
      for (int r = 0; r < 4; r++) cm[r] = __ballot(on && ((ent >> (28 + r)) & 1));
    }
    const bool touched = (cm[0] | cm[1] | cm[2] | cm[3]) != 0ull;
    if (touched) {
      const float ya = a.y0 + (float)(wy0 + (lane >> 3)) * a.dy;
      float tb = 0.0f;  // (reciprocal depth: 0 = nothing)
      if (floor) {  // the floor's depth on this lane's row: the same expressions as in the floor pass below
        const float t1 = __builtin_amdgcn_rcpf(fmaf(ya, f_uz, f_fz)) * (-f_oz);
        tb = t1 > 1e-6f ? __builtin_amdgcn_rcpf(t1) : 0.0f;
      }
#pragma unroll
      for (int r = 0; r < 4; r++) {
        unsigned ca[4] = {NOHIT, NOHIT, NOHIT, NOHIT};
        if (cm[r]) {
          const int x = pxa + 32 * r;
          const f2 xsa[2] = {f2{a.x0 + (float)x * a.dx, a.x0 + (float)(x + 1) * a.dx}, f2{a.x0 + (float)(x + 2) * a.dx, a.x0 + (float)(x + 3) * a.dx}};
          f2 best[2] = {f2{tb, tb}, f2{tb, tb}};
          for (unsigned long long m = cm[r]; m; m &= m - 1) {
            const int e = __builtin_amdgcn_readlane(ent, __builtin_ctzll(m));
            const cf4* rec = (const cf4*)(uintptr_t)(prims + (size_t)(e & 63) * PREC);
            const f4 ro = rec[0], rf = rec[1], rr = rec[2], ru = rec[3], q5 = rec[5], q6 = rec[6];
            if ((__float_as_int(ro.w) & 255) == MIR_GEOM_BOX) box_region(ro, rf, rr, ru, rec[4], rec[7], q5, ya, xsa, best, ca);
            else plane_region(ro, rf, rr, ru, q5, q6, ya, xsa, best, ca);
          }
        }
        // box layout -> store layout through the wave's LDS tile (rows of 128 pixels)
        *reinterpret_cast<uint4*>(tile + (lane >> 3) * 128 + 32 * r + 4 * (lane & 7)) = make_uint4(ca[0], ca[1], ca[2], ca[3]);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // (no barrier: a wave's LDS operations complete in order)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const uint4 v = *reinterpret_cast<const uint4*>(tile + (2 * r + (lane >> 5)) * 128 + 4 * (lane & 31));
        col[r][0] = v.x; col[r][1] = v.y; col[r][2] = v.z; col[r][3] = v.w;
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // ---- floor pass and packed RGB8 store (4 pixels = 3 dwords per lane and region; not waited for), store layout.  Two copies:
    // bands no primitive touched (most) assign the floor colour, the others fill the pixels the boxes left.
    auto finish = [&](auto touched_c) {
      constexpr bool TOUCHED = decltype(touched_c)::value;
      unsigned boff = ((unsigned)prow * (unsigned)a.W + (unsigned)px) * 3u;
#pragma unroll
      for (int r = 0; r < 4; r++, boff += 6u * (unsigned)a.W) {
        unsigned fc[4];
        if (floor) {
          const float y = fmaf((float)(prow + 2 * r), a.dy, v_y0);
          const float ez = fmaf(y, f_uz, v_fz), eu = fmaf(y, fq5.z, v_5x), ev = fmaf(y, fq6.z, v_6x);
          const float iz1 = __builtin_amdgcn_rcpf(ez);
          const bool vld = iz1 * (-f_oz) > 1e-6f;
          const unsigned c0 = vld ? ceven : v_sky, c1 = vld ? codd : v_sky;
#pragma unroll
          for (int h = 0; h < 2; h++) {
            const f2 u = (xs[h] * fq5.y + eu) * iz1, v = (xs[h] * fq6.y + ev) * iz1;
#pragma unroll
            for (int q = 0; q < 2; q++) {
              const bool odd = (__builtin_amdgcn_fractf(u[q]) >= 0.5f) != (__builtin_amdgcn_fractf(v[q]) >= 0.5f);
              fc[2 * h + q] = odd ? c1 : c0;
            }
          }
        } else {
#pragma unroll
          for (int p = 0; p < 4; p++) fc[p] = v_sky;
        }
        if (prow + 2 * r > symax || px >= a.W) continue;
        unsigned c[4];
#pragma unroll
        for (int p = 0; p < 4; p++) c[p] = TOUCHED ? (col[r][p] == NOHIT ? fc[p] : col[r][p]) : fc[p];
        u3 v;
        v.x = c[0] | c[1] << 24;
        v.y = c[1] >> 8 | c[2] << 16;
        v.z = c[2] >> 16 | c[3] << 8;
        *reinterpret_cast<u3*>(ibase + boff) = v;
      }
    };
    if (touched) finish(std::true_type{});
    else finish(std::false_type{});
  }
}

// ---- the global view of MANY envs (camera_capture_mode="global", the registry's default; GenesisEnv.render()) ------------------------
// One image, B x ngeom primitives, each a few hundred pixels of it.  The tiled kernels walk a tile's whole primitive list: 25 workgroups
// of a 480 x 640 image each cull 82 000 records of 4096 envs -- 17.8 ms.  Here k_render_setup lists the boxes that can be on screen as work
// items (a box's screen rectangle in shares of about four 32 x 8 blocks) and every item is drawn by a workgroup (lane = pixel of a
// block) into a 64-bit depth / colour buffer with atomic max on  w << 32 | colour
// (w = reciprocal depth > 0: its float bits order like unsigned integers), then one pass resolves the buffer against the planes
// (drawn once: env 0's) and the sky and stores RGB8.  Same per-pixel expressions as box_bounds / plane_region / the floor path of
// mir_render_kernel, so the two agree except where two surfaces tie in depth to the last bit (the tiled kernels keep the first one
// in list order, the maximum here keeps the larger colour word).
struct SplatArgs {
  const float* prims;
  unsigned long long* zbuf;  // (H, W): w bits << 32 | packed RGB8, 0 = nothing
  unsigned* vis;             // [0] work items, [1] [2] mask of the plane geoms, [VIS_HDR ..] box index | share << 22 | (shares - 1) << 27 (k_render_setup)
  uint8_t* pixels;
  int W, H, nprim, ngeom;
  float x0, dx, y0, dy;
  unsigned sky;
};

// one box, the 32 x 8 blocks b = first, first + stride, ... of its rectangle (row-major), lane = pixel
__device__ __forceinline__ void splat_box(const SplatArgs& a, unsigned iprim, int first, int stride) {
  const float* prim = a.prims + (size_t)iprim * PREC;
  const int tid = threadIdx.x;
  const cf4* rec = (const cf4*)(uintptr_t)prim;
  const f4 q0 = rec[0], q1 = rec[1], q2 = rec[2], q3 = rec[3], q4 = rec[4];
  const int tw = __float_as_int(q0.w);
  if ((tw & 255) != MIR_GEOM_BOX) return;
  const int xmin = __float_as_int(q1.w), xmax = min(__float_as_int(q2.w), a.W - 1), ymin = __float_as_int(q3.w), ymax = min(__float_as_int(q4.w), a.H - 1);
  if (xmin > xmax || ymin > ymax) return;
  const int nbx = (xmax >> 5) - (xmin >> 5) + 1, nb = nbx * ((ymax >> 3) - (ymin >> 3) + 1);
  const f4 q5 = rec[5], q7 = rec[7];
  const int nup = tw >> 8;
  const unsigned c0 = __float_as_uint(q5.x), c1 = __float_as_uint(q5.y), c2 = __float_as_uint(q5.z);
  for (int blk = first; blk < nb; blk += stride) {
    const int by = (ymin & ~7) + 8 * (blk / nbx), bx = (xmin & ~31) + 32 * (blk % nbx);
    const int py = by + (tid >> 5), px = bx + (tid & 31);
    if (px < xmin || px > xmax || py < ymin || py > ymax) continue;
    const float ys = a.y0 + (float)py * a.dy;
    const float e0 = fmaf(ys, q2.x, q0.x), e1 = fmaf(ys, q2.y, q0.y), e2 = fmaf(ys, q2.z, q0.z);
    const float e3 = fmaf(ys, q7.x, q3.x), e4 = fmaf(ys, q7.y, q3.y), e5 = fmaf(ys, q7.z, q3.z);
    const float xs = a.x0 + (float)px * a.dx;
    const float v0 = fmaf(xs, q1.x, e0), v1 = fmaf(xs, q1.y, e1), v2 = fmaf(xs, q1.z, e2);
    const float v3 = fmaf(xs, q4.x, e3), v4 = fmaf(xs, q4.y, e4), v5 = fmaf(xs, q4.z, e5);
    float hi, lo = fmaxf(fmaxf(v3, v4), v5);
    unsigned c;
    if (nup == 3) { hi = fminf(fminf(v0, v1), v2); c = hi == v0 ? c0 : (hi == v1 ? c1 : c2); }
    else if (nup == 2) { hi = fminf(v0, v1); lo = fmaxf(lo, v2); c = hi == v0 ? c0 : c1; }
    else { hi = v0; lo = fmaxf(lo, fmaxf(v1, v2)); c = c0; }
    if (lo <= hi && hi > 0.0f)
      atomicMax(a.zbuf + (size_t)py * a.W + px, (unsigned long long)__float_as_uint(hi) << 32 | (unsigned long long)c);
  }
}

#define GLOBAL_SPLAT_GRID 2048
__global__ __launch_bounds__(256) void k_global_splat(SplatArgs a) {
  // (a WAVE per box with the waves taking boxes round robin is twice as slow: 1.0 ms at 4096 envs -- the few boxes with large rectangles
  //  decide, and they want lanes.)  The workgroups walk the LIST of work items k_render_setup made of the boxes that can be on screen:
  //  one workgroup per box of the scene was 57 000 workgroups at 4096 envs, 99 % of which found an empty rectangle and left (15.7 us
  //  of dispatch), plus a second launch for the boxes with large rectangles (4.6 us).
  const unsigned n = a.vis[0];
  for (unsigned k = blockIdx.x; k < n; k += GLOBAL_SPLAT_GRID) {
    const unsigned it = a.vis[VIS_HDR + k];
    splat_box(a, it & 0x3fffffu, (int)(it >> 22 & 31u), (int)(it >> 27) + 1);
  }
}

// lane = 4 consecutive pixels of one row (one dwordx3 store when the width allows)
__global__ __launch_bounds__(256) void k_global_resolve(SplatArgs a) {
  // The buffer is handed back CLEAN: every key read here is zeroed again (the few that were written: most pixels show the floor), and
  // the work-item count goes back to zero -- the next render starts without a memset of 2.4 MB (a launch of 5 us).
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) a.vis[0] = 0u;  // (the list's entries are overwritten before they are read)
  const int px = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4, py = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (px >= a.W || py >= a.H) return;
  const float ys = a.y0 + (float)py * a.dy;
  float best[4];
  unsigned col[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const unsigned long long key = px + p < a.W ? a.zbuf[(size_t)py * a.W + px + p] : 0ull;
    if (key) a.zbuf[(size_t)py * a.W + px + p] = 0ull;
    best[p] = __uint_as_float((unsigned)(key >> 32));
    col[p] = key ? (unsigned)key : a.sky;
  }
  // the planes: env 0's records (the others carry an empty rectangle), by the mask the setup kernel left -- looking through all of env 0's
  // records for them was a dozen dependent scalar loads in front of every pixel
  for (unsigned long long pm = (unsigned long long)a.vis[1] | (unsigned long long)a.vis[2] << 32; pm; pm &= pm - 1ull) {
    const int g = __builtin_ctzll(pm);
    const cf4* rec = (const cf4*)(uintptr_t)(a.prims + (size_t)g * PREC);
    const f4 ro = rec[0], rf = rec[1], rr = rec[2], ru = rec[3], rh = rec[4];
    if ((__float_as_int(ro.w) & 255) == MIR_GEOM_BOX || __float_as_int(rf.w) > __float_as_int(rr.w) || __float_as_int(ru.w) > __float_as_int(rh.w)) continue;
    const f4 q5 = rec[5], q6 = rec[6];
    const unsigned ceven = __float_as_uint(q5.w), codd = __float_as_uint(q6.w);
    const float ez = fmaf(ys, ru.z, rf.z), eu = fmaf(ys, q5.z, q5.x), ev = fmaf(ys, q6.z, q6.x);
    if (rr.z == 0.0f) {  // (the row-constant depth of mir_render_kernel's floor path: the same expressions)
      const float iz1 = __builtin_amdgcn_rcpf(ez);
      const float t1 = iz1 * (-ro.z);
      const bool vld = t1 > 1e-6f;
      const float w1 = __builtin_amdgcn_rcpf(t1);
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const float xs = a.x0 + (float)(px + p) * a.dx;
        const float u = (xs * q5.y + eu) * iz1, v = (xs * q6.y + ev) * iz1;
        const bool odd = (__builtin_amdgcn_fractf(u) >= 0.5f) != (__builtin_amdgcn_fractf(v) >= 0.5f);
        const bool upd = vld && w1 > best[p];
        best[p] = upd ? w1 : best[p];
        col[p] = upd ? (odd ? codd : ceven) : col[p];
      }
    } else {
      const float nio = __builtin_amdgcn_rcpf(-ro.z);
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const float xs = a.x0 + (float)(px + p) * a.dx;
        const float dz = xs * rr.z + ez;
        const float iz = __builtin_amdgcn_rcpf(dz), w = dz * nio;
        const float u = (xs * q5.y + eu) * iz, v = (xs * q6.y + ev) * iz;
        const bool odd = (__builtin_amdgcn_fractf(u) >= 0.5f) != (__builtin_amdgcn_fractf(v) >= 0.5f);
        const bool upd = w < 1e6f && w > best[p];
        best[p] = upd ? w : best[p];
        col[p] = upd ? (odd ? codd : ceven) : col[p];
      }
    }
  }
  uint8_t* dst = a.pixels + ((size_t)py * a.W + px) * 3;
  if ((a.W & 3) == 0) {
    u3 v;
    v.x = col[0] | col[1] << 24;
    v.y = col[1] >> 8 | col[2] << 16;
    v.z = col[2] >> 16 | col[3] << 8;
    *reinterpret_cast<u3*>(dst) = v;
  } else {
#pragma unroll
    for (int p = 0; p < 4; p++)
      if (px + p < a.W) { dst[3 * p] = (uint8_t)col[p]; dst[3 * p + 1] = (uint8_t)(col[p] >> 8); dst[3 * p + 2] = (uint8_t)(col[p] >> 16); }
  }
}

void norm3(const double* v, double* o) {
  const double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  for (int k = 0; k < 3; k++) o[k] = n > 0 ? v[k] / n : 0.0;
}
void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

}  // namespace

extern "C" int mir_visual_sizeof(void) { return (int)sizeof(MirVisualSpec); }

// shared body of mir_render / mir_render_cams: cam_pos == null -> one camera (cam->pos / lookat / up) for all images
static int render_impl(MirHandle h, const MirCameraSpec* cam, const MirVisualSpec* vis, int32_t mode, const float* env_offset,
                       const float* cam_pos, const float* cam_look, const float* cam_up, uint8_t* pixels, void* stream) {
  if (!h || !cam || !vis || !pixels) return mir_set_error(MIR_E_INVALID, "mir_render: null argument");
  if (vis->struct_size != (int)sizeof(MirVisualSpec)) return mir_set_error(MIR_E_INVALID, "mir_render: MirVisualSpec size mismatch");
  if (cam->width <= 0 || cam->height <= 0 || !(cam->fov_deg > 0.0 && cam->fov_deg < 180.0))
    return mir_set_error(MIR_E_INVALID, "mir_render: bad camera (res / fov)");
  if (mode != MIR_RENDER_PER_ENV && mode != MIR_RENDER_GLOBAL) return mir_set_error(MIR_E_INVALID, "mir_render: unknown mode");
  if (!(vis->checker_size > 0.0)) return mir_set_error(MIR_E_INVALID, "mir_render: checker_size must be > 0");
  // Addressing limits of the pixel kernel: the image index is the grid's z coordinate (<= 65535 images per call), a pixel's byte
  // offset INSIDE its image is 32 bits (the image base is 64 bits: B x H x W x 3 may exceed 2^31 and 2^32; tests/test_gpu_render.py
  // renders 2400 x 480 x 640 x 3 = 3.3 GB)
  if ((unsigned long long)cam->width * (unsigned long long)cam->height * 3ull >= (1ull << 32))
    return mir_set_error(MIR_E_CAPACITY, "mir_render: one image must stay below 2^32 bytes (width x height x 3)");
  if (mode == MIR_RENDER_PER_ENV && h->B > 65535) return mir_set_error(MIR_E_CAPACITY, "mir_render: at most 65535 per-env images per call");
  double f[3] = {1, 0, 0}, r[3] = {0, 1, 0}, u[3] = {0, 0, 1};
  if (!cam_pos) {
    double d[3] = {cam->lookat[0] - cam->pos[0], cam->lookat[1] - cam->pos[1], cam->lookat[2] - cam->pos[2]};
    norm3(d, f);
    if (f[0] == 0 && f[1] == 0 && f[2] == 0) return mir_set_error(MIR_E_INVALID, "mir_render: camera pos == lookat");
    double rr[3];
    cross3(f, cam->up, rr);
    if (rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2] < 1e-24) {  // view parallel to up (a top-down camera with up = +z:
      const double ey[3] = {0, 1, 0}, ex[3] = {1, 0, 0};           // cube_stack_kitchen_batch.py:175): fall back to +y, then +x
      cross3(f, ey, rr);
      if (rr[0] * rr[0] + rr[1] * rr[1] + rr[2] * rr[2] < 1e-24) cross3(f, ex, rr);
    }
    norm3(rr, r);
    cross3(r, f, u);
  }
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != h->device) (void)hipSetDevice(h->device);
  int rc = MIR_OK;
  hipStream_t st = (hipStream_t)stream;
  const int ng = h->ngeom, B = h->B;
  do {
    if (!h->prims) {
      hipError_t e = hipMalloc((void**)&h->prims, (size_t)B * ng * PREC * sizeof(float));
      if (e != hipSuccess) { rc = mir_set_error(MIR_E_HIP, hipGetErrorString(e)); break; }
    }
    h->poses_live = 1;  // (from now on the step launches keep h->poses up to date themselves)
    if (!h->poses_current && (rc = mir_refresh_poses(h, stream)) != MIR_OK) break;
    SetupArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.geom = h->dgeom; sa.poses = h->poses; sa.pst = h->pt.pst; sa.env_offset = env_offset; sa.prims = h->prims;
    sa.cam_pos = cam_pos; sa.cam_look = cam_look; sa.cam_up = cam_up;
    for (int k = 0; k < 3; k++) sa.up0[k] = (float)cam->up[k];
    const double ty = std::tan(0.5 * cam->fov_deg * M_PI / 180.0), tx = ty * (double)cam->width / (double)cam->height;
    for (int k = 0; k < 3; k++) { sa.cam.pos[k] = (float)cam->pos[k]; sa.cam.f[k] = (float)f[k]; sa.cam.r[k] = (float)r[k]; sa.cam.u[k] = (float)u[k]; }
    sa.cam.tanx = (float)tx; sa.cam.tany = (float)ty; sa.cam.W = cam->width; sa.cam.H = cam->height;
    double L[3];
    norm3(vis->light_dir, L);
    for (int k = 0; k < 3; k++) sa.light[k] = (float)L[k];
    for (int g = 0; g < ng; g++)
      for (int k = 0; k < 3; k++) sa.rgb[g][k] = (float)vis->geom_rgb[g][k];
    sa.amb = (float)vis->ambient; sa.dif = (float)vis->diffuse; sa.inv_chk = (float)(1.0 / vis->checker_size);
    for (int k = 0; k < 3; k++) { sa.chk[0][k] = (float)vis->checker_rgb[0][k]; sa.chk[1][k] = (float)vis->checker_rgb[1][k]; }
    sa.B = B; sa.ngeom = ng; sa.global_mode = mode == MIR_RENDER_GLOBAL; sa.vis = nullptr;
    PixArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.prims = h->prims; pa.pixels = pixels; pa.W = cam->width; pa.H = cam->height;
    pa.nprim = mode == MIR_RENDER_GLOBAL ? B * ng : ng;
    pa.dx = (float)(2.0 * tx / cam->width); pa.x0 = (float)(-tx + tx / cam->width);
    pa.dy = (float)(-2.0 * ty / cam->height); pa.y0 = (float)(ty - ty / cam->height);
    {
      auto u8 = [](double c) { return (unsigned)(std::fmin(std::fmax(c, 0.0), 1.0) * 255.0 + 0.5); };
      pa.sky = u8(vis->sky_rgb[0]) | u8(vis->sky_rgb[1]) << 8 | u8(vis->sky_rgb[2]) << 16;
    }
    pa.th = h->render_th > 0 ? h->render_th : TH;
    const int nimg = mode == MIR_RENDER_GLOBAL ? 1 : B;
    // Per-env images (short primitive lists) take the binned kernel; the global view of all envs (one long list), widths
    // that are not a multiple of 4 pixels and images taller than the 11-bit row fields keep the generic kernel.
    const bool binned = mode == MIR_RENDER_PER_ENV && ng < BINW && (cam->width & 3) == 0 && cam->height <= 2048 && !h->render_generic;
    if (binned) {
      pa.th = h->render_th > 0 ? h->render_th : TH_BINNED;
      const int nsx = (cam->width + TW - 1) / TW, nsy = (cam->height + pa.th - 1) / pa.th;
      const size_t need = (size_t)nimg * nsx * nsy * BINW;
      if (need > h->bins_cap) {
        if (h->bins) (void)hipFree(h->bins);
        h->bins = nullptr; h->bins_cap = 0;
        hipError_t e = hipMalloc((void**)&h->bins, need * sizeof(int));
        if (e != hipSuccess) { rc = mir_set_error(MIR_E_HIP, hipGetErrorString(e)); break; }
        h->bins_cap = need;
      }
      BinArgs ba;
      ba.prims = h->prims; ba.bins = h->bins; ba.W = cam->width; ba.H = cam->height; ba.nprim = pa.nprim; ba.th = pa.th;
      ba.nsx = nsx; ba.nsy = nsy; ba.nimg = nimg;
      hipLaunchKernelGGL(k_render_setup<true>, dim3((B * 64 + 255) / 256), dim3(256), 0, st, sa, ba);
      pa.bins = h->bins;
      pa.nsx = nsx; pa.nsy = nsy; pa.nwg = nimg * nsx * nsy;
      hipLaunchKernelGGL(mir_render_binned, dim3((unsigned)((pa.nwg + 7) & ~7)), dim3(256), 0, st, pa);
    } else if (mode == MIR_RENDER_GLOBAL && pa.nprim > 512 && !h->render_generic) {
      // the global view of many envs: one workgroup per box into a depth / colour buffer, then a resolve pass (see k_global_splat)
      const size_t npix = (size_t)cam->width * cam->height, need = npix;
      if (need > h->zbuf_cap) {
        if (h->zbuf) (void)hipFree(h->zbuf);
        h->zbuf = nullptr; h->zbuf_cap = 0;
        hipError_t e = hipMalloc((void**)&h->zbuf, need * sizeof(unsigned long long));
        if (e != hipSuccess) { rc = mir_set_error(MIR_E_HIP, hipGetErrorString(e)); break; }
        h->zbuf_cap = need;
        // cleared ONCE: k_global_resolve leaves the pixels and the list of large boxes clean behind every render (whatever the camera's
        // resolution: it zeroes exactly what that render wrote)
        (void)hipMemsetAsync(h->zbuf, 0, need * sizeof(unsigned long long), st);
      }
      const size_t vneed = (size_t)GLOBAL_ITEMS_MAX * pa.nprim + VIS_HDR;  // (the worst case: every box fills the screen; touched only where written)
      if (pa.nprim >= (1 << 22)) { rc = mir_set_error(MIR_E_CAPACITY, "mir_render: more than 2^22 primitives in a global view"); break; }
      if (vneed > h->vis_cap) {
        if (h->vis) (void)hipFree(h->vis);
        h->vis = nullptr; h->vis_cap = 0;
        hipError_t e = hipMalloc((void**)&h->vis, vneed * sizeof(unsigned));
        if (e != hipSuccess) { rc = mir_set_error(MIR_E_HIP, hipGetErrorString(e)); break; }
        h->vis_cap = vneed;
        // (the item count -- k_global_resolve puts it back to zero behind every render -- and the mask of the scene's planes, a property
        //  of the scene that the setup kernel ORs in again with every render)
        (void)hipMemsetAsync(h->vis, 0, VIS_HDR * sizeof(unsigned), st);
      }
      sa.vis = h->vis;
      hipLaunchKernelGGL(k_render_setup<false>, dim3((B * ng + 255) / 256), dim3(256), 0, st, sa, BinArgs{});
      SplatArgs sp;
      sp.prims = h->prims; sp.zbuf = h->zbuf; sp.vis = h->vis; sp.pixels = pixels; sp.W = cam->width; sp.H = cam->height; sp.nprim = pa.nprim; sp.ngeom = ng;
      sp.x0 = pa.x0; sp.dx = pa.dx; sp.y0 = pa.y0; sp.dy = pa.dy; sp.sky = pa.sky;
      hipLaunchKernelGGL(k_global_splat, dim3(GLOBAL_SPLAT_GRID), dim3(256), 0, st, sp);
      hipLaunchKernelGGL(k_global_resolve, dim3((cam->width + 255) / 256, (cam->height + 3) / 4), dim3(256), 0, st, sp);
    } else {
      hipLaunchKernelGGL(k_render_setup<false>, dim3((B * ng + 255) / 256), dim3(256), 0, st, sa, BinArgs{});
      hipLaunchKernelGGL(mir_render_kernel, dim3((cam->width + TW - 1) / TW, (cam->height + pa.th - 1) / pa.th, nimg), dim3(256), 0, st, pa);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = mir_set_error(MIR_E_HIP, hipGetErrorString(e));
  } while (0);
  if (prev != h->device && prev >= 0) (void)hipSetDevice(prev);
  return rc;
}

extern "C" int mir_render(MirHandle h, const MirCameraSpec* cam, const MirVisualSpec* vis, int32_t mode, const float* env_offset,
                          uint8_t* pixels, void* stream) {
  return render_impl(h, cam, vis, mode, env_offset, nullptr, nullptr, nullptr, pixels, stream);
}

extern "C" int mir_debug_render_path(MirHandle h, int32_t generic, int32_t strip_rows) {
  if (!h || strip_rows < 0 || (strip_rows & 31)) return mir_set_error(MIR_E_INVALID, "mir_debug_render_path: strip_rows must be a multiple of 32");
  h->render_generic = generic != 0;
  h->render_th = strip_rows;
  return MIR_OK;
}

extern "C" int mir_render_cams(MirHandle h, const MirCameraSpec* cam, const MirVisualSpec* vis, const float* cam_pos, const float* cam_lookat,
                               const float* cam_up, uint8_t* pixels, void* stream) {
  if (!cam_pos || !cam_lookat) return mir_set_error(MIR_E_INVALID, "mir_render_cams: null camera arrays");
  return render_impl(h, cam, vis, MIR_RENDER_PER_ENV, nullptr, cam_pos, cam_lookat, cam_up, pixels, stream);
}

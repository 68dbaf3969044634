// mir_render.hip — batched tiled rasteriser for the `pixels` observation and render() (SURVEY.md 8f-2,
// BASELINE.json configs[4]).
//
// What it replaces: the reference's per-env Python loop of cam.set_pose + cam.render()
// (/root/reference/gym_genesis/tasks/franka/cube_pick.py:159-180) and GenesisEnv.render
// (/root/reference/gym_genesis/env.py:97-98), which drive Genesis's OpenGL rasteriser B times per step.
//
// MI355X mapping.  At B=1024, 480x640 the output is 943 MB per step, so the kernel is HBM-WRITE-bound
// by construction; everything else is arranged to stay below that:
//   * k_render_setup (one thread per (env, geom)): world pose of every primitive from the step kernel's
//     FK cache, the camera expressed in the primitive's frame (origin o', and the images F', R', U' of
//     the camera basis, so a pixel's ray in that frame is d' = F' + x R' + y U': 6 FMAs, no matrix
//     product), the per-face light terms, and a conservative screen rectangle.  128 B per primitive.
//   * mir_render_kernel: workgroup = 256 threads = one 128 x 32 pixel tile of one image; the tile's
//     primitives are culled by rectangle into LDS (ordered ballot compaction) and read from there as
//     wave-uniform broadcast ds_reads; a thread owns 4 consecutive pixels of a row (4 rows per thread),
//     so a wave finishes two 384-byte row segments at a time and stores them as packed RGB8 with one
//     global_store_dwordx3 per lane: every 128-byte line of the image is written exactly once, whole.
//     Primitives are additionally culled per wave (2 rows) by a scalar branch.
//   * ray/box in the box frame is a 3-slab test; depth order is resolved per pixel (strict <, list in
//     ascending primitive order => deterministic); shading is deferred to one lookup per pixel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>
#include <cstring>

#include "mir_model.h"
#include "mir_scene.h"

#define TW 128   /* tile width, pixels  */
#define TH 32    /* tile height, pixels */
#define RCAP 64  /* primitive records resident in LDS per round */
#define PREC 32  /* floats per primitive record */

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));
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

struct SetupArgs {
  const DevModel* model;
  const float* poses;       // (B, 2, 16, 4) FK cache
  const float* env_offset;  // (B, 3) or null
  float* prims;             // (B, ngeom, PREC)
  Cam cam;
  float light[3];
  float rgb[MIR_MAX_GEOM][3];
  int B, ngeom, global_mode;
};

struct PixArgs {
  const float* prims;
  uint8_t* pixels;
  int W, H, nprim;  // primitives per image
  float x0, dx, y0, dy;  // image-plane coordinates of pixel (i, j): x0 + i dx, y0 + j dy
  float amb, dif, inv_chk;
  float sky[3], chk[2][3];
};

// ---- per-(env, geom) primitive record ------------------------------------------------------------
//  [0..2] o'  [3] type | [4..6] F' [7] xmin | [8..10] R' [11] xmax | [12..14] U' [15] ymin |
//  [16..18] half extents [19] ymax | [20..22] albedo | [24..26] a_k . light
__global__ void k_render_setup(SetupArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.B * a.ngeom) return;
  const int e = i / a.ngeom, g = i % a.ngeom;
  const DevModel* __restrict__ m = a.model;
  const int b = m->g_body[g], type = m->g_type[g];
  const float* pp = a.poses + ((size_t)e * 2 * MIR_G + b) * 4;
  const V3 xp = {pp[0], pp[1], pp[2]};
  const float* qq = pp + 4 * MIR_G;
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
  const V3 cp = {a.cam.pos[0], a.cam.pos[1], a.cam.pos[2]}, cf = {a.cam.f[0], a.cam.f[1], a.cam.f[2]};
  const V3 cr = {a.cam.r[0], a.cam.r[1], a.cam.r[2]}, cu = {a.cam.u[0], a.cam.u[1], a.cam.u[2]};
  const V3 L = {a.light[0], a.light[1], a.light[2]};
  const V3 rel = cp - c;
  const V3 h = {m->g_size[g][0], m->g_size[g][1], m->g_size[g][2]};
  int xmin = 0, xmax = a.cam.W - 1, ymin = 0, ymax = a.cam.H - 1;
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
    if (behind == 8) { xmin = 1; xmax = 0; ymin = 1; ymax = 0; }  // wholly behind the camera: never drawn
    else if (behind == 0) {
      xmin = max(0, (int)fmaxf(floorf(pxmin) - 1.0f, -1.0f)); xmax = min(a.cam.W - 1, (int)fminf(ceilf(pxmax) + 1.0f, 1e9f));
      ymin = max(0, (int)fmaxf(floorf(pymin) - 1.0f, -1.0f)); ymax = min(a.cam.H - 1, (int)fminf(ceilf(pymax) + 1.0f, 1e9f));
    }  // straddling the camera plane: keep the full screen
  } else if (a.global_mode && e > 0) {
    xmin = 1; xmax = 0; ymin = 1; ymax = 0;  // an unbounded plane is drawn once (env 0's)
  }
  float* o = a.prims + (size_t)i * PREC;
  f4* o4 = reinterpret_cast<f4*>(o);
  o4[0] = f4{dot(ax[0], rel), dot(ax[1], rel), dot(ax[2], rel), __int_as_float(type)};
  o4[1] = f4{dot(ax[0], cf), dot(ax[1], cf), dot(ax[2], cf), __int_as_float(xmin)};
  o4[2] = f4{dot(ax[0], cr), dot(ax[1], cr), dot(ax[2], cr), __int_as_float(xmax)};
  o4[3] = f4{dot(ax[0], cu), dot(ax[1], cu), dot(ax[2], cu), __int_as_float(ymin)};
  o4[4] = f4{h.x, h.y, h.z, __int_as_float(ymax)};
  o4[5] = f4{a.rgb[g][0], a.rgb[g][1], a.rgb[g][2], 0.0f};
  o4[6] = f4{dot(ax[0], L), dot(ax[1], L), dot(ax[2], L), 0.0f};
  o4[7] = f4{0, 0, 0, 0};
}

__device__ __forceinline__ unsigned to_u8(float c) {
  return (unsigned)(fminf(fmaxf(c, 0.0f), 1.0f) * 255.0f + 0.5f);
}

// ---- pixels --------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mir_render_kernel(PixArgs a) {
  __shared__ __attribute__((aligned(16))) f4 s_rec[RCAP][5];  // o'|type, F'|xmin, R'|xmax, U'|ymin, h|ymax
  __shared__ int s_ids[256];
  __shared__ int s_wcnt[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH, img = blockIdx.z;
  const int px = tx0 + 4 * (tid & 31);
  const int prow = ty0 + (tid >> 5);  // rows prow, prow + 8, prow + 16, prow + 24
  const float* __restrict__ prims = a.prims + (size_t)img * a.nprim * PREC;
  const int txmax = min(tx0 + TW, a.W) - 1, tymax = min(ty0 + TH, a.H) - 1;

  float best[4][4];
  int code[4][4];
#pragma unroll
  for (int r = 0; r < 4; r++)
#pragma unroll
    for (int p = 0; p < 4; p++) { best[r][p] = 3e38f; code[r][p] = -1; }
  float xs[4];
#pragma unroll
  for (int p = 0; p < 4; p++) xs[p] = a.x0 + (float)(px + p) * a.dx;

  for (int base = 0; base < a.nprim; base += 256) {
    // ---- cull this chunk of primitives against the tile; ordered compaction of the survivors
    const int pi = base + tid;
    bool hit = false;
    if (pi < a.nprim) {
      const float* rec = prims + (size_t)pi * PREC;
      const int xmin = __float_as_int(rec[7]), xmax = __float_as_int(rec[11]), ymin = __float_as_int(rec[15]), ymax = __float_as_int(rec[19]);
      hit = xmin <= txmax && xmax >= tx0 && ymin <= tymax && ymax >= ty0;
    }
    const unsigned long long bal = __ballot(hit);
    if (lane == 0) s_wcnt[wv] = __popcll(bal);
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = s_wcnt[k];
      off += k < wv ? c : 0;
      total += c;
    }
    if (hit) s_ids[off + __popcll(bal & ((1ull << lane) - 1ull))] = pi;
    __syncthreads();
    for (int r0 = 0; r0 < total; r0 += RCAP) {
      const int nrec = min(RCAP, total - r0);
      // cooperative copy of the surviving records (first 5 quads of each) into LDS
      for (int q = tid; q < nrec * 5; q += 256) {
        const int k = q / 5, j = q - 5 * k;
        s_rec[k][j] = reinterpret_cast<const f4*>(prims + (size_t)s_ids[r0 + k] * PREC)[j];
      }
      __syncthreads();
      for (int k = 0; k < nrec; k++) {
        const f4 ro = s_rec[k][0], rf = s_rec[k][1], rr = s_rec[k][2], ru = s_rec[k][3], rh = s_rec[k][4];
        const int id = s_ids[r0 + k];
        const int xmin = __float_as_int(rf.w), xmax = __float_as_int(rr.w), ymin = __float_as_int(ru.w), ymax = __float_as_int(rh.w);
        const bool isbox = __float_as_int(ro.w) == MIR_GEOM_BOX;
#pragma unroll
        for (int r = 0; r < 4; r++) {
          // wave-uniform row cull: this wave's two rows of iteration r
          const int wy = ty0 + 8 * r + 2 * wv;
          if (ymax < wy || ymin > wy + 1) continue;
          const int y = prow + 8 * r;
          const float ys = a.y0 + (float)y * a.dy;
          const float bx = fmaf(ys, ru.x, rf.x), by = fmaf(ys, ru.y, rf.y), bz = fmaf(ys, ru.z, rf.z);
#pragma unroll
          for (int p = 0; p < 4; p++) {
            const float dxp = fmaf(xs[p], rr.x, bx), dyp = fmaf(xs[p], rr.y, by), dzp = fmaf(xs[p], rr.z, bz);
            float t;
            int face;
            bool ok;
            if (isbox) {
              const float ix = __builtin_amdgcn_rcpf(dxp), iy = __builtin_amdgcn_rcpf(dyp), iz = __builtin_amdgcn_rcpf(dzp);
              const float ax1 = (-rh.x - ro.x) * ix, ax2 = (rh.x - ro.x) * ix;
              const float ay1 = (-rh.y - ro.y) * iy, ay2 = (rh.y - ro.y) * iy;
              const float az1 = (-rh.z - ro.z) * iz, az2 = (rh.z - ro.z) * iz;
              const float nx = fminf(ax1, ax2), ny = fminf(ay1, ay2), nz = fminf(az1, az2);
              const float tn = fmaxf(fmaxf(nx, ny), nz);
              const float tf = fminf(fminf(fmaxf(ax1, ax2), fmaxf(ay1, ay2)), fmaxf(az1, az2));
              ok = tn <= tf && tn > 1e-6f && px + p >= xmin && px + p <= xmax;
              t = tn;
              face = tn == nx ? 0 : (tn == ny ? 1 : 2);
              const float dk = face == 0 ? dxp : (face == 1 ? dyp : dzp);
              face |= dk > 0.0f ? 4 : 0;  // bit 2: the ray travels along +axis, so the face normal is -axis
            } else {
              t = -ro.z * __builtin_amdgcn_rcpf(dzp);
              ok = t > 1e-6f && t < 1e30f;
              face = 2 | (ro.z < 0.0f ? 4 : 0);
            }
            if (ok && t < best[r][p]) { best[r][p] = t; code[r][p] = id * 8 + face; }
          }
        }
      }
      __syncthreads();
    }
  }

  // ---- deferred shading + packed RGB8 store -------------------------------------------------------
  const bool fast = (a.W & 3) == 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int y = prow + 8 * r;
    if (y >= a.H || px >= a.W) continue;
    const float ys = a.y0 + (float)y * a.dy;
    unsigned cr[4], cg[4], cb[4];
#pragma unroll
    for (int p = 0; p < 4; p++) {
      float R = a.sky[0], Gc = a.sky[1], Bc = a.sky[2];
      const int cd = code[r][p];
      if (cd >= 0) {
        const int id = cd >> 3, k = cd & 3;
        const f4* rec = reinterpret_cast<const f4*>(prims + (size_t)id * PREC);
        const f4 ro = rec[0];
        f4 alb = rec[5];
        const f4 lk = rec[6];
        float nl = k == 0 ? lk.x : (k == 1 ? lk.y : lk.z);
        nl = (cd & 4) ? -nl : nl;
        if (__float_as_int(ro.w) == MIR_GEOM_PLANE) {
          const f4 rf = rec[1], rr = rec[2], ru = rec[3];
          const float t = best[r][p];
          const float dxp = fmaf(xs[p], rr.x, fmaf(ys, ru.x, rf.x)), dyp = fmaf(xs[p], rr.y, fmaf(ys, ru.y, rf.y));
          const float hu = fmaf(t, dxp, ro.x), hv = fmaf(t, dyp, ro.y);
          const int par = ((int)floorf(hu * a.inv_chk) + (int)floorf(hv * a.inv_chk)) & 1;
          alb = f4{a.chk[par][0], a.chk[par][1], a.chk[par][2], 0.0f};
        }
        const float sh = a.amb + a.dif * fmaxf(nl, 0.0f);
        R = alb.x * sh; Gc = alb.y * sh; Bc = alb.z * sh;
      }
      cr[p] = to_u8(R); cg[p] = to_u8(Gc); cb[p] = to_u8(Bc);
    }
    uint8_t* dst = a.pixels + (((size_t)img * a.H + y) * a.W + px) * 3;
    if (fast) {
      u3 v;
      v.x = cr[0] | cg[0] << 8 | cb[0] << 16 | cr[1] << 24;
      v.y = cg[1] | cb[1] << 8 | cr[2] << 16 | cg[2] << 24;
      v.z = cb[2] | cr[3] << 8 | cg[3] << 16 | cb[3] << 24;
      __builtin_nontemporal_store(v, reinterpret_cast<u3*>(dst));
    } else {
#pragma unroll
      for (int p = 0; p < 4; p++)
        if (px + p < a.W) { dst[3 * p] = (uint8_t)cr[p]; dst[3 * p + 1] = (uint8_t)cg[p]; dst[3 * p + 2] = (uint8_t)cb[p]; }
    }
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

extern "C" int mir_render(MirHandle h, const MirCameraSpec* cam, const MirVisualSpec* vis, int32_t mode, const float* env_offset,
                          uint8_t* pixels, void* stream) {
  if (!h || !cam || !vis || !pixels) return mir_set_error(MIR_E_INVALID, "mir_render: null argument");
  if (vis->struct_size != (int)sizeof(MirVisualSpec)) return mir_set_error(MIR_E_INVALID, "mir_render: MirVisualSpec size mismatch");
  if (cam->width <= 0 || cam->height <= 0 || !(cam->fov_deg > 0.0 && cam->fov_deg < 180.0))
    return mir_set_error(MIR_E_INVALID, "mir_render: bad camera (res / fov)");
  if (mode != MIR_RENDER_PER_ENV && mode != MIR_RENDER_GLOBAL) return mir_set_error(MIR_E_INVALID, "mir_render: unknown mode");
  if (!(vis->checker_size > 0.0)) return mir_set_error(MIR_E_INVALID, "mir_render: checker_size must be > 0");
  double f[3], r[3], u[3], d[3] = {cam->lookat[0] - cam->pos[0], cam->lookat[1] - cam->pos[1], cam->lookat[2] - cam->pos[2]};
  norm3(d, f);
  double rr[3];
  cross3(f, cam->up, rr);
  norm3(rr, r);
  cross3(r, f, u);
  if (f[0] == 0 && f[1] == 0 && f[2] == 0) return mir_set_error(MIR_E_INVALID, "mir_render: camera pos == lookat");
  if (r[0] == 0 && r[1] == 0 && r[2] == 0) return mir_set_error(MIR_E_INVALID, "mir_render: view direction parallel to up");
  int prev = -1;
  (void)hipGetDevice(&prev);
  if (prev != h->device) (void)hipSetDevice(h->device);
  int rc = MIR_OK;
  hipStream_t st = (hipStream_t)stream;
  const int ng = h->hm.ngeom, B = h->B;
  do {
    if (!h->prims) {
      hipError_t e = hipMalloc((void**)&h->prims, (size_t)B * ng * PREC * sizeof(float));
      if (e != hipSuccess) { rc = mir_set_error(MIR_E_HIP, hipGetErrorString(e)); break; }
    }
    if ((rc = mir_refresh_poses(h, stream)) != MIR_OK) break;
    SetupArgs sa;
    memset(&sa, 0, sizeof sa);
    sa.model = h->dm; sa.poses = h->poses; sa.env_offset = env_offset; sa.prims = h->prims;
    const double ty = std::tan(0.5 * cam->fov_deg * M_PI / 180.0), tx = ty * (double)cam->width / (double)cam->height;
    for (int k = 0; k < 3; k++) { sa.cam.pos[k] = (float)cam->pos[k]; sa.cam.f[k] = (float)f[k]; sa.cam.r[k] = (float)r[k]; sa.cam.u[k] = (float)u[k]; }
    sa.cam.tanx = (float)tx; sa.cam.tany = (float)ty; sa.cam.W = cam->width; sa.cam.H = cam->height;
    double L[3];
    norm3(vis->light_dir, L);
    for (int k = 0; k < 3; k++) sa.light[k] = (float)L[k];
    for (int g = 0; g < ng; g++)
      for (int k = 0; k < 3; k++) sa.rgb[g][k] = (float)vis->geom_rgb[g][k];
    sa.B = B; sa.ngeom = ng; sa.global_mode = mode == MIR_RENDER_GLOBAL;
    hipLaunchKernelGGL(k_render_setup, dim3((B * ng + 255) / 256), dim3(256), 0, st, sa);
    PixArgs pa;
    memset(&pa, 0, sizeof pa);
    pa.prims = h->prims; pa.pixels = pixels; pa.W = cam->width; pa.H = cam->height;
    pa.nprim = mode == MIR_RENDER_GLOBAL ? B * ng : ng;
    pa.dx = (float)(2.0 * tx / cam->width); pa.x0 = (float)(-tx + tx / cam->width);
    pa.dy = (float)(-2.0 * ty / cam->height); pa.y0 = (float)(ty - ty / cam->height);
    pa.amb = (float)vis->ambient; pa.dif = (float)vis->diffuse; pa.inv_chk = (float)(1.0 / vis->checker_size);
    for (int k = 0; k < 3; k++) { pa.sky[k] = (float)vis->sky_rgb[k]; pa.chk[0][k] = (float)vis->checker_rgb[0][k]; pa.chk[1][k] = (float)vis->checker_rgb[1][k]; }
    const int nimg = mode == MIR_RENDER_GLOBAL ? 1 : B;
    hipLaunchKernelGGL(mir_render_kernel, dim3((cam->width + TW - 1) / TW, (cam->height + TH - 1) / TH, nimg), dim3(256), 0, st, pa);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) rc = mir_set_error(MIR_E_HIP, hipGetErrorString(e));
  } while (0);
  if (prev != h->device && prev >= 0) (void)hipSetDevice(prev);
  return rc;
}

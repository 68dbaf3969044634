/*
 * orc_render.c — CPU ORACLE for the `pixels` observation / render() (TEST INFRASTRUCTURE ONLY; see
 * orc_rigid.h for the rules: nothing here is imported, linked or executed by the product).
 *
 * PARITY UNPINNED against the reference: its images come from Genesis's OpenGL rasteriser over mesh
 * assets that are not in /root/reference (cam.render(), gym_genesis/tasks/franka/cube_pick.py:159-180,
 * gym_genesis/env.py:97-98).  What is restated from the reference is the camera contract:
 *     add_camera(res=(W,H), pos, lookat, fov)      gym_genesis/tasks/franka/cube_pick.py:56-63
 *     per_env: pose = envs_offset[i] + (3.5,0,2.5) -> lookat offset + (0,0,0.5)   :166-171
 *     global : one image from the camera's pose                                   :174-176
 *     output uint8 (H, W, 3), row 0 = top of the image                            :41
 * and the image definition of include/mirigid.h (pinhole, vertical fov, +z up, pixel-centre
 * sampling; nearest box / plane per pixel; Lambert + ambient; checker on planes).
 *
 * This is a brute-force float64 ray caster written from that definition: every pixel tests every
 * primitive in WORLD coordinates (no per-primitive camera frames, no tiles, no culling), so it shares
 * neither structure nor code with the HIP kernel.
 */
#define _USE_MATH_DEFINES
#define _GNU_SOURCE
#include <math.h>
#include <string.h>

#include "../include/mirigid.h"

static void q2m(const double* q, double R[3][3]) { /* wxyz -> rotation matrix (columns = frame axes) */
  double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0][0] = 1 - 2 * (y * y + z * z); R[0][1] = 2 * (x * y - w * z); R[0][2] = 2 * (x * z + w * y);
  R[1][0] = 2 * (x * y + w * z); R[1][1] = 1 - 2 * (x * x + z * z); R[1][2] = 2 * (y * z - w * x);
  R[2][0] = 2 * (x * z - w * y); R[2][1] = 2 * (y * z + w * x); R[2][2] = 1 - 2 * (x * x + y * y);
}
static void qmul(const double* a, const double* b, double* o) {
  o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3];
  o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2];
  o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1];
  o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0];
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void norm3(double* v) {
  double n = sqrt(dot3(v, v));
  if (n > 0) { v[0] /= n; v[1] /= n; v[2] /= n; }
}
static void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}

typedef struct {
  int type, geom;
  double c[3], R[3][3], h[3];
} Prim;

/* Render ONE image containing the geoms of `nenv` envs.  xpos (nenv,nbody,3), xquat (nenv,nbody,4 wxyz):
 * link poses; offsets (nenv,3) or NULL; planes are taken from env 0 only.  out (H,W,3) u8.
 * If tdepth != NULL it receives the ray parameter of the nearest hit per pixel (H,W), <0 for sky. */
int orc_render_image(const MirSceneSpec* spec, const MirCameraSpec* cam, const MirVisualSpec* vis, int nenv,
                     const double* xpos, const double* xquat, const double* offsets, unsigned char* out, double* tdepth) {
  const int W = cam->width, H = cam->height, nb = spec->nbody, ng = spec->ngeom;
  static Prim prims[65536];
  int np = 0;
  for (int e = 0; e < nenv; e++)
    for (int g = 0; g < ng; g++) {
      const MirGeomSpec* gs = &spec->geom[g];
      if (gs->type == MIR_GEOM_PLANE && e > 0) continue;
      if (np >= 65536) return -1;
      Prim* p = &prims[np++];
      const double* bp = xpos + ((size_t)e * nb + gs->body) * 3;
      const double* bq = xquat + ((size_t)e * nb + gs->body) * 4;
      double Rb[3][3], q[4];
      q2m(bq, Rb);
      for (int i = 0; i < 3; i++) p->c[i] = bp[i] + Rb[i][0] * gs->pos[0] + Rb[i][1] * gs->pos[1] + Rb[i][2] * gs->pos[2] + (offsets ? offsets[e * 3 + i] : 0.0);
      qmul(bq, gs->quat, q);
      q2m(q, p->R);
      for (int i = 0; i < 3; i++) p->h[i] = gs->size[i];
      p->type = gs->type;
      /* round geoms are drawn as their bounding boxes, as the kernel does (mir_render.hip: k_render_setup) */
      if (gs->type == MIR_GEOM_SPHERE) { p->h[1] = p->h[2] = gs->size[0]; p->type = MIR_GEOM_BOX; }
      if (gs->type == MIR_GEOM_CAPSULE) { p->h[1] = gs->size[0]; p->h[2] = gs->size[0] + gs->size[1]; p->type = MIR_GEOM_BOX; }
      if (gs->type == MIR_GEOM_HULL) { /* the bounding box of the vertices (float32, as the kernel's model holds them) */
        p->h[0] = p->h[1] = p->h[2] = 0.0;
        for (int i = (int)gs->size[0]; i < (int)gs->size[0] + (int)gs->size[1]; i++)
          for (int k = 0; k < 3; k++) { double a = fabs((double)(float)spec->vert[i][k]); if (a > p->h[k]) p->h[k] = a; }
        p->type = MIR_GEOM_BOX;
      }
      p->geom = g;
    }
  double f[3] = {cam->lookat[0] - cam->pos[0], cam->lookat[1] - cam->pos[1], cam->lookat[2] - cam->pos[2]}, r[3], u[3];
  norm3(f);
  cross3(f, cam->up, r);
  if (dot3(r, r) < 1e-24) { /* view parallel to up: fall back to +y, then +x (include/mirigid.h, mir_render_cams) */
    const double ey[3] = {0, 1, 0}, ex[3] = {1, 0, 0};
    cross3(f, ey, r);
    if (dot3(r, r) < 1e-24) cross3(f, ex, r);
  }
  norm3(r);
  cross3(r, f, u);
  const double ty = tan(0.5 * cam->fov_deg * M_PI / 180.0), tx = ty * (double)W / (double)H;
  double L[3] = {vis->light_dir[0], vis->light_dir[1], vis->light_dir[2]};
  norm3(L);
#pragma omp parallel for schedule(dynamic, 4)
  for (int j = 0; j < H; j++)
    for (int i = 0; i < W; i++) {
      const double sx = (2.0 * (i + 0.5) / W - 1.0) * tx, sy = (1.0 - 2.0 * (j + 0.5) / H) * ty;
      double d[3];
      for (int k = 0; k < 3; k++) d[k] = f[k] + sx * r[k] + sy * u[k];
      double best = 1e300, nrm[3] = {0, 0, 1};
      int bi = -1;
      for (int pi = 0; pi < np; pi++) {
        const Prim* p = &prims[pi];
        double oc[3] = {cam->pos[0] - p->c[0], cam->pos[1] - p->c[1], cam->pos[2] - p->c[2]};
        if (p->type == MIR_GEOM_PLANE) {
          double n[3] = {p->R[0][2], p->R[1][2], p->R[2][2]};
          double dn = dot3(d, n), on = dot3(oc, n);
          if (dn == 0.0) continue;
          double t = -on / dn;
          if (t > 1e-6 && t < best) {
            best = t; bi = pi;
            double s = on < 0 ? -1.0 : 1.0;
            for (int k = 0; k < 3; k++) nrm[k] = s * n[k];
          }
        } else {
          double tn = -1e300, tf = 1e300;
          int fk = -1;
          double fs = 0;
          int miss = 0;
          for (int k = 0; k < 3 && !miss; k++) {
            double ax[3] = {p->R[0][k], p->R[1][k], p->R[2][k]};
            double ok = dot3(oc, ax), dk = dot3(d, ax);
            if (dk == 0.0) { if (fabs(ok) > p->h[k]) miss = 1; continue; }
            double t1 = (-p->h[k] - ok) / dk, t2 = (p->h[k] - ok) / dk;
            double lo = t1 < t2 ? t1 : t2, hi = t1 < t2 ? t2 : t1;
            if (lo > tn) { tn = lo; fk = k; fs = dk > 0 ? -1.0 : 1.0; }
            if (hi < tf) tf = hi;
          }
          if (miss || tn > tf || tn <= 1e-6 || fk < 0) continue;
          if (tn < best) {
            best = tn; bi = pi;
            for (int k = 0; k < 3; k++) nrm[k] = fs * p->R[k][fk];
          }
        }
      }
      double rgb[3] = {vis->sky_rgb[0], vis->sky_rgb[1], vis->sky_rgb[2]};
      if (bi >= 0) {
        const Prim* p = &prims[bi];
        double alb[3] = {vis->geom_rgb[p->geom][0], vis->geom_rgb[p->geom][1], vis->geom_rgb[p->geom][2]};
        if (p->type == MIR_GEOM_PLANE) {
          double hit[3], eu[3] = {p->R[0][0], p->R[1][0], p->R[2][0]}, ev[3] = {p->R[0][1], p->R[1][1], p->R[2][1]};
          for (int k = 0; k < 3; k++) hit[k] = cam->pos[k] + best * d[k] - p->c[k];
          long a = (long)floor(dot3(hit, eu) / vis->checker_size), b = (long)floor(dot3(hit, ev) / vis->checker_size);
          int par = (int)((a + b) & 1L);
          for (int k = 0; k < 3; k++) alb[k] = vis->checker_rgb[par][k];
        }
        double nl = dot3(nrm, L);
        double sh = vis->ambient + vis->diffuse * (nl > 0 ? nl : 0.0);
        for (int k = 0; k < 3; k++) rgb[k] = alb[k] * sh;
      }
      for (int k = 0; k < 3; k++) {
        double c = rgb[k] < 0 ? 0 : (rgb[k] > 1 ? 1 : rgb[k]);
        out[((size_t)j * W + i) * 3 + k] = (unsigned char)floor(c * 255.0 + 0.5);
      }
      if (tdepth) tdepth[(size_t)j * W + i] = bi >= 0 ? best : -1.0;
    }
  return 0;
}

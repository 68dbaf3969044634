// Host build of the kernels' lane-private convex narrowphase (gym-genesis_amd/csrc/mir_convex.h) for the CPU test tier: the
// header is plain C++ apart from its qualifiers, so the SAME source that runs per lane on the GPU is compiled here with g++,
// in float32, and compared with the float64 oracle pair by pair (tests/test_convex_host.py).  Test infrastructure only.
#include <cmath>
#include <cstdint>
#define MIR_CONVEX_STANDALONE
#ifdef CONVEX_HOST_DOUBLE /* the same source in float64: separates algorithmic differences from float32 rounding */
#define float double
#define sqrtf sqrt
#define fminf fmin
#endif
#define __device__
#define __forceinline__ inline
#define MIR_GEOM_PLANE 0
#define MIR_GEOM_BOX 1
#define MIR_GEOM_SPHERE 2
#define MIR_GEOM_CAPSULE 3
#define MIR_GEOM_HULL 4
namespace {
struct f4 { float x, y, z, w; };
struct V3 { float x, y, z; };
inline V3 v3(float x, float y, float z) { return {x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(float s, V3 a) { return {s * a.x, s * a.y, s * a.z}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
}  // namespace
#include "../gym-genesis_amd/csrc/mir_convex.h"

// vertex pool the MIR_GEOM_HULL shapes index into (size = first vertex, count): rows of 4 floats like the kernel's LDS copy
static float g_pool[96][4];
extern "C" void convex_host_set_pool(const double* verts, int n) {
  for (int i = 0; i < n && i < 96; i++) {
    for (int k = 0; k < 3; k++) g_pool[i][k] = (float)verts[3 * i + k];
    g_pool[i][3] = 0;
  }
}

// in: type1, size1[3], pos1[3], quat1[4] wxyz, type2, ...; out: hit, pos[3], dist, normal[3]
extern "C" void convex_host_pairs(const float* in, float* out, int n) {
  for (int i = 0; i < n; i++) {
    const float* r = in + (size_t)i * 22;
    ShapeD S[2];
    for (int k = 0; k < 2; k++) {
      const float* p = r + 11 * k;
      float w = p[7], x = p[8], y = p[9], z = p[10];
      const float nq = std::sqrt(w * w + x * x + y * y + z * z);
      w /= nq; x /= nq; y /= nq; z /= nq;
      S[k].type = (int)p[0];
      S[k].size = v3(p[1], p[2], p[3]);
      S[k].pos = v3(p[4], p[5], p[6]);
      S[k].a0 = v3(1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y));
      S[k].a1 = v3(2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x));
      S[k].a2 = v3(2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y));
      S[k].verts = &g_pool[S[k].type == MIR_GEOM_HULL ? (int)p[1] : 0][0];
      S[k].nvert = (int)p[2];
    }
    f4 pt = {0, 0, 0, 0};
    V3 nrm = v3(0, 0, 0);
    const bool hit = convex_pair(S[0], S[1], pt, nrm);
    float* o = out + (size_t)i * 8;
    o[0] = hit ? 1.0f : 0.0f; o[1] = pt.x; o[2] = pt.y; o[3] = pt.z; o[4] = pt.w; o[5] = nrm.x; o[6] = nrm.y; o[7] = nrm.z;
  }
}

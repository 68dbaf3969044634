// store_probe.hip — what bounds the rasteriser's floor-only path: the store pattern or the arithmetic?
// Writes a 1024 x 480 x 640 x 3 image set with the rasteriser's walk (workgroup = 4 waves on a 128-px strip of 96 rows) and
// different lane -> byte mappings, with and without the floor's per-pixel arithmetic.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef unsigned u3 __attribute__((ext_vector_type(3)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
#ifndef W
#define W 640
#define H 480
#endif
#ifndef THP
#define THP 96 /* rows per workgroup strip (modes 0, 1, 3, 4); -DTHP=32: one band per wave */
#endif

struct A { uint8_t* px; float x0, dx, y0, dy, nu0, nux, nuy, nv0, nvx, nvy, fz, uz, oz; unsigned ce, co, sky; int th; /* rows per strip, mode 0 (0 = THP) */ };

// floor colour of pixel (x, y) image-plane coords: row-constant reciprocal
__device__ __forceinline__ unsigned floor_col(const A& a, float xs, float eu, float ev, float iz1, unsigned ca, unsigned cb) {
  const float u = fmaf(xs, a.nux, eu) * iz1, v = fmaf(xs, a.nvx, ev) * iz1;
  const bool odd = (__builtin_amdgcn_fractf(u) >= 0.5f) != (__builtin_amdgcn_fractf(v) >= 0.5f);
  return odd ? cb : ca;
}

// MODE 0: current pattern, dwordx3 per lane, 2 rows x 384 B per wave store.  ARITH 0: constant colour, 1: floor arithmetic
template <int MODE, int ARITH, int NT = 0>
__global__ __launch_bounds__(MODE == 4 ? 64 * (W / 128) : 256) void k(A a) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (NT == 2) {  // XCD-contiguous: the workgroups of XCD j (linear id % 8 == j) walk the j-th eighth of the images in order
    const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), T = gridDim.x * gridDim.y * gridDim.z;
    const unsigned v = (id & 7u) * (T >> 3) + (id >> 3);
    bx = v % gridDim.x; by = (v / gridDim.x) % gridDim.y; bz = v / (gridDim.x * gridDim.y);
  }
  if (NT == 3) {  // image-interleaved: XCD j takes the images j, j + 8, j + 16, ... in order (streams one image apart, not one eighth apart)
    const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z), S = gridDim.x * gridDim.y;
    const unsigned j = id & 7u, k = id >> 3, img = (k / S) * 8u + j, st = k % S;
    bx = st % gridDim.x; by = st / gridDim.x; bz = img;
    if (bz >= (int)gridDim.z) return;
  }
  const int thr = (MODE == 0 && a.th) ? a.th : THP;
  const int tx0 = bx * 128, sy0 = by * thr, img = bz;
  uint8_t* ibase = a.px + (size_t)img * H * W * 3;
  if (MODE == 0) {
    const int px = tx0 + 4 * (lane & 31);
    for (int ty0 = sy0; ty0 < sy0 + thr; ty0 += 32) {
      const int prow = ty0 + 8 * wv + (lane >> 5);
      unsigned boff = ((unsigned)prow * W + px) * 3u;
#pragma unroll
      for (int r = 0; r < 4; r++, boff += 6u * W) {
        unsigned c[4];
        if (ARITH) {
          const float ys = a.y0 + (float)(prow + 2 * r) * a.dy;
          const float ez = fmaf(ys, a.uz, a.fz), eu = fmaf(ys, a.nuy, a.nu0), ev = fmaf(ys, a.nvy, a.nv0);
          const float iz1 = __builtin_amdgcn_rcpf(ez);
          const bool vld = iz1 * (-a.oz) > 1e-6f;
          const unsigned ca = vld ? a.ce : a.sky, cb = vld ? a.co : a.sky;
#pragma unroll
          for (int p = 0; p < 4; p++) c[p] = floor_col(a, a.x0 + (float)(px + p) * a.dx, eu, ev, iz1, ca, cb);
        } else {
#pragma unroll
          for (int p = 0; p < 4; p++) c[p] = a.sky + lane;
        }
        u3 v;
        v.x = c[0] | c[1] << 24; v.y = c[1] >> 8 | c[2] << 16; v.z = c[2] >> 16 | c[3] << 8;
        if (NT == 1) __builtin_nontemporal_store(v, reinterpret_cast<u3*>(ibase + boff)); else *reinterpret_cast<u3*>(ibase + boff) = v;
      }
    }
  } else if (MODE == 1) {
    // dwordx4, 48 lanes: lane l < 48 owns 16 B of a 384-B row segment (24 lanes per row, 2 rows per store); pixels straddle
    // lanes: a lane evaluates the 6 pixels its 16 bytes touch
    if (lane >= 48) return;
    const int lr = lane % 24, rw = lane / 24;
    const int b0 = 16 * lr;            // byte offset in the 384-B segment
    const int p0 = b0 / 3, sh = b0 % 3;  // first pixel touched, bytes of it already behind
    const int px = tx0 + p0;
    for (int ty0 = sy0; ty0 < sy0 + THP; ty0 += 32) {
      const int prow = ty0 + 8 * wv + rw;
      unsigned boff = ((unsigned)prow * W + tx0) * 3u + b0;
#pragma unroll
      for (int r = 0; r < 4; r++, boff += 6u * W) {
        unsigned c[6];
        if (ARITH) {
          const float ys = a.y0 + (float)(prow + 2 * r) * a.dy;
          const float ez = fmaf(ys, a.uz, a.fz), eu = fmaf(ys, a.nuy, a.nu0), ev = fmaf(ys, a.nvy, a.nv0);
          const float iz1 = __builtin_amdgcn_rcpf(ez);
          const bool vld = iz1 * (-a.oz) > 1e-6f;
          const unsigned ca = vld ? a.ce : a.sky, cb = vld ? a.co : a.sky;
#pragma unroll
          for (int p = 0; p < 6; p++) c[p] = floor_col(a, a.x0 + (float)(px + p) * a.dx, eu, ev, iz1, ca, cb);
        } else {
#pragma unroll
          for (int p = 0; p < 6; p++) c[p] = a.sky + lane;
        }
        // 18 bytes of 6 pixels -> the 16 starting at byte `sh`
        const unsigned long long q0 = (unsigned long long)c[0] | (unsigned long long)c[1] << 24 | (unsigned long long)c[2] << 48;
        const unsigned d0 = (unsigned)q0, d1 = (unsigned)(q0 >> 32);                 // bytes 0..7 (c2's first 2 bytes at 6,7)
        const unsigned d2 = c[2] >> 16 | c[3] << 8;                                  // bytes 8..11
        const unsigned d3 = c[4] | c[5] << 24, d4 = c[5] >> 8;                       // bytes 12..15, 16..17
        u4 v;
        const unsigned s8 = 8u * sh;
        v.x = __builtin_amdgcn_alignbit(d1, d0, s8); v.y = __builtin_amdgcn_alignbit(d2, d1, s8);
        v.z = __builtin_amdgcn_alignbit(d3, d2, s8); v.w = __builtin_amdgcn_alignbit(d4, d3, s8);
        *reinterpret_cast<u4*>(ibase + boff) = v;
      }
    }
  } else if (MODE == 2) {
    // linear: the image as a byte stream, a wave store = 1024 contiguous bytes; workgroup = 4 KB chunks walked like the strip
    // (same bytes per workgroup as modes 0/1: 128 x 96 x 3 = 36864 B = 9 trips of 4 KB)
    const size_t wg = (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 36864;
    for (int t = 0; t < 9; t++) {
      const size_t off = wg + (size_t)t * 4096 + tid * 16;
      const unsigned bo = (unsigned)off;
      const int row = bo / (W * 3), b0 = bo % (W * 3);
      const int p0 = b0 / 3, sh = b0 % 3;
      unsigned c[6];
      if (ARITH) {
        const float ys = a.y0 + (float)row * a.dy;
        const float ez = fmaf(ys, a.uz, a.fz), eu = fmaf(ys, a.nuy, a.nu0), ev = fmaf(ys, a.nvy, a.nv0);
        const float iz1 = __builtin_amdgcn_rcpf(ez);
        const bool vld = iz1 * (-a.oz) > 1e-6f;
        const unsigned ca = vld ? a.ce : a.sky, cb = vld ? a.co : a.sky;
#pragma unroll
        for (int p = 0; p < 6; p++) c[p] = floor_col(a, a.x0 + (float)(p0 + p) * a.dx, eu, ev, iz1, ca, cb);
      } else {
#pragma unroll
        for (int p = 0; p < 6; p++) c[p] = a.sky + lane;
      }
      const unsigned long long q0 = (unsigned long long)c[0] | (unsigned long long)c[1] << 24 | (unsigned long long)c[2] << 48;
      const unsigned d0 = (unsigned)q0, d1 = (unsigned)(q0 >> 32);
      const unsigned d2 = c[2] >> 16 | c[3] << 8;
      const unsigned d3 = c[4] | c[5] << 24, d4 = c[5] >> 8;
      u4 v;
      const unsigned s8 = 8u * sh;
      v.x = __builtin_amdgcn_alignbit(d1, d0, s8); v.y = __builtin_amdgcn_alignbit(d2, d1, s8);
      v.z = __builtin_amdgcn_alignbit(d3, d2, s8); v.w = __builtin_amdgcn_alignbit(d4, d3, s8);
      if (NT == 1) __builtin_nontemporal_store(v, reinterpret_cast<u4*>(ibase + off)); else *reinterpret_cast<u4*>(ibase + off) = v;
    }
  } else if (MODE == 4) {
    // full-width workgroup: W / 128 waves side by side on the same 8 rows (wave = 128 x 8 band as in mode 0), walking down 96 rows:
    // at any time the workgroup writes 8 whole rows (launch with W / 128 * 64 threads, grid.x = 1)
    const int px = wv * 128 + 4 * (lane & 31);
    for (int ty0 = sy0; ty0 < sy0 + THP; ty0 += 8) {
      const int prow = ty0 + (lane >> 5);
      unsigned boff = ((unsigned)prow * W + px) * 3u;
#pragma unroll
      for (int r = 0; r < 4; r++, boff += 6u * W) {
        unsigned c[4];
        if (ARITH) {
          const float ys = a.y0 + (float)(prow + 2 * r) * a.dy;
          const float ez = fmaf(ys, a.uz, a.fz), eu = fmaf(ys, a.nuy, a.nu0), ev = fmaf(ys, a.nvy, a.nv0);
          const float iz1 = __builtin_amdgcn_rcpf(ez);
          const bool vld = iz1 * (-a.oz) > 1e-6f;
          const unsigned ca = vld ? a.ce : a.sky, cb = vld ? a.co : a.sky;
#pragma unroll
          for (int p = 0; p < 4; p++) c[p] = floor_col(a, a.x0 + (float)(px + p) * a.dx, eu, ev, iz1, ca, cb);
        } else {
#pragma unroll
          for (int p = 0; p < 4; p++) c[p] = a.sky + lane;
        }
        u3 v;
        v.x = c[0] | c[1] << 24; v.y = c[1] >> 8 | c[2] << 16; v.z = c[2] >> 16 | c[3] << 8;
        *reinterpret_cast<u3*>(ibase + boff) = v;
      }
    }
  } else if (MODE == 3) {
    // dwordx3 on a 256-px-wide strip: a wave store = ONE row segment of 768 B (lane = 4 px); wave = band of 256 x 4 rows.
    // Needs W % 256 == 0 (-DW=768 -DH=400: the same 943.7 MB); grid.x = W / 256.
    const int px = bx * 256 + 4 * lane;
    for (int ty0 = sy0; ty0 < sy0 + THP; ty0 += 16) {
      const int prow = ty0 + 4 * wv;
      unsigned boff = ((unsigned)prow * W + px) * 3u;
#pragma unroll
      for (int r = 0; r < 4; r++, boff += 3u * W) {
        unsigned c[4];
#pragma unroll
        for (int p = 0; p < 4; p++) c[p] = a.sky + lane;
        u3 v;
        v.x = c[0] | c[1] << 24; v.y = c[1] >> 8 | c[2] << 16; v.z = c[2] >> 16 | c[3] << 8;
        *reinterpret_cast<u3*>(ibase + boff) = v;
      }
    }
  }
}

// Work queue: resident waves take 128 x 8 bands from a per-XCD counter, in address order inside the XCD's eighth of the images
// (image, band row, column): at any instant an XCD's waves write one compact, advancing window.  queue[8] zeroed before the launch.
template <int ARITH>
__global__ __launch_bounds__(256) void kq(A a, unsigned* queue, int nimg) {
  const int lane = threadIdx.x & 63;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  xcc &= 7u;
  const unsigned per_img = (H / 8) * (W / 128), chunk = (unsigned)((nimg + 7) / 8) * per_img;  // bands per XCD
  // (a counter per XCD serialises: 38 400 atomics on one address = 3.5 ms.  Static round robin instead: wave s of the XCD's resident
  //  waves takes bands s, s + S, s + 2 S, ... -- the workgroups of an XCD are those with blockIdx % 8 == its slot in the dispatch order)
  (void)queue;
  const unsigned S = (gridDim.x >> 3) * 4u, s0 = (blockIdx.x >> 3) * 4u + (threadIdx.x >> 6);
  xcc = blockIdx.x & 7u;
  for (unsigned b = s0; b < chunk; b += S) {
    const unsigned g = xcc * chunk + b, img = g / per_img, r = g % per_img;
    if ((int)img >= nimg) break;
    const int by = r / (W / 128), bx = r % (W / 128);
    uint8_t* ibase = a.px + (size_t)img * H * W * 3;
    const int px = bx * 128 + 4 * (lane & 31), prow = by * 8 + (lane >> 5);
    unsigned boff = ((unsigned)prow * W + px) * 3u;
#pragma unroll
    for (int rr = 0; rr < 4; rr++, boff += 6u * W) {
      unsigned c[4];
      if (ARITH) {
        const float ys = a.y0 + (float)(prow + 2 * rr) * a.dy;
        const float ez = fmaf(ys, a.uz, a.fz), eu = fmaf(ys, a.nuy, a.nu0), ev = fmaf(ys, a.nvy, a.nv0);
        const float iz1 = __builtin_amdgcn_rcpf(ez);
        const bool vld = iz1 * (-a.oz) > 1e-6f;
        const unsigned ca = vld ? a.ce : a.sky, cb = vld ? a.co : a.sky;
#pragma unroll
        for (int p = 0; p < 4; p++) c[p] = floor_col(a, a.x0 + (float)(px + p) * a.dx, eu, ev, iz1, ca, cb);
      } else {
#pragma unroll
        for (int p = 0; p < 4; p++) c[p] = a.sky + lane;
      }
      u3 v;
      v.x = c[0] | c[1] << 24; v.y = c[1] >> 8 | c[2] << 16; v.z = c[2] >> 16 | c[3] << 8;
      *reinterpret_cast<u3*>(ibase + boff) = v;
    }
  }
}
template <int ARITH>
float runq(A a, int B, int n, int wgs) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  unsigned* q; hipMalloc((void**)&q, 32);
  for (int i = 0; i < 3; i++) { hipMemsetAsync(q, 0, 32, 0); hipLaunchKernelGGL(kq<ARITH>, dim3(wgs), dim3(256), 0, 0, a, q, B); }
  hipEventRecord(e0);
  for (int i = 0; i < n; i++) { hipMemsetAsync(q, 0, 32, 0); hipLaunchKernelGGL(kq<ARITH>, dim3(wgs), dim3(256), 0, 0, a, q, B); }
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipFree(q);
  return ms / n * 1e3f;
}

template <int MODE, int ARITH, int NT = 0>
float run(A a, int B, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  dim3 g(MODE == 3 ? W / 256 : (MODE == 4 ? 1 : W / 128), H / ((MODE == 0 && a.th) ? a.th : THP), B);
  const dim3 blk(MODE == 4 ? 64 * (W / 128) : 256);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<MODE, ARITH, NT>), g, blk, 0, 0, a);
  hipEventRecord(e0);
  for (int i = 0; i < n; i++) hipLaunchKernelGGL((k<MODE, ARITH, NT>), g, blk, 0, 0, a);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / n * 1e3f;
}

// persistent linear fill: G workgroups, each writing 4 KB per trip at stride G x 4 KB (what a fill kernel does)
template <int CONSTANT>
__global__ __launch_bounds__(256) void kp(uint8_t* px, size_t bytes, int per) {
  for (size_t off = ((size_t)blockIdx.x * per) * 4096 + threadIdx.x * 16; off < bytes; off += (size_t)gridDim.x * per * 4096) {
    for (int t = 0; t < per; t++) {
      if (off + (size_t)t * 4096 < bytes) *reinterpret_cast<u4*>(px + off + (size_t)t * 4096) = CONSTANT ? u4{7u, 7u, 7u, 7u} : u4{(unsigned)off * 2654435761u, (unsigned)t * 40503u + threadIdx.x, 7u ^ (unsigned)(off >> 7), (unsigned)threadIdx.x * 2246822519u};
    }
  }
}
template <int CONSTANT>
float runp(uint8_t* px, size_t bytes, int G, int per, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(kp<CONSTANT>, dim3(G), dim3(256), 0, 0, px, bytes, per);
  hipEventRecord(e0);
  for (int i = 0; i < n; i++) hipLaunchKernelGGL(kp<CONSTANT>, dim3(G), dim3(256), 0, 0, px, bytes, per);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / n * 1e3f;
}

// one-shot linear fill like a framework's fill kernel: workgroup i writes chunk c(i) of `per` x 4 KB and exits.
// MAP 0: c = i.  MAP 1: c = (i % 8) * (n / 8) + i / 8 (workgroups i, i+1, .. of the 8 XCDs write far apart; an XCD writes a contiguous eighth).
// MAP 2: c = i with the low 3 bits rotated by (i >> 3) (page p goes to XCD (p + p / 8) % 8: every XCD sees every page residue)
template <int MAP>
__global__ __launch_bounds__(256) void k1(uint8_t* px, size_t bytes, int per, unsigned n) {
  unsigned i = blockIdx.x, c = i;
  if (MAP == 1) c = (i & 7u) * (n >> 3) + (i >> 3);
  if (MAP == 2) c = (i & ~7u) | ((i + (i >> 3)) & 7u);
  const size_t off = (size_t)c * per * 4096 + threadIdx.x * 16;
  for (int t = 0; t < per; t++)
    if (off + (size_t)t * 4096 < bytes) *reinterpret_cast<u4*>(px + off + (size_t)t * 4096) = u4{7u, 7u, (unsigned)t, (unsigned)threadIdx.x};
}
template <int MAP>
float run1(uint8_t* px, size_t bytes, int per, int n) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned G = (unsigned)((bytes / 4096 / per) & ~7ull);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL(k1<MAP>, dim3(G), dim3(256), 0, 0, px, bytes, per, G);
  hipEventRecord(e0);
  for (int i = 0; i < n; i++) hipLaunchKernelGGL(k1<MAP>, dim3(G), dim3(256), 0, 0, px, bytes, per, G);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / n * 1e3f;
}

int main() {
  const int B = 1024;
  const size_t bytes = (size_t)B * H * W * 3;
  A a{};
  hipMalloc((void**)&a.px, bytes);
  // camera (3.5, 0, 2.5) -> (0, 0, 0.5), fov 30: numbers of the plane record in the right ballpark
  a.dx = 2.0f * 0.357f / W; a.x0 = -0.357f + 0.5f * a.dx; a.dy = -2.0f * 0.268f / H; a.y0 = 0.268f + 0.5f * a.dy;
  a.fz = -0.496f; a.uz = 0.868f; a.oz = 2.5f; a.nu0 = 1.2f; a.nux = 0.0f; a.nuy = 3.1f; a.nv0 = 0.0f; a.nvx = 2.5f; a.nvy = 0.0f;
  a.ce = 0x00c8c8c8; a.co = 0x00505050; a.sky = 0x00e6b48c;
  const double gb = bytes / 1e9;
  if (THP != 96) {  // strip-height variants: modes 0 and 4 only (mode 2's chunk arithmetic assumes 96 rows)
    static_assert(H % THP == 0 && THP % 32 == 0, "strip height");
    for (int nb = 0; nb < 3; nb++) {
      A b = a;
      hipMalloc((void**)&b.px, bytes);
      printf("strips of %d rows, buffer %p: %7.1f us | XCD-contiguous %7.1f us, with arithmetic %7.1f us\n", THP, (void*)b.px, run<0, 0>(b, B, 20), run<0, 0, 2>(b, B, 20), run<0, 1, 2>(b, B, 20));
    }
    return 0;
  }
#define R(M, AR) { float us = run<M, AR>(a, B, 20); printf("mode %d arith %d: %7.1f us  %6.0f GB/s\n", M, AR, us, gb / us * 1e6); }
  R(0, 0) R(1, 0) R(2, 0) R(0, 1) R(1, 1) R(2, 1)
  if (W % 256 == 0) R(3, 0)
#define RN(M, AR) { float us = run<M, AR, 1>(a, B, 20); printf("mode %d arith %d nontemporal: %7.1f us  %6.0f GB/s\n", M, AR, us, gb / us * 1e6); }
  RN(0, 0) RN(2, 0) RN(0, 1)
#define RX(M, AR) { float us = run<M, AR, 2>(a, B, 20); printf("mode %d arith %d XCD-contiguous images: %7.1f us  %6.0f GB/s\n", M, AR, us, gb / us * 1e6); }
  RX(0, 0) RX(0, 1) RX(1, 0)
  for (int G : {2048, 16384}) for (int per : {9}) { float us = runp<0>(a.px, bytes, G, per, 20); printf("persistent linear, varied data, G=%d per=%d: %7.1f us %6.0f GB/s\n", G, per, us, gb / us * 1e6); us = runp<1>(a.px, bytes, G, per, 20); printf("persistent linear, constant data, G=%d per=%d: %7.1f us %6.0f GB/s\n", G, per, us, gb / us * 1e6); }
  for (int per : {1, 2, 4, 9}) { float u0 = run1<0>(a.px, bytes, per, 20), u1 = run1<1>(a.px, bytes, per, 20), u2 = run1<2>(a.px, bytes, per, 20); printf("one-shot linear, %d x 4 KB per workgroup: in order %7.1f us %6.0f GB/s | XCD-contiguous eighths %7.1f us | rotated %7.1f us\n", per, u0, gb / u0 * 1e6, u1, u2); }
  // does the PLACEMENT of the buffer matter?  (tools/probes/render_align.py: the rasteriser runs at 180 us into some allocations and at
  // 213 us into others of the same process)  Five more allocations, the strided strip pattern and the linear ones on each.
  for (int nb = 0; nb < 5; nb++) {
    A b = a;
    hipMalloc((void**)&b.px, bytes);
    const float s0 = run<0, 1>(b, B, 20), sx = run<0, 1, 2>(b, B, 20), l1 = run1<0>(b.px, bytes, 1, 20), l9 = run1<0>(b.px, bytes, 9, 20);
    printf("buffer %p: strips %7.1f us | strips, XCD-contiguous %7.1f us | one-shot linear 4 KB %7.1f us | 36 KB %7.1f us", (void*)b.px, s0, sx, l1, l9);
    printf(" | strips, XCD takes every eighth IMAGE %7.1f us", run<0, 1, 3>(b, B, 20));
    for (int th : {32, 96, 160, 480}) { A c = b; c.th = th; printf(" | %d-row strips XCD-contiguous %7.1f us", th, run<0, 1, 2>(c, B, 20)); }
    printf(" | resident waves, bands round robin in address order per XCD (1536 / 3072 workgroups): %7.1f / %7.1f us, with arithmetic %7.1f / %7.1f us", runq<0>(b, B, 20, 1536), runq<0>(b, B, 20, 3072), runq<1>(b, B, 20, 1536), runq<1>(b, B, 20, 3072));
    printf(" | full-width workgroups, XCD-contiguous: %7.1f us, with arithmetic %7.1f us", run<4, 0, 2>(b, B, 20), run<4, 1, 2>(b, B, 20));
    if (W % 256 == 0) printf(" | 128-px strips no arithmetic, XCD-contiguous %7.1f us | 256-px strips (768-byte pieces), XCD-contiguous %7.1f us", run<0, 0, 2>(b, B, 20), run<3, 0, 2>(b, B, 20));
    printf("\n");
  }
  // plain fill
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) hipMemsetAsync(a.px, 7, bytes, 0);
  hipEventRecord(e0);
  for (int i = 0; i < 20; i++) hipMemsetAsync(a.px, 7, bytes, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("hipMemset: %7.1f us  %6.0f GB/s\n", ms / 20 * 1e3, gb / (ms / 20) * 1e3);
  return 0;
}

// GPU probe: the lane-private convex narrowphase compiled on its own with a chosen set of flags (see build.sh)
#include <hip/hip_runtime.h>
#include "../../../gym-genesis_amd/csrc/mir_convex.h"
namespace {
__global__ void k(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float* r = in + (size_t)i * 22;
  const M3 R1 = q2m(qnormalize(Q4{r[7], r[8], r[9], r[10]})), R2 = q2m(qnormalize(Q4{r[18], r[19], r[20], r[21]}));
  const ShapeD A = {(int)r[0], v3(r[1], r[2], r[3]), v3(r[4], r[5], r[6]), mcol(R1, 0), mcol(R1, 1), mcol(R1, 2)};
  const ShapeD B = {(int)r[11], v3(r[12], r[13], r[14]), v3(r[15], r[16], r[17]), mcol(R2, 0), mcol(R2, 1), mcol(R2, 2)};
  f4 pt = {0, 0, 0, 0};
  V3 nrm = v3(0, 0, 0);
  const bool hit = convex_pair(A, B, pt, nrm);
  float* o = out + (size_t)i * 8;
  o[0] = hit ? 1.0f : 0.0f; o[1] = pt.x; o[2] = pt.y; o[3] = pt.z; o[4] = pt.w; o[5] = nrm.x; o[6] = nrm.y; o[7] = nrm.z;
}
}  // namespace
extern "C" int probe_run(const float* in, float* out, int n) {
  hipLaunchKernelGGL(k, dim3((n + 63) / 64), dim3(64), 0, 0, in, out, n);
  return (int)hipDeviceSynchronize();
}

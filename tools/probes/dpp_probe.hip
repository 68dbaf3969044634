// probe: semantics of the DPP row primitives used by mir_step.hip (run on gfx950)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> __device__ __forceinline__ float row_ror(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xf, 0xf, false));
}
template <int K> __device__ __forceinline__ float row_bcast(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + K, 0xf, 0xf, false));
}
__device__ __forceinline__ float gsum(float v) {
  v += row_ror<1>(v); v += row_ror<2>(v); v += row_ror<4>(v); v += row_ror<8>(v);
  return v;
}
__global__ void k(float* out, const float* in) {
  float v = in[threadIdx.x];
  out[threadIdx.x] = gsum(v);
  out[64 + threadIdx.x] = row_bcast<3>(v);
  out[128 + threadIdx.x] = row_ror<1>(v);
  float w = (threadIdx.x & 1) ? v : 0.0f;   // value computed under a select, then reduced
  out[192 + threadIdx.x] = gsum(w * w);
}
int main() {
  float h[64], o[256], *d, *od;
  for (int i = 0; i < 64; i++) h[i] = (float)(i + 1);
  hipMalloc(&d, sizeof h); hipMalloc(&od, sizeof o);
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, od, d);
  hipMemcpy(o, od, sizeof o, hipMemcpyDeviceToHost);
  for (int g = 0; g < 4; g++) {
    float ref = 0, ref2 = 0; for (int i = 0; i < 16; i++) { ref += h[g*16+i]; if (i & 1) ref2 += h[g*16+i]*h[g*16+i]; }
    printf("row %d: gsum lanes0/7/15 = %g %g %g (ref %g) | bcast3 lane0/9 = %g %g (ref %g) | ror1 lane0,1 = %g %g | gsum(sel) %g (ref %g)\n", g,
           o[g*16], o[g*16+7], o[g*16+15], ref, o[64+g*16], o[64+g*16+9], h[g*16+3], o[128+g*16], o[128+g*16+1], o[192+g*16+5], ref2);
  }
  return 0;
}

// Probe: register layout of v_mfma_f32_16x16x1_4b_f32 on gfx950 (4 blocks of 16x16, K = 1).
// hipcc --offload-arch=gfx950 -O2 mfma_layout.hip -o mfma_layout && ./mfma_layout
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
  const int l = threadIdx.x;
  // A_b[i] = 1000 b + i + 1 ; B_b[j] = 1 (so D_b[i][j] = A_b[i]) -- then the reverse to find j
  v16f acc = {0};
  acc = __builtin_amdgcn_mfma_f32_16x16x1f32((float)(1000 * (l / 16) + (l % 16) + 1), 1.0f, acc, 0, 0, 0);
  v16f acc2 = {0};
  acc2 = __builtin_amdgcn_mfma_f32_16x16x1f32(1.0f, (float)(1000 * (l / 16) + (l % 16) + 1), acc2, 0, 0, 0);
  for (int v = 0; v < 16; v++) { out[(l * 16 + v) * 2] = acc[v]; out[(l * 16 + v) * 2 + 1] = acc2[v]; }
}
int main() {
  float* d; hipMalloc(&d, 64 * 16 * 2 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  static float h[64 * 16 * 2]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int l : {0, 1, 15, 16, 17, 33, 63}) {
    printf("lane %2d:", l);
    for (int v = 0; v < 16; v++) printf(" [%g|%g]", h[(l * 16 + v) * 2], h[(l * 16 + v) * 2 + 1]);
    printf("\n");
  }
  return 0;
}

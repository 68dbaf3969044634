// Host cost of one kernel launch call, three ways (run on the GPU box): hipLaunchKernelGGL with a ~400-byte by-value struct (what
// mir_launch_step does), hipModuleLaunchKernel with kernelParams, hipModuleLaunchKernel with one packed argument buffer.
//   hipcc --offload-arch=gfx950 -O2 launch_cost.hip -o launch_cost && ./launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
struct Big { void* p[40]; int i[20]; };
__global__ void k_null(Big b) { if (b.i[0] == 12345 && threadIdx.x == 0) reinterpret_cast<int*>(b.p[0])[0] = 1; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  Big b; memset(&b, 0, sizeof b);
  int* d; hipMalloc(&d, 64); b.p[0] = d;
  hipStream_t s; hipStreamCreate(&s);
  hipFunction_t f;
  if (hipGetFuncBySymbol(&f, reinterpret_cast<const void*>(&k_null)) != hipSuccess) { printf("hipGetFuncBySymbol failed\n"); return 1; }
  const int N = 20000;
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 3; rep++) {
      double acc = 0;
      for (int i = 0; i < N; i++) {
        if (i % 64 == 0) hipStreamSynchronize(s);  // (the queue never fills: the call's own cost is what is timed)
        const double t0 = now();
        if (mode == 0) hipLaunchKernelGGL(k_null, dim3(1024), dim3(128), 0, s, b);
        else if (mode == 1) { void* args[1] = {&b}; hipModuleLaunchKernel(f, 1024, 1, 1, 128, 1, 1, 0, s, args, nullptr); }
        else {
          size_t sz = sizeof b;
          void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &b, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
          hipModuleLaunchKernel(f, 1024, 1, 1, 128, 1, 1, 0, s, nullptr, extra);
        }
        acc += now() - t0;
      }
      hipStreamSynchronize(s);
      printf("mode %d (%s): %.3f us per launch call\n", mode, mode == 0 ? "hipLaunchKernelGGL" : mode == 1 ? "hipModuleLaunchKernel kernelParams" : "hipModuleLaunchKernel extra buffer", acc / N * 1e6);
    }
  }
  return 0;
}

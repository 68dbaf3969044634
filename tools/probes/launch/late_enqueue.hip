// How soon does a dependent kernel start after its predecessor ends, as a function of WHEN it was enqueued?  Kernel A spins ~20 us
// (s_memrealtime), the host waits X us after launching A and launches B on the same stream; both record their start / end on the
// device clock (100 MHz).  Prints B.start - A.end per X.  (If a packet that arrives while A runs were picked up at once, the gap
// would be the same small number for every X < 20.)
//   hipcc --offload-arch=gfx950 -O2 late_enqueue.hip -o late_enqueue && ./late_enqueue
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <vector>
__global__ void k_spin(unsigned long long* out, unsigned long long ticks, int grid_mark) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
  if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = __builtin_amdgcn_s_memrealtime();
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  unsigned long long* d; (void)hipMalloc(&d, 4 * sizeof(unsigned long long));
  hipStream_t s; (void)hipStreamCreate(&s);
  const int blocks = 1024;  // one workgroup per SIMD or so: block 0 of either kernel lands on XCD 0 (same clock)
  for (int warm = 0; warm < 50; warm++) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(128), 0, s, d, 2000ull, 0);
  (void)hipStreamSynchronize(s);
  const double xs[] = {0, 2, 5, 8, 10, 12, 14, 16, 18, 20, 22, 25, 30};
  for (double X : xs) {
    std::vector<double> gaps;
    for (int rep = 0; rep < 200; rep++) {
      hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(128), 0, s, d, 2000ull, 0);
      const double t0 = now_us();
      while (now_us() - t0 < X) {}
      hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(128), 0, s, d + 2, 200ull, 1);
      (void)hipStreamSynchronize(s);
      unsigned long long h[4];
      (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
      gaps.push_back(((double)h[2] - (double)h[1]) / 100.0);
    }
    std::sort(gaps.begin(), gaps.end());
    printf("B enqueued %5.1f us after A's launch call returned: B.start - A.end  median %6.2f us  (p10 %6.2f  p90 %6.2f)\n", X, gaps[100], gaps[20], gaps[180]);
  }
  return 0;
}

// late_enqueue.hip with the API loop's ingredients added one by one: (a) the null stream, (b) every workgroup of A stores a tagged
// word into pinned host memory 8 us into its run and the host launches B when it has seen all of them (instead of after a fixed
// wait), (c) a chain of such launches (each one is the next one's A).  Prints start-to-start and gap.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <vector>
__global__ void k_spin(unsigned long long* out, unsigned long long ticks, unsigned int* host_words, unsigned int tag, unsigned long long send_at, float* dirty, int dirty_floats) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  // (optionally dirty the L2 like a step kernel does: dirty_floats per workgroup, read-modify-write)
  for (int i = threadIdx.x; i < dirty_floats; i += blockDim.x) dirty[(size_t)blockIdx.x * dirty_floats + i] += 1.0f;
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t0;
  bool sent = host_words == nullptr;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    if (!sent && __builtin_amdgcn_s_memrealtime() - t0 >= send_at) {
      if (threadIdx.x == 0) __hip_atomic_store(host_words + blockIdx.x, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      sent = true;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = __builtin_amdgcn_s_memrealtime();
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int blocks = 1024, N = 400;
  unsigned long long* d; (void)hipMalloc(&d, 2 * N * sizeof(unsigned long long));
  unsigned int* hw; (void)hipHostMalloc(&hw, blocks * sizeof(unsigned int), hipHostMallocDefault);
  unsigned int* hw_dev; (void)hipHostGetDevicePointer((void**)&hw_dev, hw, 0);
  for (int i = 0; i < blocks; i++) hw[i] = 0;
  hipStream_t created; (void)hipStreamCreate(&created);
  float* dirty; (void)hipMalloc(&dirty, (size_t)blocks * 4096 * sizeof(float)); (void)hipMemset(dirty, 0, (size_t)blocks * 4096 * sizeof(float));
  for (int cfg = 0; cfg < 8; cfg++) {
    if (cfg & 1) continue;
    const int dirty_floats = (cfg & 4) ? 3584 : 0;  // 14 KB per workgroup = 14 MB per launch
    hipStream_t s = (cfg & 1) ? nullptr : created;
    const bool host_words = (cfg & 2) != 0;
    for (int warm = 0; warm < 20; warm++) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(128), 0, s, d, 2000ull, (unsigned int*)nullptr, 0u, 0ull, dirty, 0);
    (void)hipStreamSynchronize(s);
    unsigned int tag = 1000 * (cfg + 1);
    for (int i = 0; i < N; i++) {
      tag++;
      hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(128), 0, s, d + 2 * i, 2000ull, host_words ? hw_dev : (unsigned int*)nullptr, tag, 800ull, dirty, dirty_floats);
      if (host_words) {
        volatile unsigned int* v = hw;
        for (int b = 0; b < blocks; b++) while (v[b] != tag) {}
      } else {
        const double t0 = now_us();
        while (now_us() - t0 < 12.0) {}
      }
    }
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> h(2 * N);
    (void)hipMemcpy(h.data(), d, 2 * N * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> gap, per;
    for (int i = 50; i + 1 < N; i++) { gap.push_back(((double)h[2 * i + 2] - (double)h[2 * i + 1]) / 100.0); per.push_back(((double)h[2 * i + 2] - (double)h[2 * i]) / 100.0); }
    std::sort(gap.begin(), gap.end()); std::sort(per.begin(), per.end());
    printf("%s %s stream, next launch %s: gap median %5.2f us (p90 %5.2f), start to start median %5.2f us\n", (cfg & 4) ? "14 MB dirtied," : "nothing dirtied,", (cfg & 1) ? "null   " : "created",
           host_words ? "when all 1024 host words (stored 8 us in) are seen" : "12 us after the previous launch call        ", gap[gap.size() / 2], gap[gap.size() * 9 / 10], per[per.size() / 2]);
  }
  return 0;
}

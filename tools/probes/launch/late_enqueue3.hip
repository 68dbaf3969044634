// What at a kernel boundary costs time in a host <-> device ping-pong (see late_enqueue2.hip)?  Kernel = 20 us spin, 1024 workgroups;
// variants: how the 14 MB of per-launch results are stored (plain / nontemporal / write-through sc0 sc1), and which workgroups store
// a tagged word into pinned host memory 8 us in (all 1024, one per 64-byte line, one in all).  The host launches the next kernel when
// it has seen the words (or 12 us after the previous launch when there are none).  Prints the idle gap between consecutive kernels.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <vector>
__device__ __forceinline__ void store_wt(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory"); }
__global__ void k_spin(unsigned long long* out, unsigned long long ticks, unsigned int* host_words, unsigned int tag, unsigned long long send_at, int word_stride,
                       float* dirty, int dirty_floats, int store_kind, unsigned int* ticket) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = threadIdx.x; i < dirty_floats; i += blockDim.x) {
    float* p = dirty + (size_t)blockIdx.x * dirty_floats + i;
    const float v = (float)(tag + i);
    if (store_kind == 0) *p = v;
    else if (store_kind == 1) __builtin_nontemporal_store(v, p);
    else store_wt(p, v);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t0;
  const bool use_ticket = word_stride < 0;
  bool sent = host_words == nullptr || (!use_ticket && word_stride < 100 && (blockIdx.x % word_stride) != 0);
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    if (!sent && __builtin_amdgcn_s_memrealtime() - t0 >= send_at) {
      if (threadIdx.x == 0) {
        if (use_ticket) {  // every workgroup takes a ticket (device-scope atomic); the last one stores ONE word into host memory
          const unsigned int old = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (old == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(host_words, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          }
        } else {
          const int K = word_stride >= 100 ? word_stride - 100 : 16;
          __hip_atomic_store(host_words + (blockIdx.x / K) * 16 + (blockIdx.x % K), tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      sent = true;
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = __builtin_amdgcn_s_memrealtime();
}
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int blocks = 1024, N = 400;
  unsigned long long* d; (void)hipMalloc(&d, 2 * N * sizeof(unsigned long long));
  unsigned int* hw; (void)hipHostMalloc(&hw, blocks * 64, hipHostMallocDefault);
  unsigned int* hw_dev; (void)hipHostGetDevicePointer((void**)&hw_dev, hw, 0);
  for (int i = 0; i < blocks * 16; i++) hw[i] = 0;
  hipStream_t s; (void)hipStreamCreate(&s);
  float* dirty; (void)hipMalloc(&dirty, (size_t)blocks * 4096 * sizeof(float)); (void)hipMemset(dirty, 0, (size_t)blocks * 4096 * sizeof(float));
  unsigned int tag = 1;
  unsigned int* ticket; (void)hipMalloc(&ticket, 64); (void)hipMemset(ticket, 0, 64);
  const int strides[] = {0, 1, 116, 108, 104, 102, 101};
  for (int store_kind = 0; store_kind < 1; store_kind++)
    for (int ws = 0; ws < 7; ws++) {
      const int word_stride = strides[ws];
      for (int i = 0; i < N; i++) {
        tag++;
        hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(128), 0, s, d + 2 * i, 2000ull, word_stride ? hw_dev : (unsigned int*)nullptr, tag, 800ull, word_stride ? word_stride : 1, dirty, 3584, store_kind, ticket);
        if (word_stride) {
          volatile unsigned int* v = hw;
          if (word_stride < 0) { while (v[0] != tag) {} }
          else if (word_stride >= 100) { const int K = word_stride - 100; for (int b = 0; b < blocks; b++) while (v[(b / K) * 16 + (b % K)] != tag) {} }
          else for (int b = 0; b < blocks; b += word_stride) while (v[b] != tag) {}
        } else {
          const double t0 = now_us();
          while (now_us() - t0 < 12.0) {}
        }
      }
      (void)hipStreamSynchronize(s);
      std::vector<unsigned long long> h(2 * N);
      (void)hipMemcpy(h.data(), d, 2 * N * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      std::vector<double> gap, per;
      for (int i = 50; i + 1 < N; i++) per.push_back(((double)h[2 * i + 2] - (double)h[2 * i]) / 100.0);
      std::sort(per.begin(), per.end());
      for (int i = 50; i + 1 < N; i++) gap.push_back(((double)h[2 * i + 2] - (double)h[2 * i + 1]) / 100.0);
      std::sort(gap.begin(), gap.end());
      printf("14 MB stored %-13s | host words: %-16s | gap median %5.2f us (p90 %5.2f) | start to start %5.2f\n", store_kind == 0 ? "plain" : store_kind == 1 ? "nontemporal" : "sc0 sc1",
             word_stride == 0 ? "none" : word_stride == 1 ? "all 1024" : word_stride == 16 ? "one per line (64)" : word_stride < 0 ? "ticket + 1 word" : word_stride == 116 ? "1024, 16 per line" : word_stride == 108 ? "1024, 8 per line" : word_stride == 104 ? "1024, 4 per line" : word_stride == 102 ? "1024, 2 per line" : word_stride == 101 ? "1024, 1 per line" : "one", gap[gap.size() / 2], gap[gap.size() * 9 / 10], per[per.size() / 2]);
    }
  return 0;
}

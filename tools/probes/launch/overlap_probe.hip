// Can the tail of launch k overlap the head of launch k+1 when the dependency between them is per WORKGROUP, not per kernel?
// 1024 workgroups x 128 threads x 40 KB of LDS (the 16-lane step kernel's footprint: every slot of the chip taken by one launch).
// A workgroup "works" (spins) for `base` us, a few (15 in 1000, different ones every launch) for `slow` us -- the step kernel's
// straggler profile.  Modes:
//   0  one stream, plain launches (what env.step does today): every launch waits for the slowest workgroup of the previous one;
//   1  two streams alternating, workgroup i of launch n spins until workgroup i of launch n-1 has stored n-1 into done[i]
//      (bounded spin: 2 ms, counted as a time-out);
//   2  as 1, and every workgroup also writes / reads 4 KB of "state" through plain stores and sc0 loads (is the data of the
//      predecessor visible when its flag is?  mismatches are counted) and logs its XCC id (does workgroup i stay on one XCD?).
// Prints us per launch over 400 launches for each mode.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>

struct Args {
  uint32_t* done;      // [blocks] last finished launch number per workgroup
  uint32_t* stats;     // [0] time-outs, [1] data mismatches, [2] XCC changes
  uint32_t* xcc;       // [blocks] XCC id of the workgroup's previous launch (+1; 0 = none)
  float* state;        // [blocks][1024] floats
  uint32_t seq;
  int base_ticks, slow_ticks, mode;
  uint32_t* started;   // pinned host memory: workgroups that have begun their work, all launches (the step kernel's early bytes)
  int send_ticks;      // ... stored this long after the workgroup's start
};

// (workgroup i of two consecutive launches on different streams does NOT run on the same XCD -- measured with this probe: 429 056 changes in
//  430 080 workgroup-launches -- so flags and state go through memory: write-through stores (sc0 sc1), loads that bypass the L2 (sc0 sc1))
__device__ __forceinline__ void st_wt(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void stu_wt(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ uint32_t ld_sc0(const uint32_t* p) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ float ldf_sc0(const float* p) {
  float v;
  asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

__global__ __launch_bounds__(128) void k(Args a) {
  __shared__ float lds[10000];  // 40 KB
  const int wg = blockIdx.x, tid = threadIdx.x;
  lds[tid] = (float)tid;
  if (a.mode >= 1) {
    if (tid == 0) {
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      while (ld_sc0(a.done + wg) != a.seq - 1u) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > 200000ull) { atomicAdd(a.stats, 1u); break; }  // 2 ms at 100 MHz
        __builtin_amdgcn_s_sleep(4);
      }
    }
    __syncthreads();
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (a.mode == 2) {
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    id &= 15u;
    if (tid == 0) {
      const uint32_t prev = a.xcc[wg];
      if (prev != 0u && prev != id + 1u) atomicAdd(a.stats + 2, 1u);
      a.xcc[wg] = id + 1u;
    }
    // the predecessor's state: written with plain stores before its flag, read here with sc0 loads
    float* st = a.state + (size_t)wg * 1024;
    int bad = 0;
    for (int i = tid; i < 1024; i += 128) {
      const float v = ldf_sc0(st + i);
      if (a.seq > 1u && v != (float)(a.seq - 1u) + (float)i) bad++;
    }
    if (bad) atomicAdd(a.stats + 1, (uint32_t)bad);
  }
  const uint32_t h = (uint32_t)wg * 2654435761u + a.seq * 40503u;
  const unsigned long long ticks = (h >> 8) % 1000u < 15u ? a.slow_ticks : a.base_ticks;
  bool sent = false;
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
    lds[tid] += 1.0f;
    if (!sent && tid == 0 && __builtin_amdgcn_s_memrealtime() - t0 >= (unsigned long long)a.send_ticks) {
      __hip_atomic_store(a.started + (size_t)wg * 16, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // one 64-byte line per workgroup
      sent = true;
    }
  }
  if (a.mode == 2) {  // new state, at the very end (like the step kernel's scratch rows)
    float* st = a.state + (size_t)wg * 1024;
    for (int i = tid; i < 1024; i += 128) st_wt(st + i, (float)a.seq + (float)i);
  }
  __builtin_amdgcn_s_waitcnt(0);  // this wave's stores have been acknowledged
  __syncthreads();  // ... and every wave's
  if (tid == 0) {
    if (lds[5] < 0.0f) a.stats[3] = 1u;
    stu_wt(a.done + wg, a.seq);
  }
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const int blocks = 1024, N = 400;
  Args a{};
  (void)hipMalloc(&a.done, blocks * 4); (void)hipMalloc(&a.stats, 16); (void)hipMalloc(&a.xcc, blocks * 4);
  (void)hipMalloc(&a.state, (size_t)blocks * 1024 * 4);
  uint32_t* hs; (void)hipHostMalloc(&hs, (size_t)blocks * 64, hipHostMallocDefault);
  for (int i = 0; i < blocks * 16; i++) hs[i] = 0;
  (void)hipHostGetDevicePointer((void**)&a.started, hs, 0);
  a.send_ticks = 800;
  hipStream_t s[2];
  (void)hipStreamCreate(&s[0]); (void)hipStreamCreate(&s[1]);
  a.base_ticks = 1400; a.slow_ticks = 1900;  // 14 us / 19 us at 100 MHz
  for (int mode = 0; mode < 3; mode++) {
    for (int gap_us : {0, 1}) {  // 0: launches enqueued back to back; 1: launch n+1 enqueued when every workgroup of launch n has sent its word (8 us into its work)
      if (mode >= 1 && gap_us == 0) continue;  // (a launch enqueued before every workgroup of its predecessor is resident can take the slots they need)
      (void)hipMemset(a.done, 0, blocks * 4); (void)hipMemset(a.stats, 0, 16); (void)hipMemset(a.xcc, 0, blocks * 4);
      (void)hipDeviceSynchronize();
      for (int i = 0; i < blocks * 16; i++) hs[i] = 0;
      a.mode = mode;
      double t0 = 0;
      for (int n = 1; n <= N + 20; n++) {
        if (n == 21) { (void)hipDeviceSynchronize(); t0 = now_us(); }
        a.seq = (uint32_t)n;
        const double tl = now_us();
        hipLaunchKernelGGL(k, dim3(blocks), dim3(128), 0, s[mode == 0 ? 0 : n & 1], a);
        if (gap_us) {  // the host's wait for the early bytes of launch n
          for (int w = 0; w < blocks; w++) while (*(volatile uint32_t*)(hs + (size_t)w * 16) != (uint32_t)n) { if (now_us() - tl > 5000.0) break; }
        }
      }
      (void)hipDeviceSynchronize();
      const double us = (now_us() - t0) / N;
      uint32_t st[4];
      (void)hipMemcpy(st, a.stats, 16, hipMemcpyDeviceToHost);
      printf("mode %d, %s: %6.2f us per launch | time-outs %u, stale state words %u, XCC changes %u\n", mode, gap_us ? "next launch when every workgroup has sent its word" : "launches enqueued back to back", us, st[0], st[1], st[2]);
    }
  }
  return 0;
}

// barrier_bench.hip — how fast can 256 resident workgroups (one per CU, 8 XCDs) meet on MI355X?
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/barrier_bench.hip -o /tmp/bb && /tmp/bb
// Used to choose the grid barrier of reg_loop_kernel (warpsense_amd/csrc/registration.hip); numbers in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

#define CK(x)                                                                  \
  do                                                                           \
  {                                                                            \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess)                                                       \
    {                                                                          \
      printf("%s failed: %s\n", #x, hipGetErrorString(e));                     \
      return 1;                                                                \
    }                                                                          \
  } while (0)

constexpr int BLOCKS = 256, THREADS = 256, SLOTS = 32;

// variant 0: counter, relaxed atomic add after a release fence; poll with relaxed agent loads, acquire fence after
// variant 1: same but without the fences (what do the fences cost?)
// variant 2: per-workgroup flags (plain release store), every lane polls one flag
// variant 3: counter + each workgroup then reads all 256 x 32 partials (the real pattern)
// variant 4: flags + partial reads
// variant 5: counter, polling with an atomic RMW (fetch_add 0) instead of a load
// variant 6: NO cache-wide fences: partials written with agent-scope relaxed atomic stores (write-through, sc1),
//            s_waitcnt before the arrival, partials read with agent-scope relaxed atomic loads; data is verified
// variant 7: counter, release fence only      variant 8: counter, acquire fence only
template <int V>
__global__ __launch_bounds__(THREADS) void bar_kernel(uint32_t *bar, uint32_t *flags, int64_t *partials, int iters, int64_t *sink)
{
  __shared__ int64_t red[THREADS];
  int64_t acc = 0;
  for (int k = 1; k <= iters; ++k)
  {
    // "work": write this workgroup's partial
    if (V == 6)
    {
      if (threadIdx.x < SLOTS)
        __hip_atomic_store(&partials[(size_t)(k & 1) * BLOCKS * SLOTS + blockIdx.x * SLOTS + threadIdx.x], (int64_t)(k + threadIdx.x), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    else if (V >= 3 && V < 5 && threadIdx.x < SLOTS)
      partials[(size_t)(k & 1) * BLOCKS * SLOTS + blockIdx.x * SLOTS + threadIdx.x] = k + threadIdx.x;
    if (V == 2 || V == 4)
    {
      if (threadIdx.x < 64)
      {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], (uint32_t)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      int guard = 0;
      while (__hip_atomic_load(&flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)k && ++guard < 100000000) {}
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    else
    {
      if (threadIdx.x < 64)
      {
        if (V == 6) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // s_waitcnt only: the stores above have left the CU
        else if (V != 1 && V != 8) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (threadIdx.x == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if (threadIdx.x == 0)
      {
        const uint32_t target = (uint32_t)k * BLOCKS;
        int guard = 0;
        if (V == 5)
          while (__hip_atomic_fetch_add(bar, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++guard < 100000000) {}
        else
          while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++guard < 100000000) {}
      }
      __syncthreads();
      if (V != 1 && V != 6 && V != 7) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    if (V == 6)
    {
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      int64_t *base = partials + (size_t)(k & 1) * BLOCKS * SLOTS + ((size_t)wave * 64 + (size_t)(lane & 1) * 32) * SLOTS + (lane >> 1);
      int64_t s = 0;
#pragma unroll
      for (int i = 0; i < 32; ++i) s += __hip_atomic_load(&base[(size_t)i * SLOTS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (s != 32ll * (k + (lane >> 1))) acc += 1; // stale data
      // the next barrier (k+1) must not overwrite buffer (k+1)&1 == (k-1)&1 before everyone has read it: it cannot,
      // a workgroup arrives at k+1 only after its reads of k
    }
    else if (V >= 3 && V < 5)
    {
      // every workgroup sums all partials: lane l, slot l>>1 ... same access pattern as sum_partials
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      const int64_t *base = partials + (size_t)(k & 1) * BLOCKS * SLOTS + ((size_t)wave * 64 + (size_t)(lane & 1) * 32) * SLOTS + (lane >> 1);
      int64_t s = 0;
#pragma unroll
      for (int i = 0; i < 32; ++i) s += base[(size_t)i * SLOTS];
      red[threadIdx.x] = s;
      __syncthreads();
      acc += red[(threadIdx.x * 7) & 255];
      __syncthreads();
    }
  }
  if (V == 6)
  {
    if (acc != 0) atomicAdd((unsigned long long *)&sink[1], (unsigned long long)acc);
  }
  else if (acc == 0x7fffffffffffll)
    sink[0] = acc;
}

template <int V>
int run(const char *name, uint32_t *bar, uint32_t *flags, int64_t *partials, int64_t *sink)
{
  const int iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep)
  {
    CK(hipMemset(bar, 0, 64));
    CK(hipMemset(flags, 0, BLOCKS * 4));
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(bar_kernel<V>, dim3(BLOCKS), dim3(THREADS), 0, 0, bar, flags, partials, iters, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep == 1) printf("%-58s %8.3f us per barrier\n", name, ms * 1000.0 / iters);
    if (rep == 1 && V == 6)
    {
      int64_t h[2] = {0, 0};
      CK(hipMemcpy(h, sink, 16, hipMemcpyDeviceToHost));
      printf("    stale sums seen: %lld (must be 0)\n", (long long)h[1]);
    }
  }
  return 0;
}

// variants on top of 6 (sc1 partials, no cache-wide fences):
//  NC > 0: NC arrival counters 256 B apart (workgroup b bumps counter b % NC), lanes 0..NC-1 poll
//  NC == 0: one flag per workgroup (plain sc1 store), all 256 lanes poll one flag each
template <int NC>
__global__ __launch_bounds__(THREADS) void bar2_kernel(uint32_t *bar, uint32_t *flags, int64_t *partials, int iters, int64_t *sink)
{
  int64_t acc = 0;
  for (int k = 1; k <= iters; ++k)
  {
    if (threadIdx.x < SLOTS)
      __hip_atomic_store(&partials[(size_t)(k & 1) * BLOCKS * SLOTS + blockIdx.x * SLOTS + threadIdx.x], (int64_t)(k + threadIdx.x), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 64)
    {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the wg-scope fence emits no wait on gfx950
      if (threadIdx.x == 0)
      {
        if (NC > 0)
          __hip_atomic_fetch_add(&bar[(blockIdx.x % (NC > 0 ? NC : 1)) * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          __hip_atomic_store(&flags[blockIdx.x], (uint32_t)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (NC > 0)
    {
      if (threadIdx.x < 64)
      {
        const uint32_t target = (uint32_t)k * (BLOCKS / (NC > 0 ? NC : 1));
        int guard = 0;
        for (;;)
        {
          const bool ok = threadIdx.x >= NC || __hip_atomic_load(&bar[(threadIdx.x % (NC > 0 ? NC : 1)) * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
          if (__all(ok) || ++guard > 100000000) break;
        }
      }
    }
    else
    {
      int guard = 0;
      while (__hip_atomic_load(&flags[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)k && ++guard < 100000000) {}
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t *base = partials + (size_t)(k & 1) * BLOCKS * SLOTS + ((size_t)wave * 64 + (size_t)(lane & 1) * 32) * SLOTS + (lane >> 1);
    int64_t s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += __hip_atomic_load(&base[(size_t)i * SLOTS], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (s != 32ll * (k + (lane >> 1))) acc += 1;
  }
  if (acc != 0) atomicAdd((unsigned long long *)&sink[1], (unsigned long long)acc);
}

template <int NC>
int run2(const char *name, uint32_t *bar, uint32_t *flags, int64_t *partials, int64_t *sink)
{
  const int iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep)
  {
    CK(hipMemset(bar, 0, 64 * 256));
    CK(hipMemset(flags, 0, BLOCKS * 4));
    CK(hipMemset(sink, 0, 16));
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(bar2_kernel<NC>, dim3(BLOCKS), dim3(THREADS), 0, 0, bar, flags, partials, iters, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep == 1)
    {
      int64_t h[2] = {0, 0};
      CK(hipMemcpy(h, sink, 16, hipMemcpyDeviceToHost));
      printf("%-58s %8.3f us per barrier   (stale sums: %lld)\n", name, ms * 1000.0 / iters, (long long)h[1]);
    }
  }
  return 0;
}

// variant: the workgroups ADD their 32 values into one of NG group accumulators (agent-scope atomic add, no return;
// the accumulators are never reset: a reader subtracts what it saw two iterations ago), then arrive; readers fetch
// NG x 32 values instead of 256 x 32.
template <int NG>
__global__ __launch_bounds__(THREADS) void bar3_kernel(uint32_t *bar, int64_t *accum /* [2][NG][SLOTS] */, int iters, int64_t *sink)
{
  __shared__ int64_t prev[2][SLOTS];
  __shared__ int64_t part[THREADS / 64][SLOTS];
  if (threadIdx.x < 2 * SLOTS) prev[threadIdx.x / SLOTS][threadIdx.x % SLOTS] = 0;
  __syncthreads();
  int64_t bad = 0;
  for (int k = 1; k <= iters; ++k)
  {
    if (threadIdx.x < 64)
    {
      if (threadIdx.x < SLOTS)
        __hip_atomic_fetch_add(&accum[((size_t)(k & 1) * NG + (blockIdx.x % NG)) * SLOTS + threadIdx.x], (int64_t)(k + threadIdx.x), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (threadIdx.x == 0) __hip_atomic_fetch_add(&bar[(blockIdx.x % 16) * 64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t target = (uint32_t)k * (BLOCKS / 16);
      int guard = 0;
      for (;;)
      {
        const bool ok = threadIdx.x >= 16 || __hip_atomic_load(&bar[(threadIdx.x % 16) * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target;
        if (__all(ok) || ++guard > 100000000) break;
      }
    }
    __syncthreads();
    // NG x 32 values: lane l of the workgroup reads group (l / 32) + 8 * j, slot l % 32
    int64_t s = 0;
    for (int g = threadIdx.x / SLOTS; g < NG; g += THREADS / SLOTS)
      s += __hip_atomic_load(&accum[((size_t)(k & 1) * NG + g) * SLOTS + (threadIdx.x % SLOTS)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // combine the 8 lanes that hold the same slot: lanes l and l + 32 inside a wave, then across the 4 waves
    s += __shfl_xor(s, 32, 64);
    if ((threadIdx.x & 63) < SLOTS) part[threadIdx.x >> 6][threadIdx.x & 31] = s;
    __syncthreads();
    if (threadIdx.x < SLOTS)
    {
      const int64_t now = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
      const int64_t total = now - prev[k & 1][threadIdx.x];
      prev[k & 1][threadIdx.x] = now;
      if (total != (int64_t)BLOCKS * (k + threadIdx.x)) bad += 1;
    }
    __syncthreads();
  }
  if (bad != 0) atomicAdd((unsigned long long *)&sink[1], (unsigned long long)bad);
}

template <int NG>
int run3(const char *name, uint32_t *bar, int64_t *accum, int64_t *sink)
{
  const int iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep)
  {
    CK(hipMemset(bar, 0, 64 * 256));
    CK(hipMemset(accum, 0, 2 * 64 * SLOTS * 8));
    CK(hipMemset(sink, 0, 16));
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL(bar3_kernel<NG>, dim3(BLOCKS), dim3(THREADS), 0, 0, bar, accum, iters, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep == 1)
    {
      int64_t h[2] = {0, 0};
      CK(hipMemcpy(h, sink, 16, hipMemcpyDeviceToHost));
      printf("%-58s %8.3f us per barrier   (wrong totals: %lld)\n", name, ms * 1000.0 / iters, (long long)h[1]);
    }
  }
  return 0;
}

// variant: NO arrival counter at all.  A 64-bit value travels as two words (low / high 32 bits) whose top byte counts
// the additions: word += (1 << 56) | half.  A reader knows the word it saw two iterations ago (same parity buffer), so
// (now - then) >> 56 is the number of workgroups that have added since and the low 56 bits are their exact sum
// (16 x 2^32 per group never reaches bit 56): it polls the DATA until the count is complete.  Removes the wait for the
// adds' acknowledgement, the arrival atomic and the separate read from the critical path.
template <int NG, bool WAVE0_ONLY>
__global__ __launch_bounds__(THREADS) void bar4_kernel(uint64_t *accum /* [2][NG][64] */, int iters, int64_t *sink)
{
  __shared__ uint64_t part[THREADS / 64][64];
  constexpr int POLLERS = WAVE0_ONLY ? 64 : THREADS;
  constexpr int PER = NG * 64 / POLLERS;
  static_assert(PER >= 1, "words per lane");
  constexpr uint64_t MASK = (1ull << 56) - 1;
  uint64_t p0[PER], p1[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) p0[j] = p1[j] = 0;
  int64_t bad = 0;
  const int lane = threadIdx.x & 63;
  for (int k = 1; k <= iters; ++k)
  {
    uint64_t *buf = accum + (size_t)(k & 1) * NG * 64;
    const int slot = lane & 31;
    const int64_t v = (int64_t)(k + slot) * ((slot & 1) ? -0x123456789ll : 0x123456789ll);
    if (threadIdx.x < 64)
    {
      const uint32_t half = lane < 32 ? (uint32_t)(uint64_t)v : (uint32_t)((uint64_t)v >> 32);
      __hip_atomic_fetch_add(&buf[(size_t)(blockIdx.x % NG) * 64 + lane], (1ull << 56) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint64_t s = 0;
    if (!WAVE0_ONLY || threadIdx.x < 64)
    {
      uint64_t w[PER];
      int guard = 0;
      for (;;)
      {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < PER; ++j)
        {
          const int g = WAVE0_ONLY ? j : (int)(threadIdx.x >> 6) + (THREADS / 64) * j;
          w[j] = __hip_atomic_load(&buf[(size_t)g * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = ok && ((w[j] - p0[j]) >> 56) == (uint64_t)(BLOCKS / NG);
        }
        if (__all(ok) || ++guard > 100000000) break;
      }
#pragma unroll
      for (int j = 0; j < PER; ++j)
      {
        s += (w[j] - p0[j]) & MASK;
        const uint64_t t = p1[j]; // rotate the two parities
        p1[j] = w[j];
        p0[j] = t;
      }
    }
    part[threadIdx.x >> 6][lane] = s;
    __syncthreads();
    if (threadIdx.x < SLOTS)
    {
      uint64_t lo = 0, hi = 0;
      for (int wv = 0; wv < THREADS / 64; ++wv)
      {
        lo += part[wv][threadIdx.x];
        hi += part[wv][threadIdx.x + 32];
      }
      const uint64_t total = lo + (hi << 32);
      if (total != (uint64_t)BLOCKS * (uint64_t)v) bad += 1;
    }
    __syncthreads();
  }
  if (bad != 0) atomicAdd((unsigned long long *)&sink[1], (unsigned long long)bad);
}

template <int NG, bool WAVE0_ONLY>
int run4(const char *name, uint64_t *accum, int64_t *sink)
{
  const int iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep)
  {
    CK(hipMemset(accum, 0, 2 * 64 * 64 * 8));
    CK(hipMemset(sink, 0, 16));
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((bar4_kernel<NG, WAVE0_ONLY>), dim3(BLOCKS), dim3(THREADS), 0, 0, accum, iters, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep == 1)
    {
      int64_t h[2] = {0, 0};
      CK(hipMemcpy(h, sink, 16, hipMemcpyDeviceToHost));
      printf("%-58s %8.3f us per barrier   (wrong totals: %lld)\n", name, ms * 1000.0 / iters, (long long)h[1]);
    }
  }
  return 0;
}

// counted words, polled by wave 0 only.  MODE 0: re-read only the groups that are still incomplete; MODE 1: the same
// with s_sleep between polls; MODE 2: wave w < NW polls NG / NW groups (NW waves share the polling).
template <int NG, int MODE, int NW>
__global__ __launch_bounds__(THREADS) void bar5_kernel(uint64_t *accum /* [2][NG][64] */, int iters, int64_t *sink)
{
  __shared__ uint64_t part[THREADS / 64][64];
  constexpr int PER = NG / NW;
  constexpr uint64_t MASK = (1ull << 56) - 1;
  uint64_t p0[PER], p1[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) p0[j] = p1[j] = 0;
  int64_t bad = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 1; k <= iters; ++k)
  {
    uint64_t *buf = accum + (size_t)(k & 1) * NG * 64;
    const int slot = lane & 31;
    const int64_t v = (int64_t)(k + slot) * ((slot & 1) ? -0x123456789ll : 0x123456789ll);
    if (threadIdx.x < 64)
    {
      const uint32_t half = lane < 32 ? (uint32_t)(uint64_t)v : (uint32_t)((uint64_t)v >> 32);
      __hip_atomic_fetch_add(&buf[(size_t)(blockIdx.x % NG) * 64 + lane], (1ull << 56) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint64_t s = 0;
    if (wave < NW)
    {
      uint64_t w[PER];
      bool done[PER];
#pragma unroll
      for (int j = 0; j < PER; ++j) done[j] = false;
      int guard = 0;
      for (;;)
      {
        bool all_done = true;
#pragma unroll
        for (int j = 0; j < PER; ++j)
          if (!done[j]) w[j] = __hip_atomic_load(&buf[(size_t)(wave * PER + j) * 64 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < PER; ++j)
          if (!done[j])
          {
            done[j] = __all(((w[j] - p0[j]) >> 56) == (uint64_t)(BLOCKS / NG));
            all_done = all_done && done[j];
          }
        if (all_done || ++guard > 100000000) break;
        if (MODE == 1) __builtin_amdgcn_s_sleep(2);
      }
#pragma unroll
      for (int j = 0; j < PER; ++j)
      {
        s += (w[j] - p0[j]) & MASK;
        const uint64_t t = p1[j];
        p1[j] = w[j];
        p0[j] = t;
      }
    }
    part[wave][lane] = s;
    __syncthreads();
    if (threadIdx.x < SLOTS)
    {
      uint64_t lo = 0, hi = 0;
      for (int wv = 0; wv < NW; ++wv)
      {
        lo += part[wv][threadIdx.x];
        hi += part[wv][threadIdx.x + 32];
      }
      const uint64_t total = lo + (hi << 32);
      if (total != (uint64_t)BLOCKS * (uint64_t)v) bad += 1;
    }
    __syncthreads();
  }
  if (bad != 0) atomicAdd((unsigned long long *)&sink[1], (unsigned long long)bad);
}

template <int NG, int MODE, int NW>
int run5(const char *name, uint64_t *accum, int64_t *sink)
{
  const int iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep)
  {
    CK(hipMemset(accum, 0, 2 * 64 * 64 * 8));
    CK(hipMemset(sink, 0, 16));
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((bar5_kernel<NG, MODE, NW>), dim3(BLOCKS), dim3(THREADS), 0, 0, accum, iters, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep == 1)
    {
      int64_t h[2] = {0, 0};
      CK(hipMemcpy(h, sink, 16, hipMemcpyDeviceToHost));
      printf("%-58s %8.3f us per barrier   (wrong totals: %lld)\n", name, ms * 1000.0 / iters, (long long)h[1]);
    }
  }
  return 0;
}

// counted words, wave 0 polls with 128-bit loads: accumulators laid out [NG / 2][64 lanes][2 groups]
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
template <int NG, int NW>
__global__ __launch_bounds__(THREADS) void bar6_kernel(uint64_t *accum, int iters, int64_t *sink)
{
  __shared__ uint64_t part[THREADS / 64][64];
  constexpr int PER = NG / 2 / NW; // 128-bit loads per lane
  constexpr uint64_t MASK = (1ull << 56) - 1;
  uint64_t p0[2 * PER], p1[2 * PER];
#pragma unroll
  for (int j = 0; j < 2 * PER; ++j) p0[j] = p1[j] = 0;
  int64_t bad = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 1; k <= iters; ++k)
  {
    uint64_t *buf = accum + (size_t)(k & 1) * NG * 64;
    const int slot = lane & 31;
    const int64_t v = (int64_t)(k + slot) * ((slot & 1) ? -0x123456789ll : 0x123456789ll);
    if (threadIdx.x < 64)
    {
      const uint32_t half = lane < 32 ? (uint32_t)(uint64_t)v : (uint32_t)((uint64_t)v >> 32);
      const int g = blockIdx.x % NG;
      __hip_atomic_fetch_add(&buf[((size_t)(g >> 1) * 64 + lane) * 2 + (g & 1)], (1ull << 56) | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint64_t s = 0;
    if (wave < NW)
    {
      uint64_t w[2 * PER];
      int guard = 0;
      for (;;)
      {
        u32x4_t q[PER];
        const uint64_t *src = &buf[((size_t)(wave * PER) * 64 + lane) * 2]; // + j KB
        if constexpr (PER == 2)
          asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:1024 sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(q[0]), "=&v"(q[1]) : "v"(src) : "memory");
        else if constexpr (PER == 4)
          asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                       "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(src) : "memory");
        else
        {
          static_assert(PER == 8, "PER");
          asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                       "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]) : "v"(src) : "memory");
          asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %4, off offset:1024 sc1\n\t"
                       "global_load_dwordx4 %2, %4, off offset:2048 sc1\n\tglobal_load_dwordx4 %3, %4, off offset:3072 sc1\n\ts_waitcnt vmcnt(0)"
                       : "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7]) : "v"(src + 512) : "memory");
        }
        bool ok = true;
#pragma unroll
        for (int j = 0; j < PER; ++j)
        {
          w[2 * j] = (uint64_t)q[j].x | ((uint64_t)q[j].y << 32);
          w[2 * j + 1] = (uint64_t)q[j].z | ((uint64_t)q[j].w << 32);
          ok = ok && ((w[2 * j] - p0[2 * j]) >> 56) == (uint64_t)(BLOCKS / NG) && ((w[2 * j + 1] - p0[2 * j + 1]) >> 56) == (uint64_t)(BLOCKS / NG);
        }
        if (__all(ok) || ++guard > 2000000) break;
      }
#pragma unroll
      for (int j = 0; j < 2 * PER; ++j)
      {
        s += (w[j] - p0[j]) & MASK;
        const uint64_t t = p1[j];
        p1[j] = w[j];
        p0[j] = t;
      }
    }
    part[wave][lane] = s;
    __syncthreads();
    if (threadIdx.x < SLOTS)
    {
      uint64_t lo = 0, hi = 0;
      for (int wv = 0; wv < NW; ++wv)
      {
        lo += part[wv][threadIdx.x];
        hi += part[wv][threadIdx.x + 32];
      }
      const uint64_t total = lo + (hi << 32);
      if (total != (uint64_t)BLOCKS * (uint64_t)v) bad += 1;
    }
    __syncthreads();
  }
  if (bad != 0) atomicAdd((unsigned long long *)&sink[1], (unsigned long long)bad);
}

template <int NG, int NW>
int run6(const char *name, uint64_t *accum, int64_t *sink)
{
  const int iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int rep = 0; rep < 2; ++rep)
  {
    CK(hipMemset(accum, 0, 2 * 64 * 64 * 8));
    CK(hipMemset(sink, 0, 16));
    CK(hipEventRecord(a, 0));
    hipLaunchKernelGGL((bar6_kernel<NG, NW>), dim3(BLOCKS), dim3(THREADS), 0, 0, accum, iters, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    if (rep == 1)
    {
      int64_t h[2] = {0, 0};
      CK(hipMemcpy(h, sink, 16, hipMemcpyDeviceToHost));
      printf("%-58s %8.3f us per barrier   (wrong totals: %lld)\n", name, ms * 1000.0 / iters, (long long)h[1]);
    }
  }
  return 0;
}

int main()
{
  uint32_t *bar, *flags;
  int64_t *partials, *sink;
  CK(hipMalloc((void **)&bar, 64 * 256));
  CK(hipMalloc((void **)&flags, BLOCKS * 4));
  CK(hipMalloc((void **)&partials, 2 * BLOCKS * SLOTS * 8));
  CK(hipMalloc((void **)&sink, 16));
  CK(hipMemset(sink, 0, 16));
  if (run<0>("counter + fences", bar, flags, partials, sink)) return 1;
  if (run<1>("counter, no fences", bar, flags, partials, sink)) return 1;
  if (run<5>("counter, RMW polling", bar, flags, partials, sink)) return 1;
  if (run<2>("per-workgroup flags + fences", bar, flags, partials, sink)) return 1;
  if (run<3>("counter + fences + all-read partials (64 KB per workgroup)", bar, flags, partials, sink)) return 1;
  if (run<4>("flags + fences + all-read partials", bar, flags, partials, sink)) return 1;
  if (run<7>("counter, release fence only", bar, flags, partials, sink)) return 1;
  if (run<8>("counter, acquire fence only", bar, flags, partials, sink)) return 1;
  if (run<6>("counter, sc1 stores/loads of partials, no cache-wide fence", bar, flags, partials, sink)) return 1;
  if (run2<1>("sc1 partials, 1 counter", bar, flags, partials, sink)) return 1;
  if (run2<4>("sc1 partials, 4 counters", bar, flags, partials, sink)) return 1;
  if (run2<8>("sc1 partials, 8 counters", bar, flags, partials, sink)) return 1;
  if (run2<16>("sc1 partials, 16 counters", bar, flags, partials, sink)) return 1;
  if (run2<64>("sc1 partials, 64 counters", bar, flags, partials, sink)) return 1;
  if (run2<0>("sc1 partials, 256 flags", bar, flags, partials, sink)) return 1;
  int64_t *accum;
  CK(hipMalloc((void **)&accum, 2 * 64 * SLOTS * 8));
  if (run3<8>("atomic adds into 8 group accumulators, 16 counters", bar, accum, sink)) return 1;
  if (run3<16>("atomic adds into 16 group accumulators, 16 counters", bar, accum, sink)) return 1;
  if (run3<32>("atomic adds into 32 group accumulators, 16 counters", bar, accum, sink)) return 1;
  if (run3<64>("atomic adds into 64 group accumulators, 16 counters", bar, accum, sink)) return 1;
  uint64_t *counted;
  CK(hipMalloc((void **)&counted, 2 * 64 * 64 * 8));
  if (run4<4, false>("counted words, 4 groups, all waves poll", counted, sink)) return 1;
  if (run4<8, false>("counted words, 8 groups, all waves poll", counted, sink)) return 1;
  if (run4<16, false>("counted words, 16 groups, all waves poll", counted, sink)) return 1;
  if (run4<32, false>("counted words, 32 groups, all waves poll", counted, sink)) return 1;
  if (run4<64, false>("counted words, 64 groups, all waves poll", counted, sink)) return 1;
  if (run4<4, true>("counted words, 4 groups, wave 0 polls", counted, sink)) return 1;
  if (run4<8, true>("counted words, 8 groups, wave 0 polls", counted, sink)) return 1;
  if (run4<16, true>("counted words, 16 groups, wave 0 polls", counted, sink)) return 1;
  if (run6<8, 1>("counted, 8 groups, wave 0, 128-bit loads", counted, sink)) return 1;
  if (run6<8, 2>("counted, 8 groups, 2 waves, 128-bit loads", counted, sink)) return 1;
  if (run6<16, 1>("counted, 16 groups, wave 0, 128-bit loads", counted, sink)) return 1;
  if (run6<4, 1>("counted, 4 groups, wave 0, 128-bit loads", counted, sink)) return 1;
  if (run4<8, true>("counted words, 8 groups, wave 0 polls (again)", counted, sink)) return 1;
  if (run5<8, 0, 1>("counted, 8 groups, wave 0, re-read incomplete only", counted, sink)) return 1;
  if (run5<8, 1, 1>("counted, 8 groups, wave 0, incomplete only + sleep", counted, sink)) return 1;
  if (run5<8, 0, 2>("counted, 8 groups, 2 waves, incomplete only", counted, sink)) return 1;
  if (run5<16, 0, 1>("counted, 16 groups, wave 0, incomplete only", counted, sink)) return 1;
  if (run5<16, 0, 2>("counted, 16 groups, 2 waves, incomplete only", counted, sink)) return 1;
  if (run5<16, 1, 2>("counted, 16 groups, 2 waves, incomplete only + sleep", counted, sink)) return 1;
  if (run5<4, 0, 1>("counted, 4 groups, wave 0, incomplete only", counted, sink)) return 1;
  if (run5<2, 0, 1>("counted, 2 groups, wave 0, incomplete only", counted, sink)) return 1;
  return 0;
}

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
  return 0;
}

// xcd_exchange.hip — VERDICT r4 #5a: what would the exchange of the resident registration loop cost if the loop ran on the
// workgroups of ONE XCD (its 44 words meeting in that XCD's L2) instead of on all eight (meeting at the coherent level)?
// The exchange of reg_loop_kernel in isolation: every participating workgroup adds 64 counted words (32 int64 as low / high
// halves, the top byte counts the additions) into one of NG group accumulators, wave 0 polls all NG x 64 words until every
// count is complete.  256 workgroups are launched (one per CU); the ones whose XCC_ID is below `xcds` take part.
//   scope 0: agent-scope atomics and loads (sc1: performed at the coherent level -- what the loop does today)
//   scope 1: atomics performed in the XCD's L2 (no sc bits), polled with returning atomic adds of zero (also in the L2) -- valid ONLY when
//            all participants share one L2, i.e. xcds == 1
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_exchange.hip -o tools/xcd_exchange.out && tools/xcd_exchange.out
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int BLOCKS = 256, THREADS = 512;
__device__ __forceinline__ uint32_t xcc_id()
{
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

template <int SCOPE>
__device__ __forceinline__ void add_word(uint64_t *p, uint64_t v)
{
  if (SCOPE == 0)
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else
    asm volatile("global_atomic_add_x2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
template <int SCOPE>
__device__ __forceinline__ uint64_t load_word(const uint64_t *p)
{
  uint64_t v;
  if (SCOPE == 0)
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else
  {
    // (a load that only bypasses the L1 -- sc0 -- still hit a stale L1 line and the poll never ended: the XCD-local poll is a
    // RETURNING atomic add of zero, which is performed in the L2 like the adds it waits for)
    const uint64_t zero = 0;
    asm volatile("global_atomic_add_x2 %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p), "v"(zero) : "memory");
  }
  return v;
}

// NG groups; participants = workgroups on XCDs < xcds, numbered by a counter so that groups fill evenly
template <int SCOPE, int NG>
__global__ __launch_bounds__(THREADS) void exch_kernel(uint64_t *accum /* [2][NG][64] */, uint32_t *census /* [0] participants, [1] ticket */, int xcds, int iters,
                                                        int first_sleep, long long *out)
{
  __shared__ uint32_t s_rank, s_n;
  const uint32_t xcd = xcc_id();
  const bool takes_part = (int)xcd < xcds;
  if (threadIdx.x == 0)
  {
    s_rank = takes_part ? atomicAdd(&census[1], 1u) : 0u;
    __threadfence();
    atomicAdd(&census[2], 1u);
    // everybody (participant or not) learns the number of participants once all 256 have voted
    while (__hip_atomic_load(&census[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (uint32_t)BLOCKS) {}
    s_n = __hip_atomic_load(&census[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!takes_part) return;
  const uint32_t n = s_n, rank = s_rank;
  const uint32_t g = rank % NG;
  // participants per group (n is a multiple of NG for 32 / 64 / 256 participants and NG <= 8)
  const uint64_t per_group = n / NG;
  constexpr uint64_t MASK = (1ull << 56) - 1;
  const int lane = threadIdx.x & 63;
  uint64_t p0[NG], p1[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) p0[j] = p1[j] = 0;
  long long bad = 0;
  const long long t0 = wall_clock64();
  for (int k = 1; k <= iters; ++k)
  {
    uint64_t *buf = accum + (size_t)(k & 1) * NG * 64;
    const int slot = lane & 31;
    const int64_t v = (int64_t)(k + slot) * ((slot & 1) ? -0x123456789ll : 0x123456789ll);
    if (threadIdx.x < 64)
    {
      const uint32_t half = lane < 32 ? (uint32_t)(uint64_t)v : (uint32_t)((uint64_t)v >> 32);
      add_word<SCOPE>(&buf[(size_t)g * 64 + lane], (1ull << 56) | half);
      // first poll a little later (registration.hip: polling at once delays the very adds it waits for)
      for (int s = 0; s < first_sleep; ++s) __builtin_amdgcn_s_sleep(1);
      uint64_t w[NG];
      int guard = 0;
      for (;;)
      {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < NG; ++j)
        {
          w[j] = load_word<SCOPE>(&buf[(size_t)j * 64 + lane]);
          ok = ok && ((w[j] - p0[j]) >> 56) == per_group;
        }
        if (__all(ok) || ++guard > 20000) break;
        __builtin_amdgcn_s_sleep(2);
      }
      uint64_t s = 0;
#pragma unroll
      for (int j = 0; j < NG; ++j)
      {
        s += (w[j] - p0[j]) & MASK;
        const uint64_t t = p1[j];
        p1[j] = w[j];
        p0[j] = t;
      }
      // check: low halves in lanes 0..31, high halves in lanes 32..63
      const uint64_t hi = __shfl(s, (lane & 31) + 32, 64), lo = __shfl(s, lane & 31, 64);
      if (lane < 32 && lo + (hi << 32) != (uint64_t)n * (uint64_t)v) bad += 1;
    }
    __syncthreads();
  }
  const long long t1 = wall_clock64();
  if (rank == 0 && threadIdx.x == 0)
  {
    out[0] = t1 - t0;
    out[1] = n;
  }
  if (bad) atomicAdd((unsigned long long *)&out[2], (unsigned long long)bad);
}

template <int SCOPE, int NG>
int run(const char *name, int xcds, int first_sleep)
{
  uint64_t *accum;
  uint32_t *census;
  long long *out, h[3];
  CK(hipMalloc((void **)&accum, 2 * NG * 64 * 8));
  CK(hipMalloc((void **)&census, 64));
  CK(hipMalloc((void **)&out, 32));
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep)
  {
    CK(hipMemset(accum, 0, 2 * NG * 64 * 8));
    CK(hipMemset(census, 0, 64));
    CK(hipMemset(out, 0, 32));
    hipLaunchKernelGGL((exch_kernel<SCOPE, NG>), dim3(BLOCKS), dim3(THREADS), 0, 0, accum, census, xcds, iters, first_sleep, out);
    CK(hipDeviceSynchronize());
  }
  CK(hipMemcpy(h, out, 24, hipMemcpyDeviceToHost));
  printf("%-64s %3lld workgroups: %7.3f us per exchange   (wrong totals: %lld)\n", name, h[1], h[0] * 0.01 / iters, h[2]);
  fflush(stdout);
  hipFree(accum);
  hipFree(census);
  hipFree(out);
  return 0;
}

int main(int argc, char **argv)
{
  if (argc > 1 && argv[1][0] == 'g')
  {
    // the number of group accumulators for all 256 workgroups: fewer adds per word against more lines to poll
    for (int sl : {14, 28})
    {
      printf("first poll after %d x 64 clocks, 8 XCDs, coherent level\n", sl);
      if (run<0, 2>("2 groups of 128", 8, sl)) return 1;
      if (run<0, 4>("4 groups of 64", 8, sl)) return 1;
      if (run<0, 8>("8 groups of 32  [the loop today]", 8, sl)) return 1;
      if (run<0, 16>("16 groups of 16", 8, sl)) return 1;
      if (run<0, 32>("32 groups of 8", 8, sl)) return 1;
    }
    return 0;
  }
  for (int sl : {0, 14, 28})
  {
    printf("first poll after %d x 64 clocks\n", sl);
    fflush(stdout);
    if (run<0, 8>("8 XCDs, coherent level (sc1), 8 groups  [the loop today]", 8, sl)) return 1;
    if (run<0, 8>("2 XCDs, coherent level (sc1), 8 groups", 2, sl)) return 1;
    if (run<0, 8>("1 XCD,  coherent level (sc1), 8 groups", 1, sl)) return 1;
    if (run<0, 2>("1 XCD,  coherent level (sc1), 2 groups", 1, sl)) return 1;
    if (run<1, 8>("1 XCD,  in its L2 (atomics without sc, polled by atomics), 8 groups", 1, sl)) return 1;
    if (run<1, 2>("1 XCD,  in its L2 (atomics without sc, polled by atomics), 2 groups", 1, sl)) return 1;
    if (run<1, 1>("1 XCD,  in its L2 (atomics without sc, polled by atomics), 1 group", 1, sl)) return 1;
  }
  return 0;
}

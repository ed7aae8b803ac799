// atomic_bench.hip — throughput of 64-bit atomicMin on MI355X for the address patterns of the keyed scatter pass.
//   hipcc --offload-arch=gfx950 -O3 tools/atomic_bench.hip -o /tmp/ab && /tmp/ab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

// PATTERN 6/7: a lane stays on one random line for 4/8 successive operations (temporal, not spatial, locality).
// PATTERN 0: every lane a random word; 1: a wave hits 64 consecutive words at a random place; 2: a wave hits 8 runs of
// 8 consecutive words; 3: like 1 but 32-bit words; 4: like 1 with plain stores instead of atomics; 5: like 0, 32-bit
template <int PATTERN>
__global__ __launch_bounds__(256) void k(uint64_t *a, uint32_t *a32, uint64_t n_words, int per_lane)
{
  const uint64_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  for (int i = 0; i < per_lane; ++i)
  {
    const uint64_t r = mix(wave * 1000003ull + i);
    uint64_t idx;
    if (PATTERN == 6) idx = ((mix(mix(wave * 1000003ull + (i >> 2)) + lane * 7919ull) % (n_words / 16)) * 16) + (mix(r) & 15); // 4 successive ops of a lane on one line
    else if (PATTERN == 7) idx = ((mix(mix(wave * 1000003ull + (i >> 3)) + lane * 7919ull) % (n_words / 16)) * 16) + (mix(r) & 15); // 8 successive
    else if (PATTERN == 0 || PATTERN == 5) idx = mix(r + lane * 7919ull) % n_words;
    else if (PATTERN == 2) idx = (mix(r + (lane >> 3)) % (n_words - 8)) + (lane & 7);
    else idx = (r % (n_words - 64)) + lane;
    const uint64_t key = r + lane;
    if (PATTERN == 3 || PATTERN == 5) atomicMin(&a32[idx], (uint32_t)key);
    else if (PATTERN == 4) a[idx] = key;
    else atomicMin((unsigned long long *)&a[idx], (unsigned long long)key);
  }
}

template <int P>
int run(const char *name, uint64_t *a, uint64_t n_words)
{
  const int per_lane = 64, blocks = 256 * 32;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
  {
    CK(hipMemset(a, 0xff, n_words * 8));
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<P>, dim3(blocks), dim3(256), 0, 0, a, (uint32_t *)a, n_words, per_lane);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)blocks * 256 * per_lane;
    if (rep == 1) printf("%-64s %8.1f G ops/s  (%.0f M ops in %.3f ms)\n", name, ops / ms / 1e6, ops / 1e6, ms);
  }
  return 0;
}

int main()
{
  const uint64_t n_words = 270ull << 20; // 2.16 GB of 8-byte words, like kpos + kneg of the 513^3 map
  uint64_t *a;
  CK(hipMalloc((void **)&a, n_words * 8));
  if (run<0>("64-bit atomicMin, every lane a random word", a, n_words)) return 1;
  if (run<2>("64-bit atomicMin, 8 runs of 8 consecutive words per wave", a, n_words)) return 1;
  if (run<1>("64-bit atomicMin, 64 consecutive words per wave", a, n_words)) return 1;
  if (run<5>("32-bit atomicMin, every lane a random word", a, n_words)) return 1;
  if (run<3>("32-bit atomicMin, 64 consecutive words per wave", a, n_words)) return 1;
  if (run<4>("64-bit plain store, 64 consecutive words per wave", a, n_words)) return 1;
  if (run<6>("64-bit atomicMin, random line per lane, 4 successive ops on it", a, n_words)) return 1;
  if (run<7>("64-bit atomicMin, random line per lane, 8 successive ops on it", a, n_words)) return 1;
  return 0;
}

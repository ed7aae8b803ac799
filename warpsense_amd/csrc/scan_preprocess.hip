// scan_preprocess.hip — App::preprocess on the device (src/warpsense/app.cpp:119-148).
//
// Per point of the sensor cloud (float metres): drop it if x, y and z are all below 0.3 (:129-132, the
// reference's test is on the signed values), scale to millimetres, snap to the centre of its map voxel
// (:134-140), transform with the fixed-point pose matrix (transform_point, util/util.h:20-34), and keep every
// resulting integer point once (the reference collects them in an unordered_set, :123,142-146).
//
// The reference does this on one host thread with a hash set per scan.  Here: one lane per point, a 64-bit
// open-addressing hash set in HBM (atomicCAS on the packed point, atomicMin on the index of its first
// occurrence), and a stable three-kernel compaction — so the output is the input order with later duplicates
// removed, which is deterministic (the reference's order is whatever its hash set iterates in).
#include "ws_device.h"

namespace ws
{
constexpr uint64_t PRE_EMPTY = ~0ull;
constexpr uint32_t PRE_NONE = 0xffffffffu;
constexpr int32_t PRE_COORD_LIMIT = 1 << 20; // |coordinate| < 2^20 mm (1 km) so three of them fit a 64-bit key

struct PreArgs
{
  const float *xyz; // n points, `stride` floats apart, x y z first
  uint32_t n;
  uint32_t stride;
  int32_t M[16]; // to_int_mat(pose), column-major
  int32_t res;
  int32_t *tmp;      // [n][3] transformed points
  uint32_t *slot_of; // [n] hash slot of the point, PRE_NONE for dropped points
  uint64_t *keys;    // [mask + 1]
  uint32_t *first;   // [mask + 1] smallest input index with this key
  uint32_t mask;
  uint32_t *wg_count; // [blocks]
  uint32_t *wg_off;   // [blocks]
  uint32_t *counters; // [0] points kept, [1] error bits (1: coordinate out of range)
  int32_t *out;       // [kept][3]
};

__device__ __forceinline__ uint64_t pre_mix(uint64_t x)
{
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return x;
}

__global__ __launch_bounds__(256) void pre_insert_kernel(PreArgs a)
{
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= a.n) return;
  const float x = a.xyz[(size_t)i * a.stride + 0], y = a.xyz[(size_t)i * a.stride + 1], z = a.xyz[(size_t)i * a.stride + 2];
  uint32_t slot = PRE_NONE;
  // app.cpp:129-132 compares the floats with the double literal 0.3; non-finite input (undefined in the reference:
  // float -> int conversion of NaN) is dropped
  const bool finite = isfinite(x) && isfinite(y) && isfinite(z);
  const bool near = (double)x < 0.3 && (double)y < 0.3 && (double)z < 0.3;
  if (finite && !near)
  {
    const float res = (float)a.res;
    const float half = (float)(a.res / 2);
    // Pointf(x * 1000.f, ...); (int)(floor(p / res) * res + res / 2), all in float (app.cpp:134-140)
    const int32_t cx = (int32_t)(floorf((x * 1000.f) / res) * res + half);
    const int32_t cy = (int32_t)(floorf((y * 1000.f) / res) * res + half);
    const int32_t cz = (int32_t)(floorf((z * 1000.f) / res) * res + half);
    int32_t q[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
      q[r] = wadd(wadd(wadd(wmul(a.M[0 * 4 + r], cx), wmul(a.M[1 * 4 + r], cy)), wmul(a.M[2 * 4 + r], cz)), a.M[3 * 4 + r]) / MATRIX_RESOLUTION;
    a.tmp[3 * (size_t)i + 0] = q[0];
    a.tmp[3 * (size_t)i + 1] = q[1];
    a.tmp[3 * (size_t)i + 2] = q[2];
    bool in_range = true;
#pragma unroll
    for (int r = 0; r < 3; ++r) in_range = in_range && q[r] > -PRE_COORD_LIMIT && q[r] < PRE_COORD_LIMIT;
    if (!in_range)
    {
      atomicOr(&a.counters[1], 1u);
    }
    else
    {
      const uint64_t key = ((uint64_t)(uint32_t)(q[0] + PRE_COORD_LIMIT) << 42) | ((uint64_t)(uint32_t)(q[1] + PRE_COORD_LIMIT) << 21) |
                           (uint64_t)(uint32_t)(q[2] + PRE_COORD_LIMIT);
      uint32_t h = (uint32_t)pre_mix(key) & a.mask;
      for (;;)
      {
        const uint64_t prev = atomicCAS((unsigned long long *)&a.keys[h], (unsigned long long)PRE_EMPTY, (unsigned long long)key);
        if (prev == PRE_EMPTY || prev == key) break;
        h = (h + 1) & a.mask; // the table has at least twice as many slots as points
      }
      atomicMin(&a.first[h], i);
      slot = h;
    }
  }
  a.slot_of[i] = slot;
}

__device__ __forceinline__ bool pre_keep(const PreArgs &a, uint32_t i)
{
  if (i >= a.n) return false;
  const uint32_t s = a.slot_of[i];
  return s != PRE_NONE && a.first[s] == i;
}

__global__ __launch_bounds__(256) void pre_count_kernel(PreArgs a)
{
  __shared__ uint32_t part[4];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const unsigned long long m = __ballot(pre_keep(a, i));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) a.wg_count[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// exclusive scan of the per-workgroup counts (one workgroup; n_blocks <= 4096 for the 1 000 000-point limit)
__global__ __launch_bounds__(1024) void pre_scan_kernel(PreArgs a, uint32_t n_blocks, uint32_t *host_count)
{
  __shared__ uint32_t sums[1024];
  const uint32_t per = (n_blocks + 1023u) / 1024u;
  const uint32_t lo = threadIdx.x * per;
  uint32_t s = 0;
  for (uint32_t k = 0; k < per; ++k)
    if (lo + k < n_blocks) s += a.wg_count[lo + k];
  sums[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024u; d <<= 1)
  {
    const uint32_t v = threadIdx.x >= d ? sums[threadIdx.x - d] : 0u;
    __syncthreads();
    sums[threadIdx.x] += v;
    __syncthreads();
  }
  uint32_t run = sums[threadIdx.x] - s; // exclusive
  for (uint32_t k = 0; k < per; ++k)
    if (lo + k < n_blocks)
    {
      a.wg_off[lo + k] = run;
      run += a.wg_count[lo + k];
    }
  if (threadIdx.x == 1023)
  {
    a.counters[0] = sums[1023];
    if (host_count) __hip_atomic_store(host_count, sums[1023], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

__global__ __launch_bounds__(256) void pre_scatter_kernel(PreArgs a)
{
  __shared__ uint32_t part[4];
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const bool keep = pre_keep(a, i);
  const unsigned long long m = __ballot(keep);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) part[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (!keep) return;
  uint32_t off = a.wg_off[blockIdx.x];
  for (int w = 0; w < wave; ++w) off += part[w];
  off += (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  a.out[3 * (size_t)off + 0] = a.tmp[3 * (size_t)i + 0];
  a.out[3 * (size_t)off + 1] = a.tmp[3 * (size_t)i + 1];
  a.out[3 * (size_t)off + 2] = a.tmp[3 * (size_t)i + 2];
}

size_t pre_table_slots(size_t max_points)
{
  size_t s = 1024;
  while (s < 2 * max_points) s <<= 1;
  return s;
}

int launch_scan_preprocess(ws_scan *sc, const float *xyz_dev, size_t n, size_t stride, const int32_t M[16], int32_t res)
{
  hipStream_t s = sc->ctx->stream;
  WS_HIP(hipMemsetAsync(sc->counters, 0, 2 * sizeof(uint32_t), s));
  *(volatile uint32_t *)sc->host_count = 0;
  if (n == 0) return WS_OK;
  WS_HIP(hipMemsetAsync(sc->keys, 0xff, sc->table_slots * sizeof(uint64_t), s));
  WS_HIP(hipMemsetAsync(sc->first, 0xff, sc->table_slots * sizeof(uint32_t), s));
  PreArgs a;
  a.xyz = xyz_dev;
  a.n = (uint32_t)n;
  a.stride = (uint32_t)stride;
  for (int k = 0; k < 16; ++k) a.M[k] = M[k];
  a.res = res;
  a.tmp = sc->tmp;
  a.slot_of = sc->slot_of;
  a.keys = sc->keys;
  a.first = sc->first;
  a.mask = (uint32_t)(sc->table_slots - 1);
  a.wg_count = sc->wg_count;
  a.wg_off = sc->wg_off;
  a.counters = sc->counters;
  a.out = sc->out;
  const uint32_t blocks = (uint32_t)((n + 255) / 256);
  hipLaunchKernelGGL(pre_insert_kernel, dim3(blocks), dim3(256), 0, s, a);
  hipLaunchKernelGGL(pre_count_kernel, dim3(blocks), dim3(256), 0, s, a);
  hipLaunchKernelGGL(pre_scan_kernel, dim3(1), dim3(1024), 0, s, a, blocks, sc->host_count_dev);
  hipLaunchKernelGGL(pre_scatter_kernel, dim3(blocks), dim3(256), 0, s, a);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

} // namespace ws

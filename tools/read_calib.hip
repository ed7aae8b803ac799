// read_calib.hip — what does the FETCH_SIZE counter (rocprofv3 --pmc FETCH_SIZE, KB) report per byte LOADED for the load shapes of
// the scatter's kernels?  (MI355X_MICROARCH.md §HBM: on gfx950 a wide coalesced stream counts at half its bytes; VERDICT r5 weak
// #10a: tools/make_traffic.py doubled integrate_dense only, while tile_resolve reads avg_map as 16 B per lane too and its records
// as 8 B per lane.)  The sibling of tools/write_calib.hip; each kernel reads a KNOWN number of bytes from a 1 GiB buffer:
//   rcal_stream16     16 B per lane, coalesced                                   (integrate_dense)
//   rcal_stream8      8 B per lane, coalesced                                    (a run of records)
//   rcal_stream4      4 B per lane, coalesced
//   rcal_run16_256    16 B per lane in runs of 256 B at random places            (tile_resolve: four voxels of a column per thread, a tile's z-runs)
//   rcal_run8_256     8 B per lane in runs of 256 B at random places             (tile_resolve: one sub-chunk per half-wave)
//   rcal_scatter8     8 B per lane at a random 8-byte slot
//   rcal_scatter1     1 B per lane at a random byte                              (march_free's voxel bytes)
// Run: tools/read_calib.sh (builds, runs under rocprofv3, prints bytes counted per byte loaded).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr size_t BUF = 1ull << 30;      // 1 GiB: four times the Infinity Cache
constexpr uint32_t N = 1u << 22;        // loads per kernel (4 M lanes)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// (the value is used: a sum that is stored only if it has an impossible value)
#define SINK(v) if ((v) == 0x12345678u) out[0] = (v)

__global__ void rcal_stream16(const uint4 *p, uint32_t *out) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; const uint4 v = p[i]; SINK(v.x + v.y + v.z + v.w); }
__global__ void rcal_stream8(const uint2 *p, uint32_t *out) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; const uint2 v = p[i]; SINK(v.x + v.y); }
__global__ void rcal_stream4(const uint32_t *p, uint32_t *out) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; const uint32_t v = p[i]; SINK(v); }
__global__ void rcal_run16_256(const uint4 *p, uint32_t *out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t run = mix(i >> 4) & (uint32_t)(BUF / 256 - 1); // 16 lanes x 16 B
  const uint4 v = p[run * 16 + (i & 15)];
  SINK(v.x + v.y + v.z + v.w);
}
__global__ void rcal_run8_256(const uint2 *p, uint32_t *out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t run = mix(i >> 5) & (uint32_t)(BUF / 256 - 1); // 32 lanes x 8 B
  const uint2 v = p[run * 32 + (i & 31)];
  SINK(v.x + v.y);
}
__global__ void rcal_scatter8(const uint2 *p, uint32_t *out) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; const uint2 v = p[mix(i) & (uint32_t)(BUF / 8 - 1)]; SINK(v.x + v.y); }
__global__ void rcal_scatter1(const uint8_t *p, uint32_t *out) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; const uint32_t v = p[mix(i) & (uint32_t)(BUF - 1)]; if (v == 0x57u) out[0] = v; }

int main()
{
  void *buf;
  uint32_t *out;
  CK(hipMalloc(&buf, BUF + 4096));
  CK(hipMalloc((void **)&out, 64));
  CK(hipMemset(buf, 1, BUF));
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep)
  {
    hipLaunchKernelGGL(rcal_stream16, dim3(N / 256), dim3(256), 0, 0, (const uint4 *)buf, out);
    hipLaunchKernelGGL(rcal_stream8, dim3(N / 256), dim3(256), 0, 0, (const uint2 *)buf, out);
    hipLaunchKernelGGL(rcal_stream4, dim3(N / 256), dim3(256), 0, 0, (const uint32_t *)buf, out);
    hipLaunchKernelGGL(rcal_run16_256, dim3(N / 256), dim3(256), 0, 0, (const uint4 *)buf, out);
    hipLaunchKernelGGL(rcal_run8_256, dim3(N / 256), dim3(256), 0, 0, (const uint2 *)buf, out);
    hipLaunchKernelGGL(rcal_scatter8, dim3(N / 256), dim3(256), 0, 0, (const uint2 *)buf, out);
    hipLaunchKernelGGL(rcal_scatter1, dim3(N / 256), dim3(256), 0, 0, (const uint8_t *)buf, out);
    // (between the repetitions: push the buffer's lines out of the caches again)
    CK(hipMemset(buf, 1, BUF));
    CK(hipDeviceSynchronize());
  }
  printf("loads per kernel %u\n", N);
  return 0;
}

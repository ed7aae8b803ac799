// write_calib.hip — what does the WRITE_SIZE counter (rocprofv3 --pmc WRITE_SIZE, KB) report per store for the store shapes of the
// tail march?  (MI355X_MICROARCH.md §HBM: uncalibrated for anything but wide streams; VERDICT r4 weak #3: march_tail's 350 MB
// for 98 MB of records + 14 M byte marks cannot be read without this.)  Each kernel writes a KNOWN number of bytes:
//   calib_stream16      16 B per lane, coalesced                        (the control: must count its own bytes)
//   calib_rec8_dense    8 B per lane, consecutive                       (a sub-chunk filled at once)
//   calib_rec8_scatter  8 B per lane at a random 8-byte slot of 1 GiB   (a record arriving alone at its sub-chunk's line)
//   calib_rec8_revisit  8 B per lane, 32 visits of each 256 B sub-chunk spread over the kernel (one record per visit, other lines
//                       written in between: the tail march's pattern -- does the L2 merge them before the line leaves?)
//   calib_byte_scatter  1 B per lane at a random byte of 1 GiB          (a KEYED mark)
//   calib_byte_dense    1 B per lane, consecutive
// Run:  hipcc --offload-arch=gfx950 -O3 tools/write_calib.hip -o tools/write_calib.out
//       rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/prof_wcal -o pmc -- tools/write_calib.out ; python tools/pmc_summary.py <db>
// (tools/write_calib.sh does both and prints bytes counted per byte stored.)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr size_t BUF = 1ull << 30;      // 1 GiB: four times the Infinity Cache
constexpr uint32_t N = 1u << 22;        // stores per kernel (4 M)
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__global__ void calib_stream16(uint4 *p) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; p[i] = make_uint4(i, i, i, i); }
__global__ void calib_rec8_dense(unsigned long long *p) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; p[i] = i; }
__global__ void calib_rec8_scatter(unsigned long long *p) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; p[mix(i) & (uint32_t)(BUF / 8 - 1)] = i; }
// N / 32 sub-chunks of 256 B, scattered over the buffer; visit v of the kernel writes record v of every sub-chunk a workgroup owns
__global__ void calib_rec8_revisit(unsigned long long *p)
{
  const uint32_t subs_per_wg = 256; // 64 KB of sub-chunks per workgroup
  for (uint32_t v = 0; v < 32; ++v)
  {
    const uint32_t sub = blockIdx.x * subs_per_wg + threadIdx.x;
    const size_t base = ((size_t)(mix(sub) & (uint32_t)(BUF / 256 - 1))) * 32; // in records
    p[base + v] = sub;
    __builtin_amdgcn_s_sleep(20); // (the tail march computes ~100 instructions between two records of one sub-chunk)
  }
}
__global__ void calib_byte_scatter(uint8_t *p) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; p[mix(i) & (uint32_t)(BUF - 1)] = 1; }
__global__ void calib_byte_dense(uint8_t *p) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; p[i] = 1; }

int main()
{
  void *buf;
  CK(hipMalloc(&buf, BUF + 4096));
  CK(hipMemset(buf, 0, BUF));
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep)
  {
    hipLaunchKernelGGL(calib_stream16, dim3(N / 256), dim3(256), 0, 0, (uint4 *)buf);
    hipLaunchKernelGGL(calib_rec8_dense, dim3(N / 256), dim3(256), 0, 0, (unsigned long long *)buf);
    hipLaunchKernelGGL(calib_rec8_scatter, dim3(N / 256), dim3(256), 0, 0, (unsigned long long *)buf);
    hipLaunchKernelGGL(calib_rec8_revisit, dim3(N / 32 / 256), dim3(256), 0, 0, (unsigned long long *)buf);
    hipLaunchKernelGGL(calib_byte_scatter, dim3(N / 256), dim3(256), 0, 0, (uint8_t *)buf);
    hipLaunchKernelGGL(calib_byte_dense, dim3(N / 256), dim3(256), 0, 0, (uint8_t *)buf);
    CK(hipDeviceSynchronize());
  }
  printf("stores per kernel %u: stream16 %u B, rec8 %u B, bytes %u B\n", N, N * 16, N * 8, N);
  return 0;
}

// ta_bench.hip — what does a vector memory instruction with scattered addresses cost a compute unit?  (The free pass of round 5
// executes 28 % fewer vector ALU instructions than round 4's and takes the same 118 us: its 1.2 M scattered byte loads / stores per
// scan are the bound.)  Every wave issues ITER byte loads (or stores) whose `lanes` active lanes hit random lines of a 32 MB region
// (L2-resident: the free pass's working set), with enough independent waves per CU that latency is hidden; reported: cycles of
// one CU per wave instruction = CUs x clock x time / instructions.
//   hipcc --offload-arch=gfx950 -O3 tools/ta_bench.hip -o tools/ta_bench.out && tools/ta_bench.out
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr uint32_t REGION_MAX = 256u << 20;
constexpr int ITER = 256;
__device__ __forceinline__ uint32_t mix(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE> // 0: byte load, 1: byte store, 2: dword load, 3: byte loads whose 64 lanes share 4 lines
__global__ __launch_bounds__(256) void ta_kernel(uint8_t *buf, uint32_t lanes, uint32_t *sink, uint32_t REGION)
{
  const uint32_t lane = threadIdx.x & 63;
  uint32_t h = mix(blockIdx.x * 256u + threadIdx.x);
  uint32_t acc = 0;
  if (lane < lanes)
  {
#pragma unroll 8
    for (int i = 0; i < ITER; ++i)
    {
      h = h * 1664525u + 1013904223u;
      uint32_t off = (h >> 4) & (REGION - 1);
      if (MODE == 3) off = ((mix(blockIdx.x * 4 + (threadIdx.x >> 6) + i * 977u) & (REGION - 1)) & ~255u) + (lane & 3) * 64 + (lane >> 2);
      if (MODE == 0 || MODE == 3) acc += buf[off];
      if (MODE == 1) buf[off] = (uint8_t)i;
      if (MODE == 2) acc += *reinterpret_cast<const uint32_t *>(buf + (off & ~3u));
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int MODE>
int run(const char *name, uint8_t *buf, uint32_t *sink, uint32_t lanes, int cus, double mhz, uint32_t region = 32u << 20)
{
  const int blocks = cus * 8 * 4; // 8 workgroups of 4 waves per CU, 4 rounds
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL(ta_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, buf, lanes, sink, region);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(ta_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, buf, lanes, sink, region);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double instr = (double)blocks * 4 * ITER;
  printf("%-34s region %4u MB lanes %2u: %8.1f us, %6.1f CU-cycles per wave instruction, %6.2f per active lane\n", name, region >> 20, lanes, ms * 1e3, cus * mhz * 1e6 * ms * 1e-3 / instr,
         cus * mhz * 1e6 * ms * 1e-3 / instr / lanes);
  return 0;
}

int main()
{
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  const double mhz = p.clockRate / 1000.0;
  uint8_t *buf;
  uint32_t *sink;
  CK(hipMalloc((void **)&buf, REGION_MAX + 4096));
  CK(hipMalloc((void **)&sink, 64));
  CK(hipMemset(buf, 0, REGION_MAX));
  printf("%d CUs at %.0f MHz (nominal)\n", cus, mhz);
  for (uint32_t lanes : {64u, 32u, 16u, 4u, 1u})
    if (run<0>("byte load, random lines", buf, sink, lanes, cus, mhz)) return 1;
  for (uint32_t lanes : {64u, 16u, 1u})
    if (run<1>("byte store, random lines", buf, sink, lanes, cus, mhz)) return 1;
  for (uint32_t lanes : {64u, 16u})
    if (run<2>("dword load, random lines", buf, sink, lanes, cus, mhz)) return 1;
  if (run<3>("byte load, 64 lanes in 4 lines", buf, sink, 64, cus, mhz)) return 1;
  // the same scattered byte loads against the size of the region they fall into: 1 MB (inside every XCD's 4 MB L2) ... 256 MB
  for (uint32_t mb : {1u, 2u, 4u, 8u, 16u, 64u, 256u})
    if (run<0>("byte load, random lines", buf, sink, 64, cus, mhz, mb << 20)) return 1;
  for (uint32_t mb : {1u, 4u, 256u})
    if (run<1>("byte store, random lines", buf, sink, 64, cus, mhz, mb << 20)) return 1;
  return 0;
}

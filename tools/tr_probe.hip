#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v2i __attribute__((ext_vector_type(2)));
// What does ds_read_b64_tr_b8 (gfx950) return?  Probe behind mfma_consume (registration.hip): hipcc --offload-arch=gfx950 -O2 tools/tr_probe.hip
// LDS image: 64 "points" x 48 bytes (32 used); byte value at (point p, rowbyte r) = we store 16-bit id in two planes
__global__ void probe(uint32_t *out_lo, uint32_t *out_hi, int plane)
{
  __shared__ __attribute__((aligned(16))) unsigned char img[64 * 48];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 48; i += 64)
  {
    const int p = i / 48, r = i % 48;
    img[i] = plane == 0 ? (unsigned char)p : (unsigned char)r;
  }
  __syncthreads();
  // group g = lane >> 4, j = lane & 15: read 8 contiguous bytes of point (j >> 1), half-row (j & 1), window g & 1
  const int j = lane & 15, g = lane >> 4;
  const int point0 = (g >> 1) * 16;
  const unsigned char *addr = img + (point0 + (j >> 1)) * 48 + (g & 1) * 16 + (j & 1) * 8;
  v2i v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) v2i *)addr);
  out_lo[lane] = (uint32_t)v.x;
  out_hi[lane] = (uint32_t)v.y;
}
int main()
{
  uint32_t *lo, *hi;
  hipMalloc(&lo, 256); hipMalloc(&hi, 256);
  uint32_t hl[64], hh[64];
  for (int plane = 0; plane < 2; ++plane)
  {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, lo, hi, plane);
    hipMemcpy(hl, lo, 256, hipMemcpyDeviceToHost); hipMemcpy(hh, hi, 256, hipMemcpyDeviceToHost);
    printf("plane %d (%s)\n", plane, plane == 0 ? "point index" : "row byte index");
    for (int l = 0; l < 64; ++l)
    {
      printf("lane %2d:", l);
      for (int b = 0; b < 4; ++b) printf(" %3u", (hl[l] >> (8 * b)) & 0xff);
      for (int b = 0; b < 4; ++b) printf(" %3u", (hh[l] >> (8 * b)) & 0xff);
      printf("\n");
    }
  }
  return 0;
}

// doorbell_bench.hip -- what a host <-> resident-kernel round trip costs on this box (round 6: the resident server behind
// ws_reg_iterate, registration.hip: reg_server_kernel).  The host writes a request number, a resident kernel sees it and answers
// by writing the number into host-mapped memory, the host spins on that.  Variants:
//   bell in host-mapped memory (the GPU polls over the fabric)  |  bell in fine-grained DEVICE memory written by the host through the BAR
//   one workgroup answers  |  workgroup 0 forwards the bell through device memory and the LAST of G workgroups to see it answers
//     hipcc --offload-arch=gfx950 -O2 tools/doorbell_bench.hip -o /tmp/doorbell_bench && /tmp/doorbell_bench
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x)                                                                  \
  do                                                                           \
  {                                                                            \
    hipError_t e__ = (x);                                                      \
    if (e__ != hipSuccess)                                                     \
    {                                                                          \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e__));                 \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

struct Args
{
  uint32_t *bell;     // where the host rings (host-mapped or device memory)
  uint32_t *fwd;      // device memory: [0] forwarded bell, [16] arrivals
  uint32_t *answer;   // host-mapped
  uint32_t last;      // leave after this request
  int sleep_host, sleep_dev;
  int wide;           // workgroup 0 reads 16 words per poll (the server's pose line)
};

__global__ __launch_bounds__(512) void server(Args a)
{
  __shared__ uint32_t bell_sh;
  uint32_t served = 0;
  for (;;)
  {
    if (threadIdx.x < 64)
    {
      uint32_t b = 0;
      if (blockIdx.x == 0)
      {
        for (;;)
        {
          uint32_t w = 0;
          if (a.wide ? threadIdx.x < 16 : threadIdx.x == 0) w = __hip_atomic_load(&a.bell[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          b = (uint32_t)__builtin_amdgcn_readlane((int)w, 0);
          if (b != served) break;
          __builtin_amdgcn_s_sleep(4);
          for (int q = 0; q < a.sleep_host; q += 4) __builtin_amdgcn_s_sleep(4);
        }
        if (threadIdx.x == 0)
        {
          if (gridDim.x > 1) __hip_atomic_store(&a.fwd[0], b, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
          bell_sh = b;
        }
      }
      else if (threadIdx.x == 0)
      {
        for (;;)
        {
          b = __hip_atomic_load(&a.fwd[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
          if (b != served) break;
          __builtin_amdgcn_s_sleep(2);
          for (int q = 0; q < a.sleep_dev; q += 4) __builtin_amdgcn_s_sleep(4);
        }
        bell_sh = b;
      }
    }
    __syncthreads();
    const uint32_t b = bell_sh;
    __syncthreads();
    if (threadIdx.x == 0)
    {
      bool last = true;
      if (gridDim.x > 1) last = __hip_atomic_fetch_add(&a.fwd[16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
      if (last)
      {
        if (gridDim.x > 1) __hip_atomic_store(&a.fwd[16], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.answer, b, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    served = b;
    if (b == a.last) return;
  }
}

static double run(uint32_t *bell_host_view, uint32_t *bell_dev_view, uint32_t *fwd, uint32_t *ans_host, uint32_t *ans_dev, int grid, int n, int sleep_host,
                  int sleep_dev, int wide)
{
  volatile uint32_t *bell = bell_host_view;
  volatile uint32_t *ans = ans_host;
  for (int i = 0; i < 16; ++i) bell[i] = 0;
  *ans = 0;
  CK(hipMemset(fwd, 0, 256));
  CK(hipDeviceSynchronize());
  Args a{bell_dev_view, fwd, ans_dev, (uint32_t)n, sleep_host, sleep_dev, wide};
  hipLaunchKernelGGL(server, dim3(grid), dim3(512), 0, 0, a);
  // warm-up requests, then the timed ones
  const int warm = 200;
  std::chrono::steady_clock::time_point t0;
  for (int i = 1; i <= n; ++i)
  {
    if (i == warm) t0 = std::chrono::steady_clock::now();
    std::atomic_thread_fence(std::memory_order_release);
    bell[0] = (uint32_t)i;
    while (*ans != (uint32_t)i)
    {
    }
  }
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (n - warm + 1);
  CK(hipDeviceSynchronize());
  return us;
}

int main()
{
  uint32_t *bell_h = nullptr, *bell_h_dev = nullptr, *ans_h = nullptr, *ans_dev = nullptr, *fwd = nullptr;
  CK(hipHostMalloc((void **)&bell_h, 256, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void **)&bell_h_dev, bell_h, 0));
  CK(hipHostMalloc((void **)&ans_h, 256, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void **)&ans_dev, ans_h, 0));
  CK(hipMalloc((void **)&fwd, 256));
  const int n = 5000;
  printf("round trip host -> resident kernel -> host, us per request (%d requests)\n", n);
  for (int wide = 0; wide < 2; ++wide)
    for (int grid : {1, 32, 256})
      printf("  bell in host-mapped memory, %3d workgroups, %s poll: %6.2f us\n", grid, wide ? "64-byte" : "4-byte ",
             run(bell_h, bell_h_dev, fwd, ans_h, ans_dev, grid, n, 0, 0, wide));
  for (int sd : {4, 16, 64})
    printf("  bell in host-mapped memory, 256 workgroups, device pollers sleep +%d: %6.2f us\n", sd, run(bell_h, bell_h_dev, fwd, ans_h, ans_dev, 256, n, 0, sd, 1));
  // fine-grained device memory, written by the host through the BAR (if this box lets the host touch it)
  uint32_t *bell_d = nullptr;
  if (hipExtMallocWithFlags((void **)&bell_d, 256, hipDeviceMallocFinegrained) == hipSuccess)
  {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof at);
    (void)hipPointerGetAttributes(&at, bell_d);
    printf("  fine-grained device memory: device pointer %p, host pointer %p\n", at.devicePointer, at.hostPointer);
    if (getenv("WS_DOORBELL_BAR"))
    {
      for (int grid : {1, 256})
        printf("  bell in DEVICE memory written by the host, %3d workgroups: %6.2f us\n", grid, run(bell_d, bell_d, fwd, ans_h, ans_dev, grid, n, 0, 0, 1));
    }
  }
  uint32_t *bell_m = nullptr;
  if (hipMallocManaged((void **)&bell_m, 256) == hipSuccess)
  {
    (void)hipMemAdvise(bell_m, 256, hipMemAdviseSetPreferredLocation, 0);
    (void)hipMemAdvise(bell_m, 256, hipMemAdviseSetAccessedBy, hipCpuDeviceId);
    for (int grid : {1, 256})
      printf("  bell in managed memory (preferred location: the GPU), %3d workgroups: %6.2f us\n", grid, run(bell_m, bell_m, fwd, ans_h, ans_dev, grid, n, 0, 0, 1));
  }
  return 0;
}

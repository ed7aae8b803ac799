// valu_rate.hip -- issue rate of the vector instructions the march kernels are made of, per SIMD, on gfx950.
//
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
//
// Question: /opt/skills/guides/MI355X_MICROARCH.md gives 2 cycles per wave64 VALU instruction ("SIMD-32"),
// DESIGN.md r2 argued with 4.  Every workgroup here is 4 waves x WAVES_PER_SIMD per CU (one workgroup of 256 x W threads per CU,
// 256 workgroups), each wave runs N independent chains of ONE instruction kind, unrolled, for a fixed number of iterations;
// cycles come from s_memtime (shader clock) around the loop, the rate is reported per SIMD:
//     cycles per wave-instruction per SIMD = elapsed cycles / (instructions per wave x waves per SIMD)
// With enough waves per SIMD and independent chains the figure is the pipe's issue interval.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                         \
  do                                                                                     \
  {                                                                                      \
    hipError_t e = (x);                                                                  \
    if (e != hipSuccess)                                                                 \
    {                                                                                    \
      fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e));                      \
      exit(1);                                                                           \
    }                                                                                    \
  } while (0)

constexpr int ITERS = 16384, CHAINS = 8;

enum Kind
{
  ADD_U32,
  ADD_CO_CHAIN, // v_add_co_u32 + v_addc_co_u32 pairs (the carry-fed quotient update of the sample step)
  MUL_LO_U32,
  MUL_HI_U32,
  MUL_U24,
  MAD_U64_U32,
  CNDMASK,
  MIN_U32,
  FMA_F32,
  PK_FMA_F32,
  FMA_F64,
  SQRT_F32,
  LSHL_ADD,
  SUB_U32,
  AND_B32,
  XOR_B32,
  LSHRREV,
  ASHRREV,
  CMP_GT,
  ADD3,
  BFE_U32,
  N_KINDS
};
static const char *kind_name[N_KINDS] = {"v_add_u32",   "v_add_co+v_addc_co", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u64_u32", "v_cndmask_b32",
                                         "v_min_u32",   "v_fma_f32",          "v_pk_fma_f32", "v_fma_f64",    "v_sqrt_f32",    "v_lshl_add_u32",
                                         "v_sub_u32",   "v_and_b32",          "v_xor_b32",    "v_lshrrev_b32", "v_ashrrev_i32", "v_cmp_gt_u32 (vcc)", "v_add3_u32", "v_bfe_u32"};

template <int KIND>
__global__ __launch_bounds__(1024) void rate_kernel(unsigned long long *cycles, unsigned *sink, unsigned seed)
{
  unsigned v[CHAINS], w[CHAINS];
  float f[CHAINS];
  double d[CHAINS];
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 p[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; ++c)
  {
    v[c] = seed + threadIdx.x * 7u + c;
    w[c] = seed * 3u + c;
    f[c] = (float)(v[c] & 1023u) * 1e-3f;
    d[c] = (double)f[c];
    p[c] = f2{f[c], f[c] + 1.f};
  }
  const unsigned k1 = seed | 1u, k2 = seed + 12345u;
  const float fk = 1.0000001f;
  const unsigned long long sel = 0x5555aaaa3333ccccull ^ seed;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it)
  {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c)
    {
      if (KIND == ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[c]) : "v"(k1));
      if (KIND == ADD_CO_CHAIN) asm volatile("v_add_co_u32 %0, vcc, %0, %2\n\tv_addc_co_u32 %1, vcc, %1, %3, vcc" : "+v"(v[c]), "+v"(w[c]) : "v"(k1), "v"(k2) : "vcc");
      if (KIND == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[c]) : "v"(k1));
      if (KIND == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[c]) : "v"(k1));
      if (KIND == MUL_U24) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[c]) : "v"(k1));
      if (KIND == MAD_U64_U32)
      {
        unsigned long long acc = ((unsigned long long)w[c] << 32) | v[c];
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(v[c]), "v"(k1) : "vcc");
        v[c] = (unsigned)acc;
        w[c] = (unsigned)(acc >> 32);
      }
      if (KIND == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(k1), "s"(sel));
      if (KIND == MIN_U32) asm volatile("v_min_u32 %0, %0, %1" : "+v"(v[c]) : "v"(k2));
      if (KIND == FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[c]) : "v"(fk));
      if (KIND == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[c]) : "v"(p[(c + 1) % CHAINS]));
      if (KIND == FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[c]) : "v"(d[(c + 1) % CHAINS]));
      if (KIND == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[c]));
      if (KIND == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(v[c]) : "v"(k1));
      if (KIND == SUB_U32) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[c]) : "v"(k1));
      if (KIND == AND_B32) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[c]) : "v"(k2));
      if (KIND == XOR_B32) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[c]) : "v"(k2));
      if (KIND == LSHRREV) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(v[c]));
      if (KIND == ASHRREV) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(v[c]));
      if (KIND == CMP_GT) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(v[c]), "v"(k2) : "vcc");
      if (KIND == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(v[c]) : "v"(k1), "v"(k2));
      if (KIND == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 1, 31" : "+v"(v[c]));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  unsigned acc = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) acc += v[c] + w[c] + (unsigned)f[c] + (unsigned)d[c] + (unsigned)p[c].x;
  if (acc == 0x12345678u) sink[0] = acc;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(int waves_per_simd, unsigned long long *d_cycles, unsigned *d_sink)
{
  // one block of up to 1024 threads per CU (4 waves per SIMD); 8 waves per SIMD = two such blocks per CU
  const int threads = 256 * (waves_per_simd > 4 ? 4 : waves_per_simd), blocks = 256 * (waves_per_simd > 4 ? waves_per_simd / 4 : 1);
  const int n_waves = blocks * threads / 64;
  rate_kernel<KIND><<<blocks, threads>>>(d_cycles, d_sink, 17u);
  CHECK(hipDeviceSynchronize());
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  CHECK(hipEventRecord(a));
  rate_kernel<KIND><<<blocks, threads>>>(d_cycles, d_sink, 19u);
  CHECK(hipEventRecord(b));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  std::vector<unsigned long long> h(n_waves);
  CHECK(hipMemcpy(h.data(), d_cycles, n_waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto c : h) mean += (double)c;
  mean /= n_waves;
  const double per_wave = (double)ITERS * CHAINS * (KIND == ADD_CO_CHAIN ? 2 : 1);
  // the counter of __builtin_readcyclecounter (s_memtime) runs at a constant 100 MHz on this part: use the event time
  const double total_instr = per_wave * n_waves;
  const double rate = total_instr / (ms * 1e-3);                     // wave-instructions / s, whole chip
  const double per_simd_ns = 1e9 / (rate / 1024.0);                  // ns per wave-instruction per SIMD
  printf("%-20s %d waves/SIMD: %8.3f T wave-instr/s chip, %6.3f ns per wave-instruction per SIMD = %5.2f cycles @2.4 GHz, %5.2f @2.1 GHz (kernel %.3f ms)\n",
         kind_name[KIND], waves_per_simd, rate * 1e-12, per_simd_ns, per_simd_ns * 2.4, per_simd_ns * 2.1, ms);
  (void)mean;
  CHECK(hipEventDestroy(a));
  CHECK(hipEventDestroy(b));
}

int main()
{
  unsigned long long *d_cycles;
  unsigned *d_sink;
  CHECK(hipMalloc(&d_cycles, 512 * 16 * sizeof(unsigned long long)));
  CHECK(hipMalloc(&d_sink, 64));
  for (int w : {4, 8})
  {
    run<ADD_U32>(w, d_cycles, d_sink);
    run<ADD_CO_CHAIN>(w, d_cycles, d_sink);
    run<CNDMASK>(w, d_cycles, d_sink);
    run<MIN_U32>(w, d_cycles, d_sink);
    run<LSHL_ADD>(w, d_cycles, d_sink);
    run<MUL_U24>(w, d_cycles, d_sink);
    run<MUL_LO_U32>(w, d_cycles, d_sink);
    run<MUL_HI_U32>(w, d_cycles, d_sink);
    run<MAD_U64_U32>(w, d_cycles, d_sink);
    run<FMA_F32>(w, d_cycles, d_sink);
    run<PK_FMA_F32>(w, d_cycles, d_sink);
    run<FMA_F64>(w, d_cycles, d_sink);
    run<SQRT_F32>(w, d_cycles, d_sink);
    run<SUB_U32>(w, d_cycles, d_sink);
    run<AND_B32>(w, d_cycles, d_sink);
    run<XOR_B32>(w, d_cycles, d_sink);
    run<LSHRREV>(w, d_cycles, d_sink);
    run<ASHRREV>(w, d_cycles, d_sink);
    run<CMP_GT>(w, d_cycles, d_sink);
    run<ADD3>(w, d_cycles, d_sink);
    run<BFE_U32>(w, d_cycles, d_sink);
    printf("\n");
  }
  return 0;
}

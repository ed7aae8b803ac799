// solve_gj_bench.hip — the 6x6 solve of the Gauss-Newton update on one wave, timed as a dependent chain:
//   A  Gauss-Jordan with the pivot's reciprocal (one division per step in the chain)
//   B  cross-multiplied Gauss-Jordan: row_i <- p' * row_i - a_ik' * row_k with p' = p * 2^-E, a_ik' = a_ik * 2^-E
//      (E = exponent of the pivot: exact scalings), no division in the chain, six independent ones at the end
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math tools/solve_gj_bench.hip -o /tmp/sgb && /tmp/sgb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

__device__ __forceinline__ double lane_read(double v, int src_lane)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_gather(double v, int src_lane)
{
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double in_vgpr(double v)
{
  asm volatile("" : "+v"(v));
  return v;
}

template <int V>
__device__ __forceinline__ int solve(double a, double (&x)[6])
{
  const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
  int singular = 0;
  double inv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k)
  {
    double pv, an, rowk, colk;
    const double dk = lane_read(a, 8 * k + k);
    const bool below_larger = c == k && r > k && r < 6 && fabs(a) > fabs(dk);
    if (__ballot(below_larger) == 0ull)
    {
      pv = in_vgpr(dk);
      an = a;
      rowk = lane_gather(a, 8 * k + c);
      colk = lane_gather(a, 8 * r + k);
    }
    else
    {
      int piv = k;
      pv = in_vgpr(lane_read(a, 8 * k + k));
#pragma unroll
      for (int i = k + 1; i < 6; ++i)
      {
        const double v = in_vgpr(lane_read(a, 8 * i + k));
        const bool larger = fabs(v) > fabs(pv);
        pv = larger ? v : pv;
        piv = larger ? i : piv;
      }
      const int rr = r == k ? piv : (r == piv ? k : r);
      an = lane_gather(a, 8 * rr + c);
      rowk = lane_gather(a, 8 * piv + c);
      colk = lane_gather(a, 8 * rr + k);
    }
    singular |= pv == 0.0 ? 1 : 0;
    if (V == 0)
    {
      inv[k] = 1.0 / pv;
      const double f = colk * inv[k];
      a = (r != k && c > k) ? an - f * rowk : an;
    }
    else
    {
      const int E = __builtin_amdgcn_frexp_exp(pv);
      const double ps = __builtin_amdgcn_ldexp(pv, -E), fs = __builtin_amdgcn_ldexp(colk, -E);
      const double t = fs * (c > k ? rowk : 0.0);
      a = (r != k && (c > k || c == r)) ? __builtin_fma(ps, an, -t) : an;
    }
  }
  if (__builtin_amdgcn_readfirstlane(singular) != 0) return -1;
  if (V == 0)
  {
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = lane_read(a, 8 * i + 6) * inv[i];
  }
  else
  {
    const double d = lane_gather(a, 8 * r + r);
    const double q = a / d; // (lanes c == 6, r < 6 are the ones read)
#pragma unroll
    for (int i = 0; i < 6; ++i) x[i] = lane_read(q, 8 * i + 6);
  }
  return 0;
}

template <int V>
__global__ __launch_bounds__(64) void bench_kernel(const double *Ab, double *out, long long *cycles, int iters)
{
  const int lane = threadIdx.x;
  const int r = lane >> 3, c = lane & 7;
  double a0 = (r < 6 && c < 7) ? Ab[r * 7 + c] : (r == c ? 1.0 : 0.0);
  double x[6] = {0, 0, 0, 0, 0, 0};
  double acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it)
  {
    const double a = a0 + acc * 1e-300;
    const int rc = solve<V>(a, x);
    acc += x[0] + x[5] + rc;
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0)
  {
    cycles[0] = t1 - t0;
    for (int i = 0; i < 6; ++i) out[i] = x[i];
  }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int V>
int run(const char *name, const double *dAb, double *dout, long long *dcyc, const long double *ref)
{
  const int iters = 4000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
  {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(bench_kernel<V>, dim3(1), dim3(64), 0, 0, dAb, dout, dcyc, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double out[6];
  CK(hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost));
  double worst = 0;
  for (int i = 0; i < 6; ++i) worst = fmax(worst, fabs((double)((out[i] - ref[i]) / ref[i])));
  printf("%-60s %7.3f us per solve   worst relative error %.3g\n", name, ms * 1000.0 / iters, worst);
  return 0;
}
int main()
{
  double Ab[42];
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
  double M[6][6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) M[i][j] = rnd() * (i < 3 ? 1e6 : 1e2);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 7; ++j)
    {
      double v = 0;
      if (j < 6) for (int k = 0; k < 6; ++k) v += M[k][i] * M[k][j];
      else v = rnd() * 1e7;
      Ab[i * 7 + j] = v;
    }
  // reference in long double (Gaussian elimination, partial pivoting)
  long double L[6][7], ref[6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 7; ++j) L[i][j] = Ab[i * 7 + j];
  for (int k = 0; k < 6; ++k)
  {
    int p = k;
    for (int i = k + 1; i < 6; ++i) if (fabsl(L[i][k]) > fabsl(L[p][k])) p = i;
    for (int j = 0; j < 7; ++j) { long double t = L[k][j]; L[k][j] = L[p][j]; L[p][j] = t; }
    for (int i = 0; i < 6; ++i) if (i != k) { long double f = L[i][k] / L[k][k]; for (int j = 0; j < 7; ++j) L[i][j] -= f * L[k][j]; }
  }
  for (int i = 0; i < 6; ++i) ref[i] = L[i][6] / L[i][i];
  double *dAb, *dout;
  long long *dcyc;
  CK(hipMalloc((void **)&dAb, sizeof(Ab)));
  CK(hipMalloc((void **)&dout, 64));
  CK(hipMalloc((void **)&dcyc, 8));
  CK(hipMemcpy(dAb, Ab, sizeof(Ab), hipMemcpyHostToDevice));
  if (run<0>("A  Gauss-Jordan, pivot reciprocals (the loop today)", dAb, dout, dcyc, ref)) return 1;
  if (run<1>("B  cross-multiplied Gauss-Jordan, no division in the chain", dAb, dout, dcyc, ref)) return 1;
  return 0;
}

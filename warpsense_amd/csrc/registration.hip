// registration.hip — Point-to-TSDF registration for MI355X (gfx950).
//
// Replaces calc_jacobis_krnl + h_g_e_reduction_krnl + the host reduce() of the reference
// (src/warpsense/cuda/registration.cu:14-257,310-368) and moves the Gauss-Newton update of
// cuda::TSDFRegistration::register_cloud (src/warpsense/tsdf_registration.cpp:55-92) onto the device.
//
// One Gauss-Newton iteration, in every workgroup (256 x 512 lanes):
//   phase A  (k > 0) total of the previous iteration's partial sums (exact integer sums -> order independent,
//            bit-identical to the reference's tree); the first wave runs the 6x6 solve (lane-parallel LU in
//            double), xi -> SE(3) and the convergence test exactly like the reference's host code.  Doing this
//            redundantly per workgroup costs nothing extra and saves a broadcast.
//   phase B  fixed-point transform, voxel + 6-neighbour gather, gradient, Jacobian, per-lane accumulation
//            of the 21 unique terms of J J^T, the 6 of J v, |v| and the count in int64 registers
//            (v_mad_i64_i32), a transposing wave64 reduction (32 values in 32 exchanges instead of 32 x 6),
//            LDS across the waves, one 29-word partial per workgroup.
// No Jacobian / value / mask arrays ever reach HBM (the reference writes and re-reads 51 B per point).
//
// reg_loop_kernel (default) runs ALL iterations in one launch: resident workgroups, partial sums exchanged through
// wrapping group accumulators whose words count their additions (the exchange is the barrier), a per-lane voxel cache.
// reg_iter_kernel is one launch per iteration (state and 256 x 32 partials double buffered by launch parity, so no
// fences or atomics are needed); it is the fallback when the grid cannot be resident, and the A/B reference.
#include <cstdlib>
#include <cstring>

#include "ws_device.h"

namespace ws
{
#ifndef WS_REG_BLOCKS
#define WS_REG_BLOCKS 256
#endif
constexpr int REG_BLOCKS = WS_REG_BLOCKS; // one workgroup per CU
#ifndef WS_REG_THREADS
#define WS_REG_THREADS 512
#endif
constexpr int REG_THREADS = WS_REG_THREADS; // 8 waves: one point per lane for a 131 072-point scan (4 waves x 2 points: 10.7 vs 10.1 us)
#ifndef WS_REG_MFMA
#define WS_REG_MFMA 1 // 0: the resident loop sums with v_mad_i64_i32 + the transposing butterfly for every cloud size
#endif
#ifndef WS_SOLVE_DIAG_FIRST
#define WS_SOLVE_DIAG_FIRST 1
#endif
constexpr int REG_TERMS = 29;    // 21 h + 6 g + e + c (slots 29..31 are padding)
static_assert(REG_TERMS <= 32, "slots");
constexpr int REG_SLOTS = 32;    // padded to a power of two for the transposing reduction
static_assert(REG_BLOCKS % (REG_THREADS / 64 * 2) == 0, "sum_partials: every wave sums an equal share of the workgroups, 2 lanes per slot");

__device__ __forceinline__ int64_t shfl_xor_i64(int64_t v, int mask)
{
  int lo = __shfl_xor((int)(uint32_t)((uint64_t)v & 0xffffffffull), mask, 64);
  int hi = __shfl_xor((int)(uint32_t)((uint64_t)v >> 32), mask, 64);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

// ---- transposing wave reduction of 32 int64 values, without the LDS crossbar -------------------------
// Stage `BIT` pairs lane l with lane l ^ BIT: lanes whose BIT is clear keep the lower HALF of the values they
// still carry and receive the partner's lower half, the others keep / receive the upper half, so every stage
// halves the values per lane: 16+8+4+2+1+1 = 32 exchanges for 32 values instead of 32 x 6.
//   BIT 32, 16: gfx950's v_permlane32_swap / v_permlane16_swap exchange exactly those halves of two registers
//               (no select, no address): one VALU instruction per 32-bit register pair;
//   BIT 8 .. 1: DPP lane permutations (row_ror:8, row_half_mirror + quad_perm, quad_perm).
__device__ __forceinline__ int64_t shfl_i64(int64_t v, int src_lane)
{
  const int lo = __shfl((int)(uint32_t)((uint64_t)v & 0xffffffffull), src_lane, 64), hi = __shfl((int)(uint32_t)((uint64_t)v >> 32), src_lane, 64);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}
__device__ __forceinline__ int64_t pack64(int lo, int hi) { return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo); }

template <int BIT>
__device__ __forceinline__ void swap_add_stage(int64_t &a, const int64_t b)
{
  // a: the value kept by lanes with BIT clear, b: kept by lanes with BIT set; result in a (for every lane: its kept slot)
  const int alo = (int)(uint32_t)((uint64_t)a & 0xffffffffull), ahi = (int)(uint32_t)((uint64_t)a >> 32);
  const int blo = (int)(uint32_t)((uint64_t)b & 0xffffffffull), bhi = (int)(uint32_t)((uint64_t)b >> 32);
  if constexpr (BIT == 32)
  {
    const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = wadd64(pack64(lo[0], hi[0]), pack64(lo[1], hi[1]));
  }
  else
  {
    const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = wadd64(pack64(lo[0], hi[0]), pack64(lo[1], hi[1]));
  }
}

template <int BIT>
__device__ __forceinline__ int dpp_xor(int v)
{
  if constexpr (BIT == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, false); // row_ror:8
  if constexpr (BIT == 4)
  {
    const int m = __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false); // row_half_mirror: lane ^ 7
    return __builtin_amdgcn_update_dpp(0, m, 0x1b, 0xf, 0xf, false);         // quad_perm [3,2,1,0]: lane ^ 3
  }
  if constexpr (BIT == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4e, 0xf, 0xf, false); // quad_perm [2,3,0,1]
  return __builtin_amdgcn_update_dpp(0, v, 0xb1, 0xf, 0xf, false);                         // quad_perm [1,0,3,2]
}
template <int BIT>
__device__ __forceinline__ int64_t dpp_xor_i64(int64_t v)
{
  return pack64(dpp_xor<BIT>((int)(uint32_t)((uint64_t)v & 0xffffffffull)), dpp_xor<BIT>((int)(uint32_t)((uint64_t)v >> 32)));
}

// One pair of a transposing stage inside a 16-lane row (BIT 8 or 4): lanes without the bit end up with a + a(partner),
// lanes with it with b + b(partner), both in `a`.  Four DPP adds whose bank masks pick the two halves -- the partner comes
// over the DPP operand of the add itself (row_shl for the lower lanes, row_shr for the upper ones) -- instead of four
// selects, two to four DPP moves and two adds.  (s_nop: a VGPR written by the instruction before must not be read over DPP
// at once, and the compiler does not see what the asm reads.)
template <int BIT>
__device__ __forceinline__ void dpp_pair_add(int64_t &a, const int64_t b)
{
  static_assert(BIT == 8 || BIT == 4, "row-internal stages");
  uint32_t alo = (uint32_t)((uint64_t)a & 0xffffffffull), ahi = (uint32_t)((uint64_t)a >> 32);
  const uint32_t blo = (uint32_t)((uint64_t)b & 0xffffffffull), bhi = (uint32_t)((uint64_t)b >> 32);
  if constexpr (BIT == 8)
    asm volatile("s_nop 1\n\t"
                 "v_add_co_u32_dpp %0, vcc, %0, %0 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                 "v_add_co_u32_dpp %0, vcc, %2, %2 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                 "v_addc_co_u32_dpp %1, vcc, %3, %3, vcc row_shr:8 row_mask:0xf bank_mask:0xc"
                 : "+v"(alo), "+v"(ahi)
                 : "v"(blo), "v"(bhi)
                 : "vcc");
  else
    asm volatile("s_nop 1\n\t"
                 "v_add_co_u32_dpp %0, vcc, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_addc_co_u32_dpp %1, vcc, %1, %1, vcc row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_add_co_u32_dpp %0, vcc, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                 "v_addc_co_u32_dpp %1, vcc, %3, %3, vcc row_shr:4 row_mask:0xf bank_mask:0xa"
                 : "+v"(alo), "+v"(ahi)
                 : "v"(blo), "v"(bhi)
                 : "vcc");
  a = (int64_t)(((uint64_t)ahi << 32) | alo);
}

template <int HALF, int BIT>
__device__ __forceinline__ void reduce_stage(int64_t (&v)[REG_SLOTS], int lane)
{
  if constexpr (BIT >= 16)
  {
#pragma unroll
    for (int i = 0; i < HALF; ++i) swap_add_stage<BIT>(v[i], v[i + HALF]);
  }
  else if constexpr (BIT >= 4)
  {
#pragma unroll
    for (int i = 0; i < HALF; ++i) dpp_pair_add<BIT>(v[i], v[i + HALF]);
  }
  else
  {
    const bool upper = (lane & BIT) != 0;
#pragma unroll
    for (int i = 0; i < HALF; ++i)
    {
      const int64_t send = upper ? v[i] : v[i + HALF];
      const int64_t keep = upper ? v[i + HALF] : v[i];
      v[i] = wadd64(keep, dpp_xor_i64<BIT>(send));
    }
  }
}

// wave totals of REG_SLOTS per-lane values -> wave_part[wave][0..31] (valid after the trailing barrier)
__device__ __forceinline__ void wave_reduce32(int64_t (&v)[REG_SLOTS], int64_t (*wave_part)[REG_SLOTS])
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  reduce_stage<16, 32>(v, lane);
  reduce_stage<8, 16>(v, lane);
  reduce_stage<4, 8>(v, lane);
  reduce_stage<2, 4>(v, lane);
  reduce_stage<1, 2>(v, lane);
  v[0] = wadd64(v[0], dpp_xor_i64<1>(v[0]));
  // lane l now holds the wave total of slot (l >> 1)
  if ((lane & 1) == 0) wave_part[wave][lane >> 1] = v[0];
  __syncthreads();
}

// The same with the eight wave totals of a slot ADDED into wg_sum[slot] (LDS atomics, zero before) instead of laid side by
// side: the first wave then reads one value per slot instead of eight (valid after the trailing barrier).
// (the caller's barrier follows: a wave without points skips this call, not the barrier)
__device__ __forceinline__ void wave_reduce32_add(int64_t (&v)[REG_SLOTS], unsigned long long *wg_sum)
{
  const int lane = threadIdx.x & 63;
  reduce_stage<16, 32>(v, lane);
  reduce_stage<8, 16>(v, lane);
  reduce_stage<4, 8>(v, lane);
  reduce_stage<2, 4>(v, lane);
  reduce_stage<1, 2>(v, lane);
  v[0] = wadd64(v[0], dpp_xor_i64<1>(v[0]));
  if ((lane & 1) == 0) atomicAdd(&wg_sum[lane >> 1], (unsigned long long)v[0]);
}

// Sum REG_SLOTS per-lane values over the whole workgroup. Result: red[0..31] in LDS (valid after the
// trailing barrier).
__device__ __forceinline__ void block_reduce32(int64_t (&v)[REG_SLOTS], int64_t (*wave_part)[REG_SLOTS], int64_t *red)
{
  wave_reduce32(v, wave_part);
  if (threadIdx.x < REG_SLOTS)
  {
    int64_t s = 0;
#pragma unroll
    for (int w = 0; w < REG_THREADS / 64; ++w) s = wadd64(s, wave_part[w][threadIdx.x]);
    red[threadIdx.x] = s;
  }
  __syncthreads();
}

// row-major upper triangle index of (i <= j)
__host__ __device__ constexpr int tri_index(int i, int j) { return i * 6 - (i * (i - 1)) / 2 + (j - i); }

// 29 reduced terms -> the reference's 44 words: h 6x6 column-major (math/matrix6x6.h:112-115), g[6], e, c
__device__ __forceinline__ void expand_sums(const int64_t *terms, int64_t *sums)
{
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int i = 0; i < 6; ++i) sums[j * 6 + i] = terms[i <= j ? tri_index(i, j) : tri_index(j, i)];
#pragma unroll
  for (int i = 0; i < 6; ++i) sums[36 + i] = terms[21 + i];
  sums[42] = (int64_t)(int32_t)terms[27]; // e and c are `int` in the reference (registration.cu:16-21)
  sums[43] = (int64_t)(int32_t)terms[28];
}

// ---- 6x6 solve on one wave: Gauss-Jordan with partial pivoting in double, the same operations in the same order as
// oracle/ws_oracle.c:wso_solve6 (Eigen hf.inverse()*g, tsdf_registration.cpp:69), so the result is bit-identical
// to a serial solve.  Lane 8*r + c holds element (r, c) of the augmented matrix [A | b] (c == 6 is b): the 5
// divisions and the rank-1 update of an elimination step are one instruction each instead of 5 / 35, and no
// element ever needs a dynamic register index (a serial version spills the matrix to scratch for the row swap:
// 2.9 us per solve on one lane).  All 64 lanes of the wave must be active.
//
// The solve is one wave's chain of ~500 instructions in the middle of every Gauss-Newton iteration.  Measured with
// tools/solve_bench.hip (cycles per solve on one wave): what costs is every hop through the scalar unit.  The pivot
// candidates are uniform, so the compiler compares them into an SGPR mask, selects with s_cselect and moves the winner
// back with v_mov -- 70 cycles per candidate, 1050 of 3360 per solve.  Copied into VGPRs behind an opaque asm the
// same search is v_cmp + v_cndmask, ~20 cycles per candidate: 2660 cycles per solve.  (Tried and slower: rows that
// stay in place + DPP instead of two of the three gathers (4130), the reciprocal half of each division hoisted off the
// dependency chain (2860-3150), a tournament instead of the chain (3690): the wave is bound by instruction issue, not
// by the length of the chain.)
__device__ __forceinline__ double lane_read(double v, int src_lane /* uniform */)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_gather(double v, int src_lane /* per lane */)
{
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
// the same value in a vector register the compiler knows nothing about (keeps what follows out of the scalar unit)
__device__ __forceinline__ double in_vgpr(double v)
{
  asm volatile("" : "+v"(v));
  return v;
}

template <typename V>
__device__ __forceinline__ void pin_vgpr(V &v)
{
  static_assert(sizeof(V) == 4, "32-bit values");
  asm volatile("" : "+v"(v));
}

// a: this lane's element of [A | b].  Returns 0 and x (identical in every lane), or -1 for a singular matrix.
// Gauss-Jordan with partial pivoting, the same operations in the same order as oracle/ws_oracle.c:wso_solve6 (round 5; LU +
// back substitution before): every step clears its column in ALL other rows -- in this layout the rows above the pivot cost
// nothing, they are other lanes of the same instruction -- and the multipliers come from the pivot's reciprocal, so after the
// sixth step x[i] = b[i] / pivot i is one multiplication.  Gone: the back substitution's six chained divisions, fifteen
// multiply-subtracts and 42 operand fetches through v_readlane.
__device__ __forceinline__ int solve6_wave(double a, double (&x)[6])
{
  const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
  int singular = 0;
  double inv[6];
#pragma unroll
  for (int k = 0; k < 6; ++k)
  {
    // pivot: first row of maximal |A[i][k]|, i >= k
    double pv, an, rowk, colk;
#if WS_SOLVE_DIAG_FIRST
    // The diagonal element keeps its place unless an element BELOW it is strictly larger (the search takes the first maximum):
    // every lane of the column compares its own element with the diagonal, one ballot decides.  Then nothing changes places
    // -- no candidate chain (8 instructions per candidate), no gather of the swapped row.  The normal equations of a scan put
    // the large rotational terms first, so this is the usual case; otherwise the general search below runs.
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
#endif
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
      // rows k and piv change places; fetch the swapped element, the pivot row and the k-th column in one go
      const int rr = r == k ? piv : (r == piv ? k : r);
      an = lane_gather(a, 8 * rr + c);
      rowk = lane_gather(a, 8 * piv + c);
      colk = lane_gather(a, 8 * rr + k);
    }
    singular |= pv == 0.0 ? 1 : 0; // the exit is taken once, below (x is not used then)
    inv[k] = 1.0 / pv;
    const double f = colk * inv[k];
    a = (r != k && c > k) ? an - f * rowk : an;
  }
  if (__builtin_amdgcn_readfirstlane(singular) != 0) return -1;
#pragma unroll
  for (int i = 0; i < 6; ++i) x[i] = lane_read(a, 8 * i + 6) * inv[i];
  return 0;
}

// One Gauss-Newton update (tsdf_registration.cpp:63-92, registration/util.h:5-39), executed by one whole wave;
// every lane holds the same state and computes the same result.  H(r, c), G(r): the int64 sums.
#ifdef WS_REG_TIMING_GN
__device__ long long g_gn_ticks[5];
#define WS_GN_STAMP(i) const long long gn_t##i = wall_clock64()
#else
#define WS_GN_STAMP(i)
#endif

// First half: solve for xi and build the incremental transform `tr` (column-major 4x4).  false: no update this time
// (loop already over, no correspondences, singular matrix).
// what xi_to_transform (registration/util.h:5-39) needs to build the incremental transform
struct GnStep
{
  float L01, L02, L10, L12, L20, L21; // the skew matrix of the unit axis (all +0 for a zero rotation, like the reference's initialiser)
  float s, omc;                       // (float)sin theta, (float)(1 - cos theta)
  float t3, t4, t5;                   // (float)xi[3..5]
};

// Solve for xi and reduce it to GnStep.  false: no update this time (loop already over, no correspondences, singular matrix).
template <typename HF, typename GF>
__device__ __forceinline__ bool gn_step(GnCore &st, HF H, GF G, int32_t c, GnStep &o)
{
  if (st.finished || st.iterations >= st.max_iterations) return false;
  st.iterations += 1;
  if (c == 0)
  {
    st.finished = 1; // guard: the reference would divide by zero (tsdf_registration.cpp:80)
    return false;
  }
  WS_GN_STAMP(0);
  const double w = (double)(st.alpha * (float)c);
  const int lane = threadIdx.x & 63, lr = lane >> 3, lc = lane & 7;
  double a = 0.0;
  if (lr < 6 && lc < 6) a = (double)H(lr, lc) + (lr == lc ? w : 0.0);
  if (lr < 6 && lc == 6) a = (double)G(lr);
  double xi[6];
  WS_GN_STAMP(1);
  if (solve6_wave(a, xi) != 0)
  {
    st.finished = 1;
    return false;
  }
#pragma unroll
  for (int r = 0; r < 6; ++r) xi[r] = -xi[r];
  WS_GN_STAMP(2);

  // xi_to_transform
  const double theta = sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
  o.L01 = o.L02 = o.L10 = o.L12 = o.L20 = o.L21 = 0.f;
  if (theta != 0.0)
  {
    const double lx = xi[0] / theta, ly = xi[1] / theta, lz = xi[2] / theta;
    o.L01 = (float)-lz; o.L02 = (float)ly;
    o.L10 = (float)lz;  o.L12 = (float)-lx;
    o.L20 = (float)-ly; o.L21 = (float)lx;
  }
  double sin_t, cos_t;
  if (theta < 0.25)
  {
    // Gauss-Newton steps are small rotations: Taylor polynomials (truncation < 1e-19 below 0.25 rad) instead of the
    // library's sincos with its argument reduction (428 -> 184 cycles on the one wave everybody waits for).  What the
    // update uses are (float)sin and (float)(1 - cos): the cosine is summed with its rounding error carried (u + (e + w)),
    // so that 1 - cos_t cancels like the host's correctly rounded cos() does -- against glibc on 20 million angles in
    // [1e-9, 0.25] both floats agree in every case (the plain polynomial misses (float)(1 - cos) in 0.2 % of them).
    // (Horner steps as explicit fused multiply-adds -- half the length of the dependent chain; tools/polycheck.c: both forms
    // give libm's two floats on 20 million angles.)
    const double z = theta * theta;
    double p = fma(z, 1.0 / 6227020800.0, -1.0 / 39916800);
    p = fma(z, p, 1.0 / 362880);
    p = fma(z, p, -1.0 / 5040);
    p = fma(z, p, 1.0 / 120);
    p = fma(z, p, -1.0 / 6);
    sin_t = fma(theta * z, p, theta);
    double q = fma(z, -1.0 / 87178291200.0, 1.0 / 479001600.0);
    q = fma(z, q, -1.0 / 3628800);
    q = fma(z, q, 1.0 / 40320);
    q = fma(z, q, -1.0 / 720);
    q = fma(z, q, 1.0 / 24);
    const double t = 0.5 * z, u = 1.0 - t, e = (1.0 - u) - t, ww = z * z * q;
    cos_t = u + (e + ww);
  }
  else
    sincos(theta, &sin_t, &cos_t); // one argument reduction for both
  o.s = (float)sin_t;
  o.omc = (float)(1 - cos_t);
  o.t3 = (float)xi[3];
  o.t4 = (float)xi[4];
  o.t5 = (float)xi[5];
  WS_GN_STAMP(3);
#ifdef WS_REG_TIMING_GN
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    g_gn_ticks[0] += gn_t1 - gn_t0;
    g_gn_ticks[1] += gn_t2 - gn_t1;
    g_gn_ticks[2] += gn_t3 - gn_t2;
    g_gn_ticks[4] += 1;
  }
#endif
  return true;
}

// First half: solve for xi and build the incremental transform `tr` (column-major 4x4).  false: no update this time
// (loop already over, no correspondences, singular matrix).
template <typename HF, typename GF>
__device__ __forceinline__ bool gn_increment(GnCore &st, HF H, GF G, int32_t c, float (&tr)[16])
{
  GnStep o;
  if (!gn_step(st, H, G, c, o)) return false;
  const float L[3][3] = {{0.f, o.L01, o.L02}, {o.L10, 0.f, o.L12}, {o.L20, o.L21, 0.f}};
  const float s = o.s, omc = o.omc;
  float R[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
    {
      float ll = 0.f;
#pragma unroll
      for (int k = 0; k < 3; ++k) ll = __fadd_rn(ll, __fmul_rn(__fmul_rn(omc, L[i][k]), L[k][j]));
      R[i][j] = __fadd_rn(__fadd_rn((i == j ? 1.f : 0.f), __fmul_rn(s, L[i][j])), ll);
    }
#pragma unroll
  for (int i = 0; i < 16; ++i) tr[i] = 0.f;
  tr[15] = 1.f;
  const float oc0 = -(float)st.center[0], oc1 = -(float)st.center[1], oc2 = -(float)st.center[2];
  const float tx[3] = {o.t3, o.t4, o.t5};
#pragma unroll
  for (int i = 0; i < 3; ++i)
  {
#pragma unroll
    for (int j = 0; j < 3; ++j) tr[j * 4 + i] = R[i][j];
    const float shift = __fadd_rn(__fadd_rn(__fmul_rn(R[i][0], oc0), __fmul_rn(R[i][1], oc1)), __fmul_rn(R[i][2], oc2));
    tr[12 + i] = __fadd_rn(__fadd_rn(shift, (float)st.center[i]), tx[i]);
  }
  st.alpha = __fadd_rn(st.alpha, st.it_weight_gradient);
  return true;
}

// Second half: convergence test on the mean error (tsdf_registration.cpp:80-92)
__device__ __forceinline__ void gn_convergence(GnCore &st, int32_t e, int32_t c)
{
  const float err = __fdiv_rn((float)e, (float)c);
  if (fabsf(err - st.prev[2]) < st.epsilon && fabsf(err - st.prev[0]) < st.epsilon) st.finished = 1;
  st.prev[0] = st.prev[1];
  st.prev[1] = st.prev[2];
  st.prev[2] = st.prev[3];
  st.prev[3] = err;
}

// T = tr * T with every lane computing all 16 elements (uniform state)
__device__ __forceinline__ void pose_product(float (&T)[16], const float (&tr)[16])
{
  float out[16];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
      float acc = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) acc = __fadd_rn(acc, __fmul_rn(tr[k * 4 + i], T[j * 4 + k]));
      out[j * 4 + i] = acc;
    }
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = out[i];
}

// ---- phase B building blocks (registration.cu:194-257 + :41-118 fused) ----
struct IntTransform
{
  int32_t M[12];
  int32_t cx, cy, cz;
};

// The pose as make_int_transform needs it, next to the float pose in LDS: TI[4 j + i] = (int)(T[4 j + i] * 32768) for the rows
// i < 3, and the integer centre (int)T[12 + i] in the fourth-row places 3, 7, 11.  Written by the lane that has just computed
// the element (two instructions on the first wave) instead of 22 conversions in each of the eight waves of every iteration.
__device__ __forceinline__ void store_int_pose(int32_t *TI_sh, int lane /* < 16: element (lane & 3, lane >> 2) */, float v)
{
  const int i = lane & 3, j = lane >> 2;
  if (i < 3) TI_sh[lane] = (int32_t)(v * (float)MATRIX_RESOLUTION);
  if (j == 3 && i < 3) TI_sh[4 * i + 3] = (int32_t)v;
}
__device__ __forceinline__ IntTransform load_int_pose(const int32_t *TI_sh)
{
  IntTransform t;
  int32_t w[16];
#pragma unroll
  for (int q = 0; q < 4; ++q)
  {
    const int4 v = *reinterpret_cast<const int4 *>(TI_sh + 4 * q);
    w[4 * q + 0] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) t.M[j * 3 + i] = w[j * 4 + i];
  t.cx = w[3];
  t.cy = w[7];
  t.cz = w[11];
  return t;
}

// One Gauss-Newton update with the whole state in registers, identical in every lane of the wave
template <typename HF, typename GF>
__device__ __forceinline__ void gn_update(GnCore &st, HF H, GF G, int32_t e, int32_t c)
{
  float tr[16];
  if (!gn_increment(st, H, G, c, tr)) return;
  pose_product(st.T, tr);
  gn_convergence(st, e, c);
}

// the update fed from the 29 reduced terms in LDS (e and c are `int` in the reference, registration.cu:16-21)
__device__ __forceinline__ void gn_update_terms(GnCore &st, const int64_t *terms)
{
  gn_update(
      st, [terms](int r, int c) { return terms[r <= c ? tri_index(r, c) : tri_index(c, r)]; }, [terms](int r) { return terms[21 + r]; },
      (int32_t)terms[27], (int32_t)terms[28]);
}

// The update fed from REGISTERS (reg_loop_kernel): `total` is what the exchange left in lanes 0 .. 31 (the total of slot
// `lane`), `Tel` the pose element (lane & 3, (lane >> 2) & 3) in lanes 0 .. 15.  Lane 8 r + c fetches its element of [H | g] with
// one ds_bpermute pair and the pose's column comes over the quad with DPP: no LDS write -> read round trip between the
// exchange and the solve, none between the increment and the product.  Same operations per element as gn_update.
__device__ __forceinline__ void gn_update_total(GnCore &st, int64_t total, float &Tel, float *T_sh, int32_t *TI_sh)
{
  const int lane = threadIdx.x & 63, lr = lane >> 3, lc = lane & 7;
  int src = 29; // an empty slot
  if (lr < 6 && lc < 6) src = lr <= lc ? tri_index(lr, lc) : tri_index(lc, lr);
  if (lr < 6 && lc == 6) src = 21 + lr;
  const int lo = __builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)((uint64_t)total & 0xffffffffull));
  const int hi = __builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)((uint64_t)total >> 32));
  const int64_t mine = pack64(lo, hi);
  // (uniform values the vector unit computes with: kept out of the scalar registers, like the loop state)
  int32_t e = __builtin_amdgcn_ds_bpermute(27 << 2, (int)(uint32_t)((uint64_t)total & 0xffffffffull));
  int32_t c = __builtin_amdgcn_ds_bpermute(28 << 2, (int)(uint32_t)((uint64_t)total & 0xffffffffull));
  pin_vgpr(e);
  pin_vgpr(c);
  GnStep o;
  if (!gn_step(
          st, [mine](int, int) { return mine; }, [mine](int) { return mine; }, c, o))
    return;
  // T = tr * T: lane 4 j + i computes element (i, j) and needs ROW i of tr only -- built here per lane (the same operations
  // in the same order as gn_increment does for that row: 60 instructions instead of the 170 of all sixteen elements in every
  // lane plus twelve selects).  Row 3 of tr is (0, 0, 0, 1): its lanes select zeros and compute exactly that.
  // (The selects are v_cndmask on lane masks: as C selects over an array the compiler turned them into an INDEXED read, i.e. a
  // copy in scratch memory and a round trip to it in the middle of the chain.)
  const unsigned long long m1 = 0xaaaaaaaaaaaaaaaaull, m2 = 0xccccccccccccccccull; // lanes with bit 0 / bit 1 of the row set
  auto pick = [](float a, float b, unsigned long long mask) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
  };
  auto row4 = [&](float r0, float r1, float r2, float r3) { return pick(pick(r0, r1, m1), pick(r2, r3, m1), m2); };
  const float L[3][3] = {{0.f, o.L01, o.L02}, {o.L10, 0.f, o.L12}, {o.L20, o.L21, 0.f}};
  const float Li[3] = {row4(0.f, o.L10, o.L20, 0.f), row4(o.L01, 0.f, o.L21, 0.f), row4(o.L02, o.L12, 0.f, 0.f)}; // L[i][0..2]
  const float dl[3] = {row4(1.f, 0.f, 0.f, 0.f), row4(0.f, 1.f, 0.f, 0.f), row4(0.f, 0.f, 1.f, 0.f)};            // i == j
  float row[4];
#pragma unroll
  for (int j = 0; j < 3; ++j)
  {
    float ll = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) ll = __fadd_rn(ll, __fmul_rn(__fmul_rn(o.omc, Li[k]), L[k][j]));
    row[j] = __fadd_rn(__fadd_rn(dl[j], __fmul_rn(o.s, Li[j])), ll);
  }
  {
    const float oc0 = -(float)st.center[0], oc1 = -(float)st.center[1], oc2 = -(float)st.center[2];
    const float shift = __fadd_rn(__fadd_rn(__fmul_rn(row[0], oc0), __fmul_rn(row[1], oc1)), __fmul_rn(row[2], oc2));
    const float ci = row4((float)st.center[0], (float)st.center[1], (float)st.center[2], 0.f), ti = row4(o.t3, o.t4, o.t5, 0.f);
    row[3] = pick(__fadd_rn(__fadd_rn(shift, ci), ti), 1.f, m1 & m2); // tr[15] = 1
  }
  st.alpha = __fadd_rn(st.alpha, st.it_weight_gradient);
  const int tb = __float_as_int(Tel);
  const float tk[4] = {__int_as_float(__builtin_amdgcn_update_dpp(0, tb, 0x00, 0xf, 0xf, false)),  // quad_perm [0,0,0,0]: column j of the old
                       __int_as_float(__builtin_amdgcn_update_dpp(0, tb, 0x55, 0xf, 0xf, false)),  // [1,1,1,1]     pose sits in the lane's quad
                       __int_as_float(__builtin_amdgcn_update_dpp(0, tb, 0xaa, 0xf, 0xf, false)),  // [2,2,2,2]
                       __int_as_float(__builtin_amdgcn_update_dpp(0, tb, 0xff, 0xf, 0xf, false))}; // [3,3,3,3]
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) acc = __fadd_rn(acc, __fmul_rn(row[k], tk[k]));
  Tel = acc;
  if (lane < 16)
  {
    T_sh[lane] = acc;
    store_int_pose(TI_sh, lane, acc);
  }
  gn_convergence(st, e, c);
}

struct PointArgs
{
  const int32_t *points;
  uint32_t first;
  uint32_t end; // exclusive
  const uint32_t *map_data;
  MapParams map;
  FastDiv resdiv;
};

// ---- phase B building blocks (registration.cu:194-257 + :41-118 fused) ----

// cu_to_int_mat (cuda/util.h:24-35): (int)(float * 32768); registration.cu:208: center = (int) translation of the CURRENT transform
__device__ __forceinline__ IntTransform make_int_transform(const float *T)
{
  IntTransform t;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) t.M[j * 3 + i] = (int32_t)(T[j * 4 + i] * (float)MATRIX_RESOLUTION);
  t.cx = (int32_t)T[12];
  t.cy = (int32_t)T[13];
  t.cz = (int32_t)T[14];
  return t;
}

struct Gathered
{
  int32_t qx, qy, qz; // transformed point minus center
  uint32_t cur, xn, xl, yn, yl, zn, zl;
  bool ok;
};

// The voxel a point fell into at the previous iteration of the resident loop and the 7 entries read there.  Late in
// the Gauss-Newton loop the pose moves by a fraction of a millimetre per iteration, almost every point stays in its
// voxel, and a wave whose 128 points all stayed issues no load at all (the map does not change during the loop).
struct VoxelCache
{
  int32_t bx, by, bz;
  uint32_t cur, xn, xl, yn, yl, zn, zl;
  bool filled;
};

// transform one point and issue its 7 gathers (nothing here waits for memory)
template <bool CACHED = false>
__device__ __forceinline__ Gathered gather_point(const PointArgs &a, const IntTransform &t, int32_t px, int32_t py, int32_t pz, bool valid,
                                                 VoxelCache *cache = nullptr)
{
  Gathered g;
  // cu_transform_point (cuda/util.h:11-22), int32 wrap like the reference
  int32_t qx = wadd(wadd(wadd(wmul(t.M[0], px), wmul(t.M[3], py)), wmul(t.M[6], pz)), t.M[9]) / MATRIX_RESOLUTION;
  int32_t qy = wadd(wadd(wadd(wmul(t.M[1], px), wmul(t.M[4], py)), wmul(t.M[7], pz)), t.M[10]) / MATRIX_RESOLUTION;
  int32_t qz = wadd(wadd(wadd(wmul(t.M[2], px), wmul(t.M[5], py)), wmul(t.M[8], pz)), t.M[11]) / MATRIX_RESOLUTION;
  const int32_t bx = div_trunc(qx, a.resdiv), by = div_trunc(qy, a.resdiv), bz = div_trunc(qz, a.resdiv);
  g.qx = wsub(qx, t.cx);
  g.qy = wsub(qy, t.cy);
  g.qz = wsub(qz, t.cz);
  g.ok = valid && in_bounds_buffer(a.map, bx, by, bz, -1); // in_bounds_with_buffer_neg(buf, 1), registration.cu:217
  g.cur = g.xn = g.xl = g.yn = g.yl = g.zn = g.zl = 0;
  if (CACHED)
  {
    VoxelCache &c = *cache;
    const bool refill = g.ok && !(c.filled && c.bx == bx && c.by == by && c.bz == bz);
    if (refill) // one exec-mask region; everything else is selects
    {
      c.cur = a.map_data[get_index(a.map, bx, by, bz)];
      c.xn = a.map_data[get_index(a.map, bx + 1, by, bz)];
      c.xl = a.map_data[get_index(a.map, bx - 1, by, bz)];
      c.yn = a.map_data[get_index(a.map, bx, by + 1, bz)];
      c.yl = a.map_data[get_index(a.map, bx, by - 1, bz)];
      c.zn = a.map_data[get_index(a.map, bx, by, bz + 1)];
      c.zl = a.map_data[get_index(a.map, bx, by, bz - 1)];
      c.bx = bx;
      c.by = by;
      c.bz = bz;
      c.filled = true;
    }
    const uint32_t keep = g.ok ? 0xffffffffu : 0u;
    g.cur = c.cur & keep; g.xn = c.xn & keep; g.xl = c.xl & keep; g.yn = c.yn & keep; g.yl = c.yl & keep; g.zn = c.zn & keep; g.zl = c.zl & keep;
    return g;
  }
  if (g.ok)
  {
    // the 6 neighbours are in bounds by the test above
    g.cur = a.map_data[get_index(a.map, bx, by, bz)];
    g.xn = a.map_data[get_index(a.map, bx + 1, by, bz)];
    g.xl = a.map_data[get_index(a.map, bx - 1, by, bz)];
    g.yn = a.map_data[get_index(a.map, bx, by + 1, bz)];
    g.yl = a.map_data[get_index(a.map, bx, by - 1, bz)];
    g.zn = a.map_data[get_index(a.map, bx, by, bz + 1)];
    g.zl = a.map_data[get_index(a.map, bx, by, bz - 1)];
  }
  return g;
}

// The map's constants as the resident loop holds them: uniform values, but in VECTOR registers.  As kernel arguments they
// live in scalar registers, and the loop has more uniform state than scalar registers: the compiler spilled them to vector
// lanes and fetched them back (v_readlane + s_nop) in every iteration, and since a vector instruction takes at most one scalar
// operand it copied a further 28 of them into vector registers per point anyway.
struct LoopGather
{
  int32_t ringK[3]; // offset + size - pos: ring coordinate = ring(x + ringK, size)
  int32_t size[3];
  int32_t pos[3];
  uint32_t lim[3];  // size / 2 - 1: in_bounds_with_buffer_neg(buf, 1)
  uint32_t divM;    // division by the map resolution (FastDiv)
  int32_t divK;
};
__device__ __forceinline__ LoopGather make_loop_gather(const PointArgs &a)
{
  LoopGather c;
#pragma unroll
  for (int k = 0; k < 3; ++k)
  {
    c.ringK[k] = wsub(wadd(a.map.offset[k], a.map.size[k]), a.map.pos[k]);
    c.size[k] = a.map.size[k];
    c.pos[k] = a.map.pos[k];
    c.lim[k] = (uint32_t)(a.map.size[k] / 2 - 1); // size >= 3 (ws_map_create)
    pin_vgpr(c.ringK[k]); pin_vgpr(c.size[k]); pin_vgpr(c.pos[k]); pin_vgpr(c.lim[k]);
  }
  c.divM = (uint32_t)a.resdiv.M; // < 2^32 (make_fastdiv)
  c.divK = a.resdiv.k;
  pin_vgpr(c.divM); pin_vgpr(c.divK);
  return c;
}
__device__ __forceinline__ int64_t loop_index(const LoopGather &c, int32_t x, int32_t y, int32_t z)
{
  // get_index (ws_device.h) with x - pos + offset + size folded into one constant per axis (the same bits: wrapping adds)
  const int32_t xi = ring(wadd(x, c.ringK[0]), c.size[0]), yi = ring(wadd(y, c.ringK[1]), c.size[1]), zi = ring(wadd(z, c.ringK[2]), c.size[2]);
  const int32_t row = xi * c.size[1] + yi;
  return (int64_t)row * (int64_t)c.size[2] + zi;
}
// gather_point<true> on those constants (same arithmetic, same results)
__device__ __forceinline__ Gathered gather_point_loop(const PointArgs &a, const LoopGather &c, const IntTransform &t, int32_t px, int32_t py, int32_t pz, bool valid,
                                                      VoxelCache &vc)
{
  Gathered g;
  int32_t qx = wadd(wadd(wadd(wmul(t.M[0], px), wmul(t.M[3], py)), wmul(t.M[6], pz)), t.M[9]) / MATRIX_RESOLUTION;
  int32_t qy = wadd(wadd(wadd(wmul(t.M[1], px), wmul(t.M[4], py)), wmul(t.M[7], pz)), t.M[10]) / MATRIX_RESOLUTION;
  int32_t qz = wadd(wadd(wadd(wmul(t.M[2], px), wmul(t.M[5], py)), wmul(t.M[8], pz)), t.M[11]) / MATRIX_RESOLUTION;
  const int32_t bx = div_trunc(qx, (uint64_t)c.divM, c.divK, 0), by = div_trunc(qy, (uint64_t)c.divM, c.divK, 0), bz = div_trunc(qz, (uint64_t)c.divM, c.divK, 0);
  g.qx = wsub(qx, t.cx);
  g.qy = wsub(qy, t.cy);
  g.qz = wsub(qz, t.cz);
  g.ok = valid && (uint32_t)iabs32(wsub(bx, c.pos[0])) <= c.lim[0] && (uint32_t)iabs32(wsub(by, c.pos[1])) <= c.lim[1] &&
         (uint32_t)iabs32(wsub(bz, c.pos[2])) <= c.lim[2];
  const bool refill = g.ok && !(vc.filled && vc.bx == bx && vc.by == by && vc.bz == bz);
  if (refill)
  {
    // Ring coordinates of the voxel and of its neighbours.  The voxel is in bounds with a margin of one, so x + ringK lies in
    // [0, 3 size): ring() as two conditional subtractions written as unsigned minima (v - size wraps to a huge number when
    // v < size), and a neighbour is the voxel's own ring coordinate +- 1 with one wrap -- 30 instructions instead of the
    // nine full ring() of seven loop_index calls (54); the same indices.
    uint32_t rc[3], rn[3], rl[3];
    const int32_t b3[3] = {bx, by, bz};
#pragma unroll
    for (int k = 0; k < 3; ++k)
    {
      const uint32_t sz = (uint32_t)c.size[k];
      uint32_t v = (uint32_t)wadd(b3[k], c.ringK[k]);
      v = min(v, v - sz);
      v = min(v, v - sz);
      rc[k] = v;
      rn[k] = min(v + 1u, v + 1u - sz);      // v + 1 == size -> 0
      rl[k] = min(v - 1u, v - 1u + sz);      // v == 0 -> size - 1 (v - 1 wraps)
    }
    const uint32_t sy = (uint32_t)c.size[1], sz = (uint32_t)c.size[2];
    // (row * size_z + z as ONE v_mad_u64_u32: all three operands unsigned 32-bit, the index 64-bit for 2049^3)
    auto at = [&](uint32_t row, uint32_t z) { return a.map_data[(uint64_t)row * (uint64_t)sz + (uint64_t)z]; };
    const uint32_t row_c = rc[0] * sy + rc[1];
    vc.cur = at(row_c, rc[2]);
    vc.xn = at(rn[0] * sy + rc[1], rc[2]);
    vc.xl = at(rl[0] * sy + rc[1], rc[2]);
    vc.yn = at(rc[0] * sy + rn[1], rc[2]);
    vc.yl = at(rc[0] * sy + rl[1], rc[2]);
    vc.zn = at(row_c, rn[2]);
    vc.zl = at(row_c, rl[2]);
    vc.bx = bx;
    vc.by = by;
    vc.bz = bz;
    vc.filled = true;
  }
  const uint32_t keep = g.ok ? 0xffffffffu : 0u;
  g.cur = vc.cur & keep; g.xn = vc.xn & keep; g.xl = vc.xl & keep; g.yn = vc.yn & keep; g.yl = vc.yl & keep; g.zn = vc.zn & keep; g.zl = vc.zl & keep;
  return g;
}

// Both functions below are written without branches on purpose: nested `if`s over three gradients became nine exec-mask
// regions with a round trip through the scalar unit each (v_cmp -> SGPR -> s_and_saveexec -> s_cbranch), which cost more
// than the arithmetic they skipped; masks keep the whole point in the vector unit.
__device__ __forceinline__ int32_t central_gradient(uint32_t next, uint32_t last)
{
  // registration.cu:233-246: both neighbours observed and not of strictly opposite sign
  const int32_t nv = entry_value(next), lv = entry_value(last);
  const int32_t observed = ((next >> 16) != 0u) & ((last >> 16) != 0u);
  const int32_t opposite = (nv * lv) < 0; // 16-bit values: the product is negative iff the signs are strictly opposite
  const int32_t keep = -(observed & (opposite ^ 1));
  return ((nv - lv) / 2) & keep;
}

// J (registration.cu:224-250), the voxel's value and whether the point counts, all zero for a point that does not
__device__ __forceinline__ void point_terms(const Gathered &g, int32_t (&J)[6], int32_t &v, int32_t &used)
{
  // a point outside the map or in an unobserved voxel (registration.cu:217-222) contributes zeros
  used = (g.ok ? 1 : 0) & ((g.cur >> 16) != 0u);
  const int32_t keep = -used;
  const int32_t gx = central_gradient(g.xn, g.xl) & keep, gy = central_gradient(g.yn, g.yl) & keep, gz = central_gradient(g.zn, g.zl) & keep;
  // point.cross(gradient) in int (math/vector3.h:269-277); J = (cross, gradient) as long
  J[0] = wsub(wmul(g.qy, gz), wmul(g.qz, gy));
  J[1] = wsub(wmul(g.qz, gx), wmul(g.qx, gz));
  J[2] = wsub(wmul(g.qx, gy), wmul(g.qy, gx));
  J[3] = gx;
  J[4] = gy;
  J[5] = gz;
  v = entry_value(g.cur) & keep;
}

__device__ __forceinline__ void consume_point(const Gathered &g, int64_t (&acc)[REG_SLOTS])
{
  int32_t J[6], v, used;
  point_terms(g, J, v, used);
  // 21 unique terms of J J^T (registration.cu:55-97); int32 x int32 + int64 maps onto v_mad_i64_i32
#pragma unroll
  for (int i = 0; i < 6; ++i)
#pragma unroll
    for (int j = i; j < 6; ++j) acc[tri_index(i, j)] = wadd64(acc[tri_index(i, j)], (int64_t)J[i] * (int64_t)J[j]);
#pragma unroll
  for (int i = 0; i < 6; ++i) acc[21 + i] = wadd64(acc[21 + i], (int64_t)J[i] * (int64_t)v);
  acc[27] += (v < 0 ? -v : v);
  acc[28] += used;
}

// ---- the same sums on the matrix cores (resident loop, clouds of at most one point per lane) ---------------------------
// h = sum J J^T, g = sum J v, e = sum |v|, c = sum 1 are one Gram matrix A A^T over the points, and v_mfma_i32_32x32x32_i8
// computes exactly that -- in int32, exactly -- for rows of signed bytes.  A point's 32 rows are the bytes of 8 dwords:
//   d0..d5  J[0..5] ^ 0x00808080   limbs s0 s1 s2 s3 with J = s0 + 256 s1 + 65536 s2 + 2^24 s3 + 0x808080  (s3: the sign byte;
//           the three low bytes become SIGNED limbs by flipping their top bit, i.e. by carrying a bias of 128 each)
//   d6      (v ^ 0x80) | (n ^ 0x80) << 16   with n = -|v|: two limbs each (v, n in [-32768, 32767]), bias 128
//   d7      1 | used << 8                   a row of ones (what multiplies the biases) and the row that counts
// One instruction multiplies the 32 x 32 rows of 32 points; the SAME register is its A and its B operand (B[k][j] = A[j][k]),
// so whatever order the hardware gives the 32 points inside the operand, a row meets itself in the same order.  Lane (r, half)
// must supply row r of 16 points, while a point's rows are computed in ONE lane: the wave's 64 x 32 bytes pass through LDS,
// point-major as they are computed (two 128-bit writes per lane, points 48 bytes apart), and come back TRANSPOSED by the
// hardware: ds_read_b64_tr_b8 hands every lane of a 16-lane group one byte column of an 8-point x 16-byte block (tools/
// tr_probe.hip prints what it does), i.e. row r of 8 points per read -- four reads per wave, no byte shuffling in registers
// (the first version read [dword][point] with eight 128-bit reads and picked bytes with 24 v_perm_b32: +0.05 us).  Two
// matrix instructions per wave replace 27 v_mad_i64_i32 per lane AND the 190-instruction transposing butterfly: the K
// dimension of the product is the reduction over the lanes.  What comes out, per wave: C[a][b] = sum over its 64 points of limb a x limb b.  Lane (b, half) holds rows
// 8g + 4 half + t: the four limbs t of dword 2g + half, i.e. (Horner) the sum over the points of Js_i x (limb b % 4 of dword
// b / 4); shifted by 8 (b % 4) it goes straight into the workgroup's slot with an LDS atomic -- the 4 limb columns of a dword
// meet there.  The biases: sum (Js_i + B)(Js_j + B) = sum Js_i Js_j + B (T_i + T_j) + B^2 N with T_i = sum Js_i x 1 (the ones
// column) and N = sum 1 x 1, all from the same product; the first wave adds those terms once per iteration when it reads the
// slots (mfma_finalize).  Everything is integer arithmetic mod 2^64 like the int64 sums it replaces: bit-identical.
typedef int mf_v4i __attribute__((ext_vector_type(4)));
typedef int mf_v16i __attribute__((ext_vector_type(16)));
#ifndef WS_MF_POINT_STRIDE
#define WS_MF_POINT_STRIDE 12
#endif
constexpr int MF_POINT_STRIDE = WS_MF_POINT_STRIDE;  // words per staged point: 8 used, 48 bytes apart (128-bit writes without bank conflicts)
constexpr int MF_STAGE_WORDS = 64 * MF_POINT_STRIDE; // per wave
constexpr int MF_AUX = 8;                          // behind the 32 slots: T_0..T_5, sum vs, N
constexpr uint32_t MF_NONE = 0xffffffffu;
constexpr uint64_t MF_BJ = 0x00808080ull, MF_BV = 0x80ull;

struct MfLane // constants of a lane
{
  uint32_t rd;      // first staged word this lane reads
  uint32_t shift;   // 8 x (column limb)
  uint32_t slot[3]; // where the lane's values of g = 0, 1, 2 go (index into wg_sum[32 + MF_AUX]), MF_NONE: nowhere
  // first wave, lane -> slot (lane & 31): raw + ca * aux[ia] + cb * aux[ib] + cn * N
  uint32_t ia, ib;
  uint64_t ca, cb, cn;
};
__device__ __forceinline__ MfLane make_mf_lane()
{
  MfLane L;
  const int lane = threadIdx.x & 63, j = lane & 31, half = lane >> 5, jd = j >> 2, m = j & 3;
  // ds_read_b64_tr_b8: a 16-lane group reads a block of 8 points x 16 row bytes, every lane 8 contiguous bytes at its OWN
  // address (lane j of the group: point j / 2, bytes 8 (j % 2) .. of the 16-byte window), and gets back the block's column j:
  // the row byte (lane & 15) of the window for the 8 points.  Groups 0 / 1 take the windows of rows 0..15 / 16..31, the upper
  // half of the wave the points 16 further on.
  L.rd = (uint32_t)((16 * half + ((lane & 15) >> 1)) * MF_POINT_STRIDE * 4 + ((lane >> 4) & 1) * 16 + (lane & 1) * 8); // bytes
  L.shift = (uint32_t)(8 * m);
#pragma unroll
  for (int g = 0; g < 3; ++g)
  {
    const int i = 2 * g + half;
    uint32_t t = MF_NONE;
    if (jd < 6 && i <= jd)
      t = (uint32_t)tri_index(i, jd);
    else if (jd == 6 && m < 2)
      t = (uint32_t)(21 + i); // the value's two limbs as columns: g[i]
    else if (j == 28)
      t = (uint32_t)(REG_SLOTS + i); // the ones column: T_i
    L.slot[g] = t;
  }
  const int slot = lane & 31;
  L.ia = L.ib = 0;
  L.ca = L.cb = L.cn = 0;
  if (slot < 21)
  {
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int jj = i; jj < 6; ++jj)
        if (tri_index(i, jj) == slot)
        {
          L.ia = (uint32_t)i;
          L.ib = (uint32_t)jj;
        }
    L.ca = L.cb = MF_BJ;
    L.cn = MF_BJ * MF_BJ;
  }
  else if (slot < 27)
  {
    L.ia = (uint32_t)(slot - 21);
    L.ib = 6; // sum vs
    L.ca = MF_BV;
    L.cb = MF_BJ;
    L.cn = MF_BJ * MF_BV;
  }
  else if (slot == 27)
    L.cn = MF_BV; // e = -(sum ns + 128 N)
  return L;
}

// one point per lane (all 64 lanes active; a lane without a point has g.ok == false): C += A A^T of the wave's 64 points
__device__ __forceinline__ void mfma_consume(const Gathered &g, mf_v16i &C, uint32_t *stage /* this wave's MF_STAGE_WORDS */, const MfLane &L)
{
  int32_t J[6], v, used;
  point_terms(g, J, v, used);
  const int lane = threadIdx.x & 63;
  const int32_t n = v < 0 ? v : -v;
  uint4 d0, d1;
  d0.x = (uint32_t)J[0] ^ (uint32_t)MF_BJ; d0.y = (uint32_t)J[1] ^ (uint32_t)MF_BJ; d0.z = (uint32_t)J[2] ^ (uint32_t)MF_BJ; d0.w = (uint32_t)J[3] ^ (uint32_t)MF_BJ;
  d1.x = (uint32_t)J[4] ^ (uint32_t)MF_BJ; d1.y = (uint32_t)J[5] ^ (uint32_t)MF_BJ;
  d1.z = (((uint32_t)v ^ (uint32_t)MF_BV) & 0xffffu) | (((uint32_t)n ^ (uint32_t)MF_BV) << 16);
  d1.w = 1u | ((uint32_t)used << 8);
  *reinterpret_cast<uint4 *>(&stage[lane * MF_POINT_STRIDE]) = d0;
  *reinterpret_cast<uint4 *>(&stage[lane * MF_POINT_STRIDE + 4]) = d1;
  __builtin_amdgcn_wave_barrier(); // (LDS serves a wave's accesses in order: the reads below see all 64 lanes' writes)
  typedef int mf_v2i __attribute__((ext_vector_type(2)));
  const char *base = reinterpret_cast<const char *>(stage) + L.rd;
#pragma unroll
  for (int q = 0; q < 2; ++q)
  {
    // the transposing read delivers the operand as the instruction wants it: row (lane & 31) of 8 points per read
    const mf_v2i lo = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) mf_v2i *)(base + (32 * q) * MF_POINT_STRIDE * 4));
    const mf_v2i hi = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) mf_v2i *)(base + (32 * q + 8) * MF_POINT_STRIDE * 4));
    mf_v4i a;
    a[0] = lo[0]; a[1] = lo[1]; a[2] = hi[0]; a[3] = hi[1];
    C = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, a, C, 0, 0, 0);
  }
  __builtin_amdgcn_wave_barrier(); // the next call's writes stay behind these reads
}

// the wave's product into the workgroup's slots (wg_sum[32 + MF_AUX], zero before the iteration)
__device__ __forceinline__ void mfma_flush(const mf_v16i &C, unsigned long long *wg_sum, const MfLane &L)
{
  const int lane = threadIdx.x & 63;
  int32_t lo[4], hi[4]; // |C| <= 2^14 x 64 points: the pairs fit 32 bits
#pragma unroll
  for (int g = 0; g < 4; ++g)
  {
    lo[g] = C[4 * g + 1] * 256 + C[4 * g + 0];
    hi[g] = C[4 * g + 3] * 256 + C[4 * g + 2];
  }
  // The four limb columns of a dword are four adjacent lanes and meet in the slot itself: 48 lanes on 12 addresses per
  // instruction.  (Summing them over the quad first -- two DPP stages, one lane adds -- is slower: 5.49 vs 5.38 us per iteration.)
#pragma unroll
  for (int g = 0; g < 3; ++g)
  {
    const int64_t P = (int64_t)hi[g] * 65536 + (int64_t)lo[g];
    if (L.slot[g] != MF_NONE) atomicAdd(&wg_sum[L.slot[g]], (unsigned long long)P << L.shift);
  }
  // rows 24..27 (lower half): the limbs of v and n; rows 28, 29 (upper half): ones and used -- against the ones column (28)
  if (lane == 28 || lane == 60)
  {
    const bool up = lane == 60;
    atomicAdd(&wg_sum[up ? REG_SLOTS + 7 : REG_SLOTS + 6], (unsigned long long)(int64_t)(up ? C[12] : lo[3])); // N : sum vs
    atomicAdd(&wg_sum[up ? 28 : 27], (unsigned long long)(int64_t)(up ? C[13] : hi[3]));                       // c : sum ns
  }
}

// first wave, after the barrier: the value of slot (lane & 31) with the bias terms added; leaves the slots zero
__device__ __forceinline__ unsigned long long mfma_finalize(unsigned long long *wg_sum, const MfLane &L)
{
  const int lane = threadIdx.x & 63, slot = lane & 31;
  const unsigned long long raw = wg_sum[slot], Ta = wg_sum[REG_SLOTS + L.ia], Tb = wg_sum[REG_SLOTS + L.ib], N = wg_sum[REG_SLOTS + 7];
  __builtin_amdgcn_wave_barrier();
  if (lane < REG_SLOTS) wg_sum[slot] = 0;
  if (lane < MF_AUX) wg_sum[REG_SLOTS + lane] = 0;
  unsigned long long s = raw + L.ca * Ta + L.cb * Tb + L.cn * N;
  if (slot == 27) s = 0ull - s;
  return s;
}

constexpr uint32_t REG_STRIDE = REG_BLOCKS * REG_THREADS; // points covered by one pass of the grid

// Which point a thread takes in pass u of the grid: the grid's WAVES in the order (wave-in-workgroup, workgroup), 64 consecutive
// points each.  A cloud (or a rank's shard) smaller than one pass then fills wave 0 of every workgroup before wave 1 of any:
// the points are spread over all compute units, and a workgroup's unused waves skip the accumulate and reduce phases, so the
// phase costs a 16 384-point shard (1 wave per workgroup) a third of what it costs the full cloud (8 waves, two per SIMD).
// With workgroup-major order the same shard filled 32 workgroups to the brim and left 224 idle: no gain from sharding at all.
__device__ __forceinline__ uint32_t point_slot() { return (((threadIdx.x >> 6) * gridDim.x + blockIdx.x) << 6) + (threadIdx.x & 63u); }

// raw coordinates of this lane's first two points, loaded before anything else in the kernel
struct Prefetched
{
  int32_t p[2][3];
  bool valid[2];
};
__device__ __forceinline__ Prefetched prefetch_points(const PointArgs &a, const uint32_t REG_STRIDE = ws::REG_STRIDE)
{
  Prefetched f;
#pragma unroll
  for (int u = 0; u < 2; ++u)
  {
    const uint32_t idx = a.first + point_slot() + (uint32_t)u * REG_STRIDE;
    f.valid[u] = idx < a.end;
    const size_t o = f.valid[u] ? 3 * (size_t)idx : 0;
    f.p[u][0] = f.valid[u] ? a.points[o + 0] : 0;
    f.p[u][1] = f.valid[u] ? a.points[o + 1] : 0;
    f.p[u][2] = f.valid[u] ? a.points[o + 2] : 0;
  }
  return f;
}

template <bool CACHED = false>
__device__ __forceinline__ void accumulate_points(const PointArgs &a, const float *T, const Prefetched &f, int64_t (&acc)[REG_SLOTS],
                                                  VoxelCache *cache = nullptr, const uint32_t REG_STRIDE = ws::REG_STRIDE)
{
  const IntTransform t = make_int_transform(T);
  // the two prefetched points: 14 gathers in flight before the first is consumed
  const Gathered g0 = gather_point<CACHED>(a, t, f.p[0][0], f.p[0][1], f.p[0][2], f.valid[0], CACHED ? &cache[0] : nullptr);
  if (__ballot(f.valid[1]) != 0ull) // a whole wave without a second point (cloud <= one pass of the grid) skips its arithmetic
  {
    const Gathered g1 = gather_point<CACHED>(a, t, f.p[1][0], f.p[1][1], f.p[1][2], f.valid[1], CACHED ? &cache[1] : nullptr);
    consume_point(g0, acc);
    consume_point(g1, acc);
  }
  else
    consume_point(g0, acc);
  // clouds larger than two passes of the grid (N > 131 072)
  for (uint32_t idx = a.first + point_slot() + 2 * REG_STRIDE; idx < a.end; idx += REG_STRIDE)
  {
    const Gathered g = gather_point(a, t, a.points[3 * (size_t)idx + 0], a.points[3 * (size_t)idx + 1], a.points[3 * (size_t)idx + 2], true);
    consume_point(g, acc);
  }
}

// Sum of the partials [REG_BLOCKS][REG_SLOTS] a previous launch left in HBM -> red[0..31] in LDS.
// Lane l of wave w adds slot (l >> 1) over 32 of the wave's 64 workgroups: 32 independent, fully coalesced
// loads per lane (one memory latency), one shuffle, one LDS hop.
// COHERENT: the partials were written by other workgroups of the SAME launch -> agent-scope loads (sc1), which
// cannot be served from a stale line of this XCD's L2.
template <bool COHERENT = false>
__device__ __forceinline__ void sum_partials(const int64_t *pp, int64_t (*wave_part)[REG_SLOTS], int64_t *red)
{
  constexpr int WAVES = REG_THREADS / 64;
  constexpr int PER_LANE = REG_BLOCKS / (WAVES * 2); // workgroups summed by one lane
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slot = lane >> 1;
  const int64_t *base = pp + ((size_t)wave * (2 * PER_LANE) + (size_t)(lane & 1) * PER_LANE) * REG_SLOTS + slot;
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < PER_LANE; ++i)
  {
    const int64_t v = COHERENT ? __hip_atomic_load(const_cast<int64_t *>(&base[(size_t)i * REG_SLOTS]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : base[(size_t)i * REG_SLOTS];
    s = wadd64(s, v);
  }
  s = wadd64(s, shfl_xor_i64(s, 1));
  if ((lane & 1) == 0) wave_part[wave][slot] = s;
  __syncthreads();
  if (threadIdx.x < REG_SLOTS)
  {
    int64_t t = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) t = wadd64(t, wave_part[w][threadIdx.x]);
    red[threadIdx.x] = t;
  }
  __syncthreads();
}

struct IterArgs
{
  PointArgs pts;
  GnState *state;     // [2], double buffered by launch parity
  int64_t *partials;  // [2][REG_BLOCKS][REG_SLOTS]
  int32_t k;          // launch index 0 .. max_iterations
  int32_t *host_flag; // host-mapped: set when the loop has finished (lets the host stop enqueueing)
};

// One launch == one Gauss-Newton iteration (see the header of this file).
__global__ __launch_bounds__(REG_THREADS) void reg_iter_kernel(IterArgs a)
{
  __shared__ int64_t wave_part[REG_THREADS / 64][REG_SLOTS];
  __shared__ int64_t red[REG_SLOTS];
  __shared__ float T_sh[16];
  __shared__ int stop_sh;

#ifdef WS_REG_TIMING
  long long ts[6];
  ts[0] = wall_clock64();
#define WS_STAMP(i) ts[i] = wall_clock64()
#else
#define WS_STAMP(i)
#endif
  const GnState *prev = &a.state[(a.k + 1) & 1];
  GnState *cur = &a.state[a.k & 1];
  const bool need_update = a.k > 0 && !prev->core.finished && prev->core.iterations < prev->core.max_iterations;
  // this lane's points do not depend on the transform: fetch them now, under phase A
  const Prefetched pref = prefetch_points(a.pts);

  if (need_update)
  {
    // phase A: total of the previous launch's partials (every workgroup, redundantly)
    WS_STAMP(1);
    sum_partials(a.partials + (size_t)((a.k + 1) & 1) * REG_SLOTS * REG_BLOCKS, wave_part, red);
  }
  WS_STAMP(2);
  if (threadIdx.x < 64)
  {
    // the whole first wave: the 6x6 solve is lane-parallel, everything else is computed identically by every lane
    GnCore st = prev->core;
    if (need_update)
    {
      gn_update_terms(st, red);
      if (blockIdx.x == 0 && threadIdx.x == 0)
      {
        int64_t sums[44];
        expand_sums(red, sums);
#pragma unroll
        for (int i = 0; i < 44; ++i) cur->sums[i] = sums[i];
      }
    }
    if (threadIdx.x == 0)
    {
      const int stop = (st.finished || st.iterations >= st.max_iterations) ? 1 : 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) T_sh[i] = st.T[i];
      stop_sh = stop;
      if (blockIdx.x == 0)
      {
        cur->core = st;
        if (stop && a.host_flag) __hip_atomic_store(a.host_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  }
  __syncthreads();
  WS_STAMP(3);
  if (stop_sh) return;

  // phase B
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = T_sh[i];
  int64_t acc[REG_SLOTS];
#pragma unroll
  for (int t = 0; t < REG_SLOTS; ++t) acc[t] = 0;
  accumulate_points(a.pts, T, pref, acc);
  WS_STAMP(4);
  block_reduce32(acc, wave_part, red);
  if (threadIdx.x < REG_SLOTS)
    a.partials[(size_t)(a.k & 1) * REG_SLOTS * REG_BLOCKS + (size_t)blockIdx.x * REG_SLOTS + threadIdx.x] = red[threadIdx.x];
#ifdef WS_REG_TIMING
  WS_STAMP(5);
  if (blockIdx.x == 7 && threadIdx.x == 0 && a.k == 20)
    printf("reg_iter k=%d ticks(100MHz): load %lld reduce %lld solve %lld accumulate %lld reduce %lld\n", a.k, ts[1] - ts[0], ts[2] - ts[1],
           ts[3] - ts[2], ts[4] - ts[3], ts[5] - ts[4]);
#endif
}

// ---- the whole Gauss-Newton loop in ONE launch -------------------------------------------------------
// The launch boundary between two iterations above costs ~5.5 us (dispatch of 256 workgroups, end-of-kernel
// cache write-back, the gap to the next launch) for ~10 us of work.  reg_loop_kernel keeps the 256
// workgroups resident (one per CU, checked on the host before the launch) and replaces the boundary by a
// grid-wide exchange of the partial sums that is its own barrier (below).  The
// per-iteration structure (and every arithmetic step) is the one of reg_iter_kernel; the points of a
// lane stay in registers for the whole loop.
// Exchange of the workgroups' partial sums inside the resident loop, WITHOUT a separate barrier.  Every workgroup ADDS
// its 32 values into one of REG_GROUPS accumulators (agent-scope atomic add, no return); a 64-bit value travels as two
// words -- its low and its high 32 bits -- whose top byte counts the additions: word += (1 << 56) | half.  The
// accumulators are never reset: a reader remembers the word it completed two iterations ago (same parity buffer), so
// (now - then) >> 56 is the number of workgroups that have added since, and the low 56 bits are the exact sum of their
// halves (32 workgroups x 2^32 never reaches bit 56; the differences are taken modulo 2^64, so wrapping is harmless).
// A reader therefore polls the DATA until every word's count is complete: no wait for the adds' acknowledgement, no
// arrival counter, no second read.  tools/barrier_bench.hip (256 workgroups, no work in between): counter + group sums
// 3.1 us per exchange, counted words polled by one wave 2.3 us -- and polling by all waves, more groups or 128-bit
// loads are all slower: the polling reads queue in front of the adds in the same memory channels.
// In the loop itself the polling matters even more than in the microbenchmark: a workgroup that starts to poll right
// after its own adds keeps 256 x 4 KB of coherent reads per round in flight while the adds of the others are still on
// their way, and the exchange takes 3.2 us; sleeping ~0.9 us (the time the adds need anyway) before the FIRST poll makes
// it 1.4 us, because that poll then usually succeeds (measured with -DWS_REG_TIMING, sleeps of 12 / 20 / 26 / 34 / 40 / 50
// x 64 clocks: 2.09 / 1.48 / 1.47 / 1.52 / 1.62 / 1.88 us).  Before: counter barrier + group sums 2.95 us.
// Safe against overtaking: a workgroup can only complete the poll of iteration i + 1 after every workgroup has added
// for i + 1, i.e. after every workgroup has finished reading iteration i, so nobody adds into a parity buffer (i + 2)
// that is still being read.
#ifndef WS_REG_GROUPS
#define WS_REG_GROUPS 8 // (round 5, in the loop itself, first poll after 16 / 26 / 36 x 64 clocks: 4 groups 5.71 / 5.20 / 4.91 us per iteration, 8 groups - / 4.39 / -, 16 groups 6.82 / 5.97 / -)
#endif
constexpr int REG_GROUPS = WS_REG_GROUPS;
constexpr int REG_WORDS = 2 * REG_SLOTS; // low halves, then high halves
constexpr uint64_t REG_COUNT_ONE = 1ull << 56;
constexpr uint64_t REG_SUM_MASK = REG_COUNT_ONE - 1;
static_assert(REG_WORDS == 64 && REG_BLOCKS % REG_GROUPS == 0 && REG_BLOCKS / REG_GROUPS < 256, "counted exchange");
#ifndef WS_REG_FIRST_POLL_SLEEP
#define WS_REG_FIRST_POLL_SLEEP 26 // (round 5, after the shorter solve: 16 / 22 / 25 / 26 / 28 / 30 / 34: 4.97 / 4.47 / 4.40 / 4.40 / 4.42 / 4.46 / 4.58 us per iteration)
#endif
constexpr int REG_FIRST_POLL_SLEEP = WS_REG_FIRST_POLL_SLEEP; // x 64 clocks before the first poll
constexpr int REG_POLL_SLEEP = 2;        // between polls
#ifndef WS_REG_PEER_POLL_SLEEP
#define WS_REG_PEER_POLL_SLEEP 8 // (two ranks on one GPU: 4 -> 6.7, 12 -> 6.9, 20 -> 7.1, 28 -> 7.3 us per iteration; the mailbox is local memory, its polls are cheap)
#endif
constexpr int REG_PEER_POLL_SLEEP = WS_REG_PEER_POLL_SLEEP; // x 64 clocks before the first poll of the mailbox
// Poll limits on the 100 MHz wall clock.  The workgroups of ONE launch start within microseconds of each other, so an on-chip
// exchange that is not complete after 5 ms means that some workgroup is not on the chip (another kernel holds its CU):
// ws_register_cloud then repeats the registration with one launch per iteration, which needs no co-residency -- half a
// frame at 100 Hz lost, not the 2.5 frames at 10 Hz the 0.25 s of round 2 cost.  Ranks of a multi-GPU loop are launched by
// different processes that have just been handed the same scan: their mailboxes wait 20 ms (round 3: 0.25 s; WS_REG_PEER_TIMEOUT_MS
// in the environment at connect time changes it -- ranks that SHARE a GPU in the tests start further apart), kept in the PeerBlock.
constexpr long long REG_BARRIER_TIMEOUT_TICKS = 500000ll;
constexpr long long REG_PEER_TIMEOUT_TICKS = 2000000ll;

// first wave (all 64 lanes), after wave_reduce32: workgroup total of every slot, one half per lane, into the group accumulator
template <bool MFMA = false>
__device__ __forceinline__ void counted_publish(uint64_t *accum /* [REG_GROUPS][REG_WORDS] of this parity */, unsigned long long *wg_sum, bool publish,
                                                const uint32_t per_group = REG_BLOCKS / REG_GROUPS, const MfLane *mf = nullptr)
{
  const int lane = threadIdx.x & 63, slot = lane & (REG_SLOTS - 1);
  unsigned long long s;
  if (MFMA)
    s = mfma_finalize(wg_sum, *mf);
  else
  {
    s = wg_sum[slot];
    if (lane < REG_SLOTS) wg_sum[slot] = 0; // for the next iteration (the same wave read it one instruction ago)
  }
  if (!publish) return;
  const uint32_t half = lane < REG_SLOTS ? (uint32_t)((uint64_t)s & 0xffffffffull) : (uint32_t)((uint64_t)s >> 32);
  const int group = (int)(blockIdx.x / per_group);
  __hip_atomic_fetch_add(&accum[(size_t)group * REG_WORDS + lane], REG_COUNT_ONE | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// first wave: poll the accumulators of one parity until all workgroups have added, then red[0..31] = the totals of the
// iteration (read by the same wave afterwards).  then_cur / then_other: this lane's words as they stood when this / the
// other parity was last complete (rotated here).  false: gave up (another kernel is holding CUs this grid needs, or
// another workgroup gave up) -- every workgroup then leaves the loop.
__device__ __forceinline__ bool counted_collect(uint64_t *accum, uint32_t *abort_flag, uint64_t (&then_cur)[REG_GROUPS], uint64_t (&then_other)[REG_GROUPS],
                                                int64_t *red, const uint32_t per_group = REG_BLOCKS / REG_GROUPS, int64_t *total_out = nullptr)
{
  const int lane = threadIdx.x & 63;
  uint64_t w[REG_GROUPS];
  uint32_t spins = 0;
  long long t0 = 0;
  __builtin_amdgcn_s_sleep(REG_FIRST_POLL_SLEEP); // see above: a poll that fails is worse than a poll that starts late
  for (;;)
  {
    bool ok = true;
#pragma unroll
    for (int g = 0; g < REG_GROUPS; ++g)
    {
      w[g] = __hip_atomic_load(&accum[(size_t)g * REG_WORDS + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = ok && ((w[g] - then_cur[g]) >> 56) == (uint64_t)per_group;
    }
    if (__all(ok)) break;
    __builtin_amdgcn_s_sleep(REG_POLL_SLEEP);
    if ((++spins & 1023u) == 0)
    {
      const long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      const bool give_up = now - t0 > REG_BARRIER_TIMEOUT_TICKS || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      if (__any(give_up))
      {
        if (lane == 0) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
  uint64_t s = 0;
#pragma unroll
  for (int g = 0; g < REG_GROUPS; ++g)
  {
    s += (w[g] - then_cur[g]) & REG_SUM_MASK;
    const uint64_t t = then_other[g]; // the other parity is read next
    then_other[g] = w[g];
    then_cur[g] = t;
  }
  const uint64_t high = (uint64_t)shfl_xor_i64((int64_t)s, 32);
  const int64_t total = (int64_t)(s + (high << 32)); // lanes 0 .. 31: the total of slot `lane`
  if (lane < REG_SLOTS) red[lane] = total;
  if (total_out) *total_out = total;
  return true;
}

// ---- the same exchange ACROSS GPUs (point-sharded registration), inside the resident loop ---------------
// Every rank runs the resident loop on its shard.  After the on-chip exchange above, workgroup 0 of a rank ADDS the rank's 32
// totals -- again as low / high halves whose top byte counts the additions -- into a 2 x 64-word MAILBOX in every rank's
// HBM (its own included): fine-grained memory, peer-mapped (hipIpc) or local, system-scope atomics over xGMI.  Every
// workgroup then polls ITS OWN rank's mailbox (local memory) until the count says that all `world` ranks have added: the
// low 56 bits are the exact sums over the ranks, identical on every rank, and every rank goes on to the identical solve --
// no host, no launch, no RCCL call per iteration.  The words are never reset; what a parity held when it was last
// complete is carried in registers during a launch and in PeerBlock::then from launch to launch (all ranks run the same
// number of iterations, so at the end of a launch every addition ever made has been seen complete by every rank).
struct PeerBlock
{
  uint64_t *mailbox[8]; // [rank] -> that rank's mailbox: [2 parities][REG_WORDS]
  int32_t rank, world;
  uint32_t exchanges; // exchanges completed by all launches so far: the mailbox parity CONTINUES across launches (a rank that
                      // is already in the next registration adds into the parity its slower peers are NOT still polling)
  int32_t timeout_ticks; // poll limit of one exchange on the 100 MHz wall clock
  uint64_t then[2][REG_WORDS];
};
__device__ __forceinline__ bool peer_exchange(const PeerBlock *pb, int parity, uint64_t &then, int64_t &total /* lanes 0..31: in this rank's, out all ranks' */,
                                              int64_t *red, uint32_t *abort_flag)
{
  const int lane = threadIdx.x & 63;
  const int world = pb->world;
  const int64_t other = shfl_xor_i64(total, 32); // lanes 32 .. 63 take the total of slot lane - 32 from the lower half
  const uint64_t mine = (uint64_t)(lane < REG_SLOTS ? total : other);
  if (blockIdx.x == 0)
  {
    const uint32_t half = lane < REG_SLOTS ? (uint32_t)(mine & 0xffffffffull) : (uint32_t)(mine >> 32);
    for (int r = 0; r < world; ++r)
      __hip_atomic_fetch_add(pb->mailbox[r] + (size_t)parity * REG_WORDS + lane, REG_COUNT_ONE | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  uint64_t *own = pb->mailbox[pb->rank] + (size_t)parity * REG_WORDS + lane;
  uint64_t w;
  uint32_t spins = 0;
  long long t0 = 0;
  const long long limit = pb->timeout_ticks;
  __builtin_amdgcn_s_sleep(REG_PEER_POLL_SLEEP);
  for (;;)
  {
    w = __hip_atomic_load(own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (__all(((w - then) >> 56) == (uint64_t)world)) break;
    __builtin_amdgcn_s_sleep(REG_POLL_SLEEP);
    if ((++spins & 1023u) == 0)
    {
      const long long now = wall_clock64();
      if (t0 == 0) t0 = now;
      const bool give_up = now - t0 > limit || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
      if (__any(give_up))
      {
        if (lane == 0) __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
  const uint64_t sum = (w - then) & REG_SUM_MASK;
  then = w;
  const uint64_t high = (uint64_t)shfl_xor_i64((int64_t)sum, 32);
  total = (int64_t)(sum + (high << 32));
  if (lane < REG_SLOTS) red[lane] = total;
  return true;
}

struct LoopArgs
{
  PointArgs pts;
  GnCore init;       // the state the loop starts from (by value: no staging copy, no host synchronisation before the launch)
  GnState *state;    // out: state[0] (device copy for ws_reg_poll)
  GnState *result_host; // out: the same in host-mapped memory (the host only waits for the stream, no copy back)
  uint64_t *accum;   // [2][REG_GROUPS][REG_WORDS] counted group accumulators, zeroed before the launch; the abort flag (zeroed too)
                     // sits REG_ACCUM_OFFSET bytes in front of them (its own pointer would be the 257th byte of arguments)
  PeerBlock *peers;  // multi-GPU loop only
  uint32_t *clear_next; // the set of the NEXT launch (abort flag + accumulators): cleared on the way out
  uint32_t clear_words;
  int32_t debug_stall;  // test hook (ws_debug_reg_stall): workgroup 0 keeps its first contribution to itself.  Sits in the padding
                        // behind clear_words on purpose: 8 more bytes of kernel arguments made this kernel 30 % slower (1.07 -> 1.39 ms)
  int32_t *host_flag;
};
// Measured on MI355X / ROCm 7.0 (tools/reg_fit.py): with 264 bytes of kernel arguments instead of 256 an iteration of this
// kernel takes 7.86 us instead of 6.03 us -- same instructions, and 192 bytes are no faster than 256.  Keep them within 256.
static_assert(sizeof(LoopArgs) <= 256, "reg_loop_kernel: more than 256 bytes of kernel arguments");

constexpr size_t REG_ACCUM_OFFSET = 256; // accumulators behind the abort flag

// PEERS: this rank's shard of the points, a grid of any multiple of REG_GROUPS workgroups (ranks that share one GPU in the
// tests split the chip), and the cross-GPU exchange behind the on-chip one
// MFMA: the cloud (shard) has at most one point per lane -- every real scan: the reference's RegistrationCuda holds 131 072
// points -- and the sums come from the matrix cores (mfma_consume above)
template <bool PEERS, bool MFMA>
__global__ __launch_bounds__(REG_THREADS) void reg_loop_kernel(LoopArgs a)
{
  const uint32_t n_blocks = PEERS ? gridDim.x : (uint32_t)REG_BLOCKS, stride = n_blocks * REG_THREADS, per_group = n_blocks / REG_GROUPS;
  uint32_t *const abort_flag = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(a.accum) - REG_ACCUM_OFFSET);
  __shared__ unsigned long long wg_sum[REG_SLOTS + MF_AUX]; // the workgroup's totals of an iteration (LDS atomics of the eight waves)
  __shared__ alignas(16) uint32_t mf_stage[MFMA ? (REG_THREADS / 64) * MF_STAGE_WORDS : 4];
  const MfLane mfl = make_mf_lane();
  __shared__ int64_t red[REG_SLOTS];
  __shared__ alignas(16) float T_sh[16];
  __shared__ alignas(16) int32_t TI_sh[16];
  __shared__ int stop_sh;

  const Prefetched pref = prefetch_points(a.pts, stride);
  const bool wave_has_points = __ballot(pref.valid[0]) != 0ull; // later passes of the grid only have points where the first has
  const LoopGather lg = make_loop_gather(a.pts);
  GnCore st; // first wave only, identical in all of its lanes
  if (threadIdx.x < 64) st = a.init;
  uint64_t mb_then0 = 0, mb_then1 = 0; // first wave, PEERS: this lane's mailbox words of both parities when last complete
  uint32_t mb_base = 0;                // exchanges before this launch
  if (PEERS && threadIdx.x < 64)
  {
    mb_then0 = a.peers->then[0][threadIdx.x];
    mb_then1 = a.peers->then[1][threadIdx.x];
    mb_base = a.peers->exchanges;
  }
  // The loop state is uniform, so the compiler would keep it in scalar registers -- on top of the ~50 the kernel arguments
  // occupy, i.e. spilled to vector lanes and reloaded (v_readlane) in the middle of the first wave's dependency chain,
  // and everything the vector unit computes from it (all of it is float arithmetic) would cross between the two register
  // files.  Pinned to vector registers here it simply stays where it is used.
  pin_vgpr(st.center[0]); pin_vgpr(st.center[1]); pin_vgpr(st.center[2]);
  pin_vgpr(st.alpha); pin_vgpr(st.it_weight_gradient); pin_vgpr(st.epsilon);
  pin_vgpr(st.prev[0]); pin_vgpr(st.prev[1]); pin_vgpr(st.prev[2]); pin_vgpr(st.prev[3]);
  pin_vgpr(st.max_iterations); pin_vgpr(st.iterations); pin_vgpr(st.finished); pin_vgpr(st.error);
  uint64_t then_cur[REG_GROUPS], then_other[REG_GROUPS]; // first wave: the accumulator words of both parities when last complete
#pragma unroll
  for (int g = 0; g < REG_GROUPS; ++g) then_cur[g] = then_other[g] = 0;
  VoxelCache cache[2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
  {
    cache[u].bx = cache[u].by = cache[u].bz = 0;
    cache[u].cur = cache[u].xn = cache[u].xl = cache[u].yn = cache[u].yl = cache[u].zn = cache[u].zl = 0;
    cache[u].filled = false;
  }
#ifdef WS_REG_TIMING
  long long ts[7], tot[6] = {0, 0, 0, 0, 0, 0}, miss_ticks = 0, hit_ticks = 0;
  int miss_its = 0, miss_lanes = 0;
#define WS_LSTAMP(i) ts[i] = wall_clock64()
#else
#define WS_LSTAMP(i)
#endif
  float Tel = 0.f; // first wave, lanes 0 .. 15: the pose element (lane & 3, lane >> 2)
#pragma unroll
  for (int i = 0; i < 16; ++i) Tel = (int)threadIdx.x == i ? a.init.T[i] : Tel; // (a dynamic index into the arguments costs a scratch copy)
  if (threadIdx.x < 16)
  {
    T_sh[threadIdx.x] = Tel;
    store_int_pose(TI_sh, (int)threadIdx.x, Tel);
  }
  if (threadIdx.x < REG_SLOTS + MF_AUX) wg_sum[threadIdx.x] = 0;
  uint32_t k = 0;
  for (;; ++k)
  {
    WS_LSTAMP(0);
    WS_LSTAMP(1);
    WS_LSTAMP(2);
    if (threadIdx.x < 64)
    {
      if (k > 0)
      {
        // totals of iteration k - 1 (parity (k + 1) & 1) straight from the counted accumulators: this IS the grid barrier
        int64_t total = 0;
        bool ok = counted_collect(a.accum + (size_t)((k + 1) & 1) * REG_GROUPS * REG_WORDS, abort_flag, then_cur, then_other, red, per_group, &total);
        if (PEERS && ok) // the ranks' totals -> everybody's mailbox -> the totals over all ranks, in red[]
          ok = ((mb_base + k - 1) & 1) ? peer_exchange(a.peers, 1, mb_then1, total, red, abort_flag) : peer_exchange(a.peers, 0, mb_then0, total, red, abort_flag);
        WS_LSTAMP(2);
        if (!ok)
        {
          st.finished = 1;
          st.error = 1; // reported by the host
        }
        else
          gn_update_total(st, total, Tel, T_sh, TI_sh); // (red[] is only read again at the very end)
      }
      if (threadIdx.x == 0) stop_sh = (st.finished || st.iterations >= st.max_iterations) ? 1 : 0;
    }
    __syncthreads();
    WS_LSTAMP(3);
    if (stop_sh) break;

#ifdef WS_REG_TIMING
    const int32_t obx = cache[0].bx, oby = cache[0].by, obz = cache[0].bz;
    const bool ofilled = cache[0].filled;
#endif
    if (wave_has_points) // (uniform per wave; point_slot(): a small cloud or shard leaves whole waves of every workgroup without points)
    {
      if (MFMA)
      {
        const IntTransform t = load_int_pose(TI_sh);
        const Gathered g0 = gather_point_loop(a.pts, lg, t, pref.p[0][0], pref.p[0][1], pref.p[0][2], pref.valid[0], cache[0]);
        mf_v16i C;
#pragma unroll
        for (int i = 0; i < 16; ++i) C[i] = 0;
        mfma_consume(g0, C, mf_stage + (threadIdx.x >> 6) * MF_STAGE_WORDS, mfl);
        WS_LSTAMP(4);
        mfma_flush(C, wg_sum, mfl);
      }
      else
      {
        float T[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) T[i] = T_sh[i];
        int64_t acc[REG_SLOTS];
#pragma unroll
        for (int t = 0; t < REG_SLOTS; ++t) acc[t] = 0;
        accumulate_points<true>(a.pts, T, pref, acc, cache, stride);
        WS_LSTAMP(4);
        wave_reduce32_add(acc, wg_sum);
      }
    }
    __syncthreads();
    WS_LSTAMP(5);
    if (threadIdx.x < 64)
      counted_publish<MFMA>(a.accum + (size_t)(k & 1) * REG_GROUPS * REG_WORDS, wg_sum, !(a.debug_stall && blockIdx.x == 0 && k == 0), per_group, &mfl);
#ifdef WS_REG_TIMING
    WS_LSTAMP(6);
    for (int i = 0; i < 6; ++i) tot[i] += ts[i + 1] - ts[i];
#endif
#ifdef WS_REG_TIMING
    { // (after the stamps of the phases: the vote below costs a barrier)
      const bool changed = ofilled && cache[0].filled && (obx != cache[0].bx || oby != cache[0].by || obz != cache[0].bz);
      const int n_changed = __syncthreads_count(changed ? 1 : 0);
      if (n_changed > 0)
      {
        miss_its += 1;
        miss_ticks += ts[4] - ts[3];
        miss_lanes += n_changed;
      }
      else
        hit_ticks += ts[4] - ts[3];
    }
#endif
  }
#ifdef WS_REG_TIMING_GN
  if (blockIdx.x == 0 && threadIdx.x == 0)
    printf("gn_update x%lld, 10ns ticks: build %lld solve6 %lld xi_to_transform %lld pose+err %lld\n", g_gn_ticks[4], g_gn_ticks[0], g_gn_ticks[1],
           g_gn_ticks[2], g_gn_ticks[3]);
#endif
#ifdef WS_REG_TIMING
  if ((blockIdx.x % 37) == 0 && threadIdx.x == 0)
    printf("reg_loop wg %d iterations %u, 10ns ticks per phase: wait %lld sum %lld solve %lld accumulate %lld reduce %lld arrive %lld\n", (int)blockIdx.x, k,
           tot[0], tot[1], tot[2], tot[3], tot[4], tot[5]);
  if ((blockIdx.x % 37) == 0 && threadIdx.x == 0)
    printf("  wg %d: iterations with a moved point %d (%d lanes), accumulate ticks in those %lld, in the others %lld\n", (int)blockIdx.x, miss_its, miss_lanes,
           miss_ticks, hit_ticks);
#endif
  if (blockIdx.x == 0)
    for (uint32_t i = threadIdx.x; i < a.clear_words; i += REG_THREADS) a.clear_next[i] = 0u; // nobody touches that set during this launch
  if (PEERS && blockIdx.x == 0 && threadIdx.x < 64 && !st.error)
  {
    a.peers->then[0][threadIdx.x] = mb_then0; // where the next launch starts counting
    a.peers->then[1][threadIdx.x] = mb_then1;
    if (threadIdx.x == 0) a.peers->exchanges = mb_base + k; // k exchanges in this launch (the same number on every rank)
  }
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
#pragma unroll
    for (int i = 0; i < 16; ++i) st.T[i] = T_sh[i];
    a.state[0].core = st;
    a.result_host->core = st;
    if (k > 0 && !st.error)
    {
      int64_t sums[44];
      expand_sums(red, sums); // the totals the last update was made from
#pragma unroll
      for (int i = 0; i < 44; ++i) a.state[0].sums[i] = sums[i]; // (the host copy carries the state only: 44 fewer writes over PCIe)
    }
    // release: the result above is visible to the host before the flag (ws_register_cloud spins on the flag instead of
    // sleeping in hipStreamSynchronize)
    if (a.host_flag) __hip_atomic_store(a.host_flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- the same pieces as separate kernels: perform_registration (host gets h,g,e,c) and the multi-GPU
// path (partials -> 44 sums in HBM -> RCCL all-reduce -> solve) ----
struct AccArgs
{
  PointArgs pts;
  const float *T;       // 16 floats, column-major (device)
  const GnState *state; // null: no early exit
  int64_t *partials;    // [REG_BLOCKS][REG_SLOTS] (buffer 0)
};

// The three kernels of the sharded path hand small results to each other through HBM.  When they are replayed as nodes of a
// HIP graph, the runtime (ROCm 7.0) does not give a later node the cache maintenance a stream gives a later kernel: a
// batch of 16 iterations converged after ~25 instead of 178 because nodes read stale lines of their XCD's L2 (measured;
// one iteration per graph was fine).  So everything that crosses a kernel boundary here is written and read at agent
// scope (sc1: performed at the coherent level, like the exchange inside the resident loop).
__device__ __forceinline__ int32_t coherent_i32(const int32_t *p) { return __hip_atomic_load(const_cast<int32_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float coherent_f32(const float *p) { return __hip_atomic_load(const_cast<float *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int64_t coherent_i64(const int64_t *p) { return __hip_atomic_load(const_cast<int64_t *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void publish_i64(int64_t *p, int64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool loop_over(const GnState *state)
{
  return coherent_i32(&state->core.finished) != 0 || coherent_i32(&state->core.iterations) >= coherent_i32(&state->core.max_iterations);
}

__global__ __launch_bounds__(REG_THREADS) void reg_accumulate_kernel(AccArgs a)
{
  __shared__ int64_t wave_part[REG_THREADS / 64][REG_SLOTS];
  __shared__ int64_t red[REG_SLOTS];
  if (a.state != nullptr && loop_over(a.state)) return;
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = coherent_f32(&a.T[i]);
  int64_t acc[REG_SLOTS];
#pragma unroll
  for (int t = 0; t < REG_SLOTS; ++t) acc[t] = 0;
  accumulate_points(a.pts, T, prefetch_points(a.pts), acc);
  block_reduce32(acc, wave_part, red);
  if (threadIdx.x < REG_SLOTS) publish_i64(&a.partials[(size_t)blockIdx.x * REG_SLOTS + threadIdx.x], red[threadIdx.x]);
}

__global__ __launch_bounds__(REG_THREADS) void reg_sum_kernel(const int64_t *partials, const GnState *state, int64_t *sums_out)
{
  __shared__ int64_t wave_part[REG_THREADS / 64][REG_SLOTS];
  __shared__ int64_t red[REG_SLOTS];
  if (state != nullptr && loop_over(state)) return;
  sum_partials<true>(partials, wave_part, red);
  if (threadIdx.x == 0)
  {
    int64_t sums[44];
    expand_sums(red, sums);
#pragma unroll
    for (int k = 0; k < 44; ++k) publish_i64(&sums_out[k], sums[k]);
  }
}

// the solve alone, fed with externally (all-)reduced sums; updates state buffer 0
__global__ __launch_bounds__(64) void reg_solve_kernel(GnState *state, const int64_t *sums_dev)
{
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  GnCore st;
  {
    // the state the previous solve (or ws_reg_begin) left: read and written word by word at agent scope
    int32_t *w = reinterpret_cast<int32_t *>(&st);
    const int32_t *src = reinterpret_cast<const int32_t *>(&state->core);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(GnCore) / 4); ++i) w[i] = coherent_i32(&src[i]);
  }
  __shared__ int64_t s[44];
  if (threadIdx.x < 44) s[threadIdx.x] = coherent_i64(&sums_dev[threadIdx.x]);
  __syncthreads();
  gn_update(
      st, [](int r, int c) { return s[c * 6 + r]; }, [](int r) { return s[36 + r]; }, (int32_t)s[42], (int32_t)s[43]);
  if (threadIdx.x == 0)
  {
    const int32_t *w = reinterpret_cast<const int32_t *>(&st);
    int32_t *dst = reinterpret_cast<int32_t *>(&state->core);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(GnCore) / 4); ++i) __hip_atomic_store(&dst[i], w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < 42; ++k) publish_i64(&state->sums[k], s[k]);
    publish_i64(&state->sums[42], (int64_t)(int32_t)s[42]);
    publish_i64(&state->sums[43], (int64_t)(int32_t)s[43]);
  }
}

// ---- one launch per iteration of the point-sharded (multi-GPU) path -------------------------------------------------
// [apply the Gauss-Newton update from the all-reduced sums of the previous iteration] -> accumulate this rank's shard ->
// the 44 sums, in ONE kernel instead of reg_solve + reg_accumulate + reg_sum: every workgroup repeats the (cheap) update
// from the same sums, like the resident loop does, and the last workgroup to arrive (an arrival counter, agent-scope
// accesses as everywhere on this path: no cache-wide fences) adds the partials up AND writes the updated state.
// ONE state buffer, read by every workgroup before it arrives and written by the last one after all have arrived: the
// launch bakes in no buffer parity, so a batch of these launches captured into a HIP graph replays from whatever state
// the buffer holds (a host-side parity flip per launch made every replay restart from a stale buffer).
struct ShardArgs
{
  PointArgs pts;          // first / end: this rank's shard
  GnState *state;         // in: the state before this launch; out (if apply): the state after the update
  int64_t *sums;          // in: the all-reduced sums of the previous iteration (if apply); out: this rank's 44 sums
  int64_t *partials;      // [REG_BLOCKS][REG_SLOTS]
  uint32_t *arrived;      // zero between launches
  int32_t apply;
};

__global__ __launch_bounds__(REG_THREADS) void reg_shard_kernel(ShardArgs a)
{
  __shared__ int64_t wave_part[REG_THREADS / 64][REG_SLOTS];
  __shared__ int64_t red[REG_SLOTS];
  __shared__ int64_t s44[44];
  __shared__ float T_sh[16];
  __shared__ int stop_sh, last_sh;
  const Prefetched pref = prefetch_points(a.pts);
  GnCore st; // first wave only, identical in all of its lanes
  if (threadIdx.x < 64)
  {
    {
      int32_t *w = reinterpret_cast<int32_t *>(&st);
      const int32_t *src = reinterpret_cast<const int32_t *>(&a.state->core);
#pragma unroll
      for (int i = 0; i < (int)(sizeof(GnCore) / 4); ++i) w[i] = coherent_i32(&src[i]);
    }
    if (a.apply)
    {
      if (threadIdx.x < 44) s44[threadIdx.x] = coherent_i64(&a.sums[threadIdx.x]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); // LDS written and read by this wave only
      __builtin_amdgcn_wave_barrier();
      gn_update(
          st, [&](int r, int c) { return s44[c * 6 + r]; }, [&](int r) { return s44[36 + r]; }, (int32_t)s44[42], (int32_t)s44[43]);
    }
    if (threadIdx.x == 0)
    {
#pragma unroll
      for (int i = 0; i < 16; ++i) T_sh[i] = st.T[i];
      stop_sh = (st.finished || st.iterations >= st.max_iterations) ? 1 : 0;
    }
  }
  __syncthreads();
  const bool stop = stop_sh != 0; // every workgroup alike
  if (!stop)
  {
    float T[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) T[i] = T_sh[i];
    int64_t acc[REG_SLOTS];
#pragma unroll
    for (int t = 0; t < REG_SLOTS; ++t) acc[t] = 0;
    accumulate_points(a.pts, T, pref, acc);
    block_reduce32(acc, wave_part, red);
    if (threadIdx.x < REG_SLOTS) publish_i64(&a.partials[(size_t)blockIdx.x * REG_SLOTS + threadIdx.x], red[threadIdx.x]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the write-through stores above have been acknowledged ...
  }
  else if (!a.apply)
    return; // the loop was over before this launch: nothing to update, nothing to add (the sums stay as they are)
  __syncthreads(); // ... for all 32 lanes that made them
  if (threadIdx.x == 0)
    last_sh = __hip_atomic_fetch_add(a.arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (!last_sh) return;
  // everybody has read the state and the all-reduced sums: both may change now
  if (a.apply && threadIdx.x == 0)
  {
    const int32_t *w = reinterpret_cast<const int32_t *>(&st);
    int32_t *dst = reinterpret_cast<int32_t *>(&a.state->core);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(GnCore) / 4); ++i) __hip_atomic_store(&dst[i], w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int k = 0; k < 42; ++k) publish_i64(&a.state->sums[k], s44[k]);
    publish_i64(&a.state->sums[42], (int64_t)(int32_t)s44[42]);
    publish_i64(&a.state->sums[43], (int64_t)(int32_t)s44[43]);
  }
  if (!stop)
  {
    sum_partials<true>(a.partials, wave_part, red);
    if (threadIdx.x == 0)
    {
      int64_t sums[44];
      expand_sums(red, sums);
#pragma unroll
      for (int k = 0; k < 44; ++k) publish_i64(&a.sums[k], sums[k]);
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store(a.arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- perform_registration for a HOST loop (registration.cu:347-368 as the reference's callers use it) in one launch ----
// tsdf_registration.cpp:55-92 calls perform_registration once per Gauss-Newton iteration and needs h, g, e, c on the host
// before it can go on: the call is a latency chain, and until round 5 it was a 64-byte copy to the device, two launches, a
// 352-byte copy back into pageable memory and a stream synchronisation -- 34 us per iteration, 178 of them per scan.  Here the
// pose travels in the kernel arguments, the last workgroup to arrive adds the partials up and writes the 44 sums and, behind
// them, the call's sequence number into host-mapped memory; the host spins on that word (ws_reg_iterate).
struct HostIterArgs
{
  PointArgs pts;
  float T[16];        // column-major
  int64_t *partials;  // [REG_BLOCKS][REG_SLOTS]
  uint32_t *arrived;  // zero between launches
  int64_t *sums_host; // host-mapped: 44 sums, then (at [44]) the sequence number
  uint32_t seq;
};

__global__ __launch_bounds__(REG_THREADS) void reg_host_iter_kernel(HostIterArgs a)
{
  __shared__ int64_t wave_part[REG_THREADS / 64][REG_SLOTS];
  __shared__ int64_t red[REG_SLOTS];
  __shared__ int last_sh;
  int64_t acc[REG_SLOTS];
#pragma unroll
  for (int t = 0; t < REG_SLOTS; ++t) acc[t] = 0;
  float T[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) T[i] = a.T[i];
  accumulate_points(a.pts, T, prefetch_points(a.pts), acc);
  block_reduce32(acc, wave_part, red);
  if (threadIdx.x < REG_SLOTS) publish_i64(&a.partials[(size_t)blockIdx.x * REG_SLOTS + threadIdx.x], red[threadIdx.x]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the write-through stores above have been acknowledged ...
  __syncthreads();                                  // ... for all 32 lanes that made them
  if (threadIdx.x == 0) last_sh = __hip_atomic_fetch_add(a.arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (!last_sh) return;
  sum_partials<true>(a.partials, wave_part, red);
  if (threadIdx.x == 0)
  {
    int64_t sums[44];
    expand_sums(red, sums);
#pragma unroll
    for (int k = 0; k < 44; ++k) __hip_atomic_store(&a.sums_host[k], sums[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(a.arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&a.sums_host[44], (int64_t)a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- ... and WITHOUT a launch per call: a resident server (round 6) ----
// The reference's caller is unchanged -- perform_registration once per Gauss-Newton iteration, the solve on the host between
// two calls (tsdf_registration.cpp:55-92) -- and with one launch per call an iteration cost 19.8 us, of which the kernel is 3:
// the rest is the launch.  This kernel stays on the GPU ACROSS calls.  ws_reg_iterate writes the pose and a request number into
// one 64-byte line of host-mapped memory (ServerMail); workgroup 0 polls that line (one read gets request number and pose: only
// twelve of the pose's sixteen floats are used, cu_to_int_mat / registration.cu:208, the other slots carry the number, a stop
// word and a checksum of the snapshot), hands the pose to the other workgroups through device memory, every workgroup does its
// share of calc_jacobis + the reduction (the scan's points stay in registers from call to call), the last one to arrive writes
// the 44 sums and the request number back into host-mapped memory, where the host spins.  The server leaves when nothing has
// been asked for `idle_ticks` (50 us), or at once when the host says so (any other entry point that enqueues work on the
// stream: that work is ordered behind this kernel); ws_reg_iterate starts a new one when it finds none.  A request that
// meets a leaving server is not lost: the number stays in the line, the host sees `exited` without `done` and launches again.
struct ServerMail // host-mapped, 64-byte aligned
{
  uint32_t line[16];  // host -> device: pose words 0-2, 4-6, 8-10, 12-14 (column-major, rows 0-2); [3] request number (written
                      // last), [7] id of the launch the host wants gone, [11] checksum of the pose words and the number
  uint32_t pad[16];
  // device -> host: the 44 sums in seven 64-byte lines of 7 words + a TAG: request number << 32 | a hash of the line's seven words.
  // Every line validates itself, so nothing orders the lines against each other and the kernel does not wait for the writes'
  // acknowledgement before it says "done" (one fabric round trip less per request): the host takes an answer when all seven tags
  // carry its request number and match their lines.
  uint64_t answer[7][8];
  int64_t exited;     // device -> host: id of the last launch that has left the GPU
};
static_assert(offsetof(ServerMail, answer) == 128, "ServerMail");
__host__ __device__ inline uint64_t server_line_tag(uint32_t seq, const uint64_t *w /* 7 words */)
{
  uint64_t x = 0x9E3779B97F4A7C15ull;
#pragma unroll
  for (int k = 0; k < 7; ++k) x = (x << 7 | x >> 57) ^ w[k];
  return ((uint64_t)seq << 32) | (uint32_t)(x ^ (x >> 32));
}
size_t reg_server_mail_bytes() { return sizeof(ServerMail); }
__host__ __device__ inline uint32_t server_checksum(const uint32_t *line)
{
  uint32_t c = 0x5bd1e995u ^ line[3];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) c = (c << 5 | c >> 27) ^ line[j * 4 + i];
  return c;
}
void reg_server_mail_write(void *mail, const float T[16], uint32_t seq)
{
  ServerMail *m = static_cast<ServerMail *>(mail);
  uint32_t line[16] = {};
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 3; ++i) std::memcpy(&line[j * 4 + i], &T[j * 4 + i], 4);
  line[3] = seq;
  volatile uint32_t *dst = m->line;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 3; ++i) dst[j * 4 + i] = line[j * 4 + i];
  dst[11] = server_checksum(line);
  std::atomic_thread_fence(std::memory_order_release);
  dst[3] = seq; // the request is complete
}
void reg_server_mail_stop(void *mail, uint32_t launch_id)
{
  volatile uint32_t *dst = static_cast<ServerMail *>(mail)->line;
  dst[7] = launch_id;
}
// the answer to request `seq`, if it is there completely: 1 and the 44 sums, else 0
int reg_server_mail_answer(const void *mail, uint32_t seq, int64_t sums[44])
{
  const ServerMail *m = static_cast<const ServerMail *>(mail);
  uint64_t w[7][8];
  for (int j = 0; j < 7; ++j)
  {
    const volatile uint64_t *src = m->answer[j];
    const uint64_t tag = src[7];
    if ((uint32_t)(tag >> 32) != seq) return 0;
    for (int k = 0; k < 7; ++k) w[j][k] = src[k];
    w[j][7] = tag;
  }
  for (int j = 0; j < 7; ++j)
    if (server_line_tag(seq, w[j]) != w[j][7] || const_cast<volatile uint64_t *>(m->answer[j])[7] != w[j][7]) return 0; // (torn, or being rewritten)
  for (int k = 0; k < 44; ++k) sums[k] = (int64_t)w[k / 7][k % 7];
  return 1;
}
// host half of the protocol without a GPU (tests): a mail block in ordinary memory, an answer written as the kernel writes it, then
// everything that must make the host wait -- a stale request number, a line of the previous answer, a torn line.  0: all as expected
int reg_server_mail_selftest()
{
  ServerMail *m = new ServerMail();
  std::memset(m, 0, sizeof *m);
  int64_t want[44], got[44];
  auto fill = [&](uint32_t seq, int64_t salt) {
    for (int k = 0; k < 44; ++k) want[k] = (int64_t)(0x9E3779B97F4A7C15ull * (uint64_t)(k + 1)) ^ (salt << 17);
    for (int j = 0; j < 7; ++j)
    {
      uint64_t w[7];
      for (int k = 0; k < 7; ++k) w[k] = 7 * j + k < 44 ? (uint64_t)want[7 * j + k] : 0ull;
      for (int k = 0; k < 7; ++k) m->answer[j][k] = w[k];
      m->answer[j][7] = server_line_tag(seq, w);
    }
  };
  int bad = 0;
  if (reg_server_mail_answer(m, 1, got)) bad |= 1; // nothing written yet
  fill(5, 1);
  if (!reg_server_mail_answer(m, 5, got) || std::memcmp(got, want, sizeof want) != 0) bad |= 2;
  if (reg_server_mail_answer(m, 6, got) || reg_server_mail_answer(m, 4, got)) bad |= 4; // another request's answer
  // the next answer arrives line by line: incomplete until the last line is there
  int64_t old[44];
  std::memcpy(old, want, sizeof old);
  const ServerMail before = *m;
  fill(6, 2);
  const ServerMail after = *m;
  for (int upto = 0; upto < 7; ++upto)
  {
    *m = before;
    for (int j = 0; j <= upto; ++j) std::memcpy(m->answer[j], after.answer[j], sizeof m->answer[j]);
    const int ok = reg_server_mail_answer(m, 6, got);
    if (upto < 6 ? ok != 0 : (ok == 0 || std::memcmp(got, want, sizeof want) != 0)) bad |= 8;
  }
  // a torn line: words of the new answer under the old tag, and the other way round
  *m = after;
  m->answer[3][2] = before.answer[3][2];
  if (reg_server_mail_answer(m, 6, got)) bad |= 16;
  *m = after;
  m->answer[3][7] = before.answer[3][7];
  if (reg_server_mail_answer(m, 6, got)) bad |= 32;
  // the request line: what the host writes is what the kernel's checksum accepts
  float T[16];
  for (int i = 0; i < 16; ++i) T[i] = 0.25f * (float)i - 1.f;
  reg_server_mail_write(m, T, 77);
  uint32_t line[16];
  for (int i = 0; i < 16; ++i) line[i] = m->line[i];
  if (line[3] != 77 || line[11] != server_checksum(line)) bad |= 64;
  delete m;
  return bad;
}
uint32_t reg_server_mail_exited(const void *mail) { return (uint32_t) * const_cast<volatile int64_t *>(&static_cast<const ServerMail *>(mail)->exited); }
// Device memory of the server (ws_reg::srv_ctl, zero at creation, never reset afterwards):
//   lines[REG_GROUPS][16]   the host's line as workgroup 0 has copied it, once per group of 32 workgroups: what a workgroup
//                           polls -- request number, stop word, checksum and pose in ONE 64-byte read, 32 pollers per line
//   accum[REG_GROUPS][64]   counted accumulators (as in reg_loop_kernel): every workgroup ADDS its 32 sums, as low / high
//                           halves whose top byte counts the additions; workgroup 0 polls them until every count has moved
//                           by 32 -- no arrival counter (256 returning atomics on one address are 10 us: tools/doorbell_bench.hip,
//                           12.5 us per round trip with them against 3.0 for the host <-> kernel hop alone), no partials, no second pass
//   then[REG_GROUPS][64]    the accumulators as they stood after the last answered request (kept from launch to launch)
struct ServerCtl
{
  uint32_t lines[REG_GROUPS][16];
  uint64_t accum[REG_GROUPS][REG_WORDS];
  uint64_t then[REG_GROUPS][REG_WORDS];
};
size_t reg_server_ctl_bytes() { return sizeof(ServerCtl); }
struct ServerArgs
{
  PointArgs pts;
  ServerCtl *ctl;
  ServerMail *mail;  // device view
  uint32_t launch_id;
  uint32_t served;   // the last request number that was answered before this launch
  uint32_t idle_ticks; // of the 100 MHz clock
};

// one poll of a 16-word line by lanes 0..15 of a wave: 0 = nothing new, 1 = a request (number in `seq`, pose in `w`), 2 = leave
__device__ __forceinline__ int server_line_state(uint32_t w /* lane l < 16: word l */, uint32_t served, uint32_t launch_id, uint32_t &seq)
{
  seq = (uint32_t)__builtin_amdgcn_readlane((int)w, 3);
  // A request that is waiting is answered BEFORE the server obeys a word to leave: the word comes from another thread's call, whose
  // work is ordered behind this kernel either way, and a caller whose every new server found that word already there (a thread that
  // uploads maps in a loop asks each of them to leave before it has started) would never be answered.
  const bool leave = (uint32_t)__builtin_amdgcn_readlane((int)w, 7) == launch_id;
  if (seq == served || seq == 0) return leave ? 2 : 0;
  // a consistent snapshot?  (the writer stores the number last; a read that saw it and not all of the pose -- torn in two on the
  // way -- fails the checksum and is simply repeated)
  uint32_t c = 0x5bd1e995u ^ seq;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) c = (c << 5 | c >> 27) ^ (uint32_t)__builtin_amdgcn_readlane((int)w, j * 4 + i);
  return c == (uint32_t)__builtin_amdgcn_readlane((int)w, 11) ? 1 : 0;
}

template <bool MFMA>
__global__ __launch_bounds__(REG_THREADS) void reg_server_kernel(ServerArgs a)
{
  __shared__ int64_t wave_part[MFMA ? 1 : REG_THREADS / 64][REG_SLOTS];
  __shared__ int64_t red[REG_SLOTS];
  __shared__ unsigned long long wg_sum[REG_SLOTS + MF_AUX]; // MFMA: the workgroup's totals of a request (LDS atomics of the eight waves)
  __shared__ alignas(16) uint32_t mf_stage[MFMA ? (REG_THREADS / 64) * MF_STAGE_WORDS : 4];
  __shared__ alignas(16) float T_sh[16];
  __shared__ alignas(16) int32_t TI_sh[16];
  __shared__ uint32_t bell_sh; // 0: leave
  // the cloud does not change while a server lives (ws_reg_prepare* stops it): its points stay in registers from call to call,
  // and -- MFMA, clouds of at most one point per lane -- so does the voxel each point fell into with its seven entries (the map does
  // not change either: whoever enqueues an update asks the server to leave first)
  const Prefetched f = prefetch_points(a.pts);
  const bool wave_has_points = __ballot(f.valid[0]) != 0ull;
  const MfLane mfl = make_mf_lane();
  const LoopGather lg = make_loop_gather(a.pts);
  VoxelCache cache;
  cache.bx = cache.by = cache.bz = 0;
  cache.cur = cache.xn = cache.xl = cache.yn = cache.yl = cache.zn = cache.zl = 0;
  cache.filled = false;
  constexpr uint32_t per_group = REG_BLOCKS / REG_GROUPS;
  const int lane = threadIdx.x & 63;
  const int group = (int)(blockIdx.x / per_group);
  uint32_t served = a.served;
  uint64_t then[REG_GROUPS];
  if (blockIdx.x == 0 && threadIdx.x < 64)
  {
#pragma unroll
    for (int g = 0; g < REG_GROUPS; ++g) then[g] = a.ctl->then[g][lane];
  }
  if (MFMA && threadIdx.x < REG_SLOTS + MF_AUX) wg_sum[threadIdx.x] = 0;
  for (;;)
  {
    // ---- wave 0 waits for a request
    if (threadIdx.x < 64)
    {
      uint32_t w = 0, seq = 0;
      int state;
      if (blockIdx.x == 0)
      {
        // the host's line: lanes 0..15 read one word each, in ONE instruction (a 64-byte read of host memory)
        const long long t0 = wall_clock64();
        for (;;)
        {
          if (lane < 16) w = __hip_atomic_load(&a.mail->line[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          state = server_line_state(w, served, a.launch_id, seq);
          if (state != 0) break;
          if (wall_clock64() - t0 > (long long)a.idle_ticks)
          {
            state = 2;
            w = lane == 7 ? a.launch_id : 0u; // (a line that says "leave" to the other workgroups)
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        // hand it to the other workgroups: one copy per group
        if (lane < 16)
        {
#pragma unroll
          for (int g = 0; g < REG_GROUPS; ++g) __hip_atomic_store(&a.ctl->lines[g][lane], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      else
      {
        for (;;)
        {
          if (lane < 16) w = __hip_atomic_load(&a.ctl->lines[group][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          state = server_line_state(w, served, a.launch_id, seq);
          if (state != 0) break;
          __builtin_amdgcn_s_sleep(4);
        }
      }
      if (lane < 16)
      {
        T_sh[lane] = __uint_as_float(w);
        if (MFMA) store_int_pose(TI_sh, lane, __uint_as_float(w)); // (cu_to_int_mat of the twelve words that are the pose; the other four are never read)
      }
      if (lane == 0) bell_sh = state == 1 ? seq : 0u;
    }
    __syncthreads();
    const uint32_t bell = bell_sh;
    if (bell == 0) break;
    // ---- perform_registration (registration.cu:347-368) for that pose
    if (MFMA)
    {
      // the sums on the matrix cores, as in the resident loop (reg_loop_kernel)
      if (wave_has_points)
      {
        const IntTransform t = load_int_pose(TI_sh);
        const Gathered g0 = gather_point_loop(a.pts, lg, t, f.p[0][0], f.p[0][1], f.p[0][2], f.valid[0], cache);
        mf_v16i C;
#pragma unroll
        for (int i = 0; i < 16; ++i) C[i] = 0;
        mfma_consume(g0, C, mf_stage + (threadIdx.x >> 6) * MF_STAGE_WORDS, mfl);
        mfma_flush(C, wg_sum, mfl);
      }
      __syncthreads(); // (also keeps T_sh / TI_sh / bell_sh from being rewritten while anybody reads them)
    }
    else
    {
      float T[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) T[i] = T_sh[i];
      int64_t acc[REG_SLOTS];
#pragma unroll
      for (int t = 0; t < REG_SLOTS; ++t) acc[t] = 0;
      accumulate_points(a.pts, T, f, acc);
      block_reduce32(acc, wave_part, red); // (its barriers: see above)
    }
    if (threadIdx.x < 64)
    {
      uint64_t s;
      if (MFMA)
        s = mfma_finalize(wg_sum, mfl); // lanes 0..31 and 32..63: the total of slot lane & 31 (and wg_sum is zero again)
      else
        s = (uint64_t)red[lane & (REG_SLOTS - 1)];
      const uint32_t half = lane < REG_SLOTS ? (uint32_t)(s & 0xffffffffull) : (uint32_t)(s >> 32);
      __hip_atomic_fetch_add(&a.ctl->accum[group][lane], REG_COUNT_ONE | half, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (blockIdx.x == 0)
      {
        // ---- all 256 additions, then the 44 words of the reference's h, g, e, c to the host
        uint64_t wv[REG_GROUPS];
        for (;;)
        {
          bool ok = true;
#pragma unroll
          for (int g = 0; g < REG_GROUPS; ++g)
          {
            wv[g] = __hip_atomic_load(&a.ctl->accum[g][lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ok = ok && (((wv[g] - then[g]) >> 56) & 0xffu) == (uint64_t)per_group;
          }
          if (__all(ok)) break;
          __builtin_amdgcn_s_sleep(2);
        }
        uint64_t sum = 0;
#pragma unroll
        for (int g = 0; g < REG_GROUPS; ++g)
        {
          sum += (wv[g] - then[g]) & REG_SUM_MASK;
          then[g] = wv[g];
        }
        const uint64_t high = (uint64_t)shfl_xor_i64((int64_t)sum, 32);
        const int64_t total = (int64_t)(sum + (high << 32)); // lanes 0 .. 31: the total of slot `lane`
        // expand_sums: word k of the reference's 44 comes from the lane that holds its term; lane 8 j + i (i < 7) of the answer holds
        // word 7 j + i, lane 8 j + 7 the tag of line j
        const int line = lane >> 3, wi = lane & 7, k = 7 * line + wi;
        int src = 0;
        if (k < 36)
        {
          const int i = k % 6, j = k / 6;
          src = i <= j ? tri_index(i, j) : tri_index(j, i);
        }
        else if (k < 42)
          src = 21 + (k - 36);
        else if (k < 44)
          src = 27 + (k - 42);
        int64_t v = shfl_i64(total, src);
        if (k >= 42) v = (int64_t)(int32_t)v; // e and c are `int` in the reference (registration.cu:16-21)
        if (k >= 44 || wi == 7) v = 0;
        // the tag of a line: from its seven words, gathered into the line's last lane
        uint64_t x = 0x9E3779B97F4A7C15ull;
#pragma unroll
        for (int q = 0; q < 7; ++q) x = (x << 7 | x >> 57) ^ (uint64_t)shfl_i64(v, (lane & ~7) + q);
        const uint64_t tag = ((uint64_t)bell << 32) | (uint32_t)(x ^ (x >> 32));
        if (lane < 56) __hip_atomic_store(&a.mail->answer[line][wi], wi == 7 ? tag : (uint64_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    served = bell;
  }
  // ---- workgroup 0 leaves the accumulators' state for the next launch and tells the host that this one is gone (the others
  // have seen the same line and do nothing more; the next launch is ordered behind the end of this kernel)
  if (blockIdx.x == 0 && threadIdx.x < 64)
  {
#pragma unroll
    for (int g = 0; g < REG_GROUPS; ++g) a.ctl->then[g][lane] = then[g];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&a.mail->exited, (int64_t)a.launch_id, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// test entry: the wave solver alone, one wave per system (A row-major 6x6, b) -> x, status
__global__ __launch_bounds__(64) void solve6_test_kernel(const double *A, const double *b, double *x, int32_t *status)
{
  const int lane = threadIdx.x, r = lane >> 3, c = lane & 7;
  const double *Ai = A + (size_t)blockIdx.x * 36, *bi = b + (size_t)blockIdx.x * 6;
  double a = 0.0;
  if (r < 6 && c < 6) a = Ai[r * 6 + c];
  if (r < 6 && c == 6) a = bi[r];
  double xi[6] = {0, 0, 0, 0, 0, 0};
  const int rc = solve6_wave(a, xi);
  if (lane == 0)
  {
    status[blockIdx.x] = rc;
    for (int k = 0; k < 6; ++k) x[(size_t)blockIdx.x * 6 + k] = xi[k];
  }
}

int launch_solve6_test(ws_context *ctx, const double *A_dev, const double *b_dev, size_t n, double *x_dev, int32_t *status_dev)
{
  if (n == 0) return WS_OK;
  hipLaunchKernelGGL(solve6_test_kernel, dim3((unsigned)n), dim3(64), 0, ctx->stream, A_dev, b_dev, x_dev, status_dev);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

static PointArgs make_point_args(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, size_t first, size_t count)
{
  size_t end = first + count;
  if (end > r->n) end = r->n;
  if (flags & WS_REG_COMPAT_REFERENCE_LAUNCH)
  {
    // <<<128,512>>> covers points 0..65535 only; the reduction drops the last N % 32 points for N >= 128
    size_t lim = r->n;
    if (lim > 65536) lim = 65536;
    if (r->n >= 128)
    {
      size_t red = 32 * (r->n / 32);
      if (red < lim) lim = red;
    }
    if (end > lim) end = lim;
  }
  if (first > end) first = end;
  PointArgs p;
  p.points = r->points;
  p.first = (uint32_t)first;
  p.end = (uint32_t)end;
  p.map_data = m->data[WS_MAP_AVG];
  p.map = m->par[WS_MAP_AVG];
  p.resdiv = make_fastdiv(res);
  return p;
}

int launch_reg_accumulate(ws_reg *r, const ws_map *m, const float *T_dev_or_null, int32_t res, uint32_t flags, size_t first,
                          size_t count, int64_t *sums_dev)
{
  ws_context *ctx = r->ctx;
  AccArgs a;
  a.pts = make_point_args(r, m, res, flags, first, count);
  a.T = T_dev_or_null ? T_dev_or_null : r->state[r->latest].core.T;
  a.state = T_dev_or_null ? nullptr : &r->state[r->latest];
  a.partials = r->partials;
  prof_begin(ctx, WS_K_REG);
  hipLaunchKernelGGL(reg_accumulate_kernel, dim3(REG_BLOCKS), dim3(REG_THREADS), 0, ctx->stream, a);
  hipLaunchKernelGGL(reg_sum_kernel, dim3(1), dim3(REG_THREADS), 0, ctx->stream, (const int64_t *)r->partials, a.state, sums_dev);
  prof_end(ctx, WS_K_REG);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

int launch_reg_host_iter(ws_reg *r, const ws_map *m, const float T[16], int32_t res, uint32_t flags, uint32_t seq)
{
  ws_context *ctx = r->ctx;
  HostIterArgs a;
  a.pts = make_point_args(r, m, res, flags, 0, r->n);
  for (int i = 0; i < 16; ++i) a.T[i] = T[i];
  a.partials = r->partials;
  a.arrived = r->shard_arrived + 16; // (a word of its own, a cache line away from reg_shard_kernel's)
  a.sums_host = r->iter_host_dev;
  a.seq = seq;
  prof_begin(ctx, WS_K_REG);
  hipLaunchKernelGGL(reg_host_iter_kernel, dim3(REG_BLOCKS), dim3(REG_THREADS), 0, ctx->stream, a);
  prof_end(ctx, WS_K_REG);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

int launch_reg_server(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, uint32_t launch_id, uint32_t served, uint32_t idle_us)
{
  ws_context *ctx = r->ctx;
  ServerArgs a;
  a.pts = make_point_args(r, m, res, flags, 0, r->n);
  a.ctl = reinterpret_cast<ServerCtl *>(r->srv_ctl);
  a.mail = static_cast<ServerMail *>(r->srv_mail_dev);
  a.launch_id = launch_id;
  a.served = served;
  a.idle_ticks = idle_us * 100u; // wall_clock64: 100 MHz
  prof_begin(ctx, WS_K_REG);
  // at most one point per lane (every scan the reference's 131 072-point buffers can hold): the sums come from the matrix cores
  if (WS_REG_MFMA && (size_t)(a.pts.end - a.pts.first) <= (size_t)REG_BLOCKS * REG_THREADS)
    hipLaunchKernelGGL(reg_server_kernel<true>, dim3(REG_BLOCKS), dim3(REG_THREADS), 0, ctx->stream, a);
  else
    hipLaunchKernelGGL(reg_server_kernel<false>, dim3(REG_BLOCKS), dim3(REG_THREADS), 0, ctx->stream, a);
  prof_end(ctx, WS_K_REG);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

int launch_reg_solve(ws_reg *r, const int64_t *sums_dev)
{
  hipLaunchKernelGGL(reg_solve_kernel, dim3(1), dim3(64), 0, r->ctx->stream, &r->state[r->latest], sums_dev);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

int launch_reg_shard(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, size_t first, size_t count, int64_t *sums_dev, int apply)
{
  ws_context *ctx = r->ctx;
  ShardArgs a;
  a.pts = make_point_args(r, m, res, flags, first, count);
  a.state = &r->state[r->latest]; // one buffer, no parity: see reg_shard_kernel
  a.sums = sums_dev;
  a.partials = r->partials;
  a.arrived = r->shard_arrived;
  a.apply = apply ? 1 : 0;
  prof_begin(ctx, WS_K_REG);
  hipLaunchKernelGGL(reg_shard_kernel, dim3(REG_BLOCKS), dim3(REG_THREADS), 0, ctx->stream, a);
  prof_end(ctx, WS_K_REG);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

int launch_reg_iteration(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, int32_t k)
{
  ws_context *ctx = r->ctx;
  IterArgs a;
  a.pts = make_point_args(r, m, res, flags, 0, r->n);
  a.state = r->state;
  a.partials = r->partials;
  a.k = k;
  a.host_flag = r->host_flag_dev;
  prof_begin(ctx, WS_K_REG);
  hipLaunchKernelGGL(reg_iter_kernel, dim3(REG_BLOCKS), dim3(REG_THREADS), 0, ctx->stream, a);
  prof_end(ctx, WS_K_REG);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

// 1 if the device can hold the whole grid of reg_loop_kernel at once (required by its grid barrier)
int reg_loop_supported(int device)
{
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reg_loop_kernel<false, true>, REG_THREADS, 0) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
  return (long long)per_cu * cus >= REG_BLOCKS ? 1 : 0;
}

// Two sets of {abort flag, counted accumulators}, used by alternate launches: a launch finds its set zero because the
// launch before it cleared it on its way out (block 0, after its own loop) -- no memset kernel in front of every launch.
constexpr size_t REG_ACCUM_BYTES = sizeof(uint64_t) * 2 * REG_GROUPS * REG_WORDS;
constexpr size_t REG_SET_BYTES = REG_ACCUM_OFFSET + REG_ACCUM_BYTES;

int launch_reg_loop(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, const GnCore &init, bool peers, size_t first, size_t count)
{
  ws_context *ctx = r->ctx;
  LoopArgs a;
  a.pts = make_point_args(r, m, res, flags, peers ? first : 0, peers ? count : r->n);
  a.init = init;
  a.state = r->state;
  a.result_host = r->result_host_dev;
  if (!r->loop_sets_clear)
  {
    WS_HIP(hipMemsetAsync(r->grid_bar, 0, 2 * REG_SET_BYTES, ctx->stream));
    r->loop_sets_clear = true;
  }
  char *mine = reinterpret_cast<char *>(r->grid_bar) + (r->loop_launches & 1u) * REG_SET_BYTES;
  char *other = reinterpret_cast<char *>(r->grid_bar) + ((r->loop_launches + 1) & 1u) * REG_SET_BYTES;
  r->loop_launches += 1;
  a.accum = reinterpret_cast<uint64_t *>(mine + REG_ACCUM_OFFSET); // (the abort flag is the first word of the set)
  a.peers = reinterpret_cast<PeerBlock *>(r->peer_block_dev);
  a.clear_next = reinterpret_cast<uint32_t *>(other);
  a.clear_words = (uint32_t)(REG_SET_BYTES / sizeof(uint32_t));
  a.host_flag = r->host_flag_dev;
  a.debug_stall = r->debug_stall_next;
  r->debug_stall_next = 0;
  prof_begin(ctx, WS_K_REG);
  const unsigned blocks = peers ? (unsigned)r->peer_blocks : (unsigned)REG_BLOCKS;
  // at most one point per lane (every scan the reference's 131 072-point buffers can hold): the sums come from the matrix cores
  const bool mfma = WS_REG_MFMA && (size_t)(a.pts.end - a.pts.first) <= (size_t)blocks * REG_THREADS;
  if (peers)
  {
    if (mfma)
      hipLaunchKernelGGL((reg_loop_kernel<true, true>), dim3(blocks), dim3(REG_THREADS), 0, ctx->stream, a);
    else
      hipLaunchKernelGGL((reg_loop_kernel<true, false>), dim3(blocks), dim3(REG_THREADS), 0, ctx->stream, a);
  }
  else
  {
    if (mfma)
      hipLaunchKernelGGL((reg_loop_kernel<false, true>), dim3(blocks), dim3(REG_THREADS), 0, ctx->stream, a);
    else
      hipLaunchKernelGGL((reg_loop_kernel<false, false>), dim3(blocks), dim3(REG_THREADS), 0, ctx->stream, a);
  }
  prof_end(ctx, WS_K_REG);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

// host image of PeerBlock (api.hip fills it: the mailbox pointers are peer-mapped or local device addresses)
size_t reg_peer_block_bytes() { return sizeof(PeerBlock); }
void reg_peer_block_fill(void *host_image, void *const mailbox[8], int rank, int world)
{
  PeerBlock *pb = reinterpret_cast<PeerBlock *>(host_image);
  std::memset(pb, 0, sizeof(PeerBlock));
  for (int i = 0; i < 8; ++i) pb->mailbox[i] = reinterpret_cast<uint64_t *>(i < world ? mailbox[i] : nullptr);
  pb->rank = rank;
  pb->world = world;
  long long ticks = REG_PEER_TIMEOUT_TICKS;
  if (const char *ms = std::getenv("WS_REG_PEER_TIMEOUT_MS"))
  {
    const long long v = std::atoll(ms);
    if (v >= 1 && v <= 20000) ticks = v * 100000ll;
  }
  pb->timeout_ticks = (int32_t)ticks;
}
size_t reg_mailbox_bytes() { return sizeof(uint64_t) * 2 * REG_WORDS; }
int reg_groups() { return REG_GROUPS; }
int reg_default_blocks() { return REG_BLOCKS; }

size_t reg_barrier_bytes() { return 2 * REG_SET_BYTES; }

size_t reg_partials_bytes() { return sizeof(int64_t) * 2 * REG_SLOTS * REG_BLOCKS; }

} // namespace ws

// tsdf_update.hip — TSDF volume update for MI355X (gfx950).
//
// Replaces cu_min_tsdf_krnl + cu_avg_tsdf_krnl (src/warpsense/cuda/update_tsdf.cu:13-128) of the reference.
//
// The reference scatters with a racy CAS ("first positive-weight entry freezes the voxel",
// include/warpsense/cuda/util.h:70-102), so its result depends on thread arrival order.  This
// implementation computes the result of ONE fixed legal schedule — the serial one, ascending
// (point, ray step, fan step) — deterministically, without a single global atomic per candidate:
//
//   ray_setup/scan/scatter   per-ray constants (update_tsdf.cu:52-63), rays grouped by direction.
//   march_tail_kernel        walks the ray TAILS once (near-surface and fan candidates: everything whose result
//                            depends on the order).  Every scatter target becomes a 16-byte record
//                            (order key, tile, voxel-in-tile) appended to the workgroup's private slice with
//                            coalesced stores; the workgroup then sorts its own records by tile through an LDS hash
//                            and publishes one run descriptor per (workgroup, tile).
//   march_free_kernel        walks the steps before the tails: all free space (tau, +64) whoever comes first ->
//                            one byte per voxel.  A free-space candidate landing on a voxel the tails marked goes
//                            into a small (voxel -> earliest key) hash instead.
//   tile_count/scan/list     runs per tile -> contiguous descriptor ranges + the list of touched tiles.
//   desc_place_kernel        run descriptors grouped by tile.
//   tile_resolve_kernel      ONE workgroup per touched 4x4x64-voxel tile: all candidates of the tile are present,
//                            so the canonical accept rule is a local fold in LDS (earliest positive / smallest
//                            negative keys, then exact rounds for the voxels where a negative-weight candidate
//                            may have blocked the earliest positive one).  Writes new_map, or — fused — integrates
//                            straight into avg_map (cu_avg_tsdf_krnl folded into the write-back).
//   integrate_*_kernel       weighted average of new_map into avg_map and reset of new_map, over the touched
//                            tiles only (sparse) or over every voxel (dense, the reference's kernel).
//
// new_map after the resolve is bit-identical to what the reference kernel leaves there when its threads
// run one after the other (oracle/ws_oracle.c: wso_update_min).
#include <atomic>
#include <chrono>
#include <cstddef>

#include "ws_march.h"

namespace ws
{

struct ScatterArgs
{
  const int32_t *xyz;
  uint32_t n;
  int32_t scanner_pos[3];
  int32_t up[3];
  MapParams map; // new_map's parameters (the reference indexes new_map in the scatter, update_tsdf.cu:55-125)
  int32_t tau;
  int32_t res;
  int32_t ntx, nty, ntz;
  int32_t all_keyed;     // new_map is not (tau, 0): every candidate goes through the order keys, no free-space pass
  int32_t keyed_len_neg; // smallest ray length with off-ray (negative-weight) candidates
  int32_t keyed_slack;   // see ray_setup_kernel
  RaySetup *rays;
  uint32_t *az_hist;   // [AZ_BINS + 1] rays per direction bin (last bin: rays that contribute nothing)
  uint32_t *az_off;    // [AZ_BINS]: number of rays that contribute (written by the direction sort)
  uint2 *ray_bin;      // [n] (direction bin, rank inside the bin) of every ray: set-up blocks -> sort blocks of the same launch
  uint32_t *ray_order; // ray indices sorted by direction bin
  const int32_t *fan_steps; // [256], see tail_bound
  uint8_t *vstate;     // one byte per voxel: VOX_*
  uint8_t *tile_dirty; // one byte per tile: touched by the free-space pass
  uint32_t *tile_nruns;
  CandRecord *rec_raw;
  CandRecord *rec_sorted;
  uint32_t rec_cap;
  uint32_t scan_seq; // sequence number of this scatter (in the padding behind rec_cap: the arguments stay within 256 bytes)
  RunDesc *desc;
  uint32_t desc_cap;
  unsigned long long *fk_keys; // the values follow the keys (fk_keys + fk_mask + 1): one allocation, and the arguments stay within 256 bytes
  int32_t fk_shift; // 64 - log2(slots)
  uint32_t fk_mask;
  uint32_t *tail_stats; // records per workgroup of the tail march
  TsdfCounters *counters;
  uint32_t *status; // host-mapped: [0] sticky error bits, [4..5] record bound of the scan in flight, [6] its sequence number
};
// 264 bytes of kernel arguments instead of 256 cost reg_loop_kernel 30 % (registration.hip); the same bound here
static_assert(sizeof(ScatterArgs) <= 256, "ScatterArgs: more than 256 bytes of kernel arguments");

constexpr uint8_t VOX_KEYED = 1, VOX_TOUCHED = 2, VOX_FREEHIT = 4;
constexpr uint32_t ERR_CAPACITY = 1, ERR_RANGE = 2, ERR_FREE_BOUND = 4, ERR_INTERNAL = 8;

#ifndef WS_FUSE_SETUP
#define WS_FUSE_SETUP 0 // 1: ray set-up and direction sort in one launch (sort blocks wait for the set-up blocks)
#endif
#ifndef WS_SORT_BLOCKS
#define WS_SORT_BLOCKS 64
#endif
#ifndef WS_FUSE_PLACE
#define WS_FUSE_PLACE 1 // 1: tile scan and descriptor placement in one launch
#endif
#ifndef WS_TAIL_PERSISTENT
#define WS_TAIL_PERSISTENT 0 // (measured: 216-222 us against 187-195 us for one workgroup per item: resident workgroups run in lock step and their sort phases collide)
//  1: the tail march runs as resident workgroups that take (64 rays x 4 parts) items from a counter
#endif
#ifndef WS_SORT_U
#define WS_SORT_U 8 // (4: 188-191 us, 8: 185-187 us, 2: 196-198 us) records a thread of the tail march has in flight while it copies the workgroup's records into tile order
#endif
constexpr int SORT_U = WS_SORT_U;
#ifndef WS_FREE_PIPE
#define WS_FREE_PIPE 1 // 1: the voxel byte of a free-space candidate is requested one emit phase before it is used (126 -> 122 us)
#endif
#ifndef WS_EL_BINS
#define WS_EL_BINS 8
#endif
constexpr int AZ_ONLY_BINS = 1024, EL_BINS = WS_EL_BINS;
constexpr int AZ_BINS = AZ_ONLY_BINS * EL_BINS; // direction bins: azimuth major, elevation minor

size_t ray_setup_bytes() { return sizeof(RaySetup); }

__device__ __forceinline__ void raise_error(TsdfCounters *c, uint32_t *status, uint32_t bits)
{
  atomicOr(&c->error, bits);
  __hip_atomic_fetch_or(status, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // sticky, host visible
}

// ring-buffer storage coordinates of a world voxel (device_map.h:93-101)
__device__ __forceinline__ void storage_coords(const MapParams &m, int32_t vx, int32_t vy, int32_t vz, int32_t &sx, int32_t &sy, int32_t &sz)
{
  sx = ring(vx - m.pos[0] + m.offset[0] + m.size[0], m.size[0]);
  sy = ring(vy - m.pos[1] + m.offset[1] + m.size[1], m.size[1]);
  sz = ring(vz - m.pos[2] + m.offset[2] + m.size[2], m.size[2]);
}
__device__ __forceinline__ int64_t storage_index(const MapParams &m, int32_t sx, int32_t sy, int32_t sz)
{
  // sizes are below 2^24 and size[0] * size[1] below 2^31 (checked by ws_map_create): one full-rate 24-bit multiply-add
  const int32_t row = (int32_t)(__umul24((uint32_t)sx, (uint32_t)m.size[1]) + (uint32_t)sy);
  return (int64_t)row * (int64_t)m.size[2] + sz;
}
__device__ __forceinline__ uint32_t tile_of(int32_t nty, int32_t ntz, int32_t sx, int32_t sy, int32_t sz)
{
  // ntx * nty < 2^24 (checked by ws_map_create): full-rate 24-bit multiplies
  const uint32_t col = __umul24((uint32_t)(sx >> TILE_XB), (uint32_t)nty) + (uint32_t)(sy >> TILE_YB);
  return __umul24(col, (uint32_t)ntz) + (uint32_t)(sz >> TILE_ZB);
}
__device__ __forceinline__ uint32_t local_of(int32_t sx, int32_t sy, int32_t sz)
{
  return (uint32_t)(((sx & ((1 << TILE_XB) - 1)) << (TILE_YB + TILE_ZB)) | ((sy & ((1 << TILE_YB) - 1)) << TILE_ZB) | (sz & ((1 << TILE_ZB) - 1)));
}

// everything the scatter expects to be zero / empty, in ONE launch (five memsets cost five launch gaps)
struct PrepArgs
{
  TsdfCounters *counters;
  uint32_t *az_hist;
  uint32_t n_hist;
  uint32_t *tile_nruns;
  int64_t n_tiles;
  unsigned long long *fk; // keys and values: one allocation
  int64_t n_fk;
  unsigned long long *look; // look-back words of the tile scan
  uint32_t n_look;
};
__device__ __forceinline__ void scatter_prep(const PrepArgs &p, bool counters_too)
{
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  if (counters_too && tid < (int64_t)(offsetof(TsdfCounters, last_records) / 4)) reinterpret_cast<uint32_t *>(p.counters)[tid] = 0;
  for (int64_t i = tid; i < p.n_hist; i += stride) p.az_hist[i] = 0;
  for (int64_t i = tid; i < p.n_tiles; i += stride) p.tile_nruns[i] = 0;
  // the free-space key hash: all of it (the scans themselves only clear the slots they claimed, fk_clear_claimed)
  uint32_t *claimed = reinterpret_cast<uint32_t *>(p.fk + p.n_fk);
  for (int64_t i = tid; i < p.n_fk; i += stride) p.fk[i] = KEY_INF;
  for (int64_t i = tid; i < p.n_fk / 64; i += stride) claimed[i] = 0;
  for (int64_t i = tid; i < p.n_look; i += stride) p.look[i] = 0;
}
// the slots of the free-space key hash the PREVIOUS scan claimed (one bit per slot behind the hash) back to empty; run
// by the set-up blocks of a scan, long before its free-space pass
__device__ __forceinline__ void fk_clear_claimed(unsigned long long *fk, uint32_t slots, int64_t tid, int64_t stride)
{
  uint32_t *claimed = reinterpret_cast<uint32_t *>(fk + 2 * (size_t)slots);
  for (int64_t i = tid; i < (int64_t)(slots / 32); i += stride)
  {
    uint32_t bits = claimed[i];
    if (bits == 0) continue;
    claimed[i] = 0;
    while (bits)
    {
      const uint32_t h = (uint32_t)i * 32u + (uint32_t)__builtin_ctz(bits);
      bits &= bits - 1;
      fk[h] = KEY_INF;
      fk[(size_t)slots + h] = KEY_INF;
    }
  }
}
__global__ __launch_bounds__(256) void scatter_prep_kernel(PrepArgs p) { scatter_prep(p, true); }

// Upper bound of the scatter targets of the ray steps [k0, k1): sum of iter_steps = 2*delta_z/res + 1 (update_tsdf.cu:101-102)
// = (k1 - k0) + sum_j #{steps with delta_z >= ceil(j*res/2)}.  Exact when every sample is a candidate; additive over ranges.
// fan_steps[j] = first step k with delta_z(len_k) >= ceil(j*res/2), len_k = 1 + k*(res/2): a function of res alone, tabulated
// on the host when the map is created (fill_fan_steps) -- two 64-bit divisions per j and ray otherwise.
__device__ __forceinline__ unsigned long long tail_bound(int64_t k0, int64_t k1, int64_t len_end, const int32_t *fan_steps)
{
  if (k1 <= k0) return 0;
  unsigned long long ub = (unsigned long long)(k1 - k0);
  if ((int64_t)DZ_PER_DISTANCE * len_end >= (1ll << 31)) return ub * 256; // DZ * len wraps in the reference's int: any fan width the key admits
  for (int j = 1; j < 256; ++j)
  {
    const int64_t kj = fan_steps[j];
    if (kj >= k1) break; // non-decreasing in j
    ub += (unsigned long long)(k1 - (kj > k0 ? kj : k0));
  }
  return ub;
}

// update_tsdf.cu:52-63 for one ray per lane, plus the split of the ray into free-space steps and tail
__device__ __forceinline__ void ray_setup_block(const ScatterArgs &a, uint32_t n_setup_blocks)
{
  __shared__ unsigned long long ub_wave[4];
  const uint32_t ix = blockIdx.x * 256u + threadIdx.x;
  // the free-space key hash still holds what the previous scan claimed: back to empty, long before this scan's free pass.
  // (this thread's word of the claimed-slots bitmap is requested here and used at the END of the block: its round trip
  // runs under the ray arithmetic)
  const uint32_t fk_slots = a.fk_mask + 1u;
  uint32_t *const fk_claimed = reinterpret_cast<uint32_t *>(a.fk_keys + 2 * (size_t)fk_slots);
  const bool fk_mine = ix < fk_slots / 32u && (uint64_t)n_setup_blocks * 256u >= fk_slots / 32u;
  const uint32_t fk_bits = fk_mine ? fk_claimed[ix] : 0u;
  RaySetup r;
  r.dx = r.dy = r.dz = r.distance = r.ivx = r.ivy = r.ivz = r.steps = 0;
  r.div_m = 0;
  r.div_k = 0;
  r.pad = 0;
  r.kfirst = 0;
  r.ub = 0;
  const int32_t res = a.res, tau = a.tau, half = res / 2;
  bool ok = false;
  int32_t px = 0, py = 0, pz = 0;
  if (ix < a.n)
  {
    px = a.xyz[3 * (size_t)ix + 0];
    py = a.xyz[3 * (size_t)ix + 1];
    pz = a.xyz[3 * (size_t)ix + 2];
    // cu_to_map (cuda/util.h:111-114) + in_bounds_with_buffer_pos (update_tsdf.cu:55)
    const float fr = (float)res;
    const int32_t cx = (int32_t)floorf(__fdiv_rn((float)px, fr));
    const int32_t cy = (int32_t)floorf(__fdiv_rn((float)py, fr));
    const int32_t cz = (int32_t)floorf(__fdiv_rn((float)pz, fr));
    ok = in_bounds_buffer(a.map, cx, cy, cz, (int64_t)(tau / res / 2));
  }
  if (ok)
  {
    // cu_to_mm (cuda/util.h:116-123)
    const int32_t posx = wadd(wmul(a.scanner_pos[0], res), half);
    const int32_t posy = wadd(wmul(a.scanner_pos[1], res), half);
    const int32_t posz = wadd(wmul(a.scanner_pos[2], res), half);
    const int32_t dx = wsub(px, posx), dy = wsub(py, posy), dz = wsub(pz, posz);
    const int32_t distance = l2norm_i(dx, dy, dz);
    // distance == 0: guard (the reference divides by zero here; src/cpu/update_tsdf.cpp:593 has the guard)
    if (distance > 0)
    {
      // update_tsdf.cu:59-63, in int64 like the reference's `long`
      const int64_t MR = MATRIX_RESOLUTION;
      const int64_t ndx = div_trunc_i64(wmul64(dx, MR), distance), ndy = div_trunc_i64(wmul64(dy, MR), distance), ndz = div_trunc_i64(wmul64(dz, MR), distance);
      const int64_t ux = a.up[0], uy = a.up[1], uz = a.up[2];
      const int64_t c1x = wsub64(wmul64(ndy, uz), wmul64(ndz, uy)) / MR;
      const int64_t c1y = wsub64(wmul64(ndz, ux), wmul64(ndx, uz)) / MR;
      const int64_t c1z = wsub64(wmul64(ndx, uy), wmul64(ndy, ux)) / MR;
      int64_t ivx = wsub64(wmul64(ndy, c1z), wmul64(ndz, c1y));
      int64_t ivy = wsub64(wmul64(ndz, c1x), wmul64(ndx, c1z));
      int64_t ivz = wsub64(wmul64(ndx, c1y), wmul64(ndy, c1x));
      const int64_t inorm = l2norm_l(ivx, ivy, ivz);
      if (inorm != 0) // guard (src/cpu/update_tsdf.cpp:602)
      {
        ivx = div_trunc_i64(wmul64(ivx, MR), inorm);
        ivy = div_trunc_i64(wmul64(ivy, MR), inorm);
        ivz = div_trunc_i64(wmul64(ivz, MR), inorm);
        const int64_t len_end = (int64_t)distance + tau;
        const int64_t steps = div_trunc_i64(len_end - 1, half) + 1;
        const int64_t max_delta_z = (int64_t)DZ_PER_DISTANCE * len_end / MATRIX_RESOLUTION;
        const bool small_iv = ivx >= INT32_MIN && ivx <= INT32_MAX && ivy >= INT32_MIN && ivy <= INT32_MAX && ivz >= INT32_MIN && ivz <= INT32_MAX;
        if (steps > 65536 || (max_delta_z * 2) / res + 1 > 256 || !small_iv)
        {
          raise_error(a.counters, a.status, ERR_RANGE); // outside the range of the order key
        }
        else
        {
          r.dx = dx; r.dy = dy; r.dz = dz;
          r.distance = distance;
          r.ivx = (int32_t)ivx; r.ivy = (int32_t)ivy; r.ivz = (int32_t)ivz;
          r.steps = (int32_t)steps;
          const FastDiv fd = make_fastdiv_dev(distance);
          r.div_m = fd.M;
          r.div_k = fd.k;
          // conditions of march_steps_fast (ws_march.h): no int32 wrap in d*len, pos + d, voxel centres,
          // delta_z*iv and step*res*iv, nor in the squared distance to the hit point (|p - centre| <= len_end + 2 res)
          const int64_t dmax = max(max(llabs((long long)dx), llabs((long long)dy)), llabs((long long)dz));
          const int64_t pmax = max(max(llabs((long long)posx), llabs((long long)posy)), llabs((long long)posz));
          const int64_t ivmax = max(max(llabs(ivx), llabs(ivy)), llabs(ivz));
          // dmax <= distance: beyond ~46 m the reference's int sum of squares wraps and `distance` is not the length
          // of the ray any more — the walk's "less than one voxel per step" then fails
          const bool fast = dmax <= distance && dmax * len_end < (1ll << 31) && pmax + dmax + 2 * (int64_t)res + tau < (1ll << 30) &&
                            (2 * max_delta_z + res) * ivmax < (1ll << 31) &&
                            (len_end + 2 * (int64_t)res) * (len_end + 2 * (int64_t)res) < (1ll << 31);
          r.pad = fast ? RAY_FAST : 0;
          {
            // the whole ray, its fans included, inside the window with room to spare: the per-candidate in_bounds tests
            // (update_tsdf.cu:73,113) cannot fail.  Both ends inside a box shrunk by the fan reach (convexity does the rest).
            const int64_t margin = 4 + (max_delta_z + res) / res;
            const int64_t endx = (int64_t)posx + div_trunc_i64((int64_t)dx * len_end, distance), endy = (int64_t)posy + div_trunc_i64((int64_t)dy * len_end, distance),
                          endz = (int64_t)posz + div_trunc_i64((int64_t)dz * len_end, distance);
            const int64_t ev[3] = {div_trunc_i64(endx, res), div_trunc_i64(endy, res), div_trunc_i64(endz, res)};
            bool inside = fast;
            for (int k = 0; k < 3; ++k)
            {
              const int64_t lim = (int64_t)(a.map.size[k] / 2) - margin;
              inside = inside && llabs((long long)((int64_t)a.scanner_pos[k] - a.map.pos[k])) <= lim && llabs((long long)(ev[k] - a.map.pos[k])) <= lim;
            }
            if (inside) r.pad |= RAY_SIMPLE;
          }

          // Split of the ray.  A candidate is "free space" iff it is on-ray (positive weight) with value == +tau;
          // ALL candidates of a ray are of that kind while len < min(len_neg, distance - tau - slack): before len_neg
          // there is no fan, and a voxel centre further than tau from the hit point gives min(dist, tau) == tau.  The
          // slack covers |centre - proj| (1.5 voxels per axis for the double-width cell of trunc division + the fan
          // offset).  The bound argues with exact positions: rays whose `int` products wrap (not `fast`) and scans
          // into a non-default new_map send every step through the order keys.
          int32_t kfirst = 0;
          if (fast && !a.all_keyed)
          {
            const int32_t keyed_len = min(a.keyed_len_neg, distance - tau - a.keyed_slack);
            kfirst = keyed_len > 1 ? max(0, (keyed_len - 1) / half - 1) : 0;
            if (kfirst > (int32_t)steps) kfirst = (int32_t)steps;
          }
          r.kfirst = kfirst;
          const unsigned long long ub = tail_bound(kfirst, steps, len_end, a.fan_steps);
          r.ub = ub > 0xffffffffull ? 0xffffffffu : (uint32_t)ub;
        }
      }
    }
  }
  // direction bin of the ray (any monotone function of the direction would do: it only groups rays that lie in
  // the same vertical plane, whose voxels share tiles); bin AZ_BINS = unused ray
  if (ix < a.n)
  {
    uint32_t bin = AZ_BINS;
    if (r.steps > 0)
    {
      const float az = atan2f((float)r.dy, (float)r.dx); // [-pi, pi]
      int b = (int)((az + 3.14159265f) * ((float)AZ_ONLY_BINS / 6.2831853f));
      b = b < 0 ? 0 : (b >= AZ_ONLY_BINS ? AZ_ONLY_BINS - 1 : b);
      // elevation: sin(el) = dz / distance in [-1, 1]; LiDARs use the middle of that range: clamp +-0.5
      float se = (float)r.dz / (float)r.distance;
      int e = (int)((se + 0.5f) * (float)EL_BINS);
      e = e < 0 ? 0 : (e >= EL_BINS ? EL_BINS - 1 : e);
      bin = (uint32_t)(b * EL_BINS + e);
    }
    r.pad |= (int32_t)(bin << 1); // bits 1 .. 14 (RAY_SIMPLE is bit 30)
    // the histogram's old value is this ray's rank inside its bin: the sort blocks place it without a second atomic.
    // Both travel at agent scope (performed at the coherent level): the sort blocks run on other XCDs in the same launch.
    const uint32_t rank = __hip_atomic_fetch_add(&a.az_hist[bin], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(&a.ray_bin[ix]), (unsigned long long)bin | ((unsigned long long)rank << 32), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
    a.rays[ix] = r;
  }
  // The record slots this scan can need (sum of the per-ray bounds): one 64-bit add per workgroup; the last workgroup to
  // get here hands the total to the host (host-mapped memory: value, then the sequence number the host spins on), which
  // sizes the record buffers BEFORE it enqueues the tail march -- the capacity never rests on a guess.
  unsigned long long ub = r.ub;
  for (int d = 32; d > 0; d >>= 1) ub += __shfl_down(ub, d, 64);
  if ((threadIdx.x & 63) == 0) ub_wave[threadIdx.x >> 6] = ub;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this lane's (bin, rank) has been written through ...
  __syncthreads();                                  // ... and so has everybody else's in the workgroup
  if (threadIdx.x == 0)
  {
    const unsigned long long t = ub_wave[0] + ub_wave[1] + ub_wave[2] + ub_wave[3];
    // ONE atomic per workgroup carries both the sum and the arrival count (bits 48..: workgroups, at most 3907 of them;
    // below: record slots, < 2^32): the workgroup that finds everybody else's count in the old value also has the total in
    // it -- no second atomic, no read-back, and no cache-wide fence (an agent-scope release / acquire pair walks the XCD's
    // L2: 15-20 us on this kernel, measured; two dependent returning atomics + a coherent load: ~5 us at the kernel's tail).
    const unsigned long long before = __hip_atomic_fetch_add(&a.counters->ub_total, t + (1ull << 48), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(before >> 48) == n_setup_blocks - 1)
    {
      const unsigned long long total = (before & ((1ull << 48) - 1ull)) + t;
      __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.status + 4), total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(a.status + 6, a.scan_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // the claimed slots of the free-space hash (see the top of the block)
  if ((uint64_t)n_setup_blocks * 256u >= fk_slots / 32u)
  {
    uint32_t bits = fk_bits;
    if (bits)
    {
      fk_claimed[ix] = 0;
      while (bits)
      {
        const uint32_t h = ix * 32u + (uint32_t)__builtin_ctz(bits);
        bits &= bits - 1;
        a.fk_keys[h] = KEY_INF;
        a.fk_keys[(size_t)fk_slots + h] = KEY_INF;
      }
    }
  }
  else
    fk_clear_claimed(a.fk_keys, fk_slots, (int64_t)ix, (int64_t)n_setup_blocks * 256); // small scans: a strided loop
}

// Counting sort of the rays by direction bin, in the SAME launch as the set-up: blocks [0, S) are the set-up blocks above,
// blocks [S, 2S) wait until all of them have counted (workgroups are dispatched in index order, so a sort block can only be
// on the chip when every set-up block is there or done: no deadlock whatever the grid size) and place the rays.  One launch
// and its ~10 us of dependent start-up less on the critical path.  Every sort block scans the 8193-entry histogram itself
// (32 KB, one block scan).
template <bool FUSED>
__device__ __forceinline__ void ray_sort_block(const ScatterArgs &a, uint32_t n_setup_blocks)
{
  __shared__ uint32_t s_off[AZ_BINS + 2];
  __shared__ uint32_t wave_sums[4];
  constexpr int TOTAL = AZ_BINS + 1;
  constexpr int PER = (TOTAL + 255) / 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (FUSED)
  {
    if (threadIdx.x == 0)
      while ((uint32_t)(__hip_atomic_load(&a.counters->ub_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 48) < n_setup_blocks) __builtin_amdgcn_s_sleep(16);
    __syncthreads();
  }
  const int lo = threadIdx.x * PER, hi = min(lo + PER, TOTAL);
  uint32_t h[PER];
  uint32_t v = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j)
  {
    const int i = lo + j;
    // (a launch of its own sees the histogram through the kernel boundary; fused, it was counted on other XCDs in this launch)
    h[j] = i < hi ? (FUSED ? __hip_atomic_load(&a.az_hist[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.az_hist[i]) : 0u;
    v += h[j];
  }
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wave_sums[wave] = x;
  __syncthreads();
  uint32_t run = x - v;
  for (int w = 0; w < wave; ++w) run += wave_sums[w];
#pragma unroll
  for (int j = 0; j < PER; ++j)
  {
    const int i = lo + j;
    if (i < hi) s_off[i] = run;
    run += h[j];
  }
  if (hi == TOTAL && lo < hi) s_off[TOTAL] = run;
  __syncthreads();
  const uint32_t b = blockIdx.x - n_setup_blocks, nb = gridDim.x - n_setup_blocks;
  if (b == 0 && threadIdx.x == 0) a.az_off[AZ_BINS] = s_off[AZ_BINS]; // rays that contribute: the tail march's grid
  for (uint32_t ix = b * 256u + threadIdx.x; ix < a.n; ix += nb * 256u)
  {
    const unsigned long long br = FUSED ? __hip_atomic_load(reinterpret_cast<unsigned long long *>(&a.ray_bin[ix]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : *reinterpret_cast<const unsigned long long *>(&a.ray_bin[ix]);
    a.ray_order[s_off[(uint32_t)br] + (uint32_t)(br >> 32)] = ix;
  }
}

// n_setup_blocks == gridDim.x: set-up only (the sort follows as a launch of its own)
__global__ __launch_bounds__(256) void ray_setup_sort_kernel(ScatterArgs a, uint32_t n_setup_blocks)
{
  if (blockIdx.x < n_setup_blocks)
    ray_setup_block(a, n_setup_blocks);
  else
    ray_sort_block<true>(a, n_setup_blocks);
}
__global__ __launch_bounds__(256) void ray_sort_kernel(ScatterArgs a) { ray_sort_block<false>(a, 0); }

// ---------------------------------------------------------------------------------------------------------
// ray tails -> records, sorted by tile inside the workgroup
// ---------------------------------------------------------------------------------------------------------
constexpr int HT_BITS = 10, HT_SLOTS = 1 << HT_BITS;
#ifndef WS_TAIL_SPLIT
#define WS_TAIL_SPLIT 2
#endif
#ifndef WS_TAIL_WGS
#define WS_TAIL_WGS 6 // workgroups per CU the register budget is set for (80 VGPRs, no spills; 5 -> 6: 184 -> 181 us)
#endif
constexpr int TAIL_SPLIT = WS_TAIL_SPLIT; // workgroups that share the tails of one group of 64 rays (4 parts each)
constexpr int TAIL_QCAP = 128; // queue entries per wave of the compacting walk (one sample phase adds at most 64)
constexpr uint32_t HT_EMPTY = 0xffffffffu, REC_DONE = 0xffffffffu;

__device__ __forceinline__ int ht_insert(uint32_t *keys, uint32_t tile)
{
  uint32_t h = (tile * 0x9E3779B1u) >> (32 - HT_BITS);
  for (int p = 0; p < HT_SLOTS; ++p)
  {
    const uint32_t cur = keys[h];
    if (cur == tile) return (int)h;
    if (cur == HT_EMPTY)
    {
      const uint32_t old = atomicCAS(&keys[h], HT_EMPTY, tile);
      if (old == HT_EMPTY || old == tile) return (int)h;
    }
    h = (h + 1) & (HT_SLOTS - 1);
  }
  return -1;
}
__device__ __forceinline__ int ht_find(const uint32_t *keys, uint32_t tile)
{
  uint32_t h = (tile * 0x9E3779B1u) >> (32 - HT_BITS);
  for (int p = 0; p < HT_SLOTS; ++p)
  {
    const uint32_t cur = keys[h];
    if (cur == tile) return (int)h;
    if (cur == HT_EMPTY) return -1;
    h = (h + 1) & (HT_SLOTS - 1);
  }
  return -1;
}

// slot of the calling lane in a bump allocation shared by the lanes that reach this point together:
// one LDS atomic per wave, not per lane
__device__ __forceinline__ uint32_t lds_append(uint32_t *cursor)
{
  const unsigned long long mask = __ballot(1);
  const int lane = (int)(threadIdx.x & 63);
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(cursor, (uint32_t)__popcll(mask));
  base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader); // (the leader is uniform: no trip through LDS for the broadcast, -3 us in the tail march)
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(4))) u32x4_a4; // four consecutive voxels of a column: dword aligned only
typedef uint32_t __attribute__((aligned(1))) u32_a1; // four consecutive vstate bytes

// A workgroup takes 64 rays of neighbouring directions (ray_order) and the four quarters of their tails (one
// quarter per wave): its scatter targets fall into the same vertical slab of space, i.e. into few tiles.
// does the scan in flight fit the record buffers?  (uniform: the set-up pass has finished, its total is final)
__device__ __forceinline__ bool scan_fits(const ScatterArgs &a)
{
  // a plain (scalar) load: the total was finished by an earlier kernel.  (As a coherent load by every thread -- a million of
  // them on one address -- this line alone took the tail march from 190 to 500 us.)
  const unsigned long long need = a.counters->ub_total & ((1ull << 48) - 1ull);
  return need <= (unsigned long long)a.rec_cap;
}

// one work item: 64 direction-sorted rays x four of the 4 * TAIL_SPLIT parts of their tails
__device__ __forceinline__ void tail_item(const ScatterArgs &a, const uint32_t item)
{
  __shared__ uint32_t s_cursor, s_base, s_ub, s_overflow, s_desc_base, s_round_total;
  __shared__ uint32_t ht_key[HT_SLOTS], ht_cnt[HT_SLOTS], ht_cur[HT_SLOTS];
  __shared__ unsigned long long s_wave[4];
  __shared__ u32x4 s_queue[4 * TAIL_QCAP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t n_sorted = a.az_off[AZ_BINS];
  const uint32_t slot = (item / (uint32_t)TAIL_SPLIT) * 64u + (uint32_t)lane;
  const int part0 = (int)(item % (uint32_t)TAIL_SPLIT) * 4; // this workgroup's four parts of the tails
  const bool has_ray = slot < n_sorted;
  uint32_t ix = 0;
  RaySetup r;
  r.steps = 0;
  r.kfirst = 0;
  r.ub = 0;
  if (has_ray)
  {
    ix = a.ray_order[slot];
    r = a.rays[ix];
  }
  // ---- phase 0: reserve a slice of the raw record buffer for the upper bound of this workgroup's records
  if (threadIdx.x == 0)
  {
    s_cursor = 0;
    s_overflow = 0;
  }
  for (int i = threadIdx.x; i < HT_SLOTS; i += 256)
  {
    ht_key[i] = HT_EMPTY;
    ht_cnt[i] = 0;
  }
  if (wave == 0)
  {
    unsigned long long ub = 0;
    if (has_ray && r.steps > 0 && r.kfirst < r.steps)
    {
      const int32_t chp = (r.steps - r.kfirst + 4 * TAIL_SPLIT - 1) / (4 * TAIL_SPLIT);
      const int32_t ka = min(r.kfirst + part0 * chp, r.steps), kb = min(r.kfirst + (part0 + 4) * chp, r.steps);
      ub = TAIL_SPLIT == 1 ? (unsigned long long)r.ub : tail_bound(ka, kb, (int64_t)r.distance + a.tau, a.fan_steps);
    }
    for (int d = 32; d > 0; d >>= 1) ub += __shfl_down(ub, d, 64);
    if (lane == 0)
    {
      uint32_t base = 0xffffffffu;
      if (ub == 0)
        base = 0;
      else if (ub <= a.rec_cap)
      {
        const uint32_t b = atomicAdd(&a.counters->raw_cursor, (uint32_t)ub);
        if (b <= a.rec_cap - (uint32_t)ub) base = b;
      }
      if (base == 0xffffffffu) raise_error(a.counters, a.status, ERR_CAPACITY);
      s_base = base;
      s_ub = (uint32_t)ub;
    }
  }
  __syncthreads();
  const uint32_t base = s_base;
  if (base == 0xffffffffu) return; // record buffer exhausted: this workgroup's candidates are lost, the error is sticky
  const uint32_t ub_total = s_ub;

#ifdef WS_TAIL_TIMING
  const long long tt0 = wall_clock64(); // 100 MHz, the same clock on every CU: start / middle / end per workgroup -> ws_debug_block_stats
#endif
  // ---- phase 1: march, one record per scatter target
  const MarchFrame f = make_march_frame(a.scanner_pos, a.res, a.tau, a.map);
  const bool mark = !a.all_keyed;
  // one scatter target -> one record (vx, vy, vz: world voxel inside the window)
  auto put_record = [&](uint32_t rix, int32_t k, int32_t step, int32_t vx, int32_t vy, int32_t vz, int32_t value, bool positive) {
    const int32_t sx = ring_fast(vx, f.ringK[0], a.map.size[0]), sy = ring_fast(vy, f.ringK[1], a.map.size[1]),
                  sz = ring_fast(vz, f.ringK[2], a.map.size[2]);
    // the free-space pass must know that this voxel takes part in the key order
    if (mark) a.vstate[storage_index(a.map, sx, sy, sz)] = VOX_KEYED;
    const uint32_t p = lds_append(&s_cursor);
    if (p < ub_total)
    {
      u32x4 rec;
      const uint64_t key = record_key(order_key(rix, k, step), value, positive);
      rec.x = (uint32_t)key;
      rec.y = (uint32_t)(key >> 32);
      rec.z = tile_of(a.nty, a.ntz, sx, sy, sz);
      // the workgroup's tile histogram is built on the fly (the slot travels in the record); a full table defers
      // the record to the extra rounds of phase 2
      const int slot = ht_insert(ht_key, rec.z);
      // (one LDS atomic per lane, most of them on the same counter: the hardware takes them together -- counting the lanes
      // of a slot with ballots and adding once per (wave, tile) made the kernel 205 -> 345 us)
      if (slot >= 0)
        atomicAdd(&ht_cnt[slot], 1u);
      else
        s_overflow = 1;
      rec.w = local_of(sx, sy, sz) | ((uint32_t)(slot >= 0 ? slot : HT_SLOTS) << 10);
      *reinterpret_cast<u32x4 *>(&a.rec_raw[base + p]) = rec;
    }
    else
    {
      raise_error(a.counters, a.status, ERR_INTERNAL); // the upper bound must hold; never write out of the slice
    }
  };
  int32_t k0 = 0, k1 = 0;
  if (has_ray && r.steps > 0 && r.kfirst < r.steps)
  {
    const int32_t kbeg = r.kfirst, kend = r.steps;
    const int32_t ch = (kend - kbeg + 4 * TAIL_SPLIT - 1) / (4 * TAIL_SPLIT);
    k0 = min(kbeg + (part0 + wave) * ch, kend);
    k1 = min(k0 + ch, kend);
  }
  const bool work = k0 < k1;
  if (!__all(!work || (r.pad & RAY_SIMPLE)))
  {
    // a ray of this wave wraps in int32 or leaves the window: the general walk with all its tests
    if (work)
      march_steps<false>(f, r, k0, k1, [&](int32_t k, int32_t step, int32_t vx, int32_t vy, int32_t vz, int32_t value, bool positive) {
        put_record(ix, k, step, vx, vy, vz, value, positive);
      });
  }
  else if (__any(work))
  {
    // compacting walk (ws_march.h): the sample phase queues (position, step, ray) of every sample that enters a new
    // voxel column; the emit phase pops 64 of them and does update_tsdf.cu:81-125 with every lane busy
    u32x4 *queue = s_queue + wave * TAIL_QCAP;
    uint32_t qhead = 0, qtail = 0;
    const int32_t res = f.res, half = f.half, tau = f.tau, dist = r.distance;
    const int32_t hitx = f.posx + r.dx, hity = f.posy + r.dy, hitz = f.posz + r.dz; // the scan point (update_tsdf.cu:57)
    AxisRun ix0, iy0, iz0;
    ix0.r = ix0.ar = ix0.aq = ix0.q = ix0.spos = ix0.sm = 0;
    ix0.gap = 0x3fffffff;
    iy0 = ix0;
    iz0 = ix0;
    int32_t k = k0; // the next sample of this lane
    if (work)
    {
      const int32_t kinit = k0 > 0 ? k0 - 1 : 0;
      run_init(ix0, f, r, r.dx, f.posx, kinit, true);
      run_init(iy0, f, r, r.dy, f.posy, kinit, true);
      run_init(iz0, f, r, r.dz, f.posz, kinit, false);
    }
    // the branch-free sample step of ws_march.h (lanes that are through keep stepping, masked)
    AxisFast wx = fast_from(ix0, work ? dist : 1), wy = fast_from(iy0, work ? dist : 1), wz = fast_from(iz0, work ? dist : 1);
    auto push = [&](bool cand, bool cx, bool cy) {
      const unsigned long long mask = __ballot(cand);
      if (mask == 0) return;
      if (cand)
      {
        u32x4 e;
        e.x = (uint32_t)fast_proj(wx, cx, res);
        e.y = (uint32_t)fast_proj(wy, cy, res);
        e.z = (uint32_t)fast_proj(wz, false, res);
        e.w = (uint32_t)k | ((uint32_t)lane << 16);
        const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        queue[(qtail + rank) & (TAIL_QCAP - 1)] = e;
      }
      qtail += (uint32_t)__popcll(mask);
    };
    {
      // the sample k == 0 is compared with the voxel column (0, 0) (update_tsdf.cu:65,71) and is where the walk starts: out
      // of the loop, so that every iteration is "step, then test"
      bool first = false;
      if (work && k0 == 0) first = div_res(fast_proj(wx, false, res), f) != 0 || div_res(fast_proj(wy, false, res), f) != 0;
      push(first, false, false);
      if (work && k0 == 0) k = 1;
    }
    int32_t todo = work ? k1 - k : 0;
    for (int d = 32; d > 0; d >>= 1) todo = max(todo, __shfl_xor(todo, d, 64));
    const int32_t n_iter = __builtin_amdgcn_readfirstlane(todo);
    for (int32_t it = 0;; ++it)
    {
      const bool any_alive = it < n_iter;
      if (any_alive)
      {
        // ---- sample phase
        const bool cx = fast_step(wx, res), cy = fast_step(wy, res);
        fast_step_z(wz);
        push((cx || cy) && k < k1, cx, cy);
        k += 1;
      }
      // ---- emit phase: 64 queued samples, one per lane
      const uint32_t cnt = qtail - qhead;
      if (cnt >= 64 || (!any_alive && cnt > 0))
      {
        const uint32_t n = cnt < 64 ? cnt : 64;
        u32x4 e = {0, 0, 0, 0};
        const bool has = (uint32_t)lane < n;
        if (has) e = queue[(qhead + (uint32_t)lane) & (TAIL_QCAP - 1)];
        qhead += n;
        // constants of the ray the sample belongs to (a lane of this wave)
        const int src = (int)(e.w >> 16);
        const int32_t s_hitx = __shfl(hitx, src, 64), s_hity = __shfl(hity, src, 64), s_hitz = __shfl(hitz, src, 64);
        const int32_t s_ivx = __shfl(r.ivx, src, 64), s_ivy = __shfl(r.ivy, src, 64), s_ivz = __shfl(r.ivz, src, 64);
        const int32_t s_dist = __shfl(r.distance, src, 64);
        const uint32_t s_ix = (uint32_t)__shfl((int)ix, src, 64);
        if (has)
        {
          const int32_t ek = (int32_t)(e.w & 0xffffu);
          const int32_t projx = (int32_t)e.x, projy = (int32_t)e.y, projz = (int32_t)e.z;
          const int32_t len = 1 + ek * half;
          // update_tsdf.cu:81-98 (no int32 wrap for a RAY_SIMPLE ray: 24-bit multiplies are exact)
          const int32_t ddx = s_hitx - (__mul24(div_res(projx, f), res) + half), ddy = s_hity - (__mul24(div_res(projy, f), res) + half),
                        ddz = s_hitz - (__mul24(div_res(projz, f), res) + half);
          int32_t value = (int32_t)sqrtf((float)(__mul24(ddx, ddx) + __mul24(ddy, ddy) + __mul24(ddz, ddz)));
          value = value < tau ? value : tau;
          if (len > s_dist) value = -value;
          if (!tsdf_weight_is_zero(value, tau, f.weight_epsilon))
          {
            // update_tsdf.cu:101-125
            const int32_t delta_z = (DZ_PER_DISTANCE * len) >> 15; // len > 0
            int32_t iter_steps = 1, mid = 0;
            if (delta_z * 2 >= res)
            {
              iter_steps = (int32_t)(__umulhi((uint32_t)(delta_z * 2), f.rM32) >> f.rS) + 1;
              mid = (int32_t)(__umulhi((uint32_t)delta_z, f.rM32) >> f.rS);
            }
            const int32_t lowx = projx - trunc_shift15(__mul24(delta_z, s_ivx)), lowy = projy - trunc_shift15(__mul24(delta_z, s_ivy)),
                          lowz = projz - trunc_shift15(__mul24(delta_z, s_ivz));
            for (int32_t step = 0; step < iter_steps; ++step)
            {
              const int32_t sm = step * res;
              const int32_t vx = div_res(lowx + trunc_shift15(__mul24(sm, s_ivx)), f), vy = div_res(lowy + trunc_shift15(__mul24(sm, s_ivy)), f),
                            vz = div_res(lowz + trunc_shift15(__mul24(sm, s_ivz)), f);
              put_record(s_ix, ek, step, vx, vy, vz, value, step == mid);
            }
          }
        }
      }
      if (!any_alive && qtail == qhead) break;
    }
  }
  __syncthreads();
#ifdef WS_TAIL_TIMING
  const long long tt2 = wall_clock64();
  if (threadIdx.x == 0)
  {
    a.tail_stats[16384 + item] = (uint32_t)tt0;
    a.tail_stats[32768 + item] = (uint32_t)tt2;
    a.tail_stats[49152 + item] = (uint32_t)tt2;
  }
#endif
  const uint32_t total = min(s_cursor, ub_total);
  if (threadIdx.x == 0) a.tail_stats[item] = total;
  if (total == 0) return;

  // ---- phase 2: sort the slice by tile (counting sort over an LDS hash of the tiles this workgroup touched) and
  // publish one run per tile.  If more tiles are touched than the hash holds, the rest is binned in further rounds.
  uint32_t round_base = 0;
  for (int round = 0;; ++round)
  {
    if (round > 0)
    {
      // records the table of the previous round had no room for
      for (int i = threadIdx.x; i < HT_SLOTS; i += 256)
      {
        ht_key[i] = HT_EMPTY;
        ht_cnt[i] = 0;
      }
      __syncthreads();
      for (uint32_t i0 = threadIdx.x; i0 < total; i0 += 1024)
      {
        uint32_t tile[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
          const uint32_t i = i0 + (uint32_t)u * 256u;
          tile[u] = i < total ? a.rec_raw[base + i].tile : REC_DONE;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
        {
          if (tile[u] == REC_DONE) continue;
          const int s = ht_insert(ht_key, tile[u]);
          if (s < 0)
            s_overflow = 1;
          else
            atomicAdd(&ht_cnt[s], 1u);
        }
      }
      __syncthreads();
    }
    // exclusive scan over the slots: records (low word) and runs (high word) together
    uint32_t c[4];
    unsigned long long mine = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      c[j] = ht_cnt[threadIdx.x * 4 + j];
      mine += (unsigned long long)c[j] + (c[j] ? (1ull << 32) : 0ull);
    }
    unsigned long long incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      const unsigned long long y = __shfl_up(incl, d, 64);
      if (lane >= d) incl += y;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned long long excl = incl - mine;
    for (int w = 0; w < wave; ++w) excl += s_wave[w];
    if (threadIdx.x == 255)
    {
      const unsigned long long all = excl + mine;
      const uint32_t runs = (uint32_t)(all >> 32);
      s_round_total = (uint32_t)all;
      uint32_t db = 0xffffffffu;
      if (runs)
      {
        const uint32_t b = atomicAdd(&a.counters->desc_cursor, runs);
        if (b <= a.desc_cap && runs <= a.desc_cap - b) db = b;
        if (db == 0xffffffffu) raise_error(a.counters, a.status, ERR_CAPACITY);
      }
      s_desc_base = db;
    }
    __syncthreads();
    const uint32_t desc_base = s_desc_base;
    {
      uint32_t off = (uint32_t)excl, rank = (uint32_t)(excl >> 32);
#pragma unroll
      for (int j = 0; j < 4; ++j)
      {
        const int s = threadIdx.x * 4 + j;
        ht_cur[s] = off;
        if (c[j])
        {
          if (desc_base != 0xffffffffu)
          {
            RunDesc d;
            d.tile = ht_key[s];
            d.count = c[j];
            d.start = base + round_base + off;
            d.pad = 0;
            a.desc[desc_base + rank] = d;
            atomicAdd(&a.tile_nruns[d.tile], 1u);
          }
          rank += 1;
          off += c[j];
        }
      }
    }
    __syncthreads();
    const bool more = s_overflow != 0;
    for (uint32_t i0 = threadIdx.x; i0 < total; i0 += 256u * SORT_U)
    {
      u32x4 rec[SORT_U];
#pragma unroll
      for (int u = 0; u < SORT_U; ++u)
      {
        // unconditional (clamped) loads, SORT_U in flight; a load under a branch would be waited for on the spot
        const uint32_t i = i0 + (uint32_t)u * 256u;
        rec[u] = *reinterpret_cast<const u32x4 *>(&a.rec_raw[base + (i < total ? i : total - 1)]);
        if (i >= total) rec[u].z = REC_DONE;
      }
#pragma unroll
      for (int u = 0; u < SORT_U; ++u)
      {
        if (rec[u].z == REC_DONE) continue;
        // round 0: the slot was found when the record was written; later rounds look the tile up again
        int s = (int)(rec[u].w >> 10);
        if (round > 0)
          s = ht_find(ht_key, rec[u].z);
        else if (s >= HT_SLOTS)
          s = -1;
        if (s < 0) continue; // next round
        const uint32_t p = atomicAdd(&ht_cur[s], 1u);
        rec[u].w &= (uint32_t)(TILE_VOXELS - 1);
        *reinterpret_cast<u32x4 *>(&a.rec_sorted[base + round_base + p]) = rec[u];
        if (more) a.rec_raw[base + i0 + (uint32_t)u * 256u].tile = REC_DONE;
      }
    }
    __syncthreads();
    if (!more) break;
    round_base += s_round_total;
    __syncthreads();
    if (threadIdx.x == 0) s_overflow = 0;
  }
#ifdef WS_TAIL_TIMING
  if (threadIdx.x == 0) a.tail_stats[49152 + item] = (uint32_t)wall_clock64();
#endif
}

// Persistent workgroups (as many as the chip holds at once) that take work items from a counter: a launch of one workgroup
// per item spent a third of its time ramping up and draining (tools/tail_schedule.py: 4096 workgroups of 48 us each over 1536
// slots finished after 190 us, the sum of their durations over the slots is 127 us) -- items differ by 10x in work.
__global__ __launch_bounds__(256, WS_TAIL_WGS) void march_tail_kernel(ScatterArgs a)
{
#if WS_TAIL_PERSISTENT
  __shared__ uint32_t s_item;
#endif
  // the direction histogram has been consumed by the sort blocks of this scan: zero for the next one (no clean-up launch)
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < (uint32_t)(AZ_BINS + 1); i += gridDim.x * 256u) a.az_hist[i] = 0;
  const uint32_t n_items = ((a.n + 63u) / 64u) * (uint32_t)TAIL_SPLIT;
  // The whole update is enqueued before the host has seen the record bound of this scan (below: launch_tsdf_scatter).  If the
  // scan does not fit the record buffers, NOTHING of it may happen: the tail march and the free pass leave at once (no byte
  // of the map's state is touched, the later kernels find nothing to do), and the host grows the buffers and runs it again.
  if (scan_fits(a) == false) return;
#if WS_TAIL_PERSISTENT
  for (;;)
  {
    if (threadIdx.x == 0) s_item = atomicAdd(&a.counters->tail_next, 1u);
    __syncthreads();
    const uint32_t item = s_item;
    if (item >= n_items) break;
    tail_item(a, item);
    __syncthreads(); // everybody is done with the item's LDS state (and has read s_item)
  }
#else
  if (blockIdx.x < n_items) tail_item(a, blockIdx.x);
#endif
}

// ---------------------------------------------------------------------------------------------------------
// free space
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fk_hash(unsigned long long idx, int32_t shift) { return (uint32_t)((idx * 0x9E3779B97F4A7C15ull) >> shift); }

// What a free-space candidate does to its voxel, in two halves: the byte of the voxel is REQUESTED when the candidate is
// popped from the queue and USED one emit phase later.  The free pass is not bound by instruction issue alone: shortening
// the sample phase from ~115 to ~60 instructions moved it from 137 to 126 us, taking this load's round trip off the wave's
// path to 122 us; what remains is the scattered byte traffic itself (21 M byte loads, 9 M byte stores, one cache line each).
struct FreePending
{
  int64_t idx;   // voxel (storage index)
  uint32_t tile;
  uint32_t ix;   // ray
  int32_t k;     // ray step
  uint32_t b;    // the voxel's byte (in flight until the next emit phase)
  bool valid;
};
__device__ __forceinline__ void free_request(const ScatterArgs &a, const MarchFrame &f, FreePending &p, bool valid, uint32_t ix, int32_t k, int32_t vx, int32_t vy,
                                             int32_t vz)
{
  const int32_t sx = ring_fast(vx, f.ringK[0], a.map.size[0]), sy = ring_fast(vy, f.ringK[1], a.map.size[1]),
                sz = ring_fast(vz, f.ringK[2], a.map.size[2]);
  p.valid = valid;
  p.idx = valid ? storage_index(a.map, sx, sy, sz) : 0; // unconditional (clamped) load: nothing waits for it here
  p.tile = tile_of(a.nty, a.ntz, sx, sy, sz);
  p.ix = ix;
  p.k = k;
  p.b = a.vstate[p.idx];
}
__device__ __forceinline__ void free_finish(const ScatterArgs &a, const FreePending &p)
{
  if (!p.valid) return;
  const int64_t idx = p.idx;
  const uint32_t b = p.b;
  if (b & VOX_KEYED)
  {
    // the voxel also has ordered candidates (from the tails): this one takes part in the key order.  Only the
    // earliest free-space candidate of a voxel can matter (a later one meets a state that is at least as final).
    if (!(b & VOX_FREEHIT)) a.vstate[idx] = VOX_KEYED | VOX_FREEHIT;
    const unsigned long long t = order_key(p.ix, p.k, 0);
    uint32_t h = fk_hash((unsigned long long)idx, a.fk_shift);
    bool done = false;
    for (int q = 0; q < 128 && !done; ++q)
    {
      const unsigned long long cur = a.fk_keys[h];
      unsigned long long old = cur;
      if (cur == KEY_INF) old = atomicCAS(&a.fk_keys[h], KEY_INF, (unsigned long long)idx);
      if (old == KEY_INF) // claimed: one bit per slot behind the hash tells the next scan's set-up where to clean
        atomicOr(&reinterpret_cast<uint32_t *>(a.fk_keys + 2 * ((size_t)a.fk_mask + 1))[h >> 5], 1u << (h & 31u));
      if (old == KEY_INF || old == (unsigned long long)idx)
      {
        atomicMin(&a.fk_keys[(size_t)a.fk_mask + 1 + h], t);
        done = true;
      }
      h = (h + 1) & a.fk_mask;
    }
    if (!done) raise_error(a.counters, a.status, ERR_CAPACITY);
  }
  else if (b == 0)
  {
    // free space only (the common case): the result will be (tau, 64) whoever comes first.  (Two candidates of one voxel
    // whose loads both saw 0 both store: idempotent.)
    a.vstate[idx] = VOX_TOUCHED;
    // (remembering the tiles a workgroup has marked in an LDS set instead of this load: 126 -> 140 us, measured)
    if (a.tile_dirty[p.tile] == 0) a.tile_dirty[p.tile] = 1;
  }
}
// both halves at once (general walk)
__device__ __forceinline__ void free_emit(const ScatterArgs &a, const MarchFrame &f, uint32_t ix, int32_t k, int32_t vx, int32_t vy, int32_t vz)
{
  FreePending p;
  free_request(a, f, p, true, ix, k, vx, vy, vz);
  free_finish(a, p);
}

constexpr int FREE_QCAP = 128; // queue entries per wave (one sample phase adds at most 64)
#ifndef WS_FREE_LANES
#define WS_FREE_LANES 4
#endif
constexpr int FREE_LANES = WS_FREE_LANES; // lanes that share the free-space part of one ray

// 64 rays per workgroup, 4 lanes per ray (round 2 walk: 32 lanes 163 us, 16: 146, 8: 141, 4: 147, 1: 280; round 3 walk: 8: 123, 4: 120, 2: 131): lane c walks the steps [c*CH, (c+1)*CH) of the free-space part of its ray,
// so every lane has the same amount of work whatever the ray length.  Waves whose rays are all RAY_SIMPLE use the
// compacting walk (ws_march.h): samples for all lanes, candidates through a per-wave LDS queue, 64 at a time.
__global__ __launch_bounds__(256) void march_free_kernel(ScatterArgs a)
{
  if (scan_fits(a) == false) return; // see march_tail_kernel
  __shared__ u32x4 s_queue[4 * FREE_QCAP];
  const uint32_t ix = blockIdx.x * (uint32_t)(256 / FREE_LANES) + threadIdx.x / (uint32_t)FREE_LANES;
  const int32_t c = (int32_t)(threadIdx.x % (uint32_t)FREE_LANES);
  const int lane = threadIdx.x & 63;
  RaySetup r;
  r.steps = 0;
  r.kfirst = 0;
  r.pad = 0;
  if (ix < a.n) r = a.rays[ix];
  const int32_t kend = min(r.steps, r.kfirst);
  const int32_t ch = (kend + FREE_LANES - 1) / FREE_LANES;
  const int32_t k0 = c * ch;
  const int32_t k1 = min(k0 + ch, kend);
  const bool work = k0 < k1;
  const int32_t tau = a.tau;
  const MarchFrame f = make_march_frame(a.scanner_pos, a.res, tau, a.map);
  if (!__all(!work || (r.pad & RAY_SIMPLE)))
  {
    // a ray of this wave wraps in int32 or leaves the window: the general walk with all its tests
    if (!work) return;
    march_steps<true>(f, r, k0, k1, [&](int32_t k, int32_t step, int32_t vx, int32_t vy, int32_t vz, int32_t value, bool positive) {
      // every candidate of these steps is free space: on the ray, further than tau from the hit point
      if (!(positive && value == tau))
      {
        raise_error(a.counters, a.status, ERR_FREE_BOUND); // impossible by the bound; never lose a candidate silently
        return;
      }
      free_emit(a, f, ix, k, vx, vy, vz);
    });
    return;
  }
  if (!__any(work)) return;

  // wave-private ring buffer: LDS operations of one wave are performed in order, so no barrier between push and pop
  u32x4 *queue = s_queue + (threadIdx.x >> 6) * FREE_QCAP;
  uint32_t qhead = 0, qtail = 0;
  const int32_t res = f.res, half = f.half, dist = r.distance;
  // 64 queued candidates, one per lane (fewer at the very end): finish the batch whose voxel bytes were requested by the
  // previous emit phase, then pop the next batch and request its bytes
  FreePending pend;
  pend.valid = false;
  pend.idx = 0;
  pend.tile = pend.ix = pend.b = 0;
  pend.k = 0;
  auto emit = [&]() {
#if WS_FREE_PIPE
    free_finish(a, pend);
#endif
    const uint32_t cnt = qtail - qhead;
    const uint32_t n = cnt < 64 ? cnt : 64;
    u32x4 e = {0, 0, 0, 0};
    const bool has = (uint32_t)lane < n;
    if (has) e = queue[(qhead + (uint32_t)lane) & (FREE_QCAP - 1)];
    const uint32_t src_ix = (uint32_t)__shfl((int)ix, (int)(e.w >> 16), 64);
    free_request(a, f, pend, has, src_ix, (int32_t)(e.w & 0xffffu), div_res((int32_t)e.x, f), div_res((int32_t)e.y, f), div_res((int32_t)e.z, f));
#if !WS_FREE_PIPE
    free_finish(a, pend);
    pend.valid = false;
#endif
    qhead += n;
  };
  AxisRun ix0, iy0, iz0;
  ix0.r = ix0.ar = ix0.aq = ix0.q = ix0.spos = ix0.sm = 0;
  ix0.gap = 0x3fffffff;
  iy0 = ix0;
  iz0 = ix0;
  int32_t k = k0; // the next sample of this lane
  if (work)
  {
    const int32_t kinit = k0 > 0 ? k0 - 1 : 0;
    run_init(ix0, f, r, r.dx, f.posx, kinit, true);
    run_init(iy0, f, r, r.dy, f.posy, kinit, true);
    run_init(iz0, f, r, r.dz, f.posz, kinit, false);
  }
  AxisFast wx = fast_from(ix0, work ? dist : 1), wy = fast_from(iy0, work ? dist : 1), wz = fast_from(iz0, work ? dist : 1);
  int32_t last_dz = -1, c0x = 0, c0y = 0, c0z = 0;
  // target of the single on-ray candidate of a free-space sample (update_tsdf.cu:103-112 with iter_steps == 1): the sample
  // minus the fan base offset, which changes every 328 mm of ray
  auto push = [&](bool cand, bool cx, bool cy, int32_t dzl) {
    const unsigned long long mask = __ballot(cand);
    if (mask == 0) return;
    if (cand)
    {
      const int32_t px = fast_proj(wx, cx, res), py = fast_proj(wy, cy, res), pz = fast_proj(wz, false, res);
      const int32_t delta_z = dzl >> 15; // (DZ_PER_DISTANCE * len) >> 15, len > 0; no fan in the free-space part: delta_z * 2 < res
      if (delta_z != last_dz)
      {
        last_dz = delta_z;
        c0x = trunc_shift15(delta_z * r.ivx);
        c0y = trunc_shift15(delta_z * r.ivy);
        c0z = trunc_shift15(delta_z * r.ivz);
      }
      u32x4 e;
      e.x = (uint32_t)(px - c0x);
      e.y = (uint32_t)(py - c0y);
      e.z = (uint32_t)(pz - c0z);
      e.w = (uint32_t)k | ((uint32_t)lane << 16);
      const uint32_t rank = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
      queue[(qtail + rank) & (FREE_QCAP - 1)] = e;
    }
    qtail += (uint32_t)__popcll(mask);
  };
  // the sample k == 0 is compared with the voxel column (0, 0) (update_tsdf.cu:65,71) and is where the walk was initialised:
  // taken out of the loop, so that every iteration below is "step, then test"
  {
    bool first = false;
    if (work && k0 == 0)
    {
      const int32_t px = fast_proj(wx, false, res), py = fast_proj(wy, false, res);
      first = div_trunc(px, f.rM, f.rK, res) != 0 || div_trunc(py, f.rM, f.rK, res) != 0;
    }
    push(first, false, false, DZ_PER_DISTANCE); // len == 1
    if (work && k0 == 0) k = 1;
  }
  // iterations of the wave: the longest lane (uniform: the loop itself is scalar)
  int32_t todo = work ? k1 - k : 0;
  for (int d = 32; d > 0; d >>= 1) todo = max(todo, __shfl_xor(todo, d, 64));
  const int32_t n_iter = __builtin_amdgcn_readfirstlane(todo);
  int32_t dzl = DZ_PER_DISTANCE * (1 + k * half); // DZ_PER_DISTANCE * len of the sample k, carried (no multiply per sample)
  const int32_t dzl_step = DZ_PER_DISTANCE * half;
  for (int32_t it = 0; it < n_iter; ++it)
  {
    // ---- sample phase: every lane steps (lanes that are through keep stepping; their samples are masked)
    const bool cx = fast_step(wx, res), cy = fast_step(wy, res);
    fast_step_z(wz);
    const bool cand = (cx || cy) && k < k1;
    push(cand, cx, cy, dzl);
    k += 1;
    dzl += dzl_step;
    // ---- emit phase
    if (qtail - qhead >= 64) emit();
  }
  while (qtail != qhead) emit();
  free_finish(a, pend);
}

// ---------------------------------------------------------------------------------------------------------
// runs per tile -> descriptor ranges, list of touched tiles
// ---------------------------------------------------------------------------------------------------------
constexpr int SCAN_TILES_PER_THREAD = 16;
constexpr int SCAN_TILES_PER_BLOCK = 256 * SCAN_TILES_PER_THREAD;
uint32_t tile_scan_blocks(int64_t n_tiles) { return (uint32_t)((n_tiles + SCAN_TILES_PER_BLOCK - 1) / SCAN_TILES_PER_BLOCK); }

struct TileScanArgs
{
  uint32_t *tile_nruns;
  uint32_t *tile_begin;
  uint8_t *tile_dirty;
  TileEntry *tile_list;
  uint32_t *block_sums; // [blocks][2] (runs, listed), then [blocks][2] their exclusive scan
  uint32_t n_blocks;
  int64_t n_tiles;
  int32_t nty, ntz;
  TsdfCounters *counters;
  // run descriptors as the tail march wrote them -> grouped by tile (placement blocks of tile_scan_kernel / desc_place_kernel)
  const RunDesc *desc;
  uint32_t desc_cap;
  uint32_t *sorted_desc;
};

__device__ __forceinline__ unsigned long long block_scan_u64(unsigned long long mine, unsigned long long *wave_sums, unsigned long long &total)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned long long incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const unsigned long long y = __shfl_up(incl, d, 64);
    if (lane >= d) incl += y;
  }
  if (lane == 63) wave_sums[wave] = incl;
  __syncthreads();
  unsigned long long excl = incl - mine;
  total = 0;
  const int n_waves = blockDim.x >> 6;
  for (int w = 0; w < n_waves; ++w)
  {
    if (w < wave) excl += wave_sums[w];
    total += wave_sums[w];
  }
  __syncthreads();
  return excl;
}

__global__ __launch_bounds__(256) void tile_count_kernel(TileScanArgs a)
{
  __shared__ unsigned long long wave_sums[4];
  const int64_t t0 = (int64_t)blockIdx.x * SCAN_TILES_PER_BLOCK + (int64_t)threadIdx.x * SCAN_TILES_PER_THREAD;
  unsigned long long mine = 0; // runs in the low word, listed tiles in the high word
  for (int j = 0; j < SCAN_TILES_PER_THREAD; ++j)
  {
    const int64_t t = t0 + j;
    if (t >= a.n_tiles) break;
    const uint32_t nr = a.tile_nruns[t];
    mine += nr;
    if (nr || a.tile_dirty[t]) mine += 1ull << 32;
  }
  unsigned long long total;
  block_scan_u64(mine, wave_sums, total);
  if (threadIdx.x == 0)
  {
    a.block_sums[2 * blockIdx.x + 0] = (uint32_t)total;
    a.block_sums[2 * blockIdx.x + 1] = (uint32_t)(total >> 32);
  }
}

__global__ __launch_bounds__(1024) void tile_blockscan_kernel(TileScanArgs a)
{
  __shared__ unsigned long long wave_sums[16];
  uint32_t *out = a.block_sums + 2 * (size_t)a.n_blocks;
  unsigned long long carry = 0;
  for (uint32_t b0 = 0; b0 < a.n_blocks; b0 += 1024)
  {
    const uint32_t b = b0 + threadIdx.x;
    unsigned long long mine = 0;
    if (b < a.n_blocks) mine = (unsigned long long)a.block_sums[2 * b] | ((unsigned long long)a.block_sums[2 * b + 1] << 32);
    unsigned long long total;
    const unsigned long long excl = block_scan_u64(mine, wave_sums, total) + carry;
    if (b < a.n_blocks)
    {
      out[2 * b + 0] = (uint32_t)excl;
      out[2 * b + 1] = (uint32_t)(excl >> 32);
    }
    carry += total;
  }
  if (threadIdx.x == 0)
  {
    a.counters->n_desc_sorted = (uint32_t)carry;
    a.counters->n_listed = (uint32_t)(carry >> 32);
  }
}

__global__ __launch_bounds__(256) void tile_list_kernel(TileScanArgs a)
{
  __shared__ unsigned long long wave_sums[4];
  const int64_t t0 = (int64_t)blockIdx.x * SCAN_TILES_PER_BLOCK + (int64_t)threadIdx.x * SCAN_TILES_PER_THREAD;
  uint32_t nr[SCAN_TILES_PER_THREAD];
  uint32_t listed = 0; // bit mask
  unsigned long long mine = 0;
#pragma unroll
  for (int j = 0; j < SCAN_TILES_PER_THREAD; ++j)
  {
    const int64_t t = t0 + j;
    nr[j] = 0;
    if (t < a.n_tiles)
    {
      nr[j] = a.tile_nruns[t];
      const bool dirty = a.tile_dirty[t] != 0;
      if (dirty) a.tile_dirty[t] = 0;
      mine += nr[j];
      if (nr[j] || dirty)
      {
        mine += 1ull << 32;
        listed |= 1u << j;
      }
    }
  }
  unsigned long long total;
  unsigned long long excl = block_scan_u64(mine, wave_sums, total);
  const uint32_t *boff = a.block_sums + 2 * (size_t)a.n_blocks;
  uint32_t run_off = (uint32_t)excl + boff[2 * blockIdx.x + 0];
  uint32_t list_off = (uint32_t)(excl >> 32) + boff[2 * blockIdx.x + 1];
#pragma unroll
  for (int j = 0; j < SCAN_TILES_PER_THREAD; ++j)
  {
    if (!(listed & (1u << j))) continue;
    const int64_t t = t0 + j;
    if (nr[j]) a.tile_begin[t] = run_off;
    TileEntry e;
    e.tile = (uint32_t)t;
    e.desc_begin = run_off;
    e.nruns = nr[j];
    e.tz = (int32_t)((uint32_t)t % (uint32_t)a.ntz);
    e.ty = (int32_t)(((uint32_t)t / (uint32_t)a.ntz) % (uint32_t)a.nty);
    e.tx = (int32_t)((uint32_t)t / ((uint32_t)a.ntz * (uint32_t)a.nty));
    e.pad[0] = e.pad[1] = 0;
    a.tile_list[list_off] = e;
    run_off += nr[j];
    list_off += 1;
  }
}

// The three kernels above as ONE launch for maps whose scan fits the chip (every workgroup resident: no workgroup waits for
// one that has not started): each workgroup publishes the (runs, listed) total of its 4096 tiles in a single 64-bit word
// — status in the top two bits, so value and flag cannot be seen apart — and looks back over its predecessors' words
// (aggregate or inclusive prefix) for its own exclusive prefix.  `look` is zero at the start of a scan (prep).
constexpr unsigned long long LOOK_AGG = 1ull << 62, LOOK_INCL = 2ull << 62, LOOK_MASK = 3ull << 62;
constexpr uint32_t LOOKBACK_MAX_BLOCKS = 2048;
__device__ __forceinline__ unsigned long long look_pack(unsigned long long v) { return (v & 0x7fffffffull) | ((v >> 32) << 31); } // runs(31) | listed(31)
__device__ __forceinline__ unsigned long long look_unpack(unsigned long long w) { return (w & 0x7fffffffull) | (((w >> 31) & 0x7fffffffull) << 32); }

// run descriptors grouped by tile: the tile's counter of runs doubles as its placement cursor (and ends at zero).
// COHERENT: called by the placement blocks of tile_scan_kernel, which read what scan blocks of the same launch wrote.
template <bool COHERENT>
__device__ __forceinline__ void place_descriptors(const TileScanArgs &a, uint32_t first_block, uint32_t n_blocks)
{
  // n_desc_sorted = sum of tile_nruns = descriptors actually written: they are the first n of the array (a reservation
  // that did not fit wrote nothing and counted nothing, and every later one failed too)
  uint32_t n = COHERENT ? __hip_atomic_load(&a.counters->n_desc_sorted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.counters->n_desc_sorted;
  if (n > a.desc_cap) n = a.desc_cap;
  // four descriptors per thread and pass, every step issued for all four before the next step uses any of them (loads,
  // then the returning atomics and the coherent loads, then the stores): three round trips per pass instead of per descriptor
  constexpr int PU = 4;
  const uint32_t stride = n_blocks * 256u;
  for (uint32_t i0 = (blockIdx.x - first_block) * 256u + threadIdx.x; i0 < n; i0 += stride * PU)
  {
    RunDesc d[PU];
    uint32_t old[PU], begin[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u)
    {
      const uint32_t i = i0 + (uint32_t)u * stride;
      d[u] = a.desc[i < n ? i : n - 1u];
    }
#pragma unroll
    for (int u = 0; u < PU; ++u)
    {
      const uint32_t i = i0 + (uint32_t)u * stride;
      old[u] = i < n ? atomicSub(&a.tile_nruns[d[u].tile], 1u) : 0u;
      begin[u] = COHERENT ? __hip_atomic_load(&a.tile_begin[d[u].tile], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : a.tile_begin[d[u].tile];
    }
#pragma unroll
    for (int u = 0; u < PU; ++u)
    {
      const uint32_t i = i0 + (uint32_t)u * stride;
      if (i >= n) continue;
      const uint32_t pos = begin[u] + old[u] - 1u;
      *reinterpret_cast<uint2 *>(&a.sorted_desc[2 * (size_t)pos]) = make_uint2(d[u].start, d[u].count);
    }
  }
}

// Blocks [0, n_scan_blocks): the scan.  Blocks behind them: the placement of the run descriptors, which needs every tile's
// range -- they wait for the scan blocks' arrival count (in-order dispatch: a placement block is only on the chip when every
// scan block is there or done), instead of a launch of their own.
__global__ __launch_bounds__(256) void tile_scan_kernel(TileScanArgs a, unsigned long long *look, uint32_t n_scan_blocks)
{
  __shared__ unsigned long long wave_sums[4];
  __shared__ unsigned long long s_prefix;
  if (blockIdx.x >= n_scan_blocks)
  {
    if (threadIdx.x == 0)
      while (__hip_atomic_load(&a.counters->scan_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_scan_blocks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    place_descriptors<true>(a, n_scan_blocks, gridDim.x - n_scan_blocks);
    return;
  }
  const int64_t t0 = (int64_t)blockIdx.x * SCAN_TILES_PER_BLOCK + (int64_t)threadIdx.x * SCAN_TILES_PER_THREAD;
  uint32_t nr[SCAN_TILES_PER_THREAD];
  uint32_t listed = 0; // bit mask
  unsigned long long mine = 0;
  static_assert(SCAN_TILES_PER_THREAD == 16, "a thread's tiles are four 128-bit loads of counters and one of dirty bytes");
  uint32_t dirty_bits = 0;
  if (t0 + SCAN_TILES_PER_THREAD <= a.n_tiles)
  {
    // five 128-bit loads issued together, then one store: tile by tile (load, load, conditional store, ...) the in-order
    // memory counter made sixteen dependent round trips out of this
    u32x4 c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = *reinterpret_cast<const u32x4 *>(&a.tile_nruns[t0 + 4 * q]);
    const u32x4 d = *reinterpret_cast<const u32x4 *>(&a.tile_dirty[t0]);
#pragma unroll
    for (int q = 0; q < 4; ++q)
    {
      nr[4 * q + 0] = c[q].x;
      nr[4 * q + 1] = c[q].y;
      nr[4 * q + 2] = c[q].z;
      nr[4 * q + 3] = c[q].w;
    }
    const uint32_t dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
    for (int j = 0; j < 16; ++j) dirty_bits |= ((dw[j >> 2] >> (8 * (j & 3))) & 0xffu) ? (1u << j) : 0u;
    if (dirty_bits)
    {
      const u32x4 z = {0, 0, 0, 0};
      *reinterpret_cast<u32x4 *>(&a.tile_dirty[t0]) = z;
    }
  }
  else
  {
#pragma unroll
    for (int j = 0; j < SCAN_TILES_PER_THREAD; ++j)
    {
      const int64_t t = t0 + j;
      nr[j] = 0;
      if (t < a.n_tiles)
      {
        nr[j] = a.tile_nruns[t];
        if (a.tile_dirty[t] != 0)
        {
          a.tile_dirty[t] = 0;
          dirty_bits |= 1u << j;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < SCAN_TILES_PER_THREAD; ++j)
  {
    mine += nr[j];
    if (nr[j] || (dirty_bits & (1u << j)))
    {
      mine += 1ull << 32;
      listed |= 1u << j;
    }
  }
  unsigned long long total;
  const unsigned long long excl = block_scan_u64(mine, wave_sums, total);
  if (threadIdx.x < 64)
  {
    // The look-back, by the whole first wave: lane i reads the word of block b - 1 - i, so 64 predecessors cost ONE round
    // trip (one thread walking back word by word paid a coherent load -- 1 to 2 us -- per predecessor: the last of the 33
    // blocks of the 513^3 map waited ~30 us).  Everything up to the nearest inclusive prefix must be published (status
    // != 0); aggregates in between are added, the inclusive prefix ends the walk.
    const uint32_t b = blockIdx.x;
    const int lane = threadIdx.x;
    unsigned long long prefix = 0;
    if (b > 0)
    {
      if (lane == 0) __hip_atomic_store(&look[b], LOOK_AGG | look_pack(total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int64_t j0 = (int64_t)b - 1;
      for (;;)
      {
        const int64_t j = j0 - lane;
        unsigned long long w = LOOK_INCL; // before block 0: an inclusive prefix of zero
        if (j >= 0) w = __hip_atomic_load(&look[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long incl = __ballot((w & LOOK_MASK) == LOOK_INCL), ready = __ballot((w & LOOK_MASK) != 0);
        const int first = incl ? __ffsll((long long)incl) - 1 : 64;                            // nearest inclusive prefix among these 64
        const unsigned long long need = first >= 63 ? ~0ull : ((2ull << first) - 1ull);          // lanes 0 .. first
        if ((ready & need) != need)
        {
          __builtin_amdgcn_s_sleep(2);
          continue; // somebody in front has not published yet: read again
        }
        unsigned long long v = lane <= first ? look_unpack(w & ~LOOK_MASK) : 0ull;
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        prefix += v;
        if (first < 64) break;
        j0 -= 64;
      }
    }
    if (lane == 0)
    {
      __hip_atomic_store(&look[b], LOOK_INCL | look_pack(prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = prefix;
      if (b == n_scan_blocks - 1)
      {
        __hip_atomic_store(&a.counters->n_desc_sorted, (uint32_t)(prefix + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.counters->n_listed = (uint32_t)((prefix + total) >> 32);
      }
    }
  }
  __syncthreads();
  uint32_t run_off = (uint32_t)excl + (uint32_t)s_prefix;
  uint32_t list_off = (uint32_t)(excl >> 32) + (uint32_t)(s_prefix >> 32);
#pragma unroll
  for (int j = 0; j < SCAN_TILES_PER_THREAD; ++j)
  {
    if (!(listed & (1u << j))) continue;
    const int64_t t = t0 + j;
    if (nr[j]) __hip_atomic_store(&a.tile_begin[t], run_off, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // read by the placement blocks
    TileEntry e;
    e.tile = (uint32_t)t;
    e.desc_begin = run_off;
    e.nruns = nr[j];
    e.tz = (int32_t)((uint32_t)t % (uint32_t)a.ntz);
    e.ty = (int32_t)(((uint32_t)t / (uint32_t)a.ntz) % (uint32_t)a.nty);
    e.tx = (int32_t)((uint32_t)t / ((uint32_t)a.ntz * (uint32_t)a.nty));
    e.pad[0] = e.pad[1] = 0;
    a.tile_list[list_off] = e;
    run_off += nr[j];
    list_off += 1;
  }
  // this block's ranges (and, from the last block, the totals) have been written through: count it
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(&a.counters->scan_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the placement as a launch of its own, behind the three-kernel scan of maps with more than LOOKBACK_MAX_BLOCKS scan blocks
__global__ __launch_bounds__(256) void desc_place_kernel(TileScanArgs a) { place_descriptors<false>(a, 0, gridDim.x); }

// ---------------------------------------------------------------------------------------------------------
// exact resolve of one tile in LDS
// ---------------------------------------------------------------------------------------------------------
struct ResolveArgs
{
  const TileEntry *tile_list;
  const uint32_t *sorted_desc;
  const CandRecord *recs;
  uint32_t *new_data;
  uint32_t *avg_data;
  uint8_t *vstate;
  const unsigned long long *fk_keys;
  const unsigned long long *fk_vals;
  int32_t fk_shift;
  uint32_t fk_mask;
  MapParams map;
  int32_t nty, ntz;
  int32_t tau, max_weight;
  uint32_t wM32;     // division by tau - tau/10 of the weight ramp (update_tsdf.cu:92) as one v_mul_hi_u32 + shift
  int32_t wS;
  uint32_t desc_cap; // entries of sorted_desc
  uint32_t *resolve_stats; // [grid][2]: contested voxels, free-space hits on keyed voxels
  TsdfCounters *counters;
  uint32_t *status;
  unsigned long long *look; // look-back words of the tile scan: consumed, zeroed here for the next scan
  uint32_t n_look;
};
static_assert(sizeof(ResolveArgs) <= 256, "ResolveArgs: more than 256 bytes of kernel arguments");

constexpr int PLACE_BLOCKS = 256; // workgroups that group the run descriptors by tile
constexpr int RESOLVE_GRID = 4096; // 1024 / 2048 / 8192 workgroups: the same 170 us, 3072: 182 (block_stats holds two words per workgroup)
constexpr uint32_t M_IDLE = 0xffffffffu, M_NONE = 0x10000u; // mstate: voxel not in the ordered rounds / no earlier negative seen

// kneg: smallest |value| wins, the LATEST candidate among equal |value| (a later equal one replaces the entry)
__device__ __forceinline__ uint64_t neg_key(uint64_t key, int32_t av, int32_t value)
{
  const uint64_t t = key >> 17;
  return ((uint64_t)av << 45) | ((T_MASK - t) << 1) | (value < 0 ? 1u : 0u);
}

template <class F>
__device__ __forceinline__ void for_each_record(uint32_t desc_begin, uint32_t nruns, const uint32_t *sorted_desc, const CandRecord *recs, F &&f)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t r = (uint32_t)wave; r < nruns; r += 4)
  {
    const uint32_t start = sorted_desc[2 * (size_t)(desc_begin + r)];
    const uint32_t count = sorted_desc[2 * (size_t)(desc_begin + r) + 1];
    for (uint32_t i = (uint32_t)lane; i < count; i += 64)
    {
      const u32x4 rec = *reinterpret_cast<const u32x4 *>(&recs[start + i]);
      const uint64_t key = (uint64_t)rec.x | ((uint64_t)rec.y << 32);
      const int32_t value = (int32_t)(int16_t)(rec.x & 0xffffu);
      f(key, value, value < 0 ? -value : value, (int)(rec.w & (TILE_VOXELS - 1)));
    }
  }
}

#ifndef WS_RES_MAXR
#define WS_RES_MAXR 8
#endif
#ifndef WS_RESOLVE_WGS
#define WS_RESOLVE_WGS 4
#endif
constexpr int RES_MAXR = WS_RES_MAXR;   // records a thread keeps in registers (2048 per tile); larger tiles re-read them per pass

// what a thread needs of a tile before it can start, requested two tiles ahead
struct TilePre
{
  int64_t idx0;
  int nz;
  uint32_t my_start, my_count; // run descriptor `lane` of the tile (nruns <= 64)
  uint32_t vs;                 // four vstate bytes
  uint32_t s0[4];              // new_map entries (HAS_S0)
};
// what stays of a list entry once its voxel bytes and descriptors have been requested
struct TileRef
{
  uint32_t nruns, desc_begin;
};
// the result of a tile, written back one tile later (behind the next tile's wait for its records)
struct TilePost
{
  int64_t idx0;
  int nz;
  uint32_t vs;
  uint32_t touched; // bit j
  uint32_t value[4];
  uint32_t existing[4];
};

// One workgroup per touched tile, thread t owns the voxels 4t .. 4t+3 of the tile (one column, four consecutive z).
// HAS_S0: new_map is not (tau, 0) — the fold starts from the stored entry (a positive weight there freezes the voxel).
// FUSED: integrate the result straight into avg_map instead of writing new_map (new_map stays (tau, 0)).
//
// Per voxel, with the candidates in canonical order: the winner is the first positive-weight candidate p with
// |v_p| <= min |v_n| over the negative-weight candidates n BEFORE p (atomic_tsdf_min accepts iff the stored weight is
// <= 0 and |new| <= |stored|, cuda/util.h:70-102); if there is none, the negative candidate of smallest |value|
// (latest on ties); else the entry stays.  LDS phases per tile:
//   pass 1   kpos = earliest positive, kneg = smallest (latest) negative            (LDS atomicMin per record)
//   scan A   m = min |v_n| over the negatives before kpos with |v_n| < |v_kpos|      (only those can block it)
//   decide   no such negative -> kpos wins; no positive -> kneg; else kpos is blocked: later positives need |v| <= m
//   scan B / decide / scan A ...  the next eligible positive, until every voxel is decided (rare after the first round)
//
// Memory pipeline.  gfx950 retires vector memory operations in order behind ONE counter (loads and stores), and the
// counts here are data dependent, so every wait is a wait for everything outstanding.  The loop therefore has a single
// such point per tile — the arrival of the tile's records — and everything else is arranged around it: entry, voxel
// bytes and descriptors of later tiles and the records of the next tile are all requested together at the END of an
// iteration, and the stores of a tile are issued right AFTER the next wait, so they drain under the LDS phases.
template <bool HAS_S0, bool FUSED>
__global__ __launch_bounds__(256, WS_RESOLVE_WGS) void tile_resolve_kernel(ResolveArgs a)
{
  __shared__ unsigned long long kpos[TILE_VOXELS];
  __shared__ unsigned long long kneg[TILE_VOXELS];
  __shared__ unsigned long long klast[TILE_VOXELS]; // the positive candidate that was blocked last
  __shared__ uint32_t mstate[TILE_VOXELS];          // M_IDLE: decided; else min |value| of the blocking negatives (M_NONE: none)
  __shared__ uint16_t bound0[HAS_S0 ? TILE_VOXELS : 1]; // |stored value| + 1 (0: frozen)
  __shared__ uint32_t s_unres[2];
  const uint32_t n_list = a.counters->n_listed;
  const int32_t weight_epsilon = a.tau / 10;
  const uint32_t reset = pack_entry(a.tau, 0);
  const int lane = threadIdx.x & 63;
  // thread t owns the voxels 4t .. 4t+3 of the tile: column t >> (ZB - 2), four consecutive z
  const int col = threadIdx.x >> (TILE_ZB - 2), lx = col >> TILE_YB, ly = col & ((1 << TILE_YB) - 1), z0 = (threadIdx.x & ((1 << (TILE_ZB - 2)) - 1)) * 4;
  const int l0 = threadIdx.x * 4;
  const uint32_t G = gridDim.x;
  uint32_t n_contested = 0, n_freehit = 0;
#ifdef WS_RESOLVE_TIMING
  long long tp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = clock64();
  int t_tiles = 0, t_rounds = 0;
#define WS_TP(i)                      \
  {                                   \
    const long long now = clock64();  \
    tp[i] += now - tl;                \
    tl = now;                         \
  }
#else
#define WS_TP(i)
#endif

  // all loads unconditional (clamped addresses, results masked)
  auto request = [&](const TileEntry &te, TilePre &p) {
    const int32_t sx = (te.tx << TILE_XB) + lx, sy = (te.ty << TILE_YB) + ly, sz = (te.tz << TILE_ZB) + z0;
    const bool col_ok = sx < a.map.size[0] && sy < a.map.size[1];
    int nz = a.map.size[2] - sz;
    p.nz = !col_ok ? 0 : (nz > 4 ? 4 : (nz < 0 ? 0 : nz));
    p.idx0 = p.nz ? storage_index(a.map, sx, sy, sz) : 0;
    const uint32_t which = (uint32_t)lane;
    const bool mine = which < te.nruns && te.nruns <= 64;
    uint32_t di = te.desc_begin + (mine ? which : 0u);
    di = di < a.desc_cap ? di : a.desc_cap - 1;
    const uint2 d = *reinterpret_cast<const uint2 *>(&a.sorted_desc[2 * (size_t)di]);
    p.my_start = mine ? d.x : 0u;
    p.my_count = mine ? d.y : 0u;
    // four voxels of a column in one access each (the arrays carry 16 bytes of slack behind the last voxel)
    const uint32_t keep = p.nz >= 4 ? 0xffffffffu : ((1u << (8 * p.nz)) - 1u);
    p.vs = 0;
    if (!HAS_S0) p.vs = *reinterpret_cast<const u32_a1 *>(a.vstate + p.idx0) & keep;
    const u32x4 z4 = {reset, reset, reset, reset};
    u32x4 s4 = z4;
    if (HAS_S0) s4 = *reinterpret_cast<const u32x4_a4 *>(a.new_data + p.idx0);
    p.s0[0] = s4.x; p.s0[1] = s4.y; p.s0[2] = s4.z; p.s0[3] = s4.w;
  };
  // weight ramp of update_tsdf.cu:90-94; 64 * (tau + value) >= 0 is below 2^31
  auto weight_of = [&](int32_t value) -> int32_t {
    int32_t w = WEIGHT_RESOLUTION;
    if (value < -weight_epsilon) w = (int32_t)(__umulhi((uint32_t)(WEIGHT_RESOLUTION * (a.tau + value)), a.wM32) >> a.wS);
    return w;
  };
  auto init_lds = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      kpos[l0 + j] = KEY_INF;
      kneg[l0 + j] = KEY_INF;
      mstate[l0 + j] = M_NONE;
    }
  };
  // The records of a tile, in registers: every lane of every wave holds descriptor `lane` of the tile (nruns <= 64), the
  // records are dealt out to the 256 threads round robin, up to RES_MAXR per thread, all loads in flight together.
  // Returns false (uniform over the workgroup) when the tile has more than 64 runs or more than 256 * RES_MAXR records:
  // the waves then stream the runs from memory in every pass (wave w the runs w, w + 4, ...).
  // The loads are UNCONDITIONAL (clamped address) and all issued before the first result is touched: a load under a branch,
  // or a use right behind it, makes the compiler wait for each of them in turn (eight serial round trips instead of one).
  // (Keeping the raw 16-byte records in registers across the decide phase, to defer that one wait as well, spills.)
  unsigned long long rkey[RES_MAXR];
  uint32_t rloc[RES_MAXR]; // voxel in the tile, 0xffffffff: no record
  u32x4 ex_next = {0, 0, 0, 0}; // FUSED: the avg_map entries of the tile whose records are in flight
  auto fetch_records = [&](const TileRef &te, const TilePre &p) -> bool {
    if (FUSED) ex_next = *reinterpret_cast<const u32x4_a4 *>(a.avg_data + p.idx0);
#pragma unroll
    for (int k = 0; k < RES_MAXR; ++k) rloc[k] = 0xffffffffu;
    if (te.nruns == 0 || te.nruns > 64) return false;
    const int nruns = (int)te.nruns;
    // the tile's runs back to back: thread t takes the records t, t + 256, ... of that sequence.  Every wave walks ALL
    // descriptors (scalar reads from its lanes), so all of them see the same total and take the same route.
    uint32_t addr[RES_MAXR];
#pragma unroll
    for (int k = 0; k < RES_MAXR; ++k) addr[k] = 0xffffffffu;
    uint32_t pref = 0; // uniform
    for (int r = 0; r < nruns; ++r)
    {
      const uint32_t start = (uint32_t)__builtin_amdgcn_readlane((int)p.my_start, r);
      const uint32_t count = (uint32_t)__builtin_amdgcn_readlane((int)p.my_count, r);
      // (A run covers one or two of the eight slots; skipping the others with uniform compares and branches removes 40 % of
      // this kernel's vector instructions and makes it SLOWER, 166 -> 184 us: with four waves per SIMD the resolve is bound
      // by the length of each wave's own instruction stream, and a taken scalar branch costs more than three selects.)
#pragma unroll
      for (int k = 0; k < RES_MAXR; ++k)
      {
        const uint32_t rel = (uint32_t)(256 * k) + threadIdx.x - pref; // wraps for positions before this run
        if (rel < count) addr[k] = start + rel;
      }
      pref += count;
    }
    if (pref > (uint32_t)(256 * RES_MAXR)) return false;
    u32x4 raw[RES_MAXR];
#pragma unroll
    for (int k = 0; k < RES_MAXR; ++k) raw[k] = *reinterpret_cast<const u32x4 *>(&a.recs[addr[k] != 0xffffffffu ? addr[k] : 0u]);
#pragma unroll
    for (int k = 0; k < RES_MAXR; ++k)
    {
      rkey[k] = (unsigned long long)raw[k].x | ((unsigned long long)raw[k].y << 32);
      rloc[k] = addr[k] != 0xffffffffu ? (raw[k].w & (TILE_VOXELS - 1)) : 0xffffffffu;
    }
    return true;
  };
  auto write_back = [&](const TilePost &w) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      if (j >= w.nz) continue;
      if (!HAS_S0 && ((w.vs >> (8 * j)) & 0xffu)) a.vstate[w.idx0 + j] = 0;
      if (!(w.touched & (1u << j))) continue;
      if (FUSED)
      {
        const uint32_t updated = integrate_entry(w.existing[j], w.value[j], a.max_weight);
        if (updated != w.existing[j]) a.avg_data[w.idx0 + j] = updated;
      }
      else
      {
        a.new_data[w.idx0 + j] = w.value[j];
      }
    }
  };

  // The per-scan scratch this scan has consumed goes back to zero here, for the next scan (no clean-up launch behind the
  // update): the look-back words of the tile scan, and the cursors of the tail march after a copy for ws_tsdf_stats.
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < a.n_look; i += gridDim.x * 256u) a.look[i] = 0;
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    TsdfCounters *c = a.counters;
    c->last_slots = c->raw_cursor;
    c->last_runs = c->desc_cursor;
    c->last_listed = n_list;
    c->raw_cursor = 0;
    c->desc_cursor = 0;
    c->scan_done = 0;
    c->tail_next = 0;
    c->ub_total = 0;
  }
  const uint32_t e0 = blockIdx.x;
  if (e0 >= n_list)
  {
    if (threadIdx.x == 0) a.resolve_stats[2 * blockIdx.x + 0] = a.resolve_stats[2 * blockIdx.x + 1] = 0;
    return;
  }
  const uint32_t last = n_list - 1;
  // pipeline: tile i is processed while the voxel bytes / descriptors of tiles i+1 and i+2, the list entries up to i+3
  // and (from the middle of the iteration on) the records of tile i+1 are in flight
  TileEntry te_n2 = a.tile_list[min(e0 + 2 * G, last)], te_n3 = te_n2;
  TileRef te_cur, te_n1;
  TilePre p_cur, p_n1, p_n2;
  {
    const TileEntry t0 = a.tile_list[e0], t1 = a.tile_list[min(e0 + G, last)];
    request(t0, p_cur);
    request(t1, p_n1);
    te_cur.nruns = t0.nruns;
    te_cur.desc_begin = t0.desc_begin;
    te_n1.nruns = t1.nruns;
    te_n1.desc_begin = t1.desc_begin;
  }
  bool cached = fetch_records(te_cur, p_cur), cached_next = false;
  u32x4 ex_cur = ex_next;
  auto issue_next = [&](uint32_t e) {
    te_n3 = a.tile_list[min(e + 3 * G, last)];
    request(te_n2, p_n2);
    cached_next = fetch_records(te_n1, p_n1);
  };
  TilePost post;
  post.nz = 0;
  post.idx0 = 0;
  post.vs = post.touched = 0;
  if (threadIdx.x == 0) s_unres[0] = s_unres[1] = 0;
  init_lds();
  __syncthreads();

  for (uint32_t e = e0; e < n_list; e += G)
  {
    const TileRef te = te_cur;
    const TilePre p = p_cur;
    const int nz = p.nz;
    const int64_t idx0 = p.idx0;

    uint32_t entry[4] = {reset, reset, reset, reset};
    uint32_t touched = 0;
    uint8_t vs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vs[j] = (uint8_t)(p.vs >> (8 * j)); // <- the wait of this iteration (with the records)
    WS_TP(0)
    write_back(post); // the previous tile's stores drain under this tile's LDS phases

    if (te.nruns == 0)
    {
      issue_next(e);
      // free space only
      if (!HAS_S0)
      {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (vs[j] & VOX_TOUCHED)
          {
            entry[j] = pack_entry(a.tau, WEIGHT_RESOLUTION);
            touched |= 1u << j;
          }
      }
    }
    else
    {
      if (HAS_S0)
      {
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
          const int32_t v0 = entry_value(p.s0[j]);
          bound0[l0 + j] = (uint16_t)((j < nz && entry_weight(p.s0[j]) <= 0) ? (v0 < 0 ? -v0 : v0) + 1 : 0);
        }
        __syncthreads();
      }
      WS_TP(1)
      bool from_regs = cached; // pass 1 and scan A; the rounds stream (the registers then hold the next tile's records)
      auto scan_records = [&](auto &&f) {
        if (from_regs)
        {
#pragma unroll
          for (int k = 0; k < RES_MAXR; ++k)
            if (rloc[k] != 0xffffffffu)
            {
              const int32_t value = (int32_t)(int16_t)(rkey[k] & 0xffffu);
              const int32_t av = value < 0 ? -value : value;
              if (HAS_S0 && av >= (int32_t)bound0[rloc[k]]) continue; // rejected by the stored entry, now and for ever
              f((uint64_t)rkey[k], value, av, (int)rloc[k]);
            }
        }
        else
        {
          for_each_record(te.desc_begin, te.nruns, a.sorted_desc, a.recs, [&](uint64_t key, int32_t value, int32_t av, int l) {
            if (HAS_S0 && av >= (int32_t)bound0[l]) return;
            f(key, value, av, l);
          });
        }
      };
      // scan A: the negatives that come before the current positive candidate and can block it
      auto scan_a = [&]() {
        scan_records([&](uint64_t key, int32_t value, int32_t av, int l) {
          if (!(key & KEY_NEG_BIT)) return;
          const uint32_t m = mstate[l];
          if (m == M_IDLE || (uint32_t)av >= m) return;
          const unsigned long long P = kpos[l];
          if (P == KEY_INF || key > P) return;
          const int32_t vp = (int32_t)(int16_t)(P & 0xffffu);
          if (av < (vp < 0 ? -vp : vp)) atomicMin(&mstate[l], (uint32_t)av);
        });
      };
      auto negative_entry = [&](unsigned long long N) {
        // no positive candidate is accepted: the negatives fold to the smallest |value|, latest on ties
        const int32_t an = (int32_t)(N >> 45);
        const int32_t v = (N & 1ull) ? -an : an;
        return pack_entry(v, -weight_of(v));
      };

      // ---- pass 1: earliest positive, smallest negative per voxel
      scan_records([&](uint64_t key, int32_t value, int32_t av, int l) {
        if (key & KEY_NEG_BIT)
          atomicMin(&kneg[l], (unsigned long long)neg_key(key, av, value));
        else
          atomicMin(&kpos[l], (unsigned long long)key);
      });
      if (!HAS_S0)
      {
        // free-space candidates that hit a keyed voxel: (tau, +weight) at their earliest order key
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (vs[j] & VOX_FREEHIT)
          {
            n_freehit += 1;
            const unsigned long long want = (unsigned long long)(idx0 + j);
            uint32_t h = fk_hash(want, a.fk_shift);
            unsigned long long t = KEY_INF;
            for (int q = 0; q < 128; ++q)
            {
              const unsigned long long cur = a.fk_keys[h];
              if (cur == want)
              {
                t = a.fk_vals[h];
                break;
              }
              if (cur == KEY_INF) break;
              h = (h + 1) & a.fk_mask;
            }
            if (t != KEY_INF)
              atomicMin(&kpos[l0 + j], (unsigned long long)record_key(t, a.tau, true));
            else
              raise_error(a.counters, a.status, ERR_INTERNAL);
          }
      }
      __syncthreads();
      WS_TP(2)
      scan_a();
      __syncthreads();
      WS_TP(3)
      // this tile's records are not needed again (unless it needs ordered rounds, which stream): everything the next
      // iterations need is requested NOW and arrives under the decide phase, the barrier and the write-back
      from_regs = false;
      issue_next(e);
      WS_TP(6)

      // ---- decide
      uint32_t unres = 0; // bit j: voxel j is still open
#pragma unroll
      for (int j = 0; j < 4; ++j)
      {
        const unsigned long long P = kpos[l0 + j], N = kneg[l0 + j];
        const uint32_t m = mstate[l0 + j];
        bool open = false;
        if (j < nz && !(P == KEY_INF && N == KEY_INF))
        {
          touched |= 1u << j;
          if (P == KEY_INF)
          {
            entry[j] = negative_entry(N);
          }
          else
          {
            const int32_t vp = (int32_t)(int16_t)(P & 0xffffu);
            const uint32_t ap = (uint32_t)(vp < 0 ? -vp : vp);
            if (N != KEY_INF && ap > (uint32_t)(N >> 45)) n_contested += 1; // a negative candidate COULD have blocked it
            if (ap <= m)
            {
              entry[j] = pack_entry(vp, weight_of(vp));
            }
            else
            {
              // blocked: every later positive candidate needs |value| <= m (which stays in mstate)
              open = true;
              unres |= 1u << j;
              klast[l0 + j] = P;
              kpos[l0 + j] = KEY_INF;
            }
          }
        }
        else if (j < nz && !HAS_S0 && (vs[j] & VOX_TOUCHED))
        {
          entry[j] = pack_entry(a.tau, WEIGHT_RESOLUTION);
          touched |= 1u << j;
        }
        if (!open)
        {
          // decided: ready for the next tile
          kpos[l0 + j] = KEY_INF;
          kneg[l0 + j] = KEY_INF;
          mstate[l0 + j] = M_NONE;
        }
      }
      if (unres) atomicAdd(&s_unres[0], 1u);
      __syncthreads();
      WS_TP(4)

      if (s_unres[0] != 0)
      {
        // ---- ordered rounds (some voxel of the tile had its earliest positive candidate blocked)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (!(unres & (1u << j))) mstate[l0 + j] = M_IDLE;
        if (threadIdx.x == 0) s_unres[1] = 0;
        __syncthreads();
        int phase = 0;
        for (;;)
        {
          // B: the next positive candidate that can still be accepted
          scan_records([&](uint64_t key, int32_t value, int32_t av, int l) {
            if (key & KEY_NEG_BIT) return;
            const uint32_t m = mstate[l];
            if (m == M_IDLE || (uint32_t)av > m) return;
            if (key > klast[l]) atomicMin(&kpos[l], (unsigned long long)key);
          });
          __syncthreads();
#pragma unroll
          for (int j = 0; j < 4; ++j)
          {
            if (!(unres & (1u << j))) continue;
            if (kpos[l0 + j] == KEY_INF)
            {
              entry[j] = negative_entry(kneg[l0 + j]);
              unres &= ~(1u << j);
              mstate[l0 + j] = M_IDLE;
            }
            else
            {
              mstate[l0 + j] = M_NONE;
            }
          }
          if (unres) atomicAdd(&s_unres[phase ^ 1], 1u);
          __syncthreads();
          if (threadIdx.x == 0) s_unres[phase] = 0;
          phase ^= 1;
#ifdef WS_RESOLVE_TIMING
          t_rounds += 1;
#endif
          if (s_unres[phase] == 0) break;
          scan_a();
          __syncthreads();
#pragma unroll
          for (int j = 0; j < 4; ++j)
          {
            if (!(unres & (1u << j))) continue;
            const unsigned long long P = kpos[l0 + j];
            const int32_t vp = (int32_t)(int16_t)(P & 0xffffu);
            const uint32_t ap = (uint32_t)(vp < 0 ? -vp : vp);
            if (ap <= mstate[l0 + j])
            {
              entry[j] = pack_entry(vp, weight_of(vp));
              unres &= ~(1u << j);
              mstate[l0 + j] = M_IDLE;
            }
            else
            {
              klast[l0 + j] = P;
              kpos[l0 + j] = KEY_INF;
            }
          }
          if (unres) atomicAdd(&s_unres[phase ^ 1], 1u);
          __syncthreads();
          if (threadIdx.x == 0) s_unres[phase] = 0;
          phase ^= 1;
          if (s_unres[phase] == 0) break;
        }
        __syncthreads();
        if (threadIdx.x == 0) s_unres[0] = s_unres[1] = 0;
        init_lds();
        __syncthreads();
      }
      WS_TP(5)
#ifdef WS_RESOLVE_TIMING
      t_tiles += 1;
#endif
    }

    // ---- this tile's result waits in registers until the next iteration's loads have arrived
    post.idx0 = idx0;
    post.nz = nz;
    post.vs = p.vs;
    post.touched = touched;
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      post.value[j] = entry[j];
    }
    post.existing[0] = ex_cur.x; post.existing[1] = ex_cur.y; post.existing[2] = ex_cur.z; post.existing[3] = ex_cur.w;
    te_cur = te_n1;
    p_cur = p_n1;
    cached = cached_next;
    ex_cur = ex_next;
    te_n1.nruns = te_n2.nruns;
    te_n1.desc_begin = te_n2.desc_begin;
    p_n1 = p_n2;
    te_n2 = te_n3;
  }
  write_back(post);
#ifdef WS_RESOLVE_TIMING
  if (threadIdx.x == 0 && (blockIdx.x & 511) == 7)
    printf("resolve wg %u: %d keyed tiles, %d rounds | wait %lld staged %lld pass1 %lld scanA %lld decide %lld rounds %lld issue %lld cycles per keyed tile\n",
           blockIdx.x, t_tiles, t_rounds, tp[0] / max(t_tiles, 1), tp[1] / max(t_tiles, 1), tp[2] / max(t_tiles, 1), tp[3] / max(t_tiles, 1),
           tp[4] / max(t_tiles, 1), tp[5] / max(t_tiles, 1), tp[6] / max(t_tiles, 1));
#endif
  // statistics: one slot per workgroup, no shared counter
  for (int d = 32; d > 0; d >>= 1)
  {
    n_contested += __shfl_down(n_contested, d, 64);
    n_freehit += __shfl_down(n_freehit, d, 64);
  }
  __shared__ uint32_t s_stat[8];
  if ((threadIdx.x & 63) == 0)
  {
    s_stat[(threadIdx.x >> 6) * 2 + 0] = n_contested;
    s_stat[(threadIdx.x >> 6) * 2 + 1] = n_freehit;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    a.resolve_stats[2 * blockIdx.x + 0] = s_stat[0] + s_stat[2] + s_stat[4] + s_stat[6];
    a.resolve_stats[2 * blockIdx.x + 1] = s_stat[1] + s_stat[3] + s_stat[5] + s_stat[7];
  }
}

// ---------------------------------------------------------------------------------------------------------
// integrate
// ---------------------------------------------------------------------------------------------------------
struct IntegrateArgs
{
  uint32_t *new_data;
  uint32_t *avg_data;
  const TileEntry *tile_list;
  MapParams map;
  int32_t nty, ntz;
  int64_t n_vox;
  int32_t max_weight;
  int32_t tau;
  TsdfCounters *counters;
};

// measured on MI355X, 513^3 (2.16 GB moved), us per launch: 3072 x 8: 393-399 (the round-2 setting), 6144 x 4: 372-380,
// 16384 x 4: 365-366 (5.9 TB/s), 12288 x 2: 380-388, 65536 x 2: 372
#ifndef DENSE_GRID
#define DENSE_GRID 16384
#endif
#ifndef DENSE_UNROLL
#define DENSE_UNROLL 4
#endif
constexpr int SPARSE_GRID = 4096;

// cu_avg_tsdf_krnl (update_tsdf.cu:13-43) over the touched tiles only: one workgroup per tile, the voxel mapping of
// tile_resolve_kernel (64-byte runs along z).
__global__ __launch_bounds__(256) void integrate_sparse_kernel(IntegrateArgs a)
{
  const uint32_t n_list = a.counters->n_listed;
  const uint32_t reset = pack_entry(a.tau, 0);
  // thread t owns the voxels 4t .. 4t+3 of the tile: column t >> (ZB - 2), four consecutive z
  const int col = threadIdx.x >> (TILE_ZB - 2), lx = col >> TILE_YB, ly = col & ((1 << TILE_YB) - 1), z0 = (threadIdx.x & ((1 << (TILE_ZB - 2)) - 1)) * 4;
  for (uint32_t e = blockIdx.x; e < n_list; e += gridDim.x)
  {
    const TileEntry te = a.tile_list[e];
    const int32_t tx = te.tx, ty = te.ty, tz = te.tz;
    const int32_t sx = (tx << TILE_XB) + lx, sy = (ty << TILE_YB) + ly, sz = (tz << TILE_ZB) + z0;
    if (sx >= a.map.size[0] || sy >= a.map.size[1]) continue;
    int nz = a.map.size[2] - sz;
    nz = nz > 4 ? 4 : nz;
    const int64_t idx0 = storage_index(a.map, sx, sy, sz);
    // both arrays in one round trip (the arrays carry 16 bytes of slack behind the last voxel)
    const u32x4 f4 = *reinterpret_cast<const u32x4_a4 *>(a.new_data + idx0);
    const u32x4 e4 = *reinterpret_cast<const u32x4_a4 *>(a.avg_data + idx0);
    const uint32_t fresh[4] = {f4.x, f4.y, f4.z, f4.w}, existing[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      if (j >= nz || fresh[j] == reset) continue;
      const uint32_t updated = integrate_entry(existing[j], fresh[j], a.max_weight);
      if (updated != existing[j]) a.avg_data[idx0 + j] = updated;
      a.new_data[idx0 + j] = reset;
    }
  }
}

// statistics of the last update, on demand (ws_tsdf_stats): the hot kernels keep per-workgroup slots, no shared counters
__global__ __launch_bounds__(256) void tsdf_stats_kernel(TsdfCounters *c, const uint32_t *tail_stats, uint32_t n_tail, const uint32_t *resolve_stats,
                                                         uint32_t n_resolve)
{
  __shared__ uint32_t part[12];
  uint32_t rec = 0, con = 0, fh = 0;
  for (uint32_t i = threadIdx.x; i < n_tail; i += 256) rec += tail_stats[i];
  for (uint32_t i = threadIdx.x; i < n_resolve; i += 256)
  {
    con += resolve_stats[2 * i + 0];
    fh += resolve_stats[2 * i + 1];
  }
  for (int d = 32; d > 0; d >>= 1)
  {
    rec += __shfl_down(rec, d, 64);
    con += __shfl_down(con, d, 64);
    fh += __shfl_down(fh, d, 64);
  }
  if ((threadIdx.x & 63) == 0)
  {
    part[(threadIdx.x >> 6) * 3 + 0] = rec;
    part[(threadIdx.x >> 6) * 3 + 1] = con;
    part[(threadIdx.x >> 6) * 3 + 2] = fh;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    c->last_records = part[0] + part[3] + part[6] + part[9];
    c->last_contested = part[1] + part[4] + part[7] + part[10];
    c->last_free_keyed = part[2] + part[5] + part[8] + part[11];
  }
}
int launch_tsdf_stats(ws_map *m)
{
  hipLaunchKernelGGL(tsdf_stats_kernel, dim3(1), dim3(256), 0, m->ctx->stream, m->counters, (const uint32_t *)m->block_stats, m->tail_blocks,
                     (const uint32_t *)(m->block_stats + WS_TAIL_STATS), m->resolve_blocks);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

// cu_avg_tsdf_krnl over EVERY voxel: the HBM-roofline stream, 16 B per voxel
// (read new + existing, write existing + reset new), 4 voxels per lane as 128-bit accesses.
__global__ __launch_bounds__(256) void integrate_dense_kernel(IntegrateArgs a)
{
  const int64_t n4 = a.n_vox >> 2;
  const uint32_t reset = pack_entry(a.tau, 0);
  const u32x4 reset4 = {reset, reset, reset, reset};
  u32x4 *new4 = reinterpret_cast<u32x4 *>(a.new_data);
  u32x4 *avg4 = reinterpret_cast<u32x4 *>(a.avg_data);
  constexpr int U = DENSE_UNROLL; // 128-bit accesses in flight per lane and array
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride)
  {
    u32x4 f[U], e[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const int64_t i = base + (int64_t)u * 256;
      if (i < n4)
      {
        f[u] = __builtin_nontemporal_load(&new4[i]);
        e[u] = __builtin_nontemporal_load(&avg4[i]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const int64_t i = base + (int64_t)u * 256;
      if (i < n4)
      {
        u32x4 r = e[u];
        // untouched voxels (new == (tau, 0)) leave avg as it is: only touched ones pay for the weighted average
        if (f[u].x != reset) r.x = integrate_entry(r.x, f[u].x, a.max_weight);
        if (f[u].y != reset) r.y = integrate_entry(r.y, f[u].y, a.max_weight);
        if (f[u].z != reset) r.z = integrate_entry(r.z, f[u].z, a.max_weight);
        if (f[u].w != reset) r.w = integrate_entry(r.w, f[u].w, a.max_weight);
        __builtin_nontemporal_store(r, &avg4[i]);
        __builtin_nontemporal_store(reset4, &new4[i]);
      }
    }
  }
  // tail (n_vox is odd for the reference's odd-sized maps)
  if (blockIdx.x == 0 && threadIdx.x < (a.n_vox & 3))
  {
    const int64_t i = (n4 << 2) + threadIdx.x;
    a.avg_data[i] = integrate_entry(a.avg_data[i], a.new_data[i], a.max_weight);
    a.new_data[i] = reset;
  }
}

__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t *dst, uint32_t v, int64_t n)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}

// ---- slabs of the ring buffer <-> a dense box (device side of the map shift) ----
// box-local order: x major, z fastest, like the maps; lo/ext in world voxel coordinates, box inside the window
template <bool PACK>
__global__ __launch_bounds__(256) void box_copy_kernel(uint32_t *map_data, MapParams mp, int32_t lox, int32_t loy, int32_t loz, int32_t ex,
                                                       int32_t ey, int32_t ez, uint32_t *box)
{
  const int64_t n = (int64_t)ex * ey * ez;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
  {
    const int32_t z = (int32_t)(i % ez);
    const int32_t y = (int32_t)((i / ez) % ey);
    const int32_t x = (int32_t)(i / ((int64_t)ez * ey));
    const int64_t idx = get_index(mp, lox + x, loy + y, loz + z);
    if (PACK)
      box[i] = map_data[idx];
    else
      map_data[idx] = box[i];
  }
}

int launch_box_copy(ws_map *m, const MapParams &par, int which, const int32_t lo[3], const int32_t ext[3], uint32_t *box_dev, bool pack, hipStream_t s)
{
  const int64_t n = (int64_t)ext[0] * ext[1] * ext[2];
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (pack)
    hipLaunchKernelGGL((box_copy_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, m->data[which], par, lo[0], lo[1], lo[2],
                       ext[0], ext[1], ext[2], box_dev);
  else
    hipLaunchKernelGGL((box_copy_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s, m->data[which], par, lo[0], lo[1], lo[2],
                       ext[0], ext[1], ext[2], box_dev);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

__global__ __launch_bounds__(256) void box_fill_kernel(uint32_t *map_data, MapParams mp, int32_t lox, int32_t loy, int32_t loz, int32_t ex, int32_t ey,
                                                       int32_t ez, uint32_t value)
{
  const int64_t n = (int64_t)ex * ey * ez;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
  {
    const int32_t z = (int32_t)(i % ez);
    const int32_t y = (int32_t)((i / ez) % ey);
    const int32_t x = (int32_t)(i / ((int64_t)ez * ey));
    map_data[get_index(mp, lox + x, loy + y, loz + z)] = value;
  }
}

int launch_box_fill(ws_map *m, const MapParams &par, int which, const int32_t lo[3], const int32_t ext[3], uint32_t value, hipStream_t s)
{
  const int64_t n = (int64_t)ext[0] * ext[1] * ext[2];
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(box_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, m->data[which], par, lo[0], lo[1], lo[2], ext[0], ext[1],
                     ext[2], value);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

int fill_u32(ws_context *ctx, uint32_t *dst, uint32_t value, int64_t n)
{
  if (n <= 0) return WS_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, dst, value, n);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

constexpr int PREP_GRID = 512;
static PrepArgs make_prep_args(ws_map *m)
{
  PrepArgs p;
  p.counters = m->counters;
  p.az_hist = m->az_hist;
  p.n_hist = (uint32_t)(AZ_BINS + 1);
  p.tile_nruns = m->tile_nruns;
  p.n_tiles = m->n_tiles;
  p.fk = m->fk_keys;
  p.n_fk = (int64_t)2 * m->fk_slots;
  p.look = reinterpret_cast<unsigned long long *>(m->block_sums);
  p.n_look = m->scan_blocks <= LOOKBACK_MAX_BLOCKS ? m->scan_blocks : 0;
  return p;
}
int launch_scatter_prep(ws_map *m)
{
  hipLaunchKernelGGL(scatter_prep_kernel, dim3(PREP_GRID), dim3(256), 0, m->ctx->stream, make_prep_args(m));
  WS_HIP(hipGetLastError());
  m->prepped = true;
  return WS_OK;
}

// fan_steps[j] of tail_bound for one resolution (host side, once per map).  j = 0 is unused.
void fill_fan_steps(int32_t *fan_steps, int32_t res)
{
  const int64_t half = res / 2 > 0 ? res / 2 : 1;
  fan_steps[0] = 0;
  for (int64_t j = 1; j < 256; ++j)
  {
    const int64_t cj = (j * res + 1) / 2;                                                   // delta_z that gives 2*delta_z/res >= j
    const int64_t Lj = (cj * MATRIX_RESOLUTION + DZ_PER_DISTANCE - 1) / DZ_PER_DISTANCE;   // first length with that delta_z
    const int64_t kj = (Lj - 1 + half - 1) / half;                                          // first step with len_k >= Lj
    fan_steps[j] = (int32_t)(kj > (1ll << 30) ? (1ll << 30) : kj);                          // beyond the 65 536 steps the order key admits anyway
  }
}

// workgroups of march_tail_kernel the device holds at once (queried once)
static unsigned tail_resident_blocks(int device)
{
  static unsigned cached = 0;
  if (cached) return cached;
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, march_tail_kernel, 256, 0) != hipSuccess || per_cu < 1) per_cu = WS_TAIL_WGS;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus < 1) cus = 256;
  cached = (unsigned)per_cu * (unsigned)cus;
  return cached;
}

int launch_tsdf_scatter(ws_map *m, const int32_t *xyz_dev, size_t n, const int32_t scanner_pos[3], const int32_t up[3], bool fused)
{
  ws_context *ctx = m->ctx;
  hipStream_t s = ctx->stream;
  m->fused_done = false;
  m->tail_blocks = 0;
  if (n == 0)
  {
    m->resolve_blocks = 0;
    WS_HIP(hipMemsetAsync(m->counters, 0, sizeof(TsdfCounters), s));
    return WS_OK; // (nothing listed: a following integrate pass has nothing to do)
  }

  const bool s0 = !m->new_is_default;
  ScatterArgs sa;
  sa.xyz = xyz_dev;
  sa.n = (uint32_t)n;
  for (int k = 0; k < 3; ++k)
  {
    sa.scanner_pos[k] = scanner_pos[k];
    sa.up[k] = up[k];
  }
  sa.map = m->par[WS_MAP_NEW];
  sa.tau = m->tau;
  sa.res = m->res;
  sa.ntx = m->ntx;
  sa.nty = m->nty;
  sa.ntz = m->ntz;
  sa.all_keyed = s0 ? 1 : 0;
  {
    // Negative-weight (off-ray) candidates only exist where iter_steps >= 2, i.e. delta_z*2 >= res
    // (update_tsdf.cu:101-102): len >= ceil(ceil(res/2) * 32768 / 100).
    const int64_t dz_min = (m->res + 1) / 2;
    const int64_t len_neg = (dz_min * MATRIX_RESOLUTION + DZ_PER_DISTANCE - 1) / DZ_PER_DISTANCE;
    sa.keyed_len_neg = (int32_t)(len_neg > INT32_MAX ? INT32_MAX : len_neg);
    sa.keyed_slack = (int32_t)(2 * (dz_min + 1) + 3 * (int64_t)m->res + 4);
  }
  sa.rays = (RaySetup *)m->rays;
  sa.az_hist = m->az_hist;
  sa.az_off = m->az_off;
  sa.ray_bin = reinterpret_cast<uint2 *>(m->ray_bin);
  sa.ray_order = m->ray_order;
  sa.fan_steps = m->fan_steps;
  sa.vstate = m->vstate;
  sa.tile_dirty = m->tile_dirty;
  sa.tile_nruns = m->tile_nruns;
  sa.rec_raw = m->rec_raw;
  sa.rec_sorted = m->rec_sorted;
  sa.rec_cap = m->rec_cap;
  sa.scan_seq = ++m->scan_seq;
  sa.desc = m->desc;
  sa.desc_cap = m->desc_cap;
  sa.fk_keys = m->fk_keys;
  {
    int l = 0;
    while ((1u << l) < m->fk_slots) ++l;
    sa.fk_shift = 64 - l;
    sa.fk_mask = m->fk_slots - 1;
  }
  sa.tail_stats = m->block_stats;
  sa.counters = m->counters;
  sa.status = m->status_dev;

  const dim3 block(256);
  const dim3 grid_setup((unsigned)((n + 255) / 256));
  const dim3 grid_tail((unsigned)((n + 63) / 64) * TAIL_SPLIT);
  const dim3 grid_free((unsigned)((n + 256 / FREE_LANES - 1) / (256 / FREE_LANES)));
  m->tail_blocks = grid_tail.x;

  const bool fuse = fused && !s0;
  for (int attempt = 0;; ++attempt)
  {
  prof_begin(ctx, WS_K_SETUP);
  // normally the kernels of the previous update have left their scratch zero / empty on their way (m->prepped)
  if (!m->prepped) hipLaunchKernelGGL(scatter_prep_kernel, dim3(PREP_GRID), block, 0, s, make_prep_args(m));
  m->prepped = false;
#if WS_FUSE_SETUP
  // set-up blocks + direction-sort blocks in one launch
  hipLaunchKernelGGL(ray_setup_sort_kernel, dim3(grid_setup.x + min(grid_setup.x, (unsigned)WS_SORT_BLOCKS)), block, 0, s, sa, grid_setup.x);
#else
  hipLaunchKernelGGL(ray_setup_sort_kernel, grid_setup, block, 0, s, sa, grid_setup.x);
  hipLaunchKernelGGL(ray_sort_kernel, dim3(min(grid_setup.x, (unsigned)WS_SORT_BLOCKS)), block, 0, s, sa);
#endif
  prof_end(ctx, WS_K_SETUP);
  prof_begin(ctx, WS_K_MARCH_TAILS);
#if WS_TAIL_PERSISTENT
  hipLaunchKernelGGL(march_tail_kernel, dim3(min(grid_tail.x, tail_resident_blocks(ctx->device))), block, 0, s, sa);
#else
  hipLaunchKernelGGL(march_tail_kernel, grid_tail, block, 0, s, sa);
#endif
  prof_end(ctx, WS_K_MARCH_TAILS);
  if (!s0)
  {
    prof_begin(ctx, WS_K_MARCH_FREE);
    hipLaunchKernelGGL(march_free_kernel, grid_free, block, 0, s, sa);
    prof_end(ctx, WS_K_MARCH_FREE);
  }

  prof_begin(ctx, WS_K_TILE_BIN);
  TileScanArgs ta;
  ta.tile_nruns = m->tile_nruns;
  ta.tile_begin = m->tile_begin;
  ta.tile_dirty = m->tile_dirty;
  ta.tile_list = m->tile_list;
  ta.block_sums = m->block_sums;
  ta.n_blocks = m->scan_blocks;
  ta.n_tiles = m->n_tiles;
  ta.nty = m->nty;
  ta.ntz = m->ntz;
  ta.counters = m->counters;
  ta.desc = m->desc;
  ta.desc_cap = m->desc_cap;
  ta.sorted_desc = m->sorted_desc;
  if (m->scan_blocks <= LOOKBACK_MAX_BLOCKS)
  {
#if WS_FUSE_PLACE
    // scan blocks + descriptor-placement blocks in one launch
    hipLaunchKernelGGL(tile_scan_kernel, dim3(m->scan_blocks + PLACE_BLOCKS), block, 0, s, ta, reinterpret_cast<unsigned long long *>(m->block_sums),
                       m->scan_blocks);
#else
    hipLaunchKernelGGL(tile_scan_kernel, dim3(m->scan_blocks), block, 0, s, ta, reinterpret_cast<unsigned long long *>(m->block_sums), m->scan_blocks);
    hipLaunchKernelGGL(desc_place_kernel, dim3(PLACE_BLOCKS), block, 0, s, ta);
#endif
  }
  else
  {
    hipLaunchKernelGGL(tile_count_kernel, dim3(m->scan_blocks), block, 0, s, ta);
    hipLaunchKernelGGL(tile_blockscan_kernel, dim3(1), dim3(1024), 0, s, ta);
    hipLaunchKernelGGL(tile_list_kernel, dim3(m->scan_blocks), block, 0, s, ta);
    hipLaunchKernelGGL(desc_place_kernel, dim3(PLACE_BLOCKS), block, 0, s, ta);
  }
  prof_end(ctx, WS_K_TILE_BIN);

  ResolveArgs ra;
  ra.tile_list = m->tile_list;
  ra.sorted_desc = m->sorted_desc;
  ra.recs = m->rec_sorted;
  ra.new_data = m->data[WS_MAP_NEW];
  ra.avg_data = m->data[WS_MAP_AVG];
  ra.vstate = m->vstate;
  ra.fk_keys = m->fk_keys;
  ra.fk_vals = m->fk_vals;
  ra.fk_shift = sa.fk_shift;
  ra.fk_mask = sa.fk_mask;
  ra.map = m->par[WS_MAP_NEW];
  ra.nty = m->nty;
  ra.ntz = m->ntz;
  ra.tau = m->tau;
  ra.max_weight = m->max_weight;
  {
    const FastDiv wd = make_fastdiv(m->tau - m->tau / 10);
    ra.wM32 = (uint32_t)wd.M;
    ra.wS = wd.k - 32;
  }
  ra.desc_cap = m->desc_cap;
  ra.resolve_stats = m->block_stats + WS_TAIL_STATS;
  ra.counters = m->counters;
  ra.status = m->status_dev;
  ra.look = reinterpret_cast<unsigned long long *>(m->block_sums);
  ra.n_look = m->scan_blocks <= LOOKBACK_MAX_BLOCKS ? m->scan_blocks : 0;
  m->resolve_blocks = RESOLVE_GRID;
  prof_begin(ctx, WS_K_TILE_RESOLVE);
  if (s0)
    hipLaunchKernelGGL((tile_resolve_kernel<true, false>), dim3(RESOLVE_GRID), block, 0, s, ra);
  else if (fuse)
    hipLaunchKernelGGL((tile_resolve_kernel<false, true>), dim3(RESOLVE_GRID), block, 0, s, ra);
  else
    hipLaunchKernelGGL((tile_resolve_kernel<false, false>), dim3(RESOLVE_GRID), block, 0, s, ra);
  prof_end(ctx, WS_K_TILE_RESOLVE);
  // The set-up pass has counted the record slots this scan can need; its last workgroup writes the total and this scan's
  // sequence number into host-mapped memory.  Everything above was enqueued WITHOUT waiting for that word (the march kernels
  // check the bound themselves and do nothing if the scan does not fit), so the device runs the update back to back; the
  // host looks at the word now -- it arrived while the later launches were being enqueued -- and, should the scan not have
  // fitted, grows the buffers and runs the update again.  (Waiting for the word BEFORE the tail march was enqueued left the
  // device idle for ~10 us per scan once the set-up pass got faster than the host's flag -> launch -> doorbell path.)  A hint
  // from the previous scan is not enough: a door that opens multiplies the need.
  {
    volatile uint32_t *st = m->status_host;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    while (st[6] != sa.scan_seq)
    {
      if ((++spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50))
      {
        WS_HIP(hipStreamSynchronize(s)); // (a stream busy with much earlier work; the word is there afterwards)
        if (st[6] != sa.scan_seq)
        {
          set_error("TSDF update: the set-up pass did not report its record bound");
          return WS_ERR_INTERNAL;
        }
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    const unsigned long long need = *reinterpret_cast<volatile unsigned long long *>(m->status_host + 4);
    if (need <= m->rec_cap) break; // the normal case
    if (attempt > 0)
    {
      set_error("TSDF update: the scan did not fit the record buffers it had just been given");
      return WS_ERR_INTERNAL;
    }
    if (need > 0xfffffff0ull)
    {
      set_error("TSDF update: the scan needs more than 2^32 candidate records");
      return WS_ERR_CAPACITY;
    }
    const int rc = resize_records(m, need + need / 8); // (waits for the stream: the skipped update has drained)
    if (rc != WS_OK) return rc;
    sa.rec_raw = m->rec_raw;
    sa.rec_sorted = m->rec_sorted;
    sa.rec_cap = m->rec_cap;
    sa.desc = m->desc;
    sa.desc_cap = m->desc_cap;
    sa.scan_seq = ++m->scan_seq;
    m->prepped = true; // the skipped update put its scratch back like any other
  }
  } // attempts
  m->fused_done = fuse;
  m->prepped = true; // every kernel above has put back what it consumed (the free-space hash: the next scan's set-up blocks)
  WS_HIP(hipGetLastError());
  return WS_OK;
}

int launch_tsdf_integrate(ws_map *m)
{
  ws_context *ctx = m->ctx;
  hipStream_t s = ctx->stream;
  IntegrateArgs ia;
  ia.new_data = m->data[WS_MAP_NEW];
  ia.avg_data = m->data[WS_MAP_AVG];
  ia.tile_list = m->tile_list;
  ia.map = m->par[WS_MAP_NEW];
  ia.nty = m->nty;
  ia.ntz = m->ntz;
  ia.n_vox = m->n_vox;
  ia.max_weight = m->max_weight;
  ia.tau = m->tau;
  ia.counters = m->counters;
  const dim3 block(256);
  if (!m->fused_done)
  {
    // a non-default new_map must be streamed completely: untouched voxels carry entries too
    const bool dense = (m->integrate_mode == WS_INTEGRATE_DENSE) || !m->new_is_default;
    prof_begin(ctx, WS_K_INTEGRATE);
    if (dense)
    {
      int64_t blocks = ((m->n_vox >> 2) + 256 * DENSE_UNROLL - 1) / (256 * DENSE_UNROLL);
      if (blocks > DENSE_GRID) blocks = DENSE_GRID;
      if (blocks < 1) blocks = 1;
      hipLaunchKernelGGL(integrate_dense_kernel, dim3((unsigned)blocks), block, 0, s, ia);
    }
    else
    {
      hipLaunchKernelGGL(integrate_sparse_kernel, dim3(SPARSE_GRID), block, 0, s, ia);
    }
    prof_end(ctx, WS_K_INTEGRATE);
  }
  WS_HIP(hipGetLastError());
  m->fused_done = false;
  m->new_is_default = true;
  return WS_OK;
}

} // namespace ws

// tsdf_update.hip — TSDF volume update for MI355X (gfx950).
//
// Replaces cu_min_tsdf_krnl + cu_avg_tsdf_krnl (src/warpsense/cuda/update_tsdf.cu:13-128) of the reference.
//
// The reference scatters with a racy CAS ("first positive-weight entry freezes the voxel",
// include/warpsense/cuda/util.h:70-102), so its result depends on thread arrival order.  This
// implementation computes the result of ONE fixed legal schedule — the serial one, ascending
// (point, ray step, fan step) — deterministically:
//
//   march<EMIT>     every candidate carries an order key t; per voxel two 64-bit atomicMin words keep
//                   kpos = earliest positive-weight candidate, kneg = smallest-|value| (latest on ties)
//                   negative-weight candidate; a byte per 64-voxel tile marks touched tiles.
//   resolve         voxels whose earliest positive candidate cannot have been blocked by any negative
//                   one (|v_pos| <= min |v_neg|) — or that only saw negatives — are final: the entry
//                   is written to new_map.  The rest are "contested" and get a list head.
//   march<COLLECT>  (only if contested voxels exist) appends every candidate of a contested voxel to a
//                   linked list in an arena.
//   resolve_lists   folds each list in ascending key order with the reference's accept rule -> new_map.
//   integrate       weighted average of new_map into avg_map and reset of new_map, either over the
//                   touched tiles only (sparse) or over every voxel (dense, the reference's kernel).
//
// new_map after resolve* is bit-identical to what the reference kernel leaves there when its threads
// run one after the other (oracle/ws_oracle.c: wso_update_min).
#include <cstddef>

#include "ws_march.h"
#include "ws_tiles.h"

namespace ws
{

struct MarchArgs
{
  const int32_t *xyz;
  uint32_t n;
  int32_t scanner_pos[3];
  int32_t up[3];
  MapParams map; // new_map's parameters (the reference indexes new_map in the scatter, update_tsdf.cu:55-125)
  int32_t tau;
  int32_t res;
  FastDiv resdiv;
  RaySetup *rays;
  uint64_t *kpos;
  uint64_t *kneg;
  uint8_t *dirty;
  uint8_t *vstate;          // one byte per voxel: VOX_KEYED / VOX_TOUCHED (split scatter)
  uint32_t *az_hist;        // [AZ_BINS + 1] rays per azimuth bin (last bin: rays that contribute nothing)
  uint32_t *az_off;         // [AZ_BINS + 2] exclusive scan of az_hist
  uint32_t *ray_order;      // ray indices sorted by azimuth bin
  int32_t keyed_len_neg;    // smallest ray length with off-ray (negative-weight) candidates
  int32_t keyed_slack;      // see keyed_first_step()
  const uint32_t *new_data; // only read when HAS_S0
  TsdfCounters *counters;
  ContestedRecord *arena;
  uint32_t arena_cap;
  uint32_t arena_slice;    // records reserved per workgroup of the collect pass
  int32_t collect_min_len; // COLLECT: steps below this length cannot reach a contested voxel
  int32_t tag_in_vstate;   // COLLECT: contested voxels are marked VOX_CONTESTED in vstate (split scatter), else by kpos
};

// Bump allocation for the lanes that reach this point together: one atomic per wave, not per lane.
// (A single shared counter hit once per lane costs ~4 ns per hit on MI355X, i.e. milliseconds per pass.)
__device__ __forceinline__ uint32_t wave_alloc(uint32_t *counter)
{
  const unsigned long long mask = __ballot(1);
  const int lane = (int)(threadIdx.x & 63);
  const int leader = __ffsll((long long)mask) - 1;
  uint32_t base = 0;
  if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
  base = (uint32_t)__shfl((int)base, leader, 64);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// Record allocation for the collect pass.  A shared bump counter costs ~10-20 ns per hit on MI355X even
// when only wave leaders touch it, so every workgroup owns a fixed slice of the arena and allocates from it
// with an LDS cursor; the shared counter (placed behind the slices) is only used when a slice overflows.
__device__ __forceinline__ uint32_t block_alloc(uint32_t *lds_cursor, uint32_t slice_base, uint32_t slice_len, uint32_t *overflow_counter,
                                                uint32_t overflow_base)
{
  const unsigned long long mask = __ballot(1);
  const int lane = (int)(threadIdx.x & 63);
  const int leader = __ffsll((long long)mask) - 1;
  const uint32_t need = (uint32_t)__popcll(mask);
  uint32_t base = 0;
  if (lane == leader)
  {
    const uint32_t c = atomicAdd(lds_cursor, need); // LDS atomic
    if (c + need <= slice_len)
      base = slice_base + c;
    else
      base = overflow_base + atomicAdd(overflow_counter, need);
  }
  base = (uint32_t)__shfl((int)base, leader, 64);
  return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

enum
{
  MARCH_EMIT = 0,       // single pass: every candidate goes through the order keys (used when new_map is not default)
  MARCH_COLLECT = 1,    // candidate lists of contested voxels
  MARCH_EMIT_KEYED = 2, // pass 1 of the split scatter: the ray tails (fan and near-surface candidates) -> order keys
  MARCH_EMIT_FREE = 3   // pass 2: the steps before the tails, all free space (tau, +64) -> one byte per voxel, no atomics
};
constexpr uint8_t VOX_KEYED = 1, VOX_TOUCHED = 2, VOX_CONTESTED = 4; // CONTESTED: between resolve and resolve_lists only
#ifndef WS_EL_BINS
#define WS_EL_BINS 8
#endif
constexpr int AZ_ONLY_BINS = 1024, EL_BINS = WS_EL_BINS;
constexpr int AZ_BINS = AZ_ONLY_BINS * EL_BINS; // direction bins: azimuth major, elevation minor

// update_tsdf.cu:52-63 for one ray per lane
__global__ __launch_bounds__(256) void ray_setup_kernel(MarchArgs a)
{
  const uint32_t ix = blockIdx.x * 256u + threadIdx.x;
  if (ix >= a.n) return;
  RaySetup r;
  r.dx = r.dy = r.dz = r.distance = r.ivx = r.ivy = r.ivz = r.steps = 0;
  r.div_m = 0;
  r.div_k = 0;
  r.pad = 0;
  const int32_t res = a.res, tau = a.tau, half = res / 2;
  const int32_t px = a.xyz[3 * (size_t)ix + 0], py = a.xyz[3 * (size_t)ix + 1], pz = a.xyz[3 * (size_t)ix + 2];
  bool ok;
  {
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
      const int64_t ndx = wmul64(dx, MR) / distance, ndy = wmul64(dy, MR) / distance, ndz = wmul64(dz, MR) / distance;
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
        ivx = wmul64(ivx, MR) / inorm;
        ivy = wmul64(ivy, MR) / inorm;
        ivz = wmul64(ivz, MR) / inorm;
        const int64_t len_end = (int64_t)distance + tau;
        const int64_t steps = (len_end - 1) / half + 1;
        const int64_t max_delta_z = (int64_t)DZ_PER_DISTANCE * len_end / MATRIX_RESOLUTION;
        const bool small_iv = ivx >= INT32_MIN && ivx <= INT32_MAX && ivy >= INT32_MIN && ivy <= INT32_MAX && ivz >= INT32_MIN && ivz <= INT32_MAX;
        if (steps > 65536 || (max_delta_z * 2) / res + 1 > 256 || !small_iv)
        {
          atomicOr(&a.counters->error, 2u); // outside the range of the order key
        }
        else
        {
          r.dx = dx; r.dy = dy; r.dz = dz;
          r.distance = distance;
          r.ivx = (int32_t)ivx; r.ivy = (int32_t)ivy; r.ivz = (int32_t)ivz;
          r.steps = (int32_t)steps;
          const FastDiv fd = make_fastdiv(distance);
          r.div_m = fd.M;
          r.div_k = fd.k;
          // conditions of march_steps_fast (ws_march.h): no int32 wrap in d*len, pos + d, voxel centres,
          // delta_z*iv and step*res*iv, nor in the squared distance to the hit point (|p - centre| <= len_end + 2 res)
          const int64_t dmax = max(max(llabs((long long)dx), llabs((long long)dy)), llabs((long long)dz));
          const int64_t pmax = max(max(llabs((long long)posx), llabs((long long)posy)), llabs((long long)posz));
          const int64_t ivmax = max(max(llabs(ivx), llabs(ivy)), llabs(ivz));
          const bool fast = dmax * len_end < (1ll << 31) && pmax + dmax + 2 * (int64_t)res + tau < (1ll << 30) &&
                            (2 * max_delta_z + res) * ivmax < (1ll << 31) &&
                            (len_end + 2 * (int64_t)res) * (len_end + 2 * (int64_t)res) < (1ll << 31);
          r.pad = fast ? 1 : 0;
        }
      }
    }
  }
  // azimuth bin of the ray (any monotone function of the direction would do: it only groups rays that lie in
  // the same vertical plane, whose voxels share z-columns and therefore cache lines); bin AZ_BINS = unused ray
  uint32_t bin = AZ_BINS;
  if (r.steps > 0)
  {
    const float az = atan2f((float)r.dy, (float)r.dx); // [-pi, pi]
    int b = (int)((az + 3.14159265f) * ((float)AZ_ONLY_BINS / 6.2831853f));
    b = b < 0 ? 0 : (b >= AZ_ONLY_BINS ? AZ_ONLY_BINS - 1 : b);
    // elevation: sin(el) = dz / distance in [-1, 1]; LiDARs use the middle of that range, so bin tan-like: clamp +-0.5
    float se = (float)r.dz / (float)r.distance;
    int e = (int)((se + 0.5f) * (float)EL_BINS);
    e = e < 0 ? 0 : (e >= EL_BINS ? EL_BINS - 1 : e);
    bin = (uint32_t)(b * EL_BINS + e);
  }
  r.pad |= (int32_t)(bin << 1);
  atomicAdd(&a.az_hist[bin], 1u);
  a.rays[ix] = r;
}

// exclusive scan of the AZ_BINS + 1 histogram entries (one workgroup), histogram reset for use as cursors
__global__ __launch_bounds__(1024) void ray_scan_kernel(uint32_t *hist, uint32_t *off)
{
  __shared__ uint32_t wave_sums[16];
  constexpr int TOTAL = AZ_BINS + 1;
  constexpr int PER = (TOTAL + 1023) / 1024;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lo = threadIdx.x * PER, hi = min(lo + PER, TOTAL);
  uint32_t v = 0;
  for (int i = lo; i < hi; ++i) v += hist[i];
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
  for (int i = lo; i < hi; ++i)
  {
    const uint32_t c = hist[i];
    off[i] = run;
    hist[i] = 0;
    run += c;
  }
  if (hi == TOTAL && lo < hi) off[TOTAL] = run; // off[AZ_BINS] = rays that contribute, off[AZ_BINS + 1] = all rays
}

__global__ __launch_bounds__(256) void ray_scatter_kernel(MarchArgs a)
{
  const uint32_t ix = blockIdx.x * 256u + threadIdx.x;
  if (ix >= a.n) return;
  const uint32_t bin = (uint32_t)a.rays[ix].pad >> 1;
  a.ray_order[a.az_off[bin] + atomicAdd(&a.az_hist[bin], 1u)] = ix;
}

// 8 rays per workgroup, 32 lanes per ray: lane c walks the steps [c*CH, (c+1)*CH) of its ray
// (CH = ceil(steps/32)), so every lane has the same amount of work whatever the ray length, and a scan
// of 131 072 rays puts 4 M lanes in flight instead of 131 072 (update_tsdf.cu:67-125 per step).
template <int MODE, bool HAS_S0>
__global__ __launch_bounds__(256) void march_kernel(MarchArgs a)
{
  __shared__ uint32_t record_cursor;
  if (MODE == MARCH_COLLECT)
  {
    if (threadIdx.x == 0) record_cursor = 0;
    __syncthreads();
  }
  if (MODE == MARCH_COLLECT)
  {
    if (__hip_atomic_load(&a.counters->contested, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  }
  // Full-ray passes: 8 rays x 32 lanes per workgroup (see above).  Tail passes (ordered candidates, contested
  // lists) do scattered 8-byte atomics near the surface: there a wave takes 64 rays of the same azimuth bin
  // (ray_order) at the same quarter of the tail, so its lanes hit voxels of the same vertical plane — z-neighbours,
  // i.e. the same cache lines — and one wave-wide atomic touches a handful of lines instead of 64.
  constexpr bool TAIL = (MODE == MARCH_EMIT_KEYED); // (the collect pass is slower with this mapping: 308 vs 242 us)
#ifndef WS_FULL_LANES
#define WS_FULL_LANES 32
#endif
#ifndef WS_COLLECT_LANES
#define WS_COLLECT_LANES 32
#endif
  constexpr int FULL_LANES = MODE == MARCH_COLLECT ? WS_COLLECT_LANES : WS_FULL_LANES;
  constexpr int LANES = TAIL ? 4 : FULL_LANES;
  constexpr int RAYS_PER_BLOCK = 256 / FULL_LANES;
  uint32_t ix;
  int32_t c;
  if (TAIL)
  {
    const uint32_t slot = blockIdx.x * 64u + (threadIdx.x & 63u);
    if (slot >= a.az_off[AZ_BINS]) return;
    ix = a.ray_order[slot];
    c = (int32_t)(threadIdx.x >> 6);
  }
  else
  {
    ix = blockIdx.x * (uint32_t)RAYS_PER_BLOCK + (threadIdx.x / (uint32_t)FULL_LANES);
    if (ix >= a.n) return;
    c = (int32_t)(threadIdx.x % (uint32_t)FULL_LANES);
  }
  const RaySetup r = a.rays[ix];
  if (r.steps == 0) return;
  const int32_t res = a.res, tau = a.tau;
  const int32_t half = res / 2;
  // COLLECT only needs the steps with len = 1 + k*half >= collect_min_len
  int32_t kbeg = 0;
  if (MODE == MARCH_COLLECT && a.collect_min_len > 1) kbeg = (a.collect_min_len - 1 + half - 1) / half;
  // Split scatter: a candidate is "free space" iff it is on-ray (positive weight) with value == +tau; all of a
  // ray's candidates are of that kind while len < min(len_neg, distance - tau - slack): before len_neg there is
  // no fan, and a voxel centre further than tau from the hit point gives min(dist, tau) == tau.  The slack covers
  // |centre - proj| (1.5 voxels per axis for the double-width cell of trunc division + the fan offset).
  const int32_t keyed_len = min(a.keyed_len_neg, r.distance - tau - a.keyed_slack);
  const int32_t keyed_first = keyed_len > 1 ? max(0, (keyed_len - 1) / half - 1) : 0;
  // pass 1 (keyed) owns the steps from keyed_first on, pass 2 (free) the steps before it
  int32_t kend = r.steps;
  if (MODE == MARCH_EMIT_KEYED) kbeg = keyed_first;
  if (MODE == MARCH_EMIT_FREE) kend = min(r.steps, keyed_first);
  if (kbeg >= kend) return;
  const int32_t ch = (kend - kbeg + LANES - 1) / LANES;
  const int32_t k0 = kbeg + c * ch;
  const int32_t k1 = min(k0 + ch, kend);
  if (k0 >= k1) return;

  const MarchFrame f = make_march_frame(a.scanner_pos, res, tau, a.map);
  march_steps<MODE == MARCH_EMIT_FREE>(f, r, k0, k1, [&](int32_t k, int32_t step, int32_t vx, int32_t vy, int32_t vz, int32_t value, bool positive) {
    const int64_t idx = get_index(a.map, vx, vy, vz);
    const uint64_t t = order_key(ix, k, step);
    if (HAS_S0)
    {
      // candidates the initial new_map entry would reject can never be accepted later either
      // (the stored |value| only shrinks and a positive weight freezes the voxel): drop them here
      const uint32_t s0 = a.new_data[idx];
      const int32_t a0 = entry_value(s0) < 0 ? -entry_value(s0) : entry_value(s0);
      if (entry_weight(s0) > 0 || (value < 0 ? -value : value) > a0) return;
    }
    if (MODE == MARCH_EMIT_FREE)
    {
      // every candidate of these steps is free space: on the ray, further than tau from the hit point
      if (!(positive && value == tau))
      {
        atomicOr(&a.counters->error, 4u); // impossible by the bound above; never lose a candidate silently
        return;
      }
      const uint8_t b = a.vstate[idx];
      if (b & VOX_KEYED)
      {
        // the voxel also has ordered candidates (from pass 1): this one takes part in the key order
        const uint64_t key = make_kpos(t, value);
        if (key < a.kpos[idx]) atomicMin((unsigned long long *)&a.kpos[idx], (unsigned long long)key);
      }
      else if (b == 0)
      {
        // free space only (the common case): the result will be (tau, 64) whoever comes first
        a.vstate[idx] = VOX_TOUCHED;
        const int64_t tile = idx >> TILE_SHIFT;
        if (a.dirty[tile] == 0) a.dirty[tile] = 1;
      }
      return;
    }
    if (MODE == MARCH_EMIT_KEYED)
    {
      // Ray tails: three fire-and-forget operations per candidate, nothing the lane has to wait for.  Measured on
      // MI355X (tools/keyed_exp.sh): reading vstate / the key first to skip redundant stores and atomics makes
      // every step wait for a scattered load (610 us for the pass); unconditional stores + atomicMin 395 us, of
      // which the 7.5 M scattered 64-bit atomics are 390 (~19 G atomics/s, the same at workgroup and agent scope)
      // and the march arithmetic 153.  The tails are spread over the surfaces, so the byte stores do not pile up
      // on one address the way they would near the sensor (the full-ray EMIT pass below keeps its pre-read).
      a.vstate[idx] = VOX_KEYED;
      a.dirty[idx >> TILE_SHIFT] = 1;
      if (positive)
        atomicMin((unsigned long long *)&a.kpos[idx], (unsigned long long)make_kpos(t, value));
      else
        atomicMin((unsigned long long *)&a.kneg[idx], (unsigned long long)make_kneg(t, value));
      return;
    }
    if (MODE == MARCH_EMIT)
    {
      // A lane that still reads the "never touched" pattern marks the 64-voxel tile (plain byte store, every
      // writer stores the same value).  Only the first toucher(s) of a voxel get here, so the stores do not
      // pile up on one byte the way an unconditional mark does near the sensor (measured: +0.8 ms), and the
      // atomics stay non-returning (a returning atomicMin stalls the lane for the memory round trip).
      if (positive)
      {
        const uint64_t key = make_kpos(t, value);
        const uint64_t cur = a.kpos[idx];
        if (cur == KEY_INF) a.dirty[idx >> TILE_SHIFT] = 1;
        if (key < cur) atomicMin((unsigned long long *)&a.kpos[idx], (unsigned long long)key);
      }
      else
      {
        const uint64_t key = make_kneg(t, value);
        const uint64_t cur = a.kneg[idx];
        if (cur == KEY_INF) a.dirty[idx >> TILE_SHIFT] = 1;
        if (key < cur) atomicMin((unsigned long long *)&a.kneg[idx], (unsigned long long)key);
      }
    }
    else
    {
      // one byte per candidate where the split scatter keeps them (128 voxels per line instead of 16)
      const bool tagged = a.tag_in_vstate ? a.vstate[idx] == VOX_CONTESTED : a.kpos[idx] == KEY_CONTESTED_TAG;
      if (tagged)
      {
        const uint32_t rec = block_alloc(&record_cursor, blockIdx.x * a.arena_slice, a.arena_slice, &a.counters->records,
                                         gridDim.x * a.arena_slice);
        if (rec < a.arena_cap)
        {
          ContestedRecord cr;
          cr.key = (t << 17) | (positive ? 0ull : (1ull << 16)) | ((uint32_t)value & 0xffffu);
          // the list head of a contested voxel lives in the low half of its (now unused) kneg word
          cr.next = atomicExch(reinterpret_cast<uint32_t *>(&a.kneg[idx]), rec);
          cr.pad = 0;
          a.arena[rec] = cr;
        }
        else
        {
          atomicOr(&a.counters->error, 1u);
        }
      }
    }
  });
}

struct ResolveArgs
{
  uint64_t *kpos;
  uint64_t *kneg;
  const uint32_t *dirty_list;
  uint32_t *new_data;
  int64_t n_vox;
  int32_t tau;
  TsdfCounters *counters;
  uint8_t *vstate;
  int32_t split;                // the scatter used the keyed/free-space split
  uint32_t *contested_per_wave; // [LIST_GRID_BLOCKS * 4]
  const ContestedRecord *arena;
  uint32_t arena_cap;
};

// touched-tile flags -> list of tile ids (order irrelevant); clears the flags it consumes.
// Each workgroup owns a contiguous range of tiles: it counts its flags, reserves its part of the list with
// ONE atomic, then writes the ids (a shared counter hit once per wave was 226 us of this pass).
constexpr int COMPACT_BLOCKS = 256;
__global__ __launch_bounds__(256) void compact_dirty_kernel(uint8_t *dirty, int64_t n_tiles, uint32_t *list, TsdfCounters *counters)
{
  __shared__ uint32_t wave_total[4];
  __shared__ uint32_t block_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t per_block = ((n_tiles + COMPACT_BLOCKS - 1) / COMPACT_BLOCKS + 255) & ~255ll;
  const int64_t lo = (int64_t)blockIdx.x * per_block;
  const int64_t hi = lo + per_block < n_tiles ? lo + per_block : n_tiles;
  // pass 1: count
  uint32_t cnt = 0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += 256) cnt += dirty[i] != 0 ? 1u : 0u;
  for (int d = 32; d > 0; d >>= 1) cnt += __shfl_down(cnt, d, 64);
  if (lane == 0) wave_total[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    const uint32_t total = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
    block_base = total ? atomicAdd(&counters->dirty_tiles, total) : 0u;
  }
  __syncthreads();
  // pass 2: write ids; running offset = block_base + flags seen so far in this workgroup
  uint32_t running = block_base;
  for (int64_t i0 = lo; i0 < hi; i0 += 256)
  {
    const int64_t i = i0 + threadIdx.x;
    const bool flag = i < hi && dirty[i] != 0;
    const unsigned long long mask = __ballot(flag);
    if (lane == 0) wave_total[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += wave_total[w];
    const uint32_t chunk_total = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
    if (flag)
    {
      list[running + before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint32_t)i;
      dirty[i] = 0;
    }
    running += chunk_total;
    __syncthreads();
  }
}

#ifndef DENSE_GRID
#define DENSE_GRID 3072
#endif
#ifndef DENSE_UNROLL
#define DENSE_UNROLL 8
#endif
constexpr int LIST_GRID_BLOCKS = 2048; // persistent grid over the touched-tile list: 8192 waves
#ifndef SPARSE_UNROLL
#define SPARSE_UNROLL 4
#endif

// One wave per touched 64-voxel tile (one lane per voxel), waves stride over the tile list.
__global__ __launch_bounds__(256) void resolve_kernel(ResolveArgs a)
{
  const int lane = threadIdx.x & 63;
  const uint32_t n_dirty = __hip_atomic_load(&a.counters->dirty_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int32_t weight_epsilon = a.tau / 10;
  int n_contested = 0;
  for (uint32_t i = blockIdx.x * 4u + (threadIdx.x >> 6); i < n_dirty; i += gridDim.x * 4u)
  {
    const int64_t idx = ((int64_t)a.dirty_list[i] << TILE_SHIFT) + lane;
    if (idx >= a.n_vox) continue;
    if (a.split)
    {
      // split scatter: one byte says whether the voxel was touched at all and whether it has order keys
      const uint8_t b = a.vstate[idx];
      if (b == 0) continue;
      a.vstate[idx] = 0;
      if (!(b & VOX_KEYED))
      {
        a.new_data[idx] = pack_entry(a.tau, WEIGHT_RESOLUTION); // free-space candidates only
        continue;
      }
    }
    const uint64_t kp = a.kpos[idx], kn = a.kneg[idx];
    if (kp == KEY_INF && kn == KEY_INF) continue; // untouched voxel: new_map keeps its entry
    bool decided = true;
    int32_t value = 0;
    bool positive = false;
    if (kp != KEY_INF)
    {
      if (kp == KEY_CONTESTED_TAG) continue; // already handed to the ordered fallback
      value = (int32_t)(int16_t)(kp & 0xffffu);
      positive = true;
      if (kn != KEY_INF)
      {
        const int32_t ap = value < 0 ? -value : value;
        const int32_t an = (int32_t)(kn >> 45);
        // a negative candidate with a smaller |value| MAY have come first and blocked it: ordered fallback
        if (ap > an) decided = false;
      }
    }
    else
    {
      const int32_t an = (int32_t)(kn >> 45);
      value = (kn & 1ull) ? -an : an;
    }
    if (decided)
    {
      const int32_t w = tsdf_weight(value, a.tau, weight_epsilon);
      a.new_data[idx] = pack_entry(value, positive ? w : -w);
      a.kpos[idx] = KEY_INF;
      if (kn != KEY_INF) a.kneg[idx] = KEY_INF;
    }
    else
    {
      // contested: tag the voxel; its kneg word becomes the (empty) head of the candidate list
      a.kpos[idx] = KEY_CONTESTED_TAG;
      a.kneg[idx] = 0xffffffffull;
      if (a.split) a.vstate[idx] = VOX_CONTESTED;
      n_contested += 1;
    }
  }
  // no shared counter (even one atomic per wave on a single address costs ~100 us here): a wave that saw a
  // contested voxel raises the flag with a plain store, the exact count is kept per wave for the statistics
  for (int d = 32; d > 0; d >>= 1) n_contested += __shfl_down(n_contested, d, 64);
  if (lane == 0)
  {
    a.contested_per_wave[blockIdx.x * 4u + (threadIdx.x >> 6)] = (uint32_t)n_contested;
    if (n_contested) a.counters->contested = 1;
  }
}

// Ordered fallback: walk the touched tiles again, one lane per voxel; a lane whose voxel is tagged folds
// that voxel's candidate list in ascending key order with the accept rule of atomic_tsdf_min
// (cuda/util.h:70-102): accept iff stored weight <= 0 and |new| <= |stored|.
template <bool HAS_S0>
__global__ __launch_bounds__(256) void resolve_lists_kernel(ResolveArgs a)
{
  if (__hip_atomic_load(&a.counters->contested, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  const int lane = threadIdx.x & 63;
  const uint32_t n_dirty = __hip_atomic_load(&a.counters->dirty_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int32_t weight_epsilon = a.tau / 10;
  constexpr int U = SPARSE_UNROLL; // tiles whose tags are fetched together (most tiles have no contested voxel)
  const uint32_t stride = gridDim.x * 4u;
  for (uint32_t i0 = blockIdx.x * 4u + (threadIdx.x >> 6); i0 < n_dirty; i0 += stride * U)
  {
    int64_t idxs[U];
    bool tagged[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const uint32_t i = i0 + (uint32_t)u * stride;
      const bool ok = i < n_dirty;
      idxs[u] = ok ? (((int64_t)a.dirty_list[i] << TILE_SHIFT) + lane) : a.n_vox;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      tagged[u] = false;
      if (idxs[u] < a.n_vox) tagged[u] = a.split ? (a.vstate[idxs[u]] == VOX_CONTESTED) : (a.kpos[idxs[u]] == KEY_CONTESTED_TAG);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
    if (!tagged[u]) continue;
    const int64_t idx = idxs[u];
    if (a.split) a.vstate[idx] = 0;
    const uint32_t head = (uint32_t)(a.kneg[idx] & 0xffffffffull);
    const uint32_t state = HAS_S0 ? a.new_data[idx] : pack_entry(a.tau, 0);
    int32_t sv = entry_value(state), sw = entry_weight(state);
    int32_t sa = sv < 0 ? -sv : sv;
    uint64_t last = 0;
    bool first = true;
    while (sw <= 0)
    {
      // next record in key order (lists are short: a handful of rays reach a far voxel)
      uint64_t best = KEY_INF;
      for (uint32_t r = head; r != 0xffffffffu && r < a.arena_cap; r = a.arena[r].next)
      {
        const uint64_t k = a.arena[r].key;
        if ((first || k > last) && k < best) best = k;
      }
      if (best == KEY_INF) break;
      first = false;
      last = best;
      const int32_t v = (int32_t)(int16_t)(best & 0xffffu);
      const int32_t av = v < 0 ? -v : v;
      if (av <= sa)
      {
        const int32_t w = tsdf_weight(v, a.tau, weight_epsilon);
        sv = v;
        sa = av;
        sw = (best & (1ull << 16)) ? -w : w;
      }
    }
    a.new_data[idx] = pack_entry(sv, sw);
    a.kpos[idx] = KEY_INF;
    a.kneg[idx] = KEY_INF;
    }
  }
}

struct IntegrateArgs
{
  uint32_t *new_data;
  uint32_t *avg_data;
  const uint32_t *dirty_list;
  int64_t n_vox;
  int32_t max_weight;
  int32_t tau;
  TsdfCounters *counters;
};

// cu_avg_tsdf_krnl (update_tsdf.cu:13-43) over the touched tiles only: one wave per tile.
__global__ __launch_bounds__(256) void integrate_sparse_kernel(IntegrateArgs a)
{
  // One tile per wave and trip was latency bound (a chain of three dependent loads per 256 B: tile id, new, avg):
  // every wave works on SPARSE_UNROLL tiles at a time, all their loads are in flight before the first is used.
  constexpr int U = SPARSE_UNROLL;
  const int lane = threadIdx.x & 63;
  const uint32_t n_dirty = __hip_atomic_load(&a.counters->dirty_tiles, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint32_t reset = pack_entry(a.tau, 0);
  const uint32_t stride = gridDim.x * 4u;
  for (uint32_t i0 = blockIdx.x * 4u + (threadIdx.x >> 6); i0 < n_dirty; i0 += stride * U)
  {
    int64_t idx[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const uint32_t i = i0 + (uint32_t)u * stride;
      ok[u] = i < n_dirty;
      idx[u] = ok[u] ? (((int64_t)a.dirty_list[i] << TILE_SHIFT) + lane) : 0;
      ok[u] = ok[u] && idx[u] < a.n_vox;
    }
    uint32_t fresh[U], existing[U];
#pragma unroll
    for (int u = 0; u < U; ++u) fresh[u] = ok[u] ? a.new_data[idx[u]] : reset;
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      ok[u] = ok[u] && fresh[u] != reset;
      existing[u] = ok[u] ? a.avg_data[idx[u]] : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      if (!ok[u]) continue;
      const uint32_t updated = integrate_entry(existing[u], fresh[u], a.max_weight);
      if (updated != existing[u]) a.avg_data[idx[u]] = updated;
      a.new_data[idx[u]] = reset;
    }
  }
}

// bookkeeping after an integrate pass: remember how many tiles were streamed, restart the list
__global__ __launch_bounds__(256) void finish_update_kernel(TsdfCounters *c, const uint32_t *contested_per_wave, int n_waves)
{
  __shared__ uint32_t part[4];
  uint32_t s = 0;
  for (int i = threadIdx.x; i < n_waves; i += 256) s += contested_per_wave[i];
  for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    c->last_contested = part[0] + part[1] + part[2] + part[3];
    c->last_dirty_tiles = c->dirty_tiles;
    c->dirty_tiles = 0;
  }
}

// cu_avg_tsdf_krnl over EVERY voxel: the HBM-roofline stream, 16 B per voxel
// (read new + existing, write existing + reset new), 4 voxels per lane as 128-bit accesses.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

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
__global__ __launch_bounds__(256) void fill_u64_kernel(uint64_t *dst, uint64_t v, int64_t n)
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

int launch_box_copy(ws_map *m, int which, const int32_t lo[3], const int32_t ext[3], uint32_t *box_dev, bool pack)
{
  const int64_t n = (int64_t)ext[0] * ext[1] * ext[2];
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (pack)
    hipLaunchKernelGGL((box_copy_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, m->ctx->stream, m->data[which], m->par[which], lo[0],
                       lo[1], lo[2], ext[0], ext[1], ext[2], box_dev);
  else
    hipLaunchKernelGGL((box_copy_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, m->ctx->stream, m->data[which], m->par[which], lo[0],
                       lo[1], lo[2], ext[0], ext[1], ext[2], box_dev);
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
int fill_u64(ws_context *ctx, uint64_t *dst, uint64_t value, int64_t n)
{
  if (n <= 0) return WS_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, dst, value, n);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

int launch_tsdf_scatter(ws_map *m, const int32_t *xyz_dev, size_t n, const int32_t scanner_pos[3], const int32_t up[3], bool fused)
{
  ws_context *ctx = m->ctx;
  hipStream_t s = ctx->stream;
  // per-scatter counters (the touched-tile list survives until the integrate pass consumes it)
  WS_HIP(hipMemsetAsync(m->counters, 0, offsetof(TsdfCounters, dirty_tiles), s));
  if (n == 0) return WS_OK;

  MarchArgs ma;
  ma.xyz = xyz_dev;
  ma.n = (uint32_t)n;
  for (int k = 0; k < 3; ++k)
  {
    ma.scanner_pos[k] = scanner_pos[k];
    ma.up[k] = up[k];
  }
  ma.map = m->par[WS_MAP_NEW];
  ma.tau = m->tau;
  ma.res = m->res;
  ma.resdiv = make_fastdiv(m->res);
  ma.rays = (RaySetup *)m->rays;
  ma.kpos = m->kpos;
  ma.kneg = m->kneg;
  ma.dirty = m->dirty;
  ma.vstate = m->vstate;
  ma.az_hist = m->az_hist;
  ma.az_off = m->az_off;
  ma.ray_order = m->ray_order;
  ma.new_data = m->data[WS_MAP_NEW];
  ma.counters = m->counters;
  ma.arena = m->arena;
  ma.arena_cap = m->arena_cap;
  {
    // half of the arena is split evenly between the workgroups, the other half is the shared overflow area
    const uint32_t blocks = (uint32_t)((n + 256 / WS_COLLECT_LANES - 1) / (256 / WS_COLLECT_LANES)); // workgroups of the collect pass
    ma.arena_slice = (m->arena_cap / 2) / (blocks ? blocks : 1);
  }
  {
    // Negative-weight (off-ray) candidates only exist where iter_steps >= 2, i.e. delta_z*2 >= res
    // (update_tsdf.cu:101-102): len >= ceil(ceil(res/2) * 32768 / 100).  A contested voxel holds such a
    // candidate, and every other candidate of the same voxel has a ray length within one voxel
    // diagonal + fan of it, so the collect pass can start 4 voxels below that length.
    const int64_t dz_min = (m->res + 1) / 2;
    const int64_t len_neg = (dz_min * MATRIX_RESOLUTION + DZ_PER_DISTANCE - 1) / DZ_PER_DISTANCE;
    const int64_t lo = len_neg - 4 * (int64_t)m->res - 2 * dz_min;
    ma.collect_min_len = lo > 1 ? (int32_t)(lo > INT32_MAX ? INT32_MAX : lo) : 1;
    ma.keyed_len_neg = (int32_t)(len_neg > INT32_MAX ? INT32_MAX : len_neg);
    ma.keyed_slack = (int32_t)(2 * (dz_min + 1) + 3 * (int64_t)m->res + 4);
  }

  ResolveArgs ra;
  ra.kpos = m->kpos;
  ra.kneg = m->kneg;
  ra.dirty_list = m->dirty_list;
  ra.new_data = m->data[WS_MAP_NEW];
  ra.n_vox = m->n_vox;
  ra.tau = m->tau;
  ra.counters = m->counters;
  ra.vstate = m->vstate;
  ra.split = (m->scatter_mode != WS_SCATTER_TILES && m->new_is_default) ? 1 : 0;
  ma.tag_in_vstate = ra.split;
  ra.contested_per_wave = m->contested_per_wave;
  ra.arena = m->arena;
  ra.arena_cap = m->arena_cap;

  const dim3 block(256);
  const dim3 grid_setup((unsigned)((n + 255) / 256));
  constexpr size_t rays_per_block = 256 / WS_FULL_LANES;
  const dim3 grid_rays((unsigned)((n + rays_per_block - 1) / rays_per_block));
  const dim3 grid_tail((unsigned)((n + 63) / 64));
  constexpr size_t rays_per_collect_block = 256 / WS_COLLECT_LANES;
  const dim3 grid_collect((unsigned)((n + rays_per_collect_block - 1) / rays_per_collect_block));
  const dim3 grid_list(LIST_GRID_BLOCKS);
  const bool s0 = !m->new_is_default;

  // LDS-tile path: needs new_map == (tau, 0) (its local resolve starts every voxel from that state)
  const bool tiles = m->scatter_mode == WS_SCATTER_TILES && !s0;
  WS_HIP(hipMemsetAsync(m->az_hist, 0, (AZ_BINS + 1) * sizeof(uint32_t), s)); // the scatter pass leaves its cursors there
  hipLaunchKernelGGL(ray_setup_kernel, grid_setup, block, 0, s, ma);
  hipLaunchKernelGGL(ray_scan_kernel, dim3(1), dim3(1024), 0, s, m->az_hist, m->az_off);
  hipLaunchKernelGGL(ray_scatter_kernel, grid_setup, block, 0, s, ma);
  if (tiles)
  {
    TileArgs ta;
    ta.rays = ma.rays;
    ta.n = ma.n;
    ta.frame = make_march_frame(ma.scanner_pos, m->res, m->tau, ma.map);
    ta.grid = make_tile_grid(ma.map);
    ta.n_tiles = m->n_tiles3d;
    ta.tile_count = m->tile_count;
    ta.tile_offset = m->tile_offset;
    ta.tile_cursor = m->tile_cursor;
    ta.records = m->tile_records;
    ta.records_cap = m->tile_records_cap;
    ta.work = (uint4 *)m->tile_work;
    ta.work_cap = m->tile_work_cap;
    ta.tile_state = (TileState *)m->tile_state;
    ta.kpos = m->kpos;
    ta.kneg = m->kneg;
    ta.dirty = m->dirty;
    ta.new_data = m->data[WS_MAP_NEW];
    ta.avg_data = m->data[WS_MAP_AVG];
    ta.max_weight = m->max_weight;
    ta.counters = m->counters;
    WS_HIP(hipMemsetAsync(m->tile_state, 0, sizeof(TileState), s));
    int rc = launch_tile_path(m, ta, n, fused);
    if (rc != WS_OK) return rc;
  }
  else
  {
    prof_begin(ctx, WS_K_MARCH_EMIT);
    if (s0)
    {
      hipLaunchKernelGGL((march_kernel<MARCH_EMIT, true>), grid_rays, block, 0, s, ma);
    }
    else
    {
      hipLaunchKernelGGL((march_kernel<MARCH_EMIT_KEYED, false>), grid_tail, block, 0, s, ma);
      hipLaunchKernelGGL((march_kernel<MARCH_EMIT_FREE, false>), grid_rays, block, 0, s, ma);
    }
    prof_end(ctx, WS_K_MARCH_EMIT);
  }

  prof_begin(ctx, WS_K_RESOLVE);
  {
    hipLaunchKernelGGL(compact_dirty_kernel, dim3(COMPACT_BLOCKS), block, 0, s, m->dirty, m->n_tiles, m->dirty_list, m->counters);
  }
  hipLaunchKernelGGL(resolve_kernel, grid_list, block, 0, s, ra);
  prof_end(ctx, WS_K_RESOLVE);

  prof_begin(ctx, WS_K_MARCH_COLLECT);
  if (s0)
    hipLaunchKernelGGL((march_kernel<MARCH_COLLECT, true>), grid_collect, block, 0, s, ma);
  else
    hipLaunchKernelGGL((march_kernel<MARCH_COLLECT, false>), grid_collect, block, 0, s, ma);
  prof_end(ctx, WS_K_MARCH_COLLECT);

  prof_begin(ctx, WS_K_RESOLVE_LISTS);
  if (s0)
    hipLaunchKernelGGL((resolve_lists_kernel<true>), grid_list, block, 0, s, ra);
  else
    hipLaunchKernelGGL((resolve_lists_kernel<false>), grid_list, block, 0, s, ra);
  prof_end(ctx, WS_K_RESOLVE_LISTS);
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
  ia.dirty_list = m->dirty_list;
  ia.n_vox = m->n_vox;
  ia.max_weight = m->max_weight;
  ia.tau = m->tau;
  ia.counters = m->counters;
  const dim3 block(256);
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
    hipLaunchKernelGGL(integrate_sparse_kernel, dim3(LIST_GRID_BLOCKS), block, 0, s, ia);
  }
  prof_end(ctx, WS_K_INTEGRATE);
  hipLaunchKernelGGL(finish_update_kernel, dim3(1), dim3(256), 0, s, m->counters, (const uint32_t *)m->contested_per_wave, LIST_GRID_BLOCKS * 4);
  WS_HIP(hipGetLastError());
  m->new_is_default = true;
  return WS_OK;
}

} // namespace ws

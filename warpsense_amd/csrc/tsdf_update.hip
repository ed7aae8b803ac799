// tsdf_update.hip — TSDF volume update for MI355X (gfx950).
//
// Replaces cu_min_tsdf_krnl + cu_avg_tsdf_krnl (src/warpsense/cuda/update_tsdf.cu:13-128) of the reference.
//
// The reference scatters with a racy CAS ("first positive-weight entry freezes the voxel",
// include/warpsense/cuda/util.h:70-102), so its result depends on thread arrival order.  This
// implementation computes the result of ONE fixed legal schedule — the serial one, ascending
// (point, ray step, fan step) — deterministically, without a single global atomic per candidate:
//
//   ray_setup / ray_sort     per-ray constants (update_tsdf.cu:52-63), rays sorted by the cell of the map their END lies in
//                            (the tail of a ray lies within tau of its end: 64 neighbours in that order put their records
//                            into the same few tiles).
//   march_tail_kernel        walks the ray TAILS once (near-surface and fan candidates: everything whose result
//                            depends on the order).  Every scatter target becomes an 8-byte record and goes STRAIGHT to
//                            its place: records live in 256-byte sub-chunks that belong to one 4x4x64-voxel tile each, a
//                            wave opens sub-chunks out of its own run of ids and counts its records per tile in LDS (the
//                            count is the place), and publishes its sub-chunks when it is through -- one atomic per
//                            (wave, tile).  No wave waits for another, HBM sees a record once.  (Round 4 built two other
//                            shapes first -- through a raw buffer and a copy, and staged in LDS: DESIGN.md §5.)  Off-ray
//                            candidates of value +tau are (tau, -64) whoever makes them and never take part in the
//                            order: a byte in the second voxel plane instead of a record.
//   march_free_kernel        walks the steps before the tails: all free space (tau, +64) whoever comes first ->
//                            one byte per voxel.  A free-space candidate landing on a voxel the tails marked joins that
//                            tile's records instead.
//   tile_resolve_kernel      ONE workgroup per touched 4x4x64-voxel tile: all candidates of the tile are present,
//                            so the canonical accept rule is a local fold in LDS (earliest positive / smallest
//                            negative keys, then exact rounds for the voxels where a negative-weight candidate
//                            may have blocked the earliest positive one).  Writes new_map, or — fused — integrates
//                            straight into avg_map (cu_avg_tsdf_krnl folded into the write-back).
//   integrate_*_kernel       weighted average of new_map into avg_map and reset of new_map, over the touched
//                            tiles only (sparse) or over every voxel (dense, the reference's kernel).
//
// The list of tiles with records is built on the way (the first entries of a tile append it): no scan over the tile grid, no
// descriptors.  new_map after the resolve is bit-identical to what the reference
// kernel leaves there when its threads run one after the other (oracle/ws_oracle.c: wso_update_min).
#include <atomic>
#include <chrono>
#include <cstddef>
#include <type_traits>

#include "ws_march.h"
#include "ws_dda.h"

namespace ws
{

struct ScatterArgs
{
  const int32_t *xyz;
  int32_t *xyz_keep; // the set-up pass copies the scan here (ws_map::scan_dev: what a repeat of an aborted scan reads); NULL: xyz is that buffer
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
  uint8_t *vstate;     // two planes of one byte per voxel: VOX_* / off-ray free-space mark
  uint8_t *tile_dirty; // one byte per tile: touched by the free-space pass
  uint32_t *tile_nsub;  // [tiles] sub-chunks (entries) of the tile
  uint32_t *tile_ent;   // [tiles][TILE_DIRECT] entries: sub-chunk id << 5 | records - 1
  TileEntry *tile_list; // the tiles with records (the first entries of a tile append it; the resolve deals them out evenly)
  unsigned long long *rec; // the pool: sub-chunks of SUB_RECS records
  uint32_t sub_cap;
  uint32_t scan_seq;  // sequence number of this scatter
  unsigned long long *big_keys; // (tile, entry number) -> entry + 1 beyond TILE_DIRECT: keys, then uint32 values (big_mask + 1 slots)
  uint32_t big_mask;
  uint32_t rec_fmt;   // the scan's split of the record's key bits: S | F << 8 (rec_format, ws_internal.h)
  uint32_t *tail_stats; // records / (flush, tile) groups per workgroup of the tail march
  TsdfCounters *counters;
  uint32_t *status; // host-mapped: [0] sticky error bits, [4..5] record bound of the scan in flight, [6] its sequence number, [8] / [9] see ws_map::status_host
};
// 264 bytes of kernel arguments instead of 256 cost reg_loop_kernel 30 % (registration.hip); the same bound here
static_assert(sizeof(ScatterArgs) <= 256, "ScatterArgs: more than 256 bytes of kernel arguments");

#define REC_S(a) ((int32_t)((a).rec_fmt & 0xffu))
#define REC_F(a) ((int32_t)((a).rec_fmt >> 8))
constexpr uint8_t VOX_KEYED = 1, VOX_TOUCHED = 2;
constexpr uint8_t VOX_NEGFREE = 8; // the resolve's merged view of the second byte plane (stored there as 1)
constexpr uint32_t ERR_RANGE = 2, ERR_FREE_BOUND = 4, ERR_INTERNAL = 8;

#ifndef WS_TAIL_SPLIT
#define WS_TAIL_SPLIT (8 / WS_TAIL_WAVES) // workgroups that share the tails of one group of 64 rays: eight parts in all
#endif
#ifndef WS_FREE_THREADS
#define WS_FREE_THREADS 256 // threads per workgroup of the free pass
#endif
#ifndef WS_FREE_FIRST
#define WS_FREE_FIRST 32 // sub-chunks every wave of the free pass owns from the start (see pool_grab; 16: 136 us, 32: 121, 64: 120)
#endif
#ifndef WS_SORT_BLOCKS
#define WS_SORT_BLOCKS 128 // (64 / 128 / 256 / 512 blocks: set-up + sort 29.0 / 28.0 / 29.1 / 34 us)
#endif
#ifndef WS_SORT_RINGS
#define WS_SORT_RINGS 16 // (tail march at 4 / 8 / 16 / 32 / 64 rings: 146 / 146 / 146 / 154 / 152 us, set-up pass 36 / 31 / 28 / 30 / 28)
#endif
constexpr int AZ_BINS = 8192; // sort bins of the rays: polar cells around the sensor (ray_setup_block); bin AZ_BINS = ray without steps
static_assert(AZ_BINS == 2 * 4096, "WS_SORT_RINGS rings x 4096 / WS_SORT_RINGS sectors, above / below the sensor");

size_t ray_setup_bytes() { return sizeof(RaySetup); }

__device__ __forceinline__ void raise_error(TsdfCounters *c, uint32_t *status, uint32_t bits)
{
  atomicOr(&c->error, bits);
  __hip_atomic_fetch_or(status, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // sticky, host visible
}

__device__ __forceinline__ uint32_t tile_of(int32_t nty, int32_t ntz, int32_t sx, int32_t sy, int32_t sz)
{
  // ntx * nty < 2^24 (checked by ws_map_create): full-rate 24-bit multiplies
  const uint32_t col = __umul24((uint32_t)(sx >> TILE_XB), (uint32_t)nty) + (uint32_t)(sy >> TILE_YB);
  return __umul24(col, (uint32_t)ntz) + (uint32_t)(sz >> TILE_ZB);
}
// The voxel inside its tile, twice, straight from the storage coordinates: `local` = lx | ly | lz, z fastest -- what a record
// carries and the resolve's LDS arrays are indexed by (a column's 64 z spread over all banks; round 6 measured records in brick
// order: the resolve 107 -> 116 us, a wall's records then fall on 8 banks) -- and `vox` = vbrick(local), its byte in the tile's
// kilobyte of voxel bytes (ws_internal.h).
static_assert(TILE_XB == 2 && TILE_YB == 2 && TILE_ZB == 6, "vox_of / vbrick: 4 x 4 x 8 bricks of a 4 x 4 x 64 tile");
__device__ __forceinline__ uint32_t vox_of(int32_t sx, int32_t sy, int32_t sz)
{
  const uint32_t xy = (((uint32_t)sx & 3u) << 2) | ((uint32_t)sy & 3u);
  return (((uint32_t)sz & 0x38u) << 4) | (xy << 3) | ((uint32_t)sz & 7u);
}
__device__ __forceinline__ uint32_t local_of(int32_t sx, int32_t sy, int32_t sz)
{
  const uint32_t xy = (((uint32_t)sx & 3u) << 2) | ((uint32_t)sy & 3u);
  return (xy << TILE_ZB) | ((uint32_t)sz & 63u);
}
// vbrick backwards (the rare free-space candidate that becomes a record)
__device__ __forceinline__ uint32_t local_of_vox(uint32_t vox) { return ((vox & 0x78u) << 3) | ((vox >> 4) & 0x38u) | (vox & 7u); }
// Byte `vox` of tile `tile` in a plane of voxel bytes / record place `pos` of sub-chunk `id`.  SMALL: the planes and the pool are
// below 4 GB (every map up to 1025^3; decided per launch): the offset is ONE 32-bit instruction next to a base address in scalar
// registers, instead of a 64-bit shift and two 64-bit additions per access -- the marches are bound by vector-instruction issue.
template <bool SMALL>
__device__ __forceinline__ uint8_t *vox_ptr(uint8_t *plane, uint32_t tile, uint32_t vox)
{
  if (SMALL) return plane + (uint32_t)((tile << 10) | vox);
  return plane + (((size_t)tile << 10) | vox);
}
template <bool SMALL>
__device__ __forceinline__ unsigned long long *rec_ptr(unsigned long long *pool, uint32_t id, uint32_t pos)
{
  if (SMALL) return reinterpret_cast<unsigned long long *>(reinterpret_cast<uint8_t *>(pool) + (uint32_t)((((id << SUB_BITS) | pos)) << 3));
  return pool + (((size_t)id << SUB_BITS) | pos);
}

// everything the scatter expects to be zero / empty, in ONE launch (after map creation and after the buffers were resized:
// in steady state every kernel of a scan puts back the scratch it has consumed, there is no clean-up launch)
struct PrepArgs
{
  TsdfCounters *counters;
  uint32_t *az_hist;
  uint32_t n_hist;
  uint32_t *tile_nsub;
  uint8_t *tile_dirty;
  int64_t n_tiles;
  unsigned long long *big_keys;
  uint32_t big_slots;
};
__global__ __launch_bounds__(256) void scatter_prep_kernel(PrepArgs p)
{
  const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  if (tid < (int64_t)(offsetof(TsdfCounters, last_records) / 4)) reinterpret_cast<uint32_t *>(p.counters)[tid] = 0;
  if (tid == 0) p.counters->abort = p.counters->error = 0;
  for (int64_t i = tid; i < p.n_hist; i += stride) p.az_hist[i] = 0;
  for (int64_t i = tid; i < p.n_tiles; i += stride) p.tile_nsub[i] = 0;
  for (int64_t i = tid; i < (int64_t)(2 * tile_flag_plane_bytes(p.n_tiles)); i += stride) p.tile_dirty[i] = 0;
  uint32_t *big_vals = reinterpret_cast<uint32_t *>(p.big_keys + p.big_slots);
  for (int64_t i = tid; i < p.big_slots; i += stride)
  {
    p.big_keys[i] = KEY_INF;
    big_vals[i] = 0;
  }
}

// Sub-chunks the pool should hold for a scan of `n_points` rays whose records are bounded by `need`: the block every
// workgroup of the tail march starts with, the records themselves (the bound counts every sample as a candidate, about 2.7 x
// what a scan makes; est_shift > 1 takes a fraction of it: tests), and room for the partly filled ones.  An estimate -- the
// number of (wave, tile) pairs has no useful bound -- and never more than that: a scan that does exhaust the pool is ABORTED
// (nothing of it reaches the maps) and repeated with more (launch_tsdf_scatter).
__host__ inline unsigned long long subs_needed(unsigned long long need, uint32_t est_shift, unsigned long long n_points)
{
  const unsigned long long items = (n_points + 63) / 64 * WS_TAIL_SPLIT + 1; // (TAIL_SPLIT workgroups per 64 rays)
  const unsigned long long free_waves = (n_points + 63) / 64 * 4; // (64 rays per workgroup of the free pass)
  return items * SUB_WG_BLOCK + free_waves * WS_FREE_FIRST + ((need >> SUB_BITS) >> (est_shift ? est_shift - 1 : 0)) + 8192ull;
}

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
__device__ __forceinline__ void ray_setup_block(const ScatterArgs &a)
{
  __shared__ unsigned long long ub_wave[4];
  __shared__ uint32_t s_bkey[512], s_bcnt[512]; // bins of this workgroup's rays: key, count (then: first rank)
  __shared__ int32_t s_fan[256];                // fan_steps (tail_bound walks it entry by entry: a chain of dependent loads from memory otherwise)
  s_bkey[threadIdx.x] = s_bkey[threadIdx.x + 256] = 0xffffffffu;
  s_bcnt[threadIdx.x] = s_bcnt[threadIdx.x + 256] = 0;
  s_fan[threadIdx.x] = a.fan_steps[threadIdx.x];
  __syncthreads();
  uint32_t my_bin = 0;
  const uint32_t ix = blockIdx.x * 256u + threadIdx.x;
  if (ix == 0)
  {
    // what the marches of this scan count up (the previous scan's integrate pass has read its tile list by now)
    a.counters->chunk_cursor = 0;
    a.counters->free_cursor = 0;
    a.counters->n_listed = 0;
    a.counters->n_appended = 0;
    a.counters->abort = 0;
    a.counters->error = 0;
    a.counters->last_free_keyed = 0;
    a.counters->last_unlisted = 0;
  }
  RaySetup r;
  r.dx = r.dy = r.dz = r.distance = r.ivx = r.ivy = r.ivz = r.steps = 0;
  r.div_m = 0;
  r.spare = 0;
  r.div_k = 0;
  r.pad = 0;
  r.kfirst = 0;
  r.ub = 0;
  const int32_t res = a.res, tau = a.tau, half = res / 2;
  const bool frame_biased_ok = make_march_frame(a.scanner_pos, res, tau, a.map).biased_ok;
  bool ok = false;
  int32_t px = 0, py = 0, pz = 0;
  int32_t hvx = 0, hvy = 0, hvz = 0; // voxel of the scan point
  if (ix < a.n)
  {
    px = a.xyz[3 * (size_t)ix + 0];
    py = a.xyz[3 * (size_t)ix + 1];
    pz = a.xyz[3 * (size_t)ix + 2];
    if (a.xyz_keep)
    {
      // (the verdict on the record pool comes after ws_tsdf_update_dev has returned; a repeat must not depend on what
      // the caller has done with its buffer since)
      a.xyz_keep[3 * (size_t)ix + 0] = px;
      a.xyz_keep[3 * (size_t)ix + 1] = py;
      a.xyz_keep[3 * (size_t)ix + 2] = pz;
    }
    // cu_to_map (cuda/util.h:111-114) + in_bounds_with_buffer_pos (update_tsdf.cu:55)
    const float fr = (float)res;
    const int32_t cx = (int32_t)floorf(__fdiv_rn((float)px, fr));
    const int32_t cy = (int32_t)floorf(__fdiv_rn((float)py, fr));
    const int32_t cz = (int32_t)floorf(__fdiv_rn((float)pz, fr));
    ok = in_bounds_buffer(a.map, cx, cy, cz, (int64_t)(tau / res / 2));
    hvx = cx;
    hvy = cy;
    hvz = cz;
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
        if (steps > (int64_t)(1 << (a.rec_fmt & 0xffu)) || (max_delta_z * 2) / res + 1 > (int64_t)((1 << (a.rec_fmt >> 8)) - 1) || !small_iv)
        {
          // Outside the range of the record's step / fan fields for a scan of this many points.  With the widest split (scans of
          // up to 16 384 points: 65 536 steps, 255 fan steps) that is the end: the ray is dropped and the error is sticky.  A
          // larger scan is ABORTED instead (bit 1 of the abort word; nothing of it reaches the maps) and the host repeats it in
          // pieces of 16 384 points, each with the widest split, one after the other into new_map (settle_tsdf: the serial order
          // is the order of the points, so consecutive pieces folded on top of each other are the same schedule).
          if ((a.rec_fmt & 0xffu) == 16u && (a.rec_fmt >> 8) == 8u)
            raise_error(a.counters, a.status, ERR_RANGE);
          else
            __hip_atomic_fetch_or(&a.counters->abort, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        else
        {
          r.dx = dx; r.dy = dy; r.dz = dz;
          r.distance = distance;
          r.ivx = (int32_t)ivx; r.ivy = (int32_t)ivy; r.ivz = (int32_t)ivz;
          r.steps = (int32_t)steps;
          const FastDiv fd = make_fastdiv_dev(distance);
          r.div_m = (uint32_t)fd.M; // (below 2^32: make_fastdiv)
          r.div_k = fd.k;
          // conditions of march_steps_fast (ws_march.h): no int32 wrap in d*len, pos + d, voxel centres,
          // delta_z*iv and step*res*iv, nor in the squared distance to the hit point (|p - centre| <= len_end + 2 res)
          const int64_t dmax = max(max(llabs((long long)dx), llabs((long long)dy)), llabs((long long)dz));
          const int64_t pmax = max(max(llabs((long long)posx), llabs((long long)posy)), llabs((long long)posz));
          const int64_t ivmax = max(max(llabs(ivx), llabs(ivy)), llabs(ivz));
          // dmax <= distance: beyond ~46 m the reference's int sum of squares wraps and `distance` is not the length
          // of the ray any more — the walk's "less than one voxel per step" then fails
          // (res * distance < 2^32: the column-change walk keeps W = res * dist in 32 bits, ws_dda.h -- ADVICE r5; only a map of
          // metre-sized voxels gets near it)
          const bool fast = dmax <= distance && dmax * len_end < (1ll << 31) && pmax + dmax + 2 * (int64_t)res + tau < (1ll << 30) &&
                            (int64_t)res * distance < (1ll << 32) &&
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
            bool inside = fast && frame_biased_ok; // (div_res_b / ring_b, ws_march.h: the window's coordinates fit their bias)
            for (int k = 0; k < 3; ++k)
            {
              const int64_t lim = (int64_t)(a.map.size[k] / 2) - margin;
              inside = inside && llabs((long long)((int64_t)a.scanner_pos[k] - a.map.pos[k])) <= lim && llabs((long long)(ev[k] - a.map.pos[k])) <= lim;
            }
            if (inside) r.pad |= RAY_SIMPLE;
          }

          // Split of the ray.  A candidate is "free space" iff its value is +tau: the on-ray one is (tau, +64), the off-ray
          // ones of its fan (tau, -64), whichever ray they come from -- so neither needs a record: the first is a byte per voxel
          // (its order only matters on a voxel that also has records: earliest key, free-space hash), the second never takes
          // part in the order at all (|value| == tau cannot block a positive candidate, and it only wins where nothing else
          // landed: a second byte plane).  ALL candidates of a ray are of that kind while len < distance - tau - slack: a
          // voxel centre further than tau from the hit point gives min(dist, tau) == tau.  The slack covers |centre - proj|
          // (1.5 voxels per axis for the double-width cell of trunc division + the fan offset).  The bound argues with
          // exact positions: rays whose `int` products wrap (not `fast`) and scans into a non-default new_map send every
          // step through the order keys.  (Until round 3 the tail also began at the first step with a fan, 8.2 m at 50 mm:
          // a quarter of the benchmark scan's records were free space with a fan.)
          int32_t kfirst = 0;
          if (fast && !a.all_keyed)
          {
            const int32_t keyed_len = min(a.keyed_len_neg, distance - tau - a.keyed_slack);
            kfirst = keyed_len > 1 ? max(0, (keyed_len - 1) / half - 1) : 0;
            if (kfirst > (int32_t)steps) kfirst = (int32_t)steps;
          }
          r.kfirst = kfirst;
          // records this ray can make: the scatter targets of its tail + one per free-space step (a free-space candidate that
          // lands on a voxel with records joins them)
          const unsigned long long ub = tail_bound(kfirst, steps, len_end, s_fan) + (unsigned long long)kfirst;
          r.ub = ub > 0xffffffffull ? 0xffffffffu : (uint32_t)ub;
        }
      }
    }
  }
  // Sort bin of the ray; bin AZ_BINS = unused ray.  The tails are sorted by WHERE THE RAY ENDS -- the tail of a ray lies within
  // tau of its end, so the 64 rays of a wave of the tail march put their records into the few tiles around one cell -- in
  // polar cells around the sensor (16 rings x 256 sectors, above / below the sensor), the FAR rings first: see below.
  // (Rounds 1-3 sorted by direction, 1024 azimuths x 8 elevations: rays of one direction bin that graze the floor end metres
  // apart -- 435 k (wave, tile) pairs per scan; a 64 x 64 grid of square cells in Morton order: 297 k, tail march 178 us; the
  // polar cells: 244 k, 146 us -- two thirds of that gain is the order of the work items.)
  if (ix < a.n)
  {
    uint32_t bin = AZ_BINS;
    if (r.steps > 0)
    {
      const int32_t cwx = (a.map.size[0] + 63) / 64, cwy = (a.map.size[1] + 63) / 64;
      int bx = (hvx - a.map.pos[0] + a.map.size[0] / 2) / cwx, by = (hvy - a.map.pos[1] + a.map.size[1] / 2) / cwy;
      bx = bx < 0 ? 0 : (bx > 63 ? 63 : bx);
      by = by < 0 ? 0 : (by > 63 ? 63 : by);
      // polar cells around the sensor -- WS_SORT_RINGS rings x 4096 / WS_SORT_RINGS sectors, above / below -- the FAR rings first: rays that end far away
      // carry fans (up to three times the records), and work items in descending order of their length leave the shortest
      // for the end of the launch, when the compute units run empty
      const float fdx = (float)r.dx, fdy = (float)r.dy;
      constexpr int RINGS = WS_SORT_RINGS, SECTORS = 4096 / RINGS;
      const float ringw = (float)(a.map.size[0] > a.map.size[1] ? a.map.size[0] : a.map.size[1]) * (float)res * (0.5f / (float)RINGS);
      int ring = (int)(sqrtf(fdx * fdx + fdy * fdy) / ringw);
      ring = ring < 0 ? 0 : (ring > RINGS - 1 ? RINGS - 1 : ring);
      int sec = (int)((atan2f(fdy, fdx) + 3.14159265f) * ((float)SECTORS / 6.2831853f));
      sec = sec < 0 ? 0 : (sec > SECTORS - 1 ? SECTORS - 1 : sec);
      bin = (uint32_t)(((RINGS - 1 - ring) * 2 + (hvz >= a.scanner_pos[2] ? 1 : 0)) * SECTORS + sec);
      (void)bx;
      (void)by;
    }
    r.pad |= (int32_t)(bin << 1); // bits 1 .. 14 (RAY_SIMPLE is bit 30)
    my_bin = bin;
    a.rays[ix] = r;
  }
  // The ray's rank inside its bin comes from the bin's counter (the sort blocks place it without a second atomic) -- through
  // the workgroup: its rays count themselves per bin in LDS first, and ONE add per (workgroup, bin) reserves their ranks.  A
  // cell near the sensor holds thousands of rays, and a returning atomic per ray on such a counter took the set-up pass from
  // 30 to 69 us; the 256 rays of a workgroup are neighbours in the scan and share a few dozen bins.  (It also keeps such
  // neighbours together in the sorted order.)  Everything travels at agent scope (performed at the coherent level): the
  // sort blocks run on other XCDs in the next launch.
  __syncthreads(); // (the table was emptied on entry)
  int my_slot = -1;
  uint32_t my_lrank = 0;
  if (ix < a.n)
  {
    uint32_t h = (my_bin * 0x9E3779B1u) >> (32 - 9);
    for (;;)
    {
      const uint32_t cur = s_bkey[h];
      if (cur == my_bin) break;
      if (cur == 0xffffffffu)
      {
        const uint32_t old = atomicCAS(&s_bkey[h], 0xffffffffu, my_bin);
        if (old == 0xffffffffu || old == my_bin) break;
      }
      h = (h + 1) & 511u;
    }
    my_slot = (int)h;
    my_lrank = atomicAdd(&s_bcnt[h], 1u);
  }
  // The records this scan can make (sum of the per-ray bounds): every scan sizes the record buffers itself.
  unsigned long long ub = r.ub;
  for (int d = 32; d > 0; d >>= 1) ub += __shfl_down(ub, d, 64);
  if ((threadIdx.x & 63) == 0) ub_wave[threadIdx.x >> 6] = ub;
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 2; ++q)
  {
    const int sl = (int)threadIdx.x + 256 * q;
    const uint32_t c = s_bcnt[sl];
    if (c) s_bcnt[sl] = __hip_atomic_fetch_add(&a.az_hist[s_bkey[sl]], c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (my_slot >= 0)
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(&a.ray_bin[ix]), (unsigned long long)my_bin | ((unsigned long long)(s_bcnt[my_slot] + my_lrank) << 32),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // one fire-and-forget add per workgroup; the direction sort -- the next launch -- hands the total to the host
  if (threadIdx.x == 0) atomicAdd(&a.counters->ub_total, ub_wave[0] + ub_wave[1] + ub_wave[2] + ub_wave[3]);
}

// Counting sort of the rays by direction bin (a launch of its own behind the set-up pass: fused into it, with the sort
// blocks waiting for the set-up blocks' arrival count, it measured 50 us against 37 for two launches -- the waiting blocks'
// polls queue in front of the adds they wait for).  Every sort block scans the 8193-entry histogram itself (32 KB, one block
// scan).  Block 0 also hands the scan's record bounds to the host (host-mapped memory: the values, then the sequence number
// the host spins on), which sizes the buffers before the marches may do anything -- the capacity never rests on a guess.
__global__ __launch_bounds__(256) void ray_sort_kernel(ScatterArgs a)
{
  __shared__ uint32_t s_off[AZ_BINS + 2];
  __shared__ uint32_t wave_sums[4];
  constexpr int TOTAL = AZ_BINS + 1;
  constexpr int PER = (TOTAL + 255) / 256;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.status + 4), a.counters->ub_total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(a.status + 6, a.scan_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  const int lo = threadIdx.x * PER, hi = min(lo + PER, TOTAL);
  uint32_t h[PER];
  uint32_t v = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j)
  {
    const int i = lo + j;
    h[j] = i < hi ? a.az_hist[i] : 0u;
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
  const uint32_t b = blockIdx.x, nb = gridDim.x;
  if (b == 0 && threadIdx.x == 0) a.az_off[AZ_BINS] = s_off[AZ_BINS]; // rays that contribute: the tail march's grid
  // (four rays per trip, their loads in flight together: one after the other the trips were a chain of memory round trips)
  for (uint32_t ix0 = b * 256u + threadIdx.x; ix0 < a.n; ix0 += nb * 256u * 4u)
  {
    unsigned long long br[4];
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u)
    {
      const uint32_t ix = ix0 + u * nb * 256u;
      br[u] = *reinterpret_cast<const unsigned long long *>(&a.ray_bin[ix < a.n ? ix : ix0]);
    }
#pragma unroll
    for (uint32_t u = 0; u < 4; ++u)
    {
      const uint32_t ix = ix0 + u * nb * 256u;
      if (ix < a.n) a.ray_order[s_off[(uint32_t)br[u]] + (uint32_t)(br[u] >> 32)] = ix;
    }
  }
}
__global__ __launch_bounds__(256) void ray_setup_kernel(ScatterArgs a) { ray_setup_block(a); }

// ---------------------------------------------------------------------------------------------------------
// records of a tile: the pool of sub-chunks, the tile's entry table
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned long long big_key(uint32_t tile, uint32_t j) { return ((unsigned long long)tile << 24) | j; } // j < 2^23
__device__ __forceinline__ uint32_t big_slot(unsigned long long key, uint32_t mask) { return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 32) & mask; }

// the scan in flight ran out of sub-chunks: from here on nothing of it may reach the maps -- the resolve only puts the scratch
// back and the host repeats the scan with a larger pool (launch_tsdf_scatter)
__device__ __forceinline__ void raise_abort(const ScatterArgs &a)
{
  __hip_atomic_fetch_or(&a.counters->abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (bit 1: a ray beyond the key range, ray_setup_block)
}

// The pool, bottom to top: [one block of SUB_WG_BLOCK ids per work item of the tail march | what its waves ask for on top of
// that (chunk_cursor, upwards) ... (free_cursor, downwards) what the waves of the free pass ask for on top of | FREE_WAVE_FIRST
// ids per wave of the free pass].  The fixed parts cost no request at all: a returning atomic on ONE address takes ~40 ns
// under load (measured: 25 000 of them, one per free-space record, made the free pass 1.16 ms instead of 0.12), so the
// shared counters are for the exceptions.
constexpr uint32_t FREE_WAVE_FIRST = WS_FREE_FIRST; // (subs_needed() counts them)
__device__ __forceinline__ uint32_t tail_static_subs(const ScatterArgs &a) { return ((a.n + 63u) / 64u) * (uint32_t)WS_TAIL_SPLIT * SUB_WG_BLOCK; }
__device__ __forceinline__ uint32_t free_static_subs(const ScatterArgs &a) { return ((a.n + 63u) / 64u) * 4u * FREE_WAVE_FIRST; }
__device__ __forceinline__ bool pool_holds_static(const ScatterArgs &a)
{
  return (unsigned long long)tail_static_subs(a) + free_static_subs(a) <= (unsigned long long)a.sub_cap;
}
// n more consecutive sub-chunk ids for a wave of the tail march, or SUB_LOST
__device__ __forceinline__ uint32_t pool_grab(const ScatterArgs &a, uint32_t n)
{
  const uint32_t b = __hip_atomic_fetch_add(&a.counters->chunk_cursor, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long lo = (unsigned long long)tail_static_subs(a) + b;
  if (pool_holds_static(a) && lo + n <= (unsigned long long)a.sub_cap - free_static_subs(a)) return (uint32_t)lo;
  raise_abort(a);
  return SUB_LOST;
}
// ... for the free pass (the next launch: chunk_cursor is final), from the top down
__device__ __forceinline__ uint32_t free_grab(const ScatterArgs &a, uint32_t n)
{
  const unsigned long long lo = (unsigned long long)tail_static_subs(a) + a.counters->chunk_cursor;
  const uint32_t d = __hip_atomic_fetch_add(&a.counters->free_cursor, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long top = (unsigned long long)a.sub_cap - free_static_subs(a);
  if (pool_holds_static(a) && lo + d + n <= top) return (uint32_t)(top - d - n);
  raise_abort(a);
  return SUB_LOST;
}

// entry number j of `tile`: into the tile's table, or -- beyond TILE_DIRECT -- into the hash (value: entry + 1)
__device__ __forceinline__ void entry_publish(const ScatterArgs &a, uint32_t tile, uint32_t j, uint32_t ent)
{
  if (j < (uint32_t)TILE_DIRECT)
  {
    a.tile_ent[(size_t)tile * TILE_DIRECT + j] = ent;
    return;
  }
  const unsigned long long key = big_key(tile, j);
  uint32_t *vals = reinterpret_cast<uint32_t *>(a.big_keys + (size_t)a.big_mask + 1);
  uint32_t h = big_slot(key, a.big_mask);
  for (uint32_t probe = 0; probe <= a.big_mask; ++probe)
  {
    const unsigned long long old = atomicCAS(&a.big_keys[h], KEY_INF, key);
    if (old == KEY_INF || old == key)
    {
      // (a key stays in the table when its tile is released -- only the value goes back to 0 -- so that the probe chains
      // through it stay whole; the host empties the whole table before it fills up)
      if (old == KEY_INF) atomicAdd(&a.counters->big_inserted, 1u);
      vals[h] = ent + 1u;
      return;
    }
    h = (h + 1) & a.big_mask;
  }
  raise_error(a.counters, a.status, ERR_INTERNAL); // (the table has two slots per sub-chunk of the pool)
}
// the tile got its first entries: place `at` of the scan's tile list, and the flag byte that keeps the resolve's scan for
// tiles WITHOUT records away from it
__device__ __forceinline__ void list_tile(const ScatterArgs &a, uint32_t at, uint32_t tile)
{
  // tile -> (tx, ty, tz) by multiply-shift (constants behind the fan table, ws_map_create): exact for tile ids below 2^31
  const uint32_t Mz = (uint32_t)a.fan_steps[256], My = (uint32_t)a.fan_steps[258];
  const int32_t sz = a.fan_steps[257], sy = a.fan_steps[259];
  const uint32_t col = sz >= 0 ? __umulhi(tile, Mz) >> sz : tile;
  const uint32_t tx = sy >= 0 ? __umulhi(col, My) >> sy : col;
  TileEntry e;
  e.tile = tile;
  e.tz = (int32_t)(tile - col * (uint32_t)a.ntz);
  e.ty = (int32_t)(col - tx * (uint32_t)a.nty);
  e.tx = (int32_t)tx;
  a.tile_list[at] = e;
  a.tile_dirty[tile_flag_plane_bytes((int64_t)a.ntx * a.nty * a.ntz) + tile] = 1;
}
__device__ __forceinline__ uint32_t make_entry(uint32_t id, uint32_t fill) { return (id << SUB_BITS) | (fill - 1u); }

// one record in a sub-chunk of its own (a free-space candidate on a keyed voxel: 25 000 of the benchmark scan's 21 million)
__device__ __forceinline__ void append_single(const ScatterArgs &a, uint32_t tile, uint32_t id, unsigned long long rec)
{
  if (id == SUB_LOST) return;
  a.rec[(size_t)id << SUB_BITS] = rec;
  const uint32_t j = __hip_atomic_fetch_add(&a.tile_nsub[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  entry_publish(a, tile, j, make_entry(id, 1u));
  if (j == 0) list_tile(a, __hip_atomic_fetch_add(&a.counters->n_listed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), tile);
}

// ---------------------------------------------------------------------------------------------------------
// ray tails -> records, straight into sub-chunks of their tiles
// ---------------------------------------------------------------------------------------------------------
#ifndef WS_TAIL_KO
#define WS_TAIL_KO 0 // knock-out builds for timing (results wrong): 1 no voxel-byte stores, 2 no record store, 4 no table / record at all, 8 no rounds, 16 no publishing at the end
#endif
#ifndef WS_TAIL_WGS
#define WS_TAIL_WGS 5 // workgroups per CU the register budget is set for (six: 80 VGPRs, 16 of them spilled, 234 instead of 187 us)
#endif
constexpr int TAIL_SPLIT = WS_TAIL_SPLIT, TAIL_WAVES = WS_TAIL_WAVES; // workgroups that share the tails of one group of 64 rays (TAIL_WAVES parts each)
constexpr int TAIL_QCAP = 128; // queue entries per wave of the compacting walk (one sample phase adds at most 64)
constexpr uint32_t HT_EMPTY = 0xffffffffu;

typedef uint32_t __attribute__((aligned(1))) u32_a1; // four consecutive vstate bytes

// What a wave of the tail march keeps in LDS about the records it has made since it last published (wave_flush): nothing in
// here is shared with another wave -- no barrier, no waiting; LDS operations of one wave are performed in order.
//   key / cnt     the tiles of those records (open addressing) and how many each has: the counter's old value IS the record's
//                 place -- sub-chunk rank >> 5 of the wave's sub-chunks for that tile, position rank & 31
//   sub_of        the (local number of the) sub-chunks rank >> 5 = ..., modulo 4: one round of puts -- at most 64 records --
//                 spans three of a tile's sub-chunks at most
//   owner         local sub-chunk -> (slot, rank >> 5): what the flush publishes
//   blk           local sub-chunk l lives in pool sub-chunk blk[(l >> 5) & 7] + (l & 31): the wave's ids come in runs of 32
// Local numbers count up for the life of the wave; [flushed, n_local) are the ones not yet published, [n_local, covered) have
// an id waiting.  All of it modulo 256: flushed, rounded down to 32, and covered are never more than 256 apart.
constexpr int WT_BITS = 8, WT_SLOTS = 1 << WT_BITS;
constexpr uint32_t WT_SLOT_LIMIT = 224; // tiles in the table before the wave publishes and starts over
constexpr uint32_t WT_RING = 256;
constexpr uint32_t WT_LOCAL_LIMIT = 160; // sub-chunks in flight before it does
struct WaveTab
{
  uint32_t key[WT_SLOTS];
  uint32_t cnt[WT_SLOTS]; // (wave_flush: | first entry number << 13)
  uint8_t sub_of[WT_SLOTS][4];
  uint16_t owner[256];
  uint32_t blk[8];
  uint32_t n_local, n_slots, flushed, covered;
  uint32_t n_rec, n_groups; // statistics: records (general walk), (flush, tile) groups
};

__device__ __forceinline__ int wt_insert(WaveTab &wt, uint32_t tile, bool &fresh)
{
  uint32_t h = (tile * 0x9E3779B1u) >> (32 - WT_BITS);
  for (int p = 0; p < WT_SLOTS; ++p)
  {
    const uint32_t cur = wt.key[h];
    if (cur == tile) return (int)h;
    if (cur == HT_EMPTY)
    {
      const uint32_t old = atomicCAS(&wt.key[h], HT_EMPTY, tile);
      if (old == HT_EMPTY)
      {
        atomicAdd(&wt.n_slots, 1u);
        fresh = true;
      }
      if (old == HT_EMPTY || old == tile) return (int)h;
    }
    h = (h + 1) & (WT_SLOTS - 1);
  }
  return -1; // (never: wave_room keeps 32 slots free)
}

// The wave publishes the sub-chunks it has filled since the last time and empties its table.  Any set of lanes may call it
// (the general walk does, with whoever is there).  ONE memory round trip: a tile's entries are reserved with one atomic per
// (wave, tile) -- four tiles per lane travel together -- and written behind it; a tile that had no entries yet goes on the
// scan's tile list (one request to the list's counter per flush).
template <bool LAST = false> // LAST: the wave is through (its table is not used again: not emptied)
__device__ __forceinline__ void wave_flush(const ScatterArgs &a, WaveTab &wt)
{
  const unsigned long long act = __ballot(1);
  const int lane = threadIdx.x & 63;
  const uint32_t na = (uint32_t)__popcll(act), lr = (uint32_t)__popcll(act & ((1ull << lane) - 1ull));
  const int leader = __ffsll((long long)act) - 1;
  const uint32_t n_local = wt.n_local, flushed = wt.flushed;
  if (n_local != flushed)
  {
    for (uint32_t s0 = 0; s0 < (uint32_t)WT_SLOTS; s0 += 4u * na)
    {
      uint32_t c[4], tile[4], j0[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        const uint32_t s = s0 + lr + (uint32_t)u * na;
        c[u] = s < (uint32_t)WT_SLOTS ? wt.cnt[s] : 0u;
        tile[u] = s < (uint32_t)WT_SLOTS ? wt.key[s] : 0u;
        j0[u] = 0;
        if (c[u]) j0[u] = __hip_atomic_fetch_add(&a.tile_nsub[tile[u]], (c[u] + (uint32_t)SUB_RECS - 1u) >> SUB_BITS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      uint32_t my_first = 0, n_first = 0, n_used = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
      {
        const uint32_t s = s0 + lr + (uint32_t)u * na;
        const bool first = c[u] != 0 && j0[u] == 0;
        const unsigned long long fm = __ballot(first);
        if (first) my_first |= (((n_first + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull))) & 0x7fu) | 0x80u) << (8 * u);
        n_first += (uint32_t)__popcll(fm);
        n_used += (uint32_t)__popcll(__ballot(c[u] != 0));
        if (c[u])
        {
          if (j0[u] >= (1u << 19) - 256u) raise_error(a.counters, a.status, ERR_INTERNAL); // (half a million entries of one tile: never)
          wt.cnt[s] = c[u] | (j0[u] << 13);
        }
      }
      // (a lane's place among the firsts travels in seven bits: a flush of more than 127 new tiles takes the list places one by one)
      if (n_first)
      {
        if (n_first < 128u)
        {
          uint32_t lb = 0;
          if (lane == leader) lb = __hip_atomic_fetch_add(&a.counters->n_listed, n_first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          lb = (uint32_t)__builtin_amdgcn_readlane((int)lb, leader);
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (my_first & (0x80u << (8 * u))) list_tile(a, lb + ((my_first >> (8 * u)) & 0x7fu), tile[u]);
        }
        else
        {
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (my_first & (0x80u << (8 * u)))
              list_tile(a, __hip_atomic_fetch_add(&a.counters->n_listed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), tile[u]);
        }
      }
      if (lane == leader) wt.n_groups += n_used;
    }
    asm volatile("" ::: "memory");
    // the sub-chunks: entry number = the tile's reservation + the sub-chunk's number among the wave's for that tile
    const uint32_t nl = n_local - flushed;
    for (uint32_t q = lr; q < nl; q += na)
    {
      const uint32_t gl = (flushed + q) & (WT_RING - 1u);
      const uint32_t o = wt.owner[gl];
      const uint32_t s = o & 255u, sub = o >> 8;
      const uint32_t cj = wt.cnt[s];
      const uint32_t c = cj & 8191u, j0 = cj >> 13;
      const uint32_t ns = (c + (uint32_t)SUB_RECS - 1u) >> SUB_BITS;
      const uint32_t fill = sub + 1u == ns ? c - (sub << SUB_BITS) : (uint32_t)SUB_RECS;
      const uint32_t base = wt.blk[(gl >> 5) & (WT_RING / 32u - 1u)];
      if (sub < ns && base != SUB_LOST) entry_publish(a, wt.key[s], j0 + sub, make_entry(base + (gl & 31u), fill)); // (sub >= ns: the unused rest of a run)
    }
    asm volatile("" ::: "memory");
  }
  if (LAST) return;
  for (uint32_t s = lr; s < (uint32_t)WT_SLOTS; s += na)
  {
    wt.key[s] = HT_EMPTY;
    wt.cnt[s] = 0;
  }
  if (lane == leader)
  {
    wt.flushed = n_local;
    wt.n_slots = 0;
  }
  asm volatile("" ::: "memory");
}

// Room for `n_put` more records (n_put <= 64), whatever tiles they fall into: each can open one sub-chunk and one table slot
// at most.  Publishes and / or asks the pool for 32 more ids when it must; returns how many records the wave can put before
// it has to ask again (>= 64).  Uniform over the calling lanes.
__device__ __forceinline__ uint32_t wave_room(const ScatterArgs &a, WaveTab &wt)
{
  const unsigned long long act = __ballot(1);
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)act) - 1;
  uint32_t nl = wt.n_local, ns = wt.n_slots, fl = wt.flushed, cov = wt.covered;
  constexpr uint32_t PER_PUT = 1u; // local numbers (sub-chunks) a put can open
  constexpr uint32_t NEED = 64u * PER_PUT;             // ... a round of 64 puts
  if (nl - fl + NEED > WT_LOCAL_LIMIT || ns + 64u > WT_SLOT_LIMIT || (cov - nl < NEED && cov + NEED - (fl & ~31u) > WT_RING))
  {
    wave_flush(a, wt);
    fl = nl;
    ns = 0;
  }
  while (cov - nl < NEED)
  {
    // (flushed above if the ring of blocks had no place for more)
    uint32_t b = 0;
    if (lane == leader)
    {
      b = pool_grab(a, SUB_REFILL);
      for (uint32_t i = 0; i < SUB_REFILL; i += 32u) wt.blk[((cov + i) >> 5) & (WT_RING / 32u - 1u)] = b == SUB_LOST ? SUB_LOST : b + i;
      wt.covered = cov + SUB_REFILL;
    }
    cov += SUB_REFILL;
  }
  asm volatile("" ::: "memory");
  const uint32_t r0 = (WT_LOCAL_LIMIT - (nl - fl)) / PER_PUT, r1 = WT_SLOT_LIMIT - ns, r2 = (cov - nl) / PER_PUT;
  return min(r0, min(r1, r2));
}

// one record of the wave: its tile's slot, its rank there, the sub-chunk (opened by the record of rank 0 mod 32), its place
// Returns bit 0: the record opened a sub-chunk, bit 1: its tile is new in the table (what the caller's room shrinks by).
template <bool SMALL>
__device__ __forceinline__ uint32_t wave_put(const ScatterArgs &a, WaveTab &wt, uint32_t tile, unsigned long long rec)
{
  bool fresh = false;
  const int s = wt_insert(wt, tile, fresh);
  if (s < 0)
  {
    raise_error(a.counters, a.status, ERR_INTERNAL);
    return 0;
  }
  // (the lanes of a wave mostly hit ONE counter, and the LDS takes such atomics one lane at a time: the old value is used for
  // everything -- no second atomic on the word)
  const uint32_t rank = atomicAdd(&wt.cnt[s], 1u);
  const uint32_t sub = rank >> SUB_BITS, pos = rank & (uint32_t)(SUB_RECS - 1);
  if (pos == 0)
  {
    const uint32_t g = atomicAdd(&wt.n_local, 1u);
    wt.sub_of[s][sub & 3u] = (uint8_t)g;
    wt.owner[g & 255u] = (uint16_t)((uint32_t)s | (sub << 8));
  }
  asm volatile("" ::: "memory");
  const uint32_t gl = wt.sub_of[s][sub & 3u];
  const uint32_t base = wt.blk[(gl >> 5) & 7u];
#if WS_TAIL_KO & 2
  if (base != SUB_LOST && rec == 0x12345ull) *rec_ptr<SMALL>(a.rec, base + (gl & 31u), pos) = rec; // (never)
#else
  if (base != SUB_LOST) *rec_ptr<SMALL>(a.rec, base + (gl & 31u), pos) = rec;
#endif
  return (pos == 0 ? 1u : 0u) | (fresh ? 2u : 0u);
}

// one work item: 64 direction-sorted rays x four of the 4 * TAIL_SPLIT parts of their tails (one part per wave): the scatter
// targets of a wave fall into the same vertical slab of space, i.e. into few tiles.  Every wave is on its own: its records go
// straight from the march into sub-chunks of their tiles (wave_put) and are published when it is through (wave_flush).
// (Round 4 measured two other shapes first: the records through a slice of a raw buffer in HBM and a copy by the workgroup
// into 2 KB chunks per tile -- 226-233 us, 66 of them the copy; and staged in LDS, flushed whenever the area filled up --
// 243-258 us, it costs two workgroups per CU of occupancy.)
template <bool SMALL>
__device__ __forceinline__ void tail_item(const ScatterArgs &a, const uint32_t item)
{
  __shared__ WaveTab s_tab[TAIL_WAVES];
  __shared__ u32x4 s_queue[TAIL_WAVES * TAIL_QCAP];
  __shared__ uint32_t s_stat[2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  WaveTab &wt = s_tab[wave];
#ifdef WS_TAIL_TIMING
  const long long t_begin = wall_clock64();
#endif
  const uint32_t n_sorted = a.az_off[AZ_BINS];
  const uint32_t slot = (item / (uint32_t)TAIL_SPLIT) * 64u + (uint32_t)lane;
  const int part0 = (int)(item % (uint32_t)TAIL_SPLIT) * TAIL_WAVES; // this workgroup's parts of the tails
  const bool has_ray = slot < n_sorted;
  uint32_t ix = 0;
  RaySetup r;
  r.steps = 0;
  r.kfirst = 0;
  r.ub = 0;
  r.pad = 0;
  if (has_ray)
  {
    ix = a.ray_order[slot];
    r = a.rays[ix];
  }
  // ---- phase 0: the work item's block of sub-chunk ids (fixed: no request to anybody), every wave's table
  const uint32_t s_block = pool_holds_static(a) ? item * SUB_WG_BLOCK : SUB_LOST;
  if (threadIdx.x == 0)
  {
    if (s_block == SUB_LOST) raise_abort(a);
    s_stat[0] = s_stat[1] = 0;
    a.tail_stats[item] = 0;
    a.tail_stats[WS_TAIL_STATS + item] = 0;
  }
  for (int i = lane; i < WT_SLOTS; i += 64)
  {
    wt.key[i] = HT_EMPTY;
    wt.cnt[i] = 0;
  }
  if (lane == 0)
  {
    wt.n_local = wt.flushed = wt.n_slots = 0;
    wt.covered = SUB_WAVE_FIRST;
    wt.n_rec = wt.n_groups = 0;
  }
  __syncthreads();
  if (s_block == SUB_LOST) return; // (the scan is aborted: the host repeats it with a larger pool)
  if (lane < (int)(SUB_WAVE_FIRST / 32u)) wt.blk[lane] = s_block + (uint32_t)wave * SUB_WAVE_FIRST + (uint32_t)lane * 32u;
  int32_t k0 = 0, k1 = 0;
  if (has_ray && r.steps > 0 && r.kfirst < r.steps)
  {
    const int32_t kbeg = r.kfirst, kend = r.steps;
    const int32_t ch = (kend - kbeg + TAIL_WAVES * TAIL_SPLIT - 1) / (TAIL_WAVES * TAIL_SPLIT);
    k0 = min(kbeg + (part0 + wave) * ch, kend);
    k1 = min(k0 + ch, kend);
  }
  const bool work = k0 < k1;
  uint32_t n_written = 0; // records this wave has made (uniform; the general walk counts in LDS)

  // ---- phase 1: march, one record per scatter target
  const MarchFrame f = make_march_frame(a.scanner_pos, a.res, a.tau, a.map);
  const bool mark = !a.all_keyed;
  uint8_t *const vneg = a.vstate + vstate_plane_bytes((int64_t)a.ntx * a.nty * a.ntz);
  // a record (sx, sy, sz: storage coordinates of its voxel); returns the voxel's tile
  auto put_record = [&](uint32_t rix, int32_t k, int32_t fan_minus_mid, int32_t value, int32_t sx, int32_t sy, int32_t sz, uint32_t &used) -> uint32_t {
    // the free-space pass must know that this voxel takes part in the key order
    // (as a non-temporal store -- the marks push the half-filled sub-chunk lines out of the L2: 380 MB of writes for 98 MB of
    // records -- the kernel takes 462 instead of 183 us)
    const uint32_t tile = tile_of(a.nty, a.ntz, sx, sy, sz), vox = vox_of(sx, sy, sz);
    if (mark) *vox_ptr<SMALL>(a.vstate, tile, vox) = VOX_KEYED;
    used = wave_put<SMALL>(a, wt, tile, make_rec(rix, k, fan_minus_mid, value, local_of(sx, sy, sz), REC_S(a), REC_F(a)));
    return tile;
  };
  // an off-ray candidate of value +tau: (tau, -64) whoever makes it, never ordered (see ray_setup_block) -> a mark in the second plane
  auto mark_negative = [&](int32_t sx, int32_t sy, int32_t sz, uint32_t listed_tile) {
    const uint32_t tile = tile_of(a.nty, a.ntz, sx, sy, sz);
    *vox_ptr<SMALL>(vneg, tile, vox_of(sx, sy, sz)) = 1;
    // (the tile of the sample's on-ray record is on the list through that record -- nearly always this tile too; any other gets
    // the byte the resolve scans for.  A blind store: a load here would be a wait for everything the wave has in flight.)
    if (tile != listed_tile) a.tile_dirty[tile] = 1;
  };

  const bool general = !__all(!work || ((r.pad & RAY_SIMPLE) && r.distance >= 2)); // (>= 2: the 32-bit multiplier of ws_dda.h)
  if (general)
  {
    // a ray of this wave wraps in int32 or leaves the window: the general walk with all its tests, record by record.  (The
    // literal form with its divisions for every ray of such a wave: exact for all of them, and without the carried-remainder
    // walk's state the kernel fits 80 vector registers -- six workgroups per CU -- without a spill; such waves are rare.)
    if (work)
      march_steps_direct(f, r, k0, k1, [&](int32_t kk, int32_t step, int32_t vx, int32_t vy, int32_t vz, int32_t value, bool positive) {
        const int32_t sx = ring_fast(vx, f.ringK[0], a.map.size[0]), sy = ring_fast(vy, f.ringK[1], a.map.size[1]),
                      sz = ring_fast(vz, f.ringK[2], a.map.size[2]);
        if (mark && !positive && value == a.tau)
        {
          mark_negative(sx, sy, sz, 0xffffffffu);
          return;
        }
        // fan step - mid: update_tsdf.cu:103-104 (`positive` == the on-ray step)
        const int32_t delta_z = wmul(DZ_PER_DISTANCE, 1 + kk * f.half) / MATRIX_RESOLUTION;
        // (the lanes reach this point in varying company: room for whoever is here, counted in LDS)
        (void)wave_room(a, wt);
        atomicAdd(&wt.n_rec, 1u);
        uint32_t used = 0;
        put_record(ix, kk, step - delta_z / f.res, value, sx, sy, sz, used);
      });
  }
  else if (__any(work))
  {
    // compacting walk (ws_march.h): the sample phase queues (position, step, ray) of every sample that enters a new
    // voxel column; the emit phase pops 64 of them and does update_tsdf.cu:81-125 with every lane busy
    u32x4 *queue = s_queue + wave * TAIL_QCAP;
    uint32_t qhead = 0, qtail = 0;
    uint32_t cap_left = 0; // records the wave may put before it looks at its bookkeeping again (uniform)
    const int32_t res = f.res, half = f.half, tau = f.tau, dist = r.distance;
    // the scan point (update_tsdf.cu:57), shifted by divB - half: what the biased voxel index of div_res_b is subtracted from
    const int32_t hshift = (int32_t)f.divB - half;
    const int32_t hitbx = f.posx + r.dx + hshift, hitby = f.posy + r.dy + hshift, hitbz = f.posz + r.dz + hshift;
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
    auto push = [&](unsigned long long mask /* ballot of cand */, bool cand, bool cx, bool cy) {
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
      push(__ballot(first), first, false, false);
      if (work && k0 == 0) k = 1;
    }
    int32_t todo = work ? k1 - k : 0;
    for (int d = 32; d > 0; d >>= 1) todo = max(todo, __shfl_xor(todo, d, 64));
    const int32_t n_iter = __builtin_amdgcn_readfirstlane(todo);
    // emit phase: up to 64 queued samples, one per lane
    auto emit_batch = [&]() {
      const uint32_t cnt = qtail - qhead;
      const uint32_t n = cnt < 64 ? cnt : 64;
      u32x4 e = {0, 0, 0, 0};
      const bool has = (uint32_t)lane < n;
      if (has) e = queue[(qhead + (uint32_t)lane) & (TAIL_QCAP - 1)];
      qhead += n;
      // constants of the ray the sample belongs to (a lane of this wave)
      const int src = (int)(e.w >> 16);
      const int32_t s_hitx = __shfl(hitbx, src, 64), s_hity = __shfl(hitby, src, 64), s_hitz = __shfl(hitbz, src, 64);
      const int32_t s_ivx = __shfl(r.ivx, src, 64), s_ivy = __shfl(r.ivy, src, 64), s_ivz = __shfl(r.ivz, src, 64);
      const int32_t s_dist = __shfl(r.distance, src, 64);
      const uint32_t s_ix = (uint32_t)__shfl((int)ix, src, 64);
      const int32_t ek = (int32_t)(e.w & 0xffffu);
      const int32_t projx = (int32_t)e.x, projy = (int32_t)e.y, projz = (int32_t)e.z;
      const int32_t len = 1 + ek * half;
      // update_tsdf.cu:81-98 (no int32 wrap for a RAY_SIMPLE ray: 24-bit multiplies are exact).  The voxel's index comes biased by
      // divBq (div_res_b): centre = (q - divBq) res + half, and the hit point was shifted by divB - half once per ray
      const int32_t ddx = s_hitx - (int32_t)__umul24(div_res_b(projx, f), (uint32_t)res), ddy = s_hity - (int32_t)__umul24(div_res_b(projy, f), (uint32_t)res),
                    ddz = s_hitz - (int32_t)__umul24(div_res_b(projz, f), (uint32_t)res);
      int32_t value = (int32_t)sqrtf((float)(__mul24(ddx, ddx) + __mul24(ddy, ddy) + __mul24(ddz, ddz)));
      value = value < tau ? value : tau;
      if (len > s_dist) value = -value;
      // update_tsdf.cu:101-105
      const int32_t delta_z = (DZ_PER_DISTANCE * len) >> 15; // len > 0
      int32_t iter_steps = 0, mid = 0;
      if (has && !tsdf_weight_is_zero(value, tau, f.weight_epsilon))
      {
        iter_steps = 1;
        if (delta_z * 2 >= res)
        {
          iter_steps = (int32_t)(__umulhi((uint32_t)(delta_z * 2), f.rM32) >> f.rS) + 1;
          mid = (int32_t)(__umulhi((uint32_t)delta_z, f.rM32) >> f.rS);
        }
      }
      if (!__any(iter_steps > 0)) return;
      // the off-ray targets of a sample of value +tau are marks, not records
      const bool blind = mark && value == tau;
      const unsigned long long m_blind = mark ? __ballot(value == tau) : 0ull;
      // The fan (update_tsdf.cu:107-112): target j = (lowest + trunc(j res iv / 32768)) / res.  The products grow by res * iv
      // from one fan step to the next and their sign is iv's: `acc` carries j res iv + (iv < 0 ? 32767 : 0), the truncating
      // division by 32768 is its arithmetic shift -- one add and one shift per axis and round instead of multiply, sign, mask,
      // add, shift (round 6; the three multiply-shift divisions by res and the ring buffer: div_res_b / ring_b, ws_march.h)
      const int32_t bx = iv_bias(s_ivx), by = iv_bias(s_ivy), bz = iv_bias(s_ivz);
      const int32_t lowx = projx - trunc15_biased(delta_z, s_ivx, bx), lowy = projy - trunc15_biased(delta_z, s_ivy, by),
                    lowz = projz - trunc15_biased(delta_z, s_ivz, bz);
      const int32_t incx = __mul24(res, s_ivx), incy = __mul24(res, s_ivy), incz = __mul24(res, s_ivz);
      int32_t accx = bx, accy = by, accz = bz;
      // the parts of the record that belong to the sample (make_rec, ws_internal.h): u = step << F | fan, fan = round - mid + MID
      const int32_t recS = REC_S(a), recF = REC_F(a);
      const uint32_t rec_hi0 = s_ix << (recS + recF - 6), rec_lo0 = ((uint32_t)value & 0xffffu) << REC_VALUE_SHIFT;
      const uint32_t rec_u0 = ((uint32_t)ek << recF) + rec_fan_mid(recF) - (uint32_t)mid;
      // Rounds of at most one target per lane, fan step by fan step (update_tsdf.cu:107-125) IN THE FAN'S OWN ORDER: round j is
      // fan step j of every sample whose fan has more than j steps -- the on-ray target (always a record) where j == mid, an
      // off-ray one (a record, or a mark for a sample of value +tau) elsewhere.  The samples of a batch come from rays that
      // end in the same cell, i.e. of nearly the same length, and the fan's width depends on the length alone: the rounds run
      // 92 % full (tools/lane_model.py).  (Until round 5 the on-ray targets had a round of their own in front and every lane
      // sat out the round j == mid: 381 k rounds of 59 % instead of 244 k for the benchmark scan's 14.4 M targets.)  In front
      // of every round the wave makes sure its bookkeeping has room for the records of the round (a scalar compare, nearly always).
      uint32_t mid_tile = 0xffffffffu; // the tile of the sample's on-ray record, once that is made (it is on the list through it)
#if WS_TAIL_KO & 8
      if ((rec_hi0 ^ rec_u0 ^ (uint32_t)lowx ^ (uint32_t)lowy ^ (uint32_t)lowz ^ (uint32_t)incx ^ (uint32_t)incy ^ (uint32_t)incz ^ m_blind) == 0x12345u)
#endif
      for (int32_t round = 0;; ++round)
      {
        // (the loop bound as a ballot per round: a maximum over the lanes by shuffles is six trips through the LDS pipe per emit phase)
        const unsigned long long m_on = __ballot(round < iter_steps);
        if (m_on == 0) break;
        const bool on = round < iter_steps;
        const bool onray = round == mid;
        const bool puts = on && (onray || !blind);
        const uint32_t n_put = (uint32_t)__popcll(m_on & (__ballot(onray) | ~m_blind)); // ballot(puts), from scalar masks
        if (n_put)
        {
          // (every record of the round could open a sub-chunk and bring a new tile: room for that, then count what they did)
          if (cap_left < n_put) cap_left = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_room(a, wt));
          n_written += n_put;
        }
        // the target's storage coordinates (all lanes: the fan's state moves on in every round)
        const int32_t sx = ring_b(div_res_b(lowx + (accx >> 15), f), f.ringB[0], a.map.size[0]),
                      sy = ring_b(div_res_b(lowy + (accy >> 15), f), f.ringB[1], a.map.size[1]),
                      sz = ring_b(div_res_b(lowz + (accz >> 15), f), f.ringB[2], a.map.size[2]);
        accx += incx;
        accy += incy;
        accz += incz;
        uint32_t used = 0;
        if (on)
        {
          const uint32_t tile = tile_of(a.nty, a.ntz, sx, sy, sz), vox = vox_of(sx, sy, sz);
          if (!puts)
          {
            // an off-ray candidate of value +tau: a mark in the second plane (mark_negative)
#if !(WS_TAIL_KO & 1)
            *vox_ptr<SMALL>(vneg, tile, vox) = 1;
            if (tile != mid_tile) a.tile_dirty[tile] = 1;
#endif
          }
          else
          {
#if !(WS_TAIL_KO & 1)
            if (mark) *vox_ptr<SMALL>(a.vstate, tile, vox) = VOX_KEYED;
#endif
            const uint32_t u = rec_u0 + (uint32_t)round;
            const uint32_t hi = rec_hi0 | (u >> 6), lo = (u << REC_T_SHIFT) | rec_lo0 | local_of(sx, sy, sz);
#if WS_TAIL_KO & 4
            if ((hi ^ lo ^ tile) == 0x12345u) used = wave_put<SMALL>(a, wt, tile, ((unsigned long long)hi << 32) | lo); // (never: keeps the arithmetic alive)
#else
            used = wave_put<SMALL>(a, wt, tile, ((unsigned long long)hi << 32) | lo);
#endif
            if (onray) mid_tile = tile;
          }
        }
        if (n_put)
        {
          const uint32_t n_open = (uint32_t)__popcll(__ballot(used & 1u)), n_new = (uint32_t)__popcll(__ballot(used & 2u));
          cap_left -= n_open > n_new ? n_open : n_new;
        }
      }
    };
    for (int32_t it = 0; it < n_iter; ++it)
    {
      // ---- sample phase
      const bool cx = fast_step(wx, res), cy = fast_step(wy, res);
      fast_step_z(wz);
      // (ballots of the simple conditions, combined as scalars: the ballot of a conjunction costs two vector instructions more)
      push((__ballot(cx) | __ballot(cy)) & __ballot(k < k1), (cx || cy) && k < k1, cx, cy);
      k += 1;
      // ---- emit phase: 64 queued samples, one per lane
      if (qtail - qhead >= 64) emit_batch();
    }
    while (qtail != qhead) emit_batch();
  }
#ifdef WS_TAIL_TIMING
  const long long t_mid = wall_clock64();
#endif
  // ---- phase 2: the wave publishes what it has filled
#if !(WS_TAIL_KO & 16)
  wave_flush<true>(a, wt);
#endif
  if (lane == 0)
  {
    atomicAdd(&s_stat[0], n_written + wt.n_rec);
    atomicAdd(&s_stat[1], wt.n_groups);
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    a.tail_stats[item] = s_stat[0];
    a.tail_stats[WS_TAIL_STATS + item] = s_stat[1];
  }
#ifdef WS_TAIL_TIMING
  // (instead of the statistics: 10 ns ticks of the march and of the flush of this item, and when it started)
  if (threadIdx.x == 0)
  {
    a.tail_stats[item] = (uint32_t)(t_mid - t_begin);
    a.tail_stats[WS_TAIL_STATS + item] = (uint32_t)(wall_clock64() - t_mid);
    a.tail_stats[2 * WS_TAIL_STATS + 8192 + item] = (uint32_t)t_begin;
  }
#endif
}

template <bool SMALL> // SMALL: 32-bit offsets into the voxel bytes and the record pool (vox_ptr)
__global__ __launch_bounds__(64 * WS_TAIL_WAVES, WS_TAIL_WGS * 4 / WS_TAIL_WAVES) void march_tail_kernel(ScatterArgs a)
{
  // the direction histogram has been consumed by the sort blocks of this scan: zero for the next one (no clean-up launch)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < (uint32_t)(AZ_BINS + 1); i += gridDim.x * blockDim.x) a.az_hist[i] = 0;
  const uint32_t n_items = ((a.n + 63u) / 64u) * (uint32_t)TAIL_SPLIT;
  // (No look at counters->abort here, although a scan that the set-up pass has aborted for a ray beyond the key range could leave at
  // once: the word shares its cache line with the pool's cursor, which other workgroups of THIS launch hit with atomics -- every
  // workgroup starting with a load of it took the kernel from 135 to 210 us.  Such a scan marches in vain and is repeated in pieces.)
  if (blockIdx.x < n_items) tail_item<SMALL>(a, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// free space
// ---------------------------------------------------------------------------------------------------------
// What a free-space candidate does to its voxel, in two halves: the byte of the voxel is REQUESTED when the candidate is
// popped from the queue and USED one emit phase later.  The free pass is not bound by instruction issue alone: shortening
// the sample phase from ~115 to ~60 instructions moved it from 137 to 126 us, taking this load's round trip off the wave's
// path to 122 us; what remains is the scattered byte traffic itself (21 M byte loads, 9 M byte stores, one cache line each).
struct FreePending
{
  uint32_t tile;
  uint32_t vox;  // voxel inside the tile (brick order: vox_of)
  uint32_t ix;   // ray
  int32_t k;     // ray step
  uint32_t b;    // the voxel's byte (in flight until the next step)
  bool valid;
};
// sx, sy, sz: storage coordinates of the candidate's voxel
template <bool SMALL>
__device__ __forceinline__ void free_request(const ScatterArgs &a, FreePending &p, bool valid, uint32_t ix, int32_t k, int32_t sx, int32_t sy, int32_t sz)
{
  p.valid = valid;
  // unconditional (clamped) load: nothing waits for it here
  p.tile = valid ? tile_of(a.nty, a.ntz, sx, sy, sz) : 0u;
  p.vox = valid ? vox_of(sx, sy, sz) : 0u;
  p.ix = ix;
  p.k = k;
  p.b = *vox_ptr<SMALL>(a.vstate, p.tile, p.vox);
}
// the sub-chunks of the records the free pass makes (one each): a wave of the compacting walk keeps the rest of the 64 it
// took from the pool (fb_next, fb_left: uniform); the general walk -- lanes in varying company -- asks for what it needs
struct FreeBlock
{
  uint32_t next, left;
};
template <bool CACHED, bool SMALL>
__device__ __forceinline__ void free_finish(const ScatterArgs &a, const FreePending &p, uint32_t &n_keyed, FreeBlock &fb)
{
  const uint32_t b = p.b;
  // (the ballot of a conjunction goes through a vector register and back -- v_cndmask + v_cmp; two ballots and a scalar AND do not)
  const unsigned long long km = __ballot(p.valid) & __ballot((b & VOX_KEYED) != 0);
  const bool keyed = p.valid && (b & VOX_KEYED);
  if (km)
  {
    // the voxel also has ordered candidates (from the tails): this one, (tau, +64) at its place in the order, joins the
    // records of the tile (25 000 of the benchmark scan's 21 million free-space candidates)
    const int lane = threadIdx.x & 63;
    const uint32_t n = (uint32_t)__popcll(km);
    const int leader = __ffsll((long long)km) - 1;
    uint32_t first;
    if (CACHED)
    {
      if (fb.left < n)
      {
        uint32_t g = 0;
        if (lane == leader) g = free_grab(a, 64u);
        fb.next = (uint32_t)__builtin_amdgcn_readlane((int)g, leader);
        fb.left = fb.next == SUB_LOST ? 0u : 64u;
      }
      first = fb.left ? fb.next : SUB_LOST;
      if (fb.left)
      {
        fb.next += n;
        fb.left -= n;
      }
    }
    else
    {
      uint32_t g = 0;
      if (lane == leader) g = free_grab(a, n);
      first = (uint32_t)__builtin_amdgcn_readlane((int)g, leader);
    }
    if (keyed)
    {
      const uint32_t id = first == SUB_LOST ? SUB_LOST : first + (uint32_t)__popcll(km & ((1ull << lane) - 1ull));
      // (the answer of the atomic in there picked up one emit phase later, under the next batch's voxel bytes: no gain, measured)
      append_single(a, p.tile, id, make_rec(p.ix, p.k, 0, a.tau, local_of_vox(p.vox), REC_S(a), REC_F(a)));
      n_keyed += 1;
    }
  }
  if (p.valid && b == 0)
  {
    // free space only (the common case): the result will be (tau, 64) whoever comes first.  (Two candidates of one voxel
    // whose loads both saw 0 both store: idempotent.)
    *vox_ptr<SMALL>(a.vstate, p.tile, p.vox) = VOX_TOUCHED;
    // (remembering the tiles a workgroup has marked in an LDS set instead of this load: 126 -> 140 us, measured; an atomic
    // that puts the tile on the scan's list at its first mark: 123 -> 355 us -- the load sees stale zeros from the L1 of its
    // compute unit all through the kernel, harmless for a byte store, a blocking round trip for a returning atomic)
    if (a.tile_dirty[p.tile] == 0) a.tile_dirty[p.tile] = 1;
  }
}
// both halves at once (general walk)
template <bool SMALL>
__device__ __forceinline__ void free_emit(const ScatterArgs &a, const MarchFrame &f, uint32_t ix, int32_t k, int32_t vx, int32_t vy, int32_t vz, uint32_t &n_keyed)
{
  FreePending p;
  FreeBlock none = {0, 0};
  free_request<SMALL>(a, p, true, ix, k, ring_fast(vx, f.ringK[0], a.map.size[0]), ring_fast(vy, f.ringK[1], a.map.size[1]),
                      ring_fast(vz, f.ringK[2], a.map.size[2]));
  free_finish<false, SMALL>(a, p, n_keyed, none);
}

#ifndef WS_FREE_LANES
#define WS_FREE_LANES 4
#endif
constexpr int FREE_LANES = WS_FREE_LANES; // lanes that share the free-space part of one ray

// 64 rays per workgroup, 4 lanes per ray (round 2 walk: 32 lanes 163 us, 16: 146, 8: 141, 4: 147, 1: 280; round 3 walk: 8: 123, 4: 120, 2: 131): lane c walks the steps [c*CH, (c+1)*CH) of the free-space part of its ray,
// so every lane has the same amount of work whatever the ray length.  Waves whose rays are all RAY_SIMPLE use the
// compacting walk (ws_march.h): samples for all lanes, candidates through a per-wave LDS queue, 64 at a time.
// (Round 4 measured the free part extended over the steps that carry a fan -- 8.2 m to the tail at 50 mm, their off-ray
// targets as marks in the second byte plane: 10.8 M records instead of 14.4 M and a tail march of 146 instead of 187 us, but
// a free pass of 195-230 instead of 123 us whatever the lane layout: out there neighbouring rays are more than a voxel
// apart, every candidate is a cold cache line, and THIS pass waits for the byte it loads where the tail march only stores.)
#ifndef WS_FREE_WGS
#define WS_FREE_WGS 6 // workgroups per CU the register budget is set for (round 5's walk over column changes, 5 / 6 / 7 / 8: 105 / 103 / 102 / 120 us;
                      // six: 80 VGPRs, one spilled outside the loops; seven: 72 with 15 spilled; round 4's stepped walk: 121 / 120 / 117 / 147)
#endif
template <bool SMALL> // SMALL: 32-bit offsets into the voxel bytes (vox_ptr)
__global__ __launch_bounds__(WS_FREE_THREADS, WS_FREE_WGS * 256 / WS_FREE_THREADS) void march_free_kernel(ScatterArgs a)
{
#ifdef WS_FREE_TIMING
  const long long t_free_begin = wall_clock64();
#endif
  if (a.counters->abort != 0) return; // (the tail march ran out of sub-chunks: the host repeats the scan)
  __shared__ uint32_t s_keyed[WS_FREE_THREADS / 64];
  const uint32_t ix = blockIdx.x * (uint32_t)(WS_FREE_THREADS / FREE_LANES) + threadIdx.x / (uint32_t)FREE_LANES;
  const int32_t c = (int32_t)(threadIdx.x % (uint32_t)FREE_LANES);
  const int lane = threadIdx.x & 63;
  uint32_t n_keyed = 0;
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
  const int32_t res = f.res, half = f.half, dist = r.distance;
  if (!__all(!work || ((r.pad & RAY_SIMPLE) && r.distance >= 2)))
  {
    // a ray of this wave wraps in int32 or leaves the window: the general walk with all its tests
    if (work)
      march_steps<true>(f, r, k0, k1, [&](int32_t k, int32_t step, int32_t vx, int32_t vy, int32_t vz, int32_t value, bool positive) {
        // every candidate of these steps is free space: on the ray, further than tau from the hit point
        if (!(positive && value == tau))
        {
          raise_error(a.counters, a.status, ERR_FREE_BOUND); // impossible by the bound; never lose a candidate silently
          return;
        }
        free_emit<SMALL>(a, f, ix, k, vx, vy, vz, n_keyed);
      });
  }
  else if (__any(work))
  {
    // One loop iteration per CANDIDATE (ws_dda.h): the lane walks from one column change of its part of the ray to the next --
    // the steps at which x or y enters a new voxel are two Bresenham sequences -- and computes the sample's position from the
    // step number by one exact multiply-shift per axis.  No sample phase, no queue: rounds 3-4 stepped every sample (613 k wave
    // iterations of ~60 instructions for the benchmark scan) and moved the 21 M candidates through LDS to 333 k emit phases of
    // ~85; this loop runs 370 k times (tools/lane_model.py: 89 % of its lane slots carry a candidate).  The voxel byte of a
    // candidate is requested in one iteration and used in the next, as before.
    FreePending pend;
    FreeBlock fblock = {a.sub_cap - (blockIdx.x * (uint32_t)(WS_FREE_THREADS / 64) + (threadIdx.x >> 6) + 1u) * FREE_WAVE_FIRST, pool_holds_static(a) ? FREE_WAVE_FIRST : 0u};
    pend.valid = false;
    pend.tile = pend.vox = pend.ix = pend.b = 0;
    pend.k = 0;
    const uint32_t adx = (uint32_t)(r.dx < 0 ? -r.dx : r.dx), ady = (uint32_t)(r.dy < 0 ? -r.dy : r.dy), adz = (uint32_t)(r.dz < 0 ? -r.dz : r.dz);
    const int32_t smx = r.dx < 0 ? -1 : 0, smy = r.dy < 0 ? -1 : 0, smz = r.dz < 0 ? -1 : 0;
    const int32_t sposx = (f.posx ^ smx) - smx, sposy = (f.posy ^ smy) - smy, sposz = (f.posz ^ smz) - smz;
    // The walk lives in MIRRORED coordinates (every axis turned so that the ray travels in the positive direction: a = s pos + q),
    // and so does the rest of the step: the fan base offset c0 = trunc(delta_z * iv / 32768) (update_tsdf.cu:103-110 with one fan
    // step) with the mirrored s iv -- delta_z >= 0, so the product's sign is s iv's and the rounding toward zero a per-ray bias in
    // front of an arithmetic shift (trunc15_biased) --, the truncating division by res (trunc is odd: trunc(e / res) = s trunc(s e /
    // res)), and the sign comes back in the ONE instruction that adds the ring buffer's constant: x = s (qm - divBq) + offset - pos =
    // (qm ^ sm) + Kc, Kc = ringB for s = +1 and ringB + 2 divBq + 1 for s = -1 (v_xad_u32).  Two instructions per axis less than
    // un-mirroring the position first.
    const int32_t ivmx = (r.ivx ^ smx) - smx, ivmy = (r.ivy ^ smy) - smy, ivmz = (r.ivz ^ smz) - smz;
    const int32_t bvx = iv_bias(ivmx), bvy = iv_bias(ivmy), bvz = iv_bias(ivmz);
    const uint32_t kcx = (uint32_t)f.ringB[0] + (smx ? 2u * (uint32_t)f.divBq + 1u : 0u), kcy = (uint32_t)f.ringB[1] + (smy ? 2u * (uint32_t)f.divBq + 1u : 0u),
                   kcz = (uint32_t)f.ringB[2] + (smz ? 2u * (uint32_t)f.divBq + 1u : 0u);
    const uint32_t hdx = adx * (uint32_t)half, hdy = ady * (uint32_t)half, hdz = adz * (uint32_t)half, dzh = (uint32_t)(DZ_PER_DISTANCE * half);
    DdaRay R;
    R.M32 = r.div_m;
    R.sh = r.div_k - 32;
    DdaAxis wx, wy;
    wx.K = wy.K = wx.Ksp = wy.Ksp = DDA_NEVER;
    wx.rho = wy.rho = wx.wq = wy.wq = wx.wr = wy.wr = 0;
    wx.D = wy.D = 1;
    uint32_t k = DDA_NEVER; // the lane's next candidate (ray step)
    if (work)
    {
      const int32_t kinit = k0 > 0 ? k0 - 1 : 0;
      const int32_t len0 = 1 + kinit * half;
      const uint32_t qx = dda_q(adx, len0, R), qy = dda_q(ady, len0, R);
      dda_axis_init(wx, adx, sposx, qx, dist, res, half);
      dda_axis_init(wy, ady, sposy, qy, dist, res, half);
      k = min(wx.K, wy.K);
      // the sample k == 0 is compared with the voxel column (0, 0) (update_tsdf.cu:65,71): a candidate of its own in front
      if (k0 == 0 && (div_res(sposx + (int32_t)qx, f) != 0 || div_res(sposy + (int32_t)qy, f) != 0)) k = 0;
    }
    // One step of the walk: finish the candidate whose voxel byte the PREVIOUS step requested (it has had a whole step to
    // arrive), request the byte of the lane's next candidate, move on to the next column change.  (gfx950 retires loads and
    // stores in order behind one counter and the stores here are under branches, so the wait for a byte is a wait for
    // everything in flight; a variant that issued the same load and two stores in every step -- `vmcnt(3)` instead -- was no
    // faster: DESIGN.md section 5.)
    auto step = [&](auto special, FreePending &req) {
      const bool active = k < (uint32_t)k1;
      // ---- the sample's position (update_tsdf.cu:69) and its single on-ray target (:103-112 with iter_steps == 1)
      // (|d| * len_k = (|d| half) k + |d|, 100 * len_k = (100 half) k + 100: one multiply-add each)
      const int32_t ax = sposx + (int32_t)dda_qn(hdx * k + adx, R), ay = sposy + (int32_t)dda_qn(hdy * k + ady, R), az = sposz + (int32_t)dda_qn(hdz * k + adz, R);
      const int32_t dz = (int32_t)(dzh * k + (uint32_t)DZ_PER_DISTANCE) >> 15; // (DZ_PER_DISTANCE * len) >> 15; no fan in the free-space part: dz * 2 < res
      const int32_t ex = ax - trunc15_biased(dz, ivmx, bvx), ey = ay - trunc15_biased(dz, ivmy, bvy), ez = az - trunc15_biased(dz, ivmz, bvz);
      free_finish<true, SMALL>(a, req, n_keyed, fblock);
      free_request<SMALL>(a, req, active, ix, (int32_t)k, ring_m(div_res_b(ex, f), (uint32_t)smx, kcx, a.map.size[0]),
                          ring_m(div_res_b(ey, f), (uint32_t)smy, kcy, a.map.size[1]), ring_m(div_res_b(ez, f), (uint32_t)smz, kcz, a.map.size[2]));
      // ---- on to the next column change
      const bool cx = active && wx.K == k, cy = active && wy.K == k;
      if (cx)
      {
        const bool sp = decltype(special)::value && wx.Ksp == k;
        dda_axis_advance(wx);
        if (decltype(special)::value && sp) dda_axis_after_zero_cell(wx, adx, sposx, dist, res);
      }
      if (cy)
      {
        const bool sp = decltype(special)::value && wy.Ksp == k;
        dda_axis_advance(wy);
        if (decltype(special)::value && sp) dda_axis_after_zero_cell(wy, ady, sposy, dist, res);
      }
      if (active) k = min(wx.K, wy.K);
    };
    auto walk = [&](auto special) {
      while (__any(k < (uint32_t)k1)) step(special, pend);
    };
    // (a ray that crosses the cell around zero -- the one cell that is 2 res - 1 wide -- needs a look at every crossing: a
    // loop of its own for the waves that hold such a ray)
    if (__any(work && (wx.Ksp != DDA_NEVER || wy.Ksp != DDA_NEVER)))
      walk(std::true_type{});
    else
      walk(std::false_type{});
    free_finish<true, SMALL>(a, pend, n_keyed, fblock); // the last candidate
  }
  // statistics: free-space candidates that became records
  for (int d = 32; d > 0; d >>= 1) n_keyed += __shfl_down(n_keyed, d, 64);
  if (lane == 0) s_keyed[threadIdx.x >> 6] = n_keyed;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    uint32_t all = 0;
    for (int w = 0; w < WS_FREE_THREADS / 64; ++w) all += s_keyed[w];
    if (all) atomicAdd(&a.counters->last_free_keyed, all);
#ifdef WS_FREE_TIMING
    // (instead of the tail march's statistics: 10 ns ticks this workgroup took, and when it started -- tools/free_timing.py)
    a.tail_stats[blockIdx.x] = (uint32_t)(wall_clock64() - t_free_begin);
    a.tail_stats[WS_TAIL_STATS + blockIdx.x] = (uint32_t)t_free_begin;
#endif
  }
}

// ---------------------------------------------------------------------------------------------------------
// exact resolve of one tile in LDS
// ---------------------------------------------------------------------------------------------------------
struct ResolveArgs
{
  TileEntry *tile_list; // the tiles with records; the resolve appends the others it finds when a separate integrate pass follows
  uint32_t *tile_nsub;
  const uint32_t *tile_ent;
  uint8_t *tile_dirty;
  const unsigned long long *recs;
  unsigned long long *big_keys;
  uint32_t big_mask;
  uint32_t scan_seq;
  uint32_t sub_cap;
  uint32_t fan_mask, fan_mid; // the fan field of this scan's records (rec_format): the weight is negated iff fan != fan_mid
  uint32_t *new_data;
  uint32_t *avg_data;
  uint8_t *vstate;
  MapParams map;
  int32_t nty, ntz;
  int32_t tau, max_weight;
  uint32_t wM32;     // division by tau - tau/10 of the weight ramp (update_tsdf.cu:92) as one v_mul_hi_u32 + shift
  int32_t wS;
  uint32_t *resolve_stats; // [grid][2]: contested voxels, tiles with records
  TsdfCounters *counters;
  uint32_t *status;
  int64_t n_tiles;
};
static_assert(sizeof(ResolveArgs) <= 256, "ResolveArgs: more than 256 bytes of kernel arguments");

#ifndef WS_RESOLVE_KO
#define WS_RESOLVE_KO 0 // knock-out builds for timing (results wrong): 1 no scan A, 2 no LDS atomics in pass 1, 4 no records at all (fill = 0)
#endif
#ifndef WS_RESOLVE_GRID
#define WS_RESOLVE_GRID 1280 // 256 compute units x five resident workgroups: every workgroup is on the chip from the start (1280 / 2560 / 4096: 124 / 129 / 130 us)
#endif
constexpr int RESOLVE_GRID = WS_RESOLVE_GRID; // (block_stats holds two words per workgroup, 4096 at most)
static_assert(RESOLVE_GRID <= 4096, "resolve_stats");
constexpr uint32_t M_IDLE = 0xffffffffu, M_NONE = 0x10000u; // mstate: voxel not in the ordered rounds / no earlier negative seen
constexpr unsigned long long REC_NONE = ~0ull;

// kneg: smallest |value| wins, the LATEST candidate among equal |value| (a later equal one replaces the entry)
__device__ __forceinline__ uint64_t neg_key(uint64_t rec, int32_t av, int32_t value)
{
  const uint64_t t = rec >> REC_T_SHIFT;
  return ((uint64_t)av << 39) | ((T_MASK - t) << 1) | (value < 0 ? 1u : 0u);
}
__device__ __forceinline__ int32_t neg_key_abs(uint64_t N) { return (int32_t)(N >> 39); }

constexpr uint32_t ENT_NONE = 0xffffffffu;
// entry number j >= TILE_DIRECT of a tile: through the hash (the marches are over: everything is published)
__device__ __forceinline__ uint32_t resolve_entry(const ResolveArgs &a, uint32_t tile, uint32_t j)
{
  const unsigned long long key = big_key(tile, j);
  const uint32_t *vals = reinterpret_cast<const uint32_t *>(a.big_keys + (size_t)a.big_mask + 1);
  uint32_t h = big_slot(key, a.big_mask);
  for (uint32_t probe = 0; probe <= a.big_mask; ++probe)
  {
    const unsigned long long cur = a.big_keys[h];
    if (cur == key) return vals[h] - 1u; // (0: released -> ENT_NONE)
    if (cur == KEY_INF) break;
    h = (h + 1) & a.big_mask;
  }
  return ENT_NONE;
}

// every record of a tile, from memory: tiles of more than 8 * RES_MAXR sub-chunks, and the ordered rounds.  64 entries at a
// time -- lane l of every wave holds entry l of the batch: the first from the prefetched `cid`, the second from the tile's
// table, further ones through the hash, every lane looking up its own -- and of those WS_STREAM_U sub-chunks per half-wave in
// flight together (one after the other, each load waited for on the spot, this route took as long again as the whole fold).
#ifndef WS_STREAM_U
#define WS_STREAM_U 2 // (four: ten spilled registers in the fused resolve, 141 instead of 132 us)
#endif
template <class F>
__device__ __forceinline__ void for_each_record(const ResolveArgs &a, uint32_t tile, uint32_t nsub, uint32_t cid, uint32_t first, F &&f)
{
  const int lane = threadIdx.x & 63;
  const uint32_t half = threadIdx.x >> 5, pos = threadIdx.x & 31u;
  constexpr int U = WS_STREAM_U;
  // (first: a multiple of 8 below or at 64 -- the entries in front of it are in the caller's registers; a batch stays inside one block of
  // 64 entries: lane l of every wave holds entry 64 (b0 / 64) + l)
  for (uint32_t b0 = first; b0 < nsub; b0 = (b0 & ~63u) + 64u)
  {
    const uint32_t blk0 = b0 & ~63u, in0 = b0 & 63u;
    uint32_t ents = cid;
    if (blk0 == 64)
      ents = a.tile_ent[(size_t)tile * TILE_DIRECT + 64u + (uint32_t)lane];
    else if (blk0 > 64)
      ents = blk0 + (uint32_t)lane < nsub ? resolve_entry(a, tile, blk0 + (uint32_t)lane) : ENT_NONE;
    const uint32_t nb = min(64u, nsub - blk0); // entries of this block
    for (uint32_t j0 = in0; j0 < nb; j0 += 8u * U)
    {
      unsigned long long rec[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u)
      {
        const uint32_t j = j0 + 8u * (uint32_t)u + half;
        const uint32_t ent = (uint32_t)__shfl((int)ents, (int)(j & 63u), 64);
        ok[u] = j < nb && ent != ENT_NONE && pos <= (ent & 31u) && (ent >> SUB_BITS) < a.sub_cap;
        rec[u] = a.recs[ok[u] ? ((size_t)(ent >> SUB_BITS) << SUB_BITS) + pos : (size_t)threadIdx.x];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (ok[u])
        {
          const int32_t value = rec_value(rec[u]);
          f((uint64_t)rec[u], value, value < 0 ? -value : value, (int)rec_local(rec[u]));
        }
    }
  }
}

#ifndef WS_RES_MAXR
#define WS_RES_MAXR 8
#endif
#ifndef WS_RESOLVE_WGS
#define WS_RESOLVE_WGS 5 // (107 -> 102 VGPRs without spills, 29 KB of LDS: 4 -> 5 workgroups per CU, 139 -> 130 us)
#endif
constexpr int RES_MAXR = WS_RES_MAXR;   // records a thread keeps in registers (2048 record places = 64 sub-chunks per tile); larger tiles re-read them per pass
static_assert(RES_MAXR >= 1 && RES_MAXR * 8 <= 64 && TILE_DIRECT == 128, "the register route reads the first 8 RES_MAXR <= 64 entries of the tile's direct table, the streaming route the rest");
// (RES_MAXR 4 / 5 / 6 / 7 / 8, same box: the resolve 108-110 / 108 / 105-106 / 104-105 / 102-103 us -- fewer registers buy no sixth
// workgroup per CU, 28 KB of LDS and the other ~80 registers cap it at five, and the sub-chunks beyond the registers are streamed per pass)

// what a thread needs of a tile before it can start, requested two tiles ahead
struct TilePre
{
  int64_t idx0;
  int nz;
  uint32_t fill; // sub-chunks (entries) of the tile (uniform)
  uint32_t cid;  // lane l of every wave: entry l of the tile
  uint32_t vs;   // four vstate bytes (both planes)
  uint32_t s0[4]; // new_map entries (HAS_S0)
};
// the result of a tile, written back one tile later (behind the next tile's wait for its records)
struct TilePost
{
  int64_t idx0;
  int nz;
  uint32_t tile;
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
// bytes and entry table of later tiles and the records of the next tile are all requested together at the END of an
// iteration, and the stores of a tile are issued right AFTER the next wait, so they drain under the LDS phases.
//
// A scan that ran out of sub-chunks (counters->abort) leaves no trace: the tiles' scratch is put back as always, nothing is
// written to the maps, and the host runs the scan again with a larger buffer.
template <bool HAS_S0, bool FUSED>
__global__ __launch_bounds__(256, WS_RESOLVE_WGS) void tile_resolve_kernel(ResolveArgs a)
{
  __shared__ unsigned long long kpos[TILE_VOXELS];
  __shared__ unsigned long long kneg[TILE_VOXELS];
  __shared__ unsigned long long klast[TILE_VOXELS]; // the positive candidate that was blocked last
  __shared__ uint32_t mstate[TILE_VOXELS];          // M_IDLE: decided; else min |value| of the blocking negatives (M_NONE: none)
  __shared__ uint16_t bound0[HAS_S0 ? TILE_VOXELS : 1]; // |stored value| + 1 (0: frozen)
  __shared__ uint32_t s_unres[2];
#ifdef WS_RESOLVE_TIMING
  const long long t_begin = wall_clock64();
#endif
  const uint32_t n_list = a.counters->n_listed; // the tiles with records
  const bool aborted = a.counters->abort != 0;
  const int32_t weight_epsilon = a.tau / 10;
  const uint32_t reset = pack_entry(a.tau, 0);
  const int lane = threadIdx.x & 63;
  uint8_t *const vneg = a.vstate + vstate_plane_bytes(a.n_tiles);
  // thread t owns the voxels 4t .. 4t+3 of the tile: column t >> (ZB - 2), four consecutive z
  const int col = threadIdx.x >> (TILE_ZB - 2), lx = col >> TILE_YB, ly = col & ((1 << TILE_YB) - 1), z0 = (threadIdx.x & ((1 << (TILE_ZB - 2)) - 1)) * 4;
  const int l0 = threadIdx.x * 4;
  const uint32_t voff = vbrick((uint32_t)l0); // where the thread's four voxel bytes lie in the tile's kilobyte: one aligned word
  const uint32_t G = gridDim.x;
  uint32_t n_contested = 0;

  // all loads unconditional (clamped addresses, results masked)
  auto request = [&](const TileEntry &te, TilePre &p) {
    const int32_t sx = (te.tx << TILE_XB) + lx, sy = (te.ty << TILE_YB) + ly, sz = (te.tz << TILE_ZB) + z0;
    const bool col_ok = sx < a.map.size[0] && sy < a.map.size[1];
    int nz = a.map.size[2] - sz;
    p.nz = !col_ok ? 0 : (nz > 4 ? 4 : (nz < 0 ? 0 : nz));
    p.idx0 = p.nz ? storage_index(a.map, sx, sy, sz) : 0;
    p.fill = a.tile_nsub[te.tile];
    p.cid = a.tile_ent[(size_t)te.tile * TILE_DIRECT + (uint32_t)lane];
    // four voxels of a column in one access each (the arrays carry 16 bytes of slack behind the last voxel)
    const uint32_t keep = p.nz >= 4 ? 0xffffffffu : ((1u << (8 * p.nz)) - 1u);
    p.vs = 0;
    // both byte planes of the four voxels in one register: the second plane's mark becomes bit VOX_NEGFREE of the byte
    const size_t vb = ((size_t)te.tile << 10) + voff;
    if (!HAS_S0) p.vs = (*reinterpret_cast<const uint32_t *>(a.vstate + vb) | ((*reinterpret_cast<const uint32_t *>(vneg + vb) & 0x01010101u) << 3)) & keep;
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
  // The records of a tile, in registers: thread t takes place t & 31 of the tile's sub-chunks t >> 5, (t >> 5) + 8, ... -- one
  // coalesced 256-byte read per half-wave and sub-chunk, all RES_MAXR of them in flight together.  Returns false (uniform
  // over the workgroup) when the tile has more than 8 * RES_MAXR sub-chunks: the waves then stream them from memory in every
  // pass.  The loads are UNCONDITIONAL (clamped address) and all issued before the first result is touched: a load under a
  // branch, or a use right behind it, makes the compiler wait for each of them in turn.
  unsigned long long rrec[RES_MAXR]; // REC_NONE: no record
  u32x4 ex_next = {0, 0, 0, 0}; // FUSED: the avg_map entries of the tile whose records are in flight
  auto fetch_records = [&](const TilePre &p) -> bool {
    if (FUSED) ex_next = *reinterpret_cast<const u32x4_a4 *>(a.avg_data + p.idx0);
    const uint32_t nsub = aborted ? 0u : p.fill;
    const bool in_regs = nsub != 0; // (a tile of more than 8 * RES_MAXR sub-chunks: the first 64 here, the rest streamed)
    const uint32_t pos = threadIdx.x & 31u;
#pragma unroll
    for (int k = 0; k < RES_MAXR; ++k)
    {
      const uint32_t j = (uint32_t)(8 * k) + (threadIdx.x >> 5);
      const uint32_t ent = (uint32_t)__shfl((int)p.cid, (int)j, 64);
      const bool ok = in_regs && j < nsub && pos <= (ent & 31u) && (ent >> SUB_BITS) < a.sub_cap;
      const unsigned long long v = a.recs[ok ? ((size_t)(ent >> SUB_BITS) << SUB_BITS) + pos : (size_t)threadIdx.x];
      rrec[k] = ok ? v : REC_NONE;
    }
    return in_regs;
  };
  auto write_back = [&](const TilePost &w) {
    const size_t vb = ((size_t)w.tile << 10) + voff;
    if (w.nz == 4 && !HAS_S0) // (a non-default new_map keeps the entries of its untouched voxels: voxel by voxel below)
    {
      // the thread's four voxels as ONE access per array (byte stores are a transaction each: 14 M of them per scan were
      // most of this kernel's write traffic)
      if (w.vs & 0x07070707u) *reinterpret_cast<uint32_t *>(a.vstate + vb) = 0;
      if (w.vs & 0x08080808u) *reinterpret_cast<uint32_t *>(vneg + vb) = 0;
      if (w.touched == 0 || aborted) return;
      u32x4 out;
      if (FUSED)
      {
        out.x = (w.touched & 1u) ? integrate_entry(w.existing[0], w.value[0], a.max_weight) : w.existing[0];
        out.y = (w.touched & 2u) ? integrate_entry(w.existing[1], w.value[1], a.max_weight) : w.existing[1];
        out.z = (w.touched & 4u) ? integrate_entry(w.existing[2], w.value[2], a.max_weight) : w.existing[2];
        out.w = (w.touched & 8u) ? integrate_entry(w.existing[3], w.value[3], a.max_weight) : w.existing[3];
        if (out.x != w.existing[0] || out.y != w.existing[1] || out.z != w.existing[2] || out.w != w.existing[3])
          *reinterpret_cast<u32x4_a4 *>(a.avg_data + w.idx0) = out;
      }
      else
      {
        // (the untouched voxels of a default new_map are (tau, 0), and that is what `value` holds for them)
        out.x = w.value[0]; out.y = w.value[1]; out.z = w.value[2]; out.w = w.value[3];
        *reinterpret_cast<u32x4_a4 *>(a.new_data + w.idx0) = out;
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      if (j >= w.nz) continue;
      if (!HAS_S0 && ((w.vs >> (8 * j)) & (0xffu & ~(uint32_t)VOX_NEGFREE))) a.vstate[vb + j] = 0;
      if (!HAS_S0 && ((w.vs >> (8 * j)) & VOX_NEGFREE)) vneg[vb + j] = 0;
      if (!(w.touched & (1u << j)) || aborted) continue;
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
  // the tile's scratch goes back to zero for the next scan (no clean-up launch): its entry count (the table itself is only
  // ever read up to the count) and the values of its entries in the hash
  auto release_tile = [&](uint32_t tile, uint32_t nsub) {
    if (threadIdx.x == 8) a.tile_nsub[tile] = 0;
    // (the scan over the flag planes below owns the "listed" bytes and clears them; a scan into a non-default new_map has no
    // such pass -- ADVICE r4: the bytes of its listed tiles stayed set and hid those tiles from the NEXT scan's flag scan)
    if (HAS_S0 && threadIdx.x == 9) a.tile_dirty[tile_flag_plane_bytes(a.n_tiles) + tile] = 0;
    if (nsub > (uint32_t)TILE_DIRECT)
    {
      uint32_t *vals = reinterpret_cast<uint32_t *>(a.big_keys + (size_t)a.big_mask + 1);
      for (uint32_t j = (uint32_t)TILE_DIRECT + threadIdx.x; j < nsub; j += 256u)
      {
        const unsigned long long key = big_key(tile, j);
        uint32_t h = big_slot(key, a.big_mask);
        for (uint32_t probe = 0; probe <= a.big_mask; ++probe)
        {
          const unsigned long long cur = a.big_keys[h];
          if (cur == key)
          {
            // (the slot keeps its key: emptying it would cut the probe chains that run through it; a key of an earlier
            // scan with value 0 is "nothing" to the resolve and is overwritten by the tile's next entry j)
            vals[h] = 0;
            break;
          }
          if (cur == KEY_INF) break;
          h = (h + 1) & a.big_mask;
        }
      }
    }
  };

  // The per-scan scratch this scan has consumed goes back to zero here, for the next scan (no clean-up launch behind the
  // update); the host learns that the marches are over and whether the scan has to be repeated.
  if (blockIdx.x == 0 && threadIdx.x == 0)
  {
    TsdfCounters *c = a.counters;
    c->last_chunks = c->chunk_cursor;
    c->last_need = c->ub_total & ((1ull << 48) - 1ull);
    c->ub_total = 0;
    __hip_atomic_store(a.status + 10, c->big_inserted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(a.status + 9, a.counters->abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // (bit 0: pool exhausted, bit 1: key range)
    __hip_atomic_store(a.status + 8, a.scan_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // The tiles with records: the scan's tile list (the marches appended every tile at its first reservation), dealt out
  // evenly: workgroup b takes the entries b, b + G, ...
  const uint32_t e0 = blockIdx.x, e_end = n_list, ES = G;
  uint32_t n_mine = 0; // tiles this workgroup has folded
  if (e0 < e_end)
  {
  const uint32_t last = e_end - 1;
  // pipeline: tile i is processed while the voxel bytes / entry table of tiles i+1 and i+2, the list entries up to i+3
  // and (from the middle of the iteration on) the records of tile i+1 are in flight
  TileEntry te_n2 = a.tile_list[min(e0 + 2 * ES, last)], te_n3 = te_n2;
  uint32_t tile_cur, tile_n1;
  TilePre p_cur, p_n1, p_n2;
  {
    const TileEntry t0 = a.tile_list[e0], t1 = a.tile_list[min(e0 + ES, last)];
    request(t0, p_cur);
    request(t1, p_n1);
    tile_cur = t0.tile;
    tile_n1 = t1.tile;
  }
  bool cached = fetch_records(p_cur), cached_next = false;
  u32x4 ex_cur = ex_next;
  auto issue_next = [&](uint32_t e) {
    te_n3 = a.tile_list[min(e + 3 * ES, last)];
    request(te_n2, p_n2);
    cached_next = fetch_records(p_n1);
  };
  TilePost post;
  post.nz = 0;
  post.idx0 = 0;
  post.tile = 0;
  post.vs = post.touched = 0;
  if (threadIdx.x == 0) s_unres[0] = s_unres[1] = 0;
  init_lds();
  __syncthreads();

  for (uint32_t e = e0; e < e_end; e += ES)
  {
    {
      // The five workgroups of a compute unit start together with the same amount of work, and the SIMDs serve the OLDEST ready
      // wave first: the workgroups finished one after the other (65 ... 115 us, a round-5 instrumented build: the spread is inside the
      // compute units, not between them, and has nothing to do with the tiles a workgroup got), the compute unit ran its last
      // 25 us with one or two workgroups.  Now the issue priority goes round: a workgroup changes its priority with every tile,
      // the five of a compute unit (b, b + 256, ... in dispatch order) start at different places of the cycle -- spread inside a
      // compute unit 9.7 -> 4.9 us, the launch 110 -> 104 us.  (Priority by progress -- a quarter of the tiles done, one level
      // down: 105; time slices of 2.56 us on the shared clock: 106.)
      const uint32_t prio = (n_mine + blockIdx.x / 256u) & 3u;
      if (prio == 3u) __builtin_amdgcn_s_setprio(3);
      else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
      else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
      else __builtin_amdgcn_s_setprio(0);
    }
    n_mine += 1;
    const TilePre p = p_cur;
    const uint32_t tile = tile_cur, nsub_real = p.fill, fill = (aborted || (WS_RESOLVE_KO & 4)) ? 0u : p.fill; // (an aborted scan: the entries may be anything)
    const int nz = p.nz;
    const int64_t idx0 = p.idx0;

    uint32_t entry[4] = {reset, reset, reset, reset};
    uint32_t touched = 0;
    uint8_t vs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) vs[j] = (uint8_t)(p.vs >> (8 * j)); // <- the wait of this iteration (with the records)
    write_back(post); // the previous tile's stores drain under this tile's LDS phases

    if (fill == 0)
    {
      issue_next(e);
      // free space only: (tau, +64) where an on-ray candidate landed, else (tau, -64) where only off-ray ones did
      if (!HAS_S0)
      {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (vs[j] & (VOX_TOUCHED | VOX_NEGFREE))
          {
            entry[j] = pack_entry(a.tau, (vs[j] & VOX_TOUCHED) ? WEIGHT_RESOLUTION : -WEIGHT_RESOLUTION);
            touched |= 1u << j;
          }
      }
      __syncthreads(); // (keeps the workgroup's iterations together: release_tile below relies on it)
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
      bool from_regs = cached; // pass 1 and scan A; the rounds stream (the registers then hold the next tile's records)
      auto scan_records = [&](auto &&f) {
        if (from_regs)
        {
#pragma unroll
          for (int k = 0; k < RES_MAXR; ++k)
            if (rrec[k] != REC_NONE)
            {
              const int32_t value = rec_value(rrec[k]);
              const int32_t av = value < 0 ? -value : value;
              const int l = (int)rec_local(rrec[k]);
              if (HAS_S0 && av >= (int32_t)bound0[l]) continue; // rejected by the stored entry, now and for ever
              f((uint64_t)rrec[k], value, av, l);
            }
        }
        // what the registers do not hold: everything (the ordered rounds: the registers hold the next tile's records by then),
        // or the sub-chunks beyond the 64th of a heavy tile
        if (!from_regs || fill > (uint32_t)(8 * RES_MAXR))
        {
          for_each_record(a, tile, fill, p.cid, from_regs ? (uint32_t)(8 * RES_MAXR) : 0u, [&](uint64_t rec, int32_t value, int32_t av, int l) {
            if (HAS_S0 && av >= (int32_t)bound0[l]) return;
            f(rec, value, av, l);
          });
        }
      };
      // scan A: the negatives that come before the current positive candidate and can block it
      // (first: right behind pass 1 every voxel's mstate is still M_NONE -- nothing to look up there)
      auto scan_a = [&](auto first) {
        scan_records([&](uint64_t rec, int32_t value, int32_t av, int l) {
          if (!rec_negative(rec, a.fan_mask, a.fan_mid)) return;
          if (!decltype(first)::value)
          {
            const uint32_t m = mstate[l];
            if (m == M_IDLE || (uint32_t)av >= m) return;
          }
          const unsigned long long P = kpos[l];
          if (P == KEY_INF || rec > P) return;
          const int32_t vp = rec_value(P);
          if (av < (vp < 0 ? -vp : vp)) atomicMin(&mstate[l], (uint32_t)av);
        });
      };
      auto negative_entry = [&](unsigned long long N) {
        // no positive candidate is accepted: the negatives fold to the smallest |value|, latest on ties
        const int32_t an = neg_key_abs(N);
        const int32_t v = (N & 1ull) ? -an : an;
        return pack_entry(v, -weight_of(v));
      };

      // ---- pass 1: earliest positive, smallest negative per voxel
      scan_records([&](uint64_t rec, int32_t value, int32_t av, int l) {
        // (one LDS atomic with a selected address and key instead of two exec-mask regions per record)
        const bool neg = rec_negative(rec, a.fan_mask, a.fan_mid);
        const unsigned long long key = neg ? (unsigned long long)neg_key(rec, av, value) : (unsigned long long)rec;
#if WS_RESOLVE_KO & 2
        if (key == 0x12345ull) atomicMin(neg ? &kneg[l] : &kpos[l], key); // (never)
#else
        atomicMin(neg ? &kneg[l] : &kpos[l], key);
#endif
      });
      __syncthreads();
      // (round 6, measured and not kept: scan A only for tiles that hold off-ray records at all -- 105.9 us against 106.0:
      // nearly every tile of the benchmark scan does.  Knock-out builds, WS_RESOLVE_KO: scan A 14 us, the atomics of pass 1 16,
      // the rest of the fold 33, the kernel without any fold 50.)
      // (and: scan A only for the tiles that have a voxel whose smallest negative comes AFTER its earliest positive -- if it comes before,
      // it is itself the minimum scan A looks for -- decided per voxel behind pass 1, one more barrier: 29 % of the benchmark scan's
      // 19 535 tiles go without scan A then, 1.23 M of its voxels are contested, and the kernel takes 110.5 us against 102.3.)
#if !(WS_RESOLVE_KO & 1)
      scan_a(std::true_type{});
#endif
      __syncthreads();
      // this tile's records are not needed again (unless it needs ordered rounds, which stream): everything the next
      // iterations need is requested NOW and arrives under the decide phase, the barrier and the write-back
      from_regs = false;
      issue_next(e);

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
            const int32_t vp = rec_value(P);
            const uint32_t ap = (uint32_t)(vp < 0 ? -vp : vp);
            if (N != KEY_INF && ap > (uint32_t)neg_key_abs(N)) n_contested += 1; // a negative candidate COULD have blocked it
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
        else if (j < nz && !HAS_S0 && (vs[j] & (VOX_TOUCHED | VOX_NEGFREE)))
        {
          entry[j] = pack_entry(a.tau, (vs[j] & VOX_TOUCHED) ? WEIGHT_RESOLUTION : -WEIGHT_RESOLUTION);
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
          scan_records([&](uint64_t rec, int32_t value, int32_t av, int l) {
            if (rec_negative(rec, a.fan_mask, a.fan_mid)) return;
            const uint32_t m = mstate[l];
            if (m == M_IDLE || (uint32_t)av > m) return;
            if (rec > klast[l]) atomicMin(&kpos[l], (unsigned long long)rec);
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
          if (s_unres[phase] == 0) break;
          scan_a(std::false_type{});
          __syncthreads();
#pragma unroll
          for (int j = 0; j < 4; ++j)
          {
            if (!(unres & (1u << j))) continue;
            const unsigned long long P = kpos[l0 + j];
            const int32_t vp = rec_value(P);
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
    }

    // Nobody reads the tile's tables again: every thread's prefetch of them was consumed before the last barrier of the
    // PREVIOUS iteration, the passes that stream the sub-chunks from memory ended before the last barrier of this one.
    release_tile(tile, nsub_real);

    // ---- this tile's result waits in registers until the next iteration's loads have arrived
    post.idx0 = idx0;
    post.nz = nz;
    post.tile = tile;
    post.vs = p.vs;
    post.touched = touched;
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      post.value[j] = entry[j];
    }
    post.existing[0] = ex_cur.x; post.existing[1] = ex_cur.y; post.existing[2] = ex_cur.z; post.existing[3] = ex_cur.w;
    tile_cur = tile_n1;
    p_cur = p_n1;
    cached = cached_next;
    ex_cur = ex_next;
    tile_n1 = te_n2.tile;
    p_n1 = p_n2;
    te_n2 = te_n3;
  }
  write_back(post);
  } // listed tiles

  // ---- Tiles that are NOT on the list: no records, only marks of the free pass or of off-ray +tau candidates in the byte
  // planes -- (tau, +64) / (tau, -64) where a mark is, nothing to fold.  They are found by scanning the per-tile flag planes
  // (a byte per tile; the marches only ever STORE there: no atomic, no waiting in their loops), 16 tiles per thread and step,
  // granule g of workgroup b = b + G k so that a cluster of such tiles spreads over the grid.  The scan owns both planes
  // (it clears what it finds); the listed tiles above never look at them.
  if (!HAS_S0)
  {
    uint32_t *found = reinterpret_cast<uint32_t *>(kpos); // 2048 tile ids (the fold above is over)
    const int64_t n_gran = (a.n_tiles + 15) >> 4;
    uint8_t *const flag_mark = a.tile_dirty, *const flag_listed = a.tile_dirty + tile_flag_plane_bytes(a.n_tiles);
    __syncthreads();
    if (threadIdx.x == 0) s_unres[0] = 0;
    __syncthreads();
    for (int64_t k0 = 0; (int64_t)blockIdx.x + (int64_t)G * k0 < n_gran; k0 += 128)
    {
      const int64_t g = (int64_t)blockIdx.x + (int64_t)G * (k0 + threadIdx.x);
      if (threadIdx.x < 128 && g < n_gran)
      {
        const u32x4 d = *reinterpret_cast<const u32x4 *>(flag_mark + 16 * g), h = *reinterpret_cast<const u32x4 *>(flag_listed + 16 * g);
        const uint32_t dw[4] = {d.x, d.y, d.z, d.w}, hw[4] = {h.x, h.y, h.z, h.w};
        const u32x4 zero = {0, 0, 0, 0};
        if (d.x | d.y | d.z | d.w) *reinterpret_cast<u32x4 *>(flag_mark + 16 * g) = zero;
        if (h.x | h.y | h.z | h.w) *reinterpret_cast<u32x4 *>(flag_listed + 16 * g) = zero;
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
          const uint32_t dj = (dw[j >> 2] >> (8 * (j & 3))) & 0xffu, hj = (hw[j >> 2] >> (8 * (j & 3))) & 0xffu;
          if (dj != 0 && hj == 0 && 16 * g + j < a.n_tiles) found[atomicAdd(&s_unres[0], 1u)] = (uint32_t)(16 * g + j);
        }
      }
      __syncthreads();
      const uint32_t n_found = s_unres[0];
      if (n_found && threadIdx.x == 0) atomicAdd(&a.counters->last_unlisted, n_found);
      for (uint32_t i = 0; i < n_found; ++i)
      {
        const uint32_t tile = found[i];
        const int32_t tz = (int32_t)(tile % (uint32_t)a.ntz);
        const uint32_t colt = tile / (uint32_t)a.ntz;
        const int32_t ty = (int32_t)(colt % (uint32_t)a.nty), tx = (int32_t)(colt / (uint32_t)a.nty);
        const int32_t sx = (tx << TILE_XB) + lx, sy = (ty << TILE_YB) + ly, sz = (tz << TILE_ZB) + z0;
        const bool col_ok = sx < a.map.size[0] && sy < a.map.size[1];
        int nz = a.map.size[2] - sz;
        nz = !col_ok ? 0 : (nz > 4 ? 4 : (nz < 0 ? 0 : nz));
        const int64_t idx0 = nz ? storage_index(a.map, sx, sy, sz) : 0;
        const uint32_t keep = nz >= 4 ? 0xffffffffu : ((1u << (8 * nz)) - 1u);
        const size_t vb = ((size_t)tile << 10) + voff;
        const uint32_t vs4 = (*reinterpret_cast<const uint32_t *>(a.vstate + vb) | ((*reinterpret_cast<const uint32_t *>(vneg + vb) & 0x01010101u) << 3)) & keep;
        u32x4 ex = {0, 0, 0, 0};
        if (FUSED) ex = *reinterpret_cast<const u32x4_a4 *>(a.avg_data + idx0);
        TilePost w;
        w.idx0 = idx0;
        w.nz = nz;
        w.tile = tile;
        w.vs = vs4;
        w.touched = 0;
        w.existing[0] = ex.x; w.existing[1] = ex.y; w.existing[2] = ex.z; w.existing[3] = ex.w;
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
          const uint32_t b = (vs4 >> (8 * j)) & 0xffu;
          w.value[j] = (b & (VOX_TOUCHED | VOX_NEGFREE)) ? pack_entry(a.tau, (b & VOX_TOUCHED) ? WEIGHT_RESOLUTION : -WEIGHT_RESOLUTION) : reset;
          if (b & (VOX_TOUCHED | VOX_NEGFREE)) w.touched |= 1u << j;
        }
        write_back(w);
        if (!FUSED && !aborted && threadIdx.x == 0)
        {
          TileEntry e;
          e.tile = tile;
          e.tx = tx; e.ty = ty; e.tz = tz;
          // behind the marches' entries, through a counter of its own: n_listed is what every workgroup of this launch read on
          // entry (a workgroup that starts late must not take an appended tile for a listed one)
          a.tile_list[n_list + atomicAdd(&a.counters->n_appended, 1u)] = e;
        }
      }
      __syncthreads();
      if (threadIdx.x == 0) s_unres[0] = 0;
      __syncthreads();
    }
  }

  // statistics: one slot per workgroup, no shared counter
  for (int d = 32; d > 0; d >>= 1) n_contested += __shfl_down(n_contested, d, 64);
  __shared__ uint32_t s_stat[4];
  if ((threadIdx.x & 63) == 0) s_stat[threadIdx.x >> 6] = n_contested;
  __syncthreads();
  if (threadIdx.x == 0)
  {
    a.resolve_stats[2 * blockIdx.x + 0] = s_stat[0] + s_stat[1] + s_stat[2] + s_stat[3];
    a.resolve_stats[2 * blockIdx.x + 1] = n_mine;
#ifdef WS_RESOLVE_TIMING
    // (instead of the statistics: 10 ns ticks this workgroup was busy, and when it started)
    a.resolve_stats[2 * blockIdx.x + 0] = (uint32_t)(wall_clock64() - t_begin);
    a.resolve_stats[2 * blockIdx.x + 1] = (uint32_t)t_begin;
#endif
  }
}

constexpr int PREP_GRID = 512;
static PrepArgs make_prep_args(ws_map *m)
{
  PrepArgs p;
  p.counters = m->counters;
  p.az_hist = m->az_hist;
  p.n_hist = (uint32_t)(AZ_BINS + 1);
  p.tile_nsub = m->tile_nsub;
  p.tile_dirty = m->tile_dirty;
  p.n_tiles = m->n_tiles;
  p.big_keys = m->big_keys;
  p.big_slots = m->big_slots;
  return p;
}
int launch_scatter_prep(ws_map *m)
{
  hipLaunchKernelGGL(scatter_prep_kernel, dim3(PREP_GRID), dim3(256), 0, m->ctx->stream, make_prep_args(m));
  WS_HIP(hipGetLastError());
  m->status_host[10] = 0;
  m->prepped = true;
  return WS_OK;
}

// fan_steps[j] of tail_bound for one resolution (host side, once per map).  j = 0 is unused.
void fill_fan_steps(int32_t *fan_steps, int32_t res, int32_t ntz, int32_t nty)
{
  {
    // behind the fan table: the multiply-shift constants of the divisions by ntz and nty (tile id -> tile coordinates when a tile
    // goes on the scan's list, list_tile: two general 32-bit divisions per listed tile otherwise)
    const FastDiv dz = make_fastdiv(ntz), dy = make_fastdiv(nty);
    fan_steps[256] = (int32_t)(uint32_t)dz.M;
    fan_steps[257] = dz.k - 32; // (-1 for a divisor of 1: the quotient is the dividend)
    fan_steps[258] = (int32_t)(uint32_t)dy.M;
    fan_steps[259] = dy.k - 32;
  }
  const int64_t half = res / 2 > 0 ? res / 2 : 1;
  fan_steps[0] = 0;
  for (int64_t j = 1; j < 256; ++j)
  {
    const int64_t cj = (j * res + 1) / 2;                                                   // delta_z that gives 2*delta_z/res >= j
    const int64_t Lj = (cj * MATRIX_RESOLUTION + DZ_PER_DISTANCE - 1) / DZ_PER_DISTANCE;   // first length with that delta_z
    const int64_t kj = (Lj - 1 + half - 1) / half;                                          // first step with len_k >= Lj
    fan_steps[j] = (int32_t)(kj > (1ll << 30) ? (1ll << 30) : kj);                          // beyond the steps the record admits anyway
  }
}

uint64_t subs_for_scan(const ws_map *m, uint64_t need_records, uint64_t n_points) { return subs_needed(need_records, m->est_shift, n_points); }

// one attempt of the scatter: every kernel enqueued, nothing waited for
static int enqueue_scatter(ws_map *m, const int32_t *xyz_dev, size_t n, const int32_t scanner_pos[3], const int32_t up[3], bool fused, bool s0, uint32_t *seq_out)
{
  ws_context *ctx = m->ctx;
  hipStream_t s = ctx->stream;
  // the (tile, entry) hash keeps the keys of released tiles: empty it before it fills up
  if (m->status_host[10] > m->big_slots / 4) m->prepped = false;
  ScatterArgs sa;
  sa.xyz = xyz_dev;
  // (a scan, or a piece of one, that already lies in the map's own buffer is not copied there)
  sa.xyz_keep = (xyz_dev >= m->scan_dev && xyz_dev < m->scan_dev + 3 * MAX_SCAN_POINTS) ? nullptr : m->scan_dev;
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
  sa.tile_nsub = m->tile_nsub;
  sa.tile_ent = m->tile_ent;
  sa.tile_list = m->tile_list;
  sa.rec = m->rec;
  sa.sub_cap = m->sub_cap;
  sa.scan_seq = ++m->scan_seq;
  sa.big_keys = m->big_keys;
  sa.big_mask = m->big_slots - 1;
  {
    const RecFormat rf = rec_format(n);
    sa.rec_fmt = (uint32_t)rf.S | ((uint32_t)rf.F << 8);
  }
  sa.tail_stats = m->block_stats;
  sa.counters = m->counters;
  sa.status = m->status_dev;

  const dim3 block(256);
  const dim3 grid_setup((unsigned)((n + 255) / 256));
  const dim3 grid_tail((unsigned)((n + 63) / 64) * TAIL_SPLIT);
  const dim3 grid_free((unsigned)((n + WS_FREE_THREADS / FREE_LANES - 1) / (WS_FREE_THREADS / FREE_LANES)));
  m->tail_blocks = grid_tail.x;
  const bool fuse = fused && !s0;
  prof_begin(ctx, WS_K_SETUP);
  // normally the kernels of the previous update have left their scratch zero / empty on their way (m->prepped)
  if (!m->prepped)
  {
    hipLaunchKernelGGL(scatter_prep_kernel, dim3(PREP_GRID), block, 0, s, make_prep_args(m));
    m->status_host[10] = 0;
  }
  m->prepped = false;
  hipLaunchKernelGGL(ray_setup_kernel, grid_setup, block, 0, s, sa);
  hipLaunchKernelGGL(ray_sort_kernel, dim3(min(grid_setup.x, (unsigned)WS_SORT_BLOCKS)), block, 0, s, sa);
  prof_end(ctx, WS_K_SETUP);
  // 32-bit offsets into the voxel bytes and the record pool where both are below 4 GB (vox_ptr / rec_ptr)
  const bool small = 2ull * vstate_plane_bytes(m->n_tiles) < (1ull << 32) && (uint64_t)m->sub_cap * (SUB_RECS * 8ull) < (1ull << 32);
  prof_begin(ctx, WS_K_MARCH_TAILS);
  if (small)
    hipLaunchKernelGGL(march_tail_kernel<true>, grid_tail, dim3(64 * TAIL_WAVES), 0, s, sa);
  else
    hipLaunchKernelGGL(march_tail_kernel<false>, grid_tail, dim3(64 * TAIL_WAVES), 0, s, sa);
  prof_end(ctx, WS_K_MARCH_TAILS);
  if (!s0)
  {
    prof_begin(ctx, WS_K_MARCH_FREE);
    if (small)
      hipLaunchKernelGGL(march_free_kernel<true>, grid_free, dim3(WS_FREE_THREADS), 0, s, sa);
    else
      hipLaunchKernelGGL(march_free_kernel<false>, grid_free, dim3(WS_FREE_THREADS), 0, s, sa);
    prof_end(ctx, WS_K_MARCH_FREE);
  }

  ResolveArgs ra;
  ra.tile_list = m->tile_list;
  ra.tile_nsub = m->tile_nsub;
  ra.tile_ent = m->tile_ent;
  ra.sub_cap = m->sub_cap;
  ra.tile_dirty = m->tile_dirty;
  ra.recs = m->rec;
  ra.big_keys = m->big_keys;
  ra.big_mask = m->big_slots - 1;
  ra.scan_seq = sa.scan_seq;
  ra.fan_mask = (1u << (sa.rec_fmt >> 8)) - 1u;
  ra.fan_mid = rec_fan_mid((int32_t)(sa.rec_fmt >> 8));
  ra.new_data = m->data[WS_MAP_NEW];
  ra.avg_data = m->data[WS_MAP_AVG];
  ra.vstate = m->vstate;
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
  ra.resolve_stats = m->block_stats + 2 * WS_TAIL_STATS;
  ra.counters = m->counters;
  ra.status = m->status_dev;
  ra.n_tiles = m->n_tiles;
  m->resolve_blocks = RESOLVE_GRID;
  prof_begin(ctx, WS_K_TILE_RESOLVE);
  if (s0)
    hipLaunchKernelGGL((tile_resolve_kernel<true, false>), dim3(RESOLVE_GRID), block, 0, s, ra);
  else if (fuse)
    hipLaunchKernelGGL((tile_resolve_kernel<false, true>), dim3(RESOLVE_GRID), block, 0, s, ra);
  else
    hipLaunchKernelGGL((tile_resolve_kernel<false, false>), dim3(RESOLVE_GRID), block, 0, s, ra);
  prof_end(ctx, WS_K_TILE_RESOLVE);
  m->fused_done = fuse;
  m->prepped = true; // every kernel above has put back what it consumed
  *seq_out = sa.scan_seq;
  WS_HIP(hipGetLastError());
  return WS_OK;
}

// The pool of sub-chunks is sized by estimate (subs_needed): a scan that exhausts it raises the abort flag, the resolve --
// which is enqueued already -- then only puts the scratch back and writes nothing to the maps, and the scan is repeated with
// a larger pool: never an inexact map.  The verdict is in host-mapped memory when the resolve starts, ~0.35 ms after the
// launches (the flag, then the sequence number).  Round 4 waited for it inside ws_tsdf_update*; now the update returns after
// the launches -- like the reference's (update_tsdf.cu:165) -- and whoever takes the map next looks first: ws_register_cloud,
// ws_reg_iterate, ws_sync, the downloads, the next update.  By then the word is there; an aborted scan is repeated from the
// arguments kept in ws_map::pending before the caller's own work is enqueued.  (Paced at the sensor's 10 Hz the round-4 wait
// cost the callback 1.3 ms of its 3.5: the GPU runs the marches at idle clocks there.)
int settle_tsdf(ws_map *m)
{
  if (!m || !m->pending.active.load(std::memory_order_acquire)) return WS_OK; // (the fast path: no lock)
  // Two readers of the reference's caller can get here at once (register_cloud and the shift thread's to_host, both under
  // the SHARED lock): one of them settles the scan, the other waits here until the map is whole.
  std::lock_guard<std::mutex> lock(m->settle_mu);
  if (!m->pending.active.load(std::memory_order_acquire)) return WS_OK;
  ws_context *ctx = m->ctx;
  hipStream_t s = ctx->stream;
  volatile uint32_t *st = m->status_host;
  auto done = [&](int rc) {
    m->pending.active.store(false, std::memory_order_release);
    return rc;
  };
  // the verdict of the scatter numbered `seq`: 0 fine, bit 0 pool exhausted, bit 1 a ray beyond the key range; < 0: error
  auto verdict = [&](uint32_t seq) -> int {
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    while (st[8] != seq)
    {
      if ((++spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50))
      {
        WS_HIP(hipStreamSynchronize(s)); // (a stream busy with much earlier work; the word is there afterwards)
        if (st[8] != seq)
        {
          set_error("TSDF update: the resolve did not report the end of the marches");
          return WS_ERR_INTERNAL;
        }
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return (int)st[9];
  };
  // a larger pool for a scan of n points that has just been aborted for lack of one
  auto grow_pool = [&](size_t n) -> int {
    const unsigned long long need = *reinterpret_cast<volatile unsigned long long *>(m->status_host + 4) & ((1ull << 48) - 1ull);
    uint64_t grow_to = subs_for_scan(m, need, n);
    if (grow_to < (uint64_t)m->sub_cap * 2) grow_to = (uint64_t)m->sub_cap * 2;
    if (grow_to > SUB_ID_LIMIT)
    {
      set_error("TSDF update: the scan needs more than 2^27 record sub-chunks");
      return WS_ERR_CAPACITY;
    }
    // (waits for the stream: the aborted update has drained and put its scratch back; leaves prepped == false: the new hash is
    // filled by the preparation pass, the tile tables are zero already)
    return resize_records(m, grow_to);
  };
  for (;;)
  {
    const int v = verdict(m->pending.seq);
    if (v < 0) return done(v);
    if (v == 0) return done(WS_OK); // the normal case
    if (v & 2)
    {
      // A ray with more steps / a wider fan than the record's key holds for a scan of this many points (until
      // round 6 such a ray was dropped with WS_ERR_RANGE; the reference marches it, update_tsdf.cu:67,107).  The scan is repeated
      // in pieces of 16 384 points -- their records carry the widest split, 65 536 steps and 255 fan steps -- one after the
      // other into new_map: the first piece as the scan itself would have gone, every further one on top of what new_map holds
      // (the non-default route: the fold starts from the stored entry), which is the serial schedule of the whole scan.  Then
      // the integrate over the whole map.  Slow (every candidate of the later pieces is a record, the integrate is dense) and
      // exact; beyond 65 536 steps / 255 fan steps WS_ERR_RANGE remains.
      constexpr size_t PIECE = 16384;
      bool s0 = m->pending.s0;
      for (size_t off = 0; off < m->pending.n; off += PIECE)
      {
        const size_t cnt = std::min(PIECE, m->pending.n - off);
        for (int attempt = 0;; ++attempt)
        {
          uint32_t seq = 0;
          int rc = enqueue_scatter(m, m->scan_dev + 3 * off, cnt, m->pending.pos, m->pending.up, false, s0, &seq);
          if (rc != WS_OK) return done(rc);
          const int pv = verdict(seq);
          if (pv < 0) return done(pv);
          if (pv == 0) break;
          if ((pv & 2) || attempt >= 8)
          {
            set_error("TSDF update: a piece of the scan could not be placed");
            return done(WS_ERR_INTERNAL);
          }
          rc = grow_pool(cnt);
          if (rc != WS_OK) return done(rc);
        }
        m->new_is_default = false; // new_map carries the pieces so far
        s0 = true;
      }
      if (m->pending.integrate_after)
      {
        const int rc = launch_tsdf_integrate(m);
        if (rc != WS_OK) return done(rc);
      }
      return done(WS_OK);
    }
    if (++m->pending.attempts > 8)
    {
      set_error("TSDF update: the scan did not fit the record pool it had just been given");
      return done(WS_ERR_INTERNAL);
    }
    // The repeat reads the copy of the scan the set-up pass of the aborted attempt has left in scan_dev, and takes the route
    // (default / non-default new_map) of the first attempt.
    int rc = grow_pool(m->pending.n);
    if (rc == WS_OK) rc = enqueue_scatter(m, m->scan_dev, m->pending.n, m->pending.pos, m->pending.up, m->pending.fused, m->pending.s0, &m->pending.seq);
    if (rc == WS_OK && m->pending.integrate_after) rc = launch_tsdf_integrate(m);
    if (rc != WS_OK) return done(rc);
  }
}

int launch_tsdf_scatter(ws_map *m, const int32_t *xyz_dev, size_t n, const int32_t scanner_pos[3], const int32_t up[3], bool fused)
{
  // (one scan in flight per map: its pool and its verdict words are the map's)
  int rc = settle_tsdf(m);
  if (rc != WS_OK) return rc;
  ws_context *ctx = m->ctx;
  hipStream_t s = ctx->stream;
  m->fused_done = false;
  m->tail_blocks = 0;
  if (n == 0)
  {
    m->resolve_blocks = 0;
    // nothing listed: a following integrate pass has nothing to do
    WS_HIP(hipMemsetAsync(&m->counters->n_listed, 0, sizeof(uint32_t), s));
    WS_HIP(hipMemsetAsync(&m->counters->n_appended, 0, sizeof(uint32_t), s));
    return WS_OK;
  }
  uint32_t seq = 0;
  const bool s0 = !m->new_is_default;
  rc = enqueue_scatter(m, xyz_dev, n, scanner_pos, up, fused, s0, &seq);
  if (rc != WS_OK) return rc;
  m->pending.seq = seq;
  m->pending.n = n;
  m->pending.s0 = s0;
  for (int k = 0; k < 3; ++k)
  {
    m->pending.pos[k] = scanner_pos[k];
    m->pending.up[k] = up[k];
  }
  m->pending.fused = fused;
  m->pending.integrate_after = false;
  m->pending.attempts = 0;
  m->pending.active.store(true, std::memory_order_release);
  // A scan into a NON-default new_map (the first update after a map came from the host, update_tsdf.cu:135-136) is settled here:
  // the dense integrate that follows consumes new_map's stored entries, and must not run on the leftovers of an aborted scan.
  if (!m->new_is_default) return settle_tsdf(m);
  return WS_OK;
}

} // namespace ws

// tsdf_tiles.hip — LDS-staged scatter for the TSDF update (gfx950).
//
// The global-key scatter (tsdf_update.hip) touches one cache line per candidate: 35 M candidates move ~2 GB
// through the memory system for 0.2 GB of useful data.  This path gives every 8x8x16-voxel tile of the ring
// buffer to ONE workgroup that keeps the tile's order keys in LDS:
//
//   tile_bin_kernel<COUNT>   walk every ray (64 rays x 4 lanes per workgroup) and count, per tile, the
//                            (ray, step-run) records whose candidates can land in it.  Counts are merged in an
//                            LDS hash first: one global atomic per (workgroup, tile) instead of per record —
//                            a shared counter hit 100 k times costs milliseconds on this chip.
//   tile_scan_kernel         exclusive scan of the counts -> record offsets and the list of work items
//                            (a tile with more than TILE_PMAX records is split into several items).
//   tile_bin_kernel<FILL>    the same walk, records written to their tile's segment.
//   tile_scatter_kernel      one workgroup per work item: march the recorded step runs (ws_march.h, the exact
//                            arithmetic of update_tsdf.cu:67-125), LDS atomicMin into kpos/kneg of the tile,
//                            then either
//                              - (whole tile in one item) resolve every voxel locally in the canonical serial
//                                order — contested voxels by re-marching the tile's records round by round — and
//                                integrate straight into avg_map (cu_avg_tsdf_krnl fused into the write-back), or
//                              - (split tile) merge the LDS keys into the global key arrays with one coalesced
//                                atomicMin per touched voxel; those few tiles (all next to the sensor) finish on
//                                the global path of tsdf_update.hip.
//
// HBM then only sees the ray records and one read-modify-write of the touched voxels of avg_map.
// Results are bit-identical to the global path and to oracle/ws_oracle.c.
#include <cstddef>

#include "ws_march.h"
#include "ws_tiles.h"

namespace ws
{
constexpr int TB_X = 3, TB_Y = 3, TB_Z = 4; // tile = 8 x 8 x 16 voxels of ring-index space
constexpr int TSX = 1 << TB_X, TSY = 1 << TB_Y, TSZ = 1 << TB_Z;
constexpr int TILE_VOX = TSX * TSY * TSZ; // 1024
constexpr uint32_t TILE_PMAX = 256;       // records per work item (their ray set-ups are staged in LDS)
constexpr int BIN_RAYS = 64, BIN_LANES = 4;
constexpr int HASH_SLOTS = 512, HASH_PROBES = 24;
constexpr uint32_t HASH_EMPTY = 0xffffffffu;
constexpr int STAGE_CAP = 4096;

__device__ __forceinline__ void ring_coords(const MapParams &m, int32_t vx, int32_t vy, int32_t vz, int32_t &xi, int32_t &yi, int32_t &zi)
{
  xi = ring(vx - m.pos[0] + m.offset[0] + m.size[0], m.size[0]);
  yi = ring(vy - m.pos[1] + m.offset[1] + m.size[1], m.size[1]);
  zi = ring(vz - m.pos[2] + m.offset[2] + m.size[2], m.size[2]);
}
__device__ __forceinline__ uint32_t tile_of(const TileGrid &g, int32_t xi, int32_t yi, int32_t zi)
{
  return (uint32_t)(((xi >> TB_X) * g.nty + (yi >> TB_Y)) * g.ntz + (zi >> TB_Z));
}
__device__ __forceinline__ int32_t local_of(int32_t xi, int32_t yi, int32_t zi)
{
  return ((xi & (TSX - 1)) << (TB_Y + TB_Z)) | ((yi & (TSY - 1)) << TB_Z) | (zi & (TSZ - 1));
}

// record = ray(20) | first step(16) | number of steps(8)
__device__ __forceinline__ uint64_t make_record(uint32_t ray, int32_t k0, int32_t n) { return ((uint64_t)ray << 24) | ((uint64_t)(uint32_t)k0 << 8) | (uint64_t)(uint32_t)n; }

enum
{
  BIN_COUNT = 0,
  BIN_FILL = 1
};

// ---------------------------------------------------------------------------------------------------------
// binning
// ---------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void tile_bin_kernel(TileArgs a)
{
  __shared__ uint32_t hash_key[HASH_SLOTS];
  __shared__ uint32_t hash_cnt[HASH_SLOTS];
  __shared__ uint32_t hash_base[HASH_SLOTS];
  __shared__ uint64_t stage_rec[MODE == BIN_FILL ? STAGE_CAP : 1];
  __shared__ uint32_t stage_meta[MODE == BIN_FILL ? STAGE_CAP : 1];
  __shared__ uint32_t stage_n;

  for (int i = threadIdx.x; i < HASH_SLOTS; i += 256)
  {
    hash_key[i] = HASH_EMPTY;
    hash_cnt[i] = 0;
  }
  if (threadIdx.x == 0) stage_n = 0;
  __syncthreads();

  const MarchFrame &f = a.frame;
  const uint32_t ray = blockIdx.x * BIN_RAYS + (threadIdx.x >> 2);
  const int32_t c = threadIdx.x & (BIN_LANES - 1);

  // emit one record: merge into the workgroup's hash (count) / stage it for the write-out (fill)
  auto emit = [&](uint32_t tile, int32_t kmin, int32_t kmax) {
    const uint64_t rec = make_record(ray, kmin, kmax - kmin + 1);
    int slot = -1;
    uint32_t h = (tile * 2654435761u) >> 23; // 9 bits
    for (int p = 0; p < HASH_PROBES; ++p)
    {
      const uint32_t prev = atomicCAS(&hash_key[h], HASH_EMPTY, tile);
      if (prev == HASH_EMPTY || prev == tile)
      {
        slot = (int)h;
        break;
      }
      h = (h + 1) & (HASH_SLOTS - 1);
    }
    if (MODE == BIN_COUNT)
    {
      if (slot >= 0)
        atomicAdd(&hash_cnt[slot], 1u);
      else
        atomicAdd(&a.tile_count[tile], 1u);
    }
    else
    {
      uint32_t si = STAGE_CAP;
      if (slot >= 0) si = atomicAdd(&stage_n, 1u);
      if (si < STAGE_CAP)
      {
        const uint32_t rank = atomicAdd(&hash_cnt[slot], 1u);
        stage_rec[si] = rec;
        stage_meta[si] = ((uint32_t)slot << 16) | (rank & 0xffffu);
      }
      else
      {
        // hash or staging area full: place the record directly (any free position of the tile's segment is fine)
        const uint32_t pos = a.tile_offset[tile] + atomicAdd(&a.tile_cursor[tile], 1u);
        if (pos < a.records_cap) a.records[pos] = rec;
      }
    }
  };

  if (ray < a.n)
  {
    const RaySetup r = a.rays[ray];
    const int32_t ch = (r.steps + BIN_LANES - 1) / BIN_LANES;
    const int32_t k0 = c * ch;
    const int32_t k1 = min(k0 + ch, r.steps);

    // up to 4 open runs (tile, first step, last step)
    uint32_t otile[4] = {HASH_EMPTY, HASH_EMPTY, HASH_EMPTY, HASH_EMPTY};
    int32_t omin[4] = {0, 0, 0, 0}, omax[4] = {0, 0, 0, 0};
    auto touch = [&](uint32_t tile, int32_t k) {
      int found = -1, freeslot = -1, oldest = 0;
#pragma unroll
      for (int s = 0; s < 4; ++s)
      {
        if (otile[s] == tile) found = s;
        if (otile[s] == HASH_EMPTY && freeslot < 0) freeslot = s;
        if (omax[s] < omax[oldest]) oldest = s;
      }
      if (found < 0)
      {
        int s = freeslot;
        if (s < 0)
        {
          s = oldest;
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (q == s) emit(otile[q], omin[q], omax[q]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q == s)
          {
            otile[q] = tile;
            omin[q] = k;
            omax[q] = k;
          }
      }
      else
      {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q == found)
          {
            if (k - omin[q] >= 250)
            {
              emit(otile[q], omin[q], omax[q]);
              omin[q] = k;
            }
            omax[q] = k;
          }
      }
    };

    const int64_t MR = MATRIX_RESOLUTION;
    for (int32_t k = k0; k < k1; ++k)
    {
      const int32_t len = 1 + k * f.half;
      const int32_t projx = wadd(f.posx, div_trunc(wmul(r.dx, len), r.div_m, r.div_k, r.distance));
      const int32_t projy = wadd(f.posy, div_trunc(wmul(r.dy, len), r.div_m, r.div_k, r.distance));
      const int32_t projz = wadd(f.posz, div_trunc(wmul(r.dz, len), r.div_m, r.div_k, r.distance));
      const int32_t ixx = div_trunc(projx, f.rM, f.rK, f.res), iyy = div_trunc(projy, f.rM, f.rK, f.res), izz = div_trunc(projz, f.rM, f.rK, f.res);
      // the reference only produces candidates for steps whose on-ray voxel is inside the map (update_tsdf.cu:76-79);
      // the "same column as the previous step" skip (:71) is ignored here — a run may contain empty steps
      if (!in_bounds(f.map, ixx, iyy, izz)) continue;
      int32_t xi, yi, zi;
      ring_coords(f.map, ixx, iyy, izz, xi, yi, zi);
      const int32_t delta_z = wmul(DZ_PER_DISTANCE, len) / MATRIX_RESOLUTION;
      // every candidate of this step lies within delta_z + 1 mm of proj on each axis; if that is less than one
      // voxel and the on-ray voxel is not on a tile face, all of them fall into the on-ray voxel's tile
      const bool face = (xi & (TSX - 1)) == 0 || (xi & (TSX - 1)) == TSX - 1 || xi == f.map.size[0] - 1 ||
                        (yi & (TSY - 1)) == 0 || (yi & (TSY - 1)) == TSY - 1 || yi == f.map.size[1] - 1 ||
                        (zi & (TSZ - 1)) == 0 || (zi & (TSZ - 1)) == TSZ - 1 || zi == f.map.size[2] - 1;
      if (!face && delta_z + 2 < f.res)
      {
        touch(tile_of(a.grid, xi, yi, zi), k);
        continue;
      }
      // exact fan (update_tsdf.cu:101-125) to see which tiles the candidates reach
      const int32_t iter_steps = (delta_z * 2) / f.res + 1;
      const int32_t lowx = wsub(projx, (int32_t)(wmul64(delta_z, (int64_t)r.ivx) / MR));
      const int32_t lowy = wsub(projy, (int32_t)(wmul64(delta_z, (int64_t)r.ivy) / MR));
      const int32_t lowz = wsub(projz, (int32_t)(wmul64(delta_z, (int64_t)r.ivz) / MR));
      for (int32_t step = 0; step < iter_steps; ++step)
      {
        const int64_t sm = (int64_t)wmul(step, f.res);
        const int32_t vx = div_trunc(wadd(lowx, (int32_t)(wmul64(sm, (int64_t)r.ivx) / MR)), f.rM, f.rK, f.res);
        const int32_t vy = div_trunc(wadd(lowy, (int32_t)(wmul64(sm, (int64_t)r.ivy) / MR)), f.rM, f.rK, f.res);
        const int32_t vz = div_trunc(wadd(lowz, (int32_t)(wmul64(sm, (int64_t)r.ivz) / MR)), f.rM, f.rK, f.res);
        if (!in_bounds(f.map, vx, vy, vz)) continue;
        int32_t cx, cy, cz;
        ring_coords(f.map, vx, vy, vz, cx, cy, cz);
        touch(tile_of(a.grid, cx, cy, cz), k);
      }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (otile[s] != HASH_EMPTY) emit(otile[s], omin[s], omax[s]);
  }
  __syncthreads();

  // one global atomic per (workgroup, tile)
  for (int i = threadIdx.x; i < HASH_SLOTS; i += 256)
  {
    const uint32_t tile = hash_key[i];
    const uint32_t cnt = hash_cnt[i];
    if (tile != HASH_EMPTY && cnt)
    {
      if (MODE == BIN_COUNT)
        atomicAdd(&a.tile_count[tile], cnt);
      else
        hash_base[i] = a.tile_offset[tile] + atomicAdd(&a.tile_cursor[tile], cnt);
    }
  }
  if (MODE == BIN_FILL)
  {
    __syncthreads();
    const uint32_t staged = min(stage_n, (uint32_t)STAGE_CAP);
    for (uint32_t i = threadIdx.x; i < staged; i += 256)
    {
      const uint32_t meta = stage_meta[i];
      const uint32_t pos = hash_base[meta >> 16] + (meta & 0xffffu);
      if (pos < a.records_cap) a.records[pos] = stage_rec[i];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// scan: counts -> offsets + work items (single workgroup of 1024 lanes)
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t *wave_sums, uint32_t &total)
{
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const uint32_t y = __shfl_up(x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wave_sums[wave] = x;
  __syncthreads();
  if (wave == 0)
  {
    uint32_t w = lane < 16 ? wave_sums[lane] : 0;
#pragma unroll
    for (int d = 1; d < 16; d <<= 1)
    {
      const uint32_t y = __shfl_up(w, d, 64);
      if (lane >= d) w += y;
    }
    if (lane < 16) wave_sums[lane] = w; // inclusive
  }
  __syncthreads();
  total = wave_sums[15];
  const uint32_t before = wave ? wave_sums[wave - 1] : 0;
  __syncthreads();
  return before + x - v;
}

__global__ __launch_bounds__(1024) void tile_scan_kernel(TileArgs a)
{
  __shared__ uint32_t wave_sums[16];
  const int64_t per = (a.n_tiles + 1023) / 1024;
  const int64_t lo = (int64_t)threadIdx.x * per;
  const int64_t hi = lo + per < a.n_tiles ? lo + per : a.n_tiles;
  uint32_t cnt_sum = 0, item_sum = 0;
  for (int64_t i = lo; i < hi; ++i)
  {
    const uint32_t cnt = a.tile_count[i];
    cnt_sum += cnt;
    item_sum += (cnt + TILE_PMAX - 1) / TILE_PMAX;
  }
  uint32_t total_cnt = 0, total_items = 0;
  uint32_t off = block_exclusive_scan_1024(cnt_sum, wave_sums, total_cnt);
  uint32_t wi = block_exclusive_scan_1024(item_sum, wave_sums, total_items);
  const bool fits = total_cnt <= a.records_cap && total_items <= a.work_cap;
  for (int64_t i = lo; i < hi; ++i)
  {
    const uint32_t cnt = a.tile_count[i];
    a.tile_offset[i] = off;
    a.tile_count[i] = 0;
    a.tile_cursor[i] = 0;
    const uint32_t items = (cnt + TILE_PMAX - 1) / TILE_PMAX;
    if (fits)
      for (uint32_t j = 0; j < items; ++j)
      {
        const uint32_t first = off + j * TILE_PMAX;
        const uint32_t n = min(TILE_PMAX, cnt - j * TILE_PMAX);
        a.work[wi + j] = make_uint4((uint32_t)i, first, n, items > 1 ? 1u : 0u);
      }
    wi += items;
    off += cnt;
  }
  if (threadIdx.x == 0)
  {
    a.tile_state->work_count = fits ? total_items : 0;
    a.tile_state->total_records = total_cnt;
    if (!fits) atomicOr(&a.counters->error, 1u);
  }
}

// ---------------------------------------------------------------------------------------------------------
// scatter + resolve + integrate, one workgroup per work item
// ---------------------------------------------------------------------------------------------------------
enum
{
  VS_UNTOUCHED = 0,
  VS_FINAL = 1,
  VS_ACTIVE = 2
};

template <bool FUSED>
__global__ __launch_bounds__(256) void tile_scatter_kernel(TileArgs a)
{
  __shared__ uint64_t kpos[TILE_VOX];
  __shared__ uint64_t kneg[TILE_VOX];
  __shared__ uint32_t result[TILE_VOX];
  __shared__ uint8_t vstate[TILE_VOX];
  __shared__ uint32_t n_active;
  __shared__ uint32_t step_prefix[TILE_PMAX + 1]; // exclusive prefix of the records' step counts
  __shared__ uint32_t wave_sums[4];
  __shared__ RaySetup rays_sh[TILE_PMAX];         // one global round trip per item instead of two per record and lane
  __shared__ uint64_t recs_sh[TILE_PMAX];

  const MarchFrame &f = a.frame;
  const int32_t weight_epsilon = f.weight_epsilon;
  const uint32_t n_work = a.tile_state->work_count;
  uint32_t contested_total = 0;
#ifdef WS_TILE_TIMING
  long long tacc[6] = {0, 0, 0, 0, 0, 0};
  long long tlast = wall_clock64();
  unsigned long long nrounds = 0, nitems = 0, nrecs = 0, nsteps = 0;
#define TSTAMP(i) { long long now__ = wall_clock64(); tacc[i] += now__ - tlast; tlast = now__; }
#else
#define TSTAMP(i)
#endif

  for (uint32_t w = blockIdx.x; w < n_work; w += gridDim.x)
  {
    const uint4 item = a.work[w];
    const uint32_t tile = item.x, first = item.y, n_rec = item.z;
    const bool split = item.w != 0;
    const int32_t tz = (int32_t)(tile % (uint32_t)a.grid.ntz);
    const int32_t ty = (int32_t)((tile / (uint32_t)a.grid.ntz) % (uint32_t)a.grid.nty);
    const int32_t tx = (int32_t)(tile / ((uint32_t)a.grid.ntz * (uint32_t)a.grid.nty));

    for (int v = threadIdx.x; v < TILE_VOX; v += 256)
    {
      kpos[v] = KEY_INF;
      kneg[v] = KEY_INF;
      vstate[v] = VS_UNTOUCHED;
    }
    if (threadIdx.x == 0) n_active = 0;
    __syncthreads();
    TSTAMP(0);

    // ---- balance: the item's records hold 1..250 steps each; lanes get equal shares of STEPS, not of records
    // (one record per lane left ~70 % of the lanes idle).  step_prefix[i] = steps of records 0..i-1.
    {
      // n_rec <= TILE_PMAX == 256: one record per lane
      const uint32_t i = threadIdx.x;
      uint64_t rec = 0;
      if (i < n_rec)
      {
        rec = a.records[first + i];
        recs_sh[i] = rec;
        rays_sh[i] = a.rays[(uint32_t)(rec >> 24)];
      }
      const uint32_t cnt = (uint32_t)(rec & 0xffu);
      uint32_t x = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1)
      {
        const uint32_t y = __shfl_up(x, d, 64);
        if ((threadIdx.x & 63) >= d) x += y;
      }
      if ((threadIdx.x & 63) == 63) wave_sums[threadIdx.x >> 6] = x;
      __syncthreads();
      uint32_t before = 0;
      for (int wv = 0; wv < (int)(threadIdx.x >> 6); ++wv) before += wave_sums[wv];
      if (i < n_rec) step_prefix[i] = before + x - cnt;
      if (threadIdx.x == 0) step_prefix[n_rec] = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
      __syncthreads();
    }
    const uint32_t total_steps = step_prefix[n_rec];
    const uint32_t share = (total_steps + 255u) / 256u;
    const uint32_t my_lo = min(threadIdx.x * share, total_steps), my_hi = min(my_lo + share, total_steps);
    // first record whose step range reaches beyond my_lo
    uint32_t my_rec = 0;
    {
      uint32_t lo = 0, hi = n_rec; // invariant: step_prefix[lo] <= my_lo < step_prefix[hi]
      while (hi - lo > 1)
      {
        const uint32_t mid = (lo + hi) >> 1;
        if (step_prefix[mid] <= my_lo)
          lo = mid;
        else
          hi = mid;
      }
      my_rec = lo;
    }
    // call body(ray, setup, ka, kb) for every (partial) record of this lane's share
    auto for_my_steps = [&](auto &&body) {
      uint32_t pos = my_lo, rec_i = my_rec;
      while (pos < my_hi)
      {
        const uint64_t rec = recs_sh[rec_i];
        const uint32_t ray = (uint32_t)(rec >> 24);
        const int32_t rk0 = (int32_t)((rec >> 8) & 0xffffu);
        const uint32_t rec_lo = step_prefix[rec_i], rec_hi = step_prefix[rec_i + 1];
        const uint32_t seg_hi = min(rec_hi, my_hi);
        const RaySetup r = rays_sh[rec_i];
        body(ray, r, rk0 + (int32_t)(pos - rec_lo), rk0 + (int32_t)(seg_hi - rec_lo));
        pos = seg_hi;
        rec_i += 1;
      }
    };
    TSTAMP(5);

    // ---- pass 1: every candidate of the recorded runs that lands in this tile -> LDS keys
    for_my_steps([&](uint32_t ray, const RaySetup &r, int32_t ka, int32_t kb) {
      march_steps(f, r, ka, kb, [&](int32_t k, int32_t step, int32_t vx, int32_t vy, int32_t vz, int32_t value, bool positive) {
        int32_t xi, yi, zi;
        ring_coords(f.map, vx, vy, vz, xi, yi, zi);
        if ((xi >> TB_X) != tx || (yi >> TB_Y) != ty || (zi >> TB_Z) != tz) return;
        const int32_t v = local_of(xi, yi, zi);
        const uint64_t t = order_key(ray, k, step);
        if (positive)
          atomicMin((unsigned long long *)&kpos[v], (unsigned long long)make_kpos(t, value));
        else
          atomicMin((unsigned long long *)&kneg[v], (unsigned long long)make_kneg(t, value));
      });
    });
    __syncthreads();
    TSTAMP(1);
#ifdef WS_TILE_TIMING
    nitems += 1; nrecs += n_rec;
#endif

    // voxel v of the tile <-> linear index in the maps
    auto global_index = [&](int v, bool &inside) -> int64_t {
      const int32_t xi = (tx << TB_X) + (v >> (TB_Y + TB_Z));
      const int32_t yi = (ty << TB_Y) + ((v >> TB_Z) & (TSY - 1));
      const int32_t zi = (tz << TB_Z) + (v & (TSZ - 1));
      inside = xi < f.map.size[0] && yi < f.map.size[1] && zi < f.map.size[2];
      return ((int64_t)xi * f.map.size[1] + yi) * (int64_t)f.map.size[2] + zi;
    };

    if (split)
    {
      // ---- this tile is shared with other work items: merge into the global key arrays (coalesced along z)
      for (int v = threadIdx.x; v < TILE_VOX; v += 256)
      {
        const uint64_t kp = kpos[v], kn = kneg[v];
        if (kp == KEY_INF && kn == KEY_INF) continue;
        bool inside;
        const int64_t idx = global_index(v, inside);
        if (!inside) continue;
        if (kp != KEY_INF) atomicMin((unsigned long long *)&a.kpos[idx], (unsigned long long)kp);
        if (kn != KEY_INF) atomicMin((unsigned long long *)&a.kneg[idx], (unsigned long long)kn);
        a.dirty[idx >> TILE_SHIFT] = 1;
      }
      __syncthreads();
      continue;
    }

    // ---- classification (same rule as resolve_kernel in tsdf_update.hip)
    for (int v = threadIdx.x; v < TILE_VOX; v += 256)
    {
      const uint64_t kp = kpos[v], kn = kneg[v];
      if (kp == KEY_INF && kn == KEY_INF) continue;
      int32_t value;
      bool positive = false, decided = true;
      if (kp != KEY_INF)
      {
        value = (int32_t)(int16_t)(kp & 0xffffu);
        positive = true;
        if (kn != KEY_INF)
        {
          const int32_t ap = value < 0 ? -value : value;
          if (ap > (int32_t)(kn >> 45)) decided = false; // an earlier negative candidate may have blocked it
        }
      }
      else
      {
        const int32_t an = (int32_t)(kn >> 45);
        value = (kn & 1ull) ? -an : an;
      }
      if (decided)
      {
        const int32_t wgt = tsdf_weight(value, f.tau, weight_epsilon);
        result[v] = pack_entry(value, positive ? wgt : -wgt);
        vstate[v] = VS_FINAL;
      }
      else
      {
        // ordered fold from the empty state (tau, 0): kneg becomes (t_last + 1) << 16 | |value| of the state,
        // result the state's entry so far
        vstate[v] = VS_ACTIVE;
        result[v] = pack_entry(f.tau, 0);
        kneg[v] = (uint64_t)(uint32_t)f.tau; // nothing accepted yet
        atomicAdd(&n_active, 1u);
      }
    }
    __syncthreads();
    contested_total += (threadIdx.x == 0) ? n_active : 0;
    TSTAMP(2);

    // ---- contested voxels: one accepted candidate per round, in key order (atomic_tsdf_min's rule,
    // cuda/util.h:70-102: accept iff stored weight <= 0 and |new| <= |stored|)
    while (n_active != 0)
    {
      for (int v = threadIdx.x; v < TILE_VOX; v += 256)
        if (vstate[v] == VS_ACTIVE) kpos[v] = KEY_INF;
      __syncthreads();
      for_my_steps([&](uint32_t ray, const RaySetup &r, int32_t ka, int32_t kb) {
        march_steps(f, r, ka, kb, [&](int32_t k, int32_t step, int32_t vx, int32_t vy, int32_t vz, int32_t value, bool positive) {
          int32_t xi, yi, zi;
          ring_coords(f.map, vx, vy, vz, xi, yi, zi);
          if ((xi >> TB_X) != tx || (yi >> TB_Y) != ty || (zi >> TB_Z) != tz) return;
          const int32_t v = local_of(xi, yi, zi);
          if (vstate[v] != VS_ACTIVE) return;
          const uint64_t st = kneg[v];
          const uint64_t t = order_key(ray, k, step);
          const int32_t av = value < 0 ? -value : value;
          if (t + 1 <= (st >> 16) || av > (int32_t)(st & 0xffffu)) return; // already folded, or rejected by the current state
          atomicMin((unsigned long long *)&kpos[v], (unsigned long long)((t << 17) | (positive ? 0ull : (1ull << 16)) | ((uint32_t)value & 0xffffu)));
        });
      });
      __syncthreads();
      if (threadIdx.x == 0) n_active = 0;
      __syncthreads();
      for (int v = threadIdx.x; v < TILE_VOX; v += 256)
      {
        if (vstate[v] != VS_ACTIVE) continue;
        const uint64_t best = kpos[v];
        if (best == KEY_INF)
        {
          vstate[v] = VS_FINAL; // nothing more is accepted: the state is the result
          continue;
        }
        const int32_t value = (int32_t)(int16_t)(best & 0xffffu);
        const int32_t av = value < 0 ? -value : value;
        const int32_t wgt = tsdf_weight(value, f.tau, weight_epsilon);
        if (best & (1ull << 16))
        {
          result[v] = pack_entry(value, -wgt);
          kneg[v] = (((best >> 17) + 1) << 16) | (uint64_t)(uint32_t)av;
          atomicAdd(&n_active, 1u);
        }
        else
        {
          result[v] = pack_entry(value, wgt); // a positive weight freezes the voxel
          vstate[v] = VS_FINAL;
        }
      }
      __syncthreads();
#ifdef WS_TILE_TIMING
      nrounds += 1;
#endif
    }
    TSTAMP(3);

    // ---- write-back, rows of 16 voxels along z (64 B) per 16 lanes
    for (int v = threadIdx.x; v < TILE_VOX; v += 256)
    {
      if (vstate[v] == VS_UNTOUCHED) continue;
      const uint32_t fresh = result[v];
      if (fresh == pack_entry(f.tau, 0)) continue; // contested voxel whose every candidate was rejected cannot happen; guard
      bool inside;
      const int64_t idx = global_index(v, inside);
      if (!inside) continue;
      if (FUSED)
      {
        // cu_avg_tsdf_krnl (update_tsdf.cu:19-34) fused into the tile write-back; new_map stays (tau, 0)
        const uint32_t existing = a.avg_data[idx];
        const uint32_t updated = integrate_entry(existing, fresh, a.max_weight);
        if (updated != existing) a.avg_data[idx] = updated;
      }
      else
      {
        a.new_data[idx] = fresh;
        a.dirty[idx >> TILE_SHIFT] = 1;
      }
    }
    __syncthreads();
    TSTAMP(4);
  }
#ifdef WS_TILE_TIMING
  if (threadIdx.x == 0 && (blockIdx.x == 5 || blockIdx.x == 1000))
    printf("tile blk %d: items %llu recs %llu rounds %llu mysteps %llu | ticks(10ns): init %lld loads %lld march %lld classify %lld rounds %lld writeback %lld\n", blockIdx.x, nitems, nrecs,
           nrounds, nsteps, tacc[0], tacc[5], tacc[1], tacc[2], tacc[3], tacc[4]);
#endif
  if (threadIdx.x == 0 && contested_total) atomicAdd(&a.tile_state->contested, contested_total);
}

int launch_tile_path(ws_map *m, const TileArgs &a, size_t n, bool fused)
{
  ws_context *ctx = m->ctx;
  hipStream_t s = ctx->stream;
  const dim3 grid_bin((unsigned)((n + BIN_RAYS - 1) / BIN_RAYS));
  prof_begin(ctx, WS_K_TILE_BIN);
  hipLaunchKernelGGL((tile_bin_kernel<BIN_COUNT>), grid_bin, dim3(256), 0, s, a);
  hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, s, a);
  hipLaunchKernelGGL((tile_bin_kernel<BIN_FILL>), grid_bin, dim3(256), 0, s, a);
  prof_end(ctx, WS_K_TILE_BIN);
  prof_begin(ctx, WS_K_TILE_SCATTER);
  if (fused)
    hipLaunchKernelGGL((tile_scatter_kernel<true>), dim3(TILE_GRID_BLOCKS), dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL((tile_scatter_kernel<false>), dim3(TILE_GRID_BLOCKS), dim3(256), 0, s, a);
  prof_end(ctx, WS_K_TILE_SCATTER);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

} // namespace ws

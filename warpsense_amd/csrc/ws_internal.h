// ws_internal.h — shared host/device definitions of libwarpsense_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "warpsense_hip.h"

namespace ws
{
constexpr int MATRIX_RESOLUTION = 32768; // include/warpsense/consts.h:12-13
constexpr int WEIGHT_RESOLUTION = 64;    // include/warpsense/consts.h:9-10
constexpr int DZ_PER_DISTANCE = 100;     // (int)(tan(45/128 deg)/2 * 32768), update_tsdf.cu:49-50 (checked on the host at load)
constexpr size_t MAX_SCAN_POINTS = 1000000; // update_tsdf.h:33
constexpr int WS_FAN_TABLE = 264;           // ws_map::fan_steps: 256 fan steps + the tile grid's division constants

// ---- tiles: 4 x 4 x 64 voxels of the ring buffer's STORAGE index space (sx >> 2, sy >> 2, sz >> 6) ----
// (tall: a wave's accesses are four 256-byte z-runs, and the vertical fan of a LiDAR azimuth falls into few tiles;
// measured against 8x8x16: tile resolve 265 -> 189 us, sparse integrate 239 -> 130 us)
// A tile is the unit of the scatter's exact resolve (one workgroup folds all candidates of a tile in LDS) and of
// the sparse integrate.  Storage space, not world space: a tile's z-runs are contiguous in memory and never
// straddle the ring seam.
#ifndef WS_TILE_XB
#define WS_TILE_XB 2
#define WS_TILE_YB 2
#define WS_TILE_ZB 6
#endif
constexpr int TILE_XB = WS_TILE_XB, TILE_YB = WS_TILE_YB, TILE_ZB = WS_TILE_ZB;
static_assert(TILE_XB + TILE_YB + TILE_ZB == 10 && TILE_ZB >= 2, "a tile is 1024 voxels: 256 threads x 4 consecutive z");
constexpr int TILE_VOXELS = 1 << (TILE_XB + TILE_YB + TILE_ZB); // 1024
constexpr uint64_t KEY_INF = ~0ull;
// the voxel bytes are two planes in one allocation: [0] keyed / touched / free-space hit (VOX_*), [1] "an off-ray free-space
// candidate (tau, -weight) landed here" (blind, idempotent byte stores of the free pass)
// Where a voxel's byte lives: tile-major, and inside a tile in BRICKS of 4 x 4 x 8 voxels (128 bytes, z fastest inside a brick,
// the eight bricks of a tile one after the other): byte = tile * 1024 + vbrick(voxel in tile).  Rays of a LiDAR ring sweep
// horizontally: with one byte per voxel in the maps' own z-fastest order every voxel column is a cache line of its own and a
// wave's 64 candidates are 64 lines (2.3 CU-cycles each from the L2, 9 from beyond: tools/ta_bench.hip -- round 5's free pass,
// 28 % fewer instructions than round 4's, took the same 118 us because of them); in bricks horizontal neighbours share lines,
// a ray stays in a brick for four column changes, and a tile's bytes are ONE contiguous kilobyte for the resolve.
__host__ __device__ inline size_t vstate_plane_bytes(int64_t n_tiles) { return ((size_t)n_tiles * 1024u + 16 + 255) & ~(size_t)255; }
__host__ __device__ inline uint32_t vbrick(uint32_t local /* lx(2) ly(2) lz(6), local_of() */)
{
  return ((local & 0x38u) << 4) | ((local >> 3) & 0x78u) | (local & 7u);
}
constexpr uint32_t WS_TAIL_STATS = 65536; // per-workgroup slots of the tail march: records, flush groups (1 000 000 points / 64 rays x up to 2 workgroups), then 2 per resolve workgroup
constexpr uint32_t WS_BLOCK_STATS = 2 * WS_TAIL_STATS + 2 * 4096 + WS_TAIL_STATS; // (+ start ticks per tail workgroup of a -DWS_TAIL_TIMING build)

// ring-buffer parameters passed BY VALUE to kernels (the reference chases three device pointers per
// access, device_map.h:93-101)
struct MapParams
{
  int32_t size[3];
  int32_t pos[3];
  int32_t offset[3];
};

// ---- candidate records ---------------------------------------------------------------------
// One scatter target of a ray tail (a write_tsdf_min call, update_tsdf.cu:107-125) is ONE 64-bit word:
//   t = point | ray step | fan (38 bits) | value(16) | voxel in tile(10)
// `fan` = fan step - mid + MID (mid: the on-ray fan step, update_tsdf.cu:104; MID = 2^(F-1) - 1): ascending like the fan step
// itself, and the weight is negated iff fan != MID (update_tsdf.cu:118-121) -- no separate sign-of-weight bit.  The top 38 bits
// are unique per candidate, so ascending records == canonical serial order (the low bits never decide).
// The 38 bits are shared out PER SCAN (round 5; 20 | 13 | 5 for every scan before -- a ray of more than 8192 steps or 31 fan
// steps was dropped with WS_ERR_RANGE, which the reference has no counterpart of): a scan of n points needs P = ceil(log2 n) bits
// for the point, the rest goes to the step (up to 16 bits: 65 536 steps) and the fan (5 to 8 bits: 31 to 255 fan steps).  The
// reference's own 131 072-point scans admit 32 768 steps and 63 fan steps (40 m rays at 5 mm); a million points 8192 / 31 as
// before.  rec_format() is the one place that decides; the kernels get S and F as scalars.
constexpr int REC_VOX_BITS = 10, REC_VALUE_SHIFT = 10, REC_T_SHIFT = 26;
constexpr int T_BITS = 38;
constexpr uint64_t T_MASK = (1ull << T_BITS) - 1;
struct RecFormat
{
  int32_t S, F; // bits of the ray step and of the fan field
};
__host__ __device__ inline RecFormat rec_format(uint64_t n_points)
{
  int P = 1;
  while ((1ull << P) < n_points) ++P; // point < n <= 2^P
  const int R = T_BITS - P;            // >= 18 for n <= 2^20 (MAX_SCAN_POINTS)
  RecFormat f;
  f.F = 5 + (R - 18) / 3;
  f.S = R - f.F;
  if (f.S > 16) // (the marches carry a step in 16 bits next to the lane)
  {
    f.S = 16;
    f.F = R - 16 > 8 ? 8 : R - 16;
  }
  return f;
}
__host__ __device__ inline int32_t rec_max_steps(const RecFormat &f) { return 1 << f.S; }
__host__ __device__ inline int32_t rec_max_fan(const RecFormat &f) { return (1 << f.F) - 1; } // iter_steps the fan field holds
__host__ __device__ inline uint32_t rec_fan_mid(int32_t F) { return (1u << (F - 1)) - 1u; }
__host__ __device__ inline uint64_t make_rec(uint32_t point, int32_t step, int32_t fan_minus_mid, int32_t value, uint32_t local, int32_t S, int32_t F)
{
  // built as two 32-bit halves (the shifts are scan constants in scalar registers; 64-bit shifts by a variable are three
  // instructions each on gfx950): u = step | fan is below 2^24, t = point | u is 38 bits, the record's high word is t >> 6
  const uint32_t u = ((uint32_t)step << F) | ((uint32_t)fan_minus_mid + rec_fan_mid(F));
  const uint32_t hi = (point << (S + F - 6)) | (u >> 6);                                                                  // S + F >= 18
  const uint32_t lo = (u << REC_T_SHIFT) | (((uint32_t)value & 0xffffu) << REC_VALUE_SHIFT) | local;                      // (the top bits of u leave the word)
  return ((uint64_t)hi << 32) | lo;
}
__host__ __device__ inline int32_t rec_value(uint64_t rec) { return (int32_t)(int16_t)(uint16_t)(rec >> REC_VALUE_SHIFT); }
__host__ __device__ inline uint32_t rec_local(uint64_t rec) { return (uint32_t)rec & ((1u << REC_VOX_BITS) - 1u); }
__host__ __device__ inline bool rec_negative(uint64_t rec, uint32_t fan_mask, uint32_t fan_mid) { return (((uint32_t)(rec >> REC_T_SHIFT)) & fan_mask) != fan_mid; }

// Records live in SUB-CHUNKS of 32 (256 bytes) that belong to one tile each.  A wave of the tail march owns a run of
// sub-chunk ids (its share of the block its workgroup took from the pool, refilled 32 at a time); the first record a wave
// makes for a tile opens a sub-chunk of that run, and every later one goes straight to its place there: position = the
// record's rank among the wave's records of the tile, counted in LDS.  HBM sees a record once, where it stays.  When the wave
// is through (or its LDS bookkeeping is full) it publishes its sub-chunks: ONE atomic per tile on tile_nsub[tile] reserves
// places in the tile's entry table, tile_ent[tile][j] = id << 5 | (records - 1).  Entries beyond TILE_DIRECT go through a
// hash (tile, j) -> entry + 1.  Nobody ever waits for anybody: the resolve -- a later kernel -- reads what is there.
constexpr int SUB_BITS = 5, SUB_RECS = 1 << SUB_BITS, TILE_DIRECT = 128;
constexpr uint32_t SUB_WAVE_FIRST = 128; // sub-chunks a wave of the tail march starts with
constexpr uint32_t SUB_REFILL = 32;      // and what it asks the pool for when it runs low
#ifndef WS_TAIL_WAVES
#define WS_TAIL_WAVES 4 // waves per workgroup of the tail march (tsdf_update.hip)
#endif
constexpr uint32_t SUB_WG_BLOCK = WS_TAIL_WAVES * SUB_WAVE_FIRST; // ... the fixed block of ids of a work item (its waves' shares side by side)
constexpr uint32_t SUB_ID_LIMIT = (1u << 27) - 2u;  // an entry is id << 5 | fill - 1, + 1 in the hash
// per-tile bytes, two planes in one allocation (tile_flag_plane_bytes apart): [0] "the free pass / an off-ray mark touched the
// tile" (plain idempotent byte stores), [1] "the tile is on the scan's list" (it has records).  The resolve visits the listed
// tiles through the list and finds the others by scanning these planes.
__host__ __device__ inline size_t tile_flag_plane_bytes(int64_t n_tiles) { return ((size_t)n_tiles + 16 + 255) & ~(size_t)255; }
constexpr uint32_t SUB_LOST = 0xffffffffu; // the pool was exhausted (scan aborted): nothing is written, nothing published

struct TileEntry // 16 bytes: one touched tile of the scan in flight
{
  uint32_t tile;
  int32_t tx, ty, tz; // tile coordinates (so that 256 threads per tile do not divide the tile id again)
};

struct TsdfCounters // device-resident
{
  // ---- the first line: what the marches hit with atomics while they run
  uint32_t chunk_cursor;  // sub-chunks handed out from the bottom of the pool by the tail march (reset by the set-up pass of the next scan)
  uint32_t n_listed;      // tiles with records (what the marches put on the tile list; final when the resolve starts; survives until the next scatter)
  uint32_t pad2[2];
  unsigned long long ub_total; // bits 0..47: sum of the per-ray record upper bounds of the scan; bits 48..63: set-up blocks that have added theirs
  uint32_t big_inserted;  // keys ever put into the (tile, chunk) hash since it was last emptied (the host empties it when it fills up)
  uint32_t free_cursor;   // sub-chunks handed out from the TOP of the pool by the free pass (one record each: a free-space candidate on a keyed voxel)
  uint32_t n_appended;    // tiles WITHOUT records the non-fused resolve has put behind the listed ones (tile_list[n_listed ...]) for the separate integrate pass;
                          // a counter of its own: every workgroup of the resolve reads n_listed on entry, so nothing may move that word while the resolve runs
  uint32_t pad0;
  // statistics of the last update (ws_tsdf_stats)
  uint32_t last_records;
  uint32_t last_contested;
  uint32_t last_listed;
  uint32_t last_runs;
  uint32_t last_free_keyed;
  uint32_t last_chunks;
  unsigned long long last_need; // record bound of the last scan
  uint32_t last_unlisted; // tiles without records (marks of the byte planes only) the resolve found by its scan
  uint32_t pad1;
  // ---- a line of their own (round 6): the two words that EVERY workgroup of the free pass and of the resolve reads when it starts.
  // Next to the cursors above they shared a cache line with the launch's own atomics -- a look at `abort` at the start of the tail
  // march's workgroups took that kernel from 135 to 210 us.
  alignas(128) uint32_t abort; // != 0: the scan in flight leaves no trace and is repeated -- bit 0: it ran out of sub-chunks, bit 1: a ray beyond the key range
  uint32_t error;              // bits of this scatter (also OR-ed into the map's sticky host-visible error word)
};
static_assert(offsetof(TsdfCounters, abort) % 128 == 0 && offsetof(TsdfCounters, abort) >= 128, "TsdfCounters: abort / error in a line of their own");

// device-resident Gauss-Newton state (tsdf_registration.cpp:28-96)
struct GnCore
{
  float T[16];      // total_transform, column-major
  int32_t center[3];
  float alpha;
  float prev[4];
  float it_weight_gradient;
  float epsilon;
  int32_t max_iterations;
  int32_t iterations;
  int32_t finished;
  int32_t error; // 1: the grid barrier of reg_loop_kernel timed out
};
struct GnState
{
  GnCore core;
  int64_t sums[44]; // last h(36) g(6) e c
};

} // namespace ws

struct ws_context
{
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // profiling
  uint32_t prof_mask = 0;
  struct Span
  {
    hipEvent_t a, b;
    int cls;
  };
  std::vector<Span> spans;       // recorded, not yet resolved
  std::vector<hipEvent_t> pool;  // free events
  std::mutex lists_mu;           // guards the two lists below (objects are created and destroyed while another thread's call walks them)
  std::vector<struct ws_map *> maps; // maps of this context (sticky device-side errors are reported at ws_sync)
  std::vector<struct ws_reg *> regs; // registrations of this context (a resident server of ws_reg_iterate is asked to leave by whoever enqueues other work)
  double prof_ms[WS_K_COUNT] = {};
  int64_t prof_n[WS_K_COUNT] = {};
};

struct ws_map
{
  ws_context *ctx = nullptr;
  ws::MapParams par[2]; // [WS_MAP_AVG], [WS_MAP_NEW]
  int64_t n_vox = 0;
  uint32_t *data[2] = {nullptr, nullptr};
  uint8_t *vstate = nullptr; // two planes of one byte per voxel, tile-major in bricks (vstate_plane_bytes, vbrick): keyed / touched by free space / free-space hit on a keyed voxel; off-ray free-space hit
  void *rays = nullptr;      // per-ray set-up records (sizeof(RaySetup) x 1 000 000)
  uint32_t *az_hist = nullptr, *az_off = nullptr, *ray_order = nullptr; // rays grouped by direction bin
  void *ray_bin = nullptr;   // [1 000 000] uint2: (direction bin, rank inside the bin) per ray
  int32_t *fan_steps = nullptr;       // [256] first ray step whose fan has j + 1 targets (depends on res only), see tail_bound
  int32_t fan_steps_host[ws::WS_FAN_TABLE] = {}; // staging of the same (lives as long as the map: async upload); [256..259]: division constants of ntz, nty
  bool prepped = false; // the scatter's scratch (histograms, tile counters, free-space hash) is zero / empty
  int32_t tau = 0, max_weight = 0, res = 0;
  bool new_is_default = false; // new_map known to be (tau,0) everywhere
  int integrate_mode = WS_INTEGRATE_SPARSE;
  int32_t *scan_dev = nullptr; // 1 000 000-point upload buffer
  // tile grid (4 x 4 x 64 voxels of storage space)
  int32_t ntx = 0, nty = 0, ntz = 0;
  int64_t n_tiles = 0;
  uint32_t *tile_nsub = nullptr;    // [n_tiles] sub-chunks (= entries) of the tile in the scan in flight (zero between scans)
  uint32_t *tile_ent = nullptr;     // [n_tiles][TILE_DIRECT] entries: sub-chunk id << 5 | records - 1 (never cleared: tile_nsub says how many are valid)
  uint8_t *tile_dirty = nullptr;    // two planes of [n_tiles] bytes (tile_flag_plane_bytes): touched by the free-space pass / an off-ray mark; on the list
  ws::TileEntry *tile_list = nullptr; // [n_tiles] touched tiles of the scan in flight
  // candidate records of the ray tails: the pool of sub-chunks (32 x 8 bytes)
  unsigned long long *rec = nullptr;
  uint32_t sub_cap = 0;
  unsigned long long *big_keys = nullptr; // (tile, entry number) -> entry + 1 for entries beyond TILE_DIRECT: keys, then uint32 values
  uint32_t big_slots = 0;
  uint32_t *block_stats = nullptr; // per-workgroup statistics (no shared counters in the hot kernels)
  uint32_t tail_blocks = 0;        // workgroups of the last tail march
  uint32_t resolve_blocks = 0;     // workgroups of the last tile resolve
  bool fused_done = false;         // the last scatter already integrated into avg_map
  uint32_t scan_seq = 0;           // scatters launched on this map (ray_setup reports its record bound under this number)
  uint64_t chunk_budget_bytes = 0; // (test entry, unused since round 4's sub-chunks: the pool is always sized by estimate)
  uint32_t est_shift = 0;          // the pool's share for the records is the record bound / 32 >> (est_shift - 1) (0: the whole bound; tests shrink it to force the abort route)
  ws::TsdfCounters *counters = nullptr;
  ws::TsdfCounters *counters_host = nullptr; // pinned
  uint32_t *status_host = nullptr;           // pinned + mapped: [0] sticky error bits, [4..5] record bound of the scan in flight (u64), [6] its sequence number,
                                             // [8] sequence number of the last scan whose marches have finished, [9] != 0: that scan was aborted (pool exhausted),
                                             // [10] keys in the (tile, entry) hash
  uint32_t *status_dev = nullptr;            // device view of status_host
  uint32_t *box_stage = nullptr; // device staging for ws_map_extract_box / ws_map_insert_box
  size_t box_stage_cap = 0;
  uint32_t last_error_bits = 0;  // device error bits already taken from status_host, not yet shown by ws_tsdf_stats
  // The scan whose verdict (did its records fit the pool?) has not been looked at yet: ws_tsdf_update* return after the
  // launches, like the reference's update_tsdf (update_tsdf.cu:165); the next call that takes this map settles it first
  // (settle_tsdf: the verdict is in host-mapped memory ~0.35 ms after the launches) and repeats the scan if it was aborted.
  // The reference's caller runs "one writer or many readers" (tsdf_mapping.cpp:62-75,114-124, tsdf_registration.cpp:54), and
  // every reader entry point settles: `active` is the lock-free fast path, `settle_mu` makes the slow path (wait for the
  // verdict; for an aborted scan: new pool, the scan again) exclusive -- a second reader waits there until the map is whole.
  struct PendingScan
  {
    std::atomic<bool> active{false};
    uint32_t seq = 0;
    size_t n = 0;  // the points are in ws_map::scan_dev (the set-up pass of the scan keeps a copy there: a repeat does not depend on the caller's buffer)
    int32_t pos[3] = {0, 0, 0}, up[3] = {0, 0, 0};
    bool fused = false;
    bool s0 = false;              // the scan went into a non-default new_map (decided when it was first launched)
    bool integrate_after = false; // ws_tsdf_update*: an integrate pass follows the scatter
    int attempts = 0;
  } pending;
  std::mutex settle_mu;
  hipStream_t shift_stream = nullptr; // second stream for asynchronous slab transfers (map shift off the scan path)
  hipEvent_t shift_event = nullptr;
  uint32_t *shift_stage_dev = nullptr;  // packed leaving slabs
  uint32_t *shift_stage_host = nullptr; // pinned
  size_t shift_stage_cap = 0;           // voxels
  ws_shift *shift_open = nullptr;       // the ticket in flight
};

struct ws_shift
{
  ws_map *map = nullptr;
  int n = 0;
  int32_t leave_lo[3][3], leave_hi[3][3]; // world boxes that left (coordinates of the window before that axis moved)
  int32_t enter_lo[3][3], enter_hi[3][3]; // world boxes that entered
  size_t offset[3];                       // of slab i in the staging buffers (voxels)
  size_t total = 0;
};

struct ws_reg
{
  ws_context *ctx = nullptr;
  int32_t *points = nullptr;
  size_t cap = 0, n = 0;
  int64_t *partials = nullptr; // [2][32][REG_BLOCKS]
  ws::GnState *state = nullptr;      // [2] device, double buffered by launch parity
  ws::GnState *state_host = nullptr; // pinned staging
  ws::GnState *result_host = nullptr;     // pinned + mapped: the resident loop writes its final state here
  ws::GnState *result_host_dev = nullptr; // device view of result_host
  int32_t *host_flag = nullptr;      // pinned + mapped: the device sets it when the loop has finished
  int32_t *host_flag_dev = nullptr;  // device view of host_flag
  int latest = 0;                    // state buffer holding the newest state
  int64_t *sums_dev = nullptr;       // 44
  uint32_t *grid_bar = nullptr;      // two sets of {abort flag, counted group accumulators} of reg_loop_kernel (alternate launches)
  uint32_t *shard_arrived = nullptr;  // arrival counter of reg_shard_kernel (zero between launches); [16]: reg_host_iter_kernel's
  int64_t *iter_host = nullptr;       // pinned + mapped: the 44 sums of ws_reg_iterate, then the call's sequence number
  int64_t *iter_host_dev = nullptr;   // device view of iter_host
  uint32_t iter_seq = 0;
  // the resident server behind ws_reg_iterate (reg_server_kernel, registration.hip): requests travel through host-mapped memory
  void *srv_mail = nullptr;           // ServerMail, pinned + mapped
  void *srv_mail_dev = nullptr;       // device view
  uint32_t *srv_ctl = nullptr;        // device words of the server (bell, pose, arrival counters), zero at creation
  std::atomic<uint32_t> srv_launch{0};   // id of the last server launched (0: none yet); it lives until ServerMail::exited says so
  std::atomic<bool> srv_stopping{false}; // somebody has asked that server to leave: the next request waits for it and starts a new one
  uint32_t srv_ids = 0;               // launch ids handed out
  uint32_t srv_seq = 0;               // request numbers handed out (1 .. 2^31 - 1)
  uint32_t srv_served = 0;            // the last request that was answered
  int srv_enabled = 1;                // 0: one launch per ws_reg_iterate (reg_host_iter_kernel), WS_REG_SERVER=0
  uint32_t srv_idle_us = 50;          // the server leaves after this long without a request
  uint32_t srv_launches = 0;          // statistics (ws_debug_reg_server)
  struct
  {
    const ws_map *map = nullptr;
    const void *points = nullptr, *map_data = nullptr;
    size_t n = 0;
    int32_t res = 0;
    uint32_t flags = 0;
    ws::MapParams par;
  } srv_sig;                          // what the living server was launched for
  uint32_t loop_launches = 0;
  bool loop_sets_clear = false;
  int loop_mode = 0;                 // WS_REG_LOOP_*
  int loop_supported = 0;            // the device holds the whole grid of reg_loop_kernel at once
  int debug_stall_next = 0;          // ws_debug_reg_stall
  int resident_fallbacks = 0;        // registrations redone with one launch per iteration after a barrier timeout
  // multi-GPU resident loop (ws_reg_peer_*): the ranks' totals meet in mailboxes in each other's HBM
  void *mailbox = nullptr;           // own mailbox: fine-grained device memory, [2][64] uint64
  void *peer_mailbox[8] = {};        // every rank's mailbox as this process addresses it ([peer_rank] == mailbox)
  bool peer_opened[8] = {};          // opened with hipIpcOpenMemHandle (to be closed)
  void *peer_block_dev = nullptr;    // PeerBlock
  int peer_rank = 0, peer_world = 0; // 0: not connected
  int peer_blocks = 0;               // grid of the peer loop on this rank
  bool peer_dirty = false;           // an exchange failed or was given up: the mailboxes hold partial additions until ws_reg_peer_reset / reconnect
};

// scan pre-processing buffers (App::preprocess on the device, scan_preprocess.hip)
struct ws_scan
{
  ws_context *ctx = nullptr;
  size_t cap = 0;         // points
  size_t table_slots = 0; // power of two >= 2 * cap
  float *in_stage = nullptr;
  size_t in_stage_floats = 0;
  int32_t *tmp = nullptr;
  uint32_t *slot_of = nullptr;
  uint64_t *keys = nullptr;
  uint32_t *first = nullptr;
  uint32_t *wg_count = nullptr;
  uint32_t *wg_off = nullptr;
  uint32_t *counters = nullptr;
  int32_t *out = nullptr;
  uint32_t *host_count = nullptr;     // pinned + mapped
  uint32_t *host_count_dev = nullptr;
  size_t n_out = 0;
};

namespace ws
{
void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define WS_HIP(call)                                                     \
  do                                                                     \
  {                                                                      \
    hipError_t e__ = (call);                                             \
    if (e__ != hipSuccess) return ws::hip_fail(e__, #call, __FILE__, __LINE__); \
  } while (0)

// profiling spans around a kernel class
void prof_begin(ws_context *ctx, int cls);
void prof_end(ws_context *ctx, int cls);

// launchers implemented in the .hip files
void fill_fan_steps(int32_t *fan_steps, int32_t res, int32_t ntz, int32_t nty);
int launch_tsdf_scatter(ws_map *m, const int32_t *xyz_dev, size_t n, const int32_t scanner_pos[3], const int32_t up[3], bool fused);
int settle_tsdf(ws_map *m); // the verdict of the scan in flight (repeats an aborted scan); every entry point that takes a map calls it first
size_t ray_setup_bytes();
int launch_scatter_prep(ws_map *m);
int resize_records(ws_map *m, uint64_t sub_chunks); // api.hip: (re)allocate the pool (waits for the stream)
uint64_t subs_for_scan(const ws_map *m, uint64_t need_records, uint64_t n_points); // sub-chunks the pool should hold for a scan of that record bound
int launch_tsdf_integrate(ws_map *m);
int launch_tsdf_stats(ws_map *m); // fills the last_* statistics of TsdfCounters from the per-workgroup slots
int launch_box_copy(ws_map *m, const ws::MapParams &par, int which, const int32_t lo[3], const int32_t ext[3], uint32_t *box_dev, bool pack, hipStream_t stream);
int fill_u32(ws_context *ctx, uint32_t *dst, uint32_t value, int64_t n);
int launch_box_fill(ws_map *m, const ws::MapParams &par, int which, const int32_t lo[3], const int32_t ext[3], uint32_t value, hipStream_t stream);
int check_all_equal_host(const uint32_t *data, int64_t n, uint32_t value);

int launch_reg_accumulate(ws_reg *r, const ws_map *m, const float *T_dev_or_null, int32_t res, uint32_t flags,
                          size_t first, size_t count, int64_t *sums_dev);
int launch_reg_iteration(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, int32_t k);
int launch_reg_shard(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, size_t first, size_t count, int64_t *sums_dev, int apply);
int launch_reg_solve(ws_reg *r, const int64_t *sums_dev);
int launch_reg_host_iter(ws_reg *r, const ws_map *m, const float T[16], int32_t res, uint32_t flags, uint32_t seq);
int launch_reg_server(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, uint32_t launch_id, uint32_t served, uint32_t idle_us);
size_t reg_server_mail_bytes();
size_t reg_server_ctl_bytes();
void reg_server_mail_write(void *mail, const float T[16], uint32_t seq);
void reg_server_mail_stop(void *mail, uint32_t launch_id);
int reg_server_mail_answer(const void *mail, uint32_t seq, int64_t sums[44]);
int reg_server_mail_selftest();
uint32_t reg_server_mail_exited(const void *mail);
int launch_scan_preprocess(ws_scan *sc, const float *xyz_dev, size_t n, size_t stride, const int32_t M[16], int32_t res);
size_t pre_table_slots(size_t max_points);
int launch_reg_loop(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, const ws::GnCore &init, bool peers = false, size_t first = 0, size_t count = 0);
size_t reg_peer_block_bytes();
void reg_peer_block_fill(void *host_image, void *const mailbox[8], int rank, int world);
size_t reg_mailbox_bytes();
int reg_groups();
int reg_default_blocks();
int reg_loop_supported(int device);
int launch_solve6_test(ws_context *ctx, const double *A_dev, const double *b_dev, size_t n, double *x_dev, int32_t *status_dev);
size_t reg_barrier_bytes();
} // namespace ws

/*
 * warpsense_hip.h — C ABI of libwarpsense_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for warpsense's data-parallel hot path: every entry point below replaces one
 * member of the reference's CUDA device API (file:line under the reference tree given per function).
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.  The header-only C++
 * classes with the reference's names (cuda::TSDFCuda, cuda::RegistrationCuda, cuda::DeviceMap,
 * cuda::DeviceMapMemWrapper, cuda::pause/cleanup) live in include/warpsense_hip/compat.hpp and forward
 * to these functions; INTEGRATION.md shows the reference-side wiring.
 *
 * Conventions
 *   - all geometry is integer millimetres / voxel indices exactly as in the reference
 *     (include/warpsense/consts.h: MATRIX_RESOLUTION 32768, WEIGHT_RESOLUTION 64);
 *   - a voxel is a packed TSDFEntry: low 16 bits value, high 16 bits weight (include/map/tsdf.h:16-23);
 *   - matrices are column-major like rmagine::Matrix4x4f / Matrix6x6l and Eigen
 *     (include/warpsense/math/matrix4x4.h:175-185, matrix6x6.h:112-115);
 *   - every function returns WS_OK (0) or a negative ws_status; ws_last_error() gives the message of
 *     the calling thread's last failure.  Nothing throws across the ABI.
 *   - work is stream-ordered on the context's HIP stream; functions that return host data synchronise.
 *   - threads and devices: like the reference's CUDA classes the library launches on the CALLING thread's current device.
 *     ws_ctx_create(device_id) makes that device current for the creating thread; a further thread that uses the context's
 *     handles on a multi-GPU node calls hipSetDevice(device_id) once first (HIP's current device is per thread and starts
 *     at 0).  ws_shift_wait / _slab / _end only wait on streams and may be called from any thread as they are.
 *
 * Result semantics: the TSDF scatter is resolved in the canonical serial order of the reference kernel
 * (ascending point index, ray step, fan step — SURVEY.md §7 H1), so results are deterministic and
 * bit-identical to oracle/ws_oracle.c.
 */
#ifndef WARPSENSE_HIP_H
#define WARPSENSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ws_context ws_context; /* one per (process, GPU): device id + stream                          */
typedef struct ws_map ws_map;         /* cuda::TSDFCuda: avg_map_ + new_map_ + scan buffer (update_tsdf.h:9-34) */
typedef struct ws_scan ws_scan;       /* scan pre-processing buffers: App::preprocess (src/warpsense/app.cpp:119-148)   */
typedef struct ws_reg ws_reg;         /* cuda::RegistrationCuda (registration.h:10-45)                          */

typedef enum
{
  WS_OK = 0,
  WS_ERR_INVALID = -1,     /* bad argument                                             */
  WS_ERR_HIP = -2,         /* HIP runtime error (message in ws_last_error)              */
  WS_ERR_TOO_MANY_POINTS = -3, /* scan larger than the 1 000 000-point buffer (update_tsdf.h:33) */
  WS_ERR_CAPACITY = -4,    /* a scan that needs more than 2^27 record sub-chunks: returned by the NEXT call that takes the map (the one
                              that looks at the scan's verdict; the record pool itself cannot overflow: a scan that exhausts it is
                              repeated with a larger one) */
  WS_ERR_RANGE = -5,       /* a ray of more than 65 536 ray steps or 255 fan steps: beyond what the record's key holds even with the
                              widest split; the ray was dropped (sticky).  (A scan whose OWN split is narrower -- 32 768 / 63 for the
                              reference's 131 072-point scans, 8192 / 31 at a million points -- is repeated in pieces of 16 384
                              points with the widest split instead: exact) */
  WS_ERR_TIMEOUT = -6,     /* ws_register_cloud_peers: a peer rank did not deliver (ws_register_cloud itself retries with one
                              launch per iteration instead of returning this) */
  WS_ERR_INTERNAL = -7     /* a device-side consistency check failed (sticky)            */
} ws_status;

/* Sticky device-side errors.  ws_tsdf_update* return after ENQUEUEING the kernels (like the reference,
 * update_tsdf.cu:165), so a problem found by the later kernels (WS_ERR_RANGE / INTERNAL: the
 * map is then not bit-exact) cannot come back from that call.  It is kept in host-visible memory and returned ONCE by the
 * first call on the same map that synchronises afterwards: ws_sync, ws_map_download, ws_register_cloud, ws_tsdf_stats.
 * compat.hpp turns it into the reference's print-and-exit (common.cuh:10-21).  Capacity is not among them: a scan that does
 * not fit the record pool is aborted without touching the maps and repeated with a larger pool by the next call that takes the
 * map, before that call's own work (every such entry point looks at the verdict first; two readers that get there at once -- the
 * reference runs register_cloud and the shift thread's to_host under a SHARED lock -- are serialised inside the library). */

#define WS_MAP_AVG 0 /* TSDFCuda::avg_map() */
#define WS_MAP_NEW 1 /* TSDFCuda::new_map() */

/* integrate pass selection, ws_tsdf_set_integrate() */
#define WS_INTEGRATE_SPARSE 0 /* default: touched 4x4x64-voxel tiles only, folded into the scatter's tile resolve      */
#define WS_INTEGRATE_DENSE 1  /* stream every voxel like cu_avg_tsdf_krnl (update_tsdf.cu:13-43)                      */
#define WS_INTEGRATE_SPARSE_SEPARATE 2 /* touched tiles only, as a separate pass over new_map (the resolve writes new_map) */

/* registration flags */
#define WS_REG_ALL_POINTS 0u
#define WS_REG_COMPAT_REFERENCE_LAUNCH 1u /* reproduce the <<<128,512>>> / N%32 coverage of registration.cu:353-356 */

/* how ws_register_cloud runs the Gauss-Newton loop, ws_reg_set_loop() */
#define WS_REG_LOOP_RESIDENT 0 /* one launch for the whole loop, grid barrier between iterations (default;
                                  falls back to LAUNCHES when the device cannot hold the grid at once) */
#define WS_REG_LOOP_LAUNCHES 1 /* one launch per iteration */

const char *ws_last_error(void);
int ws_version(void);

/* ------------------------------------------------------------------ context ---- */
/* cuda runtime implicit context of the reference; device_id < 0 -> current device */
int ws_ctx_create(int device_id, ws_context **out);
int ws_ctx_destroy(ws_context *ctx);
/* Run all work of this context on a caller-owned hipStream_t (e.g. torch's current stream). NULL restores the own stream. */
int ws_ctx_set_stream(ws_context *ctx, void *hip_stream);
/* cuda::pause()  — src/warpsense/cuda/cleanup.cu:3-6 */
int ws_sync(ws_context *ctx);
/* cuda::cleanup() — src/warpsense/cuda/cleanup.cu:8-11 (hipDeviceReset; invalidates every handle) */
int ws_device_reset(void);

/* ------------------------------------------------------------------ maps ---- */
/* TSDFCuda::TSDFCuda(existing_map, tau, max_weight, map_resolution) — update_tsdf.cu:130-141.
 * Like the reference, BOTH device maps start as copies of host_data (DeviceMapMemWrapper ctor,
 * device_map_wrapper.cu:20-24). host_data may be NULL: both maps are then filled with (tau, 0). */
int ws_map_create(ws_context *ctx, const int32_t size[3], const int32_t pos[3], const int32_t offset[3],
                  const uint32_t *host_data, int32_t tau, int32_t max_weight, int32_t map_resolution, ws_map **out);
int ws_map_destroy(ws_map *map); /* TSDFCuda::~TSDFCuda + ~DeviceMapMemWrapper */
/* DeviceMapMemWrapper::to_device — device_map_wrapper.cu:35-45 (params + all voxels, host -> HBM) */
int ws_map_upload(ws_map *map, int which, const int32_t size[3], const int32_t pos[3], const int32_t offset[3],
                  const uint32_t *host_data);
/* DeviceMapMemWrapper::update_params — device_map_wrapper.cu:47-56 (params only) */
int ws_map_set_params(ws_map *map, int which, const int32_t size[3], const int32_t pos[3], const int32_t offset[3]);
/* DeviceMapMemWrapper::to_host — device_map_wrapper.cu:85-92 (synchronises) */
int ws_map_download(ws_map *map, int which, int32_t size[3], int32_t pos[3], int32_t offset[3], uint32_t *host_data);
/* Device side of TSDFMapping::map_shift (src/warpsense/tsdf_mapping.cpp:97-136 + HDF5LocalMap::shift,
 * src/map/hdf5_local_map.cpp:53-118): instead of moving the WHOLE map through the host, only the slabs that leave
 * or enter the window are packed / unpacked.  Boxes are inclusive world-voxel ranges inside the current window;
 * the host buffer is dense, x major, z fastest.  The caller updates pos/offset with ws_map_set_params in between
 * (exactly the three steps of HDF5LocalMap::shift: save, move window, load).  Both synchronise. */
int ws_map_extract_box(ws_map *map, int which, const int32_t lo[3], const int32_t hi[3], uint32_t *host_out);
int ws_map_insert_box(ws_map *map, int which, const int32_t lo[3], const int32_t hi[3], const uint32_t *host_in);
/* The same shift OFF the scan path (in the reference TSDFMapping::map_shift runs on its own thread and only blocks the
 * scans while it swaps the maps, tsdf_mapping.cpp:97-136).  ws_shift_begin is stream-ordered on the map's stream and
 * returns without waiting: per axis (x, y, z, like HDF5LocalMap::shift) the slab of avg_map that leaves the window is
 * packed into a device staging buffer, pos/offset of BOTH device maps move, and the slab that enters is filled with
 * `fill_entry` (the global map's default entry).  The staged slabs then travel to pinned host memory on a SECOND stream
 * while the next scans already run against the new window.  The caller
 *   - overwrites, with ws_map_insert_box, those parts of the entering slabs the global map already holds (revisits),
 *   - and, typically on a worker thread: ws_shift_wait (blocks on the second stream only), ws_shift_slab for each
 *     leaving slab -> global map, ws_shift_end.
 * One shift can be in flight per map: ws_shift_begin fails with WS_ERR_INVALID while a ticket is open. */
typedef struct ws_shift ws_shift;
int ws_shift_begin(ws_map *map, const int32_t new_pos[3], uint32_t fill_entry, ws_shift **out);
/* staging (device + pinned host) for shifts of up to `voxels` leaving voxels, so that no shift has to allocate */
int ws_shift_reserve(ws_map *map, uint64_t voxels);
int ws_shift_count(const ws_shift *shift);                                              /* slabs: 0 .. 3 */
int ws_shift_entering(const ws_shift *shift, int i, int32_t lo[3], int32_t hi[3]);      /* box filled with fill_entry (world voxels) */
int ws_shift_wait(ws_shift *shift);                                                     /* the host copies are complete */
int ws_shift_slab(const ws_shift *shift, int i, int32_t lo[3], int32_t hi[3], const uint32_t **host_data); /* after ws_shift_wait */
int ws_shift_end(ws_shift *shift);
/* the ring-buffer parameters the kernels currently use for `which` (no synchronisation; any of the outputs may be NULL) */
int ws_map_get_params(const ws_map *map, int which, int32_t size[3], int32_t pos[3], int32_t offset[3]);
/* device pointer of the voxel array (uint32 per voxel, z fastest) — what DeviceMap::data_ is on the device */
void *ws_map_device_data(ws_map *map, int which);
int64_t ws_map_n_voxels(const ws_map *map);

/* ------------------------------------------------------------------ TSDF update ---- */
/* TSDFCuda::update_tsdf(scan_points, scanner_pos, up) — update_tsdf.cu:143-166.
 * xyz_host: n x 3 int32 (rmagine::Pointi AoS); scanner_pos in voxel units, up scaled by 32768.
 * Returns after enqueueing, like the reference (no device sync). */
int ws_tsdf_update(ws_map *map, const int32_t *xyz_host, size_t n, const int32_t scanner_pos[3], const int32_t up[3]);
/* same with the scan already resident in HBM (no H2D copy).  Asynchronous like a stream copy: xyz_dev must stay unchanged until
 * the kernels enqueued by this call have read it (stream order: anything enqueued later on the context's stream is safe, e.g.
 * ws_scan_preprocess of the next frame).  The next call that takes this map -- ws_register_cloud, ws_reg_iterate, ws_sync, a
 * download, the next update -- looks at the scan's verdict first and repeats the scan with a larger record pool in the rare case
 * that it did not fit (never an inexact map); the repeat reads a copy of the scan that the first attempt left in a buffer of the
 * map, not xyz_dev. */
int ws_tsdf_update_dev(ws_map *map, const int32_t *xyz_dev, size_t n, const int32_t scanner_pos[3], const int32_t up[3]);
/* only the scatter (cu_min_tsdf_krnl, update_tsdf.cu:45-128): fills new_map, no integrate. For parity tests. */
int ws_tsdf_scatter_dev(ws_map *map, const int32_t *xyz_dev, size_t n, const int32_t scanner_pos[3], const int32_t up[3]);
/* only the integrate pass (cu_avg_tsdf_krnl, update_tsdf.cu:13-43) */
int ws_tsdf_integrate(ws_map *map);
int ws_tsdf_set_integrate(ws_map *map, int mode);
/* Candidate-record capacity of the scatter: records of 8 bytes in sub-chunks of 32 that belong to one 4x4x64-voxel tile each,
 * taken from a pool.  The pool is sized from the scan's own record bound (its set-up pass counts every ray step as a candidate,
 * ~2.7 x what a scan makes) plus a fixed share per work item -- an estimate: how many partly filled sub-chunks a scan leaves
 * has no useful bound.  A scan that does exhaust the pool is ABORTED -- the maps stay untouched -- and repeated with a larger
 * pool inside the same ws_tsdf_update* call, which therefore returns once the marches are over (~0.35 ms into the update, the
 * resolve still running).  Reserving up front only avoids such a re-run. */
int ws_tsdf_set_capacity(ws_map *map, uint64_t records);
/* Test entry: the pool's share for the records is (record bound / 32) >> (est_shift - 1) (0: the default, the whole bound) -- a
 * large shift forces the abort-and-repeat route on a small map.  budget_bytes is unused (rounds 1-4 had a second policy). */
int ws_debug_tsdf_chunk_policy(ws_map *map, uint64_t budget_bytes, uint32_t est_shift);

typedef struct
{
  int64_t contested_voxels; /* voxels of the last update decided by the exact ordered rounds (a negative-weight
                               candidate could have blocked the earliest positive one)                         */
  int64_t records;          /* scatter targets that became records (ray tails + free-space candidates on keyed voxels) */
  int64_t tiles;            /* touched 4x4x64-voxel tiles (resolved and integrated)                             */
  int32_t error_flags;      /* device error bits since the last call: 2 record-field range, 4 free-space bound, 8 internal */
  int32_t hash_entries;     /* entries beyond a tile's 128th that went through the (tile, number) hash since it was last emptied */
  int64_t runs;             /* (wave, tile) groups of records: one reservation in the tile's entry table each           */
  int64_t free_space_hits;  /* free-space candidates that met ordered candidates (and joined that tile's records)   */
  int64_t record_slots;     /* the scan's record bound (from its set-up pass) ...                                 */
  int64_t record_capacity;  /* ... and the record places of the pool (sub-chunks x 32)                             */
} ws_tsdf_stats_t;
int ws_tsdf_stats(ws_map *map, ws_tsdf_stats_t *out); /* synchronises */

/* ------------------------------------------------------------------ registration ---- */
/* RegistrationCuda::RegistrationCuda — registration.cu:259-281 (buffers grow on demand beyond max_points) */
int ws_reg_create(ws_context *ctx, size_t max_points, ws_reg **out);
int ws_reg_destroy(ws_reg *reg);
/* RegistrationCuda::prepare_registration — registration.cu:303-308 */
int ws_reg_prepare(ws_reg *reg, const int32_t *xyz_host, size_t n);
/* the same for a cloud in HBM: an asynchronous device-to-device copy on the context's stream (xyz_dev must stay unchanged until that
 * copy has run: anything enqueued later on the context's stream is safe; a caller with streams of its own -- torch's allocator --
 * keeps the buffer until the next ws_reg_prepare* or a ws_sync, as warpsense_amd/api.py does) */
int ws_reg_prepare_dev(ws_reg *reg, const int32_t *xyz_dev, size_t n);
/* the registration's own copy of the prepared cloud (device memory, n x 3 int32) and its point count; the pointer
 * changes when a larger cloud makes the buffer grow (callers that capture kernels into HIP graphs key on it) */
const int32_t *ws_reg_points_dev(const ws_reg *reg, size_t *n);
/* RegistrationCuda::perform_registration — registration.cu:347-368. T, h column-major. Synchronises. */
int ws_reg_iterate(ws_reg *reg, const ws_map *map, const float T[16], int32_t map_resolution, uint32_t flags,
                   int64_t h[36], int64_t g[6], int32_t *e, int32_t *c);
/* cuda::TSDFRegistration::register_cloud — src/warpsense/tsdf_registration.cpp:28-96 with the whole
 * Gauss-Newton loop on the device (points must have been given to ws_reg_prepare*). Synchronises. */
int ws_register_cloud(ws_reg *reg, const ws_map *map, const float T_in[16], int32_t max_iterations,
                      float it_weight_gradient, float epsilon, int32_t map_resolution, uint32_t flags,
                      float T_out[16], int32_t *iterations);
int ws_reg_set_loop(ws_reg *reg, int mode /* WS_REG_LOOP_* */);

/* Building blocks of the same loop for point-sharded multi-GPU runs: every rank owns the
 * points [first, first+count) of the prepared cloud, accumulates its 44 int64 partial sums
 * (h[36] column-major, g[6], e, c) into sums_dev, the caller all-reduces sums_dev (RCCL), then every
 * rank runs the identical solve. All stream-ordered, no host synchronisation. */
int ws_reg_begin(ws_reg *reg, const float T_in[16], int32_t max_iterations, float it_weight_gradient, float epsilon);
int ws_reg_accumulate_dev(ws_reg *reg, const ws_map *map, int32_t map_resolution, uint32_t flags, size_t first,
                          size_t count, int64_t *sums_dev /* 44 */);
int ws_reg_solve_dev(ws_reg *reg, const int64_t *sums_dev /* 44 */);
/* The same two steps as ONE launch per iteration: if apply_previous != 0, first the Gauss-Newton update from the 44
 * (all-reduced) sums in sums_dev -- exactly what ws_reg_solve_dev does --, then the accumulation of [first, first+count)
 * into sums_dev.  A sharded loop is: ws_reg_begin; { ws_reg_iterate_shard_dev(apply_previous = not the first); all-reduce
 * sums_dev } x n; ws_reg_solve_dev(sums_dev); ws_reg_poll.  Do not touch sums_dev between the all-reduce and the next call. */
int ws_reg_iterate_shard_dev(ws_reg *reg, const ws_map *map, int32_t map_resolution, uint32_t flags, size_t first, size_t count,
                             int64_t *sums_dev /* 44 */, int32_t apply_previous);
int ws_reg_poll(ws_reg *reg, int32_t *finished, int32_t *iterations, float T_out[16]); /* synchronises */

/* The same sharded loop WITHOUT the host in it (north_star: point-sharded registration across GPUs): every rank runs the
 * resident loop of ws_register_cloud on its points [first, first + count) and the ranks' 44 sums meet in MAILBOXES in each
 * other's HBM -- fine-grained device memory, peer-mapped through hipIpc, system-scope atomic adds over xGMI whose top byte
 * counts the ranks (exact for any rank order) -- so an iteration costs one device-side exchange instead of a launch, an RCCL
 * call and two host calls.  Set-up, once per process group:
 *   ws_reg_peer_mailbox(reg, handle)            -> this rank's mailbox as a 64-byte IPC handle (hipIpcMemHandle_t)
 *   [all-gather the handles, any transport]
 *   ws_reg_peer_connect(reg, rank, world, handles (world x 64 bytes), blocks)
 * then all ranks call ws_register_cloud_peers together for every cloud (every rank has prepared the WHOLE cloud; the map is
 * replicated).  `blocks`: workgroups of the loop on this rank, 0 = 256 (one per CU); ranks that share one GPU (tests) pass
 * 256 / ranks-per-GPU so that all of them are resident at once.  WS_ERR_TIMEOUT: a peer did not deliver within 20 ms
 * (WS_REG_PEER_TIMEOUT_MS in the environment at connect time changes the limit) -- all ranks see it; call ws_reg_peer_reset
 * on every rank (between two barriers of the caller's) and fall back to the RCCL route.  Until then every further
 * ws_register_cloud_peers on this handle is refused (WS_ERR_INVALID): the mailboxes hold the partial additions of the
 * exchange that was given up.
 * ws_reg_peer_connect_local connects ws_reg handles of ONE process (several contexts / streams on one GPU) without IPC. */
#define WS_IPC_HANDLE_BYTES 64
int ws_reg_peer_mailbox(ws_reg *reg, void *ipc_handle_out /* WS_IPC_HANDLE_BYTES, may be NULL */);
int ws_reg_peer_connect(ws_reg *reg, int32_t rank, int32_t world, const void *ipc_handles, int32_t blocks);
int ws_reg_peer_connect_local(ws_reg *reg, int32_t rank, int32_t world, ws_reg *const *regs, int32_t blocks);
int ws_reg_peer_disconnect(ws_reg *reg);
int ws_reg_peer_reset(ws_reg *reg);
int ws_register_cloud_peers(ws_reg *reg, const ws_map *map, size_t first, size_t count, const float T_in[16], int32_t max_iterations,
                            float it_weight_gradient, float epsilon, int32_t map_resolution, uint32_t flags, float T_out[16],
                            int32_t *iterations);

/* Test entry: the 6x6 solve of the Gauss-Newton update alone (LU with partial pivoting in double, one wavefront per
 * system; stands for Eigen's hf.inverse() * g, tsdf_registration.cpp:69). n systems: A row-major n x 36, b n x 6 ->
 * x n x 6, status n (0, or -1 for a singular matrix). Host pointers; synchronises. */
int ws_debug_solve6(ws_context *ctx, const double *A, const double *b, size_t n, double *x, int32_t *status);

/* Diagnostics: the per-workgroup statistics slots of the last TSDF update (records per tail workgroup, then at +65536 its
 * flush groups, then at +131072 the contested voxels per resolve workgroup).  Synchronises. */
int ws_debug_block_stats(ws_map *map, uint32_t *out, size_t words);

/* Test entry: make the NEXT resident registration of `reg` lose one workgroup's contribution to the first exchange, as if
 * another kernel kept that workgroup off the chip: the exchange times out (5 ms) and ws_register_cloud repeats the
 * registration with one launch per iteration. *fallbacks (may be NULL) receives how often that has happened on `reg`. */
int ws_debug_reg_stall(ws_reg *reg, int32_t stall_next, int32_t *fallbacks);
/* test / tuning entry: the resident server behind ws_reg_iterate (enable: 1 / 0, -1 = leave as it is; idle_us > 0: how long it
 * stays without a request, default 50); *launches = servers started so far on this handle */
int ws_debug_reg_server(ws_reg *reg, int32_t enable, int32_t idle_us, int32_t *launches);
/* Test entry, no GPU needed: the host half of the server's mail protocol (request line with checksum, the answer's seven tagged lines:
 * stale, incomplete and torn answers must not be taken).  0: as expected, else a bit per failed case. */
int ws_debug_reg_mail_selftest(void);

/* Test entry: the 44 sums (h[36] column-major, g[6], e, c -- the out-parameters of perform_registration, registration.cu:347-368)
 * the LAST Gauss-Newton update of the last ws_register_cloud / ws_register_cloud_peers on `reg` was made from.  Synchronises. */
int ws_debug_reg_sums(ws_reg *reg, int64_t sums_out[44]);

/* ------------------------------------------------------------------ scan pre-processing ---- */
/* App::preprocess — src/warpsense/app.cpp:119-148, on the device: sensor points in float metres
 * (x y z first, `stride_floats` floats per point, e.g. 3, or 4 for PointXYZI) are dropped if x, y and z are all
 * < 0.3, scaled to mm, snapped to the centre of their `map_resolution` voxel, transformed by to_int_mat(pose)
 * (pose: 4x4 column-major, translation in mm) and de-duplicated.  Output: int32 mm points in the order of their
 * first occurrence in the input (the reference's unordered_set order is unspecified), resident on the device
 * until the next call: feed ws_scan_points_dev() to ws_tsdf_update_dev / ws_reg_prepare_dev.  Synchronises
 * (the count comes back to the host).  WS_ERR_RANGE: a transformed coordinate beyond +-2^20 mm. */
int ws_scan_create(ws_context *ctx, size_t max_points, ws_scan **out);
int ws_scan_destroy(ws_scan *scan);
int ws_scan_preprocess(ws_scan *scan, const float *xyz_host, size_t n, size_t stride_floats, const float pose[16],
                       int32_t map_resolution, size_t *n_out);
int ws_scan_preprocess_dev(ws_scan *scan, const float *xyz_dev, size_t n, size_t stride_floats, const float pose[16],
                           int32_t map_resolution, size_t *n_out);
const int32_t *ws_scan_points_dev(const ws_scan *scan); /* n_out x 3 int32, device memory */
int ws_scan_download(ws_scan *scan, int32_t *xyz_host, size_t capacity_points, size_t *n_out);

/* ------------------------------------------------------------------ measurement ---- */
/* Kernel classes for hipEvent timing (bench.py's roofline leg). */
#define WS_K_SETUP 0         /* per-ray set-up + direction sort                                    */
#define WS_K_MARCH_TAILS 1   /* ray tails -> records, handed to the chunks of their tiles            */
#define WS_K_MARCH_FREE 2    /* free-space steps -> one byte per voxel                             */
#define WS_K_TILE_BIN 3      /* rounds 1-3 only (no kernel of this class since round 4: always 0)   */
#define WS_K_TILE_RESOLVE 4  /* exact per-tile fold in LDS (+ fused integrate)                     */
#define WS_K_INTEGRATE 5     /* separate sparse or dense weighted-average pass (cu_avg_tsdf_krnl)  */
#define WS_K_REG 6           /* Gauss-Newton iterations (accumulate + solve)                       */
#define WS_K_UPDATE 7        /* one span over ALL kernels of a ws_tsdf_update* call (two events per scan)  */
#define WS_K_COUNT 8
int ws_prof_enable(ws_context *ctx, uint32_t class_mask); /* 0 disables */
/* sum of event-measured durations and number of launches per class since the last reset (synchronises) */
int ws_prof_read(ws_context *ctx, int kernel_class, double *total_ms, int64_t *launches);
int ws_prof_reset(ws_context *ctx);

#ifdef __cplusplus
}
#endif
#endif /* WARPSENSE_HIP_H */

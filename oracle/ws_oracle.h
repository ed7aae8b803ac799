/*
 * ws_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's CUDA hot path, executed serially in
 * the canonical order "ascending point index, then ray step, then fan step"
 * (one legal schedule of the racy reference kernel).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  Nothing under warpsense_amd/ links or imports it.
 *
 * Pin status: pinned by the reference's own known-answer tests
 * (test/map.cpp:9-90, test/cuda.cpp:28-105,760-923,968-990, test/test.cu:48-126),
 * by oracle/_ref (the reference's own headers device_map.h, math headers, tsdf.h
 * compiled with g++, see oracle/Makefile) and by the whole-scan counters the
 * survey recorded from the reference kernel source.
 * The 6x6 solve follows Eigen's PartialPivLU (third-party, version unpinned in
 * the reference: find_package(Eigen3 3.3), CMakeLists.txt:27) — "parity unpinned"
 * for that one step; the 1e-4 m / 1e-4 rad pose tolerance absorbs it.
 */
#ifndef WS_ORACLE_H
#define WS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSO_MATRIX_RESOLUTION 32768 /* include/warpsense/consts.h:12-13 */
#define WSO_WEIGHT_RESOLUTION 64    /* include/warpsense/consts.h:9-10  */

/* Non-owning view of a ring-buffer local map (include/warpsense/cuda/device_map.h:32-164).
 * data[i] is a packed TSDFEntry: low 16 bits value, high 16 bits weight (include/map/tsdf.h:16-23). */
typedef struct
{
  int32_t size[3];
  int32_t pos[3];
  int32_t offset[3];
  uint32_t *data;
} wso_map;

typedef struct
{
  int64_t write_calls;    /* V: write_tsdf_min invocations                   */
  int64_t accepted;       /* CAS that replaced the entry                      */
  int64_t rays_in_bounds; /* rays that passed the in_bounds_with_buffer_pos   */
  int64_t rays_degenerate;/* distance==0 or interpolation_norm==0 (guarded)   */
} wso_update_stats;

/* flags for the registration functions */
#define WSO_REG_ALL_POINTS 0u
#define WSO_REG_COMPAT_REFERENCE_LAUNCH 1u /* first 65536 points only, tail N%32 dropped */

/* ---- packed entry helpers (include/map/tsdf.h) ---- */
uint32_t wso_pack(int16_t value, int16_t weight);
int16_t wso_value(uint32_t raw);
int16_t wso_weight(uint32_t raw);

/* ---- ring-buffer index math (device_map.h:14-30,93-128) ---- */
int64_t wso_get_index(const wso_map *m, int32_t x, int32_t y, int32_t z);
int wso_in_bounds(const wso_map *m, int32_t x, int32_t y, int32_t z);
int wso_in_bounds_with_buffer_pos(const wso_map *m, int32_t x, int32_t y, int32_t z, int32_t buffer);
int wso_in_bounds_with_buffer_neg(const wso_map *m, int32_t x, int32_t y, int32_t z, int32_t buffer);

/* ---- fixed-point helpers (cuda/util.h:11-35,111-123; util/util.h:8-56) ---- */
void wso_to_int_mat(const float T[16] /*col-major*/, int32_t M[16] /*col-major*/);
void wso_transform_point(const int32_t p[3], const int32_t M[16], int32_t out[3]);
void wso_to_map(const int32_t p[3], int32_t res, int32_t out[3]);
int32_t wso_dz_per_distance(void);
/* tsdf_mapping.cpp:77-85 */
void wso_convert_pose(const float pose[16], int32_t res, int32_t pos_vox[3], int32_t up[3]);

/* ---- vector math (math/vector3.h) and the per-ray set-up of update_tsdf.cu:57-63 ---- */
int32_t wso_l2norm_i(int32_t x, int32_t y, int32_t z);
int64_t wso_l2norm_l(int64_t x, int64_t y, int64_t z);
void wso_cross_i(const int32_t a[3], const int32_t b[3], int32_t out[3]);
int wso_ray_setup(const int32_t point[3], const int32_t pos_mm[3], const int32_t up[3], int32_t *distance_out, int64_t iv[3]);

/* ---- atomic_tsdf_min executed serially (cuda/util.h:70-109); returns 1 if the entry was replaced ---- */
int wso_tsdf_min(uint32_t *addr, uint32_t new_raw);

/* ---- TSDF update (update_tsdf.cu:13-128,143-166) ---- */
void wso_update_min(wso_map *new_map, const int32_t *xyz, size_t n, const int32_t scanner_pos[3],
                    const int32_t up[3], int32_t tau, int32_t res, wso_update_stats *stats);
void wso_update_avg(uint32_t *new_data, uint32_t *avg_data, int64_t n_vox, int32_t max_weight, int32_t tau);
void wso_update_tsdf(wso_map *avg_map, wso_map *new_map, const int32_t *xyz, size_t n,
                     const int32_t scanner_pos[3], const int32_t up[3], int32_t tau, int32_t max_weight,
                     int32_t res, wso_update_stats *stats);

/* ---- registration (registration.cu:14-257,310-368) ---- */
void wso_calc_jacobis(const wso_map *map, const float T[16], const int32_t *xyz, size_t n, int32_t res,
                      int64_t *jacobis /*6n*/, int16_t *values /*n*/, uint8_t *mask /*n*/, uint32_t flags);
void wso_reduce(const int64_t *jacobis, const int16_t *values, const uint8_t *mask, size_t n,
                int64_t h[36] /*col-major*/, int64_t g[6], int32_t *e, int32_t *c, uint32_t flags);
void wso_reg_iterate(const wso_map *map, const float T[16], const int32_t *xyz, size_t n, int32_t res,
                     int64_t h[36], int64_t g[6], int32_t *e, int32_t *c, uint32_t flags);

/* ---- host Gauss-Newton loop (tsdf_registration.cpp:28-96, registration/util.h:5-39) ---- */
int wso_solve6(const double A[36] /*row-major*/, const double b[6], double x[6]);
void wso_xi_to_transform(const double xi[6], const int32_t center[3], float T[16] /*col-major*/);
typedef struct
{
  float T[16]; /* total_transform, column-major */
  int32_t center[3];
  float alpha;
  float prev[4];
  float it_weight_gradient;
  float epsilon;
  int32_t max_iterations;
  int32_t iterations;
  int32_t finished;
} wso_gn_state;
void wso_gn_begin(wso_gn_state *st, const float T_in[16], int32_t max_iterations, float it_weight_gradient, float epsilon);
void wso_gn_update(wso_gn_state *st, const int64_t sums[44] /* h[36] col-major, g[6], e, c */);
/* returns iterations executed; trace (optional) receives per iteration: h[36] g[6] e c -> 44 int64 */
int wso_register_cloud(const wso_map *map, const int32_t *xyz, size_t n, const float T_in[16],
                       int32_t max_iterations, float it_weight_gradient, float epsilon, int32_t res,
                       uint32_t flags, float T_out[16], int64_t *trace, int32_t trace_cap);

/* test aid: record the candidates of one voxel during the next wso_update_min (rows: point, len, step, value, weight, accepted) */
void wso_debug_watch(int64_t idx, int32_t *out, size_t cap);
size_t wso_debug_watch_count(void);

/* ---- scan pre-processing (App::preprocess, src/warpsense/app.cpp:119-148) ----
 * xyz: n points in float metres, `stride` floats apart; pose: 4x4 column-major, translation in mm.
 * out: at most n points (int32 mm), each distinct transformed voxel centre once, in the order of its first
 * occurrence in the input (the reference's unordered_set leaves the order unspecified). Returns the count. */
size_t wso_preprocess(const float *xyz, size_t n, size_t stride, const float pose[16], int32_t res, int32_t *out);

#ifdef __cplusplus
}
#endif
#endif

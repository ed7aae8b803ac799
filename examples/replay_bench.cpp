// replay_bench.cpp — BASELINE configs[2] measured with the C++ node: warpsense::App over a synthetic OS1-128
// stream on the sliding map, paced at the sensor's rate or back to back, wall clock per stage (the reference's RuntimeEvaluator
// forms, app.cpp:68-111) NEXT TO the device's own clock for the same scans (ws_prof_*: hipEvent spans around the update's and
// the registration's kernels), so that "microseconds per Gauss-Newton iteration in the stream" is read off the GPU timeline and
// not off a host harness.
//
//   replay_bench <clouds.bin> <scans> <points_per_scan> <map_edge_voxels> <res> <tau> <max_weight> <shift_m> <hz>
//
// clouds.bin: scans x points x 3 float32 (metres, sensor frame; tools/replay_cpp.py writes it).  hz = 0: back to back.
// WS_REPLAY_ASYNC_SHIFT=1: the map shift off the scan path.  Prints ONE JSON line.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>
#include <vector>

#include "warpsense_hip/app.hpp"

int main(int argc, char **argv)
{
  if (argc != 10)
  {
    fprintf(stderr, "usage: %s clouds.bin scans points edge res tau max_weight shift_m hz\n", argv[0]);
    return 2;
  }
  const size_t scans = strtoull(argv[2], nullptr, 10), n = strtoull(argv[3], nullptr, 10);
  warpsense::AppParams p;
  const int edge = atoi(argv[4]);
  p.map_size[0] = p.map_size[1] = p.map_size[2] = edge;
  p.hot.map_resolution = atoi(argv[5]);
  p.hot.tau = atoi(argv[6]);
  p.hot.max_weight = atoi(argv[7]);
  p.max_distance = (float)p.hot.tau / 1000.f;
  p.shift = (float)atof(argv[8]);
  const double hz = atof(argv[9]);
  p.async_shift = getenv("WS_REPLAY_ASYNC_SHIFT") != nullptr;
  std::vector<float> clouds(scans * n * 3);
  {
    std::ifstream f(argv[1], std::ios::binary);
    f.read(reinterpret_cast<char *>(clouds.data()), (std::streamsize)(clouds.size() * sizeof(float)));
    if (!f)
    {
      fprintf(stderr, "cannot read %zu floats from %s\n", clouds.size(), argv[1]);
      return 2;
    }
  }
  using clk = std::chrono::steady_clock;
  const clk::time_point t_setup = clk::now();
  warpsense::App app(p, "", n);
  cuda::pause();
  const double setup_s = std::chrono::duration<double>(clk::now() - t_setup).count();
  ws_context *ctx = cuda::detail::context();
  ws_prof_enable(ctx, (1u << WS_K_UPDATE) | (1u << WS_K_REG));
  struct Row
  {
    warpsense::StageTimes t;
    int iterations;
    double dev_update_ms, dev_reg_ms;
  };
  std::vector<Row> rows;
  const clk::time_point t1 = clk::now();
  double busy_s = 0;
  for (size_t k = 0; k < scans; ++k)
  {
    if (hz > 0.0) std::this_thread::sleep_until(t1 + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double>((double)k / hz)));
    const clk::time_point tb = clk::now();
    app.cloud_callback(&clouds[k * n * 3], n, 3);
    busy_s += std::chrono::duration<double>(clk::now() - tb).count();
    Row r;
    r.t = app.last_times();
    r.iterations = app.last_iterations();
    // (outside the callback's clock: reading the spans waits for their events)
    int64_t l0 = 0, l1 = 0;
    ws_prof_read(ctx, WS_K_UPDATE, &r.dev_update_ms, &l0);
    ws_prof_read(ctx, WS_K_REG, &r.dev_reg_ms, &l1);
    ws_prof_reset(ctx);
    rows.push_back(r);
  }
  cuda::pause();
  const double stream_s = std::chrono::duration<double>(clk::now() - t1).count();
  // the first two scans build the map from nothing (pool growth, first-touch): reported apart, like tools/replay_stream.py
  const size_t skip = std::min<size_t>(2, rows.size() ? rows.size() - 1 : 0);
  double sum_pre = 0, sum_tsdf = 0, sum_reg = 0, sum_shift = 0, sum_total = 0, worst = 0, dev_upd = 0, dev_reg = 0;
  long its = 0;
  int over = 0, n_upd = 0;
  for (size_t k = skip; k < rows.size(); ++k)
  {
    const Row &r = rows[k];
    sum_pre += r.t.preprocess_us;
    sum_tsdf += r.t.tsdf_us;
    sum_reg += r.t.registration_us;
    sum_shift += r.t.shift_us;
    sum_total += r.t.total_us;
    worst = std::max(worst, r.t.total_us);
    over += r.t.total_us > 100000.0;
    dev_upd += r.dev_update_ms;
    n_upd += r.dev_update_ms > 0.0;
    dev_reg += r.dev_reg_ms;
    its += r.iterations;
  }
  const double m = (double)(rows.size() - skip);
  printf("{\"workload\": \"%zu synthetic OS1-128 scans (%zu pts), %d^3 sliding map @ %d mm, C++ warpsense::App (examples/replay_bench.cpp)\", "
         "\"hz\": %.1f, \"async_shift\": %s, \"scans_per_s\": %.2f, \"stream_s\": %.4f, \"callback_busy_s\": %.4f, \"setup_s\": %.2f, "
         "\"preprocess_ms\": %.4f, \"tsdf_ms\": %.4f, \"registration_ms\": %.4f, \"shift_ms\": %.4f, \"total_ms\": %.4f, "
         "\"slowest_scan_ms\": %.3f, \"scans_over_100ms\": %d, \"tsdf_updates\": %d, \"map_shifts\": %d, \"iterations_mean\": %.1f, "
         "\"device\": {\"update_ms_per_update\": %.4f, \"registration_ms_per_scan\": %.4f, \"us_per_iteration\": %.3f, "
         "\"note\": \"hipEvent spans on the library's stream (ws_prof_*): the kernels of the update / the registration loop of the same scans\"}, "
         "\"host_registration_us_per_iteration\": %.3f}\n",
         scans, n, edge, p.hot.map_resolution, hz, p.async_shift ? "true" : "false", (double)scans / stream_s, stream_s, busy_s, setup_s, sum_pre / m / 1e3,
         sum_tsdf / m / 1e3, sum_reg / m / 1e3, sum_shift / m / 1e3, sum_total / m / 1e3, worst / 1e3, over, app.n_updates(), app.n_shifts(), (double)its / m,
         n_upd ? dev_upd / n_upd : 0.0, dev_reg / m, its ? 1e3 * dev_reg / (double)its : 0.0, its ? sum_reg / (double)its : 0.0);
  app.terminate();
  return 0;
}

// dropin_bench.cpp — scans/s of the drop-in as a maintainer of the reference gets it (VERDICT r4 missing #3 / weak #5):
//
//   relink_only       the reference's callers UNCHANGED: TSDFCuda::update_tsdf(const std::vector<Pointi>&, ...) with its per-scan
//                     host -> device copy (update_tsdf.cu:152-154) and TSDFRegistration::register_cloud's own loop
//                     (tsdf_registration.cpp:55-92): prepare_registration(host vector), then per iteration one
//                     perform_registration (a launch, 44 sums back to the host) + the 6x6 solve and xi_to_transform on the host
//   one_line_change   the same host vectors, register_cloud's loop replaced by the resident device loop (INTEGRATION.md §1:
//                     RegistrationCuda::register_cloud -- the one call the maintainer swaps in)
//   device_clouds     clouds already resident in HBM (update_tsdf_dev / prepare_registration_dev): the route bench.py's `value` times
//
//   dropin_bench <scan.bin> <perturbed.bin> <n> <map_edge_voxels> <res> <tau> <max_weight> <scans>
//
// prints ONE JSON line.  Every route's final pose and iteration count are checked against each other.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "warpsense_hip/mapping.hpp"

namespace rm = rmagine;

static std::vector<rm::Pointi> read_points(const char *path, size_t n)
{
  std::vector<rm::Pointi> pts(n);
  std::ifstream f(path, std::ios::binary);
  f.read(reinterpret_cast<char *>(pts.data()), (std::streamsize)(n * sizeof(rm::Pointi)));
  if (!f)
  {
    fprintf(stderr, "cannot read %zu points from %s\n", n, path);
    exit(2);
  }
  return pts;
}

int main(int argc, char **argv)
{
  if (argc != 9)
  {
    fprintf(stderr, "usage: %s scan.bin perturbed.bin n edge res tau max_weight scans\n", argv[0]);
    return 2;
  }
  const size_t n = strtoull(argv[3], nullptr, 10);
  int edge = atoi(argv[4]);
  cuda::HotPathParams params;
  params.map_resolution = atoi(argv[5]);
  params.tau = atoi(argv[6]);
  params.max_weight = atoi(argv[7]);
  const int scans = atoi(argv[8]);
  if (edge % 2 == 0) edge += 1;
  rm::Pointi size(edge, edge, edge), pos(0, 0, 0), offset(edge / 2, edge / 2, edge / 2);
  // (data_ == nullptr: the library fills both device maps with the default entry -- no 540 MB host image)
  cuda::DeviceMap local_map(&size, &offset, nullptr, &pos);
  auto scan = read_points(argv[1], n);
  auto perturbed = read_points(argv[2], n);
  cuda::TSDFRegistration gpu(params, local_map);
  rm::Matrix4x4f pose;
  pose.setIdentity();
  rm::Pointi pos_rm, up_rm;
  gpu.convert_pose_to_gpu(pose, pos_rm, up_rm);

  // device copies for the third route: two registration handles used as device buffers (the C ABI has no allocator of its own)
  ws_reg *hold_scan = nullptr, *hold_pert = nullptr;
  if (ws_reg_create(cuda::detail::context(), n, &hold_scan) != WS_OK || ws_reg_create(cuda::detail::context(), n, &hold_pert) != WS_OK ||
      ws_reg_prepare(hold_scan, &scan[0].x, n) != WS_OK || ws_reg_prepare(hold_pert, &perturbed[0].x, n) != WS_OK)
  {
    fprintf(stderr, "device buffers: %s\n", ws_last_error());
    return 1;
  }
  size_t n_dev = 0;
  const int32_t *scan_dev = ws_reg_points_dev(hold_scan, &n_dev), *pert_dev = ws_reg_points_dev(hold_pert, &n_dev);

  rm::Matrix4x4f T[3];
  int its[3] = {0, 0, 0};
  double secs[3] = {0, 0, 0};
  for (int route = 0; route < 3; ++route)
  {
    auto step = [&]() {
      if (route == 2)
      {
        gpu.tsdf().update_tsdf_dev(scan_dev, n, pos_rm, up_rm);
        gpu.registration().prepare_registration_dev(pert_dev, n);
        T[route] = gpu.registration().register_cloud(gpu.tsdf().device_map(), pose, params.max_iterations, params.it_weight_gradient, params.epsilon,
                                                     params.map_resolution, &its[route]);
        return;
      }
      gpu.update_tsdf(scan, pos_rm, up_rm);
      if (route == 0)
        T[route] = gpu.register_cloud_reference_loop(perturbed, pose);
      else
        T[route] = gpu.register_cloud(perturbed, pose);
      its[route] = gpu.last_iterations();
    };
    for (int w = 0; w < 2; ++w) step();
    cuda::pause();
    const auto t0 = std::chrono::steady_clock::now();
    for (int s = 0; s < scans; ++s) step();
    cuda::pause();
    secs[route] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  const bool same = memcmp(&T[0], &T[1], sizeof(T[0])) == 0 && memcmp(&T[0], &T[2], sizeof(T[0])) == 0 && its[0] == its[1] && its[0] == its[2];
  // the update alone with a host vector (H2D included), for the per-iteration figure of the host-driven loop
  for (int w = 0; w < 2; ++w) gpu.update_tsdf(scan, pos_rm, up_rm);
  cuda::pause();
  const auto t1 = std::chrono::steady_clock::now();
  for (int s = 0; s < scans; ++s) gpu.update_tsdf(scan, pos_rm, up_rm);
  cuda::pause();
  const double upd = std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count() / scans;
  printf("{\"scans\": %d, \"iterations\": %d, \"routes_agree\": %s, "
         "\"relink_only\": {\"scans_per_s\": %.2f, \"ms_per_scan\": %.4f, \"us_per_iteration\": %.3f}, "
         "\"one_line_change\": {\"scans_per_s\": %.2f, \"ms_per_scan\": %.4f}, "
         "\"device_clouds\": {\"scans_per_s\": %.2f, \"ms_per_scan\": %.4f}, "
         "\"update_with_host_vector_ms\": %.4f}\n",
         scans, its[0], same ? "true" : "false", scans / secs[0], 1e3 * secs[0] / scans, 1e6 * (secs[0] / scans - upd) / (its[0] > 0 ? its[0] : 1),
         scans / secs[1], 1e3 * secs[1] / scans, scans / secs[2], 1e3 * secs[2] / scans, 1e3 * upd);
  ws_reg_destroy(hold_scan);
  ws_reg_destroy(hold_pert);
  return same ? 0 : 3;
}

// Forwarding header: takes the place of the reference's include/warpsense/cuda/update_tsdf.h -- cuda::TSDFCuda (update_tsdf.h:9-34) --
// and hands the declarations over to the MI355X implementation (include/warpsense_hip/compat.hpp over the C ABI of
// libwarpsense_hip.so).  Two ways to use it:
//   * put <this repo>/include AHEAD of the reference's include directory: every #include "warpsense/cuda/update_tsdf.h" of the
//     reference's sources (e.g. src/warpsense/tsdf_mapping.cpp:1, test/pcd2tsdf.cpp:20) then resolves to this file;
//   * copy the five files of this directory over the reference's own, which also covers the reference headers that reach
//     them by a path relative to themselves (include/warpsense/tsdf_mapping.h:7-9 includes "cuda/device_map.h").
// Inside the reference tree the rmagine:: math types and TSDFEntry stay the reference's own.
#pragma once
#if !defined(WARPSENSE_HIP_USE_REFERENCE_TYPES) && defined(__has_include)
#if __has_include("warpsense/math/math.h") && __has_include("map/tsdf.h")
#define WARPSENSE_HIP_USE_REFERENCE_TYPES
#endif
#endif
#include "warpsense_hip/compat.hpp"

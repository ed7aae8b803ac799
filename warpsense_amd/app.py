"""ROS-free replay node: the sequencing of warpsense::App (src/warpsense/app.cpp:30-176) over a stream of sensor
clouds, with the whole scan -> pose pipeline on the device.

    app = App(params, "/tmp/map.h5")
    for cloud in clouds:                      # (n, >=3) float32, metres, sensor frame
        app.cloud_callback(cloud)             # preprocess -> update_tsdf (if moved) -> register_cloud -> pose
    app.terminate()                           # write the local map back to the global map file

What ROS provided is passed in explicitly: the IMU pre-transform of `imu_acc_.acc_transform(stamp)` is the optional
`pretransform` argument (identity = no IMU), and the map-shift thread of TSDFMapping::map_shift
(tsdf_mapping.cpp:97-136) runs synchronously after every scan instead of polling a one-slot pose buffer, which makes
a replay deterministic.
"""
from __future__ import annotations

import time

import numpy as np

from .api import (GlobalMap, LocalMap, Params, ScanPreprocessor, TSDFRegistration, to_map)


class App:
    def __init__(self, params: Params, filename: str | None = None, ctx=None, max_points: int = 128 * 1024, async_shift: bool = False):
        # async_shift: TSDFMapping.shift_map_async — the window moves on the device inside the scan that triggers it and
        # the leaving slabs are filed into the global map by a worker thread (same maps and poses as the synchronous route)
        self.async_shift_ = bool(async_shift)
        m = params.map
        self.params_ = params
        # app.cpp:33-41: global map (file), local map around the origin, the GPU mapping/registration object
        self.hdf5_global_map_ = GlobalMap(m.tau, m.initial_weight, filename=filename, map_params=m if filename else None)
        self.hdf5_local_map_ = LocalMap(m.size[0], m.size[1], m.size[2], m.tau, m.initial_weight, self.hdf5_global_map_)
        self.gpu_ = TSDFRegistration(params, self.hdf5_local_map_, ctx)
        self.pre_ = ScanPreprocessor(max_points, ctx)
        if self.async_shift_:
            self.gpu_.reserve_shift(int(np.ceil(m.shift * 1000.0 / m.resolution)))
        self.pose_ = np.eye(4, dtype=np.float32)            # mm
        self.last_tsdf_pose_ = np.eye(4, dtype=np.float32)
        self.last_shift_pose_ = np.eye(4, dtype=np.float32)
        self.initialized_ = False
        self.shifted_ = False
        self.poses = []       # pose_ after every scan
        self.timings = []     # per scan: dict of seconds (the reference's RuntimeEvaluator forms)
        self.n_updates = 0
        self.n_shifts = 0

    def preprocess(self, cloud):
        """App::preprocess (app.cpp:119-148) -> points resident on the device."""
        return self.pre_.preprocess(cloud, self.pose_, self.params_.map.resolution)

    def update_pose_estimate(self, transform):
        """app.cpp:172-176."""
        T = np.asarray(transform, dtype=np.float32)
        f = np.float32
        R = np.zeros((3, 3), dtype=np.float32)
        for i in range(3):  # float32 products summed in index order (same as include/warpsense_hip/app.hpp)
            for j in range(3):
                acc = f(0)
                for k in range(3):
                    acc = f(acc + f(T[i, k] * self.pose_[k, j]))
                R[i, j] = acc
        self.pose_[:3, :3] = R
        self.pose_[:3, 3] += T[:3, 3]

    def map_shift(self):
        """One turn of TSDFMapping::map_shift (tsdf_mapping.cpp:104-127) for the current pose."""
        d = np.linalg.norm(self.last_shift_pose_[:3, 3] / np.float32(1000) - self.pose_[:3, 3] / np.float32(1000))
        if d >= self.params_.map.shift:
            self.last_shift_pose_ = self.pose_.copy()
            if self.async_shift_:
                self.gpu_.shift_map_async(to_map(self.pose_, self.params_.map.resolution))
            else:
                self.gpu_.shift_map(to_map(self.pose_, self.params_.map.resolution))
            self.shifted_ = True
            self.n_shifts += 1

    def cloud_callback(self, cloud, pretransform=None):
        """App::cloud_callback (app.cpp:65-117)."""
        t = {}
        t0 = time.perf_counter()
        scan_points = self.preprocess(cloud)
        t["preprocess"] = time.perf_counter() - t0
        distance_tsdf = np.linalg.norm(self.last_tsdf_pose_[:3, 3] / np.float32(1000) - self.pose_[:3, 3] / np.float32(1000))
        if not self.initialized_ or distance_tsdf > 0.3 or self.shifted_:
            self.initialized_ = True
            self.last_tsdf_pose_ = self.pose_.copy()
            t1 = time.perf_counter()
            self.gpu_.update_tsdf(scan_points, pose=self.pose_)
            t["tsdf"] = time.perf_counter() - t1
            self.shifted_ = False
            self.n_updates += 1
        pre = np.eye(4, dtype=np.float32) if pretransform is None else np.asarray(pretransform, dtype=np.float32)
        t2 = time.perf_counter()
        transform = self.gpu_.register_cloud(scan_points, pre)
        t["registration"] = time.perf_counter() - t2
        self.update_pose_estimate(transform)
        if self.hdf5_global_map_.filename() is not None:
            self.hdf5_global_map_.write_pose(self.pose_, 1000.0)
        self.map_shift()
        t["total"] = time.perf_counter() - t0
        t["points"] = len(scan_points)
        t["iterations"] = self.gpu_.last_iterations
        self.poses.append(self.pose_.copy())
        self.timings.append(t)
        return self.pose_

    def terminate(self):
        """App::terminate (app.cpp:192-224): write the map; here straight from the device map."""
        self.gpu_.wait_shift()
        if self.initialized_:
            self.gpu_.write_back()
        self.hdf5_global_map_.close()

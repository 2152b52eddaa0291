#!/usr/bin/env python3
"""The data path of stereomapper on the GPU, frame by frame, from a KITTI raw drive:

    drive (include/svh_kitti.h)  ->  VisualOdometryStereo::process  (pose, gain; include/svh.h svh_vo_*)
                                 ->  Elas::process                  (D1 on the device; svh_elas_*)
                                 ->  map fusion                     (point lists; include/svh_map.h)

i.e. what ReadFromFilesThread, VisualOdometryThread and StereoThread do between them
(readfromfilesthread.cpp:25-112, visualodometrythread.cpp:95-140, stereothread.cpp:62-170),
without the GUI.  Usage:

    python tools/stereomapper_pipeline.py <drive_dir> <calib_cam_to_cam.txt> [max_frames]

Python is glue here (ctypes over libsvhip.so); the pose accumulation H_total = H_total * inv(H_delta)
uses numpy where the reference uses Matrix::solve."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


class DeviceBuffer:
    """device memory through the HIP runtime libsvhip has loaded"""

    def __init__(self, nbytes):
        self.hip = C.CDLL("libamdhip64.so")
        self.ptr = C.c_void_p()
        if self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(nbytes)):
            raise MemoryError("hipMalloc")
        self.nbytes = nbytes

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        assert self.hip.hipMemcpy(self.ptr, C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes), 1) == 0

    def download(self, arr):
        assert self.hip.hipMemcpy(C.c_void_p(arr.ctypes.data), self.ptr, C.c_size_t(arr.nbytes), 2) == 0
        return arr

    def __del__(self):
        if self.ptr:
            self.hip.hipFree(self.ptr)


class Pipeline:
    def __init__(self, f, cu, cv, base, elas_params=None, max_dist=20.0):
        import helpers as Hh
        import svhip as S
        from svhip import mapper
        self.S = S
        self.vo = Hh.ProductVo(Hh.vo_defaults(f=f, cu=cu, cv=cv, base=base))
        self.elas = S.Elas(elas_params if elas_params is not None else Hh.robotics())
        self.map = mapper.Mapper(f, cu, cv, base, max_dist)
        self.H_total = np.eye(4)
        self.buf = None
        self.poses = []

    def push(self, I1, I2):
        """one stereo frame; returns (vo_ok, points_prev, points_curr)"""
        h, w = I1.shape
        n = w * h
        if self.buf is None or self.buf[0].nbytes != n:
            self.buf = [DeviceBuffer(n), DeviceBuffer(n), DeviceBuffer(4 * n), DeviceBuffer(4 * n)]
        # visualodometrythread.cpp:100-137
        ok = self.vo.process(I1, I2) == 1
        gain = 0.0
        if ok:
            Hd = self.vo.motion()
            gain = float(self.vo.gain(self.vo.inliers()))
            try:
                self.H_total = self.H_total @ np.linalg.inv(Hd)
            except np.linalg.LinAlgError:
                pass
        self.poses.append(self.H_total.copy())
        # stereothread.cpp:62-115: ELAS with the disparity maps left on the device
        dI1, dI2, dD1, dD2 = self.buf
        dI1.upload(I1)
        dI2.upload(I2)
        st = self.elas.process_batch_device(1, dI1.ptr.value, dI2.ptr.value, n, dD1.ptr.value, dD2.ptr.value,
                                            4 * n, w, h, w)
        if st[0] != 0:
            return ok, 0, 0
        # stereothread.cpp:166-170
        self.map.add(None, I1, self.H_total, gain, device_ptr=dD1.ptr.value)
        return ok, self.map._L.svh_map_points(self.map._h, 0, None, 0), self.map._L.svh_map_points(self.map._h, 1, None, 0)


def main():
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    from svhip import kitti
    calib = kitti.read_cam_to_cam(sys.argv[2])
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 30
    p = Pipeline(calib.f, calib.cu, calib.cv, calib.base)
    t0 = time.perf_counter()
    frames = 0
    for I1, I2, (tl, _) in kitti.Sequence(sys.argv[1]):
        ok, n0, n1 = p.push(I1, I2)
        frames += 1
        print("frame %4d  t=%.3f  vo=%d  pose z=%.2f  points kept=%d new=%d" % (
            frames - 1, tl, ok, p.H_total[2, 3], n0, n1))
        if frames >= limit:
            break
    dt = time.perf_counter() - t0
    print("%d frames in %.2f s = %.1f frames/s (PNG decode included)" % (frames, dt, frames / max(dt, 1e-9)))


if __name__ == "__main__":
    main()

"""ctypes binding of the KITTI raw reader in libsvhip.so (C-ABI: include/svh_kitti.h).

Mirrors what stereomapper's playback thread needs from a drive directory
(readfromfilesthread.cpp:25-112): the rectified calibration, and the gray stereo frames with
their capture times, in order.
"""
import ctypes as C

import numpy as np

from . import SvhError, lib

CAMERAS = 4


class Calib(C.Structure):
    """svh_kitti_calib: calib_cam_to_cam.txt (calibiokitti.cpp:227-262) + the derived rig."""
    _fields_ = [
        ("calib_time", C.c_char * 64), ("corner_dist", C.c_double),
        ("S", C.c_double * 2 * CAMERAS), ("K", C.c_double * 9 * CAMERAS), ("D", C.c_double * 5 * CAMERAS),
        ("R", C.c_double * 9 * CAMERAS), ("T", C.c_double * 3 * CAMERAS),
        ("S_rect", C.c_double * 2 * CAMERAS), ("R_rect", C.c_double * 9 * CAMERAS),
        ("P_rect", C.c_double * 12 * CAMERAS),
        ("f", C.c_double), ("cu", C.c_double), ("cv", C.c_double), ("base", C.c_double),
    ]

    def matrix(self, name, cam):
        shape = {"S": (1, 2), "K": (3, 3), "D": (1, 5), "R": (3, 3), "T": (1, 3),
                 "S_rect": (1, 2), "R_rect": (3, 3), "P_rect": (3, 4)}[name]
        return np.array(getattr(self, name)[cam][:], np.float64).reshape(shape)


def _bind():
    L = lib()
    if not getattr(L, "_kitti_bound", False):
        L.svh_kitti_read_cam_to_cam.argtypes = [C.c_char_p, C.POINTER(Calib)]
        L.svh_png_read_gray.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.svh_kitti_seq_open.restype = C.c_void_p
        L.svh_kitti_seq_open.argtypes = [C.c_char_p]
        L.svh_kitti_seq_close.argtypes = [C.c_void_p]
        L.svh_kitti_seq_count.argtypes = [C.c_void_p]
        L.svh_kitti_seq_seek.argtypes = [C.c_void_p, C.c_int32]
        L.svh_kitti_seq_next.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L._kitti_bound = True
    return L


def read_cam_to_cam(path):
    c = Calib()
    rc = _bind().svh_kitti_read_cam_to_cam(str(path).encode(), C.byref(c))
    if rc:
        raise SvhError(rc, "cannot read calibration " + str(path))
    return c


def read_png_gray(path):
    L = _bind()
    w, h = C.c_int32(), C.c_int32()
    rc = L.svh_png_read_gray(str(path).encode(), None, 0, C.byref(w), C.byref(h))
    if rc:
        raise SvhError(rc, "cannot read " + str(path))
    img = np.empty((h.value, w.value), np.uint8)
    rc = L.svh_png_read_gray(str(path).encode(), img.ctypes.data, img.size, C.byref(w), C.byref(h))
    if rc:
        raise SvhError(rc, "cannot decode " + str(path))
    return img


class Sequence:
    """for I1, I2, (t_left, t_right) in Sequence(drive_dir): ...   (times in seconds of day)"""

    def __init__(self, drive_dir, max_pixels=4096 * 2048):
        self._L = _bind()
        self._h = self._L.svh_kitti_seq_open(str(drive_dir).encode())
        if not self._h:
            raise SvhError(-1, "not a KITTI raw drive: " + str(drive_dir))
        self._cap = max_pixels

    def __len__(self):
        return self._L.svh_kitti_seq_count(self._h)

    def __iter__(self):
        return self

    def __next__(self):
        a = np.empty(self._cap, np.uint8)
        b = np.empty(self._cap, np.uint8)
        dims = (C.c_int32 * 3)()
        tv = (C.c_int64 * 4)()
        rc = self._L.svh_kitti_seq_next(self._h, a.ctypes.data, b.ctypes.data, self._cap, dims, tv)
        if rc == 1:
            raise StopIteration
        if rc:
            raise SvhError(rc, "frame could not be read")
        w, h = dims[0], dims[1]
        times = (tv[0] + tv[1] * 1e-6, tv[2] + tv[3] * 1e-6)
        return a[:w * h].reshape(h, w).copy(), b[:w * h].reshape(h, w).copy(), times

    def seek(self, frame):
        if self._L.svh_kitti_seq_seek(self._h, int(frame)):
            raise SvhError(-1, "no frame %d" % frame)

    def close(self):
        if self._h:
            self._L.svh_kitti_seq_close(self._h)
            self._h = None

    def __del__(self):
        self.close()


def load_shard(drive_dir, rank=0, world=1):
    """This rank's contiguous share of a drive: (I1 [n,h,w], I2 [n,h,w], first frame, total frames).
    Only the rank's own frames are decoded."""
    from .shard import shard_range
    seq = Sequence(drive_dir)
    total = len(seq)
    lo, hi = shard_range(total, rank, world)
    left, right = [], []
    seq.seek(lo)
    for _ in range(lo, hi):
        a, b, _t = next(seq)
        left.append(a)
        right.append(b)
    seq.close()
    if not left:
        raise SvhError(-1, "no frames for rank %d of %d in %s" % (rank, world, drive_dir))
    return np.stack(left), np.stack(right), lo, total

"""ctypes binding of the map fusion in libsvhip.so (C-ABI: include/svh_map.h): what
stereomapper's StereoThread does with a finished disparity map (stereothread.cpp:166-170)."""
import ctypes as C

import numpy as np

from . import SvhError, last_error, lib


class MapParams(C.Structure):
    """svh_map_params: the float members StereoThread keeps (stereothread.h:192-196)"""
    _fields_ = [("f", C.c_float), ("cu", C.c_float), ("cv", C.c_float), ("base", C.c_float),
                ("max_dist", C.c_float)]


def _bind():
    L = lib()
    if not getattr(L, "_map_bound", False):
        L.svh_map_create.restype = C.c_void_p
        L.svh_map_create.argtypes = [C.POINTER(MapParams)]
        L.svh_map_destroy.argtypes = [C.c_void_p]
        L.svh_map_clear.argtypes = [C.c_void_p]
        L.svh_map_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.svh_map_points.restype = C.c_int64
        L.svh_map_points.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        L.svh_map_planes.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.svh_disparity_colormap.argtypes = [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p]
        L._map_bound = True
    return L


class Mapper:
    """m = Mapper(f, cu, cv, base); m.add(D1, I1, H_total, gain); m.points(0 / 1)"""

    def __init__(self, f, cu, cv, base, max_dist=20.0):
        self._L = _bind()
        self.params = MapParams(f, cu, cv, base, max_dist)
        self._h = self._L.svh_map_create(C.byref(self.params))
        if not self._h:
            raise SvhError(-4, last_error())
        self._shape = None

    def add(self, D1, I1, H_total, gain=0.0, device_ptr=None):
        """D1: host float32 [h, w] (or device_ptr = raw device address of such a map)"""
        I1 = np.asarray(I1, np.uint8)
        h, w = I1.shape
        H = np.ascontiguousarray(H_total, np.float64)
        dims = (C.c_int32 * 3)(w, h, I1.strides[0])
        if device_ptr is None:
            D1 = np.ascontiguousarray(D1, np.float32)
            assert D1.shape == (h, w)
            rc = self._L.svh_map_add(self._h, D1.ctypes.data, 0, I1.ctypes.data, dims, H.ctypes.data, gain)
        else:
            rc = self._L.svh_map_add(self._h, device_ptr, 1, I1.ctypes.data, dims, H.ctypes.data, gain)
        if rc:
            raise SvhError(rc, last_error())
        self._shape = (h, w)

    def points(self, which):
        n = self._L.svh_map_points(self._h, which, None, 0)
        out = np.zeros((n, 4), np.float32)
        if n:
            self._L.svh_map_points(self._h, which, out.ctypes.data, n)
        return out

    def planes(self):
        h, w = self._shape
        out = np.zeros((5, h, w), np.float32)
        rc = self._L.svh_map_planes(self._h, out.ctypes.data, out.size)
        if rc:
            raise SvhError(rc, last_error())
        return out

    def clear(self):
        self._L.svh_map_clear(self._h)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.svh_map_destroy(h)


def disparity_colormap(D):
    """[h, w] float32 disparities -> [h, w, 3] float32 colours (stereothread.cpp:117-147)"""
    D = np.ascontiguousarray(D, np.float32)
    out = np.zeros(D.shape + (3,), np.float32)
    rc = _bind().svh_disparity_colormap(D.ctypes.data, 0, D.size, out.ctypes.data)
    if rc:
        raise SvhError(rc, last_error())
    return out

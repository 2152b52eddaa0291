"""ctypes binding of libsvhip.so (C-ABI: include/svh.h).

Python is plumbing here: the product is the shared library (HIP kernels for
gfx950 + C++ host engine).  This module mirrors the reference's class surface
(`Elas(parameters).process(I1, I2, D1, D2, dims)`, libelas/src/elas.h:151-165) so
parity tests read like reference call sites.  There is NO CPU fallback: a
missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SVH_LIB: another build of the library (A/B measurements of kernel changes)
LIB_PATH = os.environ.get("SVH_LIB") or os.path.join(os.path.dirname(_HERE), "libsvhip.so")

OK, ERR_FEW_SUPPORT, ERR_BAD_ARG, ERR_HIP, ERR_UNSUPPORTED, ERR_NO_DEVICE = 0, 1, -1, -2, -3, -4
ERR_BAD_DIMS = -7    # Matcher::pushBack / prefetch with bad dimensions (the reference's message, call ignored)
ROBOTICS, MIDDLEBURY = 0, 1


class ElasParams(C.Structure):
    """svh_elas_params == Elas::parameters (libelas/src/elas.h:59-148)."""
    _fields_ = [
        ("disp_min", C.c_int32), ("disp_max", C.c_int32),
        ("support_threshold", C.c_float), ("support_texture", C.c_int32),
        ("candidate_stepsize", C.c_int32), ("incon_window_size", C.c_int32),
        ("incon_threshold", C.c_int32), ("incon_min_support", C.c_int32),
        ("add_corners", C.c_int32), ("grid_size", C.c_int32),
        ("beta", C.c_float), ("gamma", C.c_float), ("sigma", C.c_float), ("sradius", C.c_float),
        ("match_texture", C.c_int32), ("lr_threshold", C.c_int32),
        ("speckle_sim_threshold", C.c_float), ("speckle_size", C.c_int32),
        ("ipol_gap_width", C.c_int32), ("filter_median", C.c_int32),
        ("filter_adaptive_mean", C.c_int32), ("postprocess_only_left", C.c_int32),
        ("subsampling", C.c_int32),
    ]


class Config(C.Structure):
    """svh_config (include/svh.h): the process-wide settings svh_init fixes."""
    _fields_ = [("size", C.c_uint32), ("hw_queues", C.c_int32), ("elas_workers", C.c_int32),
                ("elas_pairs_per_launch", C.c_int32), ("elas_stage", C.c_int32), ("wait_us", C.c_int32),
                ("read_env", C.c_int32), ("reserved_", C.c_int32 * 9)]


class RuntimeInfo(C.Structure):
    """svh_runtime_info (include/svh.h)"""
    _fields_ = [("initialised", C.c_int32), ("implicit", C.c_int32), ("hw_queues_asked", C.c_int32),
                ("hw_queues_state", C.c_int32), ("hw_queues_env", C.c_int32), ("hip_started_before", C.c_int32),
                ("env_modified", C.c_int32), ("read_env", C.c_int32), ("reserved_", C.c_int32 * 8)]

    def as_dict(self):
        names = {0: "none", 1: "applied", 2: "caller_set", 3: "too_late", 4: "hands_off"}
        d = {n: getattr(self, n) for n, _ in self._fields_[:-1]}
        d["hw_queues_state"] = names.get(self.hw_queues_state, self.hw_queues_state)
        return d


def init(**fields):
    """svh_init(&cfg) with the given svh_config fields over the defaults.  Call it before anything in the process
    starts the HIP runtime (e.g. before `import torch`): the hardware-queue count can only be asked for until then."""
    cfg = Config()
    lib().svh_config_default(C.byref(cfg))
    for k, v in fields.items():
        setattr(cfg, k, v)
    rc = lib().svh_init(C.byref(cfg))
    if rc < 0:
        raise SvhError(rc, "svh_init: bad configuration")
    return runtime_info()


class DeviceTopology(C.Structure):
    """svh_device_topology (include/svh.h)"""
    _fields_ = [("device", C.c_int32), ("numa_node", C.c_int32), ("n_cpus", C.c_int32), ("reserved_", C.c_int32),
                ("pci_bus_id", C.c_char * 32), ("cpulist", C.c_char * 256), ("cpu_mask", C.c_uint64 * 16)]


def device_topology(device):
    """PCI bus id, NUMA node and the node's CPUs of a HIP device (svh_get_device_topology)"""
    t = DeviceTopology()
    rc = lib().svh_get_device_topology(int(device), C.byref(t))
    if rc < 0:
        raise SvhError(rc, "svh_get_device_topology")
    return {"device": t.device, "pci_bus_id": t.pci_bus_id.decode(), "numa_node": t.numa_node, "cpus_of_node": t.n_cpus,
            "cpulist": t.cpulist.decode()}


def bind_host_to_device(device, max_cpus=0):
    """restrict this process (and the threads it creates from now on) to the CPUs next to the GPU; returns how many"""
    return int(lib().svh_bind_host_to_device(int(device), int(max_cpus)))


def elas_settings():
    """svh_elas_get_settings: workers, pairs per launch (0 = automatic), stage, poll interval [us]"""
    out = (C.c_int32 * 4)()
    lib().svh_elas_get_settings(out)
    return dict(workers=out[0], pairs_per_launch=out[1], stage=out[2], wait_us=out[3])


def runtime_info():
    ri = RuntimeInfo()
    lib().svh_get_runtime_info(C.byref(ri))
    return ri.as_dict()


class SvhError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsvhip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    """Load libsvhip.so; fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libsvhip.so not built (run __graft_entry__.build()): " + LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.svh_version.restype = C.c_char_p
        L.svh_last_error.restype = C.c_char_p
        L.svh_elas_create.restype = C.c_void_p
        L.svh_elas_create.argtypes = [C.POINTER(ElasParams)]
        L.svh_elas_destroy.argtypes = [C.c_void_p]
        L.svh_elas_params_default.argtypes = [C.POINTER(ElasParams), C.c_int32]
        L.svh_elas_process.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.svh_elas_process_batch.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 6
        L.svh_elas_process_batch_device.argtypes = [
            C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
            C.c_size_t, C.c_void_p, C.c_void_p]
        L.svh_elas_stream_open.restype = C.c_void_p
        L.svh_elas_stream_open.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.svh_elas_stream_push.argtypes = [C.c_void_p] * 5 + [C.POINTER(C.c_uint64)]
        L.svh_elas_stream_push_device.argtypes = [C.c_void_p] * 5 + [C.POINTER(C.c_uint64)]
        L.svh_elas_stream_flush.argtypes = [C.c_void_p]
        L.svh_elas_stream_pop.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.c_int32]
        L.svh_elas_stream_close.argtypes = [C.c_void_p]
        L.svh_elas_stream_push_device_n.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_size_t,
                                                    C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64)]
        L.svh_elas_stream_pop_n.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32)]
        L.svh_elas_stream_push_n.argtypes = [C.c_void_p, C.c_int32] + [C.c_void_p] * 4 + [C.POINTER(C.c_uint64)]
        L.svh_elas_set_taps.argtypes = [C.c_void_p, C.c_int32]
        L.svh_elas_get_stage.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_size_t,
                                         C.POINTER(C.c_size_t)]
        L.svh_elas_last_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
        L.svh_delaunay.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        _lib = L
    return _lib


SVH_ERR_EMPTY, SVH_ERR_TIMEOUT = -5, -6


class SvhTimeout(SvhError):
    """svh_elas_stream_pop: the next pair was not done within timeout_ms (it stays queued)"""


class ElasStream:
    """svh_elas_stream_*: a bounded, ordered queue of pairs in front of the engine's lanes
    (include/svh.h).  Host arrays pushed here must stay alive until their pair is popped: the
    stream keeps a reference to them."""

    def __init__(self, elas, w, h, pitch, depth=0):
        dims = (C.c_int32 * 3)(w, h, pitch)
        self._h = lib().svh_elas_stream_open(elas._h, dims, depth)
        if not self._h:
            raise SvhError(-1, last_error())
        self._keep = {}

    def push(self, I1, I2, D1, D2):
        """host arrays (uint8 [H,W] with the stream's pitch; float32 maps written in place)"""
        t = C.c_uint64(0)
        rc = lib().svh_elas_stream_push(self._h, I1.ctypes.data, I2.ctypes.data, D1.ctypes.data,
                                        D2.ctypes.data, C.byref(t))
        if rc < 0:
            raise SvhError(rc, last_error())
        self._keep[t.value] = (I1, I2, D1, D2)
        return t.value

    def push_device(self, dI1, dI2, dD1, dD2):
        """raw device pointers (ints)"""
        t = C.c_uint64(0)
        rc = lib().svh_elas_stream_push_device(self._h, dI1, dI2, dD1, dD2, C.byref(t))
        if rc < 0:
            raise SvhError(rc, last_error())
        return t.value

    def push_device_n(self, n, dI1, dI2, in_stride, dD1, dD2, out_stride):
        """n consecutive device-resident pairs in one call (blocks while the stream is full)"""
        t = C.c_uint64(0)
        rc = lib().svh_elas_stream_push_device_n(self._h, n, dI1, dI2, in_stride, dD1, dD2, out_stride, C.byref(t))
        if rc < 0:
            raise SvhError(rc, last_error())
        return t.value

    def push_n(self, I1s, I2s, D1s, D2s):
        """n host pairs in one call: arrays [n,H,W] (uint8 images, float32 maps written in place); they must stay
        alive and unread until popped -- the stream keeps a reference"""
        n = len(I1s)
        arr = C.c_void_p * n
        a = [arr(*[int(X[i].ctypes.data) for i in range(n)]) for X in (I1s, I2s, D1s, D2s)]
        t = C.c_uint64(0)
        rc = lib().svh_elas_stream_push_n(self._h, n, a[0], a[1], a[2], a[3], C.byref(t))
        if rc < 0:
            raise SvhError(rc, last_error())
        self._keep[("n", t.value)] = (I1s, I2s, D1s, D2s)
        return t.value

    def push_n_raw(self, n, a1, a2, d1, d2):
        """the same with ready-made ctypes pointer arrays (no per-call marshalling); the caller keeps the buffers alive"""
        t = C.c_uint64(0)
        rc = lib().svh_elas_stream_push_n(self._h, n, a1, a2, d1, d2, C.byref(t))
        if rc < 0:
            raise SvhError(rc, last_error())
        return t.value

    def pop_n(self, n):
        """statuses of the next n pairs (blocks until they are done)"""
        st = (C.c_int32 * n)()
        got = C.c_int32(0)
        rc = lib().svh_elas_stream_pop_n(self._h, n, st, C.byref(got))
        if rc < 0:
            raise SvhError(rc, last_error())
        return list(st)[:got.value]

    def flush(self):
        lib().svh_elas_stream_flush(self._h)

    def pop(self, timeout_ms=-1):
        """(ticket, status) of the next pair in submission order; None when nothing is in flight"""
        t, st = C.c_uint64(0), C.c_int32(0)
        rc = lib().svh_elas_stream_pop(self._h, C.byref(t), C.byref(st), timeout_ms)
        if rc == SVH_ERR_EMPTY:
            return None
        if rc == SVH_ERR_TIMEOUT:
            raise SvhTimeout(rc, "timeout")
        if rc < 0:
            raise SvhError(rc, last_error())
        self._keep.pop(t.value, None)
        return t.value, st.value

    def close(self):
        h, self._h = self._h, None
        if h:
            lib().svh_elas_stream_close(h)
        self._keep.clear()

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:
            self.close()


def last_error():
    return lib().svh_last_error().decode()


def default_params(setting=ROBOTICS, **kw):
    p = ElasParams()
    lib().svh_elas_params_default(C.byref(p), setting)
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def delaunay(pts):
    """svh_delaunay: Triangle-1.6-"zQB"-compatible triangulation (host side)."""
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    cap = 2 * len(pts) + 16
    tri = np.empty((cap, 3), np.int32)
    n = lib().svh_delaunay(pts.ctypes.data, len(pts), tri.ctypes.data, cap)
    if n < 0:
        raise SvhError(n, "svh_delaunay failed")
    return tri[:n].copy()


def _as_params(params):
    """accept any ctypes struct with the svh_elas_params layout"""
    if isinstance(params, ElasParams):
        return params
    return ElasParams.from_buffer_copy(bytes(params))


class Elas:
    """Drop-in for the reference class (libelas/src/elas.h:151-165)."""

    def __init__(self, params=None):
        self._p = _as_params(params) if params is not None else default_params()
        self._h = lib().svh_elas_create(C.byref(self._p))
        if not self._h:
            raise RuntimeError("svh_elas_create failed")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.svh_elas_destroy(h)

    @property
    def params(self):
        return self._p

    def _dshape(self, h, w):
        return (h // 2, w // 2) if self._p.subsampling else (h, w)

    def process(self, I1, I2, D1=None, D2=None):
        """Elas::process(I1,I2,D1,D2,dims).  Returns (status, D1, D2); on status 1
        (<3 support points) D1/D2 are left untouched, like the reference."""
        I1 = np.asarray(I1, np.uint8)
        I2 = np.asarray(I2, np.uint8)
        assert I1.shape == I2.shape and I1.ndim == 2
        h, w = I1.shape
        # dims[2] is a forward pitch >= width: anything else (reversed / broadcast / column views)
        # is copied first
        if I1.strides[1] != 1 or I2.strides != I1.strides or I1.strides[0] < w:
            I1 = np.ascontiguousarray(I1)
            I2 = np.ascontiguousarray(I2)
        dims = (C.c_int32 * 3)(w, h, I1.strides[0])
        if D1 is None:
            D1 = np.zeros(self._dshape(h, w), np.float32)
        if D2 is None:
            D2 = np.zeros(self._dshape(h, w), np.float32)
        for name, D in (("D1", D1), ("D2", D2)):
            # the library writes h*w packed float32 through the raw pointer
            if not (isinstance(D, np.ndarray) and D.dtype == np.float32 and D.flags.c_contiguous
                    and D.flags.writeable and D.shape == self._dshape(h, w)):
                raise ValueError("%s must be a writable C-contiguous float32 array of shape %s"
                                 % (name, (self._dshape(h, w),)))
        rc = lib().svh_elas_process(self._h, I1.ctypes.data, I2.ctypes.data, D1.ctypes.data,
                                    D2.ctypes.data, dims)
        if rc < 0:
            raise SvhError(rc, last_error())
        return rc, D1, D2

    def process_batch(self, I1s, I2s):
        """n independent pairs (host arrays [n,H,W]) pipelined over the engine lanes."""
        I1s = np.ascontiguousarray(I1s, np.uint8)
        I2s = np.ascontiguousarray(I2s, np.uint8)
        n, h, w = I1s.shape
        dh, dw = self._dshape(h, w)
        D1 = np.zeros((n, dh, dw), np.float32)
        D2 = np.zeros((n, dh, dw), np.float32)
        arr = C.c_void_p * n
        a1 = arr(*[I1s[i].ctypes.data for i in range(n)])
        a2 = arr(*[I2s[i].ctypes.data for i in range(n)])
        d1 = arr(*[D1[i].ctypes.data for i in range(n)])
        d2 = arr(*[D2[i].ctypes.data for i in range(n)])
        st = (C.c_int32 * n)()
        dims = (C.c_int32 * 3)(w, h, w)
        rc = lib().svh_elas_process_batch(self._h, n, a1, a2, d1, d2, dims, st)
        if rc < 0:
            raise SvhError(rc, last_error())
        return list(st), D1, D2

    def process_batch_device(self, n, dI1, dI2, in_stride, dD1, dD2, out_stride, w, h, pitch):
        """device-resident batch: raw device pointers (ints), see include/svh.h"""
        st = (C.c_int32 * n)()
        dims = (C.c_int32 * 3)(w, h, pitch)
        rc = lib().svh_elas_process_batch_device(self._h, n, dI1, dI2, in_stride, dD1, dD2,
                                                 out_stride, dims, st)
        if rc < 0:
            raise SvhError(rc, last_error())
        return list(st)

    def stream(self, w, h, pitch=None, depth=0):
        """streaming submission (svh_elas_stream_*): pairs in one at a time, results in order"""
        return ElasStream(self, w, h, pitch if pitch is not None else w, depth)

    # ---- parity taps -----------------------------------------------------
    def set_taps(self, enable=True):
        lib().svh_elas_set_taps(self._h, 1 if enable else 0)

    def stage(self, stage, dtype):
        n = C.c_size_t(0)
        lib().svh_elas_get_stage(self._h, stage, None, 0, C.byref(n))
        buf = np.empty(n.value, np.uint8)
        if n.value:
            rc = lib().svh_elas_get_stage(self._h, stage, buf.ctypes.data, n.value, C.byref(n))
            if rc < 0:
                raise SvhError(rc, last_error())
        return buf.view(dtype)

    def last_timing(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = lib().svh_elas_last_timing(self._h, names, ms, 16)
        return [(names[i].decode(), ms[i]) for i in range(n)]


def set_lanes(n):
    """batch workers per device (each double-buffered: two HIP streams + buffer sets)"""
    return lib().svh_elas_set_lanes(n)


def set_group(n):
    """pairs pushed through each kernel launch by one lane (1..32)"""
    return lib().svh_elas_set_group(n)


def set_stage(where):
    """E5-E7 (lattice filters, support list, Delaunay x2): 1 device, 0 host, -1 automatic"""
    return lib().svh_elas_set_stage(where)


def trim():
    """release the pooled lanes that are idle (device + pinned memory, streams); returns the count"""
    lib().svh_elas_trim.restype = C.c_int64
    return lib().svh_elas_trim()


def stage_stats():
    """(groups through the device stage, groups it handed back to the host path)"""
    a, b = C.c_int64(0), C.c_int64(0)
    lib().svh_elas_stage_stats(C.byref(a), C.byref(b))
    return a.value, b.value


def device_count():
    return lib().svh_device_count()

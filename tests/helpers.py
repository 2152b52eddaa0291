"""Shared test plumbing: ctypes bindings for the three libraries under test.

  * product   stereo-vision_amd/libsvhip.so      (HIP path, C-ABI include/svh.h)
  * oracle    oracle/liboracle.so                (scalar CPU restatement)
  * reference oracle/_ref/libref_{elas,viso}.so  (the real reference, compiled
              from /root/reference by oracle/Makefile; travels prebuilt)

Nothing here reads /root/reference at run time.
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "stereo-vision_amd")
if PKG not in sys.path:
    sys.path.insert(0, PKG)
GOLDEN = os.path.join(ROOT, "tests", "golden")


class ElasParams(C.Structure):
    """svh_elas_params (include/svh.h) == Elas::parameters (libelas/src/elas.h:59-148)."""
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

    def copy(self, **kw):
        q = ElasParams.from_buffer_copy(bytes(self))
        for k, v in kw.items():
            setattr(q, k, v)
        return q


def robotics(**kw):
    """Elas::parameters(ROBOTICS) -- libelas/src/elas.h:91-116."""
    p = ElasParams(0, 255, 0.85, 10, 5, 5, 5, 5, 0, 20, 0.02, 3.0, 1.0, 2.0, 1, 2, 1.0, 200, 3,
                   0, 1, 1, 0)
    return p.copy(**kw)


def middlebury(**kw):
    """Elas::parameters(MIDDLEBURY) -- libelas/src/elas.h:119-145."""
    p = ElasParams(0, 255, 0.95, 10, 5, 5, 5, 5, 1, 20, 0.02, 5.0, 1.0, 3.0, 0, 2, 1.0, 200, 5000,
                   1, 0, 0, 0)
    return p.copy(**kw)


# stage ids: enum svh_elas_stage (include/svh.h)
(DESC1, DESC2, DCAN_RAW, SUPPORT, TRI1, TRI2, PLANES1, PLANES2, GRID1, GRID2, D1_RAW, D2_RAW,
 D1_LR, D2_LR, D1_SEG, D2_SEG, D1_GAP, D2_GAP, STAGE_COUNT) = range(19)
D1_FINAL, D2_FINAL = STAGE_COUNT, STAGE_COUNT + 1

STAGE_DTYPE = {
    DESC1: np.uint8, DESC2: np.uint8, DCAN_RAW: np.int16, SUPPORT: np.int32, TRI1: np.int32,
    TRI2: np.int32, PLANES1: np.float32, PLANES2: np.float32, GRID1: np.int32, GRID2: np.int32,
}


def stage_dtype(stage):
    return STAGE_DTYPE.get(stage, np.float32)


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def dims_of(img):
    h, w = img.shape
    return (C.c_int32 * 3)(w, h, img.strides[0])


# ----------------------------------------------------------------------------- reference
_ref_elas = None


def ref_elas_path():
    return os.path.join(ROOT, "oracle", "_ref", "libref_elas.so")


def have_ref_elas():
    return os.path.exists(ref_elas_path())


def ref_elas():
    global _ref_elas
    if _ref_elas is None:
        lib = C.CDLL(ref_elas_path())
        lib.ref_init(1)
        lib.ref_elas_run.restype = C.c_void_p
        lib.ref_elas_run.argtypes = [C.POINTER(ElasParams), C.c_void_p, C.c_void_p, C.c_void_p]
        lib.ref_elas_run_free.argtypes = [C.c_void_p]
        lib.ref_elas_run_status.argtypes = [C.c_void_p]
        lib.ref_elas_run_get.restype = C.c_int64
        lib.ref_elas_run_get.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        lib.ref_triangulate.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32]
        _ref_elas = lib
    return _ref_elas


class StageRun:
    """dict-like view of all intermediates of one staged run."""

    def __init__(self, status, stages):
        self.status = status
        self.stages = stages

    def __getitem__(self, k):
        return self.stages[k]

    def __contains__(self, k):
        return k in self.stages


def ref_elas_run(params, I1, I2):
    """All intermediates of the reference's Elas::process on one pair."""
    lib = ref_elas()
    I1 = np.ascontiguousarray(I1, np.uint8)
    I2 = np.ascontiguousarray(I2, np.uint8)
    h = lib.ref_elas_run(C.byref(params), _p(I1), _p(I2), dims_of(I1))
    try:
        st = {}
        for s in list(range(STAGE_COUNT)) + [D1_FINAL, D2_FINAL]:
            n = lib.ref_elas_run_get(h, s, None, 0)
            if n <= 0:
                continue
            buf = np.empty(n, np.uint8)
            lib.ref_elas_run_get(h, s, _p(buf), n)
            st[s] = buf.view(stage_dtype(s))
        return StageRun(lib.ref_elas_run_status(h), st)
    finally:
        lib.ref_elas_run_free(h)


def ref_elas_process(params, I1, I2):
    lib = ref_elas()
    I1 = np.ascontiguousarray(I1, np.uint8)
    I2 = np.ascontiguousarray(I2, np.uint8)
    h, w = I1.shape
    if params.subsampling:
        h, w = h // 2, w // 2
    D1 = np.full((h, w), -7.0, np.float32)
    D2 = np.full((h, w), -7.0, np.float32)
    lib.ref_elas_process(C.byref(params), _p(I1), _p(I2), _p(D1), _p(D2), dims_of(I1))
    return D1, D2


def ref_triangulate(pts):
    pts = np.ascontiguousarray(pts, np.float32).reshape(-1, 2)
    cap = 2 * len(pts) + 16
    tri = np.empty((cap, 3), np.int32)
    n = ref_elas().ref_triangulate(_p(pts), len(pts), _p(tri), cap)
    return tri[:n].copy()


# ----------------------------------------------------------------------------- images
def read_pgm(path):
    with open(path, "rb") as f:
        data = f.read()
    assert data[:2] == b"P5"
    toks, pos = [], 2
    while len(toks) < 3:
        while data[pos:pos + 1].isspace():
            pos += 1
        if data[pos:pos + 1] == b"#":
            while data[pos:pos + 1] != b"\n":
                pos += 1
            continue
        e = pos
        while not data[e:e + 1].isspace():
            e += 1
        toks.append(int(data[pos:e]))
        pos = e
    pos += 1
    w, h, mx = toks
    assert mx == 255
    return np.frombuffer(data, np.uint8, w * h, pos).reshape(h, w).copy()


def write_pgm(path, img):
    img = np.ascontiguousarray(img, np.uint8)
    with open(path, "wb") as f:
        f.write(b"P5\n%d %d\n255\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())


def synth_pair(w, h, seed, dmax=64, planes=6, noise=2):
    """Seeded synthetic rectified pair: smooth random
    texture, piecewise-planar disparity, right = left warped by -d + noise."""
    rng = np.random.default_rng(seed)

    def value_noise(step):
        gh, gw = h // step + 3, w // step + 3
        g = rng.random((gh, gw)).astype(np.float32)
        ys = np.arange(h, dtype=np.float32) / step
        xs = np.arange(w, dtype=np.float32) / step
        y0 = ys.astype(np.int32)
        x0 = xs.astype(np.int32)
        fy = (ys - y0)[:, None]
        fx = (xs - x0)[None, :]
        a = g[y0][:, x0]
        b = g[y0][:, x0 + 1]
        c = g[y0 + 1][:, x0]
        d = g[y0 + 1][:, x0 + 1]
        return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy

    tex = 0.5 * value_noise(16) + 0.3 * value_noise(5) + 0.2 * value_noise(2)
    tex = (tex - tex.mean()) / (tex.std() + 1e-6)
    wide = w + dmax + 8
    # texture for a wider canvas so that the right view has content everywhere
    texw = np.concatenate([tex, tex[:, ::-1][:, :wide - w]], axis=1) if wide > w else tex
    left_wide = np.clip(128 + 40 * texw, 0, 255)
    # disparity: random slanted planes over vertical stripes/blocks
    disp = np.zeros((h, w), np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    bounds = np.sort(rng.integers(0, w, planes - 1))
    bounds = np.concatenate([[0], bounds, [w]])
    for i in range(planes):
        d0 = rng.uniform(4, dmax - 4)
        ax = rng.uniform(-0.02, 0.02)
        ay = rng.uniform(-0.02, 0.02)
        sl = slice(bounds[i], bounds[i + 1])
        disp[:, sl] = np.clip(d0 + ax * (xx[:, sl] - bounds[i]) + ay * (yy[:, sl] - h / 2), 1, dmax)
    left = left_wide[:, :w]
    # right(x) = left(x + d): sample the left canvas at nearest pixel
    xr = np.clip(np.rint(xx + disp).astype(np.int32), 0, wide - 1)
    right = np.take_along_axis(left_wide, xr, axis=1)
    n1 = rng.integers(-noise, noise + 1, (h, w))
    n2 = rng.integers(-noise, noise + 1, (h, w))
    I1 = np.clip(np.rint(left) + n1, 0, 255).astype(np.uint8)
    I2 = np.clip(np.rint(right) + n2, 0, 255).astype(np.uint8)
    return I1, I2


def golden_pair(name):
    """Committed input pair under tests/golden (PGM crops of the reference's own images)."""
    l = read_pgm(os.path.join(GOLDEN, name + "_left.pgm"))
    r = read_pgm(os.path.join(GOLDEN, name + "_right.pgm"))
    return l, r


def disparity_agreement(D, Dref):
    """Fraction of reference-valid pixels whose value is within +-1 (and valid) in D."""
    valid = Dref >= 0
    ok = valid & (D >= 0) & (np.abs(D - Dref) <= 1.0)
    return ok.sum() / max(int(valid.sum()), 1)


# ----------------------------------------------------------------------------- oracle
_oracle = None
TRI_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32)


def oracle_path():
    return os.path.join(ROOT, "oracle", "liboracle.so")


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(oracle_path())
        lib.orc_elas_run.restype = C.c_void_p
        lib.orc_elas_run.argtypes = [C.POINTER(ElasParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p]
        lib.orc_elas_run_free.argtypes = [C.c_void_p]
        lib.orc_elas_run_status.argtypes = [C.c_void_p]
        lib.orc_elas_run_get.restype = C.c_int64
        lib.orc_elas_run_get.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
        _oracle = lib
    return _oracle


def ref_triangulator():
    """address of the real Triangle (oracle/_ref) as an orc_triangulate_fn"""
    return C.cast(ref_elas().ref_triangulate, C.c_void_p)


def fixture_triangulator(tri_lists):
    """orc_triangulate_fn that replays golden triangle lists in call order."""
    it = iter(tri_lists)

    def fn(pts, n, tri, cap):
        t = np.ascontiguousarray(next(it), np.int32).reshape(-1, 3)
        C.memmove(tri, t.ctypes.data, t.nbytes)
        return len(t)

    cb = TRI_FN(fn)
    return cb


def oracle_elas_run(params, I1, I2, tri_fn=None):
    """All intermediates of the oracle restatement on one pair."""
    lib = oracle()
    I1 = np.ascontiguousarray(I1, np.uint8)
    I2 = np.ascontiguousarray(I2, np.uint8)
    if tri_fn is None:
        tri_fn = ref_triangulator()
    fn = tri_fn if isinstance(tri_fn, C.c_void_p) else C.cast(tri_fn, C.c_void_p)
    h = lib.orc_elas_run(C.byref(params), _p(I1), _p(I2), dims_of(I1), fn)
    try:
        st = {}
        for s in list(range(STAGE_COUNT)) + [D1_FINAL, D2_FINAL]:
            n = lib.orc_elas_run_get(h, s, None, 0)
            if n <= 0:
                continue
            buf = np.empty(n, np.uint8)
            lib.orc_elas_run_get(h, s, _p(buf), n)
            st[s] = buf.view(stage_dtype(s))
        return StageRun(lib.orc_elas_run_status(h), st)
    finally:
        lib.orc_elas_run_free(h)


STAGE_NAMES = {
    DESC1: "desc1", DESC2: "desc2", DCAN_RAW: "dcan_raw", SUPPORT: "support", TRI1: "tri1",
    TRI2: "tri2", PLANES1: "planes1", PLANES2: "planes2", GRID1: "grid1", GRID2: "grid2",
    D1_RAW: "d1_raw", D2_RAW: "d2_raw", D1_LR: "d1_lr", D2_LR: "d2_lr", D1_SEG: "d1_seg",
    D2_SEG: "d2_seg", D1_GAP: "d1_gap", D2_GAP: "d2_gap", D1_FINAL: "d1", D2_FINAL: "d2",
}


def compare_runs(a, b, stages=None, skip=()):
    """list of (stage name, n_mismatch) for stages present in both runs"""
    out = []
    for s in (stages or sorted(STAGE_NAMES)):
        if s in skip or s not in a or s not in b:
            continue
        x, y = a[s], b[s]
        if x.shape != y.shape:
            out.append((STAGE_NAMES[s], -1))
        else:
            out.append((STAGE_NAMES[s], int((x.view(np.uint8) != y.view(np.uint8)).sum())
                        if x.dtype != np.float32 else int((x != y).sum())))
    return out

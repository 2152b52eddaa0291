"""3-D reprojection + map fusion (include/svh_map.h) against the CPU
restatement oracle/map_oracle.cpp.  The per-pixel behaviour of the reference cannot be pinned (Qt
code, and it reads a freed map: see the oracle's header); its coefficient matrices are pinned
against the reference's own Matrix class through oracle/_ref."""
import ctypes as C

import numpy as np
import pytest

import helpers as H


class MapParams(C.Structure):
    _fields_ = [("f", C.c_float), ("cu", C.c_float), ("cv", C.c_float), ("base", C.c_float),
                ("max_dist", C.c_float)]


def pose(rx, ry, rz, tx, ty, tz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = (tx, ty, tz)
    return T


def oracle_map(oracle_lib):
    L = oracle_lib
    L.orc_map_create.restype = C.c_void_p
    L.orc_map_create.argtypes = [C.POINTER(MapParams)]
    L.orc_map_destroy.argtypes = [C.c_void_p]
    L.orc_map_clear.argtypes = [C.c_void_p]
    L.orc_map_add.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
    L.orc_map_points.restype = C.c_int64
    L.orc_map_points.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]
    L.orc_map_planes.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_map_coeffs.argtypes = [C.POINTER(MapParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


class OracleMapper:
    def __init__(self, L, prm):
        self.L, self.h = L, L.orc_map_create(C.byref(prm))

    def add(self, D1, I1, Ht, gain):
        D1 = np.ascontiguousarray(D1, np.float32)
        I1 = np.ascontiguousarray(I1, np.uint8)
        Ht = np.ascontiguousarray(Ht, np.float64)
        self.shape = I1.shape
        dims = (C.c_int32 * 3)(I1.shape[1], I1.shape[0], I1.shape[1])
        self.L.orc_map_add(self.h, D1.ctypes.data, I1.ctypes.data, dims, Ht.ctypes.data, gain)

    def points(self, which):
        n = self.L.orc_map_points(self.h, which, None, 0)
        out = np.zeros((n, 4), np.float32)
        if n:
            self.L.orc_map_points(self.h, which, out.ctypes.data, n)
        return out

    def planes(self):
        out = np.zeros((5,) + self.shape, np.float32)
        self.L.orc_map_planes(self.h, out.ctypes.data)
        return out

    def clear(self):
        self.L.orc_map_clear(self.h)

    def __del__(self):
        self.L.orc_map_destroy(self.h)


def synth_frames(w, h, n, seed, step=0.12, valid=0.8):
    """a camera gliding through a static piecewise-planar scene: per-frame disparity maps whose
    re-projections land on (and next to) each other, plus images, poses and gains"""
    rng = np.random.default_rng(seed)
    f, cu, cv, base = 0.9 * w, w / 2 - 3.5, h / 2 + 1.25, 0.54
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    frames = []
    img0 = rng.integers(0, 256, (h, w)).astype(np.uint8)
    for k in range(n):
        Ht = pose(0.002 * k, 0.01 * k, -0.003 * k, 0.02 * k, -0.01 * k, step * k)   # camera k -> frame 0
        depth = 4.0 + 3.0 * np.sin(xx / w * 3 + 0.3) + 2.0 * (yy / h) - step * k
        depth += (xx > w * 0.6) * 2.5                                              # a depth step
        d = (f * base / np.maximum(depth, 0.3)).astype(np.float32)
        d += rng.uniform(-0.2, 0.2, d.shape).astype(np.float32)
        d[rng.random(d.shape) > valid] = -1                                        # holes
        d[:, : w // 16] = -10                                                       # an invalid band
        d[0, 0] = 0.0
        d[1, 1] = 1e-3                                                              # beyond max_dist
        d[2, 2] = 4000.0                                                            # closer than 0.1
        img = np.roll(img0, 3 * k, axis=1)
        gain = [0.0, 1.07, 0.93, 1.0, 1.31][k % 5]
        frames.append((d, img, Ht, np.float32(gain)))
    return (np.float32(f), np.float32(cu), np.float32(cv), np.float32(base)), frames


def test_coefficients_match_the_reference_matrix_class(oracle_lib):
    """pin: Matrix::inv / getMat / operator* of the reference == the oracle's restatement"""
    if not H.have_ref_viso():
        pytest.skip("needs oracle/_ref")
    L = oracle_map(oracle_lib)
    R = H.ref_viso()
    R.ref_map_coeffs.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    rng = np.random.default_rng(2)
    for trial in range(200):
        Ht = pose(*rng.uniform(-0.4, 0.4, 3), *rng.uniform(-30, 30, 3))
        if trial % 7 == 0:
            Ht = np.eye(4)
        if trial % 11 == 0:
            Ht[:3, :3] *= rng.uniform(0.5, 2.0)        # not a rotation: inv() != transpose
        prm = MapParams(*(np.float32(v) for v in (rng.uniform(300, 1200), rng.uniform(200, 900), rng.uniform(80, 500), 0.54, 20)))
        a = [np.zeros(12, np.float32), np.zeros(4, np.float32), np.zeros(12, np.float32)]
        b = [np.zeros(12, np.float32), np.zeros(4, np.float32), np.zeros(12, np.float32)]
        Hc = np.ascontiguousarray(Ht, np.float64)
        L.orc_map_coeffs(C.byref(prm), Hc.ctypes.data, *[x.ctypes.data for x in a])
        R.ref_map_coeffs(Hc.ctypes.data, prm.f, prm.cu, prm.cv, *[x.ctypes.data for x in b])
        for x, y in zip(a, b):
            assert np.array_equal(x, y), trial


def test_oracle_fusion_behaves(oracle_lib):
    """sanity of the restated behaviour: a static scene seen twice from one pose mostly merges (the
    truncating re-projection moves some points onto a neighbour, where the 0.2 m test can fail),
    a moved camera leaves leftovers, clearing forgets the previous map"""
    L = oracle_map(oracle_lib)
    (f, cu, cv, base), frames = synth_frames(96, 64, 3, seed=5)
    m = OracleMapper(L, MapParams(f, cu, cv, base, 20))
    d, img, Ht, _ = frames[0]
    m.add(d, img, Ht, 0.0)
    n0 = len(m.points(1))
    assert n0 > 1000 and len(m.points(0)) == 0
    m.add(d, img, Ht, 0.0)                       # same frame again: every point re-projects onto itself
    assert len(m.points(0)) < 0.5 * n0
    assert n0 <= len(m.points(1)) < 1.2 * n0     # merged points stay, a few fill holes
    d2, img2, Ht2, g2 = frames[2]
    m.add(d2, img2, Ht2, g2)
    assert 0 < len(m.points(0)) < n0
    m.clear()
    m.add(d, img, Ht, 0.0)
    assert len(m.points(0)) == 0 and len(m.points(1)) == n0


def test_map_refuses_to_run_without_a_gpu():
    import svhip as S
    if S.device_count() > 0:
        pytest.skip("a GPU is present")
    from svhip import mapper
    with pytest.raises(S.SvhError) as ei:
        mapper.Mapper(600, 300, 100, 0.5)
    assert ei.value.code == S.ERR_NO_DEVICE


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,n,seed", [(96, 64, 4, 1), (322, 117, 5, 2), (1242, 375, 3, 3), (65, 33, 6, 4)])
def test_fusion_matches_oracle(w, h, n, seed, oracle_lib):
    """every frame of a sequence: both point lists (order included) and all five map planes are
    bit-identical to the oracle's"""
    from svhip import mapper
    L = oracle_map(oracle_lib)
    (f, cu, cv, base), frames = synth_frames(w, h, n, seed)
    o = OracleMapper(L, MapParams(f, cu, cv, base, 20))
    g = mapper.Mapper(f, cu, cv, base, 20)
    for k, (d, img, Ht, gain) in enumerate(frames):
        o.add(d, img, Ht, gain)
        g.add(d, img, Ht, gain)
        for which in (0, 1):
            a, b = o.points(which), g.points(which)
            assert a.shape == b.shape, (k, which, a.shape, b.shape)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (k, which)
        assert np.array_equal(o.planes().view(np.uint32), g.planes().view(np.uint32)), k
    assert len(o.points(0)) > 0 and len(o.points(1)) > 0


@pytest.mark.gpu
def test_many_points_on_one_pixel_and_strided_input(oracle_lib):
    """a far-away previous map collapses onto a few pixels of the current one (long per-pixel lists,
    replayed in scan order); the image rows come with a stride; the disparity map from the device"""
    from svhip import mapper
    L = oracle_map(oracle_lib)
    w, h = 160, 90
    (f, cu, cv, base), frames = synth_frames(w, h, 2, seed=9, valid=0.95)
    o = OracleMapper(L, MapParams(f, cu, cv, base, 60))
    g = mapper.Mapper(f, cu, cv, base, 60)
    d0, img0, H0, _ = frames[0]
    o.add(d0, img0, H0, 0.0)
    g.add(d0, img0, H0, 0.0)
    # second camera 25 m behind the first: the whole previous cloud lands in a small patch
    H1 = pose(0, 0, 0, 0, 0, -25.0)
    d1 = np.full((h, w), -1, np.float32)
    d1[h // 2 - 2:h // 2 + 2, w // 2 - 3:w // 2 + 3] = np.float32(f * base / 28.0)
    wide = np.zeros((h, w + 13), np.uint8)
    wide[:, :w] = frames[1][1]
    o.add(d1, wide[:, :w], H1, 1.2)
    # device copy of the map through the HIP runtime libsvhip already loaded (torch is not
    # imported here: its own HIP runtime must come first in a process)
    hip = C.CDLL("libamdhip64.so")
    dev = C.c_void_p()
    assert hip.hipMalloc(C.byref(dev), C.c_size_t(d1.nbytes)) == 0
    assert hip.hipMemcpy(dev, C.c_void_p(d1.ctypes.data), C.c_size_t(d1.nbytes), 1) == 0   # hipMemcpyHostToDevice
    g.add(None, wide[:, :w], H1, 1.2, device_ptr=dev.value)
    hip.hipFree(dev)
    for which in (0, 1):
        a, b = o.points(which), g.points(which)
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), which
    assert np.array_equal(o.planes().view(np.uint32), g.planes().view(np.uint32))
    assert len(o.points(0)) < 0.9 * np.count_nonzero(d0 > 0)      # many merged into the patch


@pytest.mark.gpu
def test_consumes_the_disparity_map_elas_left_on_the_device(oracle_lib):
    """ELAS -> map fusion without the map leaving the GPU: svh_elas_process_batch_device writes D1
    to device memory, svh_map_add reads it there; same points as with a downloaded copy"""
    import svhip as S
    from svhip import mapper
    hip = C.CDLL("libamdhip64.so")

    def dmalloc(nbytes):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(nbytes)) == 0
        return p

    l, r = H.golden_pair("urban3_640x240")
    h, w = l.shape
    n = w * h
    dI1, dI2, dD1, dD2 = dmalloc(n), dmalloc(n), dmalloc(4 * n), dmalloc(4 * n)
    assert hip.hipMemcpy(dI1, C.c_void_p(l.ctypes.data), C.c_size_t(n), 1) == 0
    assert hip.hipMemcpy(dI2, C.c_void_p(r.ctypes.data), C.c_size_t(n), 1) == 0
    e = S.Elas(H.robotics())
    st = e.process_batch_device(1, dI1.value, dI2.value, n, dD1.value, dD2.value, 4 * n, w, h, w)
    assert st == [0]
    D1 = np.zeros((h, w), np.float32)
    assert hip.hipMemcpy(C.c_void_p(D1.ctypes.data), dD1, C.c_size_t(4 * n), 2) == 0   # DeviceToHost
    assert (D1 > 0).mean() > 0.5
    a = mapper.Mapper(645.24, 321.0, 118.0, 0.5707)
    b = mapper.Mapper(645.24, 321.0, 118.0, 0.5707)
    o = OracleMapper(oracle_map(oracle_lib), MapParams(645.24, 321.0, 118.0, 0.5707, 20))
    for k in range(2):
        Ht = pose(0, 0.004 * k, 0, 0.01 * k, 0, 0.3 * k)
        a.add(None, l, Ht, 1.05, device_ptr=dD1.value)
        b.add(D1, l, Ht, 1.05)
        o.add(D1, l, Ht, 1.05)
        for which in (0, 1):
            pa, pb, po = a.points(which), b.points(which), o.points(which)
            assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
            assert np.array_equal(pa.view(np.uint32), po.view(np.uint32))
    assert len(a.points(1)) > 50000
    for p in (dI1, dI2, dD1, dD2):
        hip.hipFree(p)


@pytest.mark.gpu
def test_disparity_colormap_matches_oracle(oracle_lib):
    from svhip import mapper
    rng = np.random.default_rng(4)
    D = rng.uniform(-20, 260, (97, 131)).astype(np.float32)
    D[0, :12] = [0, -1, -10, 200, 199.99998, 200.00002, 1e-30, 33.333332, 66.666664, 100, 133.33333, 166.66667]
    want = np.zeros(D.shape + (3,), np.float32)
    oracle_lib.orc_disparity_colormap.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    oracle_lib.orc_disparity_colormap(D.ctypes.data, D.size, want.ctypes.data)
    got = mapper.disparity_colormap(D)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.all(got[D <= 0] == 0) and got[0, 3].tolist() == [1.0, 0.0, 0.0]

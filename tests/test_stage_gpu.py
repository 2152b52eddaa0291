"""GPU parity of the device-side stages between the two matching phases (k_lattice, k_delaunay,
k_stage_pack in csrc/elas_stage_kernels.hip): lattice filters + support list
(libelas/src/elas.cpp:174-318, 495-523) and the two Delaunay triangulations in Triangle's output
order (elas.cpp:534-600, triangle.cpp "zQB").

The same stage taps as tests/test_elas_gpu.py -- SUPPORT, TRI1, TRI2 and everything downstream --
with svh_elas_set_stage(1), against the oracle (whose triangulator is the real Triangle from
oracle/_ref, or the reference's golden triangle lists) and the reference's goldens; plus the batch
entry (device stage by default) against the single call (host stage), pair by pair.
"""
import os

import numpy as np
import pytest

import helpers as H
from test_elas_gpu import assert_same, oracle_for, product_run

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dev_stage():
    import svhip as S
    S.lib()
    assert S.device_count() > 0, "no HIP device: the product has no CPU fallback"
    S.set_stage(1)
    yield S
    S.set_stage(-1)


@pytest.mark.parametrize("case", ["urban3_demo", "urban1_robotics", "urban2_stereomapper", "cones_middlebury"])
def test_device_stage_on_golden(case, dev_stage, oracle_lib):
    z = np.load(os.path.join(H.GOLDEN, case + ".npz"))
    prm = H.ElasParams.from_buffer_copy(z["params"].tobytes())
    l, r = H.golden_pair(str(z["crop"]))
    before = dev_stage.stage_stats()
    got = product_run(dev_stage, prm, l, r)
    after = dev_stage.stage_stats()
    assert after[0] == before[0] + 1 and after[1] == before[1]     # device stage, not handed back
    assert got.status == 0
    # the reference's own lists first: support points, both triangulations (order included)
    for s in (H.SUPPORT, H.TRI1, H.TRI2):
        assert np.array_equal(got[s], z[H.STAGE_NAMES[s]]), H.STAGE_NAMES[s]
    assert_same(oracle_for(z, prm, l, r), got)
    assert np.array_equal(got[H.D1_FINAL], z["d1"]) and np.array_equal(got[H.D2_FINAL], z["d2"])


@pytest.mark.parametrize("case", ["urban3_kitti", "urban4_kitti"])
def test_device_stage_on_slim_golden(case, dev_stage):
    z = np.load(os.path.join(H.GOLDEN, case + ".npz"))
    prm = H.ElasParams.from_buffer_copy(z["params"].tobytes())
    l, r = H.golden_pair(str(z["crop"]))
    got = product_run(dev_stage, prm, l, r)
    assert got.status == 0
    for s in (H.SUPPORT, H.TRI1, H.TRI2):
        assert np.array_equal(got[s], z[H.STAGE_NAMES[s]]), H.STAGE_NAMES[s]
    assert np.array_equal(got[H.D1_FINAL], z["d1"]) and np.array_equal(got[H.D2_FINAL], z["d2"])


@pytest.mark.parametrize("seed,w,h,kw", [
    (21, 320, 200, {}),
    (22, 333, 117, {"postprocess_only_left": 0}),
    (23, 256, 160, {"support_texture": 30, "incon_window_size": 7, "incon_min_support": 7}),
    (24, 400, 240, {"disp_max": 63, "grid_size": 16, "candidate_stepsize": 4}),
    (25, 1242, 375, {}),
    (26, 322, 201, {"subsampling": 1}),
    (27, 640, 480, {"add_corners": 1, "incon_threshold": 2, "incon_min_support": 3}),
    (28, 200, 90, {"incon_window_size": 9, "incon_min_support": 12}),      # window > 15 cells wide
    (29, 96, 64, {}),                                                       # very few support points
])
@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref triangulator")
def test_device_stage_synthetic(seed, w, h, kw, dev_stage, oracle_lib):
    l, r = H.synth_pair(w, h, seed, dmax=min(48, kw.get("disp_max", 255) - 8, w // 4))
    prm = H.robotics(**kw)
    got = product_run(dev_stage, prm, l, r)
    want = H.oracle_elas_run(prm, l, r)
    assert got.status == want.status
    if want.status == 0:
        assert_same(want, got)


@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref triangulator")
@pytest.mark.parametrize("w,h,seed,dmax", [(1920, 600, 31, 120), (1920, 1080, 32, 230)])
def test_device_stage_large_image_takes_the_global_memory_forms(w, h, seed, dmax, dev_stage, oracle_lib):
    """1920x600 / 1920x1080 (BASELINE.json configs[3]): the lattice (384 x 120 / 384 x 216 cells) does not
    fit k_lattice's LDS and the support points exceed the LDS record capacity of k_delaunay, so the in-place
    global-memory lattice filter and the 32-bit L2 triangle records run, k_delaunay with 1024 threads and the
    whole LDS of a CU for its rank and cut-order phases (automatic mode picks the host stage for a single
    pair of this size)"""
    l, r = H.synth_pair(w, h, seed, dmax=dmax, planes=8)
    prm = H.robotics()
    before = dev_stage.stage_stats()
    got = product_run(dev_stage, prm, l, r)
    after = dev_stage.stage_stats()
    assert after[0] == before[0] + 1 and after[1] == before[1]
    want = H.oracle_elas_run(prm, l, r)
    assert got.status == want.status == 0
    assert len(want[H.SUPPORT]) // 3 > 2400          # beyond the LDS record capacity (2 339 points)
    assert_same(want, got)


@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref triangulator")
@pytest.mark.parametrize("stage", [0, 1])
@pytest.mark.parametrize("dmin", [7, 20, -5, 33])
@pytest.mark.parametrize("image", ["urban3_640x240", "synth_1242x375"])
def test_disp_min_moves_the_support_search(image, dmin, stage, oracle_lib):
    """Elas::parameters::disp_min (elas.h:61; settable through libelas/matlab/elasMex.cpp:60): the support search
    scans d = max(disp_min, 0) .. disp_max_valid and gives up below a range of 10 (elas.cpp:384-396).  The LDS
    support kernel builds its trip order and lane offsets on that lower bound, so the candidate map is compared
    first, then every later stage, in both forms of the middle stages."""
    import svhip as S
    S.lib()
    if image.startswith("synth"):
        l, r = H.synth_pair(1242, 375, 61 + dmin, dmax=96, planes=7)
    else:
        l, r = H.golden_pair(image)
    prm = H.robotics(disp_min=dmin)
    S.set_stage(stage)
    try:
        got = product_run(S, prm, l, r)
    finally:
        S.set_stage(-1)
    want = H.oracle_elas_run(prm, l, r)
    assert got.status == want.status == 0
    assert np.array_equal(got[H.DCAN_RAW], want[H.DCAN_RAW]), "candidate disparities (E3/E4)"
    base = H.oracle_elas_run(H.robotics(), l, r)
    if dmin > 0:
        assert not np.array_equal(base[H.DCAN_RAW], want[H.DCAN_RAW]), "disp_min had no effect on this input"
    else:
        assert np.array_equal(base[H.DCAN_RAW], want[H.DCAN_RAW])       # negative bounds clamp to 0
    assert_same(want, got)


@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref triangulator")
def test_disp_min_leaving_less_than_ten_disparities_fails_like_the_reference(capfd, oracle_lib):
    """disp_max_valid - disp_min_valid < 10 at every lattice point (elas.cpp:390): no support point, the reference's
    message, maps untouched"""
    import svhip as S
    l, r = H.golden_pair("urban3_640x240")
    prm = H.robotics(disp_min=250)
    want = H.oracle_elas_run(prm, l, r)
    D1 = np.full(l.shape, 7.0, np.float32)
    D2 = np.full(l.shape, 7.0, np.float32)
    rc, D1, D2 = S.Elas(prm).process(l, r, D1, D2)
    assert want.status != 0 and rc != 0
    assert "ERROR: Need at least 3 support points!" in capfd.readouterr().out
    assert (D1 == 7.0).all() and (D2 == 7.0).all()


FUZZ_SHAPES = [(320, 200), (401, 177), (512, 160), (288, 240)]


@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref")
@pytest.mark.parametrize("seed", range(300, 330))
def test_device_stage_param_fuzz(seed, dev_stage, oracle_lib):
    """every field of Elas::parameters moves; small candidate steps with a wide L/R threshold
    produce coincident right-image points (k_delaunay replays Triangle's quicksort for those)"""
    prm = H.fuzz_elas_params(seed)
    w, h = FUZZ_SHAPES[seed % 4]
    l, r = H.synth_pair(w, h, seed, dmax=min(48, prm.disp_max - 8))
    got = product_run(dev_stage, prm, l, r)
    want = H.oracle_elas_run(prm, l, r)
    assert got.status == want.status
    if want.status == 0:
        assert_same(want, got)


@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref")
def test_batch_device_stage_equals_single_host_stage(dev_stage, capfd):
    """svh_elas_process_batch (groups of pairs through the device stage) against
    svh_elas_process with the host stage, pair by pair; a flat pair in the middle of a group keeps
    the reference's error behaviour (status 1, message, outputs untouched: elas.cpp:69-75)"""
    S = dev_stage
    S.set_stage(-1)     # automatic: the batch takes the device stage, the single call the host
    w, h = 400, 240
    pairs = [H.synth_pair(w, h, 700 + i, dmax=40) for i in range(13)]
    flat = np.full((h, w), 90, np.uint8)
    pairs[5] = (flat, flat)
    prm = H.robotics()
    st, D1, D2 = S.Elas(prm).process_batch(np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs]))
    assert st == [0] * 5 + [1] + [0] * 7
    assert np.all(D1[5] == 0) and np.all(D2[5] == 0)          # process_batch hands in zeroed maps
    S.set_stage(0)
    for i, (l, r) in enumerate(pairs):
        if i == 5:
            continue
        rc, A1, A2 = S.Elas(prm).process(l, r)
        assert rc == 0
        assert np.array_equal(A1, D1[i]) and np.array_equal(A2, D2[i]), i
    out = capfd.readouterr().out
    assert out.count("Need at least 3 support points") == 1


@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref")
def test_coincident_points_stay_on_the_device(dev_stage, oracle_lib):
    """candidate_stepsize 2 with lr_threshold 3: two support points of one row may land on the same
    right-image pixel; Triangle keeps the one its randomised quicksort puts first.  k_delaunay replays
    that quicksort (same generator, same partition, same call order) on one lane, drops the repeats and
    triangulates the survivors: nothing comes back to the host, and every stage equals the oracle's
    (which uses the real Triangle)"""
    seen = 0
    before = dev_stage.stage_stats()
    for seed in range(40, 52):
        prm = H.robotics(candidate_stepsize=2, lr_threshold=3, incon_min_support=3, support_threshold=0.95)
        l, r = H.synth_pair(240, 120, seed, dmax=30, noise=6)
        want = H.oracle_elas_run(prm, l, r)
        if want.status != 0:
            continue
        sup = want[H.SUPPORT].reshape(-1, 3)
        key = (sup[:, 0] - sup[:, 2]).astype(np.int64) * 4096 + sup[:, 1]
        seen += len(np.unique(key)) < len(key)
        got = product_run(dev_stage, prm, l, r)
        assert got.status == 0
        assert_same(want, got)
    assert seen > 0, "no case with coincident right-image points was generated"
    assert dev_stage.stage_stats()[1] == before[1]     # no group was handed back to the host path


def test_device_resident_batch_keeps_failed_pairs_untouched(dev_stage, capfd):
    """svh_elas_process_batch_device (what bench.py calls): maps of a pair with fewer than 3 support
    points keep their previous contents, its status is 1, the others equal the host-buffer batch.
    (Device memory straight from the HIP runtime the library links: torch's own copy of the runtime
    must not be brought up after libsvhip, see INTEGRATION.md.)"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    S = dev_stage
    S.set_stage(-1)
    w, h = 400, 240
    pairs = [H.synth_pair(w, h, 800 + i, dmax=40) for i in range(5)]
    flat = np.full((h, w), 33, np.uint8)
    pairs[2] = (flat, flat)
    I1 = np.ascontiguousarray(np.stack([p[0] for p in pairs]))
    I2 = np.ascontiguousarray(np.stack([p[1] for p in pairs]))
    prm = H.robotics()
    st_h, H1, H2 = S.Elas(prm).process_batch(I1, I2)
    D0 = np.full((5, h, w), -7.0, np.float32)

    def to_device(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(p, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0   # H2D
        return p
    dI1, dI2, dD1, dD2 = to_device(I1), to_device(I2), to_device(D0), to_device(D0)
    try:
        st = S.Elas(prm).process_batch_device(5, dI1, dI2, w * h, dD1, dD2, w * h * 4, w, h, w)
        assert st == st_h == [0, 0, 1, 0, 0]
        D1, D2 = np.empty_like(D0), np.empty_like(D0)
        assert hip.hipMemcpy(C.c_void_p(D1.ctypes.data), dD1, C.c_size_t(D1.nbytes), 2) == 0   # D2H
        assert hip.hipMemcpy(C.c_void_p(D2.ctypes.data), dD2, C.c_size_t(D2.nbytes), 2) == 0
    finally:
        for p in (dI1, dI2, dD1, dD2):
            hip.hipFree(p)
    assert np.all(D1[2] == -7.0) and np.all(D2[2] == -7.0)
    for i in (0, 1, 3, 4):
        assert np.array_equal(D1[i], H1[i]) and np.array_equal(D2[i], H2[i])


def test_every_stage_with_descriptors_on_the_fly():
    """The parity tests read every stage through taps, and with taps on the engine keeps the full descriptor maps
    (the DESC stages are one of them).  Without taps -- the production path -- E1 only writes the two Sobel planes
    and the support and dense matchers assemble the descriptor rows they stage themselves
    (descriptors_on_the_fly).  SVH_DESC_FLY=2 selects that form under taps as well (the DESC taps then come from
    the full kernel run first): this file and test_elas_gpu.py run once more that way, every stage bit-exact."""
    import os
    import subprocess
    import sys
    if os.environ.get("SVH_DESC_FLY") == "2":
        pytest.skip("already the inner run")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, SVH_DESC_FLY="2")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(here, "test_stage_gpu.py"), os.path.join(here, "test_elas_gpu.py")],
                       env=env, capture_output=True, text=True, cwd=here, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.skipif(os.environ.get("SVH_DT_UNIFORM") is not None, reason="already inside the switched run")
def test_device_stage_with_the_scalar_seam_walk():
    """SVH_DT_UNIFORM=1 (read once per process): depths of the triangulation with no more nodes than waves run one
    merge per wave as wave-uniform (scalar) code -- the goldens, the global-memory forms of large images, the
    coincident-point corner and part of the parameter fuzz again in a process with the switch set: same triangle
    lists, order included"""
    import subprocess
    import sys
    env = dict(os.environ, SVH_DT_UNIFORM="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "golden or large_image or coincident or param_fuzz"],
                       env=env, capture_output=True, text=True, cwd=H.ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-500:]

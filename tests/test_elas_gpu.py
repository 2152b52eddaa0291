"""GPU parity: the HIP path (through the C-ABI of libsvhip.so) against the oracle.

Stage by stage (taps) and end to end, on the committed golden crops and on
seeded synthetic pairs.  Integer stages must be bit-exact; the float maps are
required to be bit-exact too (the kernels use non-contracted IEEE single ops in
the reference's order), with the +-1 / 99 % bar of BASELINE.json as the hard
floor reported alongside.
"""
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def svhip():
    import svhip as S
    S.lib()
    assert S.device_count() > 0, "no HIP device: the product has no CPU fallback"
    return S


def product_run(S, prm, l, r):
    e = S.Elas(prm)
    e.set_taps(True)
    rc, D1, D2 = e.process(l, r)
    st = {}
    if rc == 0:
        for s in range(H.STAGE_COUNT):
            st[s] = e.stage(s, H.stage_dtype(s))
        st[H.D1_FINAL] = D1.ravel()
        st[H.D2_FINAL] = D2.ravel()
    return H.StageRun(rc, st)


def oracle_for(case_npz, prm, l, r):
    if H.have_ref_elas():
        return H.oracle_elas_run(prm, l, r)
    return H.oracle_elas_run(prm, l, r, H.fixture_triangulator([case_npz["tri1"], case_npz["tri2"]]))


def assert_same(a, b):
    bad = [(n, c) for n, c in H.compare_runs(a, b) if c != 0]
    assert not bad, bad


GOLD = ["urban3_demo", "urban1_robotics", "urban2_stereomapper", "cones_middlebury"]


@pytest.mark.parametrize("case", GOLD)
def test_stages_match_oracle_on_golden(case, svhip, oracle_lib):
    z = np.load(os.path.join(H.GOLDEN, case + ".npz"))
    prm = H.ElasParams.from_buffer_copy(z["params"].tobytes())
    l, r = H.golden_pair(str(z["crop"]))
    got = product_run(svhip, prm, l, r)
    assert got.status == 0
    want = oracle_for(z, prm, l, r)
    # headline bar first (BASELINE.json): +-1 level at >= 99 % of valid pixels
    for s, key in ((H.D1_FINAL, "d1"), (H.D2_FINAL, "d2")):
        ref = z[key]
        assert H.disparity_agreement(got[s], ref) >= 0.99
    assert_same(want, got)
    # and against the reference's own output, bit for bit
    assert np.array_equal(got[H.D1_FINAL], z["d1"])
    assert np.array_equal(got[H.D2_FINAL], z["d2"])


@pytest.mark.parametrize("case", ["urban3_kitti", "urban4_kitti"])
def test_final_maps_match_slim_golden(case, svhip):
    """the bench headline's other two crops: support list, both triangulations and the final
    maps against the reference's"""
    z = np.load(os.path.join(H.GOLDEN, case + ".npz"))
    prm = H.ElasParams.from_buffer_copy(z["params"].tobytes())
    l, r = H.golden_pair(str(z["crop"]))
    got = product_run(svhip, prm, l, r)
    assert got.status == 0
    for s in (H.SUPPORT, H.TRI1, H.TRI2):
        assert np.array_equal(got[s], z[H.STAGE_NAMES[s]]), H.STAGE_NAMES[s]
    assert np.array_equal(got[H.D1_FINAL], z["d1"])
    assert np.array_equal(got[H.D2_FINAL], z["d2"])


@pytest.mark.parametrize("seed,w,h,kw", [
    (11, 320, 200, {}),
    (12, 333, 117, {"postprocess_only_left": 0}),        # ragged width
    (13, 256, 160, {"support_texture": 30, "ipol_gap_width": 7}),
    (14, 400, 240, {"disp_max": 63, "grid_size": 16, "candidate_stepsize": 4}),
    (15, 1242, 375, {}),                                  # BASELINE config size
    (16, 322, 201, {"subsampling": 1}),                   # half-resolution output (elas.h:83-85)
    (17, 1242, 375, {"subsampling": 1, "postprocess_only_left": 0}),
    (18, 400, 240, {"subsampling": 1, "candidate_stepsize": 4, "support_texture": 30}),
])
@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref triangulator")
def test_stages_match_oracle_synthetic(seed, w, h, kw, svhip, oracle_lib):
    l, r = H.synth_pair(w, h, seed, dmax=min(48, kw.get("disp_max", 255) - 8))
    prm = H.robotics(**kw)
    got = product_run(svhip, prm, l, r)
    want = H.oracle_elas_run(prm, l, r)
    assert got.status == want.status == 0
    assert_same(want, got)


@pytest.mark.parametrize("w,h", [(1636, 48), (1920, 40), (4096, 36)])
@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref triangulator")
def test_row_widths_at_the_lds_limits(w, h, svhip, oracle_lib):
    """launch_match picks its kernel by the LDS a row needs: 40 B per pixel cross 64 KB (the opt-in
    threshold, static s_P included) at W = 1633..1638, 1920 takes the opt-in, and 16 B per pixel of
    the ordered kernel reach 64 KB at W = 4096 (global-memory fallback)"""
    l, r = H.synth_pair(w, h, 77, dmax=40)
    prm = H.robotics()
    got = product_run(svhip, prm, l, r)
    want = H.oracle_elas_run(prm, l, r)
    assert got.status == want.status
    if want.status == 0:
        assert_same(want, got)


@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref")
def test_matches_reference_process(svhip):
    """the real reference's Elas::process on the same input"""
    l, r = H.golden_pair("urban3_640x240")
    prm = H.robotics()
    D1r, D2r = H.ref_elas_process(prm, l, r)
    rc, D1, D2 = svhip.Elas(prm).process(l, r)
    assert rc == 0
    assert H.disparity_agreement(D1, D1r) >= 0.99 and H.disparity_agreement(D2, D2r) >= 0.99
    assert np.array_equal(D1, D1r) and np.array_equal(D2, D2r)


def test_strided_input_and_few_support(svhip, capfd):
    l, r = H.golden_pair("urban3_640x240")
    prm = H.robotics()
    rc0, A1, A2 = svhip.Elas(prm).process(l, r)
    # dims[2] != width: rows embedded in a wider buffer (elas.cpp:44-56)
    big_l = np.zeros((l.shape[0], l.shape[1] + 37), np.uint8)
    big_r = np.zeros_like(big_l)
    big_l[:, :l.shape[1]] = l
    big_r[:, :r.shape[1]] = r
    rc1, B1, B2 = svhip.Elas(prm).process(big_l[:, :l.shape[1]], big_r[:, :r.shape[1]])
    assert rc0 == rc1 == 0 and np.array_equal(A1, B1) and np.array_equal(A2, B2)
    # flat image: <3 support points -> message, outputs untouched (elas.cpp:69-75)
    flat = np.full((64, 96), 77, np.uint8)
    D1 = np.full((64, 96), -7.0, np.float32)
    D2 = D1.copy()
    rc, D1, D2 = svhip.Elas(prm).process(flat, flat, D1, D2)
    assert rc == 1 and np.all(D1 == -7.0) and np.all(D2 == -7.0)
    assert "Need at least 3 support points" in capfd.readouterr().out


def test_batch_equals_single(svhip):
    """pairs shard over lanes with no cross-talk: batch == one at a time"""
    names = ["urban3_640x240"]
    l, r = H.golden_pair(names[0])
    n = 6
    I1 = np.stack([np.roll(l, 3 * i, axis=1) for i in range(n)])
    I2 = np.stack([np.roll(r, 3 * i, axis=1) for i in range(n)])
    prm = H.robotics()
    e = svhip.Elas(prm)
    st, D1, D2 = e.process_batch(I1, I2)
    assert all(s == 0 for s in st)
    for i in range(n):
        rc, a, b = svhip.Elas(prm).process(I1[i], I2[i])
        assert rc == 0 and np.array_equal(a, D1[i]) and np.array_equal(b, D2[i])


@pytest.mark.parametrize("mode,crop,case", [("demo", "urban3_640x240", "urban3_demo"),
                                            ("mapper", "urban2_1242x375", "urban2_stereomapper")])
def test_cxx_dropin_call_sites(mode, crop, case, svhip, tmp_path):
    """the reference's own call sites (main.cpp:61-64, stereothread.cpp:76-114) compiled
    against include/elas.h reproduce the reference's output bit for bit"""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(H.ROOT, "tests", "cxx"), "all"],
                          stdout=subprocess.DEVNULL)
    exe = os.path.join(H.ROOT, "tests", "cxx", "elas_dropin")
    o1, o2 = str(tmp_path / "d1.f32"), str(tmp_path / "d2.f32")
    subprocess.check_call([exe, os.path.join(H.GOLDEN, crop + "_left.pgm"),
                           os.path.join(H.GOLDEN, crop + "_right.pgm"), mode, o1, o2])
    z = np.load(os.path.join(H.GOLDEN, case + ".npz"))
    assert np.array_equal(np.fromfile(o1, np.float32), z["d1"])
    assert np.array_equal(np.fromfile(o2, np.float32), z["d2"])


@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref triangulator")
def test_full_hd_pairs_batch(svhip, oracle_lib):
    """BASELINE.json configs[3]: synthetic 1920x1080, disp_max=255, a batch sharded over
    lanes; two of the pairs are checked bit-exactly against the oracle, all of them through
    size-independent properties (determinism, L/R consistency of the outputs)."""
    n = 6
    pairs = [H.synth_pair(1920, 1080, 900 + i, dmax=200, planes=8) for i in range(n)]
    I1 = np.stack([p[0] for p in pairs])
    I2 = np.stack([p[1] for p in pairs])
    prm = H.robotics(postprocess_only_left=0)
    e = svhip.Elas(prm)
    st, D1, D2 = e.process_batch(I1, I2)
    assert all(s == 0 for s in st)
    st2, E1, E2 = e.process_batch(I1, I2)
    assert np.array_equal(D1, E1) and np.array_equal(D2, E2)          # deterministic
    for i in (0, n - 1):
        want = H.oracle_elas_run(prm, I1[i], I2[i])
        assert np.array_equal(D1[i].ravel(), want[H.D1_FINAL])
        assert np.array_equal(D2[i].ravel(), want[H.D2_FINAL])
    for i in range(n):
        d = D1[i]
        assert (d >= 0).mean() > 0.5 and d.max() <= 255
        # every valid value is a valid disparity or the invalid marker, nothing else
        assert np.all((d >= 0) | (d == -10))


@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref triangulator")
@pytest.mark.parametrize("seed,w,h,kw", [
    (31, 1242, 200, {"disp_max": 800}),      # support strips exceed LDS -> global-memory variant
    (32, 4200, 96, {"disp_max": 127}),       # row wider than the LDS row cache -> global-memory matcher
    (33, 64, 48, {"disp_max": 40}),          # tiny image: a single block per kernel, few candidates
])
def test_fallback_kernels_and_extreme_shapes(seed, w, h, kw, svhip, oracle_lib):
    l, r = H.synth_pair(w, h, seed, dmax=min(60, kw["disp_max"] - 8, w // 3))
    prm = H.robotics(**kw)
    got = product_run(svhip, prm, l, r)
    want = H.oracle_elas_run(prm, l, r)
    assert got.status == want.status
    if got.status == 0:
        assert_same(want, got)


def test_concurrent_objects_from_threads(svhip):
    """two caller threads, each with its own Elas per frame (stereothread / VO thread overlap,
    maindialog.cpp:456-465, 514-518): results identical to the serial ones"""
    import threading
    pairs = [H.golden_pair("urban3_640x240"), H.golden_pair("urban1_1242x375")]
    prm = H.robotics()
    serial = [svhip.Elas(prm).process(l, r) for l, r in pairs]
    out = {}

    def work(k):
        res = []
        for _ in range(6):
            l, r = pairs[k]
            res.append(svhip.Elas(prm).process(l, r))
        out[k] = res

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for k in range(2):
        for rc, D1, D2 in out[k]:
            assert rc == 0 and np.array_equal(D1, serial[k][1]) and np.array_equal(D2, serial[k][2])


def test_streamed_sequence_of_430_frames(svhip):
    """BASELINE.json configs[2] stand-in (the KITTI drive_0029 images are not in the reference):
    430 KITTI-sized frames, the two golden 1242x375 crops cycled, streamed from host memory through
    the lanes; every frame must reproduce the reference's output for its pair."""
    z = {c: np.load(os.path.join(H.GOLDEN, c + ".npz")) for c in ("urban1_robotics",)}
    l1, r1 = H.golden_pair("urban1_1242x375")
    l2, r2 = H.golden_pair("urban2_1242x375")
    prm = H.robotics()
    rc, A1, A2 = svhip.Elas(prm).process(l2, r2)          # expected output of the second pair
    assert rc == 0
    n = 430
    I1 = np.stack([l1 if i % 2 == 0 else l2 for i in range(n)])
    I2 = np.stack([r1 if i % 2 == 0 else r2 for i in range(n)])
    st, D1, D2 = svhip.Elas(prm).process_batch(I1, I2)
    assert all(s == 0 for s in st)
    g1 = z["urban1_robotics"]["d1"].reshape(375, 1242)
    g2 = z["urban1_robotics"]["d2"].reshape(375, 1242)
    for i in range(n):
        if i % 2 == 0:
            assert np.array_equal(D1[i], g1) and np.array_equal(D2[i], g2), i
        else:
            assert np.array_equal(D1[i], A1) and np.array_equal(D2[i], A2), i

@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref triangulator")
@pytest.mark.parametrize("seed", range(40, 56))
def test_param_fuzz_matches_oracle(seed, svhip, oracle_lib):
    """random points of Elas::parameters (every field moves, see helpers.fuzz_elas_params):
    all stages and the final maps bit-exact against the oracle"""
    prm = H.fuzz_elas_params(seed)
    w, h = [(320, 200), (401, 177), (512, 160), (288, 240)][seed % 4]
    l, r = H.synth_pair(w, h, seed, dmax=min(48, prm.disp_max - 8))
    got = product_run(svhip, prm, l, r)
    want = H.oracle_elas_run(prm, l, r)
    assert got.status == want.status
    if got.status == 0:
        assert_same(want, got)

@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref triangulator")
@pytest.mark.parametrize("gap,corners,w,h", [(17, 0, 320, 200), (40, 1, 401, 177), (5000, 1, 641, 120), (5000, 0, 288, 240),
                                             (300, 1, 1000, 67), (5000, 1, 97, 61), (5000, 1, 3100, 40)])
def test_wide_gap_interpolation_matches_oracle(gap, corners, w, h, svhip, oracle_lib):
    """ipol_gap_width beyond the local kernels' reach (MIDDLEBURY: 5000) with and without the corner extrapolation:
    the per-row ballot scan and the segmented column pass (k_gap_rows_scan, k_gap_cols_seg) against the oracle's
    sequential walk (elas.cpp:1330-1530), rows that are not a multiple of 64, columns shorter than 8 segments' worth; rows beyond 3072 px take the one-thread-per-line
    form (k_gap_lines)"""
    prm = H.robotics(ipol_gap_width=gap, add_corners=corners, speckle_size=60, lr_threshold=1)
    l, r = H.synth_pair(w, h, 900 + gap % 97 + w, dmax=40, noise=4)
    got = product_run(svhip, prm, l, r)
    want = H.oracle_elas_run(prm, l, r)
    assert got.status == want.status == 0
    assert_same(want, got)

@pytest.mark.parametrize("group,workers", [(1, 1), (2, 3), (4, 2), (8, 8)])
def test_batch_with_failing_pairs(group, workers, svhip, capfd):
    """a batch mixing good pairs with flat ones (< 3 support points): statuses per pair, failed
    pairs' outputs untouched, good pairs identical to single calls -- for several group sizes and
    worker counts of the double-buffered engine (the default is restored afterwards)"""
    l, r = H.golden_pair("urban3_640x240")
    flat = np.full_like(l, 90)
    n = 11
    bad = {2, 3, 9}
    I1 = np.stack([flat if i in bad else np.roll(l, 2 * i, axis=1) for i in range(n)])
    I2 = np.stack([flat if i in bad else np.roll(r, 2 * i, axis=1) for i in range(n)])
    prm = H.robotics()
    svhip.set_group(group)
    svhip.set_lanes(workers)
    try:
        st, D1, D2 = svhip.Elas(prm).process_batch(I1, I2)
    finally:
        svhip.set_group(4)
        svhip.set_lanes(8)
    assert st == [1 if i in bad else 0 for i in range(n)]
    assert capfd.readouterr().out.count("Need at least 3 support points") == len(bad)
    for i in range(n):
        if i in bad:
            assert np.all(D1[i] == 0) and np.all(D2[i] == 0)      # process_batch hands in zeroed maps
        else:
            rc, a, b = svhip.Elas(prm).process(I1[i], I2[i])
            assert rc == 0 and np.array_equal(a, D1[i]) and np.array_equal(b, D2[i])

@pytest.mark.timeout(180)
def test_concurrent_batches_share_the_lane_pool(svhip):
    """three threads call process_batch at once while the pool only holds two workers' lanes:
    lane pairs are taken atomically (no worker can hold one lane and wait for a second), so the
    calls queue up instead of deadlocking, and every result equals the single-call result"""
    import threading
    l, r = H.golden_pair("urban3_640x240")
    prm = H.robotics()
    n = 6
    want = {}
    for k in range(3):
        for i in range(n):
            want[(k, i)] = svhip.Elas(prm).process(np.roll(l, 5 * k + i, axis=1), np.roll(r, 5 * k + i, axis=1))
    out = {}
    svhip.set_group(2)
    svhip.set_lanes(2)

    def work(k):
        I1 = np.stack([np.roll(l, 5 * k + i, axis=1) for i in range(n)])
        I2 = np.stack([np.roll(r, 5 * k + i, axis=1) for i in range(n)])
        out[k] = svhip.Elas(prm).process_batch(I1, I2)

    try:
        th = [threading.Thread(target=work, args=(k,)) for k in range(3)]
        for x in th:
            x.start()
        for x in th:
            x.join()
    finally:
        svhip.set_group(4)
        svhip.set_lanes(8)
    for k in range(3):
        st, D1, D2 = out[k]
        assert all(s == 0 for s in st)
        for i in range(n):
            rc, a, b = want[(k, i)]
            assert rc == 0 and np.array_equal(a, D1[i]) and np.array_equal(b, D2[i])

@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref triangulator")
@pytest.mark.parametrize("seed,w,h,kw", [
    (61, 320, 200, {}), (62, 333, 117, {"postprocess_only_left": 0}),
    (63, 401, 163, {"ipol_gap_width": 4, "postprocess_only_left": 0}),
    (64, 322, 201, {"subsampling": 1}), (65, 259, 131, {"subsampling": 1, "ipol_gap_width": 6}),
    (66, 1242, 375, {"ipol_gap_width": 1}), (67, 96, 64, {"disp_max": 40}),
    (68, 641, 97, {"speckle_size": 50, "lr_threshold": 1}),
])
def test_tile_post_kernels_match_oracle(seed, w, h, kw, svhip, oracle_lib):
    """without taps the default configuration runs gap interpolation + adaptive mean as two
    fused tile kernels (k_gap_tile, k_mean_tile); the final maps must equal the oracle's, and the
    unfused kernels' (taps on) bit for bit -- ragged sizes, both resolutions, gap widths 1..4"""
    prm = H.robotics(**kw)
    l, r = H.synth_pair(w, h, seed, dmax=min(48, prm.disp_max - 8))
    rc, D1, D2 = svhip.Elas(prm).process(l, r)             # tile kernels
    want = H.oracle_elas_run(prm, l, r)
    assert rc == want.status == 0
    assert np.array_equal(D1.ravel(), np.asarray(want[H.D1_FINAL]).ravel())
    assert np.array_equal(D2.ravel(), np.asarray(want[H.D2_FINAL]).ravel())
    got = product_run(svhip, prm, l, r)                    # unfused kernels (taps)
    assert np.array_equal(D1.ravel(), np.asarray(got[H.D1_FINAL]).ravel())
    assert np.array_equal(D2.ravel(), np.asarray(got[H.D2_FINAL]).ravel())

@pytest.mark.skipif(not H.have_ref_elas(), reason="needs oracle/_ref triangulator")
def test_owner_base_wraparound(svhip, oracle_lib, monkeypatch):
    """the triangle-ownership map is never cleared: every group stores owner_base + 1 + index
    with a base above all earlier values, and re-clears only when int32 would overflow.  A lane
    sized for a new geometry starts its base from SVH_TEST_OWNER_HI; 20 000 below the limit the
    base crosses it after a few calls -- every call must still give the oracle's maps."""
    monkeypatch.setenv("SVH_TEST_OWNER_HI", str(2 ** 31 - 1 - 20000))
    l, r = H.synth_pair(523, 211, 77, dmax=40)          # a geometry no other test uses
    prm = H.robotics()
    want = H.oracle_elas_run(prm, l, r)
    e = svhip.Elas(prm)
    for k in range(12):                                  # ~3 000 triangles per call
        rc, D1, D2 = e.process(l, r)
        assert rc == want.status == 0
        assert np.array_equal(D1.ravel(), np.asarray(want[H.D1_FINAL]).ravel()), k
        assert np.array_equal(D2.ravel(), np.asarray(want[H.D2_FINAL]).ravel()), k


@pytest.mark.parametrize("w,h,sub", [(64, 48, 0), (59, 27, 0), (58, 26, 0), (117, 53, 0), (61, 33, 1),
                                      (175, 29, 0), (1242, 375, 1), (1920, 1080, 0)])
def test_descriptor_strips_and_segments(w, h, sub, svhip, oracle_lib):
    """the streaming descriptor kernel at widths / heights around its strip (58 columns) and
    segment (26 rows) sizes, full and half resolution: every byte == oracle, borders zero"""
    import ctypes as C
    rng = np.random.default_rng(w * 7 + h)
    l = rng.integers(0, 256, (h, w), dtype=np.uint8)
    r = np.ascontiguousarray(l[:, ::-1])
    prm = H.robotics(subsampling=sub, disp_max=min(255, max(8, w // 3)))
    e = svhip.Elas(prm)
    e.set_taps(True)
    e.process(l, r)          # may end with "few support points": the descriptor taps are taken before that
    for img, stage in ((l, H.DESC1), (r, H.DESC2)):
        want = np.zeros((h, w, 16), np.uint8)
        oracle_lib.orc_descriptor(H._p(img), w, h, w, sub, H._p(want))
        got = e.stage(stage, np.uint8).reshape(h, w, 16)
        assert np.array_equal(got, want), (w, h, sub, np.argwhere(got != want)[:5])


def test_owner_fix_pass_at_span_ends_equals_the_exhaustive_pass(svhip, monkeypatch):
    """k_owner<true> re-checks only the two first and two last rows of every column span of a
    triangle (contested pixels can only be there, see the kernel); SVH_OWNER_FIX_ALL=1 re-checks
    every pixel.  Raw and final maps of both forms on the goldens' inputs and a set of synthetic
    pairs (thin slanted planes make many small triangles) must be identical."""
    cases = [(H.golden_pair(n), H.robotics()) for n in
             ("urban1_1242x375", "urban2_1242x375", "urban3_1242x375", "urban4_1242x375")]
    cases.append((H.golden_pair("cones_640x480"), H.middlebury()))
    for seed in range(60, 72):
        w, h = [(320, 200), (401, 177), (512, 160), (640, 480)][seed % 4]
        cases.append((H.synth_pair(w, h, seed, dmax=48, planes=12), H.robotics(subsampling=seed % 5 == 0)))
    # 1920x1080 with large disparities: the longest edge lines and the largest |a*u + b| cancellation the
    # span-ends argument has to survive (right-image triangles reach u = -disp_max)
    for seed in (72, 73):
        cases.append((H.synth_pair(1920, 1080, seed, dmax=230, planes=14), H.robotics()))
    for (l, r), prm in cases:
        outs = []
        for mode in ("0", "1"):
            monkeypatch.setenv("SVH_OWNER_FIX_ALL", mode)
            e = svhip.Elas(prm)
            e.set_taps(True)
            rc, D1, D2 = e.process(l, r)
            assert rc == 0
            outs.append((e.stage(H.D1_RAW, np.float32).copy(), e.stage(H.D2_RAW, np.float32).copy(), D1, D2))
        for a, b in zip(*outs):
            assert np.array_equal(a, b)

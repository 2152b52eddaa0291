"""GPU parity of the Matcher path (svh_matcher_* C-ABI, HIP kernels) against the
oracle and the golden quad: filter images, feature tables (order included),
every stage of matchFeatures, final match INDICES and coordinates -- bit-exact
(BASELINE.json: "libviso2 match indices are bit-exact")."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

CASES = ["viso_quad_default", "viso_quad_predicted", "viso_stereo_default", "viso_flow_default"]


def quad():
    return {k: H.read_pgm(os.path.join(H.GOLDEN, "viso_%s.pgm" % k)) for k in ("I1p", "I2p", "I1c", "I2c")}


def push_quad(m, im):
    m.push_back(im["I1p"], im["I2p"])
    m.push_back(im["I1c"], im["I2c"])


@pytest.mark.parametrize("case", CASES)
def test_matches_golden_reference_output(case):
    z = np.load(os.path.join(H.GOLDEN, case + ".npz"))
    prm = H.MatcherParams.from_buffer_copy(z["params"].tobytes())
    method = int(z["method"])
    tr = z["tr"] if z["tr"].size else None
    m = H.ProductMatcher(prm)
    push_quad(m, quad())
    assert m.match(method, tr) == 0
    for tb in range(8):
        assert np.array_equal(m.features(tb), z["table_" + H.M_TABLES[tb]]), H.M_TABLES[tb]
    for s in range(H.M_STAGE_COUNT):
        a, b = m.stage(s), z[H.M_STAGE_NAMES[s]]
        if s == H.M_RANGES:
            ns = 4 if method == 2 else 2
            a, b = a.reshape(-1, 4, 4)[:, :, :ns], b.reshape(-1, 4, 4)[:, :, :ns]
        assert a.shape == b.shape and (a == b).all(), H.M_STAGE_NAMES[s]
    got = m.matches()
    want = z["dense"]
    assert len(got) == len(want)
    for f in ("i1p", "i2p", "i1c", "i2c"):
        assert np.array_equal(got[f], want[f]), f      # the headline: match indices bit-exact
    assert (got == want).all()


@pytest.mark.parametrize("case", ["viso_quad_default", "viso_quad_predicted"])
def test_latency_path_without_taps_matches_golden(case):
    """the form an application runs: no stage taps, so from the second frame on the dense vote's triangulation
    starts on the unrefined match list while the device refines it, on warm helper threads (fork depth 3), and the
    vote itself is split four ways.  Frames alternate so that every call is a fresh quad; the list must be the
    reference's every time, and equal to the tapped run's"""
    z = np.load(os.path.join(H.GOLDEN, case + ".npz"))
    prm = H.MatcherParams.from_buffer_copy(z["params"].tobytes())
    method = int(z["method"])
    tr = z["tr"] if z["tr"].size else None
    im = quad()
    m = H.ProductMatcher(prm)
    m.lib.svh_matcher_set_taps(C.c_void_p(m.h), 0)
    want = z["dense"]
    st0, st1 = (C.c_int64 * 4)(), (C.c_int64 * 4)()
    m.lib.svh_host_helper_stats(st0)
    for rep in range(4):
        m.push_back(im["I1p"], im["I2p"])
        m.push_back(im["I1c"], im["I2c"])
        assert m.match(method, tr) == 0
        got = m.matches()
        assert len(got) == len(want) and (got == want).all(), rep
    m.lib.svh_host_helper_stats(st1)
    if len(os.sched_getaffinity(0)) >= 3:
        # (from the second call on the vote is expected to be large: its halves and its four parts went to helpers)
        assert st1[0] > st0[0], "the dense votes ran without a helper thread"


@pytest.mark.parametrize("kw,method", [
    ({}, 2), ({"half_resolution": 0}, 2), ({"multi_stage": 0}, 2), ({"refinement": 0}, 2),
    ({"nms_n": 5, "nms_tau": 30, "match_binsize": 40}, 2), ({"half_resolution": 0}, 0),
    ({"match_radius": 120, "outlier_flow_tolerance": 3}, 1),
    ({"refinement": 2}, 2), ({"refinement": 2, "half_resolution": 0}, 1), ({"refinement": 2}, 0),
])
def test_matches_oracle_on_ragged_crop(kw, method, oracle_lib):
    """non-default parameters, width 1001 (bpl 1008); oracle = CPU restatement"""
    if not H.have_ref_viso():
        pytest.skip("oracle needs the real Triangle (oracle/_ref) for removeOutliers")
    im = {k: v[20:320, 100:1101] for k, v in quad().items()}
    prm = H.matcher_defaults(**kw)
    a, b = H.OracleMatcher(prm), H.ProductMatcher(prm)
    for m in (a, b):
        push_quad(m, im)
        assert m.match(method) == 0
    bad = [x for x in H.compare_matchers(a, b, method) if x[1] != 0]
    assert not bad, bad
    assert len(b.stage(H.M_DENSE)) > 50
    for w in range(6):
        if w in (2, 3) and not prm.half_resolution:
            continue
        x, dx = a.filter_image(w)
        y, dy = b.filter_image(w)
        mg = 2
        assert dx == dy and np.array_equal(x[mg:-mg, mg:-mg], y[mg:-mg, mg:-mg]), w


def test_ring_buffer_bucketing_gain_and_empty(oracle_lib):
    if not H.have_ref_viso():
        pytest.skip("needs oracle/_ref")
    im = quad()
    prm = H.matcher_defaults()
    a, b = H.OracleMatcher(prm), H.ProductMatcher(prm)
    libc = C.CDLL(None)
    for m in (a, b):
        assert m.match(2) == 0 and len(m.matches()) == 0        # nothing pushed: silent return
        m.push_back(im["I1p"], im["I2p"])
        assert m.match(2) == 0 and len(m.matches()) == 0        # one frame only: silent return
        m.push_back(im["I1p"], im["I2p"], replace=True)
        m.push_back(im["I1c"], im["I2c"])
        m.match(2)
        libc.srand(0)
        m.nb = m.bucket(2, 50.0, 50.0)
        m.after = m.matches()
        m.g = m.gain(np.arange(0, 100, 3))
    assert a.nb == b.nb and (a.after == b.after).all() and a.g == b.g
    # a third frame: previous <- current
    for m in (a, b):
        m.push_back(im["I1p"], im["I2p"])
        m.match(2)
    assert (a.matches() == b.matches()).all() and len(a.matches()) > 1000


def test_bad_dims_message(capfd):
    m = H.ProductMatcher(H.matcher_defaults())
    I = np.zeros((10, 10), np.uint8)
    dims = (C.c_int32 * 3)(0, 10, 10)
    rc = m.lib.svh_matcher_push_back(m.h, H._p(I), H._p(I), dims, 0)
    assert rc != 0
    assert "Image dimension mismatch" in capfd.readouterr().err


@pytest.mark.parametrize("predict", [False, True])
def test_cxx_dropin_visual_odometry_call_sequence(predict, tmp_path, oracle_lib):
    """viso_stereo.cpp:41-68 replayed through include/matcher.h reproduces the oracle"""
    import subprocess
    if not H.have_ref_viso():
        pytest.skip("needs oracle/_ref")
    subprocess.check_call(["make", "-C", os.path.join(H.ROOT, "tests", "cxx"), "all"],
                          stdout=subprocess.DEVNULL)
    exe = os.path.join(H.ROOT, "tests", "cxx", "matcher_dropin")
    out = str(tmp_path / "m.bin")
    args = [exe] + [os.path.join(H.GOLDEN, "viso_%s.pgm" % k) for k in ("I1p", "I2p", "I1c", "I2c")] + [out]
    if predict:
        args.append("predict")
    txt = subprocess.check_output(args).decode()
    got = np.fromfile(out, H.P_MATCH)
    prm = H.matcher_defaults()
    a = H.OracleMatcher(prm)
    a.set_intrinsics(645.24, 635.96, 194.13, 0.5707)
    im = quad()
    push_quad(a, im)
    tr = np.eye(4)
    tr[2, 3] = -0.75
    a.match(2, tr if predict else None)
    want = a.matches()
    assert len(got) == len(want) and (got == want).all()
    C.CDLL(None).srand(0)
    nb = a.bucket(2, 50.0, 50.0)
    g = a.gain(np.arange(0, nb, 3))
    assert "matches %d" % nb in txt
    assert "gain %.6f" % g in txt

@pytest.mark.parametrize("seed", range(100, 116))
def test_param_fuzz_matches_oracle(seed, oracle_lib):
    """random points of Matcher::parameters x method x crop x predicted motion
    (helpers.fuzz_matcher_case): tables, every stage and the match indices bit-exact"""
    if not H.have_ref_viso():
        pytest.skip("oracle needs the real Triangle (oracle/_ref) for removeOutliers")
    prm, method, crop, tr = H.fuzz_matcher_case(seed)
    im = {k: v[crop] for k, v in quad().items()}
    a, b = H.OracleMatcher(prm), H.ProductMatcher(prm)
    for m in (a, b):
        push_quad(m, im)
        assert m.match(method, tr) == 0
    bad = [x for x in H.compare_matchers(a, b, method) if x[1] != 0]
    assert not bad, bad

"""Pin the VisualOdometryStereo restatement (oracle/viso_oracle.cpp, orc_vo_*) to the
reference (oracle/_ref, viso_stereo.cpp + viso.cpp compiled from /root/reference): bucketed
matches, inlier indices, delta motion and gain, bit for bit -- the same libm on the same host,
so the fp64 RANSAC + Gauss-Newton must agree exactly -- and to the committed golden."""
import os

import numpy as np
import pytest

import helpers as H


def quad():
    return [H.read_pgm(os.path.join(H.GOLDEN, "viso_%s.pgm" % k)) for k in ("I1p", "I2p", "I1c", "I2c")]


def run(vo, im):
    r = (vo.process(im[0], im[1]), vo.process(im[2], im[3]))
    return r, vo.matches(), vo.inliers(), vo.motion(), vo.gain(vo.inliers())


@pytest.mark.skipif(not H.have_ref_viso(), reason="needs oracle/_ref (real Triangle for removeOutliers)")
def test_oracle_matches_golden(oracle_lib):
    z = np.load(os.path.join(H.GOLDEN, "vo_quad.npz"))
    prm = H.VoParams.from_buffer_copy(z["params"].tobytes())
    r, m, inl, T, gain = run(H.OracleVo(prm), quad())
    assert list(r) == list(z["ok"])
    assert m.tobytes() == z["matches"].tobytes()
    assert np.array_equal(inl, z["inliers"])
    assert np.array_equal(T, z["motion"])          # bit-exact doubles
    assert np.float32(gain) == z["gain"]
    # known answers (SURVEY 8c style): 349 bucketed matches, 277 inliers, ~0.26 m forward
    assert len(m) == 349 and len(inl) == 277 and abs(T[2, 3] + 0.2568) < 1e-3
    vo = H.OracleVo(prm)
    ok, tr = vo.estimate_motion(z["syn_matches"])
    assert ok == int(z["syn_ok"]) and np.array_equal(tr, z["syn_tr"])
    assert np.array_equal(vo.inliers(), z["syn_inliers"])


@pytest.mark.skipif(not H.have_ref_viso(), reason="needs the reference in oracle/_ref")
@pytest.mark.parametrize("kw", [
    {}, {"reweighting": 0}, {"ransac_iters": 37, "inlier_threshold": 1.2},
    {"bucket_max_features": 5, "bucket_width": 80.0, "bucket_height": 40.0},
])
def test_oracle_matches_reference_live(kw, oracle_lib):
    prm = H.vo_defaults(**kw)
    a = run(H.RefVo(prm), quad())
    b = run(H.OracleVo(prm), quad())
    assert a[0] == b[0]
    assert a[1].tobytes() == b[1].tobytes()
    assert np.array_equal(a[2], b[2])
    assert np.array_equal(a[3], b[3])
    assert a[4] == b[4]


@pytest.mark.skipif(not H.have_ref_viso(), reason="needs the reference in oracle/_ref")
@pytest.mark.parametrize("n,seed,kw", [
    (400, 1, {}), (60, 2, {"outliers": 0.5}), (7, 3, {"outliers": 0.0}), (5, 4, {}),
    (300, 5, {"outliers": 0.97}), (200, 6, {"noise": 3.0}),
])
def test_estimate_motion_matches_reference(n, seed, kw, oracle_lib):
    """estimateMotion alone: synthetic matches, incl. too few matches (<6 -> empty vector),
    too few inliers and non-converging cases; both sides start from srand(0)"""
    syn = H.synth_vo_matches(n, seed=seed, **kw)
    prm = H.vo_defaults()
    a, b = H.RefVo(prm), None
    ra = a.estimate_motion(syn)
    ia = a.inliers()
    b = H.OracleVo(prm)
    rb = b.estimate_motion(syn)
    assert ra[0] == rb[0]
    if ra[0]:
        assert np.array_equal(ra[1], rb[1])
    assert np.array_equal(ia, b.inliers())

@pytest.mark.skipif(not H.have_ref_viso(), reason="needs the reference in oracle/_ref")
@pytest.mark.parametrize("seed", range(200, 208))
def test_oracle_matches_reference_param_fuzz(seed, oracle_lib):
    prm = H.fuzz_vo_params(seed)
    a = run(H.RefVo(prm), quad())
    b = run(H.OracleVo(prm), quad())
    assert a[0] == b[0]
    assert a[1].tobytes() == b[1].tobytes()
    assert np.array_equal(a[2], b[2])
    assert np.array_equal(a[3], b[3])
    assert a[4] == b[4]

"""GPU parity of VisualOdometryStereo (svh_vo_* C-ABI; RANSAC + Gauss-Newton in
vo_kernels.hip) against the oracle and the reference's golden output.

Integer results -- bucketed matches, the libc rand() sample stream, inlier INDICES -- must be
identical.  The motion is fp64 with the reference's operation order; the only difference to
the CPU is the device libm's sin/cos (last-bit), so the tolerance is 1e-9 absolute on the six
motion parameters and on the 4x4 matrix (observed: <= 1e-13)."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
TOL = 1e-9


def quad():
    return [H.read_pgm(os.path.join(H.GOLDEN, "viso_%s.pgm" % k)) for k in ("I1p", "I2p", "I1c", "I2c")]


def run(vo, im):
    r = (vo.process(im[0], im[1]), vo.process(im[2], im[3]))
    return r, vo.matches(), vo.inliers(), vo.motion(), vo.gain(vo.inliers())


def test_process_matches_golden_reference_output():
    z = np.load(os.path.join(H.GOLDEN, "vo_quad.npz"))
    prm = H.VoParams.from_buffer_copy(z["params"].tobytes())
    r, m, inl, T, gain = run(H.ProductVo(prm), quad())
    assert list(r) == list(z["ok"])
    assert m.tobytes() == z["matches"].tobytes()      # matches after bucketing: bit-exact
    assert np.array_equal(inl, z["inliers"])          # inlier indices: identical
    assert np.abs(T - z["motion"]).max() < TOL
    assert np.float32(gain) == z["gain"]


@pytest.mark.parametrize("kw", [
    {}, {"reweighting": 0}, {"ransac_iters": 37, "inlier_threshold": 1.2},
    {"bucket_max_features": 5, "bucket_width": 80.0, "bucket_height": 40.0},
])
def test_process_matches_oracle(kw, oracle_lib):
    if not H.have_ref_viso():
        pytest.skip("oracle needs the real Triangle (oracle/_ref) for removeOutliers")
    prm = H.vo_defaults(**kw)
    a = run(H.OracleVo(prm), quad())
    b = run(H.ProductVo(prm), quad())
    assert a[0] == b[0]
    assert a[1].tobytes() == b[1].tobytes()
    assert np.array_equal(a[2], b[2])
    assert np.abs(a[3] - b[3]).max() < TOL
    assert a[4] == b[4]


@pytest.mark.parametrize("n,seed,kw", [
    (400, 1, {}), (60, 2, {"outliers": 0.5}), (7, 3, {"outliers": 0.0}), (5, 4, {}),
    (300, 5, {"outliers": 0.97}), (200, 6, {"noise": 3.0}), (3000, 8, {}),
])
def test_estimate_motion_matches_oracle(n, seed, kw, oracle_lib):
    """estimateMotion alone on synthetic matches: N < 6 (empty vector, inliers untouched),
    almost no inliers, heavy noise, a large set; both sides start from srand(0)"""
    syn = H.synth_vo_matches(n, seed=seed, **kw)
    prm = H.vo_defaults()
    a = H.OracleVo(prm)
    ra = a.estimate_motion(syn)
    ia = a.inliers()
    b = H.ProductVo(prm)
    rb = b.estimate_motion(syn)
    assert ra[0] == rb[0]
    if ra[0]:
        assert np.abs(ra[1] - rb[1]).max() < TOL
    assert np.array_equal(ia, b.inliers())


def test_golden_synthetic_estimate():
    z = np.load(os.path.join(H.GOLDEN, "vo_quad.npz"))
    vo = H.ProductVo(H.VoParams.from_buffer_copy(z["params"].tobytes()))
    ok, tr = vo.estimate_motion(z["syn_matches"])
    assert ok == int(z["syn_ok"]) and np.abs(tr - z["syn_tr"]).max() < TOL
    assert np.array_equal(vo.inliers(), z["syn_inliers"])


def test_cxx_dropin_visual_odometry_thread(tmp_path):
    """tests/cxx/vo_dropin.cpp = stereomapper/visualodometrythread.cpp's call sequence built
    against include/viso_stereo.h; its output must equal the reference's golden"""
    exe = os.path.join(H.ROOT, "tests", "cxx", "vo_dropin")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(H.ROOT, "tests", "cxx")])
    out = str(tmp_path / "vo.bin")
    args = [os.path.join(H.GOLDEN, "viso_%s.pgm" % k) for k in ("I1p", "I2p", "I1c", "I2c")]
    r = subprocess.run([exe] + args + [out], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(os.path.join(H.GOLDEN, "vo_quad.npz"))
    raw = open(out, "rb").read()
    T = np.frombuffer(raw[:128], np.float64).reshape(4, 4)
    rpyv = np.frombuffer(raw[128:160], np.float64)
    gain = np.frombuffer(raw[160:164], np.float32)[0]
    n = np.frombuffer(raw[164:172], np.int32)
    inl = np.frombuffer(raw[172:], np.int32)
    assert np.abs(T - z["motion"]).max() < TOL
    assert n[0] == len(z["matches"]) and n[1] == len(z["inliers"]) and np.array_equal(inl, z["inliers"])
    assert gain == z["gain"]
    M = z["motion"]
    want = [np.arctan2(-M[0, 1], M[0, 0]), np.arctan2(-M[1, 2], M[2, 2]),
            np.arctan2(M[0, 2], np.hypot(M[0, 0], M[0, 1])), np.hypot(M[0, 3], M[2, 3])]
    assert np.abs(rpyv - want).max() < 1e-9


def test_no_matches_and_bad_arguments():
    vo = H.ProductVo(H.vo_defaults())
    ok, _ = vo.estimate_motion(np.zeros(0, H.P_MATCH))
    assert ok == 0 and len(vo.inliers()) == 0
    assert np.array_equal(vo.motion(), np.eye(4))
    assert vo.lib.svh_vo_estimate_motion(vo.h, None, 10, None) < 0

@pytest.mark.parametrize("seed", range(200, 208))
def test_param_fuzz_matches_oracle(seed, oracle_lib):
    if not H.have_ref_viso():
        pytest.skip("oracle needs the real Triangle (oracle/_ref) for removeOutliers")
    prm = H.fuzz_vo_params(seed)
    a = run(H.OracleVo(prm), quad())
    b = run(H.ProductVo(prm), quad())
    assert a[0] == b[0]
    assert a[1].tobytes() == b[1].tobytes()
    assert np.array_equal(a[2], b[2])
    assert np.abs(a[3] - b[3]).max() < TOL
    assert a[4] == b[4]

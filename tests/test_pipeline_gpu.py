"""The whole stereomapper data path on the device (tools/stereomapper_pipeline.py): KITTI-shaped
drive on disk -> visual odometry -> ELAS -> map fusion.  Each stage has its own parity test; this
one checks that they compose: the pipeline's map equals the oracle's map fed with the pipeline's own
poses and disparity maps."""
import os
import sys

import numpy as np
import pytest

import helpers as H
import test_kitti_io as TK
import test_map as TM

sys.path.insert(0, os.path.join(H.ROOT, "tools"))


@pytest.mark.gpu
def test_drive_to_map(tmp_path, oracle_lib):
    import stereomapper_pipeline as SP
    from svhip import kitti
    # a drive of 4 frames: the reference's two consecutive quad pairs, twice
    pairs = [(H.read_pgm(os.path.join(H.GOLDEN, "viso_I1p.pgm")), H.read_pgm(os.path.join(H.GOLDEN, "viso_I2p.pgm"))),
             (H.read_pgm(os.path.join(H.GOLDEN, "viso_I1c.pgm")), H.read_pgm(os.path.join(H.GOLDEN, "viso_I2c.pgm")))]
    root = tmp_path / "drive"
    for k in range(2):
        (root / ("image_0%d" % k) / "data").mkdir(parents=True)
        lines = []
        for i in range(4):
            lines.append("2011-09-26 13:02:%02d.%09d" % (25 + i, 100000000 * i))
            TK.write_png(str(root / ("image_0%d" % k) / "data" / ("%010d.png" % i)),
                         pairs[i % 2][k][:, :, None], filters=[i % 5, 2])
        (root / ("image_0%d" % k) / "timestamps.txt").write_text("\n".join(lines) + "\n")
    f, cu, cv, base = 645.24, 635.96, 194.13, 0.5707      # libviso2 demo.cpp calibration
    p = SP.Pipeline(f, cu, cv, base)
    o = TM.OracleMapper(TM.oracle_map(oracle_lib), TM.MapParams(f, cu, cv, base, 20))
    oks = []
    for i, (I1, I2, _) in enumerate(kitti.Sequence(root)):
        assert np.array_equal(I1, pairs[i % 2][0]) and np.array_equal(I2, pairs[i % 2][1])
        ok, n0, n1 = p.push(I1, I2)
        oks.append(ok)
        # the oracle replays the fusion from the pipeline's own disparity map, pose and gain
        D1 = p.buf[2].download(np.zeros(I1.shape, np.float32))
        gain = float(p.vo.gain(p.vo.inliers())) if ok else 0.0
        o.add(D1, I1, p.poses[-1], gain)
        for which in (0, 1):
            a, b = o.points(which), p.map.points(which)
            assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), (i, which)
        assert n1 == len(p.map.points(1)) and n1 > 20000
    assert oks[0] is False and any(oks[1:])       # the first frame only fills the ring buffer
    assert not np.allclose(p.poses[-1], np.eye(4))

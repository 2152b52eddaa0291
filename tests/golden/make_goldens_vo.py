"""Generate the VisualOdometryStereo golden fixture.  RUNS ONLY IN THE BUILD CONTAINER.

Inputs : the reference's quad images (tests/golden/viso_*.pgm, see make_goldens_viso.py) and
         the calibration of libviso2/src/demo.cpp:54-58.
Outputs: what the REFERENCE itself (oracle/_ref/libref_viso.so) returns for
         VisualOdometryStereo::process on the two frames: bucketed matches, inlier indices,
         the 4x4 delta motion, gain -- plus estimateMotion on a seeded synthetic match set.

    python tests/golden/make_goldens_vo.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402


def main():
    im = [H.read_pgm(os.path.join(HERE, "viso_%s.pgm" % k)) for k in ("I1p", "I2p", "I1c", "I2c")]
    prm = H.vo_defaults()
    vo = H.RefVo(prm)
    r0 = vo.process(im[0], im[1])
    r1 = vo.process(im[2], im[3])
    out = {"params": np.frombuffer(bytes(prm), np.uint8), "ok": np.array([r0, r1]),
           "matches": vo.matches(), "inliers": vo.inliers(), "motion": vo.motion(),
           "gain": np.array(vo.gain(vo.inliers()), np.float32)}
    # estimateMotion alone on synthetic matches (fresh object => srand(0) stream)
    syn = H.synth_vo_matches(400, seed=7)
    vo2 = H.RefVo(prm)
    ok, tr = vo2.estimate_motion(syn)
    out.update(syn_matches=syn, syn_ok=np.array(ok), syn_tr=tr, syn_inliers=vo2.inliers())
    path = os.path.join(HERE, "vo_quad.npz")
    np.savez_compressed(path, **out)
    print("vo_quad", os.path.getsize(path) // 1024, "KiB", r0, r1, len(out["matches"]), len(out["inliers"]),
          "syn", ok, len(out["syn_inliers"]))


if __name__ == "__main__":
    main()

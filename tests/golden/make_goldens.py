"""Generate the ELAS golden fixtures.  RUNS ONLY IN THE BUILD CONTAINER.

Inputs : crops of the reference's own test images (/root/reference/libelas/img/*.pgm),
         written next to this script as PGM (they are data files, not source).
Outputs: per-stage results of the REFERENCE itself (oracle/_ref/libref_elas.so, i.e.
         /root/reference/libelas compiled by oracle/Makefile), stored as .npz.

    python tests/golden/make_goldens.py

The GPU box never runs this; tests only read the committed files.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402

IMG = "/root/reference/libelas/img"

# name -> (source image, x, y, w, h)
CROPS = {
    "urban1_1242x375": ("urban1", 51, 8, 1242, 375),   # KITTI-sized stand-in
    "urban2_1242x375": ("urban2", 51, 8, 1242, 375),
    "urban3_640x240": ("urban3", 300, 100, 640, 240),
    "urban3_1242x375": ("urban3", 51, 8, 1242, 375),   # with urban1/2/4: the four KITTI-size crops
    "urban4_1242x375": ("urban4", 51, 8, 1242, 375),   # of SURVEY 8(d) config 1/2 (bench headline)
    "cones_640x480": ("cones", 130, 135, 640, 480),
}

# case -> (crop, params)
CASES = {
    "urban1_robotics": ("urban1_1242x375", H.robotics()),
    # stereomapper/stereothread.cpp:76-80 setting
    "urban2_stereomapper": ("urban2_1242x375", H.robotics(support_texture=30)),
    # libelas/src/main.cpp:61-63 setting (both maps post-processed)
    "urban3_demo": ("urban3_640x240", H.robotics(postprocess_only_left=0)),
    "cones_middlebury": ("cones_640x480", H.middlebury()),
    "urban3_kitti": ("urban3_1242x375", H.robotics()),
    "urban4_kitti": ("urban4_1242x375", H.robotics()),
    # the bench's second pair with the bench's parameters (urban2_stereomapper is support_texture=30)
    "urban2_kitti": ("urban2_1242x375", H.robotics()),
}
SLIM = {"urban2_kitti", "urban3_kitti", "urban4_kitti"}   # support, triangles and final maps only

INT_STAGES = [H.D1_RAW, H.D2_RAW, H.D1_LR, H.D2_LR, H.D1_SEG, H.D2_SEG]   # integer valued


def main(only=()):
    """`only`: case names to (re)generate; default all"""
    for name, (src, x, y, w, h) in CROPS.items():
        if only and not any(CASES[c][0] == name for c in only):
            continue
        for side in ("left", "right"):
            img = H.read_pgm(os.path.join(IMG, f"{src}_{side}.pgm"))[y:y + h, x:x + w]
            H.write_pgm(os.path.join(HERE, f"{name}_{side}.pgm"), img)
    for case, (crop, prm) in CASES.items():
        if only and case not in only:
            continue
        l, r = H.golden_pair(crop)
        run = H.ref_elas_run(prm, l, r)
        assert run.status == 0
        out = {"params": np.frombuffer(bytes(prm), np.uint8), "crop": np.array(crop)}
        if case in SLIM:
            for s in (H.SUPPORT, H.TRI1, H.TRI2):
                out[H.STAGE_NAMES[s]] = run[s]
            out["d1"] = run[H.D1_FINAL]
            out["d2"] = run[H.D2_FINAL]
            np.savez_compressed(os.path.join(HERE, case + ".npz"), **out)
            print(case, "support", len(run[H.SUPPORT]) // 3, "tri", len(run[H.TRI1]) // 3, len(run[H.TRI2]) // 3)
            continue
        for s in (H.DESC1, H.DESC2):
            out[H.STAGE_NAMES[s] + "_sha256"] = np.array(hashlib.sha256(run[s].tobytes()).hexdigest())
        # a thin, exact sample of the descriptor: every 16th row
        hgt, wid = l.shape
        d1 = run[H.DESC1].reshape(hgt, wid, 16)
        out["desc1_rows16"] = d1[::16].copy()
        for s in (H.SUPPORT, H.TRI1, H.TRI2, H.PLANES1, H.PLANES2):
            out[H.STAGE_NAMES[s]] = run[s]
        for s in (H.GRID1, H.GRID2):
            g = run[s].reshape(-1, prm.disp_max + 2)
            out[H.STAGE_NAMES[s] + "_count"] = g[:, 0].astype(np.int16)
            out[H.STAGE_NAMES[s] + "_sha256"] = np.array(hashlib.sha256(run[s].tobytes()).hexdigest())
        for s in INT_STAGES:
            assert np.all(run[s] == np.rint(run[s]))
            out[H.STAGE_NAMES[s] + "_i16"] = run[s].astype(np.int16)
        out["d1_gap"] = run[H.D1_GAP]
        out["d2_gap"] = run[H.D2_GAP]
        out["d1"] = run[H.D1_FINAL]
        out["d2"] = run[H.D2_FINAL]
        path = os.path.join(HERE, case + ".npz")
        np.savez_compressed(path, **out)
        print(case, os.path.getsize(path) // 1024, "KiB", "support", len(run[H.SUPPORT]) // 3,
              "tri", len(run[H.TRI1]) // 3, len(run[H.TRI2]) // 3)


if __name__ == "__main__":
    main(tuple(sys.argv[1:]))

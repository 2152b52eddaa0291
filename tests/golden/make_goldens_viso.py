"""Generate the libviso2 Matcher golden fixtures.  RUNS ONLY IN THE BUILD CONTAINER.

Inputs : the reference's own quad-match images libviso2/img/{I1p,I2p,I1c,I2c}.png
         (demo_matching_quad.m:6-9), stored here as PGM (pixel data, not source).
Outputs: feature tables and per-stage match lists produced by the REFERENCE itself
         (oracle/_ref/libref_viso.so = /root/reference/libviso2 compiled by oracle/Makefile).

    python tests/golden/make_goldens_viso.py
"""
import os
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402

IMG = "/root/reference/libviso2/img"
# KITTI-like intrinsics only steer the optional match prediction (matcher.cpp:1312-1327)
TR = np.array([[0.9999, 0.002, -0.01, 0.02], [-0.002, 0.9999, 0.003, -0.01],
               [0.01, -0.003, 0.9999, -0.75], [0, 0, 0, 1]], np.float64)
CASES = {
    "viso_quad_default": (H.matcher_defaults(), 2, None),
    "viso_quad_predicted": (H.matcher_defaults(f=645.24, cu=635.96, cv=194.13, base=0.5707), 2, TR),
    "viso_stereo_default": (H.matcher_defaults(), 1, None),
    "viso_flow_default": (H.matcher_defaults(), 0, None),
}


def main():
    imgs = {}
    for k in ("I1p", "I2p", "I1c", "I2c"):
        imgs[k] = np.array(Image.open(os.path.join(IMG, k + ".png")).convert("L"))
        H.write_pgm(os.path.join(HERE, "viso_" + k + ".pgm"), imgs[k])
    for case, (prm, method, tr) in CASES.items():
        m = H.RefMatcher(prm)
        m.push_back(imgs["I1p"], imgs["I2p"])
        m.push_back(imgs["I1c"], imgs["I2c"])
        m.match(method, tr, staged=True)
        out = {"params": np.frombuffer(bytes(prm), np.uint8), "method": np.array(method),
               "tr": TR if tr is not None else np.zeros(0)}
        for tb in range(8):
            out["table_" + H.M_TABLES[tb]] = m.features(tb)
        for s in range(H.M_STAGE_COUNT):
            out[H.M_STAGE_NAMES[s]] = m.stage(s)
        path = os.path.join(HERE, case + ".npz")
        np.savez_compressed(path, **out)
        print(case, os.path.getsize(path) // 1024, "KiB",
              {H.M_STAGE_NAMES[s]: len(m.stage(s)) for s in range(H.M_STAGE_COUNT)})


if __name__ == "__main__":
    main()

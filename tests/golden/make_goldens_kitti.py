"""Fixture for the PNG reader.  RUNS ONLY IN THE BUILD CONTAINER.

Input : one of the reference's own PNG frames, libviso2/img/I1c.png (a data file), kept as is.
Output: tests/golden/viso_I1c.png (the file) -- its expected pixels are tests/golden/viso_I1c.pgm,
        written by make_goldens_viso.py through PIL (libpng); this script checks that they still
        correspond.

    python tests/golden/make_goldens_kitti.py
"""
import os
import shutil
import sys

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers as H  # noqa: E402

SRC = "/root/reference/libviso2/img/I1c.png"


def main():
    dst = os.path.join(HERE, "viso_I1c.png")
    shutil.copyfile(SRC, dst)
    os.chmod(dst, 0o644)
    want = H.read_pgm(os.path.join(HERE, "viso_I1c.pgm"))
    got = np.array(Image.open(dst).convert("L"))
    assert np.array_equal(want, got)
    print("viso_I1c.png", os.path.getsize(dst) // 1024, "KiB", got.shape)


if __name__ == "__main__":
    main()

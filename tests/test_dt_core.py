"""CPU: the Delaunay core of k_delaunay (stereo-vision_amd/csrc/dt_core.h: triangle records, leaves, the merge of two
halves with Triangle's tie rules and record order -- libelas/src/triangle.cpp:5638-5934, 5953-6103) compiled for the
host and driven as the kernel drives it (alternating-cut order, bottom-up by depth).  Its triangle lists must equal
csrc/delaunay.cpp's, order included -- which test_oracle_elas.py pins against the real Triangle -- on lattice points,
pixel points, collinear and tiny sets, for both record storages and both forms of the seam step."""
import os
import subprocess

import helpers as H

CXX = os.path.join(H.ROOT, "tests", "cxx")


def test_dt_core_equals_host_delaunay():
    subprocess.check_call(["make", "-C", CXX, "dt_core_check"], stdout=subprocess.DEVNULL)
    for seed in (20260929, 7):
        r = subprocess.run([os.path.join(CXX, "dt_core_check"), "1500", str(seed)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "mismatches 0" in r.stdout and "shortcuts wrong 0" in r.stdout, r.stdout

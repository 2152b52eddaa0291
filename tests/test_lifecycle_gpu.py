"""Objects come and go without leaking device memory: Matcher, VisualOdometryStereo and the map
fusion release everything they allocate; the ELAS lane pool is process-wide by design (lanes
are reused by later objects), so its footprint must stop growing once it is warm."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers as H


def free_bytes():
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


@pytest.mark.gpu
def test_create_destroy_cycles_do_not_leak():
    import gc
    import svhip as S
    from svhip import mapper
    im = [H.read_pgm(os.path.join(H.GOLDEN, "viso_%s.pgm" % k)) for k in ("I1p", "I2p", "I1c", "I2c")]
    l, r = H.golden_pair("urban3_640x240")
    D1 = np.zeros(l.shape, np.float32)
    D2 = D1.copy()

    def cycle():
        m = H.ProductMatcher(H.matcher_defaults())
        m.push_back(im[0], im[1])
        m.push_back(im[2], im[3])
        m.match(2)
        vo = H.ProductVo(H.vo_defaults())
        vo.process(im[0], im[1])
        vo.process(im[2], im[3])
        e = S.Elas(H.robotics())
        e.process(l, r, D1, D2)
        mp = mapper.Mapper(645.24, 321.0, 118.0, 0.5707)
        mp.add(D1, l, np.eye(4), 1.0)
        mp.add(D1, l, np.eye(4), 1.0)
        assert len(mp.points(1)) > 1000
        del m, vo, e, mp
        gc.collect()

    for _ in range(3):      # warm: HIP context, code objects, the ELAS lane used by process()
        cycle()
    before = free_bytes()
    for _ in range(25):
        cycle()
    after = free_bytes()
    assert before - after < 8 << 20, "device memory shrank by %.1f MB over 25 cycles" % ((before - after) / 2**20)


@pytest.mark.gpu
def test_trim_releases_the_idle_lanes():
    """svh_elas_trim(): a long-lived process gets the lane pool's memory back after a large batch
    (each lane holds device buffers for a whole group of pairs plus pinned staging)"""
    import svhip as S
    w, h = 640, 240
    l, r = H.golden_pair("urban3_640x240")
    e = S.Elas(H.robotics())
    n = 48
    S.set_lanes(6)
    S.set_group(4)
    try:
        e.process_batch(np.stack([l] * n), np.stack([r] * n))     # warm: code objects, runtime pools
        S.trim()
        base = free_bytes()
        st, A1, A2 = e.process_batch(np.stack([l] * n), np.stack([r] * n))
        assert st == [0] * n
        held = base - free_bytes()
        assert held > 64 << 20                       # the pool keeps its lanes ...
        assert S.trim() >= 6                         # ... until it is told to let go
        assert base - free_bytes() < held // 8
        st, B1, B2 = e.process_batch(np.stack([l] * 4), np.stack([r] * 4))   # and regrows on demand
        assert st == [0] * 4 and np.array_equal(A1[0], B1[0]) and np.array_equal(A2[3], B2[3])
    finally:
        S.set_lanes(8)
        S.set_group(16)
        S.trim()

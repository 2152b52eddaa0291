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


@pytest.mark.gpu
def test_a_plain_process_gets_the_engines_settings_without_any_environment_variable():
    """svh_init by first use (include/svh.h): a fresh process that loads only libsvhip -- no torch, no environment
    variable -- runs a batch; the library asked the HIP runtime for 20 hardware queues before the runtime started
    (state "applied"), and the maps are the reference's"""
    import json
    import subprocess
    import sys
    code = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import helpers as H, svhip as S
before = S.runtime_info()
z = np.load(os.path.join(H.GOLDEN, "urban3_demo.npz"))
prm = H.ElasParams.from_buffer_copy(z["params"].tobytes())
l, r = H.golden_pair(str(z["crop"]))
st, D1, D2 = S.Elas(prm).process_batch(np.stack([l] * 5), np.stack([r] * 5))
ok = st == [0] * 5 and all(np.array_equal(D1[k].ravel(), z["d1"]) and np.array_equal(D2[k].ravel(), z["d2"]) for k in range(5))
print(json.dumps({"before": before, "after": S.runtime_info(), "settings": S.elas_settings(), "ok": bool(ok)}))
''' % (os.path.join(H.ROOT, "stereo-vision_amd"), os.path.join(H.ROOT, "tests"))
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "SVH_HW_QUEUES")}
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["before"]["initialised"] == 0 and out["before"]["env_modified"] == 0
    a = out["after"]
    assert a["initialised"] == 1 and a["implicit"] == 1 and a["hip_started_before"] == 0
    assert a["hw_queues_state"] == "applied" and a["hw_queues_env"] == 20 and a["env_modified"] == 1
    assert out["settings"]["workers"] == 6 and out["ok"]

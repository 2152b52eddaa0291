"""GPU: device-error behaviour at the boundary (SURVEY 8(b) "Errors").  The reference reports a failure with a
message and returns (libelas/src/elas.cpp:69-75, libviso2/src/matcher.cpp:110-114); a HIP failure inside the
library must surface the same way: the entry returns SVH_ERR_HIP, svh_last_error() names the call, one line goes to
stderr, nothing hangs, nothing leaks, the object works on the next call.  Failures are injected with the library's
test hook svh_test_fail_at("<malloc|launch|copy|wait>:<n>[:<count>]") (include/svh.h): the n-th HIP call of that
kind is not issued and reports an error instead.  Entries driven into each kind: the single call, the batch entry
(6 workers), a stream with two producers, the lockstep Matcher and the pipelined lockstep visual odometry (K = 16,
prefetch thread)."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

KINDS = ("malloc", "copy", "launch", "wait")


@pytest.fixture(scope="module")
def svhip():
    import svhip as S
    S.lib()
    assert S.device_count() > 0, "no HIP device: the product has no CPU fallback"
    S.lib().svh_test_fail_at.argtypes = [C.c_char_p]
    return S


@pytest.fixture(autouse=True)
def disarm(svhip):
    yield
    svhip.lib().svh_test_fail_at(None)
    svhip.set_stage(-1)
    svhip.set_lanes(8)


def arm(S, spec):
    assert S.lib().svh_test_fail_at(spec.encode() if spec else None) == 0


@pytest.fixture(scope="module")
def pair():
    z = np.load(os.path.join(H.GOLDEN, "urban3_demo.npz"))
    prm = H.ElasParams.from_buffer_copy(z["params"].tobytes())
    l, r = H.golden_pair(str(z["crop"]))
    return prm, l, r, z["d1"].reshape(l.shape), z["d2"].reshape(l.shape)


def free_bytes():
    """free device memory as the HIP runtime the library runs on reports it"""
    hip = C.CDLL("libamdhip64.so")
    free, total = C.c_size_t(0), C.c_size_t(0)
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_bad_specifications_are_refused(svhip):
    for bad in ("malloc", "nothing:1", "copy:0", "wait:-3", "launch:1:-1"):
        assert svhip.lib().svh_test_fail_at(bad.encode()) == svhip.ERR_BAD_ARG, bad
    arm(svhip, "")


@pytest.mark.parametrize("kind", KINDS)
def test_single_call_reports_and_recovers(svhip, pair, kind, capfd):
    """Elas::process: every position of the failing call from the first to past the last of its kind"""
    prm, l, r, g1, g2 = pair
    e = svhip.Elas(prm)
    seen = 0
    n = 1
    while True:
        svhip.trim()                       # fresh lane: the allocations happen inside the call
        D1 = np.full(l.shape, -7.0, np.float32)
        D2 = np.full(l.shape, -7.0, np.float32)
        arm(svhip, "%s:%d" % (kind, n))
        capfd.readouterr()
        try:
            rc, _, _ = e.process(l, r, D1, D2)
            failed = False
        except svhip.SvhError as err:
            failed = True
            assert err.code == svhip.ERR_HIP
            assert "injected failure" in str(err), str(err)
            assert capfd.readouterr().err.count("svhip:") == 1       # one line, once
        arm(svhip, "")
        if not failed:
            # the call has fewer than n calls of this kind: it ran to the end, untouched by the hook
            assert rc == 0 and np.array_equal(D1, g1) and np.array_equal(D2, g2)
            break
        seen += 1
        if kind == "malloc":
            assert (D1 == -7.0).all() and (D2 == -7.0).all()     # nothing ran: the outputs are untouched
        rc, A1, A2 = e.process(l, r)        # the next call on the same object
        assert rc == 0 and np.array_equal(A1, g1) and np.array_equal(A2, g2), (kind, n)
        n += 1 if n < 6 else 5
        assert n < 400
    assert seen >= 1, "no call of kind %s inside svh_elas_process" % kind


def test_allocation_failure_leaves_outputs_untouched(svhip, pair):
    prm, l, r, g1, g2 = pair
    e = svhip.Elas(prm)
    for n in (1, 2, 5, 9):
        svhip.trim()
        D1 = np.full(l.shape, -7.0, np.float32)
        D2 = np.full(l.shape, -7.0, np.float32)
        arm(svhip, "malloc:%d" % n)
        with pytest.raises(svhip.SvhError):
            e.process(l, r, D1, D2)
        arm(svhip, "")
        assert (D1 == -7.0).all() and (D2 == -7.0).all()


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("stage", [0, 1])
def test_batch_entry_reports_and_recovers(svhip, pair, kind, stage):
    """40 pairs over 6 double-buffered workers, host stage and device stage: the groups of the failing worker carry
    SVH_ERR_HIP, the call returns it, the same call succeeds right after"""
    prm, l, r, g1, g2 = pair
    e = svhip.Elas(prm)
    svhip.set_lanes(6)
    svhip.set_group(4)
    svhip.set_stage(stage)
    I1 = np.stack([l] * 40)
    I2 = np.stack([r] * 40)
    for n in (1, 3, 7, 20):
        if kind == "malloc":
            svhip.trim()
        arm(svhip, "%s:%d" % (kind, n))
        try:
            st, _, _ = e.process_batch(I1, I2)
            failed = False
        except svhip.SvhError as err:
            failed = True
            assert err.code == svhip.ERR_HIP and "injected failure" in str(err)
        arm(svhip, "")
        if not failed:
            assert all(s == 0 for s in st)
        st, D1, D2 = e.process_batch(I1, I2)
        assert all(s == 0 for s in st)
        for k in (0, 17, 39):
            assert np.array_equal(D1[k], g1) and np.array_equal(D2[k], g2)
    svhip.set_group(16)


def test_batch_statuses_name_the_failed_groups_only(svhip, pair):
    prm, l, r, g1, g2 = pair
    e = svhip.Elas(prm)
    svhip.set_lanes(3)
    svhip.set_group(4)
    svhip.set_stage(1)
    n = 24
    I1 = np.ascontiguousarray(np.stack([l] * n))
    I2 = np.ascontiguousarray(np.stack([r] * n))
    e.process_batch(I1, I2)                       # lanes sized
    D1 = np.full((n,) + l.shape, -7.0, np.float32)
    D2 = np.full((n,) + l.shape, -7.0, np.float32)
    arr = C.c_void_p * n
    st = (C.c_int32 * n)()
    dims = (C.c_int32 * 3)(l.shape[1], l.shape[0], l.shape[1])
    arm(svhip, "launch:2")
    rc = svhip.lib().svh_elas_process_batch(e._h, n, arr(*[I1[i].ctypes.data for i in range(n)]),
                                            arr(*[I2[i].ctypes.data for i in range(n)]),
                                            arr(*[D1[i].ctypes.data for i in range(n)]),
                                            arr(*[D2[i].ctypes.data for i in range(n)]), dims, st)
    arm(svhip, "")
    st = list(st)
    assert rc == svhip.ERR_HIP and st.count(svhip.ERR_HIP) == 4, st       # one group of four
    for k in range(n):
        if st[k] == 0:
            assert np.array_equal(D1[k], g1) and np.array_equal(D2[k], g2), k
    svhip.set_group(16)


@pytest.mark.parametrize("kind", KINDS)
def test_stream_with_two_producers(svhip, pair, kind):
    """two threads push, one pops; a failure hits some group in the middle: its pairs come back with SVH_ERR_HIP in
    ticket order, every other pair is right, close() returns"""
    prm, l, r, g1, g2 = pair
    e = svhip.Elas(prm)
    svhip.set_lanes(4)
    svhip.set_group(4)
    h, w = l.shape
    s = e.stream(w, h, depth=24)
    per = 30
    D = {}
    lock = threading.Lock()

    def producer(tag):
        for i in range(per):
            d1 = np.full((h, w), -7.0, np.float32)
            d2 = np.full((h, w), -7.0, np.float32)
            with lock:
                t = s.push(l, r, d1, d2)
                D[t] = (d1, d2)

    if kind == "malloc":
        svhip.trim()
    arm(svhip, "%s:%d" % (kind, 9 if kind != "malloc" else 3))
    th = [threading.Thread(target=producer, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    got = []
    while len(got) < 2 * per:
        rr = s.pop()
        if rr is None:
            continue
        got.append(rr)
    for t in th:
        t.join()
    arm(svhip, "")
    s.close()
    assert [t for t, _ in got] == list(range(2 * per))
    bad = [t for t, st in got if st != 0]
    assert all(st in (0, svhip.ERR_HIP) for _, st in got)
    # (an allocation fails while a worker sizes its lanes ahead of the work: the group's own attempt succeeds and
    # nothing is lost; the other kinds hit a group in flight)
    assert (0 if kind == "malloc" else 1) <= len(bad) <= 8, bad
    for t, st in got:
        if st == 0:
            assert np.array_equal(D[t][0], g1) and np.array_equal(D[t][1], g2), t
    svhip.set_group(16)


def test_two_streams_share_the_lane_pool(svhip, pair):
    """(round 4's advisor finding) two open streams on one device, closed in LIFO order while both hold pairs:
    workers take lanes only while they have groups in flight, so neither stream starves the other, single calls
    still run, and both close() calls return"""
    prm, l, r, g1, g2 = pair
    e = svhip.Elas(prm)
    svhip.set_lanes(8)
    svhip.set_group(4)
    h, w = l.shape
    a = e.stream(w, h)
    b = e.stream(w, h)
    Da = [(np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)) for _ in range(24)]
    Db = [(np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)) for _ in range(24)]
    for i in range(24):
        a.push(l, r, Da[i][0], Da[i][1])
        b.push(l, r, Db[i][0], Db[i][1])
    rc, S1, S2 = e.process(l, r)                       # a single call in between
    assert rc == 0 and np.array_equal(S1, g1)
    done = []
    th = threading.Thread(target=lambda: (b.close(), a.close(), done.append(1)))
    th.start()
    th.join(60)
    assert done, "closing two streams hung"
    for D in (Da, Db):
        for d1, d2 in D:
            assert np.array_equal(d1, g1) and np.array_equal(d2, g2)
    svhip.set_group(16)


def test_close_wakes_a_producer_blocked_on_a_full_stream(svhip, pair):
    prm, l, r, g1, g2 = pair
    e = svhip.Elas(prm)
    h, w = l.shape
    s = e.stream(w, h, depth=2)
    D = [(np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)) for _ in range(4)]
    out = []

    def producer():
        try:
            for i in range(4):            # the third push blocks: nobody pops
                s.push(l, r, D[i][0], D[i][1])
            out.append("pushed all")
        except svhip.SvhError as err:
            out.append(err.code)

    th = threading.Thread(target=producer)
    th.start()
    import time
    time.sleep(0.5)
    s.close()
    th.join(30)
    assert out == [svhip.ERR_BAD_ARG], out      # "stream is closing"


def leak_trace():
    """free device memory (bytes) before every tenth of 150 injected failures -- single calls and batches, every kind,
    the lane pool trimmed before each reading -- and after the last one, then one good batch"""
    import svhip as S
    S.lib().svh_test_fail_at.argtypes = [C.c_char_p]
    z = np.load(os.path.join(H.GOLDEN, "urban3_demo.npz"))
    prm = H.ElasParams.from_buffer_copy(z["params"].tobytes())
    l, r = H.golden_pair(str(z["crop"]))
    e = S.Elas(prm)
    S.set_lanes(2)
    S.set_group(4)
    I1 = np.stack([l] * 8)
    I2 = np.stack([r] * 8)
    failed = 0
    trace = []
    for i in range(150):
        if i % 10 == 0:
            S.trim()
            trace.append(free_bytes())
        kind = KINDS[i % 4]
        if kind == "malloc":
            S.trim()
        arm(S, "%s:%d" % (kind, 1 + i % 5))
        try:
            e.process_batch(I1, I2) if i % 2 else e.process(l, r)
        except S.SvhError:
            failed += 1
        arm(S, "")
    S.trim()
    trace.append(free_bytes())
    st, D1, D2 = e.process_batch(I1, I2)
    good = all(x == 0 for x in st) and bool(np.array_equal(D1[7].ravel(), z["d1"]))
    return {"free_bytes": trace, "failed_calls": failed, "next_call_good": good}


def test_no_leak_over_150_failures():
    """in a process of its own: the HIP runtime the library is linked against, nothing else in the address space (in
    the test process torch's bundled runtime is loaded as soon as pytest has collected test_multirank.py, and under
    it plain hipMalloc / hipFree cycles of the lane pool -- with or without failures -- let the reported free memory
    drift by ~1 MB per cycle)"""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "leak_trace"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    t = d["free_bytes"]
    print("free device memory, MB below the first reading, every 10 failures:", [round((t[0] - x) / 2**20, 1) for x in t])
    assert d["failed_calls"] >= 60 and d["next_call_good"], d      # (an injection aimed past the last call of its kind misses)
    # the first ten calls load code objects and start the crew (a one-time step); from then on: flat
    # (free memory is a device-wide reading: under pytest-xdist the other workers' allocations move it)
    if "PYTEST_XDIST_WORKER" not in os.environ:
        assert max(t[1] - x for x in t[1:]) <= 4 << 20, t


# ---------------------------------------------------------------- Matcher / visual odometry
def quad():
    return [H.read_pgm(os.path.join(H.GOLDEN, "viso_%s.pgm" % k)) for k in ("I1p", "I2p", "I1c", "I2c")]


def variant(im, k):
    out = [np.roll(a, 3 * k, axis=1) for a in im]
    if k % 2:
        out = [np.ascontiguousarray(a[::-1]) for a in out]
    return out


def plain_matcher(prm):
    m = H.ProductMatcher(prm)
    m.lib.svh_matcher_set_taps(C.c_void_p(m.h), 0)
    return m


@pytest.mark.parametrize("kind", KINDS)
def test_matcher_single_object(svhip, kind, capfd):
    """Matcher::pushBack / matchFeatures: the failing call returns SVH_ERR_HIP, the matches of the call before
    stay, two more frames give exactly what an untouched object gives"""
    prm = H.matcher_defaults()
    im = quad()
    clean = plain_matcher(prm)
    clean.push_back(im[0], im[1])
    clean.push_back(im[2], im[3])
    clean.match(2)
    want = clean.matches().copy()
    for n in (1, 2, 4, 8, 15):
        m = plain_matcher(prm)
        m.push_back(im[0], im[1])
        m.push_back(im[2], im[3])
        m.match(2)
        kept = m.matches().copy()
        arm(svhip, "%s:%d" % (kind, n))
        capfd.readouterr()
        failed = 0
        try:
            m.push_back(im[0], im[1])
            m.match(2)
        except svhip.SvhError as err:
            failed = 1
            assert err.code == svhip.ERR_HIP
            assert capfd.readouterr().err.count("svhip:") == 1
        arm(svhip, "")
        if failed:
            assert (m.matches() == kept).all()          # the list of the last good call is untouched
        m.push_back(im[0], im[1])
        m.push_back(im[2], im[3])
        m.match(2)
        got = m.matches()
        assert len(got) == len(want) and (got == want).all(), (kind, n)


@pytest.mark.parametrize("kind", KINDS)
def test_matcher_lockstep(svhip, kind):
    K = 16
    prm = H.matcher_defaults()
    im = quad()
    seqs = [variant(im, k) for k in range(K)]
    clean = plain_matcher(prm)
    clean.push_back(seqs[3][0], seqs[3][1])
    clean.push_back(seqs[3][2], seqs[3][3])
    clean.match(2)
    want = clean.matches().copy()
    bat = [plain_matcher(prm) for _ in range(K)]
    for n in (1, 2, 5, 11):
        arm(svhip, "%s:%d" % (kind, n))
        try:
            H.product_matcher_batch(bat, [s[0] for s in seqs], [s[1] for s in seqs], None)
            H.product_matcher_batch(bat, [s[2] for s in seqs], [s[3] for s in seqs], 2)
        except svhip.SvhError as err:
            assert err.code == svhip.ERR_HIP
        arm(svhip, "")
        H.product_matcher_batch(bat, [s[0] for s in seqs], [s[1] for s in seqs], None)
        H.product_matcher_batch(bat, [s[2] for s in seqs], [s[3] for s in seqs], 2)
        got = bat[3].matches()
        assert len(got) == len(want) and (got == want).all(), (kind, n)


@pytest.mark.parametrize("kind", KINDS)
def test_vo_lockstep_pipelined_loop(svhip, kind):
    """K = 16 VisualOdometryStereo objects through svh_vo_process_next_batch (prefetch thread, helper pool): a
    failure anywhere in a frame returns an error or a count, never hangs; the loop goes on and estimates motion
    again"""
    K = 16
    im = quad()
    seqs = [variant(im, k) for k in range(K)]
    frames = [[np.ascontiguousarray(s[2 * (f % 2)]) for s in seqs] for f in range(2)]
    frames_r = [[np.ascontiguousarray(s[2 * (f % 2) + 1]) for s in seqs] for f in range(2)]
    shape = im[0].shape
    for n in (2, 6, 17):
        vos = [H.ProductVo(H.vo_defaults(), private_rand=0) for _ in range(K)]
        H.product_vo_prefetch_batch(vos, frames[0], frames_r[0])
        errors = 0
        for f in range(1, 9):
            if f == 3:
                arm(svhip, "%s:%d" % (kind, n))
            try:
                rc, ok = H.product_vo_process_next_batch(vos, frames[f % 2], frames_r[f % 2], shape)
            except svhip.SvhError as err:
                errors += 1
                assert err.code in (svhip.ERR_HIP, svhip.ERR_BAD_ARG), err
                # a frame that was being handed over may be lost with the failed call: start the hand-over again
                try:
                    H.product_vo_prefetch_batch(vos, frames[f % 2], frames_r[f % 2])
                except svhip.SvhError:
                    pass
            finally:
                if f == 3:
                    arm(svhip, "")
        rc, ok = H.product_vo_process_next_batch(vos, frames[1], frames_r[1], shape)
        rc, ok = H.product_vo_process_next_batch(vos, None, None, shape)
        assert rc >= K // 2, (kind, n, rc, ok)


if __name__ == "__main__":
    import json
    import sys
    sys.path.insert(0, os.path.join(H.ROOT, "stereo-vision_amd"))
    if sys.argv[1:] == ["leak_trace"]:
        print(json.dumps(leak_trace()))

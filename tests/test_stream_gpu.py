"""GPU: the streaming submission path (svh_elas_stream_*, include/svh.h) -- pairs pushed one at a
time, as the reference's producer hands them over (stereomapper/readfromfilesthread.cpp:63,104 ->
stereothread.cpp:68-176), results popped in order.  Every map is compared with the reference's own
output for its pair (tests/golden/*.npz, made from oracle/_ref)."""
import os
import threading

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

CROPS = [("urban1_robotics", "urban1_1242x375"), ("urban2_kitti", "urban2_1242x375"),
         ("urban3_kitti", "urban3_1242x375"), ("urban4_kitti", "urban4_1242x375")]


@pytest.fixture(scope="module")
def svhip():
    import svhip as S
    S.lib()
    assert S.device_count() > 0, "no HIP device: the product has no CPU fallback"
    return S


@pytest.fixture(scope="module")
def crops():
    out = []
    for npz, img in CROPS:
        z = np.load(os.path.join(H.GOLDEN, npz + ".npz"))
        l, r = H.golden_pair(img)
        out.append((l, r, z["d1"].reshape(l.shape), z["d2"].reshape(l.shape)))
    return out


def test_frames_one_at_a_time_in_order(svhip, crops, capfd):
    """70 frames pushed from host memory with a small depth (back-pressure), a flat pair among them:
    tickets come back in order, every map equals the reference's, the flat pair reports status 1
    with the reference's message and leaves its outputs untouched"""
    e = svhip.Elas(H.robotics())
    h, w = crops[0][0].shape
    s = e.stream(w, h, depth=9)
    flat = np.full((h, w), 90, np.uint8)
    n, bad = 70, {13, 40}
    D = [(np.full((h, w), -7.0, np.float32), np.full((h, w), -7.0, np.float32)) for _ in range(n)]
    popped = []

    def drain(upto):
        while len(popped) < upto:
            r = s.pop()
            assert r is not None
            popped.append(r)

    for i in range(n):
        l, r = (flat, flat) if i in bad else crops[i % 4][:2]
        assert s.push(l, r, D[i][0], D[i][1]) == i
        if i >= 8:
            drain(i - 7)          # never more than 9 in flight: push would block otherwise
    drain(n)
    assert s.pop() is None        # SVH_ERR_EMPTY
    s.close()
    assert [t for t, _ in popped] == list(range(n))
    for i, (t, st) in enumerate(popped):
        if i in bad:
            assert st == 1
            assert (D[i][0] == -7.0).all() and (D[i][1] == -7.0).all()
        else:
            assert st == 0
            assert np.array_equal(D[i][0], crops[i % 4][2]) and np.array_equal(D[i][1], crops[i % 4][3]), i
    assert capfd.readouterr().out.count("ERROR: Need at least 3 support points!") == len(bad)


def test_two_producers_one_consumer(svhip, crops):
    """pushes interleaved from two threads while a third pops: the ticket a push returns names the
    pair, results arrive in ticket order, every map equals the reference's"""
    e = svhip.Elas(H.robotics())
    h, w = crops[0][0].shape
    s = e.stream(w, h)
    per = 48
    which = {}                     # ticket -> (crop, D1, D2)
    lock = threading.Lock()

    def producer(k):
        for i in range(per):
            c = (2 * i + k) % 4
            D1 = np.zeros((h, w), np.float32)
            D2 = np.zeros((h, w), np.float32)
            with lock:             # (push + bookkeeping as one step so that `which` is complete when popped)
                t = s.push(crops[c][0], crops[c][1], D1, D2)
                which[t] = (c, D1, D2)

    got = []

    def consumer():
        while len(got) < 2 * per:
            try:
                r = s.pop(timeout_ms=20)
            except svhip.SvhTimeout:
                continue
            if r is not None:
                got.append(r)

    th = [threading.Thread(target=producer, args=(k,)) for k in range(2)] + [threading.Thread(target=consumer)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    s.close()
    assert [t for t, _ in got] == list(range(2 * per))
    for t, st in got:
        c, D1, D2 = which[t]
        assert st == 0
        assert np.array_equal(D1, crops[c][2]) and np.array_equal(D2, crops[c][3]), (t, c)


def test_device_resident_stream_equals_batch(svhip, crops):
    """frames and maps resident in HBM, pushed one by one (consecutive slices of one tensor share a
    launch; a jump in the addresses starts a new one): maps identical to the reference's"""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")     # (torch's own copy of the runtime must not come up after libsvhip)
    h, w = crops[0][0].shape
    n = 37
    I1 = np.ascontiguousarray(np.stack([crops[i % 4][0] for i in range(n)]))
    I2 = np.ascontiguousarray(np.stack([crops[i % 4][1] for i in range(n)]))
    Z = np.zeros((n, h, w), np.float32)

    def to_device(a):
        p = C.c_void_p()
        assert hip.hipMalloc(C.byref(p), C.c_size_t(a.nbytes)) == 0
        assert hip.hipMemcpy(p, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes), 1) == 0   # H2D
        return p
    dI1, dI2, dD1, dD2 = to_device(I1), to_device(I2), to_device(Z), to_device(Z)
    try:
        e = svhip.Elas(H.robotics())
        s = e.stream(w, h)
        order = list(range(0, 20)) + list(range(30, 37)) + list(range(20, 30))   # two jumps
        for i in order:
            s.push_device(dI1.value + i * w * h, dI2.value + i * w * h, dD1.value + i * w * h * 4,
                          dD2.value + i * w * h * 4)
        # an Elas::process call while the stream holds its lanes must not wait for the stream
        rc, A1, A2 = svhip.Elas(H.robotics()).process(crops[2][0], crops[2][1])
        assert rc == 0 and np.array_equal(A1, crops[2][2])
        for k in range(n):
            t, st = s.pop()
            assert t == k and st == 0
        s.close()
        d1, d2 = np.empty_like(Z), np.empty_like(Z)
        assert hip.hipMemcpy(C.c_void_p(d1.ctypes.data), dD1, C.c_size_t(d1.nbytes), 2) == 0   # D2H
        assert hip.hipMemcpy(C.c_void_p(d2.ctypes.data), dD2, C.c_size_t(d2.nbytes), 2) == 0
    finally:
        for p in (dI1, dI2, dD1, dD2):
            hip.hipFree(p)
    for i in range(n):
        assert np.array_equal(d1[i], crops[i % 4][2]) and np.array_equal(d2[i], crops[i % 4][3]), i


def test_pop_timeout_and_flush(svhip, crops):
    """a partially filled group starts on flush() (or when pop has nothing older to wait for);
    pop with a timeout returns SVH_ERR_TIMEOUT as an error, not a result"""
    e = svhip.Elas(H.robotics())
    h, w = crops[0][0].shape
    s = e.stream(w, h)
    D1, D2 = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    s.push(crops[1][0], crops[1][1], D1, D2)
    s.flush()
    t, st = s.pop(timeout_ms=5000)
    assert (t, st) == (0, 0) and np.array_equal(D1, crops[1][2])
    E1, E2 = np.zeros((h, w), np.float32), np.zeros((h, w), np.float32)
    s.push(crops[3][0], crops[3][1], E1, E2)
    with pytest.raises(svhip.SvhError):
        s.pop(timeout_ms=0)       # not done within 0 ms (the pop started it)
    t, st = s.pop()
    assert (t, st) == (1, 0) and np.array_equal(E2, crops[3][3])
    s.close()


@pytest.mark.parametrize("pinned", [True, False])
def test_host_frame_ring_through_push_n_equals_the_device_path(svhip, crops, pinned):
    """The reference's ownership contract as a stream (elas.cpp:40-56: host images in, host maps out;
    readfromfilesthread.cpp:25-112: a producer with decoded frames): a 100-frame sequence from ONE host array
    (consecutive frames: the groups' copies are strided, one per camera and group), handed over in rings of 25 through
    svh_elas_stream_push_n while a consumer pops in order.  Pinned (hipHostMalloc) and pageable frames; every map
    equals the reference's golden map of its crop, and the maps of the device-resident entry for the same frames."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    h, w = crops[0][0].shape
    n, ring = 100, 25

    def host_array(shape, dtype):
        if not pinned:
            return np.zeros(shape, dtype), None
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        assert hip.hipHostMalloc(C.byref(p), C.c_size_t(nbytes), 0) == 0
        buf = (C.c_uint8 * nbytes).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype).reshape(shape), p
    I1, p1 = host_array((n, h, w), np.uint8)
    I2, p2 = host_array((n, h, w), np.uint8)
    D1, p3 = host_array((n, h, w), np.float32)
    D2, p4 = host_array((n, h, w), np.float32)
    try:
        for i in range(n):
            I1[i], I2[i] = crops[(i * 7) % 4][:2]
        D1[:] = -3.0
        D2[:] = -3.0
        e = svhip.Elas(H.robotics())
        s = e.stream(w, h)
        st = []
        t = threading.Thread(target=lambda: st.extend(s.pop_n(n)))
        t.start()
        for r0 in range(0, n, ring):
            first = s.push_n(I1[r0:r0 + ring], I2[r0:r0 + ring], D1[r0:r0 + ring], D2[r0:r0 + ring])
            assert first == r0
        s.flush()
        t.join(120)
        assert not t.is_alive() and st == [0] * n
        s.close()
        for i in range(n):
            c = crops[(i * 7) % 4]
            assert np.array_equal(D1[i], c[2]) and np.array_equal(D2[i], c[3]), i
        # the device-resident batch entry on the same frames
        stb, B1, B2 = e.process_batch(I1[:12].copy(), I2[:12].copy())
        assert stb == [0] * 12 and np.array_equal(B1, D1[:12]) and np.array_equal(B2, D2[:12])
    finally:
        del I1, I2, D1, D2
        for p in (p1, p2, p3, p4):
            if p is not None:
                hip.hipHostFree(p)

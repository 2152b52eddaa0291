"""GPU tests of the lockstep entries (svh_matcher_push_back_batch, svh_matcher_match_features_batch,
svh_vo_process_batch): K objects driven as one launch per kernel must give, object by object, exactly what
K separate calls give -- feature tables, match indices and coordinates, motion, inliers: bit-exact.  The
unbatched calls are themselves pinned to the reference by test_matcher_gpu.py / test_vo_gpu.py; one case
here also checks a batched object against the golden reference output directly."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu

LIBC = C.CDLL(None)


def quad():
    return [H.read_pgm(os.path.join(H.GOLDEN, "viso_%s.pgm" % k)) for k in ("I1p", "I2p", "I1c", "I2c")]


def variant(im, k):
    """object k's sequence: the quad shifted by 3k columns (stereo geometry kept), every other one mirrored
    top-down, so that the K objects carry different feature counts"""
    out = [np.roll(a, 3 * k, axis=1) for a in im]
    if k % 2:
        out = [np.ascontiguousarray(a[::-1]) for a in out]
    return out


def plain_matcher(prm):
    m = H.ProductMatcher(prm)
    m.lib.svh_matcher_set_taps(C.c_void_p(m.h), 0)   # taps keep every stage and force the one-by-one path
    return m


def same_matcher_state(a, b, what):
    for tb in range(8):
        x, y = a.features(tb), b.features(tb)
        assert x.shape == y.shape and (x == y).all(), (what, "table", tb)
    x, y = a.matches(), b.matches()
    assert len(x) == len(y), (what, len(x), len(y))
    assert (x == y).all(), what


@pytest.mark.parametrize("kw,method,with_tr", [
    ({}, 2, False), ({}, 2, True), ({"half_resolution": 0}, 2, False), ({"multi_stage": 0}, 2, False),
    ({"refinement": 2}, 2, True), ({"refinement": 0}, 0, False), ({}, 1, False),
    ({"nms_n": 5, "nms_tau": 30, "match_binsize": 40}, 2, False), ({"refinement": 2, "half_resolution": 0}, 1, False),
])
def test_matcher_batch_equals_separate_calls(kw, method, with_tr):
    K = 5
    prm = H.matcher_defaults(**kw)
    prm.f, prm.cu, prm.cv, prm.base = 645.24, 635.96, 194.13, 0.5707
    im = quad()
    seqs = [variant(im, k) for k in range(K)]
    tr = None
    if with_tr:
        T = np.eye(4)
        T[2, 3] = -0.8
        T[0, 3] = 0.02
        tr = [T + 0.001 * k * np.eye(4)[[1, 0, 2, 3]] for k in range(K)]
    one = [plain_matcher(prm) for _ in range(K)]
    for k, m in enumerate(one):
        m.push_back(seqs[k][0], seqs[k][1])
        m.push_back(seqs[k][2], seqs[k][3])
        m.match(method, None if tr is None else tr[k])
    bat = [plain_matcher(prm) for _ in range(K)]
    H.product_matcher_batch(bat, [s[0] for s in seqs], [s[1] for s in seqs], None)
    H.product_matcher_batch(bat, [s[2] for s in seqs], [s[3] for s in seqs], method, tr)
    for k in range(K):
        assert len(one[k].matches()) > 50
        same_matcher_state(one[k], bat[k], ("object", k))
    # a second frame through the ring buffer, replace = 1 on top, then once more
    for rep in (True, False):
        for k, m in enumerate(one):
            m.push_back(seqs[k][0], seqs[k][1], replace=rep)
            m.match(method, None if tr is None else tr[k])
        H.product_matcher_batch(bat, [s[0] for s in seqs], [s[1] for s in seqs], method, tr, replace=rep)
        for k in range(K):
            same_matcher_state(one[k], bat[k], ("object", k, "replace", rep))


def test_matcher_batch_object_equals_golden_reference_output():
    z = np.load(os.path.join(H.GOLDEN, "viso_quad_default.npz"))
    prm = H.MatcherParams.from_buffer_copy(z["params"].tobytes())
    im = quad()
    K = 3
    seqs = [variant(im, k) for k in range(K)]
    seqs[1] = im                                   # object 1 carries the golden quad itself
    bat = [plain_matcher(prm) for _ in range(K)]
    H.product_matcher_batch(bat, [s[0] for s in seqs], [s[1] for s in seqs], None)
    H.product_matcher_batch(bat, [s[2] for s in seqs], [s[3] for s in seqs], int(z["method"]))
    got, want = bat[1].matches(), z["dense"]
    assert len(got) == len(want)
    for f in ("i1p", "i2p", "i1c", "i2c"):
        assert np.array_equal(got[f], want[f]), f
    assert (got == want).all()
    for tb in range(8):
        assert np.array_equal(bat[1].features(tb), z["table_" + H.M_TABLES[tb]])


def test_matcher_batch_with_objects_that_cannot_match_or_differ():
    """an object without a previous frame returns silently (matcher.cpp:216-259) and keeps no matches while the
    others run in lockstep; objects with other parameters send the whole call down the one-by-one path"""
    prm = H.matcher_defaults()
    im = quad()
    K = 4
    seqs = [variant(im, k) for k in range(K)]
    bat = [plain_matcher(prm) for _ in range(K)]
    H.product_matcher_batch(bat[:3], [s[0] for s in seqs[:3]], [s[1] for s in seqs[:3]], None)
    H.product_matcher_batch(bat, [s[2] for s in seqs], [s[3] for s in seqs], 2)   # object 3: first frame
    assert len(bat[3].matches()) == 0
    for k in range(3):
        m = plain_matcher(prm)
        m.push_back(seqs[k][0], seqs[k][1])
        m.push_back(seqs[k][2], seqs[k][3])
        m.match(2)
        same_matcher_state(m, bat[k], ("object", k))
    # single-image flow (I2 = NULL) in lockstep
    flow = [plain_matcher(prm) for _ in range(3)]
    H.product_matcher_batch(flow, [s[0] for s in seqs[:3]], None, None)
    H.product_matcher_batch(flow, [s[2] for s in seqs[:3]], None, 0)
    for k in range(3):
        m = plain_matcher(prm)
        m.push_back(seqs[k][0])
        m.push_back(seqs[k][2])
        m.match(0)
        same_matcher_state(m, flow[k], ("flow", k))
    # mixed parameters
    mixed = [plain_matcher(prm), plain_matcher(H.matcher_defaults(nms_n=4)), plain_matcher(prm)]
    H.product_matcher_batch(mixed, [s[0] for s in seqs[:3]], [s[1] for s in seqs[:3]], None)
    H.product_matcher_batch(mixed, [s[2] for s in seqs[:3]], [s[3] for s in seqs[:3]], 2)
    m = plain_matcher(H.matcher_defaults(nms_n=4))
    m.push_back(seqs[1][0], seqs[1][1])
    m.push_back(seqs[1][2], seqs[1][3])
    m.match(2)
    same_matcher_state(m, mixed[1], "mixed")
    # argument errors
    import svhip as S
    with pytest.raises(S.SvhError):
        H.product_matcher_batch([bat[0], bat[0]], [seqs[0][0]] * 2, [seqs[0][1]] * 2, None)


def run_vo(vos, seqs, frames, batched):
    LIBC.srand(7)
    log = []
    for i in range(frames):
        a = [s[0] if i % 2 == 0 else s[2] for s in seqs]
        b = [s[1] if i % 2 == 0 else s[3] for s in seqs]
        if batched:
            _, ok = H.product_vo_process_batch(vos, a, b)
            ok = list(ok)
        else:
            ok = [vo.process(a[k], b[k]) for k, vo in enumerate(vos)]
        log.append((ok, [vo.motion().copy() for vo in vos], [vo.inliers().copy() for vo in vos],
                    [vo.matches().copy() for vo in vos]))
    return log


@pytest.mark.parametrize("kw", [{}, {"ransac_iters": 64, "reweighting": 0}])
def test_vo_batch_equals_loop_of_process_calls(kw):
    """same srand, same draw order: the lockstep call reproduces the loop bit for bit (bootstrap frames
    included, which the batch entry runs one by one)"""
    K, frames = 6, 7
    prm = H.vo_defaults(**kw)
    im = quad()
    seqs = [variant(im, k) for k in range(K)]
    loop = run_vo([H.ProductVo(prm) for _ in range(K)], seqs, frames, False)
    bat = run_vo([H.ProductVo(prm) for _ in range(K)], seqs, frames, True)
    assert sum(sum(o == 1 for o in f[0]) for f in loop) >= K * (frames - 2)
    for i in range(frames):
        assert loop[i][0] == bat[i][0], ("return values", i)
        for k in range(K):
            assert np.array_equal(loop[i][1][k], bat[i][1][k]), ("motion", i, k)
            assert np.array_equal(loop[i][2][k], bat[i][2][k]), ("inliers", i, k)
            assert (loop[i][3][k] == bat[i][3][k]).all() and len(loop[i][3][k]) == len(bat[i][3][k]), ("matches", i, k)


def test_two_threads_each_driving_a_batch():
    """two host threads, each with its own K objects (own recorder, shared helper pool): match indices equal
    the single-threaded run"""
    prm = H.matcher_defaults()
    im = quad()
    K = 4
    seqs = [variant(im, k) for k in range(2 * K)]
    want = []
    for k in range(2 * K):
        m = plain_matcher(prm)
        m.push_back(seqs[k][0], seqs[k][1])
        m.push_back(seqs[k][2], seqs[k][3])
        m.match(2)
        want.append(m.matches())
    groups = [[plain_matcher(prm) for _ in range(K)] for _ in range(2)]
    errs = []

    def work(g):
        try:
            sq = seqs[g * K:(g + 1) * K]
            for _ in range(3):
                H.product_matcher_batch(groups[g], [s[0] for s in sq], [s[1] for s in sq], None)
                H.product_matcher_batch(groups[g], [s[2] for s in sq], [s[3] for s in sq], 2)
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(g,)) for g in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for g in range(2):
        for k in range(K):
            got = groups[g][k].matches()
            assert len(got) == len(want[g * K + k]) and (got == want[g * K + k]).all(), (g, k)


def test_private_rand_objects_in_a_batch_equal_objects_alone_in_a_process():
    """with private streams (svh_vo_set_private_rand, seed 0 = the reference constructor's srand(0)) every object
    of a lockstep call reproduces the run of ONE object that has libc rand() to itself -- which is the reference's
    situation (one VisualOdometryStereo per process) -- although K objects run interleaved here"""
    K, frames = 5, 6
    prm = H.vo_defaults()
    im = quad()
    seqs = [variant(im, k) for k in range(K)]
    alone = []
    for k in range(K):
        vo = H.ProductVo(prm)              # its constructor calls srand(0); nobody else draws meanwhile
        log = []
        for i in range(frames):
            s = seqs[k]
            ok = vo.process(s[0] if i % 2 == 0 else s[2], s[1] if i % 2 == 0 else s[3])
            log.append((ok, vo.motion().copy(), vo.inliers().copy(), vo.matches().copy()))
        alone.append(log)
        del vo
    vos = [H.ProductVo(prm, private_rand=0) for _ in range(K)]
    LIBC.srand(99)                          # the process-wide stream is not what they draw from
    for i in range(frames):
        a = [s[0] if i % 2 == 0 else s[2] for s in seqs]
        b = [s[1] if i % 2 == 0 else s[3] for s in seqs]
        _, ok = H.product_vo_process_batch(vos, a, b)
        for k in range(K):
            w = alone[k][i]
            assert int(ok[k]) == w[0], (i, k)
            assert np.array_equal(vos[k].motion(), w[1]), ("motion", i, k)
            assert np.array_equal(vos[k].inliers(), w[2]), ("inliers", i, k)
            got = vos[k].matches()
            assert len(got) == len(w[3]) and (got == w[3]).all(), ("matches", i, k)


@pytest.mark.parametrize("K", [1, 5])
def test_prefetched_frames_equal_plain_calls(K):
    """svh_matcher_prefetch_batch: frame t+1 is packed, uploaded and its features computed while frame t is matched;
    feature tables and matches of every frame equal those of plain pushBack / matchFeatures calls (K = 1: the
    one-by-one path of the entry; replace = 1 on one frame)"""
    prm = H.matcher_defaults()
    im = quad()
    seqs = [variant(im, k) for k in range(K)]
    frames = [(0, 1), (2, 3), (0, 1), (2, 3), (0, 1)]
    replace = [False, False, False, True, False]
    plain = [plain_matcher(prm) for _ in range(K)]
    want = []
    for f, (a, b) in enumerate(frames):
        for k, m in enumerate(plain):
            m.push_back(seqs[k][a], seqs[k][b], replace=replace[f])
            m.match(2)
        want.append([(m.matches().copy(), [m.features(tb).copy() for tb in range(8)]) for m in plain])
    pre = [plain_matcher(prm) for _ in range(K)]
    shape = seqs[0][0].shape
    H.product_matcher_prefetch(pre, [s[frames[0][0]] for s in seqs], [s[frames[0][1]] for s in seqs])
    for f in range(len(frames)):
        H.product_matcher_take_prefetched(pre, shape, replace=replace[f])
        if f + 1 < len(frames):   # the next frame goes out BEFORE this one is matched
            a, b = frames[f + 1]
            H.product_matcher_prefetch(pre, [s[a] for s in seqs], [s[b] for s in seqs])
        H.product_matcher_batch(pre, None, None, 2, push=False)
        for k in range(K):
            got = pre[k].matches()
            assert len(got) == len(want[f][k][0]) and (got == want[f][k][0]).all(), (f, k)
            for tb in range(8):
                assert np.array_equal(pre[k].features(tb), want[f][k][1][tb]), (f, k, tb)
    # misuse: two prefetches in a row; images while a prefetched frame is pending; nothing to take
    import svhip as S
    H.product_matcher_prefetch(pre, [s[0] for s in seqs], [s[1] for s in seqs])
    with pytest.raises(S.SvhError):
        H.product_matcher_prefetch(pre, [s[0] for s in seqs], [s[1] for s in seqs])
    with pytest.raises(S.SvhError):
        H.product_matcher_batch(pre, [s[0] for s in seqs], [s[1] for s in seqs], None)
    H.product_matcher_take_prefetched(pre, shape)
    with pytest.raises(S.SvhError):
        H.product_matcher_take_prefetched(pre, shape)


@pytest.mark.parametrize("private,K", [(True, 4), (False, 4), (True, 1)])
def test_vo_pipelined_loop_equals_plain_loop(private, K):
    """svh_vo_prefetch_batch + svh_vo_process_next_batch (frame t+1 handed over while frame t is matched) give the
    return values, motions, inliers and matches of svh_vo_process_batch with the images passed directly --
    bootstrap frames (one-by-one path) included; with libc rand() the draw order is the same in both loops;
    K = 1: a single object through the same entries"""
    frames = 8
    prm = H.vo_defaults()
    im = quad()
    seqs = [variant(im, k) for k in range(K)]
    pick = lambda i: ([s[0] if i % 2 == 0 else s[2] for s in seqs], [s[1] if i % 2 == 0 else s[3] for s in seqs])
    seed = 0 if private else None
    plain = [H.ProductVo(prm, private_rand=seed) for _ in range(K)]
    LIBC.srand(11)
    want = []
    for i in range(frames):
        _, ok = H.product_vo_process_batch(plain, *pick(i))
        want.append((list(ok), [v.motion().copy() for v in plain], [v.inliers().copy() for v in plain],
                     [v.matches().copy() for v in plain]))
    assert sum(sum(o == 1 for o in w[0]) for w in want) >= K * (frames - 2)
    pipe = [H.ProductVo(prm, private_rand=seed) for _ in range(K)]
    LIBC.srand(11)
    shape = seqs[0][0].shape
    H.product_vo_prefetch_batch(pipe, *pick(0))
    for i in range(frames):
        nxt = pick(i + 1) if i + 1 < frames else (None, None)
        _, ok = H.product_vo_process_next_batch(pipe, nxt[0], nxt[1], shape)
        assert list(ok) == want[i][0], i
        for k in range(K):
            assert np.array_equal(pipe[k].motion(), want[i][1][k]), ("motion", i, k)
            assert np.array_equal(pipe[k].inliers(), want[i][2][k]), ("inliers", i, k)
            got = pipe[k].matches()
            assert len(got) == len(want[i][3][k]) and (got == want[i][3][k]).all(), ("matches", i, k)


def test_two_threads_pipelined_vo_loops_equal_plain_loops():
    """two calling threads, each with K objects in the pipelined loop (svh_vo_prefetch_batch +
    svh_vo_process_next_batch): the packing of the next frames (prefetch thread) and the outlier votes of both
    callers are jobs in flight on the one helper pool at the same time -- every object's return values, motions,
    inliers and matches equal the plain single-threaded loop (private random streams: the draw order of a shared
    rand() would depend on the interleaving)"""
    frames, K, T = 6, 4, 2
    prm = H.vo_defaults()
    im = quad()
    seqs = [variant(im, k) for k in range(T * K)]

    def pick(g, i):
        sq = seqs[g * K:(g + 1) * K]
        return ([s[0] if i % 2 == 0 else s[2] for s in sq], [s[1] if i % 2 == 0 else s[3] for s in sq])
    want = []
    for g in range(T):
        plain = [H.ProductVo(prm, private_rand=0) for _ in range(K)]
        rows = []
        for i in range(frames):
            _, ok = H.product_vo_process_batch(plain, *pick(g, i))
            rows.append((list(ok), [v.motion().copy() for v in plain], [v.inliers().copy() for v in plain],
                         [v.matches().copy() for v in plain]))
        want.append(rows)
    shape = seqs[0][0].shape
    pipes = [[H.ProductVo(prm, private_rand=0) for _ in range(K)] for _ in range(T)]
    got = [[] for _ in range(T)]
    errs = []

    def work(g):
        try:
            H.product_vo_prefetch_batch(pipes[g], *pick(g, 0))
            for i in range(frames):
                nxt = pick(g, i + 1) if i + 1 < frames else (None, None)
                _, ok = H.product_vo_process_next_batch(pipes[g], nxt[0], nxt[1], shape)
                got[g].append((list(ok), [v.motion().copy() for v in pipes[g]], [v.inliers().copy() for v in pipes[g]],
                               [v.matches().copy() for v in pipes[g]]))
        except Exception as e:   # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=work, args=(g,)) for g in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for g in range(T):
        assert sum(sum(o == 1 for o in w[0]) for w in want[g]) >= K * (frames - 2)
        for i in range(frames):
            assert got[g][i][0] == want[g][i][0], (g, i)
            for k in range(K):
                assert np.array_equal(got[g][i][1][k], want[g][i][1][k]), ("motion", g, i, k)
                assert np.array_equal(got[g][i][2][k], want[g][i][2][k]), ("inliers", g, i, k)
                a, b = got[g][i][3][k], want[g][i][3][k]
                assert len(a) == len(b) and (a == b).all(), ("matches", g, i, k)


@pytest.mark.parametrize("seed", range(300, 312))
def test_param_fuzz_lockstep_equals_separate_calls(seed):
    """random points of Matcher::parameters x method x ragged crop x predicted motion (helpers.fuzz_matcher_case,
    the generator of the oracle fuzz in test_matcher_gpu.py), K objects with different crops of the same size:
    the lockstep entries (plain and with the frame handed over early) equal the single calls, tables and matches"""
    prm, method, crop, tr = H.fuzz_matcher_case(seed)
    rng = np.random.default_rng(seed)
    K = int(rng.integers(2, 7))
    im = quad()
    h, w = crop[0].stop - crop[0].start, crop[1].stop - crop[1].start
    seqs = []
    for k in range(K):
        x0, y0 = int(rng.integers(0, im[0].shape[1] - w)), int(rng.integers(0, im[0].shape[0] - h))
        seqs.append([np.ascontiguousarray(a[y0:y0 + h, x0:x0 + w]) for a in im])
    stereo = method != 0 or bool(rng.integers(0, 2))     # flow may run on single images
    second = (lambda s, j: s[j]) if stereo else (lambda s, j: None)
    one = [plain_matcher(prm) for _ in range(K)]
    for k, m in enumerate(one):
        m.push_back(seqs[k][0], second(seqs[k], 1))
        m.push_back(seqs[k][2], second(seqs[k], 3))
        m.match(method, tr)
    I2 = lambda j: [s[j] for s in seqs] if stereo else None
    trs = None if tr is None else [tr] * K
    bat = [plain_matcher(prm) for _ in range(K)]
    H.product_matcher_batch(bat, [s[0] for s in seqs], I2(1), None)
    H.product_matcher_batch(bat, [s[2] for s in seqs], I2(3), method, trs)
    pre = [plain_matcher(prm) for _ in range(K)]
    H.product_matcher_prefetch(pre, [s[0] for s in seqs], I2(1))
    H.product_matcher_take_prefetched(pre, (h, w))
    H.product_matcher_prefetch(pre, [s[2] for s in seqs], I2(3))
    H.product_matcher_take_prefetched(pre, (h, w))
    H.product_matcher_batch(pre, None, None, method, trs, push=False)
    for k in range(K):
        same_matcher_state(one[k], bat[k], ("lockstep", seed, k))
        same_matcher_state(one[k], pre[k], ("handed over early", seed, k))


def test_lockstep_results_do_not_depend_on_batch_composition():
    """size-independent properties of the lockstep entries: an object's tables and matches are the same whatever
    its position in the batch, whoever its neighbours are and however large K is (1344x391, K = 2 .. 24), and
    pushing the same frame again with replace = 1 changes nothing"""
    prm = H.matcher_defaults()
    im = quad()
    rng = np.random.default_rng(5)
    pool = [variant(im, k) for k in range(24)]
    ref = {}
    for K in (24, 2, 7):
        ids = [int(i) for i in rng.permutation(24)[:K]]
        if 0 not in ids:
            ids[0] = 0
        ms = [plain_matcher(prm) for _ in ids]
        H.product_matcher_batch(ms, [pool[i][0] for i in ids], [pool[i][1] for i in ids], None)
        H.product_matcher_batch(ms, [pool[i][2] for i in ids], [pool[i][3] for i in ids], 2)
        for m, i in zip(ms, ids):
            got = (m.matches().copy(), [m.features(tb).copy() for tb in range(8)])
            if i in ref:
                assert len(got[0]) == len(ref[i][0]) and (got[0] == ref[i][0]).all(), (K, i)
                for tb in range(8):
                    assert np.array_equal(got[1][tb], ref[i][1][tb]), (K, i, tb)
            else:
                ref[i] = got
        # the same current frame once more, replacing it: identical state
        H.product_matcher_batch(ms, [pool[i][2] for i in ids], [pool[i][3] for i in ids], 2, replace=True)
        for m, i in zip(ms, ids):
            got = m.matches()
            assert len(got) == len(ref[i][0]) and (got == ref[i][0]).all(), ("replace", K, i)
    assert len(ref[0][0]) > 1000


def test_cxx_classes_lockstep_loops_agree():
    """tests/cxx/vo_lockstep.cpp: demo.cpp's frame loop for K VisualOdometryStereo objects of include/viso_stereo.h --
    K process() calls, one processBatch(), and the pipelined prefetchBatch / processNextBatch loop -- compiled with
    g++ against include/ only; the three loops must report the same return values, motions and inlier sets"""
    exe = os.path.join(H.ROOT, "tests", "cxx", "vo_lockstep")
    subprocess.check_call(["make", "-C", os.path.join(H.ROOT, "tests", "cxx"), "vo_lockstep"])
    args = [os.path.join(H.GOLDEN, "viso_%s.pgm" % k) for k in ("I1p", "I2p", "I1c", "I2c")]
    r = subprocess.run([exe] + args + ["6", "7"], capture_output=True, text=True)
    assert r.returncode == 0 and "vo_lockstep: OK" in r.stdout, r.stdout + r.stderr

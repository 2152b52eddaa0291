"""the KITTI raw reader (include/svh_kitti.h) against the numpy oracle
(oracle/kitti_oracle.py) and against PIL-decoded pixels of one of the reference's own PNGs.
Host-only code: every test runs without a GPU."""
import os
import struct
import sys
import zlib

import numpy as np
import pytest

import helpers as H

sys.path.insert(0, os.path.join(H.ROOT, "oracle"))
import kitti_oracle as KO  # noqa: E402


@pytest.fixture(scope="module")
def K():
    from svhip import kitti
    return kitti


def chunk(tag, body):
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)


def write_png(path, samples, depth=8, colour=0, filters=None, idat_split=1, interlace=0, level=6):
    """samples: (h, w, channels) integers; every row filtered with filters[y % len]"""
    h, w, ch = samples.shape
    bps = depth // 8
    bpp = ch * bps
    if depth == 16:
        raw = samples.astype(">u2").tobytes()
    else:
        raw = samples.astype(np.uint8).tobytes()
    rows = np.frombuffer(raw, np.uint8).reshape(h, w * bpp).astype(np.int32)
    out = bytearray()
    filters = filters or [0]
    for y in range(h):
        f = filters[y % len(filters)]
        cur = rows[y]
        up = rows[y - 1] if y else np.zeros_like(cur)
        a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]])
        if f == 0:
            pred = 0
        elif f == 1:
            pred = a
        elif f == 2:
            pred = up
        elif f == 3:
            pred = (a + up) >> 1
        else:
            p = a + up - c
            pa, pb, pc = abs(p - a), abs(p - up), abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, up, c))
        out.append(f)
        out += ((cur - pred) & 255).astype(np.uint8).tobytes()
    z = zlib.compress(bytes(out), level)
    cuts = [len(z) * i // idat_split for i in range(idat_split + 1)]
    body = b"".join(chunk(b"IDAT", z[cuts[i]:cuts[i + 1]]) for i in range(idat_split))
    hdr = struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, interlace)
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", hdr) + chunk(b"tEXt", b"Comment\0test") + body
                 + chunk(b"IEND", b""))


def test_png_of_the_reference_decodes_like_libpng(K):
    """pin: one of the reference's PNG frames == the pixels PIL/libpng gave for it"""
    path = os.path.join(H.ROOT, "tests", "golden", "viso_I1c.png")
    want = H.read_pgm(os.path.join(H.ROOT, "tests", "golden", "viso_I1c.pgm"))
    assert np.array_equal(K.read_png_gray(path), want)
    assert np.array_equal(KO.png_read_gray(path), want)


@pytest.mark.parametrize("colour,ch", [(0, 1), (4, 2), (2, 3), (6, 4)])
@pytest.mark.parametrize("depth", [8, 16])
def test_png_variants_match_oracle(K, tmp_path, colour, ch, depth):
    rng = np.random.default_rng(colour * 10 + depth)
    h, w = 37, 53
    hi = 256 if depth == 8 else 65536
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = ((xx * 3 + yy * 5)[:, :, None] * (hi // 256) + rng.integers(0, hi // 16, (h, w, ch))) % hi
    noise = rng.integers(0, hi, (h, w, ch))
    for name, img, filters, split in (("smooth", smooth, [0, 1, 2, 3, 4], 1), ("noise", noise, [4, 3, 1], 3),
                                      ("paeth", smooth, [4], 2), ("avg", noise, [3], 1)):
        p = str(tmp_path / ("%s_%d_%d.png" % (name, colour, depth)))
        write_png(p, img, depth=depth, colour=colour, filters=filters, idat_split=split)
        got = K.read_png_gray(p)
        assert np.array_equal(got, KO.png_read_gray(p)), name
        top = (img >> (depth - 8)).astype(np.uint32)   # the byte an 8-bit reader keeps
        if ch <= 2:
            assert np.array_equal(got, top[:, :, 0])
        else:
            want = (top[:, :, 0] * 4899 + top[:, :, 1] * 9617 + top[:, :, 2] * 1868 + 8192) >> 14
            assert np.array_equal(got, want)


def test_png_errors(K, tmp_path):
    S = __import__("svhip")
    img = np.arange(20 * 30).reshape(20, 30, 1) % 256
    good = str(tmp_path / "good.png")
    write_png(good, img)
    assert np.array_equal(K.read_png_gray(good), img[:, :, 0])
    data = open(good, "rb").read()
    cases = {}
    cases["missing"] = (None, S.ERR_BAD_ARG)
    cases["not_png"] = (b"P5\n30 20\n255\n" + bytes(600), S.ERR_BAD_ARG)
    cases["truncated"] = (data[:len(data) // 2], S.ERR_BAD_ARG)
    flipped = bytearray(data)
    flipped[60] ^= 0x40          # inside the first chunk after IHDR: its CRC no longer matches
    cases["bad_crc"] = (bytes(flipped), S.ERR_BAD_ARG)
    for name, (blob, code) in cases.items():
        p = str(tmp_path / (name + ".png"))
        if blob is not None:
            open(p, "wb").write(blob)
        with pytest.raises(S.SvhError) as ei:
            K.read_png_gray(p)
        assert ei.value.code == code, name
    inter = str(tmp_path / "interlaced.png")
    write_png(inter, img, interlace=1)
    with pytest.raises(S.SvhError) as ei:
        K.read_png_gray(inter)
    assert ei.value.code == S.ERR_UNSUPPORTED
    # a buffer that is too small is refused, not overrun
    import ctypes as C
    w, h = C.c_int32(), C.c_int32()
    small = np.zeros(100, np.uint8)
    rc = S.lib().svh_png_read_gray(good.encode(), small.ctypes.data, small.size, C.byref(w), C.byref(h))
    assert rc == S.ERR_BAD_ARG and (w.value, h.value) == (30, 20) and not small.any()


def calib_text(rng, order=None, sep=" ", drop=None, extra_value=None):
    """a calib_cam_to_cam.txt in the KITTI layout with made-up numbers"""
    shapes = (("S", 2), ("K", 9), ("D", 5), ("R", 9), ("T", 3), ("S_rect", 2), ("R_rect", 9), ("P_rect", 12))
    lines = ["calib_time: 09-Jan-2012 13:57:47", "corner_dist: 9.950000e-02"]
    for i in range(4):
        for name, n in shapes:
            vals = rng.normal(0, 300, n)
            if name == "P_rect":
                vals[0] = 721.5377 + i
                vals[3] = -387.5744 * i
            key = "%s_0%d:" % (name, i)
            if key == drop:
                continue
            if key == extra_value:
                vals = np.append(vals, 1.0)
            lines.append(key + sep + sep.join("%.6e" % v for v in vals))
    if order is not None:
        lines = [lines[i] for i in order(len(lines))]
    return "\n".join(lines) + "\n"


def test_calibration_matches_oracle(K, tmp_path):
    S = __import__("svhip")
    rng = np.random.default_rng(5)
    for trial, (sep, shuffle) in enumerate(((" ", False), (" ", True), ("\t", True), (", ", False), ("; ", True))):
        text = calib_text(np.random.default_rng(trial), sep=sep,
                          order=(lambda n: rng.permutation(n)) if shuffle else None)
        p = str(tmp_path / ("calib%d.txt" % trial))
        open(p, "w").write(text)
        c = K.read_cam_to_cam(p)
        o = KO.read_cam_to_cam(p)
        assert o is not None
        for cam in range(4):
            for name in ("S", "K", "D", "R", "T", "S_rect", "R_rect", "P_rect"):
                assert np.array_equal(c.matrix(name, cam), o[(name, cam)]), (trial, name, cam)
        assert (c.f, c.cu, c.cv, c.base) == (o["f"], o["cu"], o["cv"], o["base"])
        assert c.corner_dist == float(np.float32(9.95e-2))      # values pass through a float
        assert c.calib_time == b"09-Jan-2012 13:57:47"
        assert c.base == pytest.approx(387.5744 / 722.5377, rel=1e-6)
    for kw in ({"drop": "R_rect_02:"}, {"extra_value": "P_rect_01:"}):
        p = str(tmp_path / "bad.txt")
        open(p, "w").write(calib_text(np.random.default_rng(0), **kw))
        assert KO.read_cam_to_cam(p) is None
        with pytest.raises(S.SvhError):
            K.read_cam_to_cam(p)
    with pytest.raises(S.SvhError):
        K.read_cam_to_cam(str(tmp_path / "absent.txt"))


def make_drive(root, frames, rng, unterminated=False, right_lines=None):
    stamps = []
    imgs = []
    for k in range(2):
        d = root / ("image_0%d" % k) / "data"
        d.mkdir(parents=True)
        lines = []
        n = frames if (k == 0 or right_lines is None) else right_lines
        for i in range(n):
            ns = int(rng.integers(0, 10 ** 9))
            lines.append("2011-09-26 13:%02d:%02d.%09d" % (2 + i // 60, (25 + i) % 60, ns))
        text = "\n".join(lines) + ("" if unterminated else "\n")
        (root / ("image_0%d" % k) / "timestamps.txt").write_text(text)
        stamps.append(lines)
        cam = []
        for i in range(frames):
            img = rng.integers(0, 256, (24, 40, 1))
            write_png(str(d / ("%010d.png" % i)), img, filters=[i % 5])
            cam.append(img[:, :, 0].astype(np.uint8))
        imgs.append(cam)
    return stamps, imgs


def test_sequence_playback(K, tmp_path):
    S = __import__("svhip")
    rng = np.random.default_rng(9)
    stamps, imgs = make_drive(tmp_path / "drive", 5, rng)
    seq = K.Sequence(tmp_path / "drive")
    assert len(seq) == 5
    got = list(seq)
    assert len(got) == 5
    for i, (a, b, (tl, tr)) in enumerate(got):
        assert np.array_equal(a, imgs[0][i]) and np.array_equal(b, imgs[1][i])
        for t, line in ((tl, stamps[0][i]), (tr, stamps[1][i])):
            sec, usec = KO.parse_stamp(line)
            assert t == sec + usec * 1e-6
    seq.close()
    # a last line without its newline is not a frame (the reference counts '\n')
    make_drive(tmp_path / "open_end", 3, rng, unterminated=True)
    assert len(K.Sequence(tmp_path / "open_end")) == 2
    # cameras that disagree on the frame count, or a directory that is not a drive, do not open
    make_drive(tmp_path / "uneven", 3, rng, right_lines=2)
    for bad in ("uneven", "nowhere"):
        with pytest.raises(S.SvhError):
            K.Sequence(tmp_path / bad)
    # a missing frame file is an error of that frame; playback goes on with the next one
    os.remove(tmp_path / "drive" / "image_01" / "data" / "0000000001.png")
    seq = K.Sequence(tmp_path / "drive")
    next(seq)
    with pytest.raises(S.SvhError):
        next(seq)
    a, b, _ = next(seq)
    assert np.array_equal(a, imgs[0][2]) and np.array_equal(b, imgs[1][2])


def test_sharded_load_covers_the_drive_once(K, tmp_path):
    """bench.py --workload sequence --kitti-dir: every rank decodes only its own contiguous share"""
    rng = np.random.default_rng(21)
    _, imgs = make_drive(tmp_path / "drive", 7, rng)
    seen = []
    for rank in range(3):
        a, b, lo, total = K.load_shard(tmp_path / "drive", rank, 3)
        assert total == 7 and lo == len(seen)
        for i in range(a.shape[0]):
            assert np.array_equal(a[i], imgs[0][lo + i]) and np.array_equal(b[i], imgs[1][lo + i])
            seen.append(lo + i)
    assert seen == list(range(7))


def test_png_with_absurd_header_is_refused(K, tmp_path):
    """IHDR dimensions are not trusted: a crafted 2^31-1 x 2^31-1 header returns an error code
    (no allocation of that size, no exception across the C boundary)"""
    S = __import__("svhip")
    for w, h in ((0x7fffffff, 0x7fffffff), (1 << 20, 1 << 20), (70000, 70000)):
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + \
            chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
        path = str(tmp_path / "huge.png")
        open(path, "wb").write(png)
        with pytest.raises(S.SvhError) as ei:
            K.read_png_gray(path)
        assert ei.value.code == S.ERR_BAD_ARG

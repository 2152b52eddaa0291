"""TEST INFRASTRUCTURE -- numpy restatement ("oracle") of the KITTI raw input formats.

Only tests/ may import this; the product (libsvhip.so: csrc/kitti_io.cpp) never does.

PARITY UNPINNED against the reference's own reader: stereomapper loads frames with OpenCV
(`cvLoadImage(..., CV_LOAD_IMAGE_GRAYSCALE)`, stereoimageiokitti.cpp:111) and lives in Qt
classes, neither of which exists in this image, so the reference path cannot be run here.
What IS pinned:
  * the PNG decode, against an independent decoder (PIL/libpng, the same libpng OpenCV wraps)
    on one of the reference's own PNG frames: tests/golden/viso_I1c.png must decode to
    tests/golden/viso_I1c.pgm, which tests/golden/make_goldens_viso.py wrote through PIL;
  * the text formats, against the reference's parsing rules restated below with their lines.
"""
import struct
import zlib

import numpy as np


# ---------------------------------------------------------------------------
# calib_cam_to_cam.txt  (stereomapper/calibiokitti.cpp)
# ---------------------------------------------------------------------------
def split_line(line):
    """calibiokitti.cpp:110-138: tokens end at blank, tab, comma, semicolon, newline"""
    out, cur = [], ""
    for ch in line:
        if ch in " \t,;\n\r":
            if cur:
                out.append(cur)
            cur = ""
            if ch == "\n":
                break
        else:
            cur += ch
    if cur:
        out.append(cur)
    return out


def read_matrix(lines, name, m, n):
    """calibiokitti.cpp:176-224: first line starting with `name`; values pass through a float"""
    for line in lines:
        tok = split_line(line)
        if tok and tok[0] == name:
            if len(tok) - 1 != m * n:
                return None
            return np.array([np.float32(t) for t in tok[1:]], np.float32).astype(np.float64).reshape(m, n)
    return None


def read_cam_to_cam(path):
    """calibiokitti.cpp:227-262 + the rig of stereothread.cpp:444-447"""
    lines = open(path).read().splitlines(True)
    out = {}
    shapes = (("S", 1, 2), ("K", 3, 3), ("D", 1, 5), ("R", 3, 3), ("T", 1, 3),
              ("S_rect", 1, 2), ("R_rect", 3, 3), ("P_rect", 3, 4))
    out["corner_dist"] = read_matrix(lines, "corner_dist:", 1, 1)
    ok = out["corner_dist"] is not None
    for i in range(4):
        for name, m, n in shapes:
            v = read_matrix(lines, "%s_0%d:" % (name, i), m, n)
            ok = ok and v is not None
            out[(name, i)] = v
    if not ok:
        return None
    P0, P1 = out[("P_rect", 0)], out[("P_rect", 1)]
    out["f"], out["cu"], out["cv"] = P0[0, 0], P0[0, 2], P0[1, 2]
    out["base"] = -P1[0, 3] / P1[0, 0]
    return out


# ---------------------------------------------------------------------------
# timestamps.txt  (stereomapper/stereoimageiokitti.cpp:100-105)
# ---------------------------------------------------------------------------
def parse_stamp(line):
    sec = int(line[11:13]) * 3600 + int(line[14:16]) * 60 + int(line[17:19])
    usec = int(line[20:29]) // 1000
    return sec, usec


# ---------------------------------------------------------------------------
# PNG -> 8-bit gray  (RFC 2083; colour -> gray with OpenCV's 14-bit fixed-point weights)
# ---------------------------------------------------------------------------
def png_read_gray(path):
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    at, idat, hdr = 8, b"", None
    while at < len(b):
        n, = struct.unpack(">I", b[at:at + 4])
        t = b[at + 4:at + 8]
        body = b[at + 8:at + 8 + n]
        if t == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif t == b"IDAT":
            idat += body
        at += 12 + n
    w, h, depth, colour, _, _, interlace = hdr
    assert interlace == 0 and depth in (8, 16)
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[colour]
    bpp = ch * depth // 8
    stride = bpp * w
    flat = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, stride + 1)
    raw = np.zeros((h, stride), np.int32)
    for y in range(h):
        f = int(flat[y, 0])
        row = flat[y, 1:].astype(np.int32)
        up = raw[y - 1] if y else np.zeros(stride, np.int32)
        cur = raw[y]
        if f == 0:
            cur[:] = row
        elif f == 2:
            cur[:] = (row + up) & 255
        else:
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                c = up[i - bpp] if i >= bpp else 0
                bb = up[i]
                if f == 1:
                    p = a
                elif f == 3:
                    p = (a + bb) >> 1
                else:
                    pp = a + bb - c
                    pa, pb, pc = abs(pp - a), abs(pp - bb), abs(pp - c)
                    p = a if (pa <= pb and pa <= pc) else (bb if pb <= pc else c)
                cur[i] = (row[i] + p) & 255
    px = raw.reshape(h, w, ch, depth // 8)[:, :, :, 0].astype(np.uint32)   # high byte of each sample
    if ch <= 2:
        return px[:, :, 0].astype(np.uint8)
    return ((px[:, :, 0] * 4899 + px[:, :, 1] * 9617 + px[:, :, 2] * 1868 + 8192) >> 14).astype(np.uint8)

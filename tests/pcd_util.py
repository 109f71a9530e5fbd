"""Minimal PCD writer/reader + PNG decoder for the CLI tests (test helpers)."""
import struct
import zlib

import numpy as np

from bev_amd import POINT_DTYPE

PACKED = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("intensity", "<f4"), ("row", "<u2"), ("col", "<u2"),
                   ("t", "<u4"), ("label", "<i2")])  # 26 bytes, what PCL writes for PointXYZIRCT


def _header(n, data, width=None, height=1):
    return (f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity row col t label\n"
            f"SIZE 4 4 4 4 2 2 4 2\nTYPE F F F F U U U I\nCOUNT 1 1 1 1 1 1 1 1\nWIDTH {n if width is None else width}\n"
            f"HEIGHT {height}\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA {data}\n").encode()


def to_packed(pts):
    out = np.empty(len(pts), PACKED)
    for k in PACKED.names:
        out[k] = pts[k]
    return out


def write_pcd_binary(path, pts, width=None, height=1):
    with open(path, "wb") as f:
        f.write(_header(len(pts), "binary", width, height))
        f.write(to_packed(pts).tobytes())


def write_pcd_ascii(path, pts):
    with open(path, "wb") as f:
        f.write(_header(len(pts), "ascii"))
        for p in pts:
            f.write((" ".join([repr(float(p["x"])), repr(float(p["y"])), repr(float(p["z"])),
                               repr(float(p["intensity"])), str(int(p["row"])), str(int(p["col"])), str(int(p["t"])),
                               str(int(p["label"]))]) + "\n").encode())


def read_pcd_binary(path):
    raw = open(path, "rb").read()
    marker = b"DATA binary\n"
    i = raw.index(marker) + len(marker)
    head = raw[:i].decode()
    n = int([l for l in head.splitlines() if l.startswith("POINTS")][0].split()[1])
    packed = np.frombuffer(raw[i:i + 26 * n], PACKED)
    out = np.zeros(n, POINT_DTYPE)
    for k in PACKED.names:
        out[k] = packed[k]
    return head, out


def read_png_gray8(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(raw):
        ln, tag = struct.unpack(">I4s", raw[pos:pos + 8])
        body = raw[pos + 8:pos + 8 + ln]
        crc = struct.unpack(">I", raw[pos + 8 + ln:pos + 12 + ln])[0]
        assert zlib.crc32(tag + body) == crc
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 0)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + ln
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w + 1)
    assert (rows[:, 0] == 0).all()
    return rows[:, 1:]

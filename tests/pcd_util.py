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


def lzf_compress(data: bytes) -> bytes:
    """Small LZF encoder (liblzf format, as PCL's binary_compressed uses): greedy 3-byte hash matches
    within 8 KiB, literal runs of at most 32 bytes.  Good enough to exercise both token kinds."""
    out = bytearray()
    n = len(data)
    table = {}
    lit = bytearray()

    def flush():
        nonlocal lit
        i = 0
        while i < len(lit):
            run = lit[i:i + 32]
            out.append(len(run) - 1)
            out.extend(run)
            i += 32
        lit = bytearray()

    i = 0
    while i < n:
        m = None
        if i + 2 < n:
            key = data[i:i + 3]
            j = table.get(key)
            table[key] = i
            if j is not None and 0 < i - j <= 8191:
                ln = 3
                while i + ln < n and ln < 264 and data[j + ln] == data[i + ln]:
                    ln += 1
                m = (i - j - 1, ln)
        if m is None:
            lit.append(data[i])
            i += 1
            continue
        flush()
        off, ln = m
        l2 = ln - 2
        if l2 < 7:
            out.append((l2 << 5) | (off >> 8))
        else:
            out.append((7 << 5) | (off >> 8))
            out.append(l2 - 7)
        out.append(off & 0xFF)
        i += ln
    flush()
    return bytes(out)


def write_pcd_binary_compressed(path, pts):
    packed = to_packed(pts)
    soa = b"".join(np.ascontiguousarray(packed[k]).tobytes() for k in PACKED.names)  # field-by-field
    comp = lzf_compress(soa)
    with open(path, "wb") as f:
        f.write(_header(len(pts), "binary_compressed"))
        f.write(struct.pack("<II", len(comp), len(soa)))
        f.write(comp)

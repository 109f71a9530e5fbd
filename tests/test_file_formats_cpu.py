"""CPU: the on-disk formats of the CLI (host/FileFormats.cpp) without a GPU: PCD in all three DATA
kinds round-trips through the reader, the writer's output is what PCL writes for this point type (header restated from
memory: framing parity unpinned), malformed files are rejected instead of crashing, PNG files decode to the same pixels
with a standard inflate, the CSV text has the framing the reference's cv::format produces (250,656 bytes for 224 x 224)."""
import numpy as np
import pytest

import bev_amd
import hostcheck_lib as hc
import pcd_util
from bev_amd import synth

P = bev_amd.params_for_sensor("HDL_32E")
FIELDS = ("x", "y", "z", "intensity", "row", "col", "t", "label")


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(a[f], b[f], equal_nan=True) for f in FIELDS)


@pytest.mark.parametrize("kind", ["binary", "ascii", "binary_compressed", "binary_w0h0"])
def test_pcd_reader_round_trip(tmp_path, kind):
    pts = synth.adversarial(P, 3000, 11) if kind == "ascii" else synth.sweep(P, 7)
    path = tmp_path / "a.pcd"
    {"binary": pcd_util.write_pcd_binary, "ascii": pcd_util.write_pcd_ascii,
     "binary_compressed": pcd_util.write_pcd_binary_compressed,
     "binary_w0h0": lambda p_, x: pcd_util.write_pcd_binary(p_, x, width=0, height=0)}[kind](path, pts)
    rc, got, w, h = hc.pcd_load(path)
    assert rc == 0 and _same(got, pts)
    assert (got["_pad0"] == 0).all() if "_pad0" in got.dtype.names else True


def test_pcd_writer_matches_pcl_layout(tmp_path):
    pts = synth.sweep(P, 3)
    path = tmp_path / "out.pcd"
    assert hc.pcd_save(path, pts) == 0
    raw = path.read_bytes()
    head = (f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity row col t label\n"
            f"SIZE 4 4 4 4 2 2 4 2\nTYPE F F F F U U U I\nCOUNT 1 1 1 1 1 1 1 1\nWIDTH {len(pts)}\nHEIGHT 1\n"
            f"VIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(pts)}\nDATA binary\n").encode()
    assert raw[:len(head)] == head and len(raw) == len(head) + 26 * len(pts)   # 26-byte packed records
    assert raw[len(head):] == pcd_util.to_packed(pts).tobytes()
    rc, back, w, h = hc.pcd_load(path)
    assert rc == 0 and _same(back, pts) and (w, h) == (len(pts), 1)
    # empty cloud
    assert hc.pcd_save(tmp_path / "e.pcd", pts[:0]) == 0
    rc, back, _, _ = hc.pcd_load(tmp_path / "e.pcd")
    assert rc == 0 and len(back) == 0


def test_pcd_reader_other_field_layouts(tmp_path):
    """extra fields, other sizes / types / counts and a different field order (what a generic PCD may hold)"""
    n = 50
    rng = np.random.default_rng(0)
    rec = np.dtype([("label", "<i4"), ("normal", "<f4", (3,)), ("x", "<f8"), ("col", "<u1"), ("y", "<f4"), ("z", "<f4"),
                    ("row", "<u4")])
    a = np.zeros(n, rec)
    a["label"], a["x"], a["col"] = rng.integers(-5, 5, n), rng.normal(0, 10, n), rng.integers(0, 200, n)
    a["y"], a["z"], a["row"] = rng.normal(0, 10, n), rng.normal(0, 1, n), rng.integers(0, 31, n)
    head = (f"VERSION 0.7\nFIELDS label normal x col y z row\nSIZE 4 4 8 1 4 4 4\nTYPE I F F U F F U\n"
            f"COUNT 1 3 1 1 1 1 1\nWIDTH {n}\nHEIGHT 1\nPOINTS {n}\nDATA binary\n").encode()
    (tmp_path / "g.pcd").write_bytes(head + a.tobytes())
    rc, got, _, _ = hc.pcd_load(tmp_path / "g.pcd")
    assert rc == 0 and len(got) == n
    assert np.array_equal(got["x"], a["x"].astype(np.float32)) and np.array_equal(got["y"], a["y"])
    assert np.array_equal(got["row"], a["row"]) and np.array_equal(got["col"], a["col"]) and np.array_equal(got["label"], a["label"])
    assert (got["intensity"] == 0).all() and (got["t"] == 0).all()          # absent fields stay zero


def test_malformed_pcd_files_are_rejected(tmp_path):
    pts = synth.sweep(P, 1)[:1000]
    good = tmp_path / "good.pcd"
    pcd_util.write_pcd_binary(good, pts)
    raw = good.read_bytes()
    hdr_end = raw.index(b"DATA binary\n") + len(b"DATA binary\n")
    cases = {
        "truncated": raw[:hdr_end + 26 * 500],
        "huge_points": raw[:hdr_end].replace(b"POINTS 1000", b"POINTS 99999999999999") + raw[hdr_end:],
        "negative_points": raw[:hdr_end].replace(b"POINTS 1000", b"POINTS -5") + raw[hdr_end:],
        "bad_size": raw[:hdr_end].replace(b"SIZE 4 4 4 4 2 2 4 2", b"SIZE 4 4 4 4 2 3 4 2") + raw[hdr_end:],
        "zero_size": raw[:hdr_end].replace(b"SIZE 4 4 4 4 2 2 4 2", b"SIZE 0 4 4 4 2 2 4 2") + raw[hdr_end:],
        "negative_count": raw[:hdr_end].replace(b"COUNT 1 1", b"COUNT -1 1") + raw[hdr_end:],
        "bad_type": raw[:hdr_end].replace(b"TYPE F F F F U U U I", b"TYPE F F F F U U U Q") + raw[hdr_end:],
        "no_data_line": raw[:hdr_end - len(b"DATA binary\n")],
        "unknown_data": raw[:hdr_end].replace(b"DATA binary", b"DATA zipped") + raw[hdr_end:],
        "no_fields": b"VERSION 0.7\nPOINTS 3\nDATA binary\n" + b"\0" * 100,
        "empty_file": b"",
        "width_overflow": raw[:hdr_end].replace(b"WIDTH 1000", b"WIDTH 18446744073709551615").replace(b"HEIGHT 1", b"HEIGHT 7").replace(b"POINTS 1000\n", b"") + raw[hdr_end:],
        "ascii_short": raw[:hdr_end].replace(b"DATA binary", b"DATA ascii") + b"1 2 3 4 5 6 7 8\n" * 10,
    }
    comp = tmp_path / "c.pcd"
    pcd_util.write_pcd_binary_compressed(comp, pts)
    craw = comp.read_bytes()
    cend = craw.index(b"DATA binary_compressed\n") + len(b"DATA binary_compressed\n")
    cases["compressed_truncated"] = craw[:cend + 8 + 100]
    cases["compressed_bad_sizes"] = craw[:cend] + (2**31).to_bytes(4, "little") + craw[cend + 4:]
    cases["compressed_wrong_uncomp"] = craw[:cend + 4] + (12345).to_bytes(4, "little") + craw[cend + 8:]
    body = bytearray(craw[cend + 8:])
    body[10:40] = bytes([0xff]) * 30          # back references before the start of the output / overlong runs
    cases["compressed_garbage"] = craw[:cend + 8] + bytes(body)
    for name, data in cases.items():
        f = tmp_path / f"{name}.pcd"
        f.write_bytes(data)
        rc, got, _, _ = hc.pcd_load(f)
        assert rc != 0 and len(got) == 0, name
    assert hc.pcd_load(tmp_path / "missing.pcd")[0] != 0
    # garbage VALUES are accepted and converted without undefined behaviour (NaN -> 0, out of range saturates)
    (tmp_path / "vals.pcd").write_bytes(raw[:hdr_end].replace(b"DATA binary", b"DATA ascii").replace(b"POINTS 1000", b"POINTS 2").replace(b"WIDTH 1000", b"WIDTH 2")
                                        + b"nan inf -inf 1e40 1e9 -7 1e30 -1e30\n1 2 3 4 70000 65535 4294967295 -32768\n")
    rc, got, _, _ = hc.pcd_load(tmp_path / "vals.pcd")
    assert rc == 0 and len(got) == 2
    assert got["row"][1] == 65535 and got["col"][1] == 65535 and got["t"][1] == 4294967295 and got["label"][1] == -32768
    # "nan" / "inf" tokens are numbers to PCL's loader: the point keeps them AND the fields that follow on the line
    (tmp_path / "nan.pcd").write_bytes(raw[:hdr_end].replace(b"DATA binary", b"DATA ascii").replace(b"POINTS 1000", b"POINTS 3").replace(b"WIDTH 1000", b"WIDTH 3")
                                       + b"nan nan nan 0.5 17 901 77 -2\n1.5 inf -inf -1 3 4 5 -2\nbogus 2 3 4 5 6 7 8\n")
    rc, got, _, _ = hc.pcd_load(tmp_path / "nan.pcd")
    assert rc == 0 and len(got) == 3
    assert np.isnan(got["x"][0]) and np.isnan(got["z"][0]) and got["intensity"][0] == np.float32(0.5)
    assert (got["row"][0], got["col"][0], got["t"][0], got["label"][0]) == (17, 901, 77, -2)
    assert got["x"][1] == np.float32(1.5) and np.isposinf(got["y"][1]) and np.isneginf(got["z"][1]) and got["row"][1] == 3
    assert got["x"][2] == 0 and got["y"][2] == 2 and got["label"][2] == 8   # an unreadable token costs that value only


def test_png_and_csv(tmp_path):
    rng = np.random.default_rng(5)
    for img in (np.zeros((224, 224), np.uint8), rng.integers(0, 256, (224, 224)).astype(np.uint8),
                (rng.random((201, 77)) < 0.02).astype(np.uint8) * 255, np.full((1, 1), 9, np.uint8)):
        assert hc.png_write(tmp_path / "i.png", img) == 0
        assert np.array_equal(pcd_util.read_png_gray8(tmp_path / "i.png"), img)
        text = hc.csv_u8(img)
        assert text == "".join(", ".join("%3d" % v for v in row) + "\n" for row in img)
    assert len(hc.csv_u8(np.zeros((224, 224), np.uint8))) == 250656

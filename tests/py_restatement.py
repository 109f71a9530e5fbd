"""A SECOND, independent restatement of the hot path: plain Python / numpy scalars, written loop by loop from the
reference text (BatchMultiBevGen.cpp:94-117, 119-252, 261-292, 331-356; BatchMultiBevGen.h:73-99), with every
float / double conversion of the C++ expressions spelled out.  Test infrastructure only: tests/test_oracle_vs_python.py
holds the C oracle against it on small sensors (pure-Python loops are slow).  Like the oracle it adopts the float
overloads of atan2 / sqrt / abs / round and x86-64's cvttss2si / cvttsd2si for the unchecked
float -> int casts."""
import math

import numpy as np

from bev_amd import POINT_DTYPE

F, D = np.float32, np.float64
INT_MIN = -2**31


def cvtt(v):
    """static_cast<int>(floating value) on x86-64: truncation; NaN / out of range -> INT_MIN"""
    v = float(v)
    if math.isnan(v) or math.isinf(v):
        return INT_MIN
    t = math.trunc(v)
    return t if -2**31 <= t < 2**31 else INT_MIN


def round_half_away(v):
    """round() / roundf(): half away from zero (on a value that already has the right precision)"""
    v = float(v)
    if math.isnan(v) or math.isinf(v):
        return v
    return math.floor(v + 0.5) if v >= 0 else -math.floor(-v + 0.5)


def belonging_grid(x, y):
    """BatchMultiBevGen.h:73-99"""
    nx = F(D(x) + 75.0)                        # float normalized_x = x + 75.0;
    ny = F(D(y) + 50.0)
    with np.errstate(all="ignore"):
        r = cvtt(np.floor(D(nx) / 2.0))        # static_cast<int>(std::floor(normalized_x / 2.0))
        c = cvtt(np.floor(D(ny) / 2.0))
    if r >= 75:
        r = 74
    if r < 0:
        r = 0
    if c >= 50:
        c = 49
    if c < 0:
        c = 0
    return r, c


def order_cloud(N, H, pts):
    """BatchMultiBevGen.cpp:94-117"""
    out = np.zeros(N * H, POINT_DTYPE)           # resize(): value-initialised points
    for p in pts:
        row, col = int(p["row"]), int(p["col"])
        if row < 0 or row >= N:
            continue
        if col < 0 or col >= H:
            continue
        out[row * H + col] = p
    return out


def c_mod(a, b):
    """C++ % (sign of the dividend)"""
    return int(math.fmod(a, b))


def mark_ground(N, H, G, cloud):
    """BatchMultiBevGen.cpp:119-252; labels are written into `cloud` in place"""
    gm = np.zeros((N, H), np.int8)
    sums = np.zeros((75, 50), F)
    cnts = np.full((75, 50), F(F(1.0) * 0.01), F)          # 0.01 * Mat::ones(CV_32F)
    pts = cloud
    with np.errstate(all="ignore"):
        for col in range(H):
            for row in range(N - 1, N - G - 1, -1):
                lower = row * H + col
                upper = (row - 1) * H + col
                if pts[upper]["intensity"] == -1:
                    upper = (row - 1) * H + c_mod(col + 2, H)
                if pts[upper]["intensity"] == -1:
                    upper = (row - 1) * H + c_mod(col - 2, H)
                if pts[upper]["intensity"] == -1 and row >= 2:
                    upper = (row - 2) * H + col
                if pts[lower]["intensity"] == -1 or pts[upper]["intensity"] == -1:
                    gm[row, col] = -1
                    continue
                dx = F(pts[upper]["x"] - pts[lower]["x"])
                dy = F(pts[upper]["y"] - pts[lower]["y"])
                dz = F(pts[upper]["z"] - pts[lower]["z"])
                horiz = np.sqrt(F(F(dx * dx) + F(dy * dy)))         # sqrtf
                angle = F(D(np.arctan2(dz, horiz)) * 180.0 / math.pi)  # atan2f, then double arithmetic, stored to float
                if abs(F(angle - F(0.0))) <= F(10.0):
                    gm[row, col] = 1
                    gm[row - 1, col] = 1
        for row in range(N):
            for col in range(H):
                if gm[row, col] != 1:
                    continue
                i = row * H + col
                r, c = belonging_grid(pts[i]["x"], pts[i]["y"])
                sums[r, c] = F(sums[r, c] + pts[i]["z"])
                cnts[r, c] = F(cnts[r, c] + F(1))
        avg = (sums / cnts).astype(F)
        for row in range(N):
            for col in range(H):
                i = row * H + col
                r, c = belonging_grid(pts[i]["x"], pts[i]["y"])
                for dr, dc in ((-1, 0), (0, 1), (0, -1), (1, 0)):
                    nr, nc = r + dr, c + dc
                    if nr < 0 or nr >= 75 or nc < 0 or nc >= 50:
                        continue
                    if D(F(pts[i]["z"] - avg[nr, nc])) > 0.30:
                        gm[row, col] = 0
                        break
                if gm[row, col] == 1:
                    pts[i]["label"] = 0
    return gm, avg


def bev_bin(p, max_range, interval):
    with np.errstate(all="ignore"):
        v = D(F(F(p + F(max_range)) / F(interval))) + 0.5        # (pi.x + MAX_RANGE) / interval in float, + 0.5 in double
    return cvtt(round_half_away(v))


def multi_bev(cloud, height_res, interval=1.0):
    """BatchMultiBevGen.cpp:266-292"""
    M = cvtt(F(112 * 2) / F(interval))
    out = np.zeros((24, M, M), np.uint8)
    with np.errstate(all="ignore"):
        for p in cloud:
            x = bev_bin(p["x"], 112, interval)
            y = bev_bin(p["y"], 112, interval)
            layer = cvtt(round_half_away(F(F(p["z"] / F(height_res)) + F(2.0))))   # roundf of a float expression
            if x < 0 or x >= M or y < 0 or y >= M or layer < 0 or layer >= 24 or p["label"] == 0:
                continue
            if out[layer, x, y] == 0:
                out[layer, x, y] = 255
    return out


def single_bev(cloud, interval=1.0):
    """BatchMultiBevGen.cpp:336-356"""
    M = cvtt(F(112 * 2) / F(interval))
    out = np.zeros((M, M), np.uint8)
    with np.errstate(all="ignore"):
        for p in cloud:
            x = bev_bin(p["x"], 112, interval)
            y = bev_bin(p["y"], 112, interval)
            height = cvtt(D(F(p["z"] + F(2.0))) * 4.0)
            height = min(max(0, height), 255)
            if x < 0 or x >= M or y < 0 or y >= M or p["label"] == 0:
                continue
            if out[x, y] < height:
                out[x, y] = height
    return out


def process_frame(N, H, G, height_res, pts):
    ordered = order_cloud(N, H, pts)
    gm, avg = mark_ground(N, H, G, ordered)
    return ordered, gm, avg, multi_bev(ordered, height_res), single_bev(ordered)

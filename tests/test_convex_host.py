"""CPU tier: the kernels' lane-private convex narrowphase (gym-genesis_amd/csrc/mir_convex.h: GJK on the cores, MPR when they
overlap) compiled for the HOST with g++ (tests/convex_host.cpp: the same source, float32) against the float64 oracle, pair by
pair.  Catches logic errors in the device code without a GPU; the GPU tests then only have to show that the device build of the
same source behaves the same."""
import ctypes as C
import os
import subprocess

import numpy as np

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")


def host_lib():
    os.makedirs(BUILD, exist_ok=True)
    so = os.path.join(BUILD, "libconvex_host.so")
    srcs = [os.path.join(HERE, "convex_host.cpp"), os.path.join(HERE, "..", "gym-genesis_amd", "csrc", "mir_convex.h")]
    if not os.path.exists(so) or any(os.path.getmtime(so) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-ffp-contract=off", srcs[0], "-o", so])
    return C.CDLL(so)


def random_pairs(n, seed):
    rng = np.random.default_rng(seed)
    T = rng.choice([1, 2, 3], (n, 2))       # box, sphere, capsule
    T[(T == 1).all(1), 1] = 3               # (box-box has its own narrowphase)
    rows = np.zeros((n, 22), np.float32)
    rows[:, 0], rows[:, 11] = T[:, 0], T[:, 1]
    rows[:, 1:4], rows[:, 12:15] = rng.uniform(0.03, 0.15, (n, 3)), rng.uniform(0.03, 0.15, (n, 3))
    rows[:, 4:7], rows[:, 15:18] = rng.uniform(-0.2, 0.2, (n, 3)), rng.uniform(-0.2, 0.2, (n, 3))
    for c in (7, 18):
        q = rng.normal(size=(n, 4))
        rows[:, c:c + 4] = q / np.linalg.norm(q, axis=1, keepdims=True)
    return rows


def oracle_pairs(rows):
    lib = orc.load(False)
    lib.orc_narrowphase.restype = C.c_int
    out = np.zeros((rows.shape[0], 8))
    arr = lambda v: (C.c_double * len(v))(*v)  # noqa: E731
    for i, r in enumerate(rows.astype(np.float64)):
        pts, nrm = (C.c_double * 32)(), (C.c_double * 3)()
        cnt = lib.orc_narrowphase(int(r[0]), arr(r[1:4]), arr(r[4:7]), arr(r[7:11]), int(r[11]), arr(r[12:15]), arr(r[15:18]), arr(r[18:22]), pts, nrm)
        if cnt:
            out[i] = [1, pts[0], pts[1], pts[2], pts[3], nrm[0], nrm[1], nrm[2]]
    return out


def compare(rows, got, ref):
    """Shallow contacts (GJK on the cores): depth 2e-6, normal 2e-4, position 1e-5.  Deep ones (MPR, float32 against float64 of
    the same portal refinement): hit agreed; depth within 5 % everywhere and within 1e-4 for 90 % of the pairs; normal within
    2e-2 rad for 98 % of the pairs and 2e-3 rad for 90 % (a portal that stops one refinement apart in the two precisions can swing
    the normal of a near-degenerate pair: measured tail 0.29 rad on 1 of 193)."""
    shallow = deep = 0
    drel, dang = [], []
    for i in range(rows.shape[0]):
        hit, dist = bool(ref[i, 0]), ref[i, 4]
        if (hit and dist > -1e-4) or (not hit and got[i, 0] and got[i, 4] > -1e-4):
            continue  # grazing
        assert bool(got[i, 0]) == hit, (i, rows[i], got[i], ref[i])
        if not hit:
            continue
        r = rows[i]
        radii = (0 if r[0] == 1 else r[1]) + (0 if r[11] == 1 else r[12])
        if -dist < 0.9 * radii:
            assert abs(got[i, 4] - dist) < 2e-6, (i, got[i], ref[i])
            assert np.abs(got[i, 5:8] - ref[i, 5:8]).max() < 2e-4 * max(1.0, 0.01 / (radii + dist)), (i, got[i], ref[i])
            assert np.abs(got[i, 1:4] - ref[i, 1:4]).max() < 1e-5, (i, got[i], ref[i])
            shallow += 1
        elif -dist > 1.1 * radii:
            drel.append(abs(got[i, 4] - dist) / abs(dist))
            dang.append(float(np.arccos(np.clip(np.dot(got[i, 5:8], ref[i, 5:8]), -1.0, 1.0))))
            assert drel[-1] < 0.05, (i, got[i], ref[i])
            deep += 1
    if deep >= 50:
        drel, dang = np.array(drel), np.array(dang)
        assert np.quantile(drel, 0.9) < 1e-4, np.quantile(drel, 0.9)
        assert np.quantile(dang, 0.98) < 2e-2 and np.quantile(dang, 0.9) < 2e-3, (np.quantile(dang, 0.98), np.quantile(dang, 0.9))
    return shallow, deep


def test_host_build_of_the_device_narrowphase_matches_the_oracle():
    rows = random_pairs(4000, 21)
    got = np.zeros((rows.shape[0], 8), np.float32)
    host_lib().convex_host_pairs(rows.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), rows.shape[0])
    shallow, deep = compare(rows, got, oracle_pairs(rows))
    assert shallow > 300 and deep > 100, (shallow, deep)


def analytic_deep_pairs():
    """Deep pairs whose minimum-translation answer is known in closed form AND lies along the line of centres (where portal
    refinement started from the centre difference is exact): a sphere whose centre is inside a box, displaced from the box centre
    along one face normal only -- depth = r + (h - offset), normal = that face normal; a capsule parallel to a box edge with its
    whole axis inside the box, displaced the same way.  Box A first, axis-aligned or rotated by a quaternion applied to both."""
    rng = np.random.default_rng(5)
    rows, want = [], []
    for k in range(60):
        h = rng.uniform(0.05, 0.12, 3)
        ax = int(rng.integers(3))
        sgn = 1.0 if rng.random() < 0.5 else -1.0
        off = rng.uniform(0.2, 0.8) * h[ax]
        r = rng.uniform(0.01, 0.04)
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        if k % 2 == 0:
            q = np.array([1.0, 0, 0, 0])
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        ca = rng.uniform(-0.1, 0.1, 3)
        e = np.zeros(3); e[ax] = sgn
        cb = ca + R @ (off * e)
        row = np.zeros(22, np.float32)
        row[0] = 1; row[1:4] = h; row[4:7] = ca; row[7:11] = q
        if k % 3 == 2:   # capsule along another box axis, short enough to stay inside
            ax2 = (ax + 1) % 3
            hl = 0.5 * h[ax2]
            # capsule z axis -> box axis ax2 (in the box frame), then the common rotation
            qz = {0: np.array([np.sqrt(0.5), 0, np.sqrt(0.5), 0]), 1: np.array([np.sqrt(0.5), -np.sqrt(0.5), 0, 0]), 2: np.array([1.0, 0, 0, 0])}[ax2]
            qq = np.array([q[0] * qz[0] - q[1:] @ qz[1:], *(q[0] * qz[1:] + qz[0] * q[1:] + np.cross(q[1:], qz[1:]))])
            row[11] = 3; row[12:15] = (r, hl, 0); row[15:18] = cb; row[18:22] = qq
        else:
            row[11] = 2; row[12:15] = (r, 0, 0); row[15:18] = cb; row[18:22] = (1, 0, 0, 0)
        rows.append(row)
        want.append((r + (h[ax] - off), R @ e))   # depth, normal from A to B
    return np.array(rows, np.float32), want


def test_mpr_on_analytic_cases_depth_5_percent_normal_2e_2_rad():
    rows, want = analytic_deep_pairs()
    got = np.zeros((rows.shape[0], 8), np.float32)
    host_lib().convex_host_pairs(rows.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), rows.shape[0])
    ref = oracle_pairs(rows)
    for i, (depth, n) in enumerate(want):
        for name, res in (("device source (host build, float32)", got[i]), ("oracle", ref[i])):
            assert res[0] == 1, (name, i, rows[i], res)
            assert abs(-res[4] - depth) < 0.05 * depth, (name, i, -res[4], depth)
            ang = np.arccos(np.clip(np.dot(res[5:8], n), -1, 1))
            assert ang < 2e-2, (name, i, ang, res[5:8], n)

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
    """Shallow contacts (GJK on the cores): depth 2e-6, normal 2e-4, position 1e-5; deep ones (MPR): hit agreed, depth 20 %."""
    shallow = deep = 0
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
            assert abs(got[i, 4] - dist) < 0.2 * abs(dist), (i, got[i], ref[i])
            deep += 1
    return shallow, deep


def test_host_build_of_the_device_narrowphase_matches_the_oracle():
    rows = random_pairs(4000, 21)
    got = np.zeros((rows.shape[0], 8), np.float32)
    host_lib().convex_host_pairs(rows.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), rows.shape[0])
    shallow, deep = compare(rows, got, oracle_pairs(rows))
    assert shallow > 300 and deep > 100, (shallow, deep)

"""ctypes driver for the CPU oracle (oracle/liborc64.so / liborc32.so).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (gym-genesis_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

ORACLE_DIR = os.path.dirname(os.path.abspath(__file__))

(F_QPOS, F_QVEL, F_TARGET, F_QACC_WS, F_XPOS, F_XQUAT, F_XIPOS, F_M, F_MT, F_QFRC_BIAS, F_QFRC_SMOOTH,
 F_QACC_SMOOTH, F_QACC, F_CPOS, F_CDIST, F_CFRAME, F_J, F_AREF, F_EFCD, F_EFCFORCE, F_DOF_INVWEIGHT0,
 F_BODY_INVWEIGHT0, F_MEANINERTIA, F_QFRC_ACT, F_QFRC_PASSIVE, F_EFCPOS, F_DBG_IMP, F_DBG_GN, F_DBG_ALPHA, F_DBG_LS) = range(30)


def build_oracle() -> None:
    """Compile the oracle with its committed Makefile if the .so files are missing/stale."""
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR], stdout=subprocess.DEVNULL)


_libs = {}


def load(f32: bool = False, flops=False, variant=None):
    """liborc64.so (float64 oracle), liborc32.so / liborc32_big.so (float32 port at the pick-task / full capacities; f32 = True / "big") or liborc_flops.so / liborc_flops_big.so (the float32 port with
    counted arithmetic at the pick-task / full capacities; flops = True / "big")."""
    key = ("_flops_big" if flops == "big" else "_flops") if flops else (("32_big" if f32 == "big" else "32") if f32 else "64")
    if variant == "norules":  # (the float64 oracle without the stopping rules it shares with the kernels: `make -C oracle norules`)
        key = "64_norules"
    if key == "64" and os.environ.get("ORC_SANITIZE"):  # (tests/test_sanitizers_cpu.py: the ASan + UBSan build, `make -C oracle asan`)
        key = "64_asan"
    if key not in _libs:
        path = os.path.join(ORACLE_DIR, f"liborc{key}.so")
        srcs = [os.path.join(ORACLE_DIR, "orc_rigid.c"), os.path.join(ORACLE_DIR, "orc_render.c"), os.path.join(ORACLE_DIR, "orc_rigid.h"),
                os.path.join(ORACLE_DIR, "orc_flops.h"), os.path.join(ORACLE_DIR, "..", "include", "mirigid.h")]
        if not os.path.exists(path) or any(os.path.getmtime(path) < os.path.getmtime(s) for s in srcs):
            if key == "64_asan":
                subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "asan"], stdout=subprocess.DEVNULL)
            elif key == "64_norules":
                subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "norules"], stdout=subprocess.DEVNULL)
            else:
                build_oracle()
        lib = C.CDLL(path)
        lib.orc_read.restype = C.c_int
        lib.orc_compile.restype = C.c_int
        _libs[key] = lib
    return _libs[key]


class Oracle:
    """One compiled model + a batch of per-env data blocks."""

    def __init__(self, spec, num_envs: int = 1, f32: bool = False, flops=False, variant=None):
        self.lib = load(f32, flops, variant)
        self.spec = spec
        self.B = num_envs
        self.model = C.create_string_buffer(self.lib.orc_sizeof_model())
        rc = self.lib.orc_compile(C.byref(spec), self.model)
        if rc != 0:
            raise RuntimeError(f"orc_compile failed: {rc}")
        self.dsize = self.lib.orc_sizeof_data()
        self.data = C.create_string_buffer(self.dsize * num_envs)
        self._dptr = C.addressof(self.data)
        for e in range(num_envs):
            self.lib.orc_init_data(self.model, C.c_void_p(self._dptr + e * self.dsize))
        # dims via a first read
        self.nq = len(self.read(F_QPOS))
        self.nv = len(self.read(F_QVEL))
        self.nbody = spec.nbody
        self.n_grip = spec.task.n_grip
        self.nfree = sum(1 for b in range(1, spec.nbody) if spec.body[b].jtype == 3)
        n_arm = sum(1 for b in range(1, spec.nbody) if spec.body[b].jtype in (1, 2))
        self.agent_dim = n_arm if spec.task.agent_mode == 1 else 7 + spec.task.n_grip
        self.env_dim = 14 if spec.task.obj2_body >= 0 else 11
        self.nu = sum(1 for i in range(spec.ndof) if spec.dof[i].ctrl_mode == 1)
        self.u_dofs = [i for i in range(spec.ndof) if spec.dof[i].ctrl_mode == 1]

    FLOP_KINDS = ("add", "mul", "div", "sqrt", "trans", "cmp")

    def flops_reset(self) -> None:
        self.lib.orc_flops_reset()

    def flops(self) -> dict:
        """Operations the CALLING thread has executed in this library since flops_reset() (liborc_flops.so; zeros otherwise)."""
        out = (C.c_ulonglong * 6)()
        self.lib.orc_flops_read(out)
        return dict(zip(self.FLOP_KINDS, (int(x) for x in out)))

    def d(self, e: int = 0):
        return C.c_void_p(self._dptr + e * self.dsize)

    # ---- single-env accessors ------------------------------------------------------------
    def read(self, field: int, e: int = 0) -> np.ndarray:
        out = np.zeros(1 << 15, dtype=np.float64)  # the largest field: J, (4 x 48 + 48) rows x 48 dofs
        n = self.lib.orc_read(self.model, self.d(e), field, out.ctypes.data_as(C.c_void_p))
        if n < 0:
            raise KeyError(field)
        return out[:n].copy()

    def write(self, field: int, val, e: int = 0) -> None:
        v = np.ascontiguousarray(val, dtype=np.float64)
        self.lib.orc_write(self.model, self.d(e), field, v.ctypes.data_as(C.c_void_p))

    def counts(self, e: int = 0):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        self.lib.orc_counts(self.d(e), C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value

    def fk(self, e: int = 0):
        self.lib.orc_fk(self.model, self.d(e))

    def forward(self, e: int = 0):
        self.lib.orc_forward(self.model, self.d(e))

    def step(self, e: Optional[int] = None):
        for i in (range(self.B) if e is None else [e]):
            self.lib.orc_step(self.model, self.d(i))

    def aba(self, e: int = 0) -> np.ndarray:
        out = np.zeros(self.nv)
        self.lib.orc_aba(self.model, self.d(e), out.ctypes.data_as(C.c_void_p))
        return out

    # ---- batched, mirroring the C ABI ----------------------------------------------------
    def reset(self, obj_pos, obj_quat, arm_qpos):
        """obj_pos (B, nfree, 3) / obj_quat (B, nfree, 4): poses of all free bodies in body order ((B,3)/(B,4) with one)."""
        obj_pos = np.ascontiguousarray(obj_pos, dtype=np.float64).reshape(self.B, self.nfree * 3)
        obj_quat = np.ascontiguousarray(obj_quat, dtype=np.float64).reshape(self.B, self.nfree * 4)
        arm_qpos = np.ascontiguousarray(arm_qpos, dtype=np.float64).reshape(self.B, -1)
        for e in range(self.B):
            self.lib.orc_reset(self.model, self.d(e), obj_pos[e].ctypes.data_as(C.c_void_p),
                               obj_quat[e].ctypes.data_as(C.c_void_p), arm_qpos[e].ctypes.data_as(C.c_void_p))

    def set_targets(self, tgt):
        tgt = np.ascontiguousarray(tgt, dtype=np.float64).reshape(self.B, -1)
        for e in range(self.B):
            self.lib.orc_set_targets(self.model, self.d(e), tgt[e].ctypes.data_as(C.c_void_p))

    def step_batch(self, action=None, nthreads: int = 0):
        a = None
        if action is not None:
            a = np.ascontiguousarray(action, dtype=np.float32)
        self.lib.orc_step_batch(self.model, self.data, C.c_int(self.B),
                                a.ctypes.data_as(C.c_void_p) if a is not None else None, C.c_int(nthreads))

    def get_obs(self):
        agent = np.zeros((self.B, self.agent_dim))
        env = np.zeros((self.B, self.env_dim))
        rew = np.zeros(self.B)
        term = np.zeros(self.B, dtype=np.uint8)
        for e in range(self.B):
            self.lib.orc_get_obs(self.model, self.d(e), agent[e].ctypes.data_as(C.c_void_p),
                                 env[e].ctypes.data_as(C.c_void_p), rew[e:e + 1].ctypes.data_as(C.c_void_p),
                                 term[e:e + 1].ctypes.data_as(C.c_void_p))
        return agent, env, rew, term

    def ik(self, link: int, target_pos, target_quat=None, init_q=None, max_iters=20, damping=0.05, pos_tol=5e-4, rot_tol=5e-3,
           max_step=0.5, respect_limits=True):
        """Damped-least-squares IK of include/mirigid.h on the oracle's own kinematics, one env at a time.
        target_pos (n,3), target_quat (n,4) or None, init_q (n,n_arm).  Returns (q (n,n_arm), err (n,2))."""
        tp = np.ascontiguousarray(target_pos, dtype=np.float64).reshape(-1, 3)
        n = tp.shape[0]   # (the solver reads the model only: any number of rows -- the test double's mir_inverse_kinematics_rows passes len(envs_idx))
        tq = None if target_quat is None else np.ascontiguousarray(target_quat, dtype=np.float64).reshape(n, 4)
        q = np.ascontiguousarray(init_q, dtype=np.float64).reshape(n, -1).copy()
        err = np.zeros((n, 2))
        self.lib.orc_ik.restype = C.c_int
        self.ik_iters = np.zeros(n, np.int32)  # (iterations each row took in the last call)
        for e in range(n):
            self.ik_iters[e] = self.lib.orc_ik(self.model, C.c_int(link), tp[e].ctypes.data_as(C.c_void_p), tq[e].ctypes.data_as(C.c_void_p) if tq is not None else None,
                            q[e].ctypes.data_as(C.c_void_p), C.c_int(max_iters), C.c_double(damping), C.c_double(pos_tol), C.c_double(rot_tol),
                            C.c_double(max_step), C.c_int(1 if respect_limits else 0), err[e].ctypes.data_as(C.c_void_p))
        return q, err

    def read_all(self, field: int, n: int) -> np.ndarray:
        """(B, n) float64: field `field` (of length <= n) of every env in one call (orc_read_batch)."""
        out = np.zeros((self.B, n))
        self.lib.orc_read_batch.restype = C.c_int
        got = self.lib.orc_read_batch(self.model, self.data, C.c_int(self.B), C.c_int(field), out.ctypes.data_as(C.c_void_p), C.c_int(n))
        if got < 0:
            raise KeyError(field)
        return out[:, :got]

    def write_all(self, field: int, vals) -> None:
        """field `field` of every env from the rows of vals (B, n) in one call (orc_write_batch)"""
        v = np.ascontiguousarray(vals, dtype=np.float64)
        assert v.ndim == 2 and v.shape[0] == self.B
        self.lib.orc_write_batch.restype = C.c_int
        if self.lib.orc_write_batch(self.model, self.data, C.c_int(self.B), C.c_int(field), v.ctypes.data_as(C.c_void_p), C.c_int(v.shape[1])) < 0:
            raise KeyError(field)

    def counts_all(self):
        """(ncon, nefc, niter) int32 arrays over the envs (orc_counts_batch)."""
        a, b, c = (np.zeros(self.B, np.int32) for _ in range(3))
        self.lib.orc_counts_batch(self.data, C.c_int(self.B), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
        return a, b, c

    def ncand_all(self) -> np.ndarray:
        """candidate contact points of every env's last collision pass, before the capacity was applied (orc_ncand_batch)"""
        a = np.zeros(self.B, np.int32)
        self.lib.orc_ncand_batch(self.data, C.c_int(self.B), a.ctypes.data_as(C.c_void_p))
        return a

    def get_obs_all(self):
        """get_obs() in one call (orc_get_obs_batch)."""
        agent, env = np.zeros((self.B, self.agent_dim)), np.zeros((self.B, self.env_dim))
        rew, term = np.zeros(self.B), np.zeros(self.B, dtype=np.uint8)
        self.lib.orc_get_obs_batch(self.model, self.data, C.c_int(self.B), agent.ctypes.data_as(C.c_void_p), C.c_int(self.agent_dim),
                                   env.ctypes.data_as(C.c_void_p), C.c_int(self.env_dim), rew.ctypes.data_as(C.c_void_p), term.ctypes.data_as(C.c_void_p))
        return agent, env, rew, term

    def state(self):
        return self.read_all(F_QPOS, self.nq), self.read_all(F_QVEL, self.nv)


def render_image(spec, cam, vis, xpos, xquat, offsets=None, want_depth=False):
    """Brute-force float64 ray cast of one image (orc_render.c).  xpos (nenv,nbody,3), xquat (nenv,nbody,4)."""
    lib = load(False)
    xpos = np.ascontiguousarray(xpos, dtype=np.float64)
    xquat = np.ascontiguousarray(xquat, dtype=np.float64)
    nenv = xpos.shape[0]
    out = np.zeros((cam.height, cam.width, 3), dtype=np.uint8)
    depth = np.zeros((cam.height, cam.width)) if want_depth else None
    off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.float64)
    lib.orc_render_image.restype = C.c_int
    rc = lib.orc_render_image(C.byref(spec), C.byref(cam), C.byref(vis), C.c_int(nenv), xpos.ctypes.data_as(C.c_void_p),
                              xquat.ctypes.data_as(C.c_void_p), off.ctypes.data_as(C.c_void_p) if off is not None else None,
                              out.ctypes.data_as(C.c_void_p), depth.ctypes.data_as(C.c_void_p) if want_depth else None)
    if rc != 0:
        raise RuntimeError(f"orc_render_image failed: {rc}")
    return (out, depth) if want_depth else out

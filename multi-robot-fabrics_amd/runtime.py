"""Thin torch-tensor front of the C ABI (include/mrf.h): one `FabricHandle` per (config, device).

torch is plumbing here -- device memory, streams -- and nothing else: every numeric result comes from
the HIP kernels behind csrc/libmrf_hip.so.  There is no CPU path; constructing a handle without the
library or without a GPU raises.
"""
import ctypes as C
import os

import torch

from . import abi


class MrfError(RuntimeError):
    pass


def _dtype(cfg):
    return torch.float64 if cfg.scalar == abi.F64 else torch.float32


def device_topology(cap=16):
    """The node as HIP shows it to this process (mrf_device_topology; no handle, no context on other devices):
    {"n_devices", "can_access_peer" [n][n], "link_type" [n][n] (names), "hops" [n][n]}."""
    lib = abi.load_library()
    n = C.c_int32(0)
    can, lt, hp = ((C.c_int32 * (cap * cap))() for _ in range(3))
    rc = lib.mrf_device_topology(C.byref(n), can, lt, hp, cap)
    m = max(0, min(int(n.value), cap))
    grid = lambda a, f=int: [[f(a[i * cap + j]) for j in range(m)] for i in range(m)]
    return {"n_devices": int(n.value), "status": abi.STATUS_TEXT.get(rc, str(rc)), "can_access_peer": grid(can),
            "link_type": grid(lt, lambda v: abi.LINK_TYPE_NAMES.get(int(v), str(int(v)))), "hops": grid(hp)}


def streams_concurrent(device_index, stream_a, stream_b):
    """True when kernels on the two torch streams really run side by side (mrf_streams_concurrent)."""
    lib = abi.load_library()
    out = C.c_int32(0)
    rc = lib.mrf_streams_concurrent(int(device_index), C.c_void_p(stream_a.cuda_stream), C.c_void_p(stream_b.cuda_stream), C.byref(out))
    if rc != 0:
        raise MrfError(f"mrf_streams_concurrent: {abi.STATUS_TEXT.get(rc, rc)}")
    return bool(out.value)


class FabricHandle:
    """Owns an `mrf_handle` (immutable constants on the device).  Not thread-safe, like the C handle."""

    def __init__(self, cfg, device=None):
        """device: index / torch.device; None = torch's current device (so that a rank that called
        torch.cuda.set_device(local_rank) gets its own GPU).  The C handle pins its device: calls run there whatever
        the current device is, and leave the current device untouched."""
        self.lib = abi.load_library()
        if not torch.cuda.is_available():
            raise MrfError("no HIP device visible to torch; the fabric solve has no CPU fallback")
        self.cfg = cfg.copy()
        if os.environ.get("MRF_ABI_ANY") == "1":        # A/B timing of an older build (abi.load_library)
            self.cfg.abi_version = self.lib.mrf_abi_version()
        if device is None:
            device = torch.cuda.current_device()
        if not isinstance(device, int):
            d = torch.device(device)
            device = torch.cuda.current_device() if d.index is None else d.index
        self.device = torch.device("cuda", device)
        self.dtype = _dtype(cfg)
        self.dof = 7 if cfg.model == abi.MODEL_PANDA7 else 3
        self._h = C.c_void_p()
        rc = self.lib.mrf_create(C.byref(self.cfg), self.device.index, C.byref(self._h))
        if rc != 0:
            msg = self.lib.mrf_last_error(self._h).decode() if self._h else ""
            if self._h:
                self.lib.mrf_destroy(self._h)
                self._h = C.c_void_p()
            raise MrfError(f"mrf_create failed: {abi.STATUS_TEXT.get(rc, rc)} {msg}")

    @property
    def constants_source(self):
        """The reconciled-constants file in force when planners are built in this process (config.reconciled_constants_source),
        or None: the recalled defaults."""
        from . import config
        return config.reconciled_constants_source()

    def close(self):
        if getattr(self, "_h", None):
            self.lib.mrf_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _check(self, rc):
        if rc != 0:
            raise MrfError(f"{abi.STATUS_TEXT.get(rc, rc)}: {self.lib.mrf_last_error(self._h).decode()}")

    def tensor(self, a):
        """Host/any array -> contiguous device tensor of the handle's scalar type."""
        return torch.as_tensor(a, dtype=self.dtype).to(self.device).contiguous()

    def upload(self, *arrays):
        """Several host arrays -> device tensors of the handle's scalar type with ONE host-to-device copy
        (the single-row planner calls are latency-bound by the number of transfers, not by their size)."""
        import numpy as np
        np_dtype = np.float64 if self.dtype == torch.float64 else np.float32
        arrs = [np.ascontiguousarray(a, dtype=np_dtype) for a in arrays]
        sizes = [a.size for a in arrs]
        flat = np.concatenate([a.reshape(-1) for a in arrs]) if arrs else np.zeros(0, dtype=np_dtype)
        dev = torch.from_numpy(flat).to(self.device)
        out, off = [], 0
        for a, n in zip(arrs, sizes):
            out.append(dev[off:off + n].view(a.shape))
            off += n
        return out

    def _arg(self, t, shape=None, name="tensor"):
        if t is None:
            return None
        if not isinstance(t, torch.Tensor) or t.dtype != self.dtype or t.device != self.device or not t.is_contiguous():
            raise MrfError(f"{name}: expected a contiguous {self.dtype} tensor on {self.device}")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise MrfError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return C.c_void_p(t.data_ptr())

    def _stream(self, stream):
        """Stream to launch on: the caller's, or torch's current stream OF THE HANDLE'S DEVICE."""
        s = torch.cuda.current_stream(self.device) if stream is None else stream
        if getattr(s, "device", self.device) != self.device:
            raise MrfError(f"stream belongs to {s.device}, the handle to {self.device}")
        return C.c_void_p(s.cuda_stream)

    # ------------------------------------------------------------------ entry points
    def compute_action(self, q, qdot, params, obst_x=None, obst_v=None, obst_a=None, obst_r=None,
                       want_qddot=False, n_static=0, stream=None):
        """q,qdot [dof,rows]; params [29,rows]; obst_x/v/a [M,3,rows]; obst_r [M,rows] -> action [dof,rows].
        The first n_static obstacles are static leaves (3-D, no motion), the rest dynamic leaves."""
        rows = q.shape[1]
        M = 0 if obst_x is None else obst_x.shape[0]
        act = torch.empty((self.dof, rows), dtype=self.dtype, device=self.device)
        qdd = torch.empty_like(act) if want_qddot else None
        rc = self.lib.mrf_compute_action(
            self._h, rows, self._arg(q, (self.dof, rows), "q"), self._arg(qdot, (self.dof, rows), "qdot"),
            self._arg(params, (abi.NPARAM, rows), "params"), M, n_static, self._arg(obst_x, (M, 3, rows), "obst_x"),
            self._arg(obst_v, (M, 3, rows), "obst_v"), self._arg(obst_a, (M, 3, rows), "obst_a"),
            self._arg(obst_r, (M, rows), "obst_r"), self._arg(qdd), self._arg(act), self._stream(stream))
        self._check(rc)
        return (act, qdd) if want_qddot else act

    def compute_action_coupled(self, q, qdot, params, use_accel=False, want_qddot=False, stream=None):
        """compute_action of every robot against the spheres of the other robots of its scenario (EXJ:394-448 on the
        device); rows = n_scenarios * n_robots -> action [7, rows]."""
        rows = q.shape[1]
        N = self.cfg.n_robots
        if rows % N:
            raise MrfError("rows must be a multiple of n_robots")
        act = torch.empty((self.dof, rows), dtype=self.dtype, device=self.device)
        qdd = torch.empty_like(act) if want_qddot else None
        rc = self.lib.mrf_compute_action_coupled(self._h, rows // N, self._arg(q, (self.dof, rows), "q"),
                                                 self._arg(qdot, (self.dof, rows), "qdot"),
                                                 self._arg(params, (abi.NPARAM, rows), "params"), int(bool(use_accel)),
                                                 self._arg(qdd), self._arg(act), self._stream(stream))
        self._check(rc)
        return (act, qdd) if want_qddot else act

    def rollout(self, q0, qdot0, params, want_traj=False, stream=None):
        """Coupled joint-space rollout; rows = n_scenarios * n_robots -> avg_vel [rows] (, traj_q, traj_qdot)."""
        rows = q0.shape[1]
        N, H = self.cfg.n_robots, self.cfg.horizon
        if rows % N:
            raise MrfError("rows must be a multiple of n_robots")
        avg = torch.empty((rows,), dtype=self.dtype, device=self.device)
        tq = torch.empty((H, self.dof, rows), dtype=self.dtype, device=self.device) if want_traj else None
        tqd = torch.empty_like(tq) if want_traj else None
        rc = self.lib.mrf_rollout(self._h, rows // N, self._arg(q0, (self.dof, rows), "q0"),
                                  self._arg(qdot0, (self.dof, rows), "qdot0"),
                                  self._arg(params, (abi.NPARAM, rows), "params"), self._arg(avg), self._arg(tq),
                                  self._arg(tqd), self._stream(stream))
        self._check(rc)
        return (avg, tq, tqd) if want_traj else avg

    def rollout_clock(self):
        """Shader clock of the last row-per-lane rollout launch, measured inside the kernel (mrf_rollout_clock; this
        synchronises the device): dict with the clock seen by the first / last workgroup [GHz] and their lifetimes [ms]."""
        v = (C.c_double * 5)()
        self._check(self.lib.mrf_rollout_clock(self._h, v, 5))
        ghz = [x for x in (v[0], v[1]) if x > 0]
        return {"shader_ghz_first_workgroup": v[0], "shader_ghz_last_workgroup": v[1], "first_workgroup_ms": v[2],
                "last_workgroup_ms": v[3], "wall_clock_ghz": v[4], "shader_ghz": sum(ghz) / len(ghz) if ghz else None}

    def rollout_cartesian(self, q0, qdot0, params, obst_x0, obst_v, obst_a, obst_r, want_traj=False, n_static=0,
                          stream=None):
        rows = q0.shape[1]
        H = self.cfg.horizon
        M = 0 if obst_x0 is None else obst_x0.shape[0]
        avg = torch.empty((rows,), dtype=self.dtype, device=self.device)
        tq = torch.empty((H, self.dof, rows), dtype=self.dtype, device=self.device) if want_traj else None
        tqd = torch.empty_like(tq) if want_traj else None
        rc = self.lib.mrf_rollout_cartesian(
            self._h, rows, self._arg(q0, (self.dof, rows), "q0"), self._arg(qdot0, (self.dof, rows), "qdot0"),
            self._arg(params, (abi.NPARAM, rows), "params"), M, n_static, self._arg(obst_x0, (M, 3, rows), "obst_x0"),
            self._arg(obst_v, (M, 3, rows), "obst_v"), self._arg(obst_a, (M, 3, rows), "obst_a"),
            self._arg(obst_r, (M, rows), "obst_r"), self._arg(avg), self._arg(tq), self._arg(tqd),
            self._stream(stream))
        self._check(rc)
        return (avg, tq, tqd) if want_traj else avg

    def rollout_cartesian_coupled(self, q0, qdot0, params, want_traj=False, stream=None):
        """Cartesian Rollout Fabrics of every robot against the other robots' configured spheres moving at constant
        velocity, obstacle assembly on the device (EXC:330-399); rows = n_scenarios * n_robots -> avg_vel [rows]."""
        rows = q0.shape[1]
        N, H = self.cfg.n_robots, self.cfg.horizon
        if rows % N:
            raise MrfError("rows must be a multiple of n_robots")
        avg = torch.empty((rows,), dtype=self.dtype, device=self.device)
        tq = torch.empty((H, self.dof, rows), dtype=self.dtype, device=self.device) if want_traj else None
        tqd = torch.empty_like(tq) if want_traj else None
        rc = self.lib.mrf_rollout_cartesian_coupled(self._h, rows // N, self._arg(q0, (self.dof, rows), "q0"),
                                                    self._arg(qdot0, (self.dof, rows), "qdot0"),
                                                    self._arg(params, (abi.NPARAM, rows), "params"), self._arg(avg),
                                                    self._arg(tq), self._arg(tqd), self._stream(stream))
        self._check(rc)
        return (avg, tq, tqd) if want_traj else avg

    def fk_spheres(self, q, qdot=None, stream=None):
        """-> x, v, a  each [S,3,rows]  (v, a None when qdot is None)."""
        rows = q.shape[1]
        S = self.cfg.n_spheres
        x = torch.empty((S, 3, rows), dtype=self.dtype, device=self.device)
        v = torch.empty_like(x) if qdot is not None else None
        a = torch.empty_like(x) if qdot is not None else None
        rc = self.lib.mrf_fk_spheres(self._h, rows, self._arg(q, (self.dof, rows), "q"),
                                     self._arg(qdot, (self.dof, rows), "qdot"), self._arg(x), self._arg(v),
                                     self._arg(a), self._stream(stream))
        self._check(rc)
        return x, v, a

    def rollout_sphere_traj(self, qdot0, traj_q, traj_qd, stream=None):
        """Sphere states every robot publishes at each step of a rollout (FPJ:211-225): x, v, a each [H, S, 3, rows]."""
        H, _, rows = traj_q.shape
        S = self.cfg.n_spheres
        x = torch.empty((H, S, 3, rows), dtype=self.dtype, device=self.device)
        v, a = torch.empty_like(x), torch.empty_like(x)
        rc = self.lib.mrf_rollout_sphere_traj(self._h, rows // self.cfg.n_robots, self._arg(qdot0, (self.dof, rows), "qdot0"),
                                              self._arg(traj_q, (self.cfg.horizon, self.dof, rows), "traj_q"),
                                              self._arg(traj_qd, (self.cfg.horizon, self.dof, rows), "traj_qd"),
                                              self._arg(x), self._arg(v), self._arg(a), self._stream(stream))
        self._check(rc)
        return x, v, a

    @property
    def exchange_spheres(self):
        """Spheres per robot in the sharded exchange buffers (coincident link origins travel once)."""
        return int(self.lib.mrf_exchange_spheres(self._h))

    def step_prepare(self, n_scen, robot_first, robot_count, q, qdot, params, stream=None):
        """-> params with the RF-CV goal estimate applied for the owned robots in cfg.goal_estimate_mask."""
        rows = n_scen * robot_count
        out = torch.empty_like(params)
        rc = self.lib.mrf_step_prepare(self._h, n_scen, robot_first, robot_count, self._arg(q, (self.dof, rows), "q"),
                                       self._arg(qdot, (self.dof, rows), "qdot"),
                                       self._arg(params, (abi.NPARAM, rows), "params"), self._arg(out),
                                       self._stream(stream))
        self._check(rc)
        return out

    def step_predict(self, n_scen, robot_first, robot_count, q_io, qdot, sph_own, stream=None):
        rows = n_scen * robot_count
        S = self.exchange_spheres
        rc = self.lib.mrf_step_predict(self._h, n_scen, robot_first, robot_count,
                                       self._arg(q_io, (self.dof, rows), "q_io"),
                                       self._arg(qdot, (self.dof, rows), "qdot"),
                                       self._arg(sph_own, (robot_count, S, 9, n_scen), "sph_own"),
                                       self._stream(stream))
        self._check(rc)

    def step_action(self, n_scen, robot_first, robot_count, q, qdot_io, params, sph_all, sumsq_io, stream=None):
        rows = n_scen * robot_count
        S, N = self.exchange_spheres, self.cfg.n_robots
        rc = self.lib.mrf_step_action(self._h, n_scen, robot_first, robot_count, self._arg(q, (self.dof, rows), "q"),
                                      self._arg(qdot_io, (self.dof, rows), "qdot_io"),
                                      self._arg(params, (abi.NPARAM, rows), "params"),
                                      self._arg(sph_all, (N, S, 9, n_scen), "sph_all"),
                                      self._arg(sumsq_io, (rows,), "sumsq_io"), self._stream(stream))
        self._check(rc)

    @property
    def exchange_scalars(self):
        """Scalars one robot puts on the wire per scenario and step under cfg.exchange: 21 (joints) or 9 * exchange_spheres."""
        return int(self.lib.mrf_exchange_scalars(self._h))

    def step_predict_joints(self, n_scen, robot_first, robot_count, q_io, qdot, jst_own, stream=None):
        """q_io += dt*qdot for the owned robots; cos q, sin q, qdot -> jst_own [count, 21, n_scen] (MRF_EXCHANGE_JOINTS)."""
        rows = n_scen * robot_count
        rc = self.lib.mrf_step_predict_joints(self._h, n_scen, robot_first, robot_count,
                                              self._arg(q_io, (self.dof, rows), "q_io"),
                                              self._arg(qdot, (self.dof, rows), "qdot"),
                                              self._arg(jst_own, (robot_count, abi.JOINT_STATE_SCALARS, n_scen), "jst_own"),
                                              self._stream(stream))
        self._check(rc)

    def step_action_joints(self, n_scen, robot_first, robot_count, q, qdot_io, params, jst_all, sumsq_io, stream=None):
        """Fabric solve of the owned robots: the owned robots exchange on chip, the others are re-walked from jst_all
        [n_robots, 21, n_scen]; qdot_io := action, sumsq_io += |action|^2."""
        rows = n_scen * robot_count
        rc = self.lib.mrf_step_action_joints(self._h, n_scen, robot_first, robot_count, self._arg(q, (self.dof, rows), "q"),
                                             self._arg(qdot_io, (self.dof, rows), "qdot_io"),
                                             self._arg(params, (abi.NPARAM, rows), "params"),
                                             self._arg(jst_all, (self.cfg.n_robots, abi.JOINT_STATE_SCALARS, n_scen), "jst_all"),
                                             self._arg(sumsq_io, (rows,), "sumsq_io"), self._stream(stream))
        self._check(rc)

    def step_action_predict_joints(self, n_scen, robot_first, robot_count, q_io, qdot_io, params, jst_all, sumsq_io, jst_next_own,
                                   stream=None):
        """step_action_joints of this step AND step_predict_joints of the following one in one launch: q_io += dt * action,
        the next joint state -> jst_next_own [count, 21, n_scen] (the block the next all-gather sends)."""
        rows = n_scen * robot_count
        rc = self.lib.mrf_step_action_predict_joints(
            self._h, n_scen, robot_first, robot_count, self._arg(q_io, (self.dof, rows), "q_io"),
            self._arg(qdot_io, (self.dof, rows), "qdot_io"), self._arg(params, (abi.NPARAM, rows), "params"),
            self._arg(jst_all, (self.cfg.n_robots, abi.JOINT_STATE_SCALARS, n_scen), "jst_all"),
            self._arg(sumsq_io, (rows,), "sumsq_io"),
            self._arg(jst_next_own, (robot_count, abi.JOINT_STATE_SCALARS, n_scen), "jst_next_own"), self._stream(stream))
        self._check(rc)

    # ------------------------------------------------------------------ host-buffer entry points (numpy in, numpy out)
    @staticmethod
    def _host(a, shape, name):
        """float64 C-contiguous numpy array of the given shape (converted if needed) -> (array, pointer)."""
        import numpy as np
        if a is None:
            return None, None
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != tuple(shape):
            raise MrfError(f"{name}: expected shape {tuple(shape)}, got {a.shape}")
        return a, C.c_void_p(a.ctypes.data)

    def compute_action_host(self, q, qdot, params, obst_x=None, obst_v=None, obst_a=None, obst_r=None, n_static=0,
                            want_qddot=False):
        """compute_action for host arrays (layouts of compute_action): one packed copy each way inside the library, no
        torch tensors -- the path of the reference-shaped single-scenario calls.  -> action [dof, rows] (numpy)."""
        import numpy as np
        q, pq = self._host(q, (self.dof, np.shape(q)[1]), "q")
        rows = q.shape[1]
        M = 0 if obst_x is None else np.shape(obst_x)[0]
        qd, pqd = self._host(qdot, (self.dof, rows), "qdot")
        prm, pp = self._host(params, (abi.NPARAM, rows), "params")
        ox, pox = self._host(obst_x, (M, 3, rows), "obst_x")
        ov, pov = self._host(obst_v, (M, 3, rows), "obst_v")
        oa, poa = self._host(obst_a, (M, 3, rows), "obst_a")
        orad, por = self._host(obst_r, (M, rows), "obst_r")
        act = np.empty((self.dof, rows))
        qdd = np.empty((self.dof, rows)) if want_qddot else None
        rc = self.lib.mrf_compute_action_host(self._h, rows, pq, pqd, pp, M, n_static, pox, pov, poa, por,
                                              None if qdd is None else C.c_void_p(qdd.ctypes.data), C.c_void_p(act.ctypes.data))
        self._check(rc)
        return (act, qdd) if want_qddot else act

    def rollout_host(self, q0, qdot0, params, want_traj=False):
        import numpy as np
        q, pq = self._host(q0, (self.dof, np.shape(q0)[1]), "q0")
        rows = q.shape[1]
        N, H = self.cfg.n_robots, self.cfg.horizon
        if rows % N:
            raise MrfError("rows must be a multiple of n_robots")
        qd, pqd = self._host(qdot0, (self.dof, rows), "qdot0")
        prm, pp = self._host(params, (abi.NPARAM, rows), "params")
        avg = np.empty((rows,))
        tq = np.empty((H, self.dof, rows)) if want_traj else None
        tqd = np.empty((H, self.dof, rows)) if want_traj else None
        p = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
        self._check(self.lib.mrf_rollout_host(self._h, rows // N, pq, pqd, pp, p(avg), p(tq), p(tqd)))
        return (avg, tq, tqd) if want_traj else avg

    def rollout_cartesian_host(self, q0, qdot0, params, obst_x0, obst_v, obst_a, obst_r, want_traj=False, n_static=0):
        import numpy as np
        q, pq = self._host(q0, (self.dof, np.shape(q0)[1]), "q0")
        rows = q.shape[1]
        H = self.cfg.horizon
        M = 0 if obst_x0 is None else np.shape(obst_x0)[0]
        qd, pqd = self._host(qdot0, (self.dof, rows), "qdot0")
        prm, pp = self._host(params, (abi.NPARAM, rows), "params")
        ox, pox = self._host(obst_x0, (M, 3, rows), "obst_x0")
        ov, pov = self._host(obst_v, (M, 3, rows), "obst_v")
        oa, poa = self._host(obst_a, (M, 3, rows), "obst_a")
        orad, por = self._host(obst_r, (M, rows), "obst_r")
        avg = np.empty((rows,))
        tq = np.empty((H, self.dof, rows)) if want_traj else None
        tqd = np.empty((H, self.dof, rows)) if want_traj else None
        p = lambda a: None if a is None else C.c_void_p(a.ctypes.data)
        self._check(self.lib.mrf_rollout_cartesian_host(self._h, rows, pq, pqd, pp, M, n_static, pox, pov, poa, por,
                                                        p(avg), p(tq), p(tqd)))
        return (avg, tq, tqd) if want_traj else avg

    def fk_spheres_host(self, q, qdot=None):
        import numpy as np
        q, pq = self._host(q, (self.dof, np.shape(q)[1]), "q")
        rows = q.shape[1]
        S = self.cfg.n_spheres
        qd, pqd = self._host(qdot, (self.dof, rows), "qdot")
        x = np.empty((S, 3, rows))
        v = np.empty((S, 3, rows)) if qd is not None else None
        a = np.empty((S, 3, rows)) if qd is not None else None
        p = lambda t: None if t is None else C.c_void_p(t.ctypes.data)
        self._check(self.lib.mrf_fk_spheres_host(self._h, rows, pq, pqd, p(x), p(v), p(a)))
        return x, v, a

    # ------------------------------------------------------------------ robot-sharded rollout inside the library
    def comm_unique_id(self):
        """Rank 0 of a robot group: a fresh RCCL communicator id (bytes) to hand to every rank's comm_init_rccl."""
        buf = (C.c_ubyte * abi.COMM_ID_BYTES)()
        rc = self.lib.mrf_comm_unique_id(buf)
        if rc != 0:
            raise MrfError("mrf_comm_unique_id failed (librccl not loadable?)")
        return bytes(buf)

    def comm_init_rccl(self, rank, world, unique_id=None):
        """RCCL transport (ncclAllGather per rollout step, issued from C++ on the caller's stream)."""
        uid = None if unique_id is None else (C.c_ubyte * abi.COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self.lib.mrf_comm_init(self._h, rank, world, uid))

    def comm_peer_open(self, rank, world, max_scenarios):
        """PEER transport, phase 1: allocates the exchange buffer, returns this rank's IPC handle (bytes)."""
        buf = (C.c_ubyte * abi.IPC_HANDLE_BYTES)()
        self._check(self.lib.mrf_comm_peer_open(self._h, rank, world, int(max_scenarios), buf))
        return bytes(buf)

    def comm_peer_connect(self, handles):
        """PEER transport, phase 2: `handles` = every rank's IPC handle in rank order."""
        blob = b"".join(handles)
        self._check(self.lib.mrf_comm_peer_connect(self._h, (C.c_ubyte * len(blob)).from_buffer_copy(blob)))

    def comm_peer_local_base(self):
        """Device pointer (int) of this rank's exchange buffer, for groups whose ranks live in one process."""
        base = C.c_void_p()
        if self.lib.mrf_comm_peer_local_base(self._h, C.byref(base)) != 0:
            raise MrfError("mrf_comm_peer_local_base: comm_peer_open first")
        return int(base.value)

    def comm_peer_connect_local(self, bases):
        """PEER transport inside ONE process: `bases` = comm_peer_local_base() of every rank's handle, in rank order."""
        arr = (C.c_void_p * len(bases))(*[C.c_void_p(b) for b in bases])
        self._check(self.lib.mrf_comm_peer_connect_local(self._h, arr))

    def comm_info(self):
        """What the communicator itself reports (mrf_comm_info): transport, rank, world, robot block, and -- RCCL -- the
        rank count / user rank / device ncclCommCount, ncclCommUserRank and ncclCommCuDevice return."""
        vals = (C.c_int32 * len(abi.COMM_INFO_KEYS))()
        if self.lib.mrf_comm_info(self._h, vals, len(abi.COMM_INFO_KEYS)) != 0:
            raise MrfError("mrf_comm_info failed")
        out = dict(zip(abi.COMM_INFO_KEYS, (int(v) for v in vals)))
        out["transport"] = {abi.TRANSPORT_NONE: "none", abi.TRANSPORT_RCCL: "rccl", abi.TRANSPORT_PEER: "peer"}[out["transport"]]
        out["exchange"] = abi.EXCHANGE_NAMES.get(out["exchange"], str(out["exchange"]))
        return out

    def comm_peer_info(self):
        """Connected PEER communicator: per rank of the group, where its exchange buffer really is as seen from this rank --
        owning device of the mapping, hipDeviceCanAccessPeer, link type and hop count (mrf_comm_peer_info)."""
        world = self.comm_info()["world"]
        vals = (C.c_int32 * (world * len(abi.PEER_INFO_KEYS)))()
        if self.lib.mrf_comm_peer_info(self._h, vals, len(vals)) != 0:
            raise MrfError("mrf_comm_peer_info: no connected PEER communicator")
        n = len(abi.PEER_INFO_KEYS)
        out = []
        for g in range(world):
            d = dict(zip(abi.PEER_INFO_KEYS, (int(v) for v in vals[g * n:(g + 1) * n])))
            d["rank"] = g
            d["link"] = abi.LINK_TYPE_NAMES.get(d["link_type"], str(d["link_type"]))
            out.append(d)
        return out

    def comm_partition(self):
        first, count = C.c_int32(), C.c_int32()
        if self.lib.mrf_comm_partition(self._h, C.byref(first), C.byref(count)) != 0:
            raise MrfError("no communicator")
        return first.value, count.value

    def rollout_sharded(self, q_io, qdot_io, params, stream=None):
        """H rollout steps of the owned robots with the per-step exchange inside the library; q_io / qdot_io
        [7, n_scen*count] advance in place -> avg_vel [n_scen*count]."""
        first, count = self.comm_partition()
        rows = q_io.shape[1]
        if rows % count:
            raise MrfError("rows must be a multiple of the owned robot count")
        avg = torch.empty((rows,), dtype=self.dtype, device=self.device)
        rc = self.lib.mrf_rollout_sharded(self._h, rows // count, self._arg(q_io, (self.dof, rows), "q_io"),
                                          self._arg(qdot_io, (self.dof, rows), "qdot_io"),
                                          self._arg(params, (abi.NPARAM, rows), "params"), self._arg(avg),
                                          self._stream(stream))
        self._check(rc)
        return avg

    def comm_status(self):
        """Synchronises the last sharded rollout; raises if a peer exchange timed out."""
        self._check(self.lib.mrf_comm_status(self._h))

    def comm_reset(self):
        """Clears a timed-out peer exchange (every rank of the group, after a barrier of the caller's)."""
        self._check(self.lib.mrf_comm_reset(self._h))

    def comm_destroy(self):
        self.lib.mrf_comm_destroy(self._h)

    # ------------------------------------------------------------------ device-resident control step (include/mrf.h)
    def control_prepare(self, q, qdot, params_nominal, params_work, apply_estimate=True, stream=None):
        """Hand FK + RF-CV goal estimate (EXJ:325-329,346-348) -> x_ee [3,rows]; fills params_work."""
        rows = q.shape[1]
        x_ee = torch.empty((3, rows), dtype=self.dtype, device=self.device)
        rc = self.lib.mrf_control_prepare(self._h, rows // self.cfg.n_robots, self._arg(q, (self.dof, rows), "q"),
                                          self._arg(qdot, (self.dof, rows), "qdot"),
                                          self._arg(params_nominal, (abi.NPARAM, rows), "params_nominal"),
                                          self._arg(params_work, (abi.NPARAM, rows), "params_work"),
                                          int(bool(apply_estimate)), self._arg(x_ee), self._stream(stream))
        self._check(rc)
        return x_ee

    def deadlock_state(self, n_scen, stream=None):
        """Fresh per-scenario deadlock state: (int32 [DL_NSTATE, n_scen], goal [3, n_scen])  (DP:9-34, EXJ:273)."""
        st = torch.empty((abi.DL_NSTATE, n_scen), dtype=torch.int32, device=self.device)
        goal = torch.empty((3, n_scen), dtype=self.dtype, device=self.device)
        self._check(self.lib.mrf_deadlock_init(self._h, n_scen, C.c_void_p(st.data_ptr()), self._arg(goal),
                                               self._stream(stream)))
        return st, goal

    def deadlock_config(self, point_mass=False):
        c = abi.DeadlockConfig()
        self.lib.mrf_default_deadlock_config(C.byref(c), int(bool(point_mass)))
        return c

    @staticmethod
    def _i32(t, shape, name):
        if t is None:
            return None
        if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != tuple(shape):
            raise MrfError(f"{name}: expected a contiguous int32 device tensor of shape {tuple(shape)}")
        return C.c_void_p(t.data_ptr())

    def deadlock_step(self, dl_cfg, x_ee, avg_vel, params_work, dl_state, dl_goal, sm_state=None, time_step=-1,
                      stream=None):
        """deadlock_checking (DP:50-118) for every scenario; rewrites x_goal_0 / weight_goal_0 rows of params_work."""
        rows = x_ee.shape[1]
        n_scen = rows // self.cfg.n_robots
        rc = self.lib.mrf_deadlock_step(self._h, n_scen, C.byref(dl_cfg), int(time_step), self._arg(x_ee, (3, rows), "x_ee"),
                                        self._arg(avg_vel, (rows,), "avg_vel"), self._i32(sm_state, (rows,), "sm_state"),
                                        self._arg(params_work, (abi.NPARAM, rows), "params_work"),
                                        self._i32(dl_state, (abi.DL_NSTATE, n_scen), "dl_state"),
                                        self._arg(dl_goal, (3, n_scen), "dl_goal"), self._stream(stream))
        self._check(rc)

    # ------------------------------------------------------------------ pick-and-place state machine (SM:133-214)
    def state_machine_config(self, nr_blocks, model=0):
        c = abi.StateMachineConfig()
        self.lib.mrf_default_state_machine_config(C.byref(c), int(nr_blocks))
        c.model = int(model)
        return c

    def state_machine_state(self, start_goal, stream=None):
        """Fresh per-row state for start goals [3, rows]: (int32 [SM_NSTATE, rows], goal [SM_NGOAL, rows])."""
        rows = start_goal.shape[1]
        st = torch.empty((abi.SM_NSTATE, rows), dtype=torch.int32, device=self.device)
        sg = torch.empty((abi.SM_NGOAL, rows), dtype=self.dtype, device=self.device)
        self._check(self.lib.mrf_state_machine_init(self._h, rows, self._arg(start_goal, (3, rows), "start_goal"),
                                                    C.c_void_p(st.data_ptr()), self._arg(sg), self._stream(stream)))
        return st, sg

    def state_machine_step(self, sm_cfg, x_ee, start_goal, blocks, q_gripper_io, sm_state, sm_goal, params_work=None,
                           skip_robot_mask=0, stream=None):
        """get_state_machine_panda + get_gripper_action_panda for every row -> gripper velocity command [2, rows]."""
        rows = x_ee.shape[1]
        nb = blocks.shape[0]
        act = torch.empty((2, rows), dtype=self.dtype, device=self.device)
        rc = self.lib.mrf_state_machine_step(
            self._h, rows, C.byref(sm_cfg), self._arg(x_ee, (3, rows), "x_ee"), self._arg(start_goal, (3, rows), "start_goal"),
            self._arg(blocks, (nb, 3, rows), "blocks"), nb, self._arg(q_gripper_io, (2, rows), "q_gripper_io"),
            self._i32(sm_state, (abi.SM_NSTATE, rows), "sm_state"), self._arg(sm_goal, (abi.SM_NGOAL, rows), "sm_goal"),
            self._arg(params_work, (abi.NPARAM, rows), "params_work"), int(skip_robot_mask), self._arg(act),
            self._stream(stream))
        self._check(rc)
        return act

    def apply_action(self, q_io, qdot_io, action_io, vel_limit, stop_margin=-1.0, stream=None):
        """action := clip(action, +-vel_limit); q += dt*action; qdot := action  (EXJ:452-453)."""
        rows = q_io.shape[1]
        vl = (C.c_double * 7)(*[float(v) for v in vel_limit])
        rc = self.lib.mrf_apply_action(self._h, rows, self._arg(q_io, (self.dof, rows), "q_io"),
                                       self._arg(qdot_io, (self.dof, rows), "qdot_io"),
                                       self._arg(action_io, (self.dof, rows), "action_io"), vl, float(stop_margin),
                                       self._stream(stream))
        self._check(rc)


class ControlLoop:
    """n control steps on the device without a host round trip (mrf_episode_run): per step
    prepare -> Rollout Fabrics -> deadlock logic -> coupled compute_action -> apply (EXJ:280-458 minus the simulator).

    `h_rollout` may be None (plain MRDF: no rollouts, no deadlock logic); `deadlock=False` keeps the rollouts as a
    monitor only.  State (q, qdot, deadlock state) lives in this object's device tensors and is advanced in place."""

    def __init__(self, h_action, h_rollout, q, qdot, params, vel_limit, deadlock=True, apply_estimate=True,
                 stop_margin=1e-3, sm_state=None, use_graph=True, pick_place=None, cartesian_rollouts=False):
        """pick_place: dict(start_goal [3,rows], blocks [nb,3,rows] (already lifted by 0.1, EXJ:303), nr_blocks,
        q_gripper [2,rows], model=1, h_grasp=None) -- runs the pick-and-place state machine on the device every step
        (mrf_episode_set_pick_place); its state is then in self.sm_state / self.sm_goal / self.q_gripper."""
        self.ha, self.hr = h_action, h_rollout
        # cartesian_rollouts: the rollouts are the per-robot constant-velocity ones of example_pandas_cartesian.py
        # (mrf_rollout_cartesian_coupled) instead of the coupled joint-space rollout
        self.rollout_kind = abi.ROLLOUT_CARTESIAN if cartesian_rollouts else abi.ROLLOUT_JOINTSPACE
        if h_rollout is not None and (h_rollout.dtype != h_action.dtype or h_rollout.device != h_action.device or
                                      h_rollout.cfg.n_robots != h_action.cfg.n_robots):
            raise MrfError("rollout and action handles must agree in scalar type, device and n_robots")
        if q.dim() != 2 or q.shape[0] != h_action.dof or q.shape[1] % h_action.cfg.n_robots:
            raise MrfError(f"q: expected [{h_action.dof}, n_scenarios * {h_action.cfg.n_robots}], got {tuple(q.shape)}")
        rows = q.shape[1]
        # own, contiguous copies of the state (advanced in place); `params` is NOT copied: the caller may rewrite goals
        # between run() calls (the replayed graph reads the same buffer), so it must already be a valid device array
        self.q, self.qdot = q.clone().contiguous(), qdot.clone().contiguous()
        self.params = params
        h_action._arg(self.q, (h_action.dof, rows), "q")
        h_action._arg(self.qdot, (h_action.dof, rows), "qdot")
        h_action._arg(self.params, (abi.NPARAM, rows), "params")
        if len(vel_limit) != h_action.dof:
            raise MrfError(f"vel_limit: expected {h_action.dof} values")
        self.n_scen = rows // h_action.cfg.n_robots
        self.params_work = torch.empty_like(params)
        self.x_ee = torch.empty((3, rows), dtype=h_action.dtype, device=h_action.device)
        self.avg = torch.zeros((rows,), dtype=h_action.dtype, device=h_action.device)
        self.action = torch.zeros_like(q)
        self.vel_limit = (C.c_double * 7)(*[float(v) for v in vel_limit])
        self.stop_margin = float(stop_margin)
        self.apply_estimate = bool(apply_estimate)
        self.sm_state = sm_state
        FabricHandle._i32(sm_state, (rows,), "sm_state")
        if sm_state is not None and sm_state.device != h_action.device:
            raise MrfError(f"sm_state lives on {sm_state.device}, the handles on {h_action.device}")
        self.use_graph = bool(use_graph)
        self.pick_place = None
        if pick_place is not None:
            pp = dict(pick_place)
            h = h_action
            # start_goal and blocks are read IN PLACE by every run() (a caller in model 0 rewrites `blocks` with new
            # observations between calls): no private copy is made, so they must already be valid device arrays
            self.start_goal, self.blocks = pp["start_goal"], pp["blocks"]
            if self.blocks.dim() != 3:
                raise MrfError(f"blocks: expected [n_block_arrays, 3, {rows}], got {tuple(self.blocks.shape)}")
            h._arg(self.start_goal, (3, rows), "start_goal")
            self.q_gripper = pp["q_gripper"].clone().contiguous()
            self.sm_cfg = h.state_machine_config(pp["nr_blocks"], model=pp.get("model", 1))
            self.sm_state, self.sm_goal = h.state_machine_state(self.start_goal)
            self.gripper_action = torch.zeros((2, rows), dtype=h.dtype, device=h.device)
            self.h_grasp = pp.get("h_grasp")
            self.action_grasp = torch.zeros_like(self.q) if self.h_grasp is not None else None
            nb = self.blocks.shape[0]
            h._arg(self.blocks, (nb, 3, rows), "blocks")
            h._arg(self.q_gripper, (2, rows), "q_gripper")
            self.pick_place = True
        self.dl_cfg = h_action.deadlock_config() if (deadlock and h_rollout is not None) else None
        self.dl_state, self.dl_goal = (h_rollout.deadlock_state(self.n_scen) if self.dl_cfg is not None else (None, None))
        self.recorder = None

    def attach_recorder(self, capacity, done_state=10, keep_q=True):
        """Have the control steps themselves record what a host loop would read back after every step
        (mrf_episode_set_recorder), for the next `capacity` steps: -> dict of device tensors q_hist [capacity, dof, rows]
        (None without keep_q), sm_hist [capacity, rows], t_begin / t_end [capacity] (wall-clock ticks), done_at [rows]
        (first step with state == done_state, -1 before), counter [1].  run(n) can then queue n steps back to back."""
        h, rows = self.ha, self.q.shape[1]
        dev = h.device
        self.recorder = dict(
            q_hist=torch.empty((capacity, h.dof, rows), dtype=h.dtype, device=dev) if keep_q else None,
            sm_hist=torch.zeros((capacity, rows), dtype=torch.int32, device=dev),
            t_begin=torch.zeros((capacity,), dtype=torch.int64, device=dev), t_end=torch.zeros((capacity,), dtype=torch.int64, device=dev),
            done_at=torch.full((rows,), -1, dtype=torch.int32, device=dev), counter=torch.zeros((1,), dtype=torch.int32, device=dev),
            capacity=int(capacity), done_state=int(done_state))
        return self.recorder

    def run(self, n_steps, stream=None):
        ha, hr = self.ha, self.hr
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        # (re)attach this loop's pick-and-place buffers: the attachment lives in the action handle, which loops may share
        if self.pick_place:
            ha._check(ha.lib.mrf_episode_set_pick_place(
                ha._h, C.byref(self.sm_cfg), p(self.start_goal), p(self.blocks), self.blocks.shape[0], p(self.q_gripper),
                p(self.sm_state), p(self.sm_goal), p(self.gripper_action),
                self.h_grasp._h if self.h_grasp is not None else None, p(self.action_grasp)))
        else:
            ha._check(ha.lib.mrf_episode_set_pick_place(ha._h, None, None, None, 0, None, None, None, None, None, None))
        if hr is not None:
            hr._check(hr.lib.mrf_episode_set_rollout(hr._h, self.rollout_kind))
        rec = self.recorder
        if rec is not None:
            ha._check(ha.lib.mrf_episode_set_recorder(ha._h, p(rec["q_hist"]), p(rec["sm_hist"]), p(rec["t_begin"]), p(rec["t_end"]),
                                                      p(rec["done_at"]), p(rec["counter"]), rec["capacity"], rec["done_state"]))
        else:
            ha._check(ha.lib.mrf_episode_set_recorder(ha._h, None, None, None, None, None, None, 0, 0))
        rc = ha.lib.mrf_episode_run(hr._h if hr is not None else None, ha._h, self.n_scen, int(n_steps),
                                    C.byref(self.dl_cfg) if self.dl_cfg is not None else None, int(self.apply_estimate),
                                    self.vel_limit, self.stop_margin, p(self.q), p(self.qdot), p(self.params),
                                    p(self.params_work), p(self.sm_state), p(self.dl_state), p(self.dl_goal), p(self.x_ee),
                                    p(self.avg), p(self.action), int(self.use_graph), ha._stream(stream))
        ha._check(rc)
        return self.action

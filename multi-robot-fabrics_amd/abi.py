"""ctypes mirror of include/mrf.h and the loader of the HIP shared library.

The product path has no CPU fallback: if csrc/libmrf_hip.so is missing or cannot be loaded,
`load_library()` raises, and every planner / rollout class raises with it.
"""
import ctypes as C
import os

MRF_ABI_VERSION = 6
MRF_MAX_ROBOTS = 16
MRF_MAX_SPHERES = 32
MRF_DOF_MAX = 7
MRF_N_EGO = 6

# per-row parameter vector (include/mrf.h)
P_X_GOAL_0 = 0
P_WEIGHT_GOAL_0 = 3
P_ANGLE_GOAL_1 = 4
P_X_GOAL_1 = 13
P_WEIGHT_GOAL_1 = 16
P_X_GOAL_2 = 17
P_WEIGHT_GOAL_2 = 18
P_CONSTRAINT_0 = 19
P_RADIUS_BODY = 23
NPARAM = 29

MODEL_PANDA7, MODEL_PLANAR3 = 0, 1
MODE_ACC, MODE_VEL = 0, 1
F64, F32 = 0, 1
FAMILY_POW, FAMILY_LOGISTIC = 0, 1
GATE_NONE, GATE_NEG = 0, 1

STATUS_TEXT = {0: "MRF_OK", -1: "MRF_E_ARG", -2: "MRF_E_CONFIG", -3: "MRF_E_DEVICE", -4: "MRF_E_LAUNCH"}


class LeafFn(C.Structure):
    _fields_ = [("family", C.c_int32), ("gate", C.c_int32), ("p", C.c_int32), ("reserved", C.c_int32),
                ("k", C.c_double), ("c", C.c_double), ("s", C.c_double)]

    def as_tuple(self):
        return (self.family, self.gate, self.p, self.k, self.c, self.s)


class Config(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("model", C.c_int32), ("scalar", C.c_int32), ("mode", C.c_int32),
        ("n_robots", C.c_int32), ("n_spheres", C.c_int32), ("horizon", C.c_int32), ("dynamic", C.c_int32),
        ("n_ego", C.c_int32), ("n_planes", C.c_int32), ("use_limits", C.c_int32), ("n_goals", C.c_int32),
        ("plane_abs", C.c_int32), ("zero_small_action", C.c_int32), ("obst_dim", C.c_int32),
        ("goal_estimate_mask", C.c_int32),
        ("dt", C.c_double), ("eps", C.c_double), ("jdot_sign", C.c_double), ("goal_estimate_T", C.c_double),
        ("base_mass", C.c_double),
        ("attr_k", C.c_double), ("attr_alpha", C.c_double),
        ("attr_mu", C.c_double), ("attr_ml", C.c_double), ("attr_a", C.c_double),
        ("beta_a", C.c_double), ("beta_r", C.c_double), ("beta_b", C.c_double), ("beta_s", C.c_double),
        ("eta_a", C.c_double), ("eta_s", C.c_double),
        ("mount", (C.c_double * 12) * MRF_MAX_ROBOTS),
        ("limits", (C.c_double * 2) * MRF_DOF_MAX),
        ("sphere_link", C.c_int32 * MRF_MAX_SPHERES),
        ("sphere_offset", (C.c_double * 3) * MRF_MAX_SPHERES),
        ("sphere_radius", C.c_double * MRF_MAX_SPHERES),
        ("collision_geometry", LeafFn), ("collision_finsler", LeafFn),
        ("plane_geometry", LeafFn), ("plane_finsler", LeafFn),
        ("limit_geometry", LeafFn), ("limit_finsler", LeafFn),
        ("kernel_select", C.c_int32), ("ego_link_mask", C.c_int32),
        ("exchange", C.c_int32), ("reserved_tail", C.c_int32),
    ]

    def copy(self):
        out = Config()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(Config))
        return out


_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmrf_hip.so")

EXPORTS = [
    "mrf_default_config_panda", "mrf_default_config_planar3", "mrf_create", "mrf_destroy", "mrf_last_error",
    "mrf_abi_version", "mrf_build_has_f32", "mrf_build_has_wp", "mrf_config_sizeof", "mrf_compute_action", "mrf_compute_action_coupled", "mrf_rollout", "mrf_rollout_cartesian",
    "mrf_fk_spheres", "mrf_rollout_sphere_traj", "mrf_exchange_spheres", "mrf_step_prepare", "mrf_step_predict", "mrf_step_action",
    "mrf_default_deadlock_config", "mrf_deadlock_config_sizeof", "mrf_deadlock_init", "mrf_control_prepare", "mrf_deadlock_step", "mrf_apply_action",
    "mrf_episode_run",
    "mrf_compute_action_host", "mrf_rollout_host", "mrf_rollout_cartesian_host", "mrf_fk_spheres_host",
    "mrf_default_state_machine_config", "mrf_state_machine_config_sizeof", "mrf_state_machine_init", "mrf_state_machine_step",
    "mrf_episode_set_pick_place", "mrf_rollout_cartesian_coupled", "mrf_episode_set_rollout", "mrf_rollout_clock", "mrf_episode_set_recorder",
    "mrf_comm_unique_id", "mrf_comm_init", "mrf_comm_peer_open", "mrf_comm_peer_connect", "mrf_comm_partition",
    "mrf_comm_info", "mrf_comm_transport", "mrf_rollout_sharded", "mrf_comm_status", "mrf_comm_reset", "mrf_comm_destroy",
    "mrf_step_predict_joints", "mrf_step_action_joints", "mrf_step_action_predict_joints", "mrf_exchange_scalars", "mrf_comm_peer_info", "mrf_device_topology",
    "mrf_comm_peer_local_base", "mrf_comm_peer_connect_local", "mrf_streams_concurrent",
]

ROLLOUT_JOINTSPACE, ROLLOUT_CARTESIAN = 0, 1
TRANSPORT_NONE, TRANSPORT_RCCL, TRANSPORT_PEER = 0, 1, 2
EXCHANGE_JOINTS, EXCHANGE_SPHERES = 0, 1        # mrf_config.exchange: what a robot-sharded rollout puts on the wire
EXCHANGE_NAMES = {EXCHANGE_JOINTS: "joints", EXCHANGE_SPHERES: "spheres"}
JOINT_STATE_SCALARS = 21
COMM_ID_BYTES, IPC_HANDLE_BYTES = 128, 64
COMM_INFO_KEYS = ("transport", "rank", "world", "robot_first", "robot_count", "rccl_comm_count", "rccl_user_rank",
                  "rccl_device", "hip_device", "peer_buffers_mapped", "exchange", "exchange_scalars_per_robot",
                  "peers_one_hop", "coresident_workgroups", "paired_blocks", "tagged_payload")
PEER_INFO_KEYS = ("device", "can_access_peer", "link_type", "hops")
LINK_TYPE_NAMES = {0: "same device", 1: "hypertransport", 2: "pcie", 3: "qpi", 4: "xgmi", -1: "unknown"}

# rows of the int32 deadlock state (include/mrf.h MRF_DL_*)
DL_LEADER, DL_FOLLOWER, DL_DEAD0, DL_DEAD1, DL_TIME_IN_DEADLOCK, DL_TIME_DEADLOCK_OUT, DL_TIME_STEP, DL_NONFINITE, DL_NSTATE = range(9)


class DeadlockConfig(C.Structure):
    """ctypes mirror of mrf_deadlock_config (thresholds of deadlock_prevention.py:12-27, literals of :50-118)."""
    _fields_ = [("avg_vel_constant", C.c_double), ("dist_constant", C.c_double), ("goal_weight_follower", C.c_double),
                ("goal_weight_leader", C.c_double), ("nr_goal_scale", C.c_double), ("ee_distance", C.c_double),
                ("follower_offset", C.c_double), ("min_goal_norm", C.c_double), ("z_floor", C.c_double),
                ("time_wait", C.c_int32), ("min_time_step", C.c_int32), ("grasp_state", C.c_int32),
                ("grasp_timeout", C.c_int32)]

# rows of the pick-and-place state (include/mrf.h MRF_SM_*)
SM_STATE, SM_PICKED, SM_FAILED, SM_T_GRIP, SM_GRIPPER, SM_STOP, SM_NSTATE = range(7)
SM_GOAL, SM_GOAL_ABOVE, SM_WEIGHT, SM_NGOAL = 0, 3, 6, 7


class StateMachineConfig(C.Structure):
    """ctypes mirror of mrf_state_machine_config (literals of state_machine.py:133-214 and :66-86)."""
    _fields_ = [("reach_home", C.c_double), ("reach_pregrasp", C.c_double), ("reach_block", C.c_double),
                ("reach_lift", C.c_double), ("reach_drop", C.c_double), ("pregrasp_height", C.c_double),
                ("lift_height", C.c_double), ("grip_steps", C.c_double), ("open_tol", C.c_double),
                ("dropped_below_z", C.c_double), ("weight_high", C.c_double), ("weight_low", C.c_double),
                ("gripper_open", C.c_double * 2), ("v_close", C.c_double), ("v_open", C.c_double),
                ("nr_blocks", C.c_int32), ("model", C.c_int32)]


_lib = None


class MrfLibraryError(RuntimeError):
    pass


def load_library(path=None):
    """Load csrc/libmrf_hip.so (built by __graft_entry__.build()).  Raises if it is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("MRF_HIP_LIB") or LIB_PATH  # MRF_HIP_LIB: A/B-ing kernel builds (tools/)
    if not os.path.exists(p):
        raise MrfLibraryError(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the fabric solve.")
    # torch ships its own libamdhip64; load it first so that this library binds to the same HIP runtime the
    # tensors live in (two runtimes in one process do not see each other's devices / allocations)
    import torch  # noqa: F401
    lib = C.CDLL(p)
    vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
    lib.mrf_default_config_panda.argtypes = [C.POINTER(Config), i32, i32]
    lib.mrf_default_config_panda.restype = None
    lib.mrf_default_config_planar3.argtypes = [C.POINTER(Config), i32]
    lib.mrf_default_config_planar3.restype = None
    lib.mrf_create.argtypes = [C.POINTER(Config), i32, C.POINTER(vp)]
    lib.mrf_create.restype = C.c_int
    lib.mrf_destroy.argtypes = [vp]
    lib.mrf_destroy.restype = None
    lib.mrf_last_error.argtypes = [vp]
    lib.mrf_last_error.restype = C.c_char_p
    lib.mrf_abi_version.argtypes = []
    lib.mrf_abi_version.restype = C.c_int
    if hasattr(lib, "mrf_build_has_f32"):      # absent in older builds under A/B timing (MRF_HIP_LIB + MRF_ABI_ANY)
        lib.mrf_build_has_f32.argtypes = []
        lib.mrf_build_has_f32.restype = C.c_int
    lib.mrf_config_sizeof.argtypes = []
    lib.mrf_config_sizeof.restype = C.c_int64
    lib.mrf_compute_action.argtypes = [vp, i64, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.mrf_compute_action.restype = C.c_int
    lib.mrf_compute_action_coupled.argtypes = [vp, i64, vp, vp, vp, i32, vp, vp, vp]
    lib.mrf_compute_action_coupled.restype = C.c_int
    lib.mrf_rollout.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp]
    lib.mrf_rollout.restype = C.c_int
    lib.mrf_rollout_cartesian.argtypes = [vp, i64, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.mrf_rollout_cartesian.restype = C.c_int
    lib.mrf_fk_spheres.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    lib.mrf_fk_spheres.restype = C.c_int
    lib.mrf_rollout_sphere_traj.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp]
    lib.mrf_rollout_sphere_traj.restype = C.c_int
    lib.mrf_exchange_spheres.argtypes = [vp]
    lib.mrf_exchange_spheres.restype = i32
    lib.mrf_step_prepare.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, vp]
    lib.mrf_step_prepare.restype = C.c_int
    lib.mrf_step_predict.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp]
    lib.mrf_step_predict.restype = C.c_int
    lib.mrf_step_action.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.mrf_step_action.restype = C.c_int
    dlp = C.POINTER(DeadlockConfig)
    lib.mrf_default_deadlock_config.argtypes = [dlp, i32]
    lib.mrf_default_deadlock_config.restype = None
    lib.mrf_deadlock_config_sizeof.argtypes = []
    lib.mrf_deadlock_config_sizeof.restype = C.c_int64
    lib.mrf_deadlock_init.argtypes = [vp, i64, vp, vp, vp]
    lib.mrf_deadlock_init.restype = C.c_int
    lib.mrf_control_prepare.argtypes = [vp, i64, vp, vp, vp, vp, i32, vp, vp]
    lib.mrf_control_prepare.restype = C.c_int
    lib.mrf_deadlock_step.argtypes = [vp, i64, dlp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.mrf_deadlock_step.restype = C.c_int
    lib.mrf_apply_action.argtypes = [vp, i64, vp, vp, vp, C.POINTER(C.c_double), C.c_double, vp]
    lib.mrf_apply_action.restype = C.c_int
    lib.mrf_episode_run.argtypes = [vp, vp, i64, i32, dlp, i32, C.POINTER(C.c_double), C.c_double, vp, vp, vp, vp, vp,
                                    vp, vp, vp, vp, vp, i32, vp]
    lib.mrf_episode_run.restype = C.c_int
    lib.mrf_compute_action_host.argtypes = [vp, i64, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.mrf_compute_action_host.restype = C.c_int
    lib.mrf_rollout_host.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp]
    lib.mrf_rollout_host.restype = C.c_int
    lib.mrf_rollout_cartesian_host.argtypes = [vp, i64, vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.mrf_rollout_cartesian_host.restype = C.c_int
    lib.mrf_fk_spheres_host.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    lib.mrf_fk_spheres_host.restype = C.c_int
    smp = C.POINTER(StateMachineConfig)
    lib.mrf_default_state_machine_config.argtypes = [smp, i32]
    lib.mrf_default_state_machine_config.restype = None
    lib.mrf_state_machine_config_sizeof.argtypes = []
    lib.mrf_state_machine_config_sizeof.restype = C.c_int64
    lib.mrf_state_machine_init.argtypes = [vp, i64, vp, vp, vp, vp]
    lib.mrf_state_machine_init.restype = C.c_int
    lib.mrf_state_machine_step.argtypes = [vp, i64, smp, vp, vp, vp, i32, vp, vp, vp, vp, i32, vp, vp]
    lib.mrf_state_machine_step.restype = C.c_int
    lib.mrf_episode_set_pick_place.argtypes = [vp, smp, vp, vp, i32, vp, vp, vp, vp, vp, vp]
    lib.mrf_episode_set_pick_place.restype = C.c_int
    # MRF_ABI_ANY=1 (with MRF_HIP_LIB): an older build under A/B timing may lack the entry points added since
    any_abi = bool(path or os.environ.get("MRF_HIP_LIB")) and os.environ.get("MRF_ABI_ANY") == "1"
    if not (any_abi and not hasattr(lib, "mrf_rollout_cartesian_coupled")):
        lib.mrf_rollout_cartesian_coupled.argtypes = [vp, i64, vp, vp, vp, vp, vp, vp, vp]
        lib.mrf_rollout_cartesian_coupled.restype = C.c_int
        lib.mrf_episode_set_rollout.argtypes = [vp, i32]
        lib.mrf_episode_set_rollout.restype = C.c_int
    if not (any_abi and not hasattr(lib, "mrf_episode_set_recorder")):
        lib.mrf_episode_set_recorder.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32]
        lib.mrf_episode_set_recorder.restype = C.c_int
    if not (any_abi and not hasattr(lib, "mrf_rollout_clock")):
        lib.mrf_rollout_clock.argtypes = [vp, C.POINTER(C.c_double), i32]
        lib.mrf_rollout_clock.restype = C.c_int
    lib.mrf_comm_unique_id.argtypes = [vp]
    lib.mrf_comm_unique_id.restype = C.c_int
    lib.mrf_comm_init.argtypes = [vp, i32, i32, vp]
    lib.mrf_comm_init.restype = C.c_int
    lib.mrf_comm_peer_open.argtypes = [vp, i32, i32, i64, vp]
    lib.mrf_comm_peer_open.restype = C.c_int
    lib.mrf_comm_peer_connect.argtypes = [vp, vp]
    lib.mrf_comm_peer_connect.restype = C.c_int
    lib.mrf_comm_partition.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    lib.mrf_comm_partition.restype = C.c_int
    if not (any_abi and not hasattr(lib, "mrf_comm_info")):
        lib.mrf_comm_info.argtypes = [vp, C.POINTER(i32), i32]
        lib.mrf_comm_info.restype = C.c_int
    lib.mrf_comm_transport.argtypes = [vp]
    lib.mrf_comm_transport.restype = i32
    if not (any_abi and not hasattr(lib, "mrf_exchange_scalars")):
        lib.mrf_build_has_wp.argtypes = []
        lib.mrf_build_has_wp.restype = C.c_int
        lib.mrf_step_predict_joints.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp]
        lib.mrf_step_predict_joints.restype = C.c_int
        lib.mrf_step_action_joints.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, vp, vp]
        lib.mrf_step_action_joints.restype = C.c_int
        if hasattr(lib, "mrf_step_action_predict_joints") or not any_abi:
            lib.mrf_step_action_predict_joints.argtypes = [vp, i64, i32, i32, vp, vp, vp, vp, vp, vp, vp]
            lib.mrf_step_action_predict_joints.restype = C.c_int
        lib.mrf_exchange_scalars.argtypes = [vp]
        lib.mrf_exchange_scalars.restype = i32
        lib.mrf_comm_peer_info.argtypes = [vp, C.POINTER(i32), i32]
        lib.mrf_comm_peer_info.restype = C.c_int
        lib.mrf_device_topology.argtypes = [C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32]
        lib.mrf_device_topology.restype = C.c_int
        lib.mrf_comm_peer_local_base.argtypes = [vp, C.POINTER(vp)]
        lib.mrf_comm_peer_local_base.restype = C.c_int
        lib.mrf_comm_peer_connect_local.argtypes = [vp, C.POINTER(vp)]
        lib.mrf_comm_peer_connect_local.restype = C.c_int
        lib.mrf_streams_concurrent.argtypes = [i32, vp, vp, C.POINTER(i32)]
        lib.mrf_streams_concurrent.restype = C.c_int
    lib.mrf_rollout_sharded.argtypes = [vp, i64, vp, vp, vp, vp, vp]
    lib.mrf_rollout_sharded.restype = C.c_int
    lib.mrf_comm_status.argtypes = [vp]
    lib.mrf_comm_status.restype = C.c_int
    lib.mrf_comm_reset.argtypes = [vp]
    lib.mrf_comm_reset.restype = C.c_int
    lib.mrf_comm_destroy.argtypes = [vp]
    lib.mrf_comm_destroy.restype = None
    if lib.mrf_abi_version() != MRF_ABI_VERSION and not any_abi:
        raise MrfLibraryError(f"ABI mismatch: library {lib.mrf_abi_version()} != python {MRF_ABI_VERSION}")
    if lib.mrf_config_sizeof() != C.sizeof(Config):
        raise MrfLibraryError(f"mrf_config size mismatch: C {lib.mrf_config_sizeof()} != ctypes {C.sizeof(Config)}")
    if lib.mrf_deadlock_config_sizeof() != C.sizeof(DeadlockConfig):
        raise MrfLibraryError("mrf_deadlock_config size mismatch")
    if lib.mrf_state_machine_config_sizeof() != C.sizeof(StateMachineConfig):
        raise MrfLibraryError("mrf_state_machine_config size mismatch")
    if path is None:
        _lib = lib
    return lib


def has_wp():
    """True when the loaded library carries the wave-pair rollout kernel (built with -DMRF_WITH_WP; the default build does
    not).  A library without the query symbol predates the switch (round 5 and earlier: the kernel was always built)."""
    lib = load_library()
    return bool(lib.mrf_build_has_wp()) if hasattr(lib, "mrf_build_has_wp") else True


def has_f32():
    """True when the loaded library carries the float32 kernels (built with -DMRF_WITH_F32; the default build does not).
    A library without the query symbol (an older A/B build under MRF_ABI_ANY) is probed: mrf_create of a float32 handle
    fails with MRF_E_CONFIG at validation -- before any device is touched -- exactly when the kernels are absent."""
    lib = load_library()
    if hasattr(lib, "mrf_build_has_f32"):
        return bool(lib.mrf_build_has_f32())
    cfg = Config()
    lib.mrf_default_config_panda(C.byref(cfg), 2, 2)
    cfg.scalar = F32
    h = C.c_void_p()
    rc = lib.mrf_create(C.byref(cfg), -1, C.byref(h))
    if h:
        lib.mrf_destroy(h)
    return rc != -2

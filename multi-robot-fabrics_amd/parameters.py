"""Scenario constants and flags of the Panda examples, with the attribute names the reference's drivers read
(examples/parameters_manipulators.py:4-181, examples/configs/panda_config.yaml:1-8).

URDF locations (PM:62-79, get_urdf_locations :154) point at the kinematics-only files that tools/make_kinematic_urdf.py
writes under examples/simulation_environments/urdfs/ (no meshes: there is no physics engine here); the tray / table
entries are kept as names only.  N > 3 robots (the reference stops at 3, PM:72-73,119-120) use the build-defined ring
of config.mount_positions.
"""
import copy
import os

import numpy as np

from . import config as _config
from . import scenarios as _scenarios


_URDF_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "simulation_environments", "urdfs")


class manipulator_parameters:
    def __init__(self, nr_robots, n_obst_per_link=1, urdf_dir=None):
        N = nr_robots
        urdf_dir = _URDF_DIR if urdf_dir is None else urdf_dir
        self.URDF_file_panda = os.path.join(urdf_dir, "panda_with_finger.urdf")                 # PM:63
        self.URDF_file_kinova = os.path.join(urdf_dir, "kinova_gen_3_lite.urdf")                # not shipped
        shallow = N == 3                                                                          # PM:65-70
        self.URDF_tray_location = os.path.join(urdf_dir, "tray", "tray_shallow.urdf" if shallow else "tray.urdf")
        self.URDF_table = os.path.join(urdf_dir, "table", "table_enlarged.urdf" if shallow else "table.urdf")
        self.urdf_links = {"URDF_file_panda": self.URDF_file_panda, "URDF_file_kinova": self.URDF_file_kinova,
                           "URDF_tray_location": self.URDF_tray_location, "URDF_table": self.URDF_table}
        if N == 3:                                                                                # PM:116-118
            self.tray_positions, self.tray_orientations = [[0.2, 0.77, 0.64], [0.9, -0.63, 0.62]], [[0, 0, 0, 1], [0, 0, 1, 1]]
            self.table_position = [0.5, 0.4, 0]
        else:                                                                                     # PM:96-98
            self.tray_positions, self.tray_orientations = [[0.2, 0.67, 0.57], [0.9, -0.67, 0.57]], [[0, 0, -1, 1], [0, 0, 1, 1]]
            self.table_position = [0.5, 0, 0]
        self.dt = 0.01                       # PM:8
        self.n_cubes = 6
        self.nr_robots = N
        self.dof = [7] * N
        self.fabrics_mode = "vel"            # PM:12
        self.nu = self.dof
        self.nx = self.dof
        self.nr_obsts = [0] * N
        self.radius_obsts = [[] for _ in range(N)]
        self.n_obst_per_link = n_obst_per_link
        self.radius_sphere = 0.08            # PM:23
        self.nr_constraints = [1] * N
        self.collision_links_nrs = [list(range(1, 9)) for _ in range(N)]
        self.collision_links = [["panda_link%d" % l for l in range(1, 9)] for _ in range(N)]
        self.nr_obsts_dyn = [8 * (N - 1)] * N                      # one sphere per link in the rollouts (PM:28-33)
        self.nr_obsts_dyn_all = [8 * (N - 1) * n_obst_per_link] * N
        self.robot_types = ["panda"] * N
        self.r_robots = [[self.radius_sphere] * 8 for _ in range(N)]
        self.radius_body_panda_links = {str(l): np.array(self.radius_sphere) for l in range(3, 9)}   # PM:41-43
        n_all = self.nr_obsts_dyn_all[0]
        self.a_dyns_obsts = [[np.zeros(3)] * n_all for _ in range(N)]
        self.r_dyns_obsts = [[self.radius_sphere] * n_all for _ in range(N)]

        self.ROLLOUT_FABRICS = False
        self.ROLLOUTS_PLOTTING = False
        self.STATIC_OR_DYN_FABRICS = 0
        self.RESOLVE_DEADLOCKS = True
        self.ESTIMATE_GOAL = False
        self.N_HORIZON = 2                   # PM:60
        self.MPC_LAYER = False

        self.z_table = _config.Z_TABLE
        pos, yaw = _config.mount_positions(N)
        self.mount_positions = pos
        self.mount_yaws = yaw
        self.mount_orientations = [np.array([0.0, 0.0, np.sin(y / 2), np.cos(y / 2)]) for y in yaw]
        self.mount_transform = [_config.mount_transform(p, y) for p, y in zip(pos, yaw)]
        self.mount_param = {"z_table": self.z_table, "mount_positions": pos, "mount_orientations": self.mount_orientations}
        p0 = _scenarios.pos0(N)
        self.pos0 = np.array([np.concatenate([p, [0.02, 0.02]]) for p in p0])
        self.rotation_matrix_pandas = [_scenarios.ROT_GOAL_1.copy() for _ in range(N)]    # PM:121-122
        self.start_goals = [list(g) for g in _scenarios.start_goals(N)]
        self.constraints = [np.array([0.0, 0.0, 1.0, -self.z_table])] * N                  # PM:136

    def get_mount_parameters(self):
        return self.mount_param

    def get_urdf_locations(self):
        return self.urdf_links

    def define_settings(self, ROLLOUT_FABRICS=False, ROLLOUTS_PLOTTING=False, STATIC_OR_DYN_FABRICS=0,
                        RESOLVE_DEADLOCKS=True, ESTIMATE_GOAL=False, N_HORIZON=10, MPC_LAYER=False, n_obst_per_link=1):
        self.ROLLOUT_FABRICS = ROLLOUT_FABRICS
        self.ROLLOUTS_PLOTTING = ROLLOUTS_PLOTTING
        self.STATIC_OR_DYN_FABRICS = STATIC_OR_DYN_FABRICS
        self.RESOLVE_DEADLOCKS = RESOLVE_DEADLOCKS
        self.ESTIMATE_GOAL = ESTIMATE_GOAL
        self.N_HORIZON = N_HORIZON
        self.MPC_LAYER = MPC_LAYER
        self.n_obst_per_link = n_obst_per_link
        return self.get_settings() + [n_obst_per_link]

    def get_settings(self):
        return [self.ROLLOUT_FABRICS, self.ROLLOUTS_PLOTTING, self.STATIC_OR_DYN_FABRICS, self.RESOLVE_DEADLOCKS,
                self.ESTIMATE_GOAL, self.N_HORIZON, self.MPC_LAYER]

    def obstacle_counts(self, i_robot):
        """(static, dynamic) obstacle spheres the MAIN planner of robot i is built for: every sphere of the other robots
        (n_obst_per_link on each of their eight links), static or moving by STATIC_OR_DYN_FABRICS."""
        n = self.nr_obsts_dyn_all[i_robot]
        return (0, n) if self.STATIC_OR_DYN_FABRICS else (n, 0)

    def rollout_planner_spec(self, i_robot):
        """(dof, static, dynamic, collision links) of robot i's ROLLOUT planner: it sees one sphere per link of the other
        robots -- their link origins, nr_obsts_dyn -- whatever n_obst_per_link the main planners use."""
        return self.dof[i_robot], self.nr_obsts[i_robot], self.nr_obsts_dyn[i_robot], self.collision_links_nrs[i_robot]

    def apply_yaml(self, setup):
        """define_settings from a loaded panda_config.yaml (its n_robots key belongs to the constructor)."""
        return self.define_settings(**{k: v for k, v in setup.items() if k != "n_robots"})

    def set_horizon(self, n_horizon):
        self.N_HORIZON = n_horizon

    def copy(self):
        return copy.deepcopy(self)


def read_yaml(path=None):
    """The eight keys of examples/configs/panda_config.yaml as a dict (path None: the shipped file)."""
    import yaml
    if path is None:
        path = os.path.join(os.path.dirname(_URDF_DIR), os.pardir, "configs", "panda_config.yaml")
    with open(path, "r") as f:
        return yaml.safe_load(f)


def load_yaml_settings(path=None, apply_flags=True):
    """-> (manipulator_parameters, settings dict); apply_flags=False leaves the flags at the constructor's values (the
    joint-space driver builds its main planners in that state)."""
    setup = read_yaml(path)
    p = manipulator_parameters(nr_robots=setup["n_robots"], n_obst_per_link=setup["n_obst_per_link"])
    if apply_flags:
        p.apply_yaml(setup)
    return p, setup

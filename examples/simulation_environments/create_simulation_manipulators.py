"""examples/simulation_environments/create_simulation_manipulators.py of the reference: `create_manipulators_simulation`.
pybullet / urdfenvs are not part of this build; the class is the kinematic stand-in of multi-robot-fabrics_amd/scene.py
(exact velocity integration, device sphere kinematics, cubes that travel with a closed gripper), behind the reference's
import path (`from examples.simulation_environments import create_simulation_manipulators`, example_pandas_Jointspace.py:7)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

from multi_robot_fabrics_amd.scene import BoxObstacle, KinematicManipulatorEnv, create_manipulators_simulation  # noqa: E402,F401

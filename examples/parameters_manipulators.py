"""examples/parameters_manipulators.py of the reference: `manipulator_parameters(nr_robots, n_obst_per_link=1)`.
The class lives in the package (multi-robot-fabrics_amd/parameters.py); this module keeps the reference's import path
(`import examples.parameters_manipulators as parameters_manipulators`, example_pandas_Jointspace.py:16)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from multi_robot_fabrics_amd.parameters import manipulator_parameters  # noqa: E402,F401

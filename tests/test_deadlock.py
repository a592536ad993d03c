"""Deadlock detection / resolution (DP:50-118) against sequences recorded from the reference's own
module (tests/golden/make_deadlock_golden.py imports it in the build container; only inputs/outputs are stored)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "deadlock_sequences.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files if not k.startswith("velavg")})
VELAVG = sorted({k.split("/")[0] for k in GOLD.files if k.startswith("velavg")})


def case(name):
    return {k.split("/")[1]: GOLD[k] for k in GOLD.files if k.startswith(name + "/")}


@pytest.mark.parametrize("name", CASES)
def test_host_mirror_replays_reference_sequences(name):
    from multi_robot_fabrics_amd.deadlock import deadlockprevention
    c = case(name)
    n = c["x"].shape[1]
    dp = deadlockprevention([2 if name.startswith("point") else 7] * n, n, 10)
    for t in range(c["x"].shape[0]):
        g, w, t_out = dp.deadlock_checking([x.copy() for x in c["x"][t]], [g.copy() for g in c["goals_in"][t]],
                                           list(c["weights_in"][t]), int(c["time_step"][t]), int(c["t_out_in"][t]),
                                           float(c["avg"][t]), list(c["sm"][t]))
        assert t_out == c["t_out_out"][t], (name, t)
        np.testing.assert_array_equal(np.array(g, dtype=float), c["goals_out"][t])
        np.testing.assert_array_equal(np.array(w, dtype=float), c["weights_out"][t])
        assert (dp.i_leader, dp.i_follower) == (c["leader"][t], c["follower"][t])
        assert list(dp.i_robots_dead) == list(c["dead"][t])
        assert dp.time_in_deadlock == c["time_in_deadlock"][t]


@pytest.mark.parametrize("name", VELAVG)
def test_compute_velocity_average_matches_the_reference_method(name):
    """DP:36-43 through the reference's own method (recorded by make_deadlock_golden.velocity_average_vectors): uniform and
    mixed dof lists, an all-zero rollout.  The reference sums sqrt(x**2) term by term; the mirror sums |x| per joint."""
    from multi_robot_fabrics_amd.deadlock import deadlockprevention
    c = case(name)
    dof, H = [int(d) for d in c["dof"]], int(c["H"])
    dp = deadlockprevention(dof, len(dof), H)
    for k in range(c["qdot"].shape[0]):
        d = {f"robot_{i}": [list(c["qdot"][k, i, j]) for j in range(n)] for i, n in enumerate(dof)}
        assert abs(dp.compute_velocity_average(d) - float(c["avg"][k])) <= 1e-14 * max(1.0, abs(float(c["avg"][k])))
    assert float(c["avg"][0]) == 0.0


@pytest.mark.parametrize("name", CASES)
def test_oracle_replays_reference_sequences(name):
    """The oracle used by the GPU parity tests is itself pinned by the reference's recorded sequences."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle"))
    import deadlock_oracle as do
    c = case(name)
    K = do.constants(point_mass=name.startswith("point"))
    st = do.initial_state()
    for t in range(c["x"].shape[0]):
        st["time_deadlock_out"] = int(c["t_out_in"][t])      # the driver owns this variable (EXJ:273,377-383)
        st, g, w = do.step(st, K, c["x"][t], c["goals_in"][t], c["weights_in"][t], float(c["avg"][t]),
                           int(c["time_step"][t]), [int(v) for v in c["sm"][t]])
        assert st["time_deadlock_out"] == c["t_out_out"][t], (name, t)
        np.testing.assert_allclose(g, c["goals_out"][t], rtol=0, atol=1e-15)
        np.testing.assert_array_equal(w, c["weights_out"][t])
        assert (st["leader"], st["follower"]) == (c["leader"][t], c["follower"][t])
        assert list(st["dead"]) == list(c["dead"][t])
        assert st["time_in_deadlock"] == c["time_in_deadlock"][t]

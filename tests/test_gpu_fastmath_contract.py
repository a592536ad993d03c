"""The contract of the -ffast-math solve kernels (include/mrf.h, "Numerical contract"): a row whose barrier coordinate is
<= 0 (spheres overlapping: the reference's even-power barriers, EXJ:88-89, have no meaning there) returns unspecified
values -- finite or not -- but ONLY that scenario does: the rows of every other scenario, including those that share its
wave and its LDS exchange tile, agree with a run without the penetration to round-off (1e-12; not bit for bit: the rollout
kernels choose between the incremental and the full sincos by a wave-wide vote on |dt*qdot|, and a diverging row changes
that vote for its wave)."""
import numpy as np
import pytest
import torch

from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle

pytestmark = pytest.mark.gpu
RTOL = 1e-12


def _batches(cfg, n_scen, bad):
    """Two batches that differ in the scenarios `bad` only: there every body radius is 1.5 m, so every collision leaf of
    those rows has x = d / (r_o + r_b) - 1 < 0 (the arms are ~1 m apart)."""
    good = scenarios.panda_batch(cfg, n_scen, seed=77, x_min=0.08)
    pen = {k: np.array(v, copy=True) for k, v in good.items() if isinstance(v, np.ndarray)}
    N = cfg.n_robots
    for s in bad:
        pen["params"][abi.P_RADIUS_BODY:abi.P_RADIUS_BODY + 6, s * N:(s + 1) * N] = 1.5
    return good, pen


@pytest.mark.parametrize("kernel", [1, 3])      # row-per-lane and wave-pair kernels (both exchange through a shared tile)
def test_penetrating_scenario_stays_in_its_rows_rollout(kernel):
    cfg = config.panda_config(n_robots=3, horizon=8)
    cfg.goal_estimate_mask = 0b110
    cfg.kernel_select = kernel
    n_scen, bad = 45, (0, 7, 20, 21, 44)        # 21 scenarios per wave: first / middle / last row of a wave, next wave's first
    good, pen = _batches(cfg, n_scen, bad)
    h = FabricHandle(cfg, 0)
    outs = []
    for b in (good, pen):
        avg, tq, tqd = h.rollout(h.tensor(b["q"]), h.tensor(b["qdot"]), h.tensor(b["params"]), want_traj=True)
        torch.cuda.synchronize()
        outs.append((avg.cpu().numpy(), tq.cpu().numpy(), tqd.cpu().numpy()))
    N = cfg.n_robots
    keep = np.ones(n_scen * N, dtype=bool)
    for s in bad:
        keep[s * N:(s + 1) * N] = False
    for a, b in zip(outs[0], outs[1]):
        assert np.isfinite(b[..., keep]).all()
        assert np.abs(a[..., keep] - b[..., keep]).max() <= RTOL * np.abs(a[..., keep]).max()
    assert np.isfinite(outs[0][0]).all()
    # the penetrating rows did change (the test would be vacuous otherwise)
    assert not np.array_equal(outs[0][0][~keep], outs[1][0][~keep])


def test_penetrating_scenario_stays_in_its_rows_coupled_action():
    cfg = config.panda_config(n_robots=3, horizon=1)
    cfg.kernel_select = 1
    n_scen, bad = 45, (0, 7, 20, 21, 44)
    good, pen = _batches(cfg, n_scen, bad)
    h = FabricHandle(cfg, 0)
    acts = []
    for b in (good, pen):
        acts.append(h.compute_action_coupled(h.tensor(b["q"]), h.tensor(b["qdot"]), h.tensor(b["params"]), use_accel=False).cpu().numpy())
    N = cfg.n_robots
    keep = np.ones(n_scen * N, dtype=bool)
    for s in bad:
        keep[s * N:(s + 1) * N] = False
    assert np.array_equal(acts[0][:, keep], acts[1][:, keep])      # a single evaluation has no wave-wide decision: bit for bit
    assert not np.array_equal(acts[0][:, ~keep], acts[1][:, ~keep])

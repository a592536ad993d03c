"""The pin against the REAL reference (fabrics 0.9.5 / CasADi), SURVEY 8c.

tests/golden/reference_*.npz hold the outputs of the reference's own planner objects on the seeded inputs of the
committed golden files (tests/golden/make_reference_golden.py, which needs the reference's python<3.10 environment and
therefore cannot run in the build container).  While those files are absent every test here SKIPS and parity of rows
a-e stays "unpinned" (DESIGN.md section 3); the moment they are committed the same tests compare the float64 oracle (CPU)
and the HIP kernels (-m gpu, through the C ABI) against them -- no code change needed.

Tolerance against the reference: 1e-6 relative.  The reference evaluates (M + eps I)^-1 through CasADi's pinv (normal
equations: cond(M)^2 * 2^-53 of round-off, DESIGN.md deviation 2), so its own results carry ~1e-8 near barriers; the
build's 1e-9 GPU-vs-oracle tolerance stays what it is."""
import numpy as np
import pytest

import reference_cases as rc

TOL = 1e-6


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


@pytest.mark.skipif(not rc.have("panda_actions"), reason=rc.HOW)
def test_oracle_panda_actions_match_the_reference(oracle):
    want = np.load(rc.FILES["panda_actions"])["action"]
    got = rc.oracle_actions(oracle, rc.panda_action_cases())
    for i, c in enumerate(rc.panda_action_cases()):
        assert rel(got[i], want[i]) < TOL, (c[0], got[i], want[i])


@pytest.mark.skipif(not rc.have("planar_actions"), reason=rc.HOW)
def test_oracle_planar_actions_match_the_reference(oracle):
    want = np.load(rc.FILES["planar_actions"])["action"]
    got = rc.oracle_actions(oracle, rc.planar_action_cases())
    assert rel(got, want) < TOL


@pytest.mark.skipif(not rc.have("panda_rollout"), reason=rc.HOW)
def test_oracle_rollouts_match_the_reference(oracle):
    want = np.load(rc.FILES["panda_rollout"])
    got = rc.oracle_rollouts(oracle, rc.rollout_cases())
    for k, v in got.items():
        assert rel(v, want[k]) < TOL, k


def _hip_actions(cases):
    import torch
    from multi_robot_fabrics_amd.runtime import FabricHandle
    out = []
    for kind, cfg, q, qd, prm, ox, ov, oa, orad, ns in cases:
        h = FabricHandle(cfg, 0)
        t = h.tensor
        act = h.compute_action(t(q), t(qd), t(prm), t(ox), t(ov), t(oa), t(orad), n_static=ns)
        torch.cuda.synchronize()
        out.append(act.cpu().numpy()[:, 0])
    return np.stack(out)


@pytest.mark.gpu
@pytest.mark.skipif(not rc.have("panda_actions"), reason=rc.HOW)
def test_hip_panda_actions_match_the_reference():
    want = np.load(rc.FILES["panda_actions"])["action"]
    got = _hip_actions(rc.panda_action_cases())
    for i, c in enumerate(rc.panda_action_cases()):
        assert rel(got[i], want[i]) < TOL, (c[0], got[i], want[i])


@pytest.mark.gpu
@pytest.mark.skipif(not rc.have("planar_actions"), reason=rc.HOW)
def test_hip_planar_actions_match_the_reference():
    want = np.load(rc.FILES["planar_actions"])["action"]
    assert rel(_hip_actions(rc.planar_action_cases()), want) < TOL


@pytest.mark.gpu
@pytest.mark.skipif(not rc.have("panda_rollout"), reason=rc.HOW)
def test_hip_rollouts_match_the_reference():
    from multi_robot_fabrics_amd.runtime import FabricHandle
    want = np.load(rc.FILES["panda_rollout"])
    for name, cfg, q0, qd0, prm in rc.rollout_cases():
        for select in (1, 2):       # row-per-lane and cooperative kernels
            cfg.kernel_select = select
            h = FabricHandle(cfg, 0)
            avg, tq, tqd = h.rollout(h.tensor(q0), h.tensor(qd0), h.tensor(prm), want_traj=True)
            assert rel(tq.cpu().numpy().transpose(2, 0, 1), want[name + "_q"]) < TOL
            assert rel(tqd.cpu().numpy().transpose(2, 0, 1), want[name + "_qd"]) < TOL
            assert rel(avg.cpu().numpy(), want[name + "_avg"]) < TOL


def test_the_recipe_evaluates_the_committed_inputs(oracle):
    """Runs always: the case builders used above reproduce the committed (autodiff) golden outputs through the oracle,
    so the day reference_*.npz appear the comparison is between like and like."""
    g = np.load(rc.GOLD + "/panda_actions.npz")
    got = rc.oracle_actions(oracle, rc.panda_action_cases())
    for i, kind in enumerate(g["kinds"]):
        assert rel(got[i], g["action"][i]) < (1e-7 if kind in ("near", "nogoal") else 1e-10), kind
    assert rel(rc.oracle_actions(oracle, rc.planar_action_cases()), np.load(rc.GOLD + "/planar_actions.npz")["action"]) < 1e-10
    r = rc.oracle_rollouts(oracle, rc.rollout_cases())
    gr = np.load(rc.GOLD + "/panda_rollout.npz")
    for k, v in r.items():
        assert rel(v, gr[k]) < 1e-10, k

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


@pytest.mark.skipif(not rc.have("panda_rollout_c4"), reason=rc.HOW)
def test_oracle_c4_rollout_matches_the_reference(oracle):
    """BASELINE config 4: 3-Panda RF-CV H=30 (rows a6, a7, a12)."""
    want = np.load(rc.FILES["panda_rollout_c4"])
    got = rc.oracle_c4(oracle, rc.c4_case())
    for k, v in got.items():
        assert rel(v, want[k]) < TOL, k


@pytest.mark.skipif(not rc.have("panda_cartesian"), reason=rc.HOW)
def test_oracle_cartesian_rollouts_match_the_reference(oracle):
    """FabricsRollouts (row a11) incl. the RF-CV goal estimate inputs of robot 1 (row a12)."""
    want = np.load(rc.FILES["panda_cartesian"])
    got = rc.oracle_cartesian(oracle, rc.cartesian_cases())
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


@pytest.mark.gpu
@pytest.mark.skipif(not rc.have("panda_rollout_c4"), reason=rc.HOW)
def test_hip_c4_rollout_matches_the_reference():
    from multi_robot_fabrics_amd.runtime import FabricHandle
    want = np.load(rc.FILES["panda_rollout_c4"])
    cfg, q0, qd0, prm = rc.c4_case()
    for select in (1, 2):
        cfg.kernel_select = select
        h = FabricHandle(cfg, 0)
        avg, tq, tqd = h.rollout(h.tensor(q0), h.tensor(qd0), h.tensor(prm), want_traj=True)
        assert rel(avg.cpu().numpy(), want["avg"]) < TOL
        assert rel(tq[-1].cpu().numpy().T, want["q_last"]) < TOL and rel(tqd[-1].cpu().numpy().T, want["qd_last"]) < TOL


@pytest.mark.gpu
@pytest.mark.skipif(not rc.have("panda_cartesian"), reason=rc.HOW)
def test_hip_cartesian_rollouts_match_the_reference():
    from multi_robot_fabrics_amd.runtime import FabricHandle
    want = np.load(rc.FILES["panda_cartesian"])
    for name, cfg, q0, qd0, prm, ox, ov, orad in rc.cartesian_cases():
        h = FabricHandle(cfg, 0)
        t = h.tensor
        avg, tq, tqd = h.rollout_cartesian(t(q0), t(qd0), t(prm), t(ox), t(ov), None, t(orad), want_traj=True)
        assert rel(tq.cpu().numpy()[:, :, 0], want[name + "_q"]) < TOL
        assert rel(tqd.cpu().numpy()[:, :, 0], want[name + "_qd"]) < TOL
        assert rel(avg.cpu().numpy()[0], want[name + "_avg"]) < TOL


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
    c4 = rc.oracle_c4(oracle, rc.c4_case())
    g4 = np.load(rc.GOLD + "/panda_rollout_c4.npz")
    for k, v in c4.items():
        assert rel(v, g4[k]) < 1e-9, k                  # 30 coupled steps of a 3-robot cell
    ca = rc.oracle_cartesian(oracle, rc.cartesian_cases())
    gc = np.load(rc.GOLD + "/panda_cartesian.npz")
    for k, v in ca.items():
        assert rel(v, gc[k]) < 1e-10, k


def test_composition_conventions_are_diagnosable(oracle):
    """tests/reconcile_constants.py's numpy composition (SURVEY A.3 from the oracle's pulled specs) reproduces the
    oracle's own actions, and each alternative convention moves them by an amount the pin tolerance can see or just
    cannot -- so a 1e-6 miss on first contact with the reference is classified, not debugged."""
    import reconcile_constants as recon
    got = rc.oracle_actions(oracle, rc.panda_action_cases())
    v = recon.composition_variants(got)
    labels = list(v)
    assert v[labels[0]] < 1e-10
    assert 1e-8 < v[labels[1]] < 1e-5        # eps at every stage: ~1e-6 relative, the size of TOL
    assert all(1e-9 < v[k] < 1e-4 for k in labels[2:])


def test_attractor_norm_conventions_are_diagnosable(oracle):
    """How ca.norm_2 of the attractor strings behaves at x = 0 cannot be checked here; the committed case "rest" sits
    exactly on it (q[6] == x_goal_2, the reference's own start pose PM:93 / EXJ:428), so the reference's vector for that
    case decides between the candidates reconcile_constants.py enumerates: the build convention and CasADi-as-recalled
    agree to round-off, "x / sqrt(x.x) in every dimension" turns that one case into NaN (a different NaN pattern), and a
    regularised norm moves finite digits only."""
    import reconcile_constants as recon
    cases = rc.panda_action_cases()
    rest = [i for i, c in enumerate(cases) if c[0] == "rest"]
    assert len(rest) == 1 and float(cases[rest[0]][2][6, 0] - cases[rest[0]][4][17, 0]) == 0.0     # q[6] - x_goal_2
    got = rc.oracle_actions(oracle, cases)
    assert np.isfinite(got).all()
    v = recon.norm_variants(got)
    labels = list(v)
    assert v[labels[0]] == (0.0, True)
    assert v[labels[1]][0] < 1e-15 and v[labels[1]][1]                 # |x| with sign(0) = 0: the same numbers
    assert not v[labels[2]][1]                                        # 0/0 in 1-D: the "rest" case becomes NaN
    assert all(v[k][1] and v[k][0] < 1e-8 for k in labels[3:])         # regularised norms: far below the pin tolerance
    oracle.set_attractor_norm(3, 0.0)
    try:
        bad = rc.oracle_actions(oracle, cases)
    finally:
        oracle.set_attractor_norm(0, 0.0)
    assert [i for i in range(len(cases)) if not np.isfinite(bad[i]).all()] == rest


def test_reconciliation_writes_constants_that_config_loads(oracle, tmp_path, monkeypatch):
    """Zero-code reconciliation: stand-in "reference" vectors made by the oracle with another discrete
    convention (attractor metric M = A instead of the Hessian 2A) -> reconcile_constants.py --write finds the convention and
    writes the JSON file -> every planner configuration built afterwards carries it ($MRF_CONSTANTS), and the oracle on
    that configuration reproduces the stand-in vectors."""
    import sys
    import reconcile_constants as recon
    from multi_robot_fabrics_amd import config
    truth = {"attr_mu": 1.0, "attr_ml": 0.15}            # M = A: both metric constants halved
    files = {k: str(tmp_path / f"reference_{k}.npz") for k in rc.KINDS}
    acts = rc.oracle_actions(oracle, rc.panda_action_cases(truth))
    np.savez(files["panda_actions"], action=acts)
    monkeypatch.setattr(rc, "FILES", files)
    out = str(tmp_path / "constants.json")
    monkeypatch.setattr(sys, "argv", ["reconcile_constants.py", "--write", out])
    assert recon.main() == 0
    monkeypatch.setenv("MRF_CONSTANTS", out)
    cfg = config.panda_config(n_robots=1, horizon=1)
    assert cfg.attr_mu == 1.0 and cfg.attr_ml == 0.15 and cfg.jdot_sign == -1.0
    again = rc.oracle_actions(oracle, rc.panda_action_cases())          # built through config.panda_config: reconciled
    assert rel(again, acts) < 1e-12
    monkeypatch.delenv("MRF_CONSTANTS")
    assert config.panda_config(n_robots=1, horizon=1).attr_mu == 2.0


@pytest.mark.gpu
def test_reference_recipe_dry_run_on_the_mirrors(tmp_path):
    """tests/golden/make_reference_golden.py cannot run where the reference's wheels are missing -- but its plumbing
    can: with this build's mirror classes standing in under the reference's import names (--dry-run-with-mirrors) the
    whole recipe runs on the GPU, writes every reference_*.npz key the pin tests read, and -- because the mirrors are the
    kernels -- reproduces the committed (autodiff) fixtures.  Not a parity statement: the day the real stack runs it,
    the call sequence, keyword names, shapes and keys are already known to be right."""
    import os
    import subprocess
    import sys
    script = os.path.join(rc.GOLD, "make_reference_golden.py")
    out = subprocess.run([sys.executable, script, "--dry-run-with-mirrors", "--out", str(tmp_path)], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    refuse = subprocess.run([sys.executable, script, "--dry-run-with-mirrors"], capture_output=True, text=True, timeout=120)
    assert refuse.returncode != 0 and "scratch directory" in (refuse.stderr + refuse.stdout)   # never next to the fixtures
    got = {k: np.load(os.path.join(str(tmp_path), f"reference_{k}.npz")) for k in rc.KINDS}
    g = np.load(rc.GOLD + "/panda_actions.npz")
    for i, kind in enumerate(g["kinds"]):
        assert rel(got["panda_actions"]["action"][i], g["action"][i]) < (1e-7 if kind in ("near", "nogoal") else 1e-9), kind
    assert rel(got["planar_actions"]["action"], np.load(rc.GOLD + "/planar_actions.npz")["action"]) < 1e-9
    gr = np.load(rc.GOLD + "/panda_rollout.npz")
    for k in ("dyn_q", "dyn_qd", "dyn_avg", "stat_q", "stat_qd", "stat_avg"):
        assert rel(got["panda_rollout"][k], gr[k]) < 1e-9, k
    g4 = np.load(rc.GOLD + "/panda_rollout_c4.npz")
    assert rel(got["panda_rollout_c4"]["avg"], g4["avg"]) < 1e-8
    assert rel(got["panda_rollout_c4"]["estimated_goal_1"], g4["estimated_goal_1"]) < 1e-12
    gc = np.load(rc.GOLD + "/panda_cartesian.npz")
    for k in ("r0_q", "r0_qd", "r0_avg", "r1_q", "r1_qd", "r1_avg"):
        assert rel(got["panda_cartesian"][k], gc[k]) < 1e-9, k

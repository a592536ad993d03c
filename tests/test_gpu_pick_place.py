"""Pick-and-place state machine on the device (include/mrf.h mrf_state_machine_*).

 * k_state_machine replays the sequences RECORDED FROM THE REFERENCE's own state_machine.py
   (tests/golden/state_machine_sequences.npz, 4 800 control steps, all 8 states): states, counters and gripper status
   exact, goals and gripper commands to 1e-12;
 * a device-resident episode with the state machine attached (mrf_episode_set_pick_place: minimal block / gripper
   model, grasp planner in state 2, zero action in states 3 and 5) against the same loop stepped from the host with the
   pinned Python mirror (pick_place.StateMachine) making the decisions."""
import contextlib
import io
import os

import numpy as np
import pytest
import torch

from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.pick_place import StateMachine
from multi_robot_fabrics_amd.runtime import ControlLoop, FabricHandle

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "state_machine_sequences.npz"))
CASES = sorted({k.split("/")[0] for k in GOLD.files})


@pytest.mark.parametrize("name", CASES)
def test_state_machine_kernel_replays_reference_sequences(name):
    c = {k.split("/")[1]: GOLD[k] for k in GOLD.files if k.startswith(name + "/")}
    nb = int(c["meta"][0])
    cfg = config.panda_config(n_robots=2, horizon=1)
    h = FabricHandle(cfg, 0)
    rows = 2                                        # the same sequence in both rows (row 1 checks the indexing)
    t2 = lambda v: h.tensor(np.repeat(np.asarray(v, dtype=float)[:, None], rows, axis=1))
    start = t2(c["start"])
    sm = h.state_machine_config(nb, model=0)
    st, sg = h.state_machine_state(start)
    prm = torch.zeros((abi.NPARAM, rows), dtype=torch.float64, device="cuda")
    T = len(c["state"])
    states, goals, weights, picked, acts, status = [], [], [], [], [], []
    for t in range(T):
        grip = t2(c["grip"][t])
        act = h.state_machine_step(sm, t2(c["x_ee"][t]), start, t2(c["block"][t])[None].contiguous(), grip, st, sg, prm,
                                   skip_robot_mask=0b10)
        states.append(st[abi.SM_STATE].clone()); goals.append(sg[0:3].clone()); weights.append(sg[abi.SM_WEIGHT].clone())
        picked.append(st[abi.SM_PICKED].clone()); acts.append(act.clone()); status.append(st[abi.SM_GRIPPER].clone())
        # row 0 (robot 0) gets goal and weight; row 1 (robot 1, RF-CV estimated) keeps its goal but still gets the state
        # machine's weight -- weight_low = 0 while gripping (EXJ:313-316 feed weight_goals0 whatever ESTIMATE_GOAL says)
        assert torch.equal(prm[abi.P_WEIGHT_GOAL_0], sg[abi.SM_WEIGHT])
        if t in (0, T // 2, T - 1):
            assert torch.equal(prm[0:3, 0], sg[0:3, 0])
            assert float(prm[0:3, 1].abs().max()) == 0.0
    S = torch.stack(states).cpu().numpy()
    for r in range(rows):
        assert (S[:, r] == c["state"]).all(), (name, int(np.argmax(S[:, r] != c["state"])))
        assert (torch.stack(picked).cpu().numpy()[:, r] == c["picked"]).all()
        assert (torch.stack(status).cpu().numpy()[:, r] == c["status"]).all()
        assert np.abs(torch.stack(goals).cpu().numpy()[:, :, r] - c["goal"]).max() < 1e-12
        assert np.abs(torch.stack(weights).cpu().numpy()[:, r] - c["weight"]).max() == 0.0
        assert np.abs(torch.stack(acts).cpu().numpy()[:, :, r] - c["grip_act"]).max() < 1e-12
    assert sorted(set(S[:, 0].tolist())) == sorted(set(c["state"].tolist()))
    assert 0.0 in set(torch.stack(weights).cpu().numpy()[:, 1].tolist()) or 3 not in set(c["state"].tolist())


def _host_episode(hr, ha, hg, q, qd, prm, start, blocks, grip0, nb, steps, dl_cfg, vel_limit, apply_estimate=False):
    """The control step of mrf_episode_run, one call at a time, with the Python mirror of the reference's state
    machine (pinned by tests/test_pick_place.py) deciding goals, gripper and action selection on the host."""
    N = ha.cfg.n_robots
    rows = q.shape[1]
    q, qd = q.clone(), qd.clone()
    x = {}
    sms = []
    for r in range(rows):
        sms.append(StateMachine(start_goal=start[:, r].cpu().numpy().copy(), nr_robots=N, nr_blocks=nb,
                                fk_fun_ee=(lambda _q, r=r: x["ee"][:, r].copy()), robot_types=["panda"] * N))
    grip = grip0.cpu().numpy().copy()
    blk = blocks.cpu().numpy()
    dl_state, dl_goal = hr.deadlock_state(rows // N)
    work = torch.empty_like(prm)
    hist = []
    for t in range(steps):
        x_ee = hr.control_prepare(q, qd, prm, work, apply_estimate=apply_estimate)
        x["ee"] = x_ee.cpu().numpy()
        w = work.cpu().numpy()
        mask = hr.cfg.goal_estimate_mask if apply_estimate else 0
        states = np.zeros(rows, dtype=np.int32)
        for r, sm in enumerate(sms):
            bi = min(sm.get_nr_blocks_picked(), nb - 1)
            block = blk[bi, :, r].copy()
            if sm.gripper_panda == "close" and sm.state_machine_panda in (12, 4):
                block = x["ee"][:, r].copy()
            with contextlib.redirect_stdout(io.StringIO()):
                states[r] = sm.get_state_machine_panda(q_robot=None, q_robot_gripper=grip[:, r].copy(), goal_block=block,
                                                       robot_type="panda")
            if not (mask >> (r % N)) & 1:                            # EXJ:346-348: the estimate replaces the goal only
                w[0:3, r] = np.asarray(sm.get_goal_robot(), dtype=float)
            w[abi.P_WEIGHT_GOAL_0, r] = sm.get_weight_goal0()
            a = sm.get_gripper_action_panda(grip[:, r])
            grip[:, r] = np.clip(grip[:, r] + ha.cfg.dt * a, 0.0, 0.04)
        work = ha.tensor(w)
        sm_dev = torch.as_tensor(states, device="cuda")
        avg = hr.rollout(q, qd, work)
        hr.deadlock_step(dl_cfg, x_ee, avg, work, dl_state, dl_goal, sm_state=sm_dev)
        act = ha.compute_action_coupled(q, qd, work)
        act_g = hg.compute_action(q, qd, work)
        sel = torch.as_tensor(states, device="cuda")
        act = torch.where((sel == 2)[None], act_g, act)
        act = torch.where(((sel == 3) | (sel == 5))[None], torch.zeros_like(act), act)
        ha.apply_action(q, qd, act, vel_limit, stop_margin=-1.0)
        hist.append(states.copy())
    return q, qd, np.array(hist), grip, [sm.get_nr_blocks_picked() for sm in sms]


@pytest.mark.parametrize("apply_estimate", [False, True])
def test_episode_with_state_machine_matches_host_stepped_loop(apply_estimate):
    N, B, H, NB, STEPS = 2, 3, 4, 2, 900
    cfg_act = config.panda_config(n_robots=N, horizon=1)
    cfg_roll = config.panda_config(n_robots=N, horizon=H)
    cfg_roll.goal_estimate_mask = 0b10 if apply_estimate else 0
    cfg_grasp = config.panda_config(n_robots=N, horizon=1, n_ego=0)
    ha, hr, hg = FabricHandle(cfg_act, 0), FabricHandle(cfg_roll, 0), FabricHandle(cfg_grasp, 0)
    batch = scenarios.panda_batch(cfg_act, B, seed=11, qd_spread=0.0)
    rows = B * N
    q, qd, prm = (ha.tensor(batch[k]) for k in ("q", "qdot", "params"))
    rng = np.random.default_rng(3)
    start = prm[0:3].clone()                                       # start goals = the scenario's goals
    # blocks on the table in front of each robot, already lifted by 0.1 (EXJ:303); reachable within a few hundred steps
    xe = hr.control_prepare(q, qd, prm, torch.empty_like(prm), apply_estimate=False).cpu().numpy()
    blocks = np.zeros((NB, 3, rows))
    for b in range(NB):
        blocks[b] = xe + np.array([[0.05 * (b + 1)], [-0.08], [-0.25]]) + rng.uniform(-0.02, 0.02, (3, rows))
    blocks = ha.tensor(blocks)
    grip0 = ha.tensor(np.full((2, rows), 0.04))
    loop = ControlLoop(ha, hr, q, qd, prm, config.PANDA_VEL_LIMITS, deadlock=True, apply_estimate=apply_estimate, stop_margin=-1.0,
                       use_graph=True, pick_place=dict(start_goal=start, blocks=blocks, nr_blocks=NB, q_gripper=grip0,
                                                       model=1, h_grasp=hg))
    hist_dev = []
    for t in range(STEPS):
        loop.run(1)
        hist_dev.append(loop.sm_state[abi.SM_STATE].cpu().numpy().copy())
    torch.cuda.synchronize()
    hq, hqd, hist, hgrip, hpicked = _host_episode(hr, ha, hg, q, qd, prm, start, blocks, grip0, NB, STEPS, loop.dl_cfg,
                                                  config.PANDA_VEL_LIMITS, apply_estimate=apply_estimate)
    hist_dev = np.array(hist_dev)
    visited = sorted(set(hist.ravel().tolist()))
    assert {1, 2, 3, 12, 4}.issubset(visited), visited              # the cycle is actually exercised (by robot 0 at least)
    assert (hist_dev == hist).all(), int(np.argmax((hist_dev != hist).any(axis=1)))
    assert loop.sm_state[abi.SM_PICKED].cpu().tolist() == hpicked
    assert float((loop.q - hq).abs().max()) < 1e-9 and float((loop.qdot - hqd).abs().max()) < 1e-8
    assert np.abs(loop.q_gripper.cpu().numpy() - hgrip).max() < 1e-12

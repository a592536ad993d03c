"""Edge cases of the entry points added in round 2: empty batches, ragged tails, the largest robot count, NULL optional
arrays -- through the C ABI on the GPU, against the fused kernel / the oracle."""
import numpy as np
import pytest
import torch

from multi_robot_fabrics_amd import abi, config, scenarios
from multi_robot_fabrics_amd.runtime import FabricHandle
from multi_robot_fabrics_amd.sharded import ShardedRollout

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def test_empty_batches_are_accepted_everywhere():
    cfg = config.panda_config(n_robots=2, horizon=3)
    h = FabricHandle(cfg, 0)
    z7, zp = np.zeros((7, 0)), np.zeros((abi.NPARAM, 0))
    assert h.compute_action_host(z7, z7, zp).shape == (7, 0)
    assert h.rollout_host(z7, z7, zp).shape == (0,)
    x, v, a = h.fk_spheres_host(z7, z7)
    assert x.shape == (8, 3, 0)
    assert h.rollout_cartesian_host(z7, z7, zp, None, None, None, None).shape == (0,)
    for transport in ("rccl", "peer"):
        sr = ShardedRollout(cfg, 0, 1, device_index=0, transport=transport, max_scenarios=4)
        e7 = torch.zeros((7, 0), dtype=torch.float64, device="cuda")
        ep = torch.zeros((abi.NPARAM, 0), dtype=torch.float64, device="cuda")
        assert sr.rollout(e7, e7.clone(), ep).shape == (0,)
        sr.backend.h.comm_status()
    st, sg = h.state_machine_state(torch.zeros((3, 0), dtype=torch.float64, device="cuda"))
    assert st.shape == (abi.SM_NSTATE, 0)


@pytest.mark.parametrize("n_robots,n_scen", [(16, 5), (7, 10), (1, 70)])
def test_peer_kernel_with_many_robots_and_ragged_workgroups(oracle, n_robots, n_scen):
    """cnt_max = n_robots on a single rank: floor(64/n_robots) scenarios per workgroup, the last workgroup partly
    filled; 16 is MRF_MAX_ROBOTS, 1 robot has no exchange partner at all."""
    cfg = config.panda_config(n_robots=n_robots, horizon=3)
    cfg.goal_estimate_mask = ((1 << n_robots) - 1) & ~1
    b = scenarios.panda_batch(cfg, n_scen, seed=4, x_min=0.2 if n_robots > 3 else 0.05, q_spread=0.15 if n_robots > 3 else 0.3)
    want_avg, _, want_qd = oracle.rollout(cfg, b["q"], b["qdot"], b["params"], traj=True)
    for transport in ("peer", "rccl"):
        sr = ShardedRollout(cfg, 0, 1, device_index=0, transport=transport, max_scenarios=n_scen)
        h = sr.backend.h
        q, qd, prm = (h.tensor(b[k]) for k in ("q", "qdot", "params"))
        avg = sr.rollout(q, qd, prm)
        h.comm_status()
        assert rel(avg.cpu().numpy(), want_avg) < 1e-9 and rel(qd.cpu().numpy(), want_qd[-1]) < 1e-9, transport


def test_host_path_optional_arrays_and_static_obstacles(oracle):
    """obst_v / obst_a = NULL (static fabrics hand over positions only), a static prefix, trajectories on and off."""
    cfg = config.panda_config(n_robots=2, horizon=4, dynamic=0)
    b = scenarios.panda_batch(cfg, 6, seed=8)
    h = FabricHandle(cfg, 0)
    sx, _, _ = oracle.fk_spheres(cfg, b["q"], b["qdot"])
    ox, _, _, orad = scenarios.other_robot_obstacles(cfg, b, sx)
    _, want = oracle.compute_action(cfg, b["q"], b["qdot"], b["params"], ox, None, None, orad, n_static=3)
    got = h.compute_action_host(b["q"], b["qdot"], b["params"], ox, None, None, orad, n_static=3)
    assert rel(got, want) < 1e-9
    wavg, wq, wqd = oracle.rollout_cartesian(cfg, b["q"], b["qdot"], b["params"], ox, 0 * ox, 0 * ox, orad, traj=True, n_static=3)
    gavg, gq, gqd = h.rollout_cartesian_host(b["q"], b["qdot"], b["params"], ox, 0 * ox, None, orad, want_traj=True, n_static=3)
    assert rel(gavg, wavg) < 1e-9 and rel(gqd, wqd) < 1e-9
    assert rel(h.rollout_cartesian_host(b["q"], b["qdot"], b["params"], ox, 0 * ox, None, orad, n_static=3), wavg) < 1e-9


def test_collision_mask_in_the_cooperative_kernels_with_five_robots(oracle):
    cfg = config.panda_config(n_robots=5, horizon=3)
    cfg.ego_link_mask = 0b010010
    cfg.kernel_select = 2
    b = scenarios.panda_batch(cfg, 4, seed=3, x_min=0.2, q_spread=0.15)
    want_avg, _, want_qd = oracle.rollout(cfg, b["q"], b["qdot"], b["params"], traj=True)
    h = FabricHandle(cfg, 0)
    avg, tq, tqd = h.rollout(*(h.tensor(b[k]) for k in ("q", "qdot", "params")), want_traj=True)
    assert rel(tqd.cpu().numpy(), want_qd) < 1e-9 and rel(avg.cpu().numpy(), want_avg) < 1e-9


def test_rollout_cartesian_coupled_edge_cases():
    """mrf_rollout_cartesian_coupled: an empty batch is a no-op, ragged row counts and missing arrays are refused, the
    planar model is refused, and the float32 instantiation agrees with the float64 one to float32 accuracy on a
    well-conditioned batch."""
    import ctypes as C
    from multi_robot_fabrics_amd.runtime import MrfError
    cfg = config.panda_config(n_robots=2, horizon=4)
    links, offs = config.sphere_offsets_per_link(2)
    config.set_spheres(cfg, links, offs)
    batch = scenarios.panda_batch(cfg, 12, seed=9, x_min=0.2)
    h = FabricHandle(cfg, 0)
    q, qd, prm = (h.tensor(batch[k]) for k in ("q", "qdot", "params"))
    empty = h.rollout_cartesian_coupled(q[:, :0].contiguous(), qd[:, :0].contiguous(), prm[:, :0].contiguous())
    assert empty.shape == (0,)
    with pytest.raises(MrfError):
        h.rollout_cartesian_coupled(q[:, :3].contiguous(), qd[:, :3].contiguous(), prm[:, :3].contiguous())      # 3 rows, 2 robots
    rc = h.lib.mrf_rollout_cartesian_coupled(h._h, 12, C.c_void_p(q.data_ptr()), None, C.c_void_p(prm.data_ptr()),
                                             C.c_void_p(q.data_ptr()), None, None, None)
    assert rc == abi_status("MRF_E_ARG")
    hp = FabricHandle(config.planar3_config(n_robots=2), 0)
    z = torch.zeros((3, 2), dtype=torch.float64, device="cuda")
    rc = hp.lib.mrf_rollout_cartesian_coupled(hp._h, 1, C.c_void_p(z.data_ptr()), C.c_void_p(z.data_ptr()),
                                              C.c_void_p(z.data_ptr()), C.c_void_p(z.data_ptr()), None, None, None)
    assert rc == abi_status("MRF_E_CONFIG")
    want = h.rollout_cartesian_coupled(q, qd, prm)
    cfg32 = cfg.copy()
    cfg32.scalar = abi.F32
    if abi.has_f32():
        h32 = FabricHandle(cfg32, 0)
        got = h32.rollout_cartesian_coupled(*(h32.tensor(batch[k]) for k in ("q", "qdot", "params")))
        assert torch.allclose(got.double(), want, rtol=2e-3, atol=1e-5)
    else:                                   # the default build: float64 only, and it says so
        from multi_robot_fabrics_amd.runtime import MrfError
        with pytest.raises(MrfError, match="float32"):
            FabricHandle(cfg32, 0)


def abi_status(name):
    return {v: k for k, v in abi.STATUS_TEXT.items()}[name]

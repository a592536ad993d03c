"""Robot-sharded Rollout Fabrics: one robot (or a contiguous group of robots) per GPU, with an exchange between the
ranks after every rollout step (the exchange step is FPJ:211-225).  What a robot sends is cfg.exchange
(include/mrf.h mrf_exchange_kind): its JOINT STATE (cos q, sin q, qdot: 21 scalars -- the receivers re-walk its chain;
the default since round 6) or its predicted collision SPHERES (SX x 9 scalars, the literal all-gather of sphere
centres).  Robots that live on the same rank exchange on chip either way.

    world = sum of group sizes   the ranks are filled with groups of N ranks (one robot per GPU, the north-star layout),
                     then ONE smaller group takes the remaining ranks: 3 robots on 1/2/4/8 GPUs -> [1], [2], [3, 1],
                     [3, 3, 2]; 8 robots on 8 GPUs -> [8].  Inside a group the robots are split into contiguous blocks
                     whose sizes differ by at most one (mrf_comm_partition); the groups are data-parallel replicas on
                     different scenario batches, sized in inverse proportion to the group's robots per rank so that
                     all ranks finish together.
    per rollout step on every rank:
        mrf_step_predict[_joints]  q += dt*qdot for the owned robots; their payload -> own [cnt, XS, B]
        all_gather                 over the ranks of the group (RCCL over xGMI on GPUs; gloo in the CPU tests)
        mrf_step_action[_joints]   fabric solve of the owned robots against everybody else's spheres; qdot := action
    XS = 21 (joints) or SX*9 (spheres), SX = mrf_exchange_spheres.

Three transports for the exchange:
    "torch"  the per-step loop in Python with torch.distributed.all_gather_into_tensor (gloo in the CPU tests of the
             partitioning logic, nccl = RCCL on GPUs) -- the fallback;
    "rccl"   mrf_rollout_sharded with the communicator inside the library: the H-step loop and the ncclAllGather calls
             are issued from C++ on one stream (include/mrf.h); torch.distributed only carries the 128-byte id;
    "peer"   mrf_rollout_sharded over peer-mapped exchange buffers: one persistent kernel per rollout, device-side stores
             into the peers' memory and flag polling; torch.distributed only carries the 64-byte IPC handles.

The collective moves XS*B scalars per owned robot per step: 21*B with the joint payload (B=1: 168 B in f64, whatever
the sphere table), SX*9*B with the sphere payload (SX = 6 for the reference's 8 link-origin spheres, whose coincident
origins of links 1/2 and 5/6 travel once: 432 B; 20 spheres: 1 440 B).  At small B it is latency-bound; batching
scenarios is what makes the link time matter.  The compute backend is injectable so that the
partitioning / gather logic is testable on CPU ranks (tests pass an oracle-backed stand-in); the default backend
is the HIP kernels and there is no CPU fallback.
"""
import os
import time

import torch
import torch.distributed as dist

from . import abi


def group_layout(n_robots, world):
    """Sizes of the robot groups that `world` ranks form: groups of n_robots ranks (one robot per rank) first, then one
    smaller group with the ranks that are left ("G <= N, remaining GPUs split the scenario batch")."""
    if world < 1 or n_robots < 1:
        raise ValueError("world and n_robots must be positive")
    full, rem = divmod(world, n_robots)
    return [n_robots] * full + ([rem] if rem else [])


def rank_placement(n_robots, world, rank):
    """-> (group index, rank inside the group, first global rank of the group, group size)."""
    first = 0
    for gi, size in enumerate(group_layout(n_robots, world)):
        if rank < first + size:
            return gi, rank - first, first, size
        first += size
    raise ValueError(f"rank {rank} outside world {world}")


def robots_per_rank_max(n_robots, G):
    return -(-n_robots // G)


def group_scenarios(n_robots, world, scenarios):
    """Scenario batch of every group when a one-robot-per-rank group gets `scenarios`: a group whose ranks carry c
    robots each takes scenarios // c, so that a rollout costs every rank of the job about the same time."""
    return [max(1, scenarios // robots_per_rank_max(n_robots, size)) for size in group_layout(n_robots, world)]


def robot_partition(n_robots, G):
    """Contiguous robot blocks per group rank: list of (first, count); counts differ by at most one."""
    base, extra = divmod(n_robots, G)
    out, first = [], 0
    for g in range(G):
        cnt = base + (1 if g < extra else 0)
        out.append((first, cnt))
        first += cnt
    return out


def agree_on_error(err, world):
    """Every rank of the job passes its error text (or None) and gets the first one any rank reported: either all ranks
    raise or none does, so that no rank is left alone in a later world-wide collective."""
    if world <= 1 or not dist.is_initialized():
        return err
    errs = [None] * world
    dist.all_gather_object(errs, err)
    return next((f"rank {r}: {e}" for r, e in enumerate(errs) if e), None)


def device_identity(device_index):
    """Name and UUID of the GPU a rank runs on (for logs that must show N ranks sat on N different devices)."""
    props = torch.cuda.get_device_properties(device_index)
    return {"name": props.name, "uuid": str(getattr(props, "uuid", "")), "index": int(device_index)}


_GROUPS = {}     # ranks tuple -> process group: the bench builds four ShardedRollout objects in a row (two transports x two
                 # payloads) over the same robot groups; a communicator per group is made once (every rank asks in the same order)


def _group_of(ranks):
    key = (id(dist.distributed_c10d._get_default_group()), ranks)      # a re-initialised default group starts over
    if key not in _GROUPS:
        _GROUPS[key] = dist.new_group(ranks=list(ranks))
    return _GROUPS[key]


class HipStepBackend:
    def __init__(self, cfg, device_index):
        from .runtime import FabricHandle
        self.h = FabricHandle(cfg, device_index)
        self.dtype, self.device = self.h.dtype, self.h.device
        self.exchange_spheres = self.h.exchange_spheres      # 6 of the 8 link-origin spheres: two pairs coincide

    def prepare(self, n_scen, first, count, q, qd, prm):
        return self.h.step_prepare(n_scen, first, count, q, qd, prm)

    def predict(self, n_scen, first, count, q_io, qd, sph_own):
        self.h.step_predict(n_scen, first, count, q_io, qd, sph_own)

    def action(self, n_scen, first, count, q, qd_io, prm, sph_all, sumsq):
        self.h.step_action(n_scen, first, count, q, qd_io, prm, sph_all, sumsq)

    def predict_joints(self, n_scen, first, count, q_io, qd, jst_own):
        self.h.step_predict_joints(n_scen, first, count, q_io, qd, jst_own)

    def action_joints(self, n_scen, first, count, q, qd_io, prm, jst_all, sumsq):
        self.h.step_action_joints(n_scen, first, count, q, qd_io, prm, jst_all, sumsq)

    def action_predict_joints(self, n_scen, first, count, q_io, qd_io, prm, jst_all, sumsq, jst_next_own):
        """The action of this step and the position update + joint state of the following one in ONE launch."""
        self.h.step_action_predict_joints(n_scen, first, count, q_io, qd_io, prm, jst_all, sumsq, jst_next_own)


class ShardedRollout:
    def __init__(self, cfg, rank, world, backend=None, device_index=0, transport="torch", max_scenarios=None):
        self.cfg = cfg.copy()
        self.N, self.S, self.H = cfg.n_robots, cfg.n_spheres, cfg.horizon
        self.rank, self.world = rank, world
        self.groups = group_layout(self.N, world)
        self.replica, self.grank, self.group_first, self.G = rank_placement(self.N, world, rank)
        self.D = len(self.groups)
        self.parts = robot_partition(self.N, self.G)
        self.first, self.count = self.parts[self.grank]
        self.cnt_max = max(c for _, c in self.parts)
        self.uniform = all(c == self.cnt_max for _, c in self.parts)
        self.backend = backend if backend is not None else HipStepBackend(cfg, device_index)
        self.dtype, self.device = self.backend.dtype, self.backend.device
        self.S = getattr(self.backend, "exchange_spheres", cfg.n_spheres)   # spheres per robot on the wire (sphere payload)
        if cfg.exchange not in abi.EXCHANGE_NAMES:
            raise ValueError("cfg.exchange must be abi.EXCHANGE_JOINTS or abi.EXCHANGE_SPHERES")
        self.exchange = abi.EXCHANGE_NAMES[cfg.exchange]
        self.joints = cfg.exchange == abi.EXCHANGE_JOINTS
        self.XS = abi.JOINT_STATE_SCALARS if self.joints else 9 * self.S     # scalars per robot, scenario and step on the wire
        self.group = None
        if world > 1:
            # one communicator per replica; every rank must take part in every new_group call
            first = 0
            for d, size in enumerate(self.groups):
                g = _group_of(tuple(range(first, first + size))) if size > 1 else None
                if d == self.replica:
                    self.group = g
                first += size
        if not self.uniform:
            # scatter plan from the padded gather buffer [G, cnt_max] to robots [N]
            idx = []
            for g, (f, c) in enumerate(self.parts):
                idx += [g * self.cnt_max + l for l in range(c)]
            self._unpad = torch.tensor(idx, device=self.device)
        self.transport = transport
        if transport not in ("torch", "rccl", "peer"):
            raise ValueError("transport must be 'torch', 'rccl' or 'peer'")
        if transport != "torch":
            # The C handle owns the communicator; only its bootstrap blob travels through torch.  A rank whose local
            # setup fails still takes part in every collective below (with an error marker), so that the whole group
            # raises together instead of leaving the healthy ranks waiting.
            h = self.backend.h
            err = None
            if transport == "rccl":
                box = [None]
                if self.grank == 0:
                    try:
                        box = [h.comm_unique_id()]
                    except Exception as e:       # noqa: BLE001
                        box = ["ERR " + str(e)]
                if self.G > 1:
                    dist.broadcast_object_list(box, src=self.group_first, group=self.group)
                if isinstance(box[0], str):
                    err = box[0]
                else:
                    try:
                        h.comm_init_rccl(self.grank, self.G, box[0])
                    except Exception as e:       # noqa: BLE001
                        err = str(e)
            else:
                if max_scenarios is None:
                    raise ValueError("transport 'peer' needs max_scenarios (capacity of the exchange buffers)")
                try:
                    mine = h.comm_peer_open(self.grank, self.G, max_scenarios)
                except Exception as e:           # noqa: BLE001
                    mine, err = None, str(e)
                handles = [mine]
                if self.G > 1:
                    handles = [None] * self.G
                    dist.all_gather_object(handles, mine, group=self.group)
                if any(x is None for x in handles):
                    err = err or "a peer could not allocate / export its exchange buffer"
                else:
                    try:
                        h.comm_peer_connect(handles)
                    except Exception as e:       # noqa: BLE001
                        err = str(e)
            # agreed over the WORLD, not just the group: the callers go on with world-wide barriers and reductions, which
            # a group that raised alone would leave the other groups waiting in
            err = agree_on_error(err, world)
            if err:
                raise RuntimeError(f"robot-group transport '{transport}' could not be set up: {err}")
            assert h.comm_partition() == (self.first, self.count)

    def own_rows(self, n_scen):
        """Global row indices (scenario*N + robot) of the rows this rank owns, in its local row order."""
        s = torch.arange(n_scen).repeat_interleave(self.count)
        r = torch.arange(self.first, self.first + self.count).repeat(n_scen)
        return s * self.N + r

    def rollout(self, q, qd, prm):
        """q, qd [7, B*count], prm [29, B*count] for the owned robots (local row = scenario*count + l).
        Advances q, qd in place over the horizon; returns avg_vel [B*count]."""
        if self.transport != "torch":
            return self.backend.h.rollout_sharded(q, qd, prm)
        n_scen = q.shape[1] // self.count
        H = self.H
        shape = (abi.JOINT_STATE_SCALARS,) if self.joints else (self.S, 9)      # one robot's payload per scenario
        predict = self.backend.predict_joints if self.joints else self.backend.predict
        action = self.backend.action_joints if self.joints else self.backend.action
        pad = torch.zeros((self.G, self.cnt_max) + shape + (n_scen,), dtype=self.dtype, device=self.device)
        own = pad[self.grank] if self.G == 1 else torch.zeros((self.cnt_max,) + shape + (n_scen,), dtype=self.dtype,
                                                               device=self.device)
        sumsq = torch.zeros((n_scen * self.count,), dtype=self.dtype, device=self.device)
        if (self.cfg.goal_estimate_mask >> self.first) & ((1 << self.count) - 1):
            prm = self.backend.prepare(n_scen, self.first, self.count, q, qd, prm)     # RF-CV goal estimate (EXC:355-357)
        # joint payload: from the second step on, the action launch of step k also does the position update of step k + 1 and
        # writes its joint state into the send block (mrf_step_action_predict_joints: what the in-library RCCL loop runs)
        fused = self.joints and hasattr(self.backend, "action_predict_joints")
        for k in range(H):
            if k == 0 or not fused:
                predict(n_scen, self.first, self.count, q, qd, own[:self.count])
            if self.G > 1:
                dist.all_gather_into_tensor(pad.view(-1), own.view(-1), group=self.group)
            everybody = pad.view((self.G * self.cnt_max,) + shape + (n_scen,))
            if not self.uniform:
                everybody = everybody.index_select(0, self._unpad)
            if fused and k + 1 < H:
                self.backend.action_predict_joints(n_scen, self.first, self.count, q, qd, prm, everybody, sumsq, own[:self.count])
            else:
                action(n_scen, self.first, self.count, q, qd, prm, everybody, sumsq)
        return sumsq / (H * 7)

    # ------------------------------------------------------------------ bench leg (bench.py --shard robots)
    @staticmethod
    def bench(cfg, batch, args, rank, world, local_rank):
        import numpy as np
        transport = getattr(args, "transport", "rccl")
        per_group = group_scenarios(cfg.n_robots, world, args.scenarios)
        gi = rank_placement(cfg.n_robots, world, rank)[0]
        B = per_group[gi]              # this group's batch: args.scenarios for a one-robot-per-rank group
        sr = ShardedRollout(cfg, rank, world, device_index=local_rank, transport=transport, max_scenarios=B)
        if batch is None:
            # the ranks of one robot group work on the SAME scenarios (one batch per group)
            batch = ShardedRollout.replica_batch(cfg, B, rank, world)
        elif batch["q"].shape[1] != B * cfg.n_robots:
            batch = {k: v[:, :B * cfg.n_robots] for k, v in batch.items() if hasattr(v, "shape") and v.ndim == 2}
        rows = sr.own_rows(B).numpy()
        h = sr.backend.h
        q0, qd0, prm = (h.tensor(np.ascontiguousarray(batch[k][:, rows])) for k in ("q", "qdot", "params"))

        def barrier():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        fresh = lambda t: t.clone()
        # Several ranks on ONE device (the single-GPU test layout): every rollout starts from a device synchronisation and a
        # barrier with NOTHING queued in front of its peer kernel.  Measured in round 6: with the other ranks' peer kernels already
        # waiting on the shared device, a small kernel (a copy) that one rank still has to run before its own peer kernel can
        # be held back for seconds -- 2.6 s once, past the 5 s time-out in most runs at 3 x 43 008 scenarios; with no such kernel
        # in front, 5 of 5 runs pass (DESIGN.md section 6 "Residency").  One rank per device -- any real run -- queues its
        # rollouts back to back with their copies.
        shared_device = os.environ.get("MRF_BENCH_SHARE_GPU") == "1" and world > 1

        def one_rollout():
            qi, qdi = fresh(q0), fresh(qd0)
            if shared_device:
                torch.cuda.synchronize()
                dist.barrier()
            return sr.rollout(qi, qdi, prm)

        for _ in range(args.warmup):
            one_rollout()
        barrier()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record()
        for _ in range(args.steps):
            avg = one_rollout()
        ev1.record()
        barrier()
        elapsed = time.perf_counter() - t0
        own_ms = ev0.elapsed_time(ev1) / args.steps          # this rank's own device time per rollout, barrier not included
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        err = None
        if transport != "torch":
            try:
                h.comm_status()          # a timed-out peer exchange is reported here
            except Exception as e:       # noqa: BLE001
                err = str(e)
        if err is None and not bool(torch.isfinite(avg).all()):
            err = "non-finite rollout result"
        if err:     # every rank's own account (the agreed error below is the first rank's only)
            import sys
            print(f"[robot-sharded bench, rank {rank}] {err}", file=sys.stderr, flush=True)
        err = agree_on_error(err, world)
        if err:
            raise RuntimeError(f"robot-sharded rollout ({transport}): {err}")
        # what every rank saw: its communicator (as RCCL itself reports it), its device, its own time per rollout
        mine = {"rank": rank, "group": gi, "rollout_ms": own_ms, "device": device_identity(local_rank),
                "comm": h.comm_info() if transport != "torch" else None,
                "peers": h.comm_peer_info() if transport == "peer" else None}
        ranks = [mine]
        if world > 1:
            ranks = [None] * world
            dist.all_gather_object(ranks, mine)
        # parity of the sharded result with the fused single-GPU kernel on the first scenarios of the replica's batch
        from .runtime import FabricHandle
        nchk = min(B, 64)
        ref = FabricHandle(cfg, local_rank)
        fq, fqd, fprm = (ref.tensor(np.ascontiguousarray(batch[k][:, :nchk * cfg.n_robots])) for k in ("q", "qdot", "params"))
        want = ref.rollout(fq, fqd, fprm)[sr.own_rows(nchk).to(ref.device)]
        got = avg[:nchk * sr.count]
        perr = float((got - want).abs().max() / want.abs().max().clamp_min(1e-300))
        if world > 1:
            t = torch.tensor([perr], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            perr = float(t.item())
        N, H, S = cfg.n_robots, cfg.horizon, sr.S
        sb = 8 if cfg.scalar == abi.F64 else 4
        if transport != "torch":
            h.comm_destroy()        # the exchange buffers (2 x N x XS x B scalars per rank) go back before the next block
        rate = sum(per_group) * args.steps / elapsed            # every group's scenarios over the slowest rank's time
        per_rank = [c for size in sr.groups for _, c in robot_partition(N, size)]
        return {
            "metric": f"rollout control-steps/s, {N}-Panda RF-CV H={H}, robots sharded over GPUs with per-step all-gather",
            "value": rate, "unit": "control-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{N}-Panda RF-CV H={H} coupled rollout only", "scenarios_per_replica": B,
                       "robot_groups": sr.groups, "scenarios_per_group": per_group, "robots_per_rank_all": per_rank,
                       "robot_group_ranks": sr.G, "replicas": sr.D, "robots_per_rank": [c for _, c in sr.parts],
                       "sharding": f"robots ({sr.exchange} payload: {sr.XS} scalars per remote robot, scenario and rollout "
                                   "step; robots of one rank exchange on chip)",
                       "exchange": sr.exchange, "exchange_scalars_per_robot": sr.XS,
                       "exchanged_spheres_per_robot": S if not sr.joints else None},
            "rollout_steps_per_s": rate * N * H,
            "allgather_bytes_per_rank_per_step": sr.cnt_max * sr.XS * B * sb,
            "exchange": sr.exchange,
            "transport": transport,
            "rccl_ranks_seen": sorted({r["comm"]["rccl_comm_count"] for r in ranks if r["comm"]}) if transport == "rccl" else None,
            "rollout_ms_per_rank": {"min": min(r["rollout_ms"] for r in ranks), "max": max(r["rollout_ms"] for r in ranks),
                                    "all": [round(r["rollout_ms"], 4) for r in ranks]},
            "devices": [r["device"] for r in ranks],
            "distinct_devices": len({(r["device"]["uuid"] or r["device"]["index"]) for r in ranks}),
            "ranks": ranks,
            "parity_vs_fused_kernel": {"max_rel_err": perr, "tol": 1e-9 if cfg.scalar == abi.F64 else 2e-3,
                                       "ok": perr <= (1e-9 if cfg.scalar == abi.F64 else 2e-3), "scenarios": nchk},
            "roofline": ShardedRollout.roofline(cfg, sr, B, sb, elapsed / args.steps),
        }

    @staticmethod
    def replica_batch(cfg, B, rank, world):
        from . import scenarios
        return scenarios.panda_batch(cfg, B, seed=1000 + rank_placement(cfg.n_robots, world, rank)[0])

    @staticmethod
    def roofline(cfg, sr, B, sb, seconds_per_rollout):
        """Two views of one robot-sharded rollout (H exchange steps).  Link view: on the xGMI mesh every rank sends its
        payload block to each of the G-1 peers over a separate link, so one directed link carries cnt_max*XS*B scalars
        per step (XS = 21 joint-state scalars or SX*9 sphere scalars; MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU).
        HBM view: SURVEY 8d's algorithmic bytes per rollout-step of the exchanged formulation (spheres through memory)."""
        N, H, S = cfg.n_robots, cfg.horizon, sr.S
        XS = getattr(sr, "XS", 9 * S)
        link_bytes = sr.cnt_max * XS * B * sb if sr.G > 1 else 0
        per_step = seconds_per_rollout / H
        alg = sb * (28 + 9 * S + 9 * S * (N - 1)) + sb * 23 / H            # per (robot, horizon step)
        hbm = alg * sr.count * B / per_step                                # this rank's rows
        nrem = N - sr.count                                                # robots of other ranks, per owned robot
        out = {"bound": "xgmi_link" if sr.G > 1 else "hbm",
               "exchange_bytes_per_row_step": {"read_from_the_local_buffer": XS * sb * nrem,
                                               "stored_into_peer_buffers": XS * sb * (sr.G - 1),
                                               "on_chip_partners": sr.count - 1},
               "hbm_algorithmic": {"achieved": hbm / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": hbm / 8.0e12,
                                   "bytes_per_unit": alg},
               "link": {"bytes_per_link_per_step": link_bytes, "achieved": link_bytes / per_step / 1e9, "peak": 153.0,
                        "unit": "GB/s", "frac": link_bytes / per_step / 153e9, "steps": H}}
        top = out["link"] if sr.G > 1 else out["hbm_algorithmic"]
        out.update(achieved=top["achieved"], peak=top["peak"], unit="GB/s", frac=top["frac"], traffic=None)
        # measured HBM traffic of the transport's kernels (separate --pmc passes, profiles/traffic.json keys
        # sharded_<transport>_<dtype>: bytes per owned row and rollout step), scaled to this rank's rows
        xname = getattr(sr, "exchange", "spheres")
        tr = ShardedRollout._traffic_entry(f"sharded_{sr.transport}_{xname}_{'f64' if sb == 8 else 'f32'}")
        if tr is not None:
            out["traffic"] = tr["bytes_per_row_step"] * sr.count * B / per_step / 1e9
            out["traffic_unit"], out["traffic_key"], out["traffic_kernels"] = "GB/s", tr["key"], tr.get("kernels")
        # Link model, stated so that the first run on real links audits itself: one directed xGMI link
        # carries cnt_max*SX*9*B scalars per step at <= 153 GB/s; the exchanged step is link-bound once that exceeds the
        # fixed cost of an exchange (flag round trip of the peer kernel / launch + collective latency of the RCCL path).
        one = XS * sb                                        # bytes per scenario, link and step with ONE robot per rank
        cm = sr.cnt_max if sr.G > 1 else 1
        fixed_us = {"peer": 4.0, "rccl": 25.0}               # assumed fixed cost per exchange [us]: measured 1.7 us flag
        out["link"].update(                                  # round trip on one die (tools/ipc_probe.hip); RCCL: typical
            predicted_ms_per_step=cm * one * B / 153e9 * 1e3,
            predicted_ms_per_rollout=cm * one * B / 153e9 * 1e3 * H,
            measured_ms_per_step=per_step * 1e3,
            model={"bytes_per_scenario_link_step": cm * one, "payload": xname, "scalars_per_robot": XS,
                   "spheres_payload_bytes_per_scenario_link_step": cm * 9 * S * sb,
                   "link_GBps": 153.0, "assumed_fixed_us_per_exchange": fixed_us,
                   "link_bound_above_scenarios": {k: int(v * 1e-6 * 153e9 / (cm * one)) for k, v in fixed_us.items()},
                   "peer_vs_rccl": "per step the peer kernel costs max(kernel, link) + flag round trip, the RCCL path "
                                   "predict + all-gather + action in stream order = kernels + latency + link: in this "
                                   "model the peer transport is ahead at every batch size; the RCCL path can only cross "
                                   "if the peer kernel's 8-byte remote stores stay below the link rate RCCL reaches -- "
                                   "compare achieved link GB/s of the two transports at the largest batch"})
        return out

    @staticmethod
    def _traffic_entry(key):
        import json
        path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
        try:
            with open(path) as f:
                e = json.load(f).get(key)
        except (OSError, ValueError):
            return None
        return dict(e, key=key) if isinstance(e, dict) and "bytes_per_row_step" in e else None


class InProcessGroup:
    """Every rank of ONE robot group inside this process: one FabricHandle and one stream per rank, the exchange buffers
    connected by their device pointers (mrf_comm_peer_connect_local).  The ranks' persistent peer kernels run side by side
    on the device(s) of the process and exchange exactly as ranks in separate processes do -- without the dispatch
    interference several PROCESSES on one device show, so the multi-rank kernels can be tested and
    timed at production grid sizes on one GPU.  devices: one device index per rank (default: all on device 0)."""

    def __init__(self, cfg, G, max_scenarios, devices=None):
        from .runtime import FabricHandle
        self.cfg, self.G, self.N = cfg.copy(), G, cfg.n_robots
        self.parts = robot_partition(self.N, G)
        devices = list(devices) if devices is not None else [0] * G
        self.handles = [FabricHandle(cfg, devices[g]) for g in range(G)]
        # one stream per rank whose kernels REALLY run beside those of the ranks already placed on the same device: HIP maps
        # streams onto a few hardware queues, and two ranks on one queue would wait for each other until the bounded wait
        # gives up (seen once in ~10 runs of the test suite before this check)
        from .runtime import streams_concurrent
        self.streams = []
        for g in range(G):
            for attempt in range(24):
                st = torch.cuda.Stream(device=devices[g])
                mates = [s for s, d in zip(self.streams, devices) if d == devices[g]]
                if all(st.cuda_stream != m.cuda_stream and streams_concurrent(devices[g], m, st) and streams_concurrent(devices[g], st, m)
                       for m in mates):
                    self.streams.append(st)
                    break
            else:
                raise RuntimeError(f"no stream found whose kernels run beside those of the {len(mates)} ranks already on device "
                                   f"{devices[g]} (hardware queues exhausted: GPU_MAX_HW_QUEUES)")
        for g, h in enumerate(self.handles):
            h.comm_peer_open(g, G, max_scenarios)
        bases = [h.comm_peer_local_base() for h in self.handles]
        for h in self.handles:
            h.comm_peer_connect_local(bases)
            assert h.comm_partition() == self.parts[h.comm_info()["rank"]]

    def own_rows(self, g, n_scen):
        first, count = self.parts[g]
        s = torch.arange(n_scen).repeat_interleave(count)
        r = torch.arange(first, first + count).repeat(n_scen)
        return s * self.N + r

    def rollout(self, states):
        """states[g] = (q, qdot, params) of rank g's owned rows; advances q, qdot in place -> [avg_vel of rank g].  All
        ranks' rollouts are issued (each on its own stream) before anything is waited for."""
        torch.cuda.synchronize()
        out = []
        for h, st, (q, qd, prm) in zip(self.handles, self.streams, states):
            with torch.cuda.stream(st):
                out.append(h.rollout_sharded(q, qd, prm, stream=st))
        for st in self.streams:
            st.synchronize()
        for h in self.handles:
            h.comm_status()
        return out

    def close(self):
        for h in self.handles:
            h.comm_destroy()

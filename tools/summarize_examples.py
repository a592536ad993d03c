#!/usr/bin/env python3
"""gpurun_out/ex/*.json (tools/run_examples.sh) -> one record for profiles/: the headline fields of every driver."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "gpurun_out", "ex")
KEEP = ("success_rate", "n_steps_panda", "n_steps_robot2", "total_time", "dt", "solver_time_mean", "solver_time_std",
        "step_time_mean", "min clearance", "control_steps", "blocks_picked", "states_visited", "time_in_deadlock_steps", "config",
        "solver_time_is", "all_scenes", "distance_to_goal_m", "min_clearance_m")


def load(name):
    path = os.path.join(EX, name)
    try:
        with open(path) as f:
            txt = f.read()
        return json.loads(txt[txt.index("{"):] if txt.lstrip().startswith("{") else txt[txt.index("["):])
    except (OSError, ValueError) as e:
        err = ""
        try:
            err = open(path.rsplit(".", 1)[0] + ".err").read()[-400:]
        except OSError:
            pass
        return {"error": f"{type(e).__name__}: {e}", "stderr_tail": err}


out = {"_doc": "Every example driver and evaluation script run under its reference entry point on one MI355X "
               "(tools/run_examples.sh): default examples/configs/panda_config.yaml (2 Pandas, Rollout Fabrics H=10, deadlock "
               "resolution, n_obst_per_link=4), 6 cubes; the control loop is device-resident (cell.PandaCell), solver times "
               "are device times of one control step (HIP events)."}
for key, name in (("example_pandas_Jointspace --steps 7000", "jointspace.json"),
                  ("example_pandas_cartesian --steps 7000", "cartesian.json"),
                  ("example_pandas_Jointspace --steps 7000 --scenes 64", "jointspace_64scenes.json"),
                  ("example_pointmasses_static", "point_static.json"), ("example_pointmasses_dynamic", "point_dynamic.json")):
    r = load(name)
    out[key] = {k: r[k] for k in KEEP if k in r} if "error" not in r else r
    if "all_scenes" in out[key]:
        a = out[key]["all_scenes"]
        out[key]["all_scenes"] = {"scenes": len(a["success"]), "successes": int(sum(a["success"])),
                                  "min_clearance_m_min": min(a["min_clearance_m"])}
h = load("horizon.json")
out["evaluate_horizon --steps 100"] = h.get("solver_time", h)
try:
    txt = open(os.path.join(EX, "random.txt")).read()
    i = txt.index("{")
    out["evaluate_random_dynamic_scenarios --runs 16 --steps 7000"] = {"table": txt[:i].strip().split("\n"),
                                                                        "cases": json.loads(txt[i:])["cases"]}
except (OSError, ValueError) as e:
    out["evaluate_random_dynamic_scenarios --runs 16 --steps 7000"] = {"error": str(e)}
out["evaluate_random_dynamic_scenarios --device --scenarios 512 --steps 4000 --blocks 2"] = load("random_device.json")
print(json.dumps(out, indent=1))

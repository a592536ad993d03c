#!/usr/bin/env python3
"""Measured-versus-algorithmic HBM bytes per kernel.
usage: kernel_byte_ratios.py <TAG_kernels_pmc.json> <prof_kernels_f64.json>
For every kernel timed by tools/prof_kernels.py: algorithmic bytes per launch (its bytes_per_unit x units), the bytes the
PMC passes saw ((2*FETCH_SIZE + WRITE_SIZE)*1024, FETCH doubled per MI355X_MICROARCH.md), their ratio, and the rates both
give against the 8 TB/s peak, next to the issue-side counters (VALU busy share, wait share, scratch, LDS conflicts)."""
import json
import sys

MATCH = {  # prof_kernels.py label -> substring of the short kernel name (tools/summarize_prof.py)
    "k_fk_spheres_panda": "k_fk_spheres_panda<double>",
    "k_action_panda (M=16 from HBM)": "k_action_panda<double, LS_reference, true>",
    "k_action_panda (M=16 from HBM, obst_a = NULL)": "k_action_panda<double, LS_reference, false>",
    "k_action_coupled": "k_action_coupled<double",
    "k_rollout_panda (H=30)": "k_rollout_panda<double, LS_reference, true>",
    "k_rollout_cart_panda (H=30, M=16)": "k_rollout_cart_panda<double, LS_reference, true, true>",
    "k_rollout_cart_panda (H=30, M=16, obst_a = NULL)": "k_rollout_cart_panda<double, LS_reference, true, false>",
    "k_step_predict": "k_step_predict<double",
    "k_step_action": "k_step_action<double",
}


def main():
    pmc = json.load(open(sys.argv[1]))
    timed = json.load(open(sys.argv[2]))
    out = {"dtype": timed["dtype"], "scenarios": timed["scenarios"], "robots": timed["robots"], "kernels": []}
    for k in timed["kernels"]:
        rec = dict(k)
        sub = MATCH.get(k["kernel"])
        hit = [n for n in pmc if sub and sub in n]
        if len(hit) == 1:
            c = pmc[hit[0]]
            unit_key = next(x for x in k if x.endswith("_per_s"))
            units_per_launch = k[unit_key] * k["ms"] * 1e-3
            alg = units_per_launch * k["bytes_per_unit"]
            meas = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0
            rec.update(pmc_kernel=hit[0], algorithmic_bytes_per_launch=alg, measured_hbm_bytes_per_launch=meas,
                       measured_over_algorithmic=meas / alg if alg else None,
                       measured_GBps=meas / (k["ms"] * 1e-3) / 1e9, measured_frac_of_8TBps=meas / (k["ms"] * 1e-3) / 8e12,
                       vgpr=c.get("vgpr"), agpr=c.get("agpr"), lds_bytes=c.get("lds_bytes"), scratch_bytes=c.get("scratch_bytes"))
            if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
                rec["valu_busy_share_of_wave_cycles"] = c.get("SQ_ACTIVE_INST_VALU", 0.0) / c["SQ_WAVE_CYCLES"]
                rec["wait_inst_any_share_of_wave_cycles"] = c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
            if "SQ_INSTS_LDS" in c and c["SQ_INSTS_LDS"]:
                rec["lds_bank_conflict_cycles_per_lds_inst"] = c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_INSTS_LDS"]
            if "TCC_HIT_sum" in c:
                tot = c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0)
                rec["l2_hit_rate"] = c["TCC_HIT_sum"] / tot if tot else None
        else:
            rec["pmc_kernel"] = None
        out["kernels"].append(rec)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""profiles/r05_wp_pmc.json from the two counter listings tools/pmc_wp.sh writes (gpurun_out/pmc_<tag>.txt): the wave-pair
rollout kernel (MRF_ROLLOUT_WP=1) beside the row kernel on the same box.
usage: python3 tools/wp_pmc_json.py gpurun_out/pmc_r05wp.txt gpurun_out/pmc_r05row.txt [B] > profiles/r05_wp_pmc.json"""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    d = {}
    for line in open(path):
        m = re.match(r"\s+(\S+)\s+([\d.]+)\s+\(n=(\d+)\)", line)
        if m:
            d[m.group(1)] = float(m.group(2))
    return d


def derived(d, waves_per_row, H=30):
    flops = 2 * d["SQ_INSTS_VALU_FMA_F64"] + d["SQ_INSTS_VALU_MUL_F64"] + d["SQ_INSTS_VALU_ADD_F64"]
    cyc = d["GRBM_GUI_ACTIVE"] / 8          # the counter is summed over the 8 XCDs
    return {
        "kernel_cycles_per_xcd": cyc,
        "f64_flop_wave_instructions_per_launch": flops,
        "f64_flop_per_lane_and_rollout_step": flops / d["SQ_WAVES"] / H * waves_per_row,
        "valu_instr_per_wave_step": d["SQ_INSTS_VALU"] / d["SQ_WAVES"] / H,
        "valu_other_than_f64_arith": d["SQ_INSTS_VALU"] - d["SQ_INSTS_VALU_FMA_F64"] - d["SQ_INSTS_VALU_MUL_F64"]
        - d["SQ_INSTS_VALU_ADD_F64"] - d["SQ_INSTS_VALU_TRANS_F64"],
        "valu_busy_fraction_of_simd_time": d["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / cyc,
        "wait_inst_any_fraction_of_wave_cycles": d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"],
        "hbm_write_MB": d["WRITE_SIZE"] / 1e3,
        "FETCH_SIZE_KB_raw": d["FETCH_SIZE"],
    }


def main():
    wp, row = parse(sys.argv[1]), parse(sys.argv[2])
    src = open(os.path.join(ROOT, "multi-robot-fabrics_amd", "csrc", "mrf_rollout_wp.hpp"), "rb").read()
    out = {
        "_meta": {
            "tag": "r05",
            "what": "same box, same inputs (tools/prof_rollout.py 129024 f64): k_rollout_panda_wp (MRF_ROLLOUT_WP=1, a pair of waves "
                    "per row, two waves per SIMD) against k_rollout_panda (one wave per SIMD); separate --pmc passes by "
                    "tools/pmc_wp.sh; counters are per launch, averaged over the launches of a pass; GRBM_GUI_ACTIVE is summed over "
                    "the 8 XCDs",
            "kernel_source_sha256_wp": hashlib.sha256(src).hexdigest(),
            "reading": "the wave pair executes ~99 % of the row kernel's VALU instructions (no VGPR<->AGPR moves: other-than-f64 VALU "
                       "~148 M against ~197 M; +1.9 % f64 work from the second, partial chain walk) but keeps the SIMD's VALU busy ~72 % "
                       "of the time against ~78 %: ~22 % of its wave cycles are s_waitcnt waits (0.9 % in the row kernel, whose single "
                       "wave prefetches every operand one item ahead -- at 128 f64 values per lane the second operand buffer does not "
                       "fit: built, 212 B of scratch per lane, 3.72 against 3.43 ms), ~425 MB of scratch stores per launch, and the "
                       "tail of every step (the two solves, the action, system_step, the next walk) runs on one wave of the pair at a "
                       "time (profiles/r05_wp_phases.txt).",
        },
        "k_rollout_panda_wp": {"counters": wp, "derived": derived(wp, 2), "vgpr": 256, "agpr": 0, "lds_bytes": 40384,
                               "scratch_bytes_per_lane": 140, "waves_per_simd": 2},
        "k_rollout_panda": {"counters": row, "derived": derived(row, 1), "vgpr": 256, "agpr": 250, "lds_bytes": 36992,
                            "scratch_bytes_per_lane": 0, "waves_per_simd": 1},
    }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

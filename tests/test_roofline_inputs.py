"""bench.py's roofline inputs are derived, not copied: every entry of profiles/traffic.json that names
a PMC summary as its `source` must equal what tools/make_traffic.py computes from that file --
bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 and flops_per_unit = (2*FMA + MUL + ADD)/SQ_WAVES/H -- and carry the
kernel-source hash the summary was taken with."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_traffic  # noqa: E402


def test_traffic_json_equals_its_sources():
    with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
        tj = json.load(f)
    checked = 0
    for key, rec in tj.items():
        if key.startswith("_") or ("kernel" not in rec and "kernels" not in rec):
            continue                                   # round-1 record kept for history (hand-entered then)
        if re.fullmatch(r"sharded_(rccl|peer|torch)_(joints|spheres)_(f64|f32)", key):   # robot-sharded transports: bytes per row and step
            with open(os.path.join(ROOT, rec["source"])) as f:
                pmc = json.load(f)
            want = make_traffic.sharded_entry(pmc, [k for k in rec["kernels"]], rec["rows"], rec["steps_per_launch"])
            assert rec["bytes_per_row_step"] == want["bytes_per_row_step"], key
            continue
        m = re.fullmatch(r"rollout_(f64|f32)_N(\d+)_H(\d+)_B(\d+)", key)
        mc = re.fullmatch(r"config_([A-Za-z0-9]+)_(f64|f32)_B(\d+)", key)       # bench.py's `configs` block
        assert m or mc, key
        dtype = m.group(1) if m else mc.group(2)
        with open(os.path.join(ROOT, rec["source"])) as f:
            pmc = json.load(f)
        hits = [k for k in pmc if k.startswith(rec["kernel"]) and not k.startswith("_")]
        assert len(hits) == 1, (key, hits)
        kernel = hits[0]
        want = make_traffic.derive(pmc[kernel], rec["horizon"], dtype, rec.get("work_waves"))
        if rec.get("work_waves"):       # a persistent kernel: the blocks of work follow from the batch in the key
            N = {"C2": 2}[mc.group(1)]
            assert rec["work_waves"] == -(-int(mc.group(3)) // (64 // N)) and pmc[kernel]["SQ_WAVES"] < rec["work_waves"], key
        assert rec["bytes_per_launch"] == want["bytes_per_launch"], key
        assert rec["flops_per_unit"] == want["flops_per_unit"], key
        assert rec["kernel_source_sha256"] == pmc.get("_meta", {}).get("kernel_source_sha256"), key
        if m:
            # the batch in the key is the batch the counters were taken at: grid = waves * 64 lanes, 64 // N scenarios per wave
            N, B = int(m.group(2)), int(m.group(4))
            assert int(m.group(3)) == rec["horizon"] and int(pmc[kernel]["SQ_WAVES"]) == -(-B // (64 // N)), key
        checked += 1
    assert checked >= 5          # the headline and the four configurations of the `configs` block


def test_kernel_source_hash_covers_every_kernel_source():
    """The stamp covers the files kernels are compiled from: the fused kernels, the device header, the exchange tiles and the
    robot-sharded kernels (round 6: the latter three files are new)."""
    import hashlib
    h = hashlib.sha256()
    files = ("mrf_kernels.hip", "mrf_device.hpp", "mrf_tile.hpp", "mrf_shard.hpp", "mrf_comm.hip", "mrf_shard_step.hip")
    for rel in files:
        with open(os.path.join(ROOT, "multi-robot-fabrics_amd", "csrc", rel), "rb") as f:
            h.update(f.read())
    assert make_traffic.kernel_source_sha256() == h.hexdigest()
    assert tuple(os.path.basename(p) for p in make_traffic.KERNEL_SOURCES) == files

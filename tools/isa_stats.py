#!/usr/bin/env python3
"""Instruction statistics of the gfx950 kernels from the compiler's assembly (no GPU needed).
usage: python3 tools/isa_stats.py [substring-of-demangled-kernel-name ...]
Emits, per kernel: VGPR/AGPR/SGPR/scratch/LDS from the .amdhsa_ metadata, static instruction counts by class,
and the same counts for the largest backward-branch loop body (the time-step loop of the rollout kernels)."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# MRF_SRC=mrf_rollout_wp.hip selects the wave-pair kernel's translation unit (its phase markers are the WP_STAMP points)
SRC = os.path.join(ROOT, "multi-robot-fabrics_amd", "csrc", os.environ.get("MRF_SRC", "mrf_kernels.hip"))
ASM = os.environ.get("MRF_ASM", "/tmp/" + os.path.basename(SRC).replace(".hip", ".s"))


def classify(op):
    if op.startswith("v_accvgpr") or (op.startswith("v_mov") and False):
        return "acc_mov"
    if op.startswith(("v_fma_f64", "v_fmac_f64")):
        return "fma64"
    if op.startswith(("v_mul_f64", "v_add_f64")):
        return "muladd64"
    if op.endswith("_f64") or "_f64_" in op:
        return "other64"
    if op.startswith(("v_pk_",)):
        return "pk32"
    if op.startswith("v_mov") or op.startswith("v_cndmask") or op.startswith("v_readlane") or op.startswith("v_writelane"):
        return "mov/sel"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "wait/nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    if not os.path.exists(ASM) or os.path.getmtime(ASM) < max(
            os.path.getmtime(SRC), os.path.getmtime(os.path.join(os.path.dirname(SRC), "mrf_device.hpp")),
            os.path.getmtime(os.path.join(os.path.dirname(SRC), "mrf_rollout_wp.hpp"))):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17",
                               "-ffast-math", "--cuda-device-only", "-S", "-DMRF_ISA_MARKS", "-o", ASM, SRC], stderr=subprocess.DEVNULL)
    lines = open(ASM).read().split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r"^(_Z\w+):", l))]
    names = subprocess.run(["c++filt"], input="\n".join(n for _, n in starts), capture_output=True, text=True).stdout.split("\n")
    pats = sys.argv[1:]
    for (i, sym), dem in zip(starts, names):
        if pats and not all(p in dem for p in pats):
            continue
        end = next(j for j in range(i, len(lines)) if lines[j].startswith("\t.end_amdhsa_kernel") or lines[j].startswith(".Lfunc_end"))
        body = lines[i + 1:end]
        labels, instrs, marks = {}, [], []
        for l in body:
            s = l.strip()
            if s.startswith("; MRFMARK"):
                marks.append((s.split()[2], len(instrs)))
                continue
            if not s or s.startswith((";", ".", "//")) and not re.match(r"^\.LBB\w+:", s):
                if re.match(r"^\.LBB\w+:", s):
                    labels[s.split(":")[0]] = len(instrs)
                continue
            if re.match(r"^\.?\w+:", s):
                labels[s.split(":")[0]] = len(instrs)
                continue
            instrs.append(s.split(";")[0].strip())
        meta = {}
        k0 = next((j for j in range(i, len(lines)) if lines[j].strip() == ".amdhsa_kernel " + sym), len(lines))
        for l in lines[k0:k0 + 80]:
            m = re.match(r"\s*\.amdhsa_(next_free_vgpr|next_free_sgpr|accum_offset|group_segment_fixed_size|private_segment_fixed_size)\s+(\S+)", l)
            if m:
                meta[m.group(1)] = m.group(2)
            if ".end_amdhsa_kernel" in l:
                break
        tot = collections.Counter(classify(x.split()[0]) for x in instrs)
        # largest loop: backward branch with the largest span
        best = (0, 0, 0)
        for k, x in enumerate(instrs):
            p = x.split()
            if p[0].startswith(("s_cbranch", "s_branch")) and p[-1] in labels and labels[p[-1]] <= k:
                if k - labels[p[-1]] > best[0]:
                    best = (k - labels[p[-1]], labels[p[-1]], k)
        print(f"== {dem[:160]}")
        print(f"   meta {meta}")
        print(f"   static total {len(instrs)}: {dict(tot)}")
        if best[0]:
            loop = collections.Counter(classify(x.split()[0]) for x in instrs[best[1]:best[2] + 1])
            print(f"   largest loop {best[0] + 1}: {dict(loop)}")
        prev = None
        for name, pos in marks:             # -DMRF_ISA_MARKS builds: instructions between consecutive markers
            if prev is not None:
                c = collections.Counter(classify(y.split()[0]) for y in instrs[prev[1]:pos])
                print(f"   phase {prev[0]:>14} -> {name:<14} [{prev[1]}:{pos}] {pos - prev[1]:5d}: {dict(c)}")
            prev = (name, pos)
        for k, x in enumerate(instrs):      # every loop (backward branch) with its span
            p = x.split()
            if p[0].startswith(("s_cbranch", "s_branch")) and p[-1] in labels and labels[p[-1]] <= k:
                a = labels[p[-1]]
                c = collections.Counter(classify(y.split()[0]) for y in instrs[a:k + 1])
                print(f"   loop [{a}:{k}] {k - a + 1} instr: {dict(c)}")


if __name__ == "__main__":
    main()

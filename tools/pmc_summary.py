#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc CSV output per kernel: mean counter value per dispatch, and -- when the passes of
tools/pmc_passes.sh are all there -- what follows from them per kernel: instructions per wavefront, wavefronts
in flight per SIMD, VALU instructions in execution per SIMD, lanes active per VALU instruction, cycles per VALU
instruction, L2 hit rate, HBM bytes per launch.  (SQ_WAVE_CYCLES, SQ_ACTIVE_INST_* and SQ_WAIT_* count in units
of four cycles; GRBM_GUI_ACTIVE is the sum over the 8 XCDs; the chip has 1 024 SIMDs.)"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def kernel_name(raw):
    """'void (anonymous namespace)::k_generate<4u, (anonymous namespace)::FastMath>(...)' -> 'k_generate<4u, FastMath>'"""
    n = raw.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"[A-Za-z_0-9:]+(?:<[^()]*>)?", n)
    return m.group(0) if m else n


def main():
    root = sys.argv[1]
    means = defaultdict(dict)  # kernel -> counter -> mean per dispatch (the last pass that has the counter wins)
    for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        with open(path) as f:
            for row in csv.DictReader(f):
                name = kernel_name(row["Kernel_Name"])
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", os.path.relpath(path, root))
        for k, counters in acc.items():
            if not k.startswith("k_"):
                continue
            parts = [f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(counters.items())]
            for c, v in counters.items():
                means[k][c] = sum(v) / len(v)
            n = len(next(iter(counters.values())))
            print(f"  {k} (n={n}): " + " ".join(parts))

    SIMDS, XCDS = 1024, 8
    need = ("SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "GRBM_GUI_ACTIVE")
    rows = [(k, m) for k, m in means.items() if all(c in m for c in need)]
    if rows:
        print("== derived (means over the dispatches of each kernel)")
        for k, m in rows:
            cycles = m["GRBM_GUI_ACTIVE"] / XCDS  # duration of a launch in shader cycles
            line = (f"  {k}: {m['SQ_INSTS_VALU'] / m['SQ_WAVES']:.0f} VALU + {m['SQ_INSTS_SALU'] / m['SQ_WAVES']:.0f} SALU instructions per wavefront, "
                    f"{4 * m['SQ_WAVE_CYCLES'] / (cycles * SIMDS):.2f} wavefronts in flight per SIMD, "
                    f"{4 * m['SQ_ACTIVE_INST_VALU'] / (cycles * SIMDS):.2f} VALU instructions in execution per SIMD, "
                    f"{m['SQ_THREAD_CYCLES_VALU'] / m['SQ_INSTS_VALU']:.1f} lane-cycles per VALU instruction (64 = all lanes; packed ones count more), "
                    f"{cycles * SIMDS / m['SQ_INSTS_VALU']:.2f} cycles per VALU instruction and SIMD")
            if "SQ_WAIT_INST_ANY" in m:
                line += f", waiting for memory {m['SQ_WAIT_INST_ANY'] / m['SQ_WAVE_CYCLES']:.0%} of a wavefront's time"
            if "TCC_HIT_sum" in m and "TCC_REQ_sum" in m and m["TCC_REQ_sum"]:
                line += f", L2 hit rate {m['TCC_HIT_sum'] / m['TCC_REQ_sum']:.0%}"
            if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
                gb = (m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024 / 1e9
                line += f", HBM {gb:.3f} GB per launch = {gb / (cycles / 2.4e9) / 1e3:.2f} TB/s at 2.4 GHz"
            print(line)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel hardware-counter figures for bench.py's roofline objects, from the passes of tools/pmc_passes.sh:

    python tools/kernel_counters.py <config> <summary file name under profiles/> <pmc dir>[:kernel,kernel...] ...   ->   profiles/kernel_counters.json

(a directory may be restricted to some kernels: the exhaustive-scan frames of a run also launch the path kernels, with
more live paths than the default mode has, so k_mesh is taken from a run with those frames and the rest from one without)

Per kernel (mean over its dispatches in the profiled run): VALU wave-instructions per launch (SQ_INSTS_VALU) and HBM
bytes per launch = FETCH_SIZE x correction + WRITE_SIZE (both in KB).  On gfx950 FETCH_SIZE tallies a wide coalesced
read (16 bytes per lane) at half its bytes, so it is doubled for the kernels whose reads are of that kind -- the path
kernels and k_bin64 (state records and queue entries at consecutive positions) and k_mesh (the ray blocks); the reads
of k_group_clusters are 32-byte gathers through the group lists, a width the counter is not calibrated for: taken as
is, a lower bound.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import kernel_name  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALIASES = {"k_group_clusters": "k_group_clusters", "k_shade_fast": "k_shade_fast", "k_shade": "k_shade", "k_mesh": "k_mesh",
           "k_bin64": "k_bin64", "k_generate": None, "k_trace_spheres": None, "k_accumulate": "k_accumulate"}
FETCH_X2 = ("k_shade", "k_shade_fast", "k_generate", "k_trace_spheres", "k_bin64", "k_mesh", "k_accumulate")


def short(name):
    base = name.split("<")[0]
    if base == "k_shade_fast_first":  # the first bounce of the fast arithmetic is a kernel of its own: one of the frame's bounce launches
        return "k_shade_fast"
    if base in ("k_generate", "k_trace_spheres"):  # the arithmetic is a template argument
        return base + ("_fast" if "FastMath" in name else "")
    return base


def main():
    config, source = sys.argv[1], sys.argv[2]
    acc = defaultdict(lambda: defaultdict(list))
    for spec in sys.argv[3:]:
        root, _, only = spec.partition(":")
        only = set(only.split(",")) if only else None
        for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    name = kernel_name(row["Kernel_Name"])
                    if name.startswith("k_") and (only is None or short(name) in only):
                        acc[short(name)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, counters in sorted(acc.items()):
        if not all(c in counters for c in ("SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE")):
            continue
        mean = {c: sum(v) / len(v) for c, v in counters.items()}
        x2 = k.split("_fast")[0] in FETCH_X2 or k in FETCH_X2
        out[k] = {
            "valu_wave_instructions_per_launch": mean["SQ_INSTS_VALU"],
            "FETCH_SIZE_KB": mean["FETCH_SIZE"],
            "WRITE_SIZE_KB": mean["WRITE_SIZE"],
            "fetch_correction": 2 if x2 else 1,
            "hbm_bytes_per_launch": (mean["FETCH_SIZE"] * (2 if x2 else 1) + mean["WRITE_SIZE"]) * 1024,
            "dispatches": len(counters["SQ_INSTS_VALU"]),
            "source": source,
            "note": "rocprofv3 --pmc passes of `bench.py --steps 1 --warmup 1`, mean per dispatch; FETCH_SIZE "
                    + ("doubled (16-byte-per-lane coalesced reads count at half on gfx950)" if x2 else
                       "as reported (32-byte gathers: a width the counter is not calibrated for, a lower bound)"),
        }
    path = os.path.join(ROOT, "profiles", "kernel_counters.json")
    doc = json.load(open(path)) if os.path.exists(path) else {}
    doc[config] = out
    with open(path, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    for k, v in out.items():
        print(f"{k:24s} {v['valu_wave_instructions_per_launch']:.4g} VALU  {v['hbm_bytes_per_launch'] / 1e6:.1f} MB  x{v['dispatches']}")


if __name__ == "__main__":
    main()

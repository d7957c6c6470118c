#!/usr/bin/env python3
"""Registers, LDS and scratch of every kernel, from the assembly `make -C raytracer_amd/csrc asm` writes.
usage: tools/kernel_resources.py [substring of a kernel name ...]"""
import os
import re
import subprocess
import sys

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracer_amd", "csrc", "pathtrace.s")
text = open(path).read()
wanted = sys.argv[1:]
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", text, re.S):
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if wanted and not any(w in name for w in wanted):
        continue
    body = m.group(2)
    get = lambda key: re.search(key + r"\s+(\S+)", body).group(1)
    print(f"{name:44s} vgpr {get(r'.amdhsa_next_free_vgpr'):>4s} sgpr {get(r'.amdhsa_next_free_sgpr'):>4s} "
          f"lds {get(r'.amdhsa_group_segment_fixed_size'):>6s} scratch {get(r'.amdhsa_private_segment_fixed_size'):>4s}")

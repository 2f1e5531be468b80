#!/usr/bin/env python
"""Per-kernel register / scratch / occupancy table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.
usage: python tools/kernel_resources.py dynamic_llava_amd/csrc/linear_packed.hip [name filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dynamic_llava_amd.build_ext import FLAGS, HIPCC
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
r = subprocess.run([HIPCC, *FLAGS, "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur, rows = None, []
for line in r.stderr.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"dl::|\(.*\)$|void ", "", name)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs): (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).split(" ")[0]] = int(m.group(2))
    if "error" in line:
        print(line)
for c in rows:
    if flt in c["name"]:
        print(f"{c['name'][:110]:110s} v={c.get('VGPRs')} a={c.get('AGPRs')} s={c.get('SGPRs')} scratch={c.get('ScratchSize')} occ={c.get('Occupancy')}")

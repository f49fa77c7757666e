#!/usr/bin/env python3
"""dev tool: per-function register / scratch usage from a gfx950 assembly listing (hipcc -save-temps): every function,
not only kernels (the kernel-resource-usage remarks cover kernels only).  usage: fn_resources.py file.s [substring]"""
import re, subprocess, sys
path = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur = None; rows = []
for line in open(path, errors="replace"):
  m = re.match(r"^(_Z\w+):\s", line)
  if m: cur = m.group(1); info = {}
  m = re.match(r"^; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|Occupancy|NumSgprs|codeLenInByte)(?::| =) (\d+)", line)
  if m and cur:
    info[m.group(1)] = int(m.group(2))
    if m.group(1) == "ScratchSize": rows.append((cur, dict(info)))
  m = re.match(r"^; (Occupancy): (\d+)", line)
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
for (mn, info), nm in zip(rows, names):
  nm = re.sub(r"myriad::", "", nm)
  if pat in nm:
    print(f"{info.get('NumVgprs',0):4d} vgpr {info.get('NumAgprs',0):4d} agpr {info.get('ScratchSize',0):6d} scratch {info.get('codeLenInByte',0):7d} B  {nm[:150]}")

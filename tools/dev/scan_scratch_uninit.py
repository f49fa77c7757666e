#!/usr/bin/env python3
"""dev tool / build guard candidate: private-segment (scratch) slots that a function LOADS but never STORES.
A `scratch_load ... offset:N` (frame-relative: off / s32 / s33 based) whose offset no `scratch_store` of the same function covers reads what the
previous kernel on the queue left there -- unless the slot is an incoming stack argument (loads at the caller's outgoing offsets: listed separately,
positive offsets from s32 at function entry).  Flow-insensitive (a store anywhere in the function counts), so it under-reports; what it reports is real.
  usage: scan_scratch_uninit.py file.s [function-substring]"""
import re, subprocess, sys
path = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else ""
funcs = {}; cur = None
for ln, line in enumerate(open(path, errors="replace"), 1):
  m = re.match(r"^(_Z\w+):\s", line)
  if m: cur = m.group(1); funcs[cur] = {"st": {}, "ld": {}}; continue
  if cur is None: continue
  m = re.match(r"\s+scratch_(load|store)_(dword|dwordx2|dwordx3|dwordx4|ubyte|short|sbyte|sshort)\S*\s+(.*)", line)
  if not m: continue
  kind, width, ops = m.group(1), m.group(2), m.group(3)
  nd = {"dword": 1, "dwordx2": 2, "dwordx3": 3, "dwordx4": 4}.get(width, 1)
  off = re.search(r"offset:(-?\d+)", ops); off = int(off.group(1)) if off else 0
  base = "vaddr" if re.search(r"v\d+|v\[", ops.split(",")[0 if kind == "load" and False else 0]) and "off" not in ops.split("offset")[0] else ("s33" if "s33" in ops else ("s32" if "s32" in ops else "off"))
  d = funcs[cur]["st" if kind == "store" else "ld"]
  for k in range(nd): d.setdefault((base, off + 4 * k), ln)
names = dict(zip(funcs, subprocess.run(["c++filt"], input="\n".join(funcs), capture_output=True, text=True).stdout.splitlines()))
bad = 0
for f, d in funcs.items():
  nm = names[f]
  if pat not in nm: continue
  miss = sorted((k, ln) for k, ln in d["ld"].items() if k not in d["st"] and k[0] != "vaddr")
  if miss:
    bad += 1
    print(f"{nm[:140]}\n   loads {len(d['ld'])} slots, stores {len(d['st'])}; loaded but never stored here: " + ", ".join(f"{b}+{o} (line {ln})" for (b, o), ln in miss[:24]) + (" ..." if len(miss) > 24 else ""))
print(f"{bad} function(s) with scratch slots loaded but never stored")

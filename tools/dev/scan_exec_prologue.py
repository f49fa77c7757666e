#!/usr/bin/env python3
"""build guard (round 5): vector instructions AHEAD of the EXEC restore at the top of a join block.

The structurizer ends a divergent region with `s_or_b64 exec, exec, sN` as the first instruction of the join block: from there on every lane that
entered the region is active again.  ROCm 7.2's register allocator sometimes places a SPILL (v_accvgpr_write_b32 aN, vM -- AGPRs serve as spill space
on gfx950 -- or a scratch_store, or a rematerialised v_mov) at the top of that block, in FRONT of the restore: the spill then runs under the region's
partial mask, the lanes that skipped the region never reach the spill slot, and the reload -- under the full mask -- hands them whatever the register
held before: what the previous kernel on that SIMD left there.  That is the "handle-to-handle nondeterminism" of the W > 1 fused kernels with the
speculative rung + the called sweep (round 4; tools/dev/exp/exp57..59: the result follows the inherited contents of a54:a55, the AGPR pair written
at the top of the block that ends the hessian pass, ahead of `s_or_b64 exec`), and it needs no race and no read-before-write in the source.
Reported: label, line, the offending instructions.  Not every hit is fatal (a value that is dead in the inactive lanes may be spilled under a partial
mask), so __graft_entry__.build() prints them as warnings and fails only on AGPR / scratch spills of registers that are RELOADED outside the block.
  usage: scan_exec_prologue.py file.s [more.s ...]"""
import re, sys

VEC = re.compile(r"^\s+(v_accvgpr_write_b32|scratch_store_\w+|v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_read_b32|scratch_load_\w+|v_\w+|ds_\w+|global_\w+|buffer_\w+)\s+(.*)")
RESTORE = re.compile(r"^\s+s_or_b64\s+exec,\s*exec,\s*s\[\d+:\d+\]")

def scan(path, window=24):
  hits = []
  lines = open(path, errors="replace").read().split("\n")
  fn = None
  i = 0
  while i < len(lines):
    m = re.match(r"^(_Z\w+):", lines[i])
    if m: fn = m.group(1)
    if re.match(r"^\.LBB\w+:", lines[i]):
      label = lines[i].split(":")[0]
      pend = []
      j = i + 1; n = 0
      while j < len(lines) and n < window:
        l = lines[j]
        if re.match(r"^\.LBB\w+:|^_Z\w+:|^\s+s_cbranch|^\s+s_branch|^\s+s_endpgm|^\s+s_setpc|^\s+s_swappc|^\s+s_barrier", l): break
        if l.strip().startswith(";") or not l.strip(): j += 1; continue
        n += 1
        if RESTORE.match(l):
          # the signature of the misplaced spill: NOTHING but spill stores (and constant / register moves) between the label and the restore --
          # computation in front of a restore is ordinary code of an enclosing divergent region that an inner region's join interrupts
          spills = [t for k, t in pend if re.match(r"\s+(v_accvgpr_write_b32|scratch_store_)", t)]
          other = [t for k, t in pend if not re.match(r"\s+(v_accvgpr_write_b32|scratch_store_\w+|v_mov_b32_e32|v_mov_b64_e32)\s", t)]
          if spills and not other:
            hits.append((fn, label, j + 1, [t.strip() for k, t in pend]))
          break
        if VEC.match(l): pend.append((j + 1, l))
        j += 1
    i += 1
  return hits

def scan_strict(path, window=48):
  """The wider net (round 6, after the advisor's note that scan() drops a hit as soon as any other vector instruction sits between the label and the
  restore): EVERY AGPR / scratch spill store in a join block's prologue whose slot is reloaded after the restore, whatever else stands in the prologue.
  It also catches what is harmless -- spills of lane-local values in kernels whose lanes are independent trajectories (the lane kernels), and regions
  whose masks are all-or-none by the algorithm (the network passes' wave-uniform branches) -- so it REPORTS (build log, `--strict`) and does not refuse;
  the gate that decides is the register-fill test of tests/test_gpu_poison.py, which runs every multi-wavefront instantiation."""
  hits = []
  lines = open(path, errors="replace").read().split("\n")
  fn = None
  for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m: fn = m.group(1)
    if not re.match(r"^\.LBB\w+:", l): continue
    label = l.split(":")[0]; pend = []; j = i + 1; n = 0
    while j < len(lines) and n < window:
      t = lines[j]
      if re.match(r"^\.LBB\w+:|^_Z\w+:|^\s+s_cbranch|^\s+s_branch|^\s+s_endpgm|^\s+s_setpc|^\s+s_swappc|^\s+s_barrier", t): break
      if t.strip().startswith(";") or not t.strip(): j += 1; continue
      n += 1
      if RESTORE.match(t):
        end = next((q for q in range(j, len(lines)) if lines[q].startswith(".Lfunc_end")), len(lines))
        for k, s in pend:
          m1 = re.match(r"\s+v_accvgpr_write_b32\s+(a\d+),", s)
          m2 = re.match(r"\s+scratch_store_\w+\s+off,\s*\S+,\s*off(?:\s+offset:(\d+))?.*Folded Spill", s)
          if not (m1 or m2): continue
          for q in range(j + 1, end):
            u = lines[q]
            if m1:
              if re.search(r"v_accvgpr_read_b32\s+v\d+,\s*%s\b" % m1.group(1), u): hits.append((fn, label, k, [s.strip(), "reloaded at line %d" % (q + 1)])); break
              if re.search(r"v_accvgpr_write_b32\s+%s," % m1.group(1), u): break
            else:
              off = m2.group(1) or "0"
              if re.search(r"scratch_load_\w+\s+\S+,\s*off,\s*off\s+offset:%s\b" % off, u): hits.append((fn, label, k, [s.strip(), "reloaded at line %d" % (q + 1)])); break
        break
      pend.append((j + 1, t)); j += 1
  return hits


if __name__ == "__main__":
  if "--strict" in sys.argv:
    total = 0
    for p in [a for a in sys.argv[1:] if a != "--strict"]:
      for fn, label, ln, ins in scan_strict(p):
        total += 1
        print(f"{p}:{ln}: {label} in {fn[:90]}: " + "; ".join(ins))
    print(f"{total} spill store(s) in front of an EXEC restore whose slot is reloaded behind it (report only)")
    sys.exit(0)
  total = 0
  for p in sys.argv[1:]:
    for fn, label, ln, ins in scan(p):
      total += 1
      print(f"{p}:{ln}: {label} in {fn[:90]}: spill ahead of the EXEC restore: " + "; ".join(ins[:6]))
  print(f"{total} block(s) with a spill in front of `s_or_b64 exec`")

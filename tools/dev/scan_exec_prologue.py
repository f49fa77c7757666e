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

if __name__ == "__main__":
  total = 0
  for p in sys.argv[1:]:
    for fn, label, ln, ins in scan(p):
      total += 1
      print(f"{p}:{ln}: {label} in {fn[:90]}: spill ahead of the EXEC restore: " + "; ".join(ins[:6]))
  print(f"{total} block(s) with a spill in front of `s_or_b64 exec`")

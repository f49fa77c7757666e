"""Scan an AMDGPU listing for inline-asm DPP moves that sit in the shadow of a matrix instruction.

The hazard recogniser of the compiler inserts the wait states CDNA3/4 require between an MFMA and a VALU instruction that
reads or overwrites its registers -- for instructions it knows.  An `asm volatile("v_mov_b32_dpp ...")` block is opaque
to it: no wait states are inserted in front of the block.  For v_mfma_f64_16x16x4_f64 (16 passes) the documented
distances are 11 wait states from the MFMA to a VALU read / write of its destination and a few to a write of its
source C.  This script reports every asm DPP whose source or destination registers overlap the operands of an MFMA
issued fewer than WINDOW wait states earlier (straight-line count: one per instruction, N+1 per s_nop N).
Usage: python tools/dev/scan_asm_hazard.py x.s [WINDOW]"""
import re, sys

def vregs(tok):
  m = re.match(r'^v\[(\d+):(\d+)\]$', tok)
  if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
  m = re.match(r'^v(\d+)$', tok)
  if m: return {int(m.group(1))}
  return set()

def scan(path, window=19):
  lines = open(path).read().split('\n')
  func = None; hits = []
  hist = []      # (line, text)
  in_asm = False
  for i, l in enumerate(lines):
    t = l.strip()
    if l.startswith('_Z') and t.endswith(':') or (l.startswith('_Z') and ':' in l and '@' in l):
      func = l.split(':')[0][:110]; hist = []
      continue
    if t.startswith(';;#ASMSTART'): in_asm = True; continue
    if t.startswith(';;#ASMEND'): in_asm = False; continue
    if not t or t.startswith((';', '.', '//')) or t.endswith(':'):
      continue
    if in_asm and t.startswith('v_mov_b32_dpp'):
      ops = [o.strip() for o in t.split(None, 1)[1].split(',')]
      d = vregs(ops[0]); s = vregs(ops[1].split()[0])
      ws = 0
      for (j, h) in reversed(hist):
        if h.startswith('s_nop'):
          ws += int(h.split()[1]) + 1
        else:
          ws += 1
        if ws > window: break
        if h.startswith('v_mfma'):
          mo = [o.strip() for o in h.split(None, 1)[1].split(',')]
          dst, a, b, c = (vregs(x.split()[0]) for x in mo[:4])
          kind = []
          if s & dst: kind.append('RAW(reads MFMA result)')
          if d & dst: kind.append('WAW(overwrites MFMA result)')
          if d & c and not (d & dst): kind.append('WAR(overwrites srcC)')
          if d & (a | b): kind.append('WAR(overwrites srcA/B)')
          if kind:
            hits.append(f"{func} line {i+1}: `{t[:60]}` {ws - 1} wait states after line {j+1} `{h[:70]}`: {', '.join(kind)}")
    hist.append((i, t))
    if len(hist) > 80: hist.pop(0)
  return hits

if __name__ == "__main__":
  h = scan(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 19)
  for x in h: print(x)
  print("asm DPP in MFMA shadow:", len(h))

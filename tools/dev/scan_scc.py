"""Scan an AMDGPU assembly listing for a select on SCC whose condition was produced by a VALU compare into VCC.

Seen with ROCm 7.2 hipcc (round 2, shoot_solver_wave.h): a wave-uniform f64 compare feeding `cond ? 1.0 : 0.0` inside a
one-lane region became  v_cmp_nlt_f64 vcc, ...  ;  s_cselect_b32 s, 0x3ff00000, 0  -- the s_cselect reads SCC, which a
vector compare does not write, so the flag took whatever the previous scalar instruction left there.  The pattern flagged:
an s_cselect / s_cbranch_scc whose most recent SCC-writing scalar instruction lies BEFORE the most recent v_cmp .. vcc
(within 6 lines), and either that instruction is neither a scalar compare nor the mask-to-SCC idiom
`s_and_b64 d, mask, exec` (how a wave-uniform condition held as a lane mask legitimately reaches SCC), or the select is
between two literals (the `cond ? 1.0 : 0.0` shape: with two flags in a row the second select re-used the SCC of the
first -- tools/dev/repro/scc_select.hip reproduces exactly that in 25 lines) AND the vector compare's own result is never
read or that SCC has already fed an earlier select (a first select whose neighbouring compare has its own v_cndmask / branch
on vcc is the legitimate interleaving; the miscompiled second flag re-uses a consumed SCC).  A heuristic: it finds the real instances
and nothing else in this library.  __graft_entry__.build() runs it on every translation unit.
Usage: hipcc -S --cuda-device-only ... -o x.s ; python tools/dev/scan_scc.py x.s"""
import re
import sys

# (s_mulk_i32 is NOT in the list: it leaves SCC alone -- round 6: a legitimate `s_cmp_gt_i32 .. ; s_mulk_i32 .. ; <vector code> ; s_cselect_b64` was flagged)
_W = re.compile(r'^\s*s_(add|sub|addc|subb|min|max|and|or|xor|andn2|orn2|nand|nor|xnor|lshl|lshr|ashr|bfe|absdiff|abs|not|wqm|bcnt|cmp|bitcmp|quadmask|addk|cmpk)\w*\s')


_VCC_SRC = re.compile(r'(,\s*vcc\b)|(s_cbranch_vcc)')


def _vcc_consumed(lines, vcmp, sel, ahead=12):
  """True when an instruction after the vector compare at line `vcmp` reads vcc before vcc is written again (looked for up to
  `ahead` lines past the select at line `sel`)."""
  for k in range(vcmp + 1, min(len(lines), sel + 1 + ahead)):
    t = lines[k].strip()
    if not t or t.startswith((';', '.')):
      continue
    if _VCC_SRC.search(t):
      return True
    if re.match(r'^(v_cmp\w*\s+vcc\b|\w+\s+vcc\s*,)', t):      # vcc overwritten unread
      return False
  return False


def scan(path):
  hits = []
  func = None; last_vcmp = None; last_scc = None; last_scc_txt = ''; scc_used = False
  with open(path) as f:
    lines = f.read().split('\n')
  for i, l in enumerate(lines):
    if l.startswith('_Z') and ':' in l:
      func = l.split(':')[0][:90]; last_vcmp = last_scc = None
    t = l.strip()
    if t.startswith('v_cmp') and ' vcc' in t:
      last_vcmp = i
    if _W.match(l):
      last_scc = i; last_scc_txt = t; scc_used = False
    if t.startswith('s_cselect') or t.startswith('s_cbranch_scc'):
      if last_vcmp is not None and last_scc is not None and last_scc < last_vcmp and i - last_vcmp <= 6:
        ops = [o.strip() for o in t.split(None, 1)[1].split(',')][1:] if t.startswith('s_cselect_b32') else []
        # `cond ? 1.0 : 0.0`: the high word of a double constant against 0
        fp_literal_select = len(ops) == 2 and any(o.startswith('0x') for o in ops) and not any(o.startswith(('s', 'v', 'exec', 'm0')) for o in ops)
        scalar_compare = last_scc_txt.startswith(('s_cmp', 's_bitcmp', 's_cmpk'))
        mask_idiom = last_scc_txt.startswith(('s_and_b64', 's_andn2_b64', 's_or_b64')) and re.search(r',\s*exec\b', last_scc_txt) is not None
        if scalar_compare or (mask_idiom and not fp_literal_select):
          continue
        # the vector compare has a consumer of its own (v_cndmask .., vcc / a scalar mask operation / a vcc branch before the
        # next write of vcc): the select then belongs to the mask the scalar instruction turned into SCC, the two are merely
        # interleaved by the scheduler (ocml's pow() is full of this shape)
        if mask_idiom and not scc_used and _vcc_consumed(lines, last_vcmp, i):
          scc_used = True
          continue
        hits.append(f"{func} line {i + 1} | {last_scc_txt} ... {lines[last_vcmp].strip()} -> {t}")
      scc_used = True
  return hits


if __name__ == "__main__":
  h = scan(sys.argv[1])
  for x in h:
    print(x)
  print("suspicious", len(h))

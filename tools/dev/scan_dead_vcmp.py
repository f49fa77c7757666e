"""Scan an AMDGPU assembly listing for SCC consumers that sit behind a DEAD vector compare.

The select-on-stale-SCC miscompile (tools/dev/scan_scc.py, tools/dev/repro/scc_select.hip) leaves a tell-tale: the
compare whose result the program meant is still there as a VALU compare (v_cmp_* vcc / s[n:n+1]), but NOTHING reads
its result -- the consumer was lowered to s_cselect / s_cbranch_scc on an SCC that the vector compare never writes.
A compiler does not emit a dead compare otherwise (DCE removes it), so "a v_cmp whose destination is overwritten or
left unread, followed within a few instructions by an SCC consumer with no scalar compare in between" is a much
wider net than scan_scc.py's shape matching.  Straight-line liveness only: the look-ahead stops at labels and
branches (then the compare is assumed live).
Usage: python tools/dev/scan_dead_vcmp.py x.s"""
import re
import sys

_SREG = re.compile(r'\bs\[(\d+):(\d+)\]|\bs(\d+)\b|\bvcc(_lo|_hi)?\b')
_SCC_W = re.compile(r'^s_(add|sub|addc|subb|min|max|and|or|xor|andn2|orn2|nand|nor|xnor|lshl|lshr|ashr|bfe|absdiff|abs|not|wqm|bcnt|cmp|bitcmp|quadmask|addk|cmpk|mulk|bfm|ff|flbit|sext|brev|lshl1|lshl2|lshl3|lshl4)\w*\s')
_SCC_R = re.compile(r'^(s_cselect|s_cbranch_scc|s_addc|s_subb|s_cmov)')


def _regs(tok):
  """set of 32-bit scalar register names a textual operand covers"""
  out = set()
  tok = tok.strip()
  m = re.match(r'^s\[(\d+):(\d+)\]$', tok)
  if m:
    return {f"s{i}" for i in range(int(m.group(1)), int(m.group(2)) + 1)}
  if re.match(r'^s\d+$', tok):
    return {tok}
  if tok == 'vcc':
    return {'vcc_lo', 'vcc_hi'}
  if tok in ('vcc_lo', 'vcc_hi'):
    return {tok}
  return out


def _operands(t):
  parts = t.split(None, 1)
  if len(parts) < 2:
    return []
  body = parts[1].split(';')[0]
  # keep s[a:b] together
  ops = re.findall(r's\[\d+:\d+\]|[^,\s]+', body)
  return ops


def scan(path, ahead=400, near=12):
  with open(path) as f:
    lines = f.read().split('\n')
  ins = []     # (line_no, text)
  func = {}
  cur = None
  for i, l in enumerate(lines):
    if l.startswith('_Z') and l.rstrip().endswith(':'):
      cur = l.split(':')[0][:100]
    t = l.strip()
    if not t or t.startswith((';', '.', '//')):
      if t.startswith('.LBB') or (t.endswith(':') and not t.startswith(';')):
        ins.append((i, 'LABEL'))
      continue
    if t.endswith(':'):
      ins.append((i, 'LABEL'))
      continue
    ins.append((i, t))
    func[i] = cur
  hits = []
  for idx, (ln, t) in enumerate(ins):
    if not t.startswith('v_cmp') or t.startswith('v_cmpx'):
      continue
    ops = _operands(t)
    if not ops:
      continue
    # VOPC e32 writes vcc implicitly: "v_cmp_lt_f64_e32 vcc, v[..], v[..]"; e64: first operand is the sdst
    dst = _regs(ops[0])
    if not dst:
      continue
    live = None
    scc_consumer = None
    scalar_cmp_between = False
    for k in range(idx + 1, min(len(ins), idx + 1 + ahead)):
      ln2, t2 = ins[k]
      if t2 == 'LABEL' or t2.startswith(('s_branch', 's_cbranch', 's_endpgm', 's_setpc', 's_swappc')):
        if t2.startswith('s_cbranch_scc') and scc_consumer is None and not scalar_cmp_between and k - idx <= near:
          scc_consumer = (ln2, t2)
        if t2.startswith('s_cbranch_vcc') and dst & {'vcc_lo', 'vcc_hi'}:
          live = True
        break
      ops2 = _operands(t2)
      mnem = t2.split()[0]
      # reads: every operand except the first (destination) -- plus implicit vcc reads
      srcs = set()
      for o in ops2[1:]:
        srcs |= _regs(o)
      if mnem.startswith(('v_cndmask', 'v_addc', 'v_subb', 'v_subbrev', 'v_div_fmas')) and len(ops2) <= 3:
        srcs |= {'vcc_lo', 'vcc_hi'}      # e32 forms read vcc implicitly
      if mnem.startswith(('s_and_saveexec', 's_or_saveexec', 's_andn2_saveexec', 's_xor_saveexec')):
        pass
      # stores / branches with a single operand read it
      if len(ops2) == 1:
        srcs |= _regs(ops2[0]) if not mnem.startswith(('s_mov', 'v_mov')) else set()
      if srcs & dst:
        live = True
        break
      if _SCC_R.match(t2) and scc_consumer is None and not scalar_cmp_between and k - idx <= near:
        scc_consumer = (ln2, t2)
      if _SCC_W.match(t2):
        scalar_cmp_between = True
      wr = _regs(ops2[0]) if ops2 else set()
      if mnem.startswith('v_cmp') and not mnem.startswith('v_cmpx') and len(ops2) and not wr:
        wr = set()
      if wr and wr >= dst:
        live = False
        break
      if wr & dst:
        dst = dst - wr
        if not dst:
          live = False
          break
    if live is False and scc_consumer is not None:
      hits.append(f"{func.get(ln)} line {ln + 1}: dead `{t}` then line {scc_consumer[0] + 1} `{scc_consumer[1]}`")
  return hits


if __name__ == "__main__":
  h = scan(sys.argv[1])
  for x in h:
    print(x)
  print("suspicious", len(h))

"""dev tool: scan an AMDGPU assembly listing for a select on SCC whose condition was produced by a VALU compare into VCC
(seen with ROCm 7.2 hipcc: a wave-uniform f64 compare feeding `cond ? 1.0 : 0.0` became v_cmp_*_f64 vcc ; s_cselect_b32 --
the s_cselect then reads whatever the last scalar instruction left in SCC).  Usage: hipcc -S --cuda-device-only ... -o x.s;
python tools/dev/scan_scc.py x.s"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
W = re.compile(r'^\s*s_(add|sub|addc|subb|min|max|and|or|xor|andn2|orn2|nand|nor|xnor|lshl|lshr|ashr|bfe|absdiff|abs|not|wqm|bcnt|cmp|bitcmp|quadmask|addk|cmpk|mulk)\w*\s')
func = None; last_vcmp = None; last_scc = None; last_scc_txt = ''; hits = 0
for i, l in enumerate(lines):
  if l.startswith('_Z') and ':' in l:
    func = l.split(':')[0][:90]; last_vcmp = last_scc = None
  t = l.strip()
  if t.startswith('v_cmp') and ' vcc' in t: last_vcmp = i
  if W.match(l): last_scc = i; last_scc_txt = t
  if t.startswith('s_cselect') or t.startswith('s_cbranch_scc'):
    if last_vcmp is not None and last_scc is not None and last_scc < last_vcmp and i - last_vcmp <= 6 and not last_scc_txt.startswith(('s_cmp', 's_bitcmp', 's_cmpk')):
      hits += 1
      print(func, 'line', i + 1, '|', last_scc_txt, '...', lines[last_vcmp].strip(), '->', t)
print("suspicious", hits)

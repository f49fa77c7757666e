"""The build's guard against a hipcc miscompile seen in round 2 (tools/dev/scan_scc.py, run by __graft_entry__.build() on the
device assembly of every translation unit): a wave-uniform f64 compare selecting between two constants was lowered to
v_cmp .. vcc + s_cselect_b32 on SCC, which the vector compare does not write."""
import importlib.util
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _scanner():
  spec = importlib.util.spec_from_file_location("scan_scc", os.path.join(ROOT, "tools", "dev", "scan_scc.py"))
  m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
  return m


def test_scanner_flags_the_pattern_and_passes_the_legitimate_idioms(tmp_path):
  bad = tmp_path / "bad.s"
  bad.write_text("_Z1kv:\n\ts_subb_u32 s1, s15, s11\n\tv_cmp_nlt_f64_e32 vcc, v[2:3], v[4:5]\n\ts_cselect_b32 s10, 0x3ff00000, 0\n\ts_endpgm\n"
                 "_Z2k2v:\n\ts_and_b64 s[0:1], s[0:1], exec\n\tv_cmp_lt_f64_e32 vcc, v[10:11], v[12:13]\n\ts_cselect_b32 s0, 0x3ff00000, 0\n\ts_endpgm\n")
  good = tmp_path / "good.s"
  good.write_text("_Z1gv:\n\ts_cmp_lg_u32 s50, 11\n\tv_cmp_nlt_f64_e32 vcc, s[4:5], v[66:67]\n\ts_cselect_b64 s[4:5], -1, 0\n\ts_endpgm\n"
                  "_Z2g2v:\n\ts_and_b64 s[14:15], s[76:77], exec\n\tv_cmp_gt_f64_e64 vcc, v[10:11], |v[94:95]|\n\ts_cselect_b32 s53, s3, s17\n\ts_endpgm\n"
                  "_Z2g3v:\n\tv_cmp_nlt_f64_e64 s[0:1], v[10:11], v[12:13]\n\ts_and_b64 s[0:1], s[0:1], exec\n\ts_cselect_b32 s0, 0x3ff00000, 0\n\ts_endpgm\n"
                  # ocml pow(): the mask-to-SCC idiom and its literal select interleaved with an unrelated compare whose vcc is consumed
                  "_Z2g4v:\n\ts_and_b64 s[4:5], s[4:5], exec\n\tv_cndmask_b32_e32 v14, v16, v14, vcc\n\tv_cmp_class_f64_e32 vcc, s[14:15], v15\n"
                  "\ts_cselect_b32 s4, 0, 0x7ff00000\n\tv_mov_b32_e32 v15, s4\n\tv_cndmask_b32_e32 v12, v12, v13, vcc\n\ts_endpgm\n")
  sc = _scanner()
  assert len(sc.scan(str(bad))) == 2
  assert sc.scan(str(good)) == []


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_reproducer_of_the_select_on_scc_lowering(tmp_path):
  """tools/dev/repro/scc_select.hip: either this compiler still produces the pattern and the scanner finds it, or the compiler
  has been fixed and the listing holds no select on SCC between the two double constants without a scalar compare."""
  out = tmp_path / "scc_select.s"
  subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only", os.path.join(ROOT, "tools", "dev", "repro", "scc_select.hip"),
                  "-o", str(out)], check=True, capture_output=True)
  txt = out.read_text()
  hits = _scanner().scan(str(out))
  selects = re.findall(r"s_cselect_b32 s\d+, 0x3ff00000, 0", txt)
  if selects:                      # the flags are still materialised through SCC: the second one has no compare of its own
    assert hits, "scanner missed the known miscompile"
  else:
    assert hits == []

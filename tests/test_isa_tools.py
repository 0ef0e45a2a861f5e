"""tools/isa_branches.py and tools/isa_sections.py read a `-DHLX_MARKS -S` listing: checked here on a small synthetic kernel
(the layout rules they serve -- rare paths out of line, not jumped over -- are DESIGN.md section 5)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LISTING = """
\t.text
_ZN12_GLOBAL__N_114hlx_env_kernelILj1ELi0ELb0ELb0ELb1ELi1EEEvv:
\ts_load_dword s0, s[4:5], 0x0
\t; HLXMARK 0
\tv_add_f32_e32 v1, v2, v3
\ts_and_saveexec_b64 s[2:3], vcc
\ts_cbranch_execz .LBB0_2
\tv_mul_f32_e32 v1, v1, v1
\tv_mul_f32_e32 v1, v1, v1
.LBB0_2:
\ts_or_b64 exec, exec, s[2:3]
\t; HLXMARK 1
\ts_cbranch_execnz .LBB0_9
.LBB0_3:
\tv_add_f64 v[4:5], v[4:5], v[6:7]
\tglobal_store_dword v0, v1, s[0:1]
\ts_endpgm
.LBB0_9:
\tv_mov_b32_e32 v1, 0
\ts_branch .LBB0_3
\t.section\t.rodata
\t.end_amdhsa_kernel
"""


def run(tool, *args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), *args], check=True, capture_output=True, text=True).stdout


def test_branch_listing_reports_sections_targets_and_distances(tmp_path):
    p = tmp_path / "k.s"
    p.write_text(LISTING)
    out = run("isa_branches.py", str(p), "ILj1ELi0ELb0ELb0ELb1ELi1").splitlines()
    rows = [l.split() for l in out if l.strip() and l.split()[0].isdigit()]
    # position, "sec", section, opcode, target, distance
    assert [(r[2], r[3], r[4], r[5]) for r in rows] == [("0", "s_cbranch_execz", ".LBB0_2", "2"), ("1", "s_cbranch_execnz", ".LBB0_9", "3"),
                                                        ("1", "s_branch", ".LBB0_3", "-5")]
    assert out[-1].startswith("instructions 13") and out[-1].endswith("2")        # two short forward conditional skips
    only = run("isa_branches.py", str(p), "ILj1ELi0ELb0ELb0ELb1ELi1", "1", "1")
    assert "s_cbranch_execz" not in only and "s_cbranch_execnz" in only


def test_section_counts_split_at_the_marks(tmp_path):
    p = tmp_path / "k.s"
    p.write_text(LISTING)
    out = run("isa_sections.py", str(p), "ILj1ELi0ELb0ELb0ELb1ELi1")
    assert "total 13" in out
    rows = [l for l in out.splitlines() if l.startswith(("entry", "after mark"))]
    assert [int(l.split("%")[0].split()[-2]) for l in rows] == [1, 6, 6]
    assert "'f64': 1" in out and "'mem': 1" in out

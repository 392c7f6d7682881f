"""tools/asm_profile.py (the dynamic instruction profile of DESIGN.md section 7) on a synthetic piece of assembly: every
straight-line run gets its counter, the kernel descriptor gets room for the extra registers and LDS, and the report
multiplies execution counts with run lengths.  (The real thing needs a GPU: tools/profile_round.sh.)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "asm_profile.py")
K = "_ZN2pt16integrate_kernelILb0ELb0ELb0ELb0ELb0ELi0EEEvNS_10RenderArgsE"

ASM = f"""\t.file\t0 "/x" "pt_kernels.hip"
\t.text
{K}:
\t.loc\t0 10 1
\ts_load_dwordx2 s[2:3], s[0:1], 0x0
\tv_mov_b32_e32 v1, 0
.LBB0_1:
\t.loc\t0 20 1
\tv_add_u32_e32 v1, 1, v1
\tv_cmp_gt_u32_e32 vcc, 5, v1
\ts_cbranch_vccnz .LBB0_1
\t.loc\t0 30 1
\tv_mov_b32_e32 v2, v1
\ts_endpgm
.Lfunc_end0:
\t.amdhsa_kernel {K}
\t\t.amdhsa_group_segment_fixed_size 5128
\t\t.amdhsa_next_free_vgpr 80
\t\t.amdhsa_next_free_sgpr 100
\t\t.amdhsa_accum_offset 80
\t.end_amdhsa_kernel
"""


def test_instrument_and_report(tmp_path):
    src, out, mp, cnt = (str(tmp_path / n) for n in ("k.s", "k_inst.s", "map.json", "counts.txt"))
    open(src, "w").write(ASM)
    subprocess.run([sys.executable, TOOL, "instrument", src, out, mp, "--kernels", K, "--kernarg-offset", "256"], check=True)
    text = open(out).read()
    runs = json.load(open(mp))["runs"][K]
    # entry run, loop body (label), fall-through after the branch
    assert [len(r) for r in runs] == [2, 3, 1]
    assert text.count("ds_add_u32 v110, v111") == 3
    assert text.count("s_mov_b64 exec, s[100:101]") == 3
    assert "s_load_dwordx2 s[100:101], s[0:1], 256" in text             # the counters' base comes from the kernel argument
    assert text.index("global_atomic_add") < text.index("s_endpgm")       # flushed before the wave ends
    assert ".amdhsa_next_free_vgpr 120" in text and ".amdhsa_next_free_sgpr 102" in text
    assert ".amdhsa_group_segment_fixed_size 12288" in text
    # every original instruction is still there, in order
    orig = [l.strip() for l in ASM.split("\n") if l.startswith("\t") and not l.strip().startswith(".")]
    pos = 0
    for ins in orig:
        pos = text.index(ins, pos) + 1
    # the loop ran 5 times in each of 7 waves
    open(cnt, "w").write("0 7\n1 35\n2 7\n")
    r = subprocess.run([sys.executable, TOOL, "report", mp, cnt, "--kernel", K, "--wave-segments", "7"],
                       capture_output=True, text=True, check=True).stdout
    assert f"{7 * 2 + 35 * 3 + 7 * 1} instructions executed" in r
    assert "pt_kernels.hip:20" in r


def test_pipe_cycle_table():
    """The weights of the report's "pipe cycles" follow tools/issue_cost.hip (profiles/r02_issue_cost.txt): 2 cycles for the plain
    fp32 / add / logic forms with vector operands only, 4 with a scalar operand, as DPP and for the rest, 8 for v_rcp / v_rsq."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import asm_profile as A
    assert A.pipe_cycles("v_fma_f32", "v1, v2, v3, v4") == ("valu", 2.0)
    assert A.pipe_cycles("v_mul_f32_e32", "v1, 0x40490fdb, v2") == ("valu", 2.0)          # a literal is not a scalar register
    assert A.pipe_cycles("v_mul_f32_e32", "v1, s7, v2") == ("valu", 4.0)
    assert A.pipe_cycles("v_fma_f32", "v1, v2, s[4:5], v4")[1] == 4.0
    assert A.pipe_cycles("v_cndmask_b32_e32", "v1, v2, v3, vcc") == ("valu", 4.0)
    assert A.pipe_cycles("v_add_u32_dpp", "v1, v2, v1 row_shr:1 row_mask:0xf bank_mask:0xf") == ("valu", 4.0)
    assert A.pipe_cycles("v_cvt_f32_ubyte0_e32", "v1, v2") == ("valu", 4.0)
    assert A.pipe_cycles("v_pk_fma_f32", "v[0:1], v[2:3], v[4:5], v[6:7]") == ("valu", 4.0)
    assert A.pipe_cycles("v_rcp_f32_e32", "v1, v2") == ("valu", 8.0)
    assert A.pipe_cycles("s_add_u32", "s1, s2, s3") == ("salu", 4.0)
    assert A.pipe_cycles("s_cbranch_vccnz", ".LBB0_1") == ("salu", 4.0)
    assert A.pipe_cycles("s_nop", "0")[1] < 1.0
    assert A.pipe_cycles("ds_write_b32", "v1, v2")[0] == "lds" and A.pipe_cycles("global_load_dwordx4", "v[0:3], v[4:5], off")[0] == "vmem"

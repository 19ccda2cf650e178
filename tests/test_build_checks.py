"""The build-time assembly checks of blockcg_amd/csrc/Makefile do fail when they should: tools/check_dpp_hazard.py (a VALU
write of a v_fmac_f64_dpp broadcast operand too close in front of it) and tools/check_async_regs.py (a register of a
hand-waited load touched between its issue and the wait that retires it), on small synthetic listings."""
import os
import subprocess
import sys

from conftest import ROOT


def _run(tool, text, tmp_path):
    f = tmp_path / "k.s"
    f.write_text(text)
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), str(f)], capture_output=True, text=True)


DPP = "\tv_fmac_f64_dpp v[10:11], v[20:21], v[30:31] row_newbcast:3 row_mask:0xf bank_mask:0xf\n"


def test_dpp_hazard_check(tmp_path):
    ok = "\tds_read_b128 v[20:23], v5\n\ts_waitcnt lgkmcnt(0)\n" + DPP + DPP
    assert _run("check_dpp_hazard.py", ok, tmp_path).returncode == 0
    # a copy into the broadcast operand right in front of its use: one instruction of distance, two are needed
    bad = "\tv_mov_b32_e32 v20, v40\n\ts_mov_b32 s0, s1\n" + DPP
    r = _run("check_dpp_hazard.py", bad, tmp_path)
    assert r.returncode == 1 and "1 with a VALU write" in r.stdout
    # far enough (two other instructions in between), or the s_nop the hazard asks for
    assert _run("check_dpp_hazard.py", "\tv_mov_b32_e32 v20, v40\n\ts_mov_b32 s0, s1\n\ts_mov_b32 s2, s3\n" + DPP, tmp_path).returncode == 0
    assert _run("check_dpp_hazard.py", "\tv_mov_b32_e32 v20, v40\n\ts_nop 1\n" + DPP, tmp_path).returncode == 0
    # a VALU write of EXEC within five instructions
    assert _run("check_dpp_hazard.py", "\tv_cmpx_gt_u32_e32 v1, v2\n\ts_mov_b32 s0, s1\n\ts_mov_b32 s2, s3\n" + DPP, tmp_path).returncode == 1
    # a listing without the instruction is reported, not failed (the row-kernel file has none)
    assert _run("check_dpp_hazard.py", "\ts_mov_b32 s0, s1\n", tmp_path).returncode == 0


def _kernel(body):
    return "_ZN3bcg12_GLOBAL__N_17k_hop4bILi16ELi0EEEv:\n" + body + "\ts_endpgm\n"


def test_async_register_check(tmp_path):
    issue = ("\t; ASYNC_ISSUE n1\n\tglobal_load_dwordx4 v[8:11], v[2:3], off\n\t; ASYNC_ISSUED n1\n")
    retire = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(3) ; ASYNC_RETIRE n1\n\tv_mov_b64 v[40:41], v[8:9]\n\t;;#ASMEND\n"
    good = _kernel(issue + "\tv_fma_f64 v[20:21], v[22:23], v[24:25], v[20:21]\n" + retire)
    assert _run("check_async_regs.py", good, tmp_path).returncode == 0
    # the allocator copies the destination in front of the wait: read before it arrives
    bad = _kernel(issue + "\tv_mov_b32_e32 v50, v9\n" + retire)
    r = _run("check_async_regs.py", bad, tmp_path)
    assert r.returncode == 1 and "touched in between: 1" in r.stdout
    # no hand-waited groups at all is an error too (not a PIPE build)
    assert _run("check_async_regs.py", _kernel("\ts_mov_b32 s0, s1\n"), tmp_path).returncode == 1


def test_async_register_check_fails_on_a_loop_without_a_retire(tmp_path):
    """A path that comes round to the next issue (or leaves the kernel) without passing a retiring wait: the loads of the
    step before are still in flight when their registers are re-issued.  Tolerated only as the two statically infeasible
    paths of the groups n1 / n2 (the walker is not path-sensitive, tools/check_async_regs.py); one of group p, or a second
    one of n1, fails the build."""
    def step(tag, reg):
        return (f"\t; ASYNC_ISSUE {tag}\n\tglobal_load_dwordx4 v[{reg}:{reg + 3}], v[2:3], off\n\t; ASYNC_ISSUED {tag}\n")
    retire = lambda tag: f"\t;;#ASMSTART\n\ts_waitcnt vmcnt(3) ; ASYNC_RETIRE {tag}\n\t;;#ASMEND\n"  # noqa: E731
    # group p in a loop whose back edge skips the retire
    loop_p = _kernel(".LBB0_1:\n" + step("p", 8) + "\ts_cbranch_scc1 .LBB0_1\n" + retire("p"))
    r = _run("check_async_regs.py", loop_p, tmp_path)
    assert r.returncode == 1 and "NOT on the allow-list" in r.stdout and "reached the next issue" in r.stdout
    # the same shape for n1 is the tolerated infeasible path ...
    loop_n1 = _kernel(".LBB0_1:\n" + step("n1", 8) + "\ts_cbranch_scc1 .LBB0_1\n" + retire("n1"))
    assert _run("check_async_regs.py", loop_n1, tmp_path).returncode == 0
    # ... but two back edges without a retire are one too many
    loop_n1_twice = _kernel(".LBB0_1:\n" + step("n1", 8) + "\ts_cbranch_scc1 .LBB0_1\n\ts_cbranch_vccz .LBB0_2\n" + retire("n1")
                            + "\ts_branch .LBB0_3\n.LBB0_2:\n\ts_nop 0\n.LBB0_4:\n" + step("n1", 8) + retire("n1") + ".LBB0_3:\n")
    r = _run("check_async_regs.py", loop_n1_twice, tmp_path)
    assert r.returncode == 1 and "NOT on the allow-list" in r.stdout
    # a path that runs off the end of the kernel with group p in flight
    off_end = _kernel(step("p", 8) + "\ts_cbranch_scc1 .LBB0_9\n" + retire("p") + ".LBB0_9:\n\ts_nop 0\n")
    r = _run("check_async_regs.py", off_end, tmp_path)
    assert r.returncode == 1 and "ran off the end" in r.stdout

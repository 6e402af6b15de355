"""CPU: tools/vgpr_liveness.py -- the backward-dataflow tool the register-resident kernels were tuned with.  A hand-made
kernel with a loop and a diamond: the live set at the maximum and at the labels must be what the dataflow says, and a
register that is only written in the two arms of a branch must NOT be reported live above them (the tool works on the
text, it does not model the backend's phantom path -- that is what comparing its numbers with the compiler's shows)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "vgpr_liveness.py")

KERNEL = """
toy_kernel:                             ; @toy_kernel
	v_mov_b32_e32 v0, 0
	v_mov_b32_e32 v1, 1
.LBB0_1:
	s_cmp_eq_u32 s0, 0
	s_cbranch_scc1 .LBB0_3
	v_add_f64 v[4:5], v[0:1], v[0:1]
	s_branch .LBB0_4
.LBB0_3:
	v_mul_f64 v[4:5], v[0:1], v[0:1]
.LBB0_4:
	v_fma_f64 v[6:7], v[4:5], v[0:1], v[4:5]
	global_store_dwordx2 v2, v[6:7], s[2:3]
	s_add_i32 s0, s0, -1
	s_cmp_lg_u32 s0, 0
	s_cbranch_scc1 .LBB0_1
	s_endpgm
.Lfunc_end0:
"""


def test_live_sets_of_a_loop_with_a_diamond(tmp_path):
    f = tmp_path / "toy.s"
    f.write_text(KERNEL)
    r = subprocess.run([sys.executable, TOOL, str(f), "toy_kernel", "-v"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0].startswith("max live VGPRs: 5 ")                 # v0, v1, v2 (store address) and the pair v[4:5] / v[6:7]
    at = {l.split()[0]: l for l in lines[2:]}
    # loop head: the loop-carried v0, v1 and the never-defined store address v2; v[4:5] are defined in both arms below
    assert " live   3 " in at[".LBB0_1"] and "v[0:2]" in at[".LBB0_1"]
    assert " live   5 " in at[".LBB0_4"] and "v[0:2] v[4:5]" in at[".LBB0_4"]


def test_runs_on_the_built_engine():
    s = os.path.join(ROOT, "phyly_amd", "csrc", "build", "plk_engine-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(s):
        import pytest
        pytest.skip("engine not built with --save-temps in this tree")
    r = subprocess.run([sys.executable, TOOL, s, "_Z15k_up_nodes_mfmaILi4EEv7MUpArgs"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    n = int(r.stdout.split()[3])
    assert 96 <= n <= 168            # three or four vectors of 32 registers plus addressing: the kernel's design point

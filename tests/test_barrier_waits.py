"""Every s_barrier of the product's kernels is reached with no LDS store in flight (tools/check_barrier_waits.py): the defect behind the
round-2..4 concurrency failures was a `__syncthreads()` whose LDS wait hipcc had dropped at a loop header."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import check_barrier_waits as cbw  # noqa: E402

DEFECT = """
k_loop:
	s_load_dword s0, s[0:1], 0x0
	s_waitcnt lgkmcnt(0)
.LBB0_1:
	s_barrier
	ds_read_b64 v[2:3], v1
	s_waitcnt lgkmcnt(0)
	v_add_f64 v[2:3], v[2:3], v[2:3]
	ds_write_b64 v1, v[2:3]
	s_add_i32 s0, s0, -1
	s_cmp_lg_u32 s0, 0
	s_cbranch_scc1 .LBB0_1
	s_endpgm
.Lfunc_end0:
"""


def test_the_checker_sees_a_barrier_behind_a_back_edge_with_pending_lds_stores():
    assert cbw.check_text(DEFECT) == [("k_loop", 1)]
    fixed = DEFECT.replace("\ts_add_i32 s0, s0, -1", "\ts_waitcnt lgkmcnt(0)\n\ts_add_i32 s0, s0, -1")
    assert cbw.check_text(fixed) == [("k_loop", 0)]
    partial = DEFECT.replace("\ts_add_i32 s0, s0, -1", "\ts_waitcnt lgkmcnt(1)\n\ts_add_i32 s0, s0, -1")  # (not a full wait)
    assert cbw.check_text(partial) == [("k_loop", 1)]


def test_no_product_kernel_has_a_barrier_with_lds_stores_in_flight():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_barrier_waits.py")], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 with an unprotected barrier" in r.stdout

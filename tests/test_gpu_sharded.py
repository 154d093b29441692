"""GPU tests of the row-sharded path: two ranks share the one GPU, slices are exchanged through host memory
(gloo), and the result must equal the single-rank run bit for bit; plus the RCCL self-test."""
import pytest

from test_dist_gloo import launch

pytestmark = pytest.mark.gpu


def test_rccl_selftest(qn):
    qn.default_context().comm_selftest()


def test_two_ranks_on_one_gpu_equal_single_rank_bitwise(tmp_path):
    res = launch("gpu", tmp_path, timeout=600)
    assert len(res) == 2
    for r in res:
        assert r["iters"] == 25
        assert r["trace_equal"] and r["x_equal"] and r["h_equal"], r
        assert r["objective_rows_ok"] and r["eval_equal"] and r["dfp_bt_equal"], r
        assert r["lse_eval_close"] and r["lse_dfp_close"], r

"""GPU tests of the symmetric-storage fast path (csrc/qn_sym.hip.h: only the upper block triangle of H and Q is streamed).
It must agree with the fused ROW kernels on the full matrices (tiling -3) and with the oracle to the stated tolerance, be
bit-identical between pipelined and synchronous mode and with / without the deferred update step, leave a complete,
bitwise symmetric H behind (the lower triangle is mirrored back when minimize returns), and step aside for a
non-symmetric user-installed H."""
import numpy as np
import pytest

import problems as P

pytestmark = pytest.mark.gpu


def sym_bytes(n):
    """bytes one pass of the symmetric-storage path streams: off-diagonal 128 x 128 tiles whole, diagonal tiles as their upper
    triangle (wave w of a diagonal tile reads 64 - 8 w lanes of 16 bytes per row: 73 728 of its 131 072 bytes)"""
    nb = n // 128
    return nb * (nb - 1) // 2 * 128 * 128 * 8 + nb * 73728


def _run(qn, method, lsname, obj, x0, iters, tiling=None, sync=None):
    s = (qn.BFGS if method == "bfgs" else qn.DFP)(1e-10, x0)
    s.set_trace(iters, with_x=True)
    if tiling:
        s.configure(*tiling)
    if sync is not None:
        s.set_sync_mode(sync)
    ls = qn.MoreThuente() if lsname == "mt" else qn.BackTracking(1e-4, 0.5)
    st = 0
    try:
        s.minimize(ls, obj, iters, 20)
    except qn.MaxIterReached:
        st = 1
    return s, st


@pytest.mark.parametrize("n", [1024, 1152, 2048])
@pytest.mark.parametrize("method,lsname", [("bfgs", "mt"), ("dfp", "mt"), ("bfgs", "bt")])
def test_symmetric_path_vs_row_kernels_and_oracle(qn, qo, n, method, lsname):
    q, b, x0, diag = P.synth_problem(qo, n)
    obj = qn.Quadratic(q, b)
    iters = 30
    s, st = _run(qn, method, lsname, obj, x0, iters)
    assert s.stats()["matrix_bytes_per_pass"] == sym_bytes(n)  # the path under test did run
    r, st_r = _run(qn, method, lsname, obj, x0, iters, tiling=("symmetric_storage", 0))
    assert r.stats()["matrix_bytes_per_pass"] == n * n * 8
    (tr, xs), (tr_r, xs_r) = s.trace(), r.trace()
    assert st == st_r and len(tr) == len(tr_r)
    assert [(a["n_evals"], a["ls_cases"]) for a in tr] == [(a["n_evals"], a["ls_cases"]) for a in tr_r]
    assert np.allclose([a["t"] for a in tr], [a["t"] for a in tr_r], rtol=1e-9, atol=0)
    assert np.linalg.norm(xs[-1] - xs_r[-1]) <= 1e-9 * max(1.0, np.linalg.norm(xs_r[-1]))
    # the inverse Hessian left behind: complete, bitwise symmetric, equal to the row kernels' to rounding
    h, h_r = s.approx_inv_hessian(), r.approx_inv_hessian()
    assert np.array_equal(h, h.T)
    assert np.abs(h - h_r).max() <= 1e-9 * np.abs(h_r).max()
    # and the oracle (rank-2 restatement), same tolerance as the parity sweep
    ref = qo.Solver(qo.BFGS if method == "bfgs" else qo.DFP, 1e-10, x0, qo.UPDATE_RANK2)
    ls = qo.morethuente() if lsname == "mt" else qo.backtracking(1e-4, 0.5)
    ref.minimize(ls, qo.QuadraticOracle(q, b), iters, 20, trace_cap=iters, trace_x=True)
    k = min(len(tr), len(ref.trace))
    assert k >= 10
    for a, c in zip(tr[:k], ref.trace[:k]):
        assert a["n_evals"] == c["n_evals"] and abs(a["t"] - c["t"]) <= 1e-9 * abs(c["t"])
    assert np.linalg.norm(xs[k - 1] - ref.trace_x[k - 1]) <= 1e-9 * max(1.0, np.linalg.norm(ref.trace_x[k - 1]))


def test_symmetric_path_is_bitwise_reproducible_across_modes(qn, qo):
    n = 1280
    q, b, x0, _ = P.synth_problem(qo, n)
    obj = qn.Quadratic(q, b)
    runs = []
    for tiling, sync in ((None, 0), (None, 1), (("deferred_update_step", 0), 0), (("deferred_update_step", 0), 1)):  # pipelined / synchronous, with / without the deferred step
        s, st = _run(qn, "bfgs", "mt", obj, x0, 35, tiling=tiling, sync=sync)
        tr, xs = s.trace()
        runs.append((st, tr, xs, s.approx_inv_hessian()))
    for other in runs[1:]:
        assert other[0] == runs[0][0] and other[1] == runs[0][1]
        assert np.array_equal(other[2], runs[0][2]) and np.array_equal(other[3], runs[0][3])


@pytest.mark.parametrize("method,lsname", [("bfgs", "mt"), ("dfp", "mt"), ("bfgs", "bt")])
def test_folded_accept_reduce_is_the_same_run_bit_for_bit(qn, qo, method, lsname):
    """set_option("folded_accept_reduce", 1): the launch that runs the update tiles also turns the accepted evaluation's slots into vectors (four
    launches per iteration; s2_hpass_kernel<.., FOLD>).  Same sums in the same order: the trace, the iterates and the inverse
    Hessian must equal the default five-launch pattern bit for bit, pipelined and synchronous, and across a continued call."""
    n = 1280
    q, b, x0, _ = P.synth_problem(qo, n)
    obj = qn.Quadratic(q, b)
    base, st0 = _run(qn, method, lsname, obj, x0, 35)
    tr0, xs0 = base.trace()
    for sync in (0, 1):
        s, st = _run(qn, method, lsname, obj, x0, 35, tiling=("folded_accept_reduce", 1), sync=sync)
        tr, xs = s.trace()
        assert st == st0 and tr == tr0
        assert np.array_equal(xs, xs0) and np.array_equal(s.approx_inv_hessian(), base.approx_inv_hessian())
        assert s.stats()["launches"] < base.stats()["launches"] or sync
    # a run that converges (the speculative tiles of the last accepted point are not followed by a pass) and is then continued
    tol = 1e-3
    pair = []  # (tol 1e-3: converges in a few dozen iterations)
    for tiling in (None, ("folded_accept_reduce", 1)):
        s = (qn.BFGS if method == "bfgs" else qn.DFP)(tol, x0)
        if tiling:
            s.configure(*tiling)
        ls = qn.MoreThuente() if lsname == "mt" else qn.BackTracking(1e-4, 0.5)
        try:
            s.minimize(ls, obj, 200, 20)
        except qn.MaxIterReached:
            pass  # (fixed-ratio backtracking does not get there in 200 iterations: the continued call is still compared)
        k1, x1 = s.k(), s.x()
        s.set_x(x1 + 0.25)  # ... from a new point, on the inverse Hessian the first call left
        try:
            s.minimize(ls, obj, 15, 20)
        except qn.MaxIterReached:
            pass
        pair.append((k1, s.k(), s.x(), s.approx_inv_hessian()))
    assert pair[0][0] == pair[1][0] and pair[0][1] == pair[1][1]
    assert np.array_equal(pair[0][2], pair[1][2]) and np.array_equal(pair[0][3], pair[1][3])


@pytest.mark.parametrize("method,lsname", [("bfgs", "mt"), ("dfp", "mt"), ("bfgs", "bt")])
def test_row_slivers_at_4096(qn, qo, method, lsname):
    """n = 4096: 528 tiles on 256 workgroups.  The sixteen diagonal tiles left over after two rounds are cut into 8-row slivers, one
    per workgroup (qn_sym2.hip.h, qn_s2_eval_sliver); set_option("row_slivers", 0) switches back to whole tiles (round 2's work lists).  Other
    association of the sums, same run: decisions equal, steps and iterates to the parity tolerance, H complete and bitwise
    symmetric; pipelined and synchronous identical; and a solver may change between the two layouts from call to call."""
    n = 4096
    q, b, x0, _ = P.synth_problem(qo, n)
    obj = qn.Quadratic(q, b)
    iters = 24
    s, st = _run(qn, method, lsname, obj, x0, iters)
    r, st_r = _run(qn, method, lsname, obj, x0, iters, tiling=("row_slivers", 0))
    assert s.stats()["path"] & 16 and r.stats()["path"] & 16
    (tr, xs), (tr_r, xs_r) = s.trace(), r.trace()
    assert st == st_r and len(tr) == len(tr_r) >= 10
    assert [(a["n_evals"], a["ls_cases"]) for a in tr] == [(a["n_evals"], a["ls_cases"]) for a in tr_r]
    assert np.allclose([a["t"] for a in tr], [a["t"] for a in tr_r], rtol=1e-9, atol=0)
    assert np.allclose([a["f"] for a in tr], [a["f"] for a in tr_r], rtol=1e-9, atol=0)
    assert np.linalg.norm(xs[-1] - xs_r[-1]) <= 1e-9 * max(1.0, np.linalg.norm(xs_r[-1]))
    h, h_r = s.approx_inv_hessian(), r.approx_inv_hessian()
    assert np.array_equal(h, h.T) and np.array_equal(h_r, h_r.T)
    assert np.abs(h - h_r).max() <= 1e-9 * np.abs(h_r).max()
    sy, st_y = _run(qn, method, lsname, obj, x0, iters, sync=1)
    assert st_y == st and sy.trace()[0] == tr and np.array_equal(sy.approx_inv_hessian(), h)
    # the evaluation kernel's two-items-and-a-sliver instance (s2_eval_kernel<true>) against the general body on the same lists:
    # the same sums in the same order
    g, st_g = _run(qn, method, lsname, obj, x0, iters, tiling=("eval_pair_instance", 0))
    assert st_g == st and g.trace()[0] == tr and np.array_equal(g.approx_inv_hessian(), h)
    # against the oracle, as the parity sweep does
    ref = qo.Solver(qo.BFGS if method == "bfgs" else qo.DFP, 1e-10, x0, qo.UPDATE_RANK2)
    ls = qo.morethuente() if lsname == "mt" else qo.backtracking(1e-4, 0.5)
    ref.minimize(ls, qo.QuadraticOracle(q, b), iters, 20, trace_cap=iters, trace_x=True)
    k = min(len(tr), len(ref.trace))
    for a, c in zip(tr[:k], ref.trace[:k]):
        assert a["n_evals"] == c["n_evals"] and abs(a["t"] - c["t"]) <= 1e-9 * abs(c["t"])
    assert np.linalg.norm(xs[k - 1] - ref.trace_x[k - 1]) <= 1e-9 * max(1.0, np.linalg.norm(ref.trace_x[k - 1]))
    # layouts alternate on one solver: slivers -> whole tiles -> slivers, 8 iterations each, against the run above
    m = (qn.BFGS if method == "bfgs" else qn.DFP)(1e-10, x0)
    ls = qn.MoreThuente() if lsname == "mt" else qn.BackTracking(1e-4, 0.5)
    for leg in range(3):
        try:
            m.minimize(ls, obj, 8, 20)
        except qn.MaxIterReached:
            pass
        m.set_option("row_slivers", 0)
    assert np.linalg.norm(m.x() - xs[-1]) <= 1e-9 * max(1.0, np.linalg.norm(xs[-1]))
    hm = m.approx_inv_hessian()
    assert np.array_equal(hm, hm.T) and np.abs(hm - h).max() <= 1e-9 * np.abs(h).max()


@pytest.mark.parametrize("n,method,lsname", [(4096, "bfgs", "mt"), (4096, "dfp", "mt"), (4096, "bfgs", "bt"), (1152, "bfgs", "mt"),
                                             (3200, "dfp", "mt"), (8192, "bfgs", "mt")])
def test_tail_reduce_is_the_reduce_launch_bit_for_bit(qn, qo, n, method, lsname):
    """Round 5, set_option("tail_reduce", 1): the update-reduce in the TAIL of the update-tile launch (s2_hpass_kernel<.., TRED>: the workgroup
    whose slot completes a block-row sums that block-row's slots, in slot order) -- 4 launches per iteration instead of 5.  Measured
    slower than the launch it removes and off by default (note in front of the kernel); kept as a tested variant.  The arrival
    order decides who sums, never in what order: trace, iterates and inverse Hessian must equal the default run bit for bit,
    pipelined and synchronous, with row slivers (n = 4096), without (1152, 3200: lists of one and two items) and with lists of
    nine items (8192); repeated, because a stale slot would depend on timing."""
    diag = P.synth_diag(n); b, x0 = P.synth_vectors(n)
    obj = qn.Quadratic.synthetic(n, P.SEED, diag, b)
    iters = 24 if n <= 4096 else 8
    base, st0 = _run(qn, method, lsname, obj, x0, iters)
    tr0, xs0 = base.trace()
    h0 = base.approx_inv_hessian() if n <= 4096 else None
    assert base.stats()["path"] & 16
    for rep, sync in enumerate((0, 1, 0, 0)):
        s, st = _run(qn, method, lsname, obj, x0, iters, tiling=("tail_reduce", 1), sync=sync)
        tr, xs = s.trace()
        assert st == st0 and tr == tr0 and np.array_equal(xs, xs0), (rep, sync)
        if h0 is not None:
            assert np.array_equal(s.approx_inv_hessian(), h0)
        if not sync:
            assert s.stats()["launches"] < base.stats()["launches"]  # one launch less per iteration


@pytest.mark.parametrize("method,lsname", [("bfgs", "mt"), ("dfp", "mt"), ("bfgs", "bt")])
def test_ring_evaluation_is_the_pair_instance_bit_for_bit(qn, qo, method, lsname):
    """Round 6: at n = 4096 the evaluation tiles run as s2_evalr_kernel (csrc/qn_sym2r.hip.h) -- a 16-wave workgroup whose eight
    mover waves stream the two tiles and the sliver as round 5's kernel does, while eight multiplier waves take the FIRST tile's rows out of the LDS
    park as they land (the trial point staged once per workgroup) and the movers multiply the second tile out of their own registers; the lanes'
    shares of the scalar sums change hands through LDS.  The same products, sums and exchanges in the same order as round 5's two-items-and-a-
    sliver instance: trace, iterates and inverse Hessian must be equal bit for bit, pipelined and synchronous; repeated, because a row consumed
    before it was parked would depend on timing.  14.3 us per launch against 15.3 (profiles/r06_a_*): the default since round 6; set_option("eval_mover_multiplier", 0)
    selects round 5's kernel.  Backtracking evaluates at points it rejects: launches whose request is not an evaluation's leave early."""
    n = 4096
    diag = P.synth_diag(n); b, x0 = P.synth_vectors(n)
    obj = qn.Quadratic.synthetic(n, P.SEED, diag, b)
    iters = 30
    base, st0 = _run(qn, method, lsname, obj, x0, iters, tiling=("eval_mover_multiplier", 0))
    tr0, xs0 = base.trace()
    h0 = base.approx_inv_hessian()
    assert base.stats()["path"] & 16 and len(tr0) == iters
    assert all(np.isfinite(r["f"]) for r in tr0)
    for rep, sync in enumerate((0, 1, 0, 0, 1, 0)):
        s, st = _run(qn, method, lsname, obj, x0, iters, sync=sync)
        tr, xs = s.trace()
        assert st == st0 and tr == tr0 and np.array_equal(xs, xs0), (rep, sync)
        assert np.array_equal(s.approx_inv_hessian(), h0)


@pytest.mark.parametrize("method,lsname", [("bfgs", "mt"), ("bfgs", "bt")])
def test_zigzag_and_touch_workgroups_change_no_bit(qn, qo, method, lsname):
    """Round 6, the XCDs' L2 across kernel boundaries (tools/l2_keep_probe*.hip, profiles/r06_e_*): (1) ZIG-ZAG -- s2_evalr_kernel streams its two
    tiles in the other order in launches of odd parity, so that an evaluation launch right behind another one starts with the tile the L2 still
    holds; the items change hands between the multipliers and the movers, every sum keeps its operands and its order.  (2) TOUCH WORKGROUPS -- the
    accept-reduce and the update-reduce launches carry G workgroups that only LOAD rows of the tile the next tile launch's workgroup of the same
    index streams first.  Both are cache policy: trace, iterates and inverse Hessian equal bit for bit with each switched off, with both off, and
    with whole tiles touched; pipelined and synchronous; the launches per iteration do not change."""
    n = 4096
    diag = P.synth_diag(n); b, x0 = P.synth_vectors(n)
    obj = qn.Quadratic.synthetic(n, P.SEED, diag, b)
    iters = 24
    def run(opts, sync=0):
        s = (qn.BFGS if method == "bfgs" else qn.DFP)(1e-10, x0)
        s.set_trace(iters, with_x=True)
        for k, v in opts.items():
            s.set_option(k, v)
        s.set_sync_mode(sync)
        ls = qn.MoreThuente() if lsname == "mt" else qn.BackTracking(1e-4, 0.5)
        st = 0
        try:
            s.minimize(ls, obj, iters, 20)
        except qn.MaxIterReached:
            st = 1
        return s, st
    base, st0 = run({"eval_zigzag": 0, "touch_h_rows": 0, "touch_q_rows": 0})
    tr0, xs0 = base.trace()
    h0 = base.approx_inv_hessian()
    l0 = base.stats()["launches"]
    assert base.stats()["path"] & 16 and len(tr0) == iters and all(np.isfinite(r["f"]) for r in tr0)
    for opts in ({}, {"touch_h_rows": 0, "touch_q_rows": 0}, {"eval_zigzag": 0}, {"touch_h_rows": 16, "touch_q_rows": 16}, {"touch_h_rows": 4, "touch_q_rows": 12},
                 {"eval_zigzag": 0, "touch_h_rows": 10, "touch_q_rows": 0}):
        for sync in (0, 1):
            s, st = run(opts, sync)
            tr, xs = s.trace()
            assert st == st0 and tr == tr0 and np.array_equal(xs, xs0), (opts, sync)
            assert np.array_equal(s.approx_inv_hessian(), h0), (opts, sync)
            if sync == 0:
                assert s.stats()["launches"] == l0, (opts, s.stats()["launches"], l0)
    with pytest.raises(qn.ErrorInputParams):
        base.set_option("touch_h_rows", 5)


@pytest.mark.parametrize("n", [1152, 3200])
def test_second_generation_without_slivers_and_without_the_pair_instance(qn, qo, n):
    """Sizes whose work lists carry NO row slivers (sl_per == 0) and where the two-items-and-a-sliver instance of the evaluation
    kernel cannot run (pair == 0): n = 1152 (nb = 9: 45 items, one per workgroup) and n = 3200 (nb = 25: 325 items on 256
    workgroups -- lists of one and of two items side by side).  Round 3's memory fault lived exactly here: the kernels divided by
    the sliver count unconditionally and the compiler deleted the guarded no-sliver path (qn_s2_sliver).  The host now also refuses
    work lists whose sliver count does not tile the grid (solver_alloc_sym2).  Against the row kernels, pipelined = synchronous."""
    q, b, x0, _ = P.synth_problem(qo, n)
    obj = qn.Quadratic(q, b)
    iters = 16
    s, st = _run(qn, "bfgs", "mt", obj, x0, iters)
    assert s.stats()["path"] & 16 and s.stats()["matrix_bytes_per_pass"] == sym_bytes(n)
    r, st_r = _run(qn, "bfgs", "mt", obj, x0, iters, tiling=("symmetric_storage", 0))
    (tr, xs), (tr_r, xs_r) = s.trace(), r.trace()
    assert st == st_r and len(tr) == len(tr_r) == iters
    assert [(a["n_evals"], a["ls_cases"]) for a in tr] == [(a["n_evals"], a["ls_cases"]) for a in tr_r]
    assert np.linalg.norm(xs[-1] - xs_r[-1]) <= 1e-9 * max(1.0, np.linalg.norm(xs_r[-1]))
    y, st_y = _run(qn, "bfgs", "mt", obj, x0, iters, sync=1)
    assert st_y == st and y.trace()[0] == tr and np.array_equal(y.trace()[1], xs)
    h = s.approx_inv_hessian()
    assert np.array_equal(h, h.T) and np.array_equal(y.approx_inv_hessian(), h)


def test_warm_restart_continues_on_the_mirrored_hessian(qn, qo):
    """Two minimize calls of 10 iterations follow one of 20: the lower triangle restored between the calls is the right one
    (the second call's first pass applies the pending update to it)."""
    n = 1024
    q, b, x0, _ = P.synth_problem(qo, n)
    obj = qn.Quadratic(q, b)
    one, _ = _run(qn, "bfgs", "mt", obj, x0, 20)
    two = qn.BFGS(1e-10, x0)
    for _ in range(2):
        with pytest.raises(qn.MaxIterReached):
            two.minimize(qn.MoreThuente(), obj, 10, 20)
    # (to rounding, not bit for bit: a fresh call forms its first direction with a pass over H, a running one with the lazy formula)
    assert np.linalg.norm(one.x() - two.x()) <= 1e-9 * np.linalg.norm(one.x())
    h1, h2 = one.approx_inv_hessian(), two.approx_inv_hessian()
    assert np.array_equal(h2, h2.T) and np.abs(h1 - h2).max() <= 1e-9 * np.abs(h1).max()
    # ... and the generic path can take over from the mirrored matrix (it reads every entry)
    three = qn.BFGS(1e-10, x0)
    with pytest.raises(qn.MaxIterReached):
        three.minimize(qn.MoreThuente(), obj, 10, 20)
    three.set_option("generic_kernels", 1)
    with pytest.raises(qn.MaxIterReached):
        three.minimize(qn.MoreThuente(), obj, 10, 20)
    assert np.linalg.norm(three.x() - one.x()) <= 1e-9 * np.linalg.norm(one.x())


@pytest.mark.parametrize("tiling", [None, ("second_generation", 0), ("symmetric_storage", 0)])
def test_continued_calls_are_one_run_bit_for_bit(qn, qo, tiling):
    """minimize resets only k (ls_solver.rs:74): a second call on the same device objective, with nothing touched in between,
    continues the first one -- the memoised evaluation at x_k and the lazily formed direction carry over, so two calls of 10
    iterations ARE one call of 20, bit for bit, and cost no extra evaluation or pass over H.  Touching the state (set_x) or
    changing the objective makes the next call start from scratch: same iterates to rounding, one more evaluation."""
    n = 1024
    q, b, x0, _ = P.synth_problem(qo, n)
    obj = qn.Quadratic(q, b)
    one, _ = _run(qn, "bfgs", "mt", obj, x0, 20, tiling=tiling)
    two = qn.BFGS(1e-10, x0)
    if tiling:
        two.configure(*tiling)
    evals = passes = 0
    for _ in range(2):
        with pytest.raises(qn.MaxIterReached):
            two.minimize(qn.MoreThuente(), obj, 10, 20)
        evals += two.stats()["oracle_evals"]
        passes += two.stats()["h_passes"]
    assert np.array_equal(one.x(), two.x())
    assert (evals, passes) == (one.stats()["oracle_evals"], one.stats()["h_passes"])
    assert np.array_equal(one.approx_inv_hessian(), two.approx_inv_hessian())
    three = qn.BFGS(1e-10, x0)
    if tiling:
        three.configure(*tiling)
    with pytest.raises(qn.MaxIterReached):
        three.minimize(qn.MoreThuente(), obj, 10, 20)
    three.set_x(three.x())  # same values, but the solver cannot know that
    with pytest.raises(qn.MaxIterReached):
        three.minimize(qn.MoreThuente(), obj, 10, 20)
    assert three.stats()["oracle_evals"] == two.stats()["oracle_evals"] + 1 and three.stats()["h_passes"] == two.stats()["h_passes"] + 1
    assert np.linalg.norm(three.x() - one.x()) <= 1e-9 * np.linalg.norm(one.x())
    four = qn.BFGS(1e-10, x0)
    if tiling:
        four.configure(*tiling)
    with pytest.raises(qn.MaxIterReached):
        four.minimize(qn.MoreThuente(), obj, 10, 20)
    obj2 = qn.Quadratic(q, b)  # an equal objective is still another oracle
    with pytest.raises(qn.MaxIterReached):
        four.minimize(qn.MoreThuente(), obj2, 10, 20)
    assert four.stats()["oracle_evals"] == two.stats()["oracle_evals"] + 1
    assert np.linalg.norm(four.x() - one.x()) <= 1e-9 * np.linalg.norm(one.x())


def test_objective_destroyed_and_recreated_between_calls_is_not_a_continuation(qn, qo):
    """The warm continuation must know the objective by identity, not by address: run to the iteration cap on A, destroy A,
    create B (same size: the allocator is likely to hand the address back) with another Q and b, continue.  The second call has
    to evaluate B at x_k and form the direction from scratch -- as the reference does on every minimize() (ls_solver.rs:79) --
    and so equal a solver that was handed (x_k, H_k) and ran on B cold."""
    n = 1024
    q, b, x0, _ = P.synth_problem(qo, n)
    rng = np.random.default_rng(11)
    q2 = q + np.diag(rng.uniform(0.5, 3.0, n))
    b2 = b + rng.standard_normal(n)
    for attempt in range(4):  # several tries: each destroy / create is a chance for the address to come back
        a_obj = qn.Quadratic(q, b)
        s = qn.BFGS(1e-10, x0)
        with pytest.raises(qn.MaxIterReached):
            s.minimize(qn.MoreThuente(), a_obj, 8, 20)
        xk, hk = s.x(), s.approx_inv_hessian()
        # (the getters above export nothing that would end a continuation on the fast path: re-run to the cap to arm it again)
        s2 = qn.BFGS(1e-10, x0)
        with pytest.raises(qn.MaxIterReached):
            s2.minimize(qn.MoreThuente(), a_obj, 8, 20)
        a_obj.close()
        b_obj = qn.Quadratic(q2, b2)
        with pytest.raises(qn.MaxIterReached):
            s2.minimize(qn.MoreThuente(), b_obj, 6, 20)
        cold = qn.BFGS(1e-10, xk)
        cold.set_approx_inv_hessian(hk)
        with pytest.raises(qn.MaxIterReached):
            cold.minimize(qn.MoreThuente(), b_obj, 6, 20)
        assert s2.stats()["oracle_evals"] == cold.stats()["oracle_evals"]  # the evaluation at x_k on B was made
        assert np.linalg.norm(s2.x() - cold.x()) <= 1e-9 * max(1.0, np.linalg.norm(cold.x()))
        b_obj.close()


def test_non_symmetric_user_hessian_uses_the_full_matrix(qn, qo):
    n = 1024
    q, b, x0, _ = P.synth_problem(qo, n)
    obj = qn.Quadratic(q, b)
    rng = np.random.default_rng(11)
    h0 = np.eye(n) + 1e-3 * rng.standard_normal((n, n))  # not symmetric: the reference would use it as it is
    s = qn.BFGS(1e-10, x0)
    s.set_approx_inv_hessian(h0)
    with pytest.raises(qn.MaxIterReached):
        s.minimize(qn.MoreThuente(), obj, 5, 20)
    assert s.stats()["matrix_bytes_per_pass"] == n * n * 8
    ref = qo.Solver(qo.BFGS, 1e-10, x0, qo.UPDATE_RANK2)
    ref.set_inv_hessian(h0)
    ref.minimize(qo.morethuente(), qo.QuadraticOracle(q, b), 5, 20)
    assert np.linalg.norm(s.x() - ref.x) <= 1e-9 * np.linalg.norm(ref.x)
    # a symmetric one re-enables the tiles
    s2 = qn.BFGS(1e-10, x0)
    s2.set_approx_inv_hessian(0.5 * (h0 + h0.T))
    with pytest.raises(qn.MaxIterReached):
        s2.minimize(qn.MoreThuente(), obj, 5, 20)
    assert s2.stats()["matrix_bytes_per_pass"] < n * n * 8


# ---------------------------------------------------------------------------------------------
# the generic path's H pass on the same tiles (closures, log-sum-exp objective, SR1, bounded variants)
# ---------------------------------------------------------------------------------------------

def _generic_pair(make_solver, minimize):
    out = []
    for full in (False, True):
        s = make_solver()
        if full:
            s.set_option("symmetric_storage", 0)  # H pass on the full row-major matrix
        s.set_trace(40, with_x=True)
        st = 0
        try:
            minimize(s)
        except Exception as e:  # noqa: BLE001
            st = type(e).__name__
        out.append((s, st))
    return out


def _assert_same_run(a, b, n, half_expected=True):
    (s, st), (r, st_r) = a, b
    assert st == st_r
    if half_expected:  # (the generic path's H pass runs on whole 128 x 128 tiles, diagonal ones included)
        assert s.stats()["matrix_bytes_per_pass"] == (n // 128) * (n // 128 + 1) // 2 * 128 * 128 * 8
    assert r.stats()["matrix_bytes_per_pass"] == n * n * 8
    (tr, xs), (tr_r, xs_r) = s.trace(), r.trace()
    assert len(tr) == len(tr_r) and len(tr) >= 5
    assert [x["n_evals"] for x in tr] == [x["n_evals"] for x in tr_r]
    assert np.linalg.norm(xs[-1] - xs_r[-1]) <= 1e-9 * max(1.0, np.linalg.norm(xs_r[-1]))
    h, h_r = s.approx_inv_hessian(), r.approx_inv_hessian()
    assert np.array_equal(h, h.T) and np.abs(h - h_r).max() <= 1e-9 * np.abs(h_r).max()


@pytest.mark.parametrize("method", ["bfgs", "dfp"])
def test_generic_path_device_objective_on_tiles(qn, qo, method):
    n = 1024
    q, b, x0, _ = P.synth_problem(qo, n)
    obj = qn.Quadratic(q, b)

    def make():
        s = (qn.BFGS if method == "bfgs" else qn.DFP)(1e-10, x0)
        s.set_option("generic_kernels", 1)  # generic kernels (the control step does the vector work)
        return s
    a, b_ = _generic_pair(make, lambda s: s.minimize(qn.MoreThuente(), obj, 25, 20))
    _assert_same_run(a, b_, n)


def test_generic_path_host_closure_on_tiles(qn, qo):
    n = 1024
    q, b, x0, _ = P.synth_problem(qo, n)
    fn = lambda x: (0.5 * x @ (q @ x) - b @ x, q @ x - b)  # noqa: E731
    a, b_ = _generic_pair(lambda: qn.BFGS(1e-10, x0), lambda s: s.minimize(qn.MoreThuente(), fn, 12, 20))
    _assert_same_run(a, b_, n)
    ref = qo.Solver(qo.BFGS, 1e-10, x0, qo.UPDATE_RANK2)
    ref.minimize(qo.morethuente(), qo.QuadraticOracle(q, b), 12, 20)
    assert np.linalg.norm(a[0].x() - ref.x) <= 1e-8 * np.linalg.norm(ref.x)


def test_generic_path_logsumexp_dfp_on_tiles(qn, qo):
    rng = np.random.default_rng(21)
    m = n = 1024
    amat = rng.standard_normal((m, n)) * (3.0 / np.sqrt(n))
    c = rng.standard_normal(m)
    x0 = rng.standard_normal(n)
    obj = qn.LogSumExp(amat, c, 0.1)

    def generic_dfp():  # (round 5: the default for this objective is the second-generation structure, qn_sym2g.hip.h; -4 keeps the generic path)
        s = qn.DFP(1e-10, x0)
        s.set_option("second_generation", 0)
        return s
    a, b_ = _generic_pair(generic_dfp, lambda s: s.minimize(qn.MoreThuente(), obj, 15, 20))
    assert a[0].stats()["path"] & 4 and not a[0].stats()["path"] & 16  # generic path, H pass on the symmetric tiles
    _assert_same_run(a, b_, n)
    # ... and the default path against both
    d = qn.DFP(1e-10, x0)
    d.set_trace(40, with_x=True)
    with pytest.raises(qn.MaxIterReached):
        d.minimize(qn.MoreThuente(), obj, 15, 20)
    assert d.stats()["path"] & 16 and d.stats()["path"] & 8  # second-generation structure, pipelined
    _assert_same_run((d, "MaxIterReached"), b_, n)


def test_generic_path_sr1b_bounded_on_tiles(qn, qo):
    n = 1024
    q, b, x0, _ = P.synth_problem(qo, n)
    rng = np.random.default_rng(5)
    xs = np.linalg.solve(q, b)
    lb = np.where(rng.random(n) < 0.3, xs + 0.05, -np.inf)
    ub = np.where(rng.random(n) < 0.3, np.maximum(xs - 0.05, lb + 0.1), np.inf)
    obj = qn.Quadratic(q, b)
    def mk():  # (SR1B of this shape takes the second-generation path since round 5: this test is about the GENERIC path's tiles)
        s = qn.SR1B.new(1e-10, x0, lb, ub)
        s.set_option("bounded_second_generation", 0)
        return s
    a, b_ = _generic_pair(mk, lambda s: s.minimize(qn.MoreThuente(), obj, 20, 20))
    _assert_same_run(a, b_, n)
    x = a[0].x()
    assert np.all(x >= lb - 1e-12) and np.all(x <= ub + 1e-12)


def test_callback_sees_a_complete_hessian_mid_run(qn, qo):
    """ls_solver.rs:105-107: the callback gets `&Self`; approx_inv_hessian() called from it must be the whole matrix although
    the run itself only maintains the upper block triangle."""
    n = 1024
    q, b, x0, _ = P.synth_problem(qo, n)
    obj = qn.Quadratic(q, b)
    seen = []

    def cb(solver):
        h = solver.approx_inv_hessian()
        seen.append((solver.k(), bool(np.array_equal(h, h.T)), h))

    s = qn.BFGS(1e-10, x0)
    with pytest.raises(qn.MaxIterReached):
        s.minimize(qn.MoreThuente(), obj, 6, 20, cb)
    assert [k for k, _, _ in seen] == [1, 2, 3, 4, 5, 6] and all(sym for _, sym, _ in seen)
    ref = qo.Solver(qo.BFGS, 1e-10, x0, qo.UPDATE_RANK2)
    hs = []
    ref.minimize(qo.morethuente(), qo.QuadraticOracle(q, b), 6, 20, callback=lambda r: hs.append(r.approx_inv_hessian))
    for (_, _, h), h_ref in zip(seen, hs):
        assert np.abs(h - h_ref).max() <= 1e-8 * np.abs(h_ref).max()


def test_non_symmetric_objective_matrix_is_multiplied_as_given(qn, qo):
    """A caller may hand qn_quadratic_create a Q that is not symmetric (g = Qx - b is then not the gradient of f, but that is the
    caller's business): the evaluation must multiply by the matrix as given, i.e. stay on the row kernels."""
    n = 1024
    q, b, x0, _ = P.synth_problem(qo, n)
    rng = np.random.default_rng(2)
    qa = q + 1e-3 * np.triu(rng.standard_normal((n, n)), 1)  # upper triangle perturbed only
    obj = qn.Quadratic(qa, b)
    ev = obj(x0)
    assert np.allclose(ev.g(), qa @ x0 - b, rtol=1e-12, atol=1e-12)
    s = qn.BFGS(1e-10, x0)
    try:
        s.minimize(qn.MoreThuente(), obj, 3, 20)
    except qn.SolverError:
        pass
    assert s.stats()["matrix_bytes_per_pass"] == n * n * 8

"""CPU test ("additional pins the build must create itself"): the oracle's f64 trajectories against an independent
HIGH-PRECISION evaluation of the same recurrences (mpmath, 60 significant digits), written from the mathematics of the reference
path rather than from its operation order -- matrix form of the BFGS / DFP updates (bfgs.rs:112-127, dfp.rs:110-116), the
line-search decisions of morethuente.rs:165-297 and backtracking.rs:20-58 on phi(t) = (f(x + t d), g(x + t d).d).

What it pins: the f64 restatement (and with it the HIP path, which is compared with the restatement) computes the algorithm the
reference's text describes, to the stated parity tolerance, over the pre-convergence window -- same More-Thuente case per inner
iteration, same number of evaluations, |t - t*| <= 1e-9 |t*|, ||x - x*|| <= 1e-9 max(1, ||x*||), |f - f*| <= 1e-10 max(1, |f*|).
It is not the reference (which cannot run here): parity stays unpinned at the iterate level, this narrows what could be wrong to
a misreading that the C restatement, the pure-Python restatement and this one all share."""
import mpmath as mp
import numpy as np
import pytest

import mt_workloads as W
import problems as P

mp.mp.dps = 60


def _mpv(a):
    return mp.matrix([mp.mpf(float(v)) for v in a])


def _mpm(a):
    return mp.matrix([[mp.mpf(float(v)) for v in row] for row in a])


class _Quadratic:
    """f = 1/2 x'Qx - b'x in high precision, from the f64 entries of the test problem (exactly representable)"""

    def __init__(self, q, b):
        self.q, self.b, self.calls = _mpm(q), _mpv(b), 0

    def __call__(self, x):
        self.calls += 1
        qx = self.q * x
        return (x.T * qx)[0] / 2 - (self.b.T * x)[0], qx - self.b


def _dot(a, b):
    return (a.T * b)[0]


def _more_thuente(phi, f0, gd0, max_iter, t_max=mp.inf, c1=mp.mpf("1e-4"), c2=mp.mpf("0.9"), t_min=mp.mpf(0),
                  delta=mp.mpf("0.66")):
    """morethuente.rs:165-297; returns (t, case digits, evaluations in the reference's sequence)"""
    def cubic(ta, tb, f_ta, f_tb, g_ta, g_tb):  # :93-108
        s = 3 * (f_tb - f_ta) / (tb - ta)
        z = s - g_ta - g_tb
        r = z * z - g_ta * g_tb
        if r < 0:
            return None  # the reference's sqrt gives NaN here
        w = mp.sqrt(r)
        return ta + (tb - ta) * ((w - g_ta - z) / (g_tb - g_ta + 2 * w))

    def quad1(ta, tb, f_ta, f_tb, g_ta):  # :110-121
        lin = (f_ta - f_tb) / (ta - tb)
        return ta - ((ta - tb) * g_ta) / (g_ta - lin) / 2

    def quad2(ta, tb, g_ta, g_tb):  # :123-132
        return ta - g_ta * ((ta - tb) / (g_ta - g_tb))

    digits, evals = [], 0
    use_mod, conv = False, False
    t = min(max(mp.mpf(1), t_min), t_max)
    tl, tu = t_min, t_max
    for _ in range(max_iter):
        f_t, gd_t = phi(t)
        evals += 1
        if f_t - f0 <= c1 * t * gd0 and abs(gd_t) <= c2 * abs(gd0):  # strong Wolfe (mod.rs:72-83)
            return t, digits, evals
        if conv or t == tl or t == tu:
            return t, digits, evals
        psi_f, psi_g = f_t - f0 - c1 * t * gd0, gd_t - c1 * gd0
        if not use_mod and psi_f <= 0 and gd_t > 0:
            use_mod = True
        f_tl_raw, gd_tl_raw = phi(tl)  # :217 re-evaluation at tl
        evals += 1
        if use_mod:
            f_tl, g_tl, ft, gt = f_tl_raw, gd_tl_raw, f_t, gd_t
        else:
            f_tl, g_tl, ft, gt = f_tl_raw - f0 - c1 * tl * gd0, gd_tl_raw - c1 * gd0, psi_f, psi_g
        if ft > f_tl:
            digits.append(1)
            tc, tq = cubic(tl, t, f_tl, ft, g_tl, gt), quad1(tl, t, f_tl, ft, g_tl)
            t_new = tc if abs(tc - tl) < abs(tq - tl) else (tq + tc) / 2
        elif gt * g_tl < 0:
            digits.append(2)
            tc, ts = cubic(tl, t, f_tl, ft, g_tl, gt), quad2(tl, t, g_tl, gt)
            t_new = tc if abs(tc - t) >= abs(ts - t) else ts
        elif abs(gt) <= abs(g_tl):
            digits.append(3)
            tc, ts = cubic(tl, t, f_tl, ft, g_tl, gt), quad2(tl, t, g_tl, gt)
            t_plus = tc if abs(tc - t) < abs(ts - t) else ts
            bound = t + delta * (tu - t)
            t_new = min(t_plus, bound) if t > tl else max(t_plus, bound)
        else:
            digits.append(4)
            if tu == mp.inf:
                return None, digits, evals + 1  # the reference evaluates at x + inf d here: outside real arithmetic
            f_tu_raw, gd_tu_raw = phi(tu)
            evals += 1
            f_tu, g_tu = (f_tu_raw, gd_tu_raw) if use_mod else (f_tu_raw - f0 - c1 * tu * gd0, gd_tu_raw - c1 * gd0)
            t_new = cubic(tu, t, ft, f_tu, gt, g_tu)
        if t_new is None:
            assert digits[-1] == 4, "a NaN minimiser outside case 4 is not modelled here"
            t_new = min(t_min, t_max)  # :290 `t.max(t_min).min(t_max)`: Rust's max / min drop the NaN operand
        t_new = min(max(t_new, t_min), t_max)
        # update_interval (:64-91) as the reference calls it (:293): the NEW t with the OLD trial's f_t, g_t -- reproduced, not
        # "fixed" (the textbook updates the bracket with the trial that was evaluated)
        t = t_new
        if ft > f_tl:
            tu = t
        elif gt * (tl - t) > 0:
            tl = t
        elif gt * (tl - t) < 0:
            tl, tu = t, tl
        else:
            conv = True
    return t, digits, evals


def _backtracking(phi, f0, gd0, max_iter, c1=mp.mpf("1e-4"), beta=mp.mpf("0.5")):
    t, evals = mp.mpf(1), 0
    for _ in range(max_iter):
        f_t, _ = phi(t)
        evals += 1
        if f_t - f0 <= c1 * t * gd0:
            return t, [], evals
        t *= beta
    return t, [], evals


def _run_mp(method, lsname, q, b, x0, iters, tol=1e-10, h0=None, t_max=mp.inf):
    n = len(x0)
    obj = _Quadratic(q, b)
    x = _mpv(x0)
    h = mp.eye(n) * (mp.mpf(h0) if h0 is not None else 1)
    eye = mp.eye(n)
    s_norm = y_norm = None
    out = []
    for _ in range(iters):
        f, g = obj(x)
        if (s_norm is not None and s_norm < tol) or (y_norm is not None and y_norm < tol) or mp.norm(g) < tol:
            break
        d = -(h * g)
        gd0 = _dot(g, d)

        def phi(t, x=x, d=d):
            ft, gt = obj(x + t * d)
            return ft, _dot(gt, d)
        if lsname == "mt":
            t, digits, evals = _more_thuente(phi, f, gd0, 20, t_max=t_max)
        else:
            t, digits, evals = _backtracking(phi, f, gd0, 20)
        if t is None:
            out.append(dict(t=None, digits=digits, n_evals=None, x=None, f=f))
            break
        xn = x + t * d
        s = xn - x
        _, gn = obj(xn)  # bfgs.rs:98
        y = gn - g
        s_norm, y_norm = mp.norm(s), mp.norm(y)
        out.append(dict(t=t, digits=digits, n_evals=evals + 2, x=xn, f=f, gnorm=mp.norm(g)))
        x = xn
        if s_norm < tol or y_norm < tol:
            continue
        ys = _dot(y, s)
        if method == "bfgs":  # (I - rho s y') H (I - rho y s') + rho s s'
            rho = 1 / ys
            h = (eye - rho * (s * y.T)) * h * (eye - rho * (y * s.T)) + rho * (s * s.T)
        else:  # H + s s'/(s'y) - H y y' H/(y' H y)
            hy = h * y
            h = h + (s * s.T) / ys - (hy * hy.T) / _dot(y, hy)
    return out


def _compare(ref_trace, ref_xs, mp_trace, label, t_tol=1e-9):
    g0 = ref_trace[0]["gnorm"]
    checked = 0
    for k, (r, m) in enumerate(zip(ref_trace, mp_trace)):
        if r["gnorm"] < 1e-6 * g0 or k >= 50:
            break  # outside the pre-convergence window the trajectories are a conditioning statement, not a parity one
        assert W.case_digits(r["ls_cases"]) == m["digits"], (label, k, oct(r["ls_cases"]), m["digits"])
        if m["t"] is None:
            checked += 1
            break  # case 4 at tu = +inf: decisions agree up to here, the rest is IEEE non-finite arithmetic
        assert r["n_evals"] == m["n_evals"], (label, k, r["n_evals"], m["n_evals"])
        assert abs(mp.mpf(r["t"]) - m["t"]) <= t_tol * abs(m["t"]), (label, k, r["t"], mp.nstr(m["t"], 17))
        assert abs(mp.mpf(r["f"]) - m["f"]) <= 1e-10 * max(1, abs(m["f"])), (label, k)
        dx = mp.norm(_mpv(ref_xs[k]) - m["x"])
        assert dx <= 1e-9 * max(1, mp.norm(m["x"])), (label, k, mp.nstr(dx, 5))
        checked += 1
    return checked


@pytest.mark.parametrize("n", [2, 3, 8, 17, 64])
@pytest.mark.parametrize("method", ["bfgs", "dfp"])
@pytest.mark.parametrize("lsname", ["mt", "bt"])
def test_oracle_trajectory_vs_high_precision_recurrences(qo, n, method, lsname):
    q, b, x0, _ = P.synth_problem(qo, n, kappa=1e3 if n > 3 else 10.0)
    iters = 25 if n <= 17 else 12
    ref = qo.Solver(qo.BFGS if method == "bfgs" else qo.DFP, 1e-10, x0, qo.UPDATE_AS_WRITTEN)
    ls = qo.morethuente() if lsname == "mt" else qo.backtracking(1e-4, 0.5)
    ref.minimize(ls, qo.QuadraticOracle(q, b), iters, 20, trace_cap=iters, trace_x=True)
    hp = _run_mp(method, lsname, q, b, x0, iters)
    assert _compare(ref.trace, ref.trace_x, hp, (n, method, lsname)) >= min(3, len(ref.trace))


@pytest.mark.parametrize("name", list(W.WORKLOADS))
@pytest.mark.parametrize("method", ["bfgs", "dfp"])
def test_morethuente_branch_workloads_vs_high_precision(qo, name, method):
    """the workloads that reach cases 2, 3 and 4 and the modified-updating switch (tests/mt_workloads.py), n = 17"""
    n = 17
    w = W.WORKLOADS[name]
    s, _, q = W.run_oracle(qo, n, name, method)
    diag, b, x0 = W.inputs(n, name)
    hp = _run_mp(method, "mt", q, b, x0, w["iters"], h0=w["h0"], t_max=mp.mpf(w["t_max"]) if w["t_max"] is not None else mp.inf)
    t_tol = w.get("t_tol", 1e-9)
    if "t_amp" in w:  # kappa = 1: the gradient shrinks ~50x per iteration and t's conditioning with it (mt_workloads.py)
        t_tol = 1e-6
    assert _compare(s.trace, s.trace_x, hp, (name, method), t_tol=t_tol) >= 1
    if method == "bfgs":  # (that the workloads reach their branches at the GPU test sizes is test_oracle_mt_cases.py's business)
        digits = [d for r in s.trace for d in W.case_digits(r["ls_cases"])]
        for d in w["expect"]:
            assert d in digits

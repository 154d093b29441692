"""Workloads that drive More-Thuente through its cases 2, 3 and 4, the sticky modified-updating switch and the
`tu = +inf` evaluation (morethuente.rs:212-215, 243-293) on the device-resident quadratic objective, i.e. on the
fused / symmetric-storage (bench) path as well as the generic one.

Why these inputs (phi(t) = f(x + t d) is a parabola on a quadratic, t* its minimiser, first trial t = 1, tl = 0):
  * case 1  -- t = 1 overshoots so far that sufficient decrease fails (t* < 0.50005): the benchmark family's usual case;
  * case 2 + modified updating -- sufficient decrease holds, phi'(1) > 0 and strong curvature fails: t* in (0.50005, 0.526).
    Q ~ I (kappa = 1: eigenvalues 1 +- 2/sqrt(3n)) with H0 = 1.95 I puts every fresh direction there;
  * case 3  -- t = 1 undershoots by more than 10x (t* > 10): H0 = 0.05 I.  With t_max = +inf the safeguard
    `t + delta (tu - t)` is +inf and the extrapolated t_plus wins; with a finite t_max it is the safeguard that wins;
  * case 4  -- needs |phi'(1)| > |phi'(0)| with the same sign: negative curvature, so an INDEFINITE Q (the objective accepts any
    symmetric matrix).  It evaluates the oracle at x + tu d: with t_max = +inf that point is non-finite (f = NaN), the cubic
    returns NaN, Rust's max/min drop it, t = 0 and the run ends through `next iterate too close`; with a finite t_max the
    search goes on from the cubic through (t, tu).
`iters` bounds each run to the stretch over which the as-written and rank-2 restatements agree to ~1e-15 (explored on CPU; the
non-convex runs blow up afterwards), so the stated tolerance of the parity sweep applies unchanged -- with ONE exception,
`t_tol` of "case3_inf": there the accepted step is the cubic / secant EXTRAPOLATION from [0, 1] to t* ~ 20 of a function that is
a parabola.  For a parabola the cubic's discriminant z^2 - g_a g_b collapses to (g_a - g_b)^2 / 4 = O(g_a^2 / t*^2): the
~1e-13 absolute difference that two summation orders leave in f(x + d) - f(x) ~ 1e2 comes out of that step as ~1e-11 relative in
t at iteration 0, and iteration 1 -- whose gradient is the small remainder after that near-exact line minimisation -- amplifies
it again to ~1e-8 (measured: 1.5e-8 between the HIP path and the oracle at n = 1024; the two restatements, which share their
summation orders, agree to 1e-16).  That is the conditioning of morethuente.rs:256-272 at tu = +inf, not a property of the
implementation, so this workload states 1e-6 for t (the iterates still meet 1e-9: the step error enters x scaled by |t d| / |x|).
"case2_mod" has the milder form of the same thing: on Q ~ I every exact line minimisation shrinks the gradient ~50x (n = 1024),
so g_k is the small difference of large terms and a relative perturbation eps of x_k shows up as eps ||g_0|| / ||g_k|| in
phi'(t) and in the interpolated step; its step tolerance is therefore max(1e-9, 1e-10 ||g_0|| / ||g_k||) (measured: 2e-9 ... 8e-9
at ||g_k|| / ||g_0|| = 3e-4, 3e-6 at 7e-6).  The benchmark family (kappa = 1e3) loses two orders of gradient in ~100
iterations, not in one, which is why the parity sweep's flat 1e-9 holds there.
"""


def t_tol(name, gnorm_k, gnorm_0, base=1e-9):
    """step-length tolerance of workload `name` at an iteration whose loop-top gradient norm is gnorm_k"""
    w = WORKLOADS[name]
    tol = max(base, w.get("t_tol", 0.0))
    if w.get("t_amp"):
        tol = max(tol, w["t_amp"] * gnorm_0 / gnorm_k)
    return tol

import numpy as np

import problems as P

MOD_BIT = 1 << 30


def case_digits(ls_cases):
    """base-8 digits of a trace record's ls_cases (without the modified-updating bit), first inner iteration first"""
    c = ls_cases & ~MOD_BIT
    out = []
    while c:
        out.append(c & 7)
        c >>= 3
    return out


def count_cases(trace):
    """{digit: occurrences} over a trace, plus the number of line searches that threw the modified-updating switch"""
    cnt = {1: 0, 2: 0, 3: 0, 4: 0}
    mod = 0
    for r in trace:
        for d in case_digits(r["ls_cases"]):
            cnt[d] += 1
        mod += 1 if r["ls_cases"] & MOD_BIT else 0
    return cnt, mod


# name -> (kappa, fraction of negative diagonal entries, H0 = c I or None, t_max or None, iterations, digits it must produce)
WORKLOADS = {
    "case2_mod":      dict(kappa=1.0, neg=0.0, h0=1.95, t_max=None, iters=4, expect={2: 3}, expect_mod=3, t_amp=1e-10),
    "case3_inf":      dict(kappa=1.0, neg=0.0, h0=0.05, t_max=None, iters=3, expect={3: 2}, expect_mod=0, t_tol=1e-6),
    "case3_tmax":     dict(kappa=1.0, neg=0.0, h0=0.05, t_max=4.0, iters=8, expect={3: 3}, expect_mod=0),
    "case4_inf":      dict(kappa=10.0, neg=0.1, h0=None, t_max=None, iters=8, expect={4: 1}, expect_mod=0),
    "case4_tmax4":    dict(kappa=10.0, neg=0.1, h0=None, t_max=4.0, iters=4, expect={4: 1}, expect_mod=0),
    "case4_tmax2":    dict(kappa=10.0, neg=0.1, h0=None, t_max=2.0, iters=4, expect={4: 1}, expect_mod=0),
}


def inputs(n, name, seed=P.SEED):
    """diag (possibly indefinite), b, x0 of workload `name` at dimension n"""
    w = WORKLOADS[name]
    diag = P.synth_diag(n, w["kappa"]).copy()
    k = int(w["neg"] * n)
    diag[:k] = -diag[:k]
    b, x0 = P.synth_vectors(n, seed)
    return diag, b, x0


def run_oracle(qo, n, name, method="bfgs", mode=None, threads=1, seed=P.SEED, memo_q=None):
    """the oracle on workload `name`; returns (solver, status, q)"""
    w = WORKLOADS[name]
    diag, b, x0 = inputs(n, name, seed)
    q = memo_q if memo_q is not None else qo.synth_rows(n, 0, n, seed, diag, nthreads=threads)
    s = qo.Solver(qo.BFGS if method == "bfgs" else qo.DFP, 1e-10, x0, qo.UPDATE_AS_WRITTEN if mode is None else mode, nthreads=threads)
    if w["h0"] is not None:
        s.set_inv_hessian(w["h0"] * np.eye(n))
    kw = {} if w["t_max"] is None else {"t_max": w["t_max"]}
    st = s.minimize(qo.morethuente(**kw), qo.QuadraticOracle(q, b, nthreads=threads), w["iters"], 20, trace_cap=w["iters"], trace_x=True)
    return s, st, q

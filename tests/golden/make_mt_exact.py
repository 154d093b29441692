"""Generates tests/golden/mt_exact_n1024.json: 60-digit traces of the two More-Thuente branch workloads whose step tolerance
tests/mt_workloads.py relaxes ("case2_mod", "case3_inf"), at the GPU test size n = 1024.

Why: the relaxed tolerances (1e-6 on t for the collapsing cubic; 1e-10 ||g_0|| / ||g_k|| on Q ~ I) are argued from conditioning.
This file is the independent pin: the recurrences of the reference path (matrix form of the BFGS / DFP updates, bfgs.rs:112-127 /
dfp.rs:110-116; the line-search decisions of morethuente.rs:165-297 -- the same code as tests/test_oracle_mpmath.py) evaluated in
60-digit arithmetic, against which the GPU tests bound |t_gpu - t_exact| by a small multiple of |t_oracle - t_exact|: the HIP
path must be as close to the truth as the f64 restatement is, not merely close to the restatement.

How it is affordable at n = 1024: (1) the objective's mat-vec is EXACT -- for these workloads (kappa = 1, n a power of two) every
entry of Q is an integer multiple of 2^-62, x is carried as a 300-bit fixed-point integer, and Q x is a product of integer
matrices (numpy object arrays); (2) the inverse Hessian is never formed: H_0 = c I and each update is applied to a vector through
its defining formula, recursively (depth <= 4).

BUILD-GENERATED, NOT REFERENCE-GENERATED (the Rust reference cannot run in this image).  Run: python tests/golden/make_mt_exact.py
(about a minute); the CPU test tests/test_oracle_mt_exact.py re-derives one record of it and checks the C oracle against it."""
import json
import os
import sys

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import qn_oracle as qo  # noqa: E402
import mt_workloads as W  # noqa: E402
from test_oracle_mpmath import _more_thuente  # noqa: E402  (the line search in high precision: one source)

mp.mp.dps = 60
QBITS, XBITS = 62, 300


class ExactQuadratic:
    """f = 1/2 x'Qx - b'x, g = Qx - b with an exact mat-vec (see the module docstring)"""

    def __init__(self, q, b):
        scaled = np.ldexp(q, QBITS)
        assert np.array_equal(scaled, np.rint(scaled)) and np.max(np.abs(scaled)) < 2.0 ** 63, "Q is not a multiple of 2^-62"
        self.qi = np.array([[int(v) for v in row] for row in scaled.astype(np.int64)], dtype=object)
        self.b = [mp.mpf(float(v)) for v in b]
        self.n = len(b)
        self.calls = 0

    def __call__(self, x):
        self.calls += 1
        xi = np.array([int(mp.nint(mp.ldexp(v, XBITS))) for v in x], dtype=object)
        qx = [mp.ldexp(mp.mpf(int(v)), -(QBITS + XBITS)) for v in self.qi.dot(xi)]
        f = mp.fsum(a * c for a, c in zip(x, qx)) / 2 - mp.fsum(a * c for a, c in zip(self.b, x))
        return f, [a - c for a, c in zip(qx, self.b)]


def dot(a, b):
    return mp.fsum(u * v for u, v in zip(a, b))


def norm(a):
    return mp.sqrt(dot(a, a))


def run_exact(method, q, b, x0, iters, h0, t_max, tol=1e-10):
    obj = ExactQuadratic(q, b)
    x = [mp.mpf(float(v)) for v in x0]
    c0 = mp.mpf(h0 if h0 is not None else 1)
    updates = []  # BFGS: (s, y, rho); DFP: (s, hy, ys, yhy)

    def h_apply(v, depth=None):
        depth = len(updates) if depth is None else depth
        if depth == 0:
            return [c0 * u for u in v]
        if method == "bfgs":  # (I - rho s y') H (I - rho y s') v + rho s (s'v)
            s, y, rho = updates[depth - 1]
            sv = dot(s, v)
            w = h_apply([u - rho * sv * yy for u, yy in zip(v, y)], depth - 1)
            yw = dot(y, w)
            return [a - rho * yw * ss + rho * sv * ss for a, ss in zip(w, s)]
        s, hy, ys, yhy = updates[depth - 1]  # H v + s (s'v) / (s'y) - (H y) ((H y)'v) / (y' H y)
        w = h_apply(v, depth - 1)
        sv, hv = dot(s, v), dot(hy, v)
        return [a + ss * sv / ys - hh * hv / yhy for a, ss, hh in zip(w, s, hy)]

    out = []
    s_norm = y_norm = None
    for _ in range(iters):
        f, g = obj(x)
        gn0 = norm(g)
        if (s_norm is not None and s_norm < tol) or (y_norm is not None and y_norm < tol) or gn0 < tol:
            break
        d = [-u for u in h_apply(g)]
        gd0 = dot(g, d)

        def phi(t, x=x, d=d):
            ft, gt = obj([a + t * c for a, c in zip(x, d)])
            return ft, dot(gt, d)
        t, digits, evals = _more_thuente(phi, f, gd0, 20, t_max=t_max)
        if t is None:
            out.append(dict(t=None, digits=digits))
            break
        xn = [a + t * c for a, c in zip(x, d)]
        s = [a - c for a, c in zip(xn, x)]
        _, gn = obj(xn)
        y = [a - c for a, c in zip(gn, g)]
        s_norm, y_norm = norm(s), norm(y)
        out.append(dict(t=t, digits=digits, n_evals=evals + 2, x=xn, f=f, gnorm=gn0, gd0=gd0, s_norm=s_norm))
        x = xn
        if s_norm < tol or y_norm < tol:
            continue
        ys = dot(y, s)
        if method == "bfgs":
            updates.append((s, y, 1 / ys))
        else:
            hy = h_apply(y)
            updates.append((s, hy, ys, dot(y, hy)))
    return out


def dd(v):
    """an mpf as a double-double (hi, lo) in C99 hex: ~32 significant digits, plenty against f64 trajectories"""
    hi = float(v)
    lo = float(v - mp.mpf(hi))
    return [hi.hex(), lo.hex()]


def generate(n, names=("case2_mod", "case3_inf"), methods=("bfgs", "dfp")):
    cases = []
    for name in names:
        w = W.WORKLOADS[name]
        diag, b, x0 = W.inputs(n, name)
        q = qo.synth_rows(n, 0, n, W.P.SEED, diag)
        for method in methods:
            tr = run_exact(method, q, b, x0, w["iters"], w["h0"], mp.mpf(w["t_max"]) if w["t_max"] is not None else mp.inf)
            cases.append(dict(workload=name, method=method, n=n, seed=W.P.SEED, iters=w["iters"],
                              records=[dict(t=mp.nstr(r["t"], 50), t_dd=dd(r["t"]), digits=r["digits"], n_evals=r["n_evals"],
                                            f=mp.nstr(r["f"], 50), gnorm=mp.nstr(r["gnorm"], 30), gd0=mp.nstr(r["gd0"], 30),
                                            s_norm=mp.nstr(r["s_norm"], 30),
                                            x_dd=[dd(v) for v in r["x"]]) for r in tr if r["t"] is not None]))
            print(name, method, "iterations", len(tr), "t =", [mp.nstr(r["t"], 20) for r in tr if r["t"] is not None], flush=True)
    return cases


def main():
    out = {"_comment": "build-generated by tests/golden/make_mt_exact.py: 60-digit evaluation of the reference path's recurrences (exact "
                       "integer mat-vec, implicit inverse Hessian); t as a 50-digit decimal and as a double-double, x as double-doubles "
                       "(C99 hex)", "cases": generate(1024)}
    with open(os.path.join(HERE, "mt_exact_n1024.json"), "w") as fh:
        json.dump(out, fh, separators=(",", ":"))
    print("wrote", os.path.join(HERE, "mt_exact_n1024.json"))


if __name__ == "__main__":
    main()

"""Independent pure-Python restatement of the reference path, used ONLY to cross-check the C oracle on
small cases.  Python floats are IEEE f64 and never fuse a*b+c, like rustc.

Follows: src/ls_solver.rs:66-111, src/quasi_newton/bfgs.rs:42-127, src/quasi_newton/dfp.rs:115-120,
src/line_search/mod.rs:25-83, backtracking.rs:20-58, morethuente.rs:64-297, with nalgebra's operation
order as recalled in SURVEY.md 8(a).  Not the reference; parity unpinned.
"""
import math

INF = float("inf")


def rmax(a, b):  # Rust f64::max: the non-NaN operand
    if a != a:
        return b
    if b != b:
        return a
    return a if a > b else b


def rmin(a, b):
    if a != a:
        return b
    if b != b:
        return a
    return a if a < b else b


def dot(a, b):
    n = len(a)
    acc = [0.0] * 8
    i = 0
    while n - i >= 8:
        for k in range(8):
            acc[k] += a[i + k] * b[i + k]
        i += 8
    res = 0.0
    res += acc[0] + acc[4]
    res += acc[1] + acc[5]
    res += acc[2] + acc[6]
    res += acc[3] + acc[7]
    while i < n:
        res += a[i] * b[i]
        i += 1
    return res


def norm(a):
    return math.sqrt(dot(a, a))


def gemv(h, x):  # h[i][j]; column sweep
    n = len(x)
    y = [h[i][0] * x[0] for i in range(n)]
    for j in range(1, n):
        for i in range(n):
            y[i] = h[i][j] * x[j] + y[i]
    return y


def matmul(a, b):
    n = len(a)
    c = [[0.0] * n for _ in range(n)]
    for j in range(n):
        for i in range(n):
            c[i][j] = a[i][0] * b[0][j]
        for k in range(1, n):
            for i in range(n):
                c[i][j] = a[i][k] * b[k][j] + c[i][j]
    return c


def axpy(x, t, d):
    return [xi + t * di for xi, di in zip(x, d)]


class MoreThuente:
    def __init__(self):
        self.c1, self.c2, self.t_min, self.t_max = 1e-4, 0.9, 0.0, INF
        self.delta_min, self.delta, self.delta_max = 0.58333333, 0.66, 1.1

    @staticmethod
    def update_interval(f_tl, f_t, g_t, tl, t, tu):
        if f_t > f_tl:
            return False, tl, t
        if g_t * (tl - t) > 0.0:
            return False, t, tu
        if g_t * (tl - t) < 0.0:
            return False, t, tl
        return True, tl, tu

    @staticmethod
    def cubic(ta, tb, f_ta, f_tb, g_ta, g_tb):
        s = _div(3.0 * (f_tb - f_ta), tb - ta)
        z = s - g_ta - g_tb
        r = z * z - g_ta * g_tb
        w = math.sqrt(r) if r >= 0 else float("nan")
        return ta + ((tb - ta) * _div(w - g_ta - z, g_tb - g_ta + 2.0 * w))

    @staticmethod
    def quad1(ta, tb, f_ta, f_tb, g_ta):
        lin = _div(f_ta - f_tb, ta - tb)
        return ta - 0.5 * _div((ta - tb) * g_ta, g_ta - lin)

    @staticmethod
    def quad2(ta, tb, g_ta, g_tb):
        return ta - g_ta * _div(ta - tb, g_ta - g_tb)

    def compute_step_len(self, x, f0, g0, d, oracle, max_iter):
        use_mod = False
        conv = False
        t = rmin(rmax(1.0, self.t_min), self.t_max)
        tl, tu = self.t_min, self.t_max
        for _ in range(max_iter):
            f_et, g_et = oracle(axpy(x, t, d))
            if (f_et - f0 <= self.c1 * t * dot(g0, d)) and (abs(dot(g_et, d)) <= self.c2 * abs(dot(g0, d))):
                return t
            elif conv:
                return t
            elif t == tl:
                return t
            elif t == tu:
                return t
            phi_t = (f_et, dot(g_et, d))
            phi_0 = (f0, dot(g0, d))
            psi_t = (phi_t[0] - phi_0[0] - self.c1 * t * phi_0[1], phi_t[1] - self.c1 * phi_0[1])
            if (not use_mod) and psi_t[0] <= 0.0 and phi_t[1] > 0.0:
                use_mod = True
            f_etl, g_etl = oracle(axpy(x, tl, d))
            phi_tl = (f_etl, dot(g_etl, d))
            if use_mod:
                f_tl, g_tl, f_t, g_t = phi_tl[0], phi_tl[1], phi_t[0], phi_t[1]
            else:
                f_tl = phi_tl[0] - phi_0[0] - self.c1 * tl * phi_0[1]
                g_tl = phi_tl[1] - self.c1 * phi_0[1]
                f_t, g_t = psi_t
            if f_t > f_tl:
                tc = self.cubic(tl, t, f_tl, f_t, g_tl, g_t)
                tq = self.quad1(tl, t, f_tl, f_t, g_tl)
                t = tc if abs(tc - tl) < abs(tq - tl) else 0.5 * (tq + tc)
            elif g_t * g_tl < 0.0:
                tc = self.cubic(tl, t, f_tl, f_t, g_tl, g_t)
                ts = self.quad2(tl, t, g_tl, g_t)
                t = tc if abs(tc - t) >= abs(ts - t) else ts
            elif abs(g_t) <= abs(g_tl):
                tc = self.cubic(tl, t, f_tl, f_t, g_tl, g_t)
                ts = self.quad2(tl, t, g_tl, g_t)
                t_plus = tc if abs(tc - t) < abs(ts - t) else ts
                if t > tl:
                    t = rmin(t_plus, t + self.delta * (tu - t))
                else:
                    t = rmax(t_plus, t + self.delta * (tu - t))
            else:
                f_etu, g_etu = oracle(axpy(x, tu, d))
                phi_tu = (f_etu, dot(g_etu, d))
                if use_mod:
                    f_tu, g_tu = phi_tu
                else:
                    f_tu = phi_tu[0] - phi_0[0] - self.c1 * tu * phi_0[1]
                    g_tu = phi_tu[1] - self.c1 * phi_0[1]
                t = self.cubic(tu, t, f_t, f_tu, g_t, g_tu)
            t = rmin(rmax(t, self.t_min), self.t_max)
            conv, tl, tu = self.update_interval(f_tl, f_t, g_t, tl, t, tu)
        return t


def _div(a, b):
    """IEEE division (Python raises on /0)."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0.0:
            return float("nan")
        neg = (math.copysign(1.0, a) < 0) != (math.copysign(1.0, b) < 0)
        return -INF if neg else INF


class BackTracking:
    def __init__(self, c1, beta):
        self.c1, self.beta = c1, beta

    def compute_step_len(self, x, f0, g0, d, oracle, max_iter):
        t, i = 1.0, 0
        while max_iter > i:
            f1, _ = oracle(axpy(x, t, d))
            if f1 != f1 or f1 in (INF, -INF):
                t *= self.beta
                continue
            if f1 - f0 <= self.c1 * t * dot(g0, d):
                return t
            t *= self.beta
            i += 1
        return t


def minimize(method, tol, x0, ls, oracle, max_iter, max_iter_ls):
    """Returns (status, x, k, H, calls, steps) with status in {'ok','max_iter','out_of_domain'}."""
    n = len(x0)
    x = list(x0)
    h = [[1.0 if i == j else 0.0 for j in range(n)] for i in range(n)]
    ident = [[1.0 if i == j else 0.0 for j in range(n)] for i in range(n)]
    s_norm = y_norm = None
    calls = [0]
    steps = []

    def orc(p):
        calls[0] += 1
        f, g = oracle(p)
        return float(f), [float(v) for v in g]

    k = 0
    while max_iter > k:
        f, g = orc(x)
        if f != f or f in (INF, -INF):
            return "out_of_domain", x, k, h, calls[0], steps
        if (s_norm is not None and s_norm < tol) or (y_norm is not None and y_norm < tol) or norm(g) < tol:
            return "ok", x, k, h, calls[0], steps
        d = [-v for v in gemv(h, g)]
        t = ls.compute_step_len(x, f, g, d, orc, max_iter_ls)
        steps.append(t)
        xn = axpy(x, t, d)
        s = [a - b for a, b in zip(xn, x)]
        s_norm = norm(s)
        _, gn = orc(xn)
        y = [a - b for a, b in zip(gn, g)]
        y_norm = norm(y)
        x = xn
        if not (s_norm < tol) and not (y_norm < tol):
            if method == "bfgs":
                ys = dot(y, s)
                rho = _div(1.0, ys)
                left = [[ident[i][j] - (s[i] * y[j]) * rho for j in range(n)] for i in range(n)]
                right = [[ident[i][j] - (s[j] * y[i]) * rho for j in range(n)] for i in range(n)]
                tmp = matmul(matmul(left, h), right)
                h = [[tmp[i][j] + (s[i] * s[j]) * rho for j in range(n)] for i in range(n)]
            else:
                sy = dot(s, y)
                yhy = dot(y, gemv(h, y))
                yy = [[y[i] * y[j] for j in range(n)] for i in range(n)]
                t2 = matmul(matmul(h, yy), h)
                h = [[h[i][j] + (_div(s[i] * s[j], sy) - _div(t2[i][j], yhy)) for j in range(n)] for i in range(n)]
        k += 1
    return "max_iter", x, k, h, calls[0], steps

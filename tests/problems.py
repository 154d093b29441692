"""Shared problem definitions for the tests: the reference's own test/example problems (SURVEY.md 8(c)
G1-G9) and the seeded synthetic SPD quadratic family of SURVEY.md 8(d)."""
import numpy as np

SEED = 0x5EED0001


def g1_quadratic_rs():
    """examples/quadratic.rs:10-43 : f = x'Ix, g = 2Ix, x0 = (1,1), BFGS tol 1e-6, MT default, caps 100/10."""
    m = np.eye(2)
    return dict(fn=lambda x: (x @ (m @ x), 2.0 * (m @ x)), x0=[1.0, 1.0], tol=1e-6, caps=(100, 10))


def g2_bfgs_rs(gamma=1.0):
    """bfgs.rs:141-239 : f = 1/2((x0+1)^2 + gamma (x1-1)^2), x0 = (180,152), tol 1e-12, caps 1000/100000."""
    def fn(x):
        return (0.5 * ((x[0] + 1.0) ** 2 + gamma * (x[1] - 1.0) ** 2), np.array([x[0] + 1.0, gamma * (x[1] - 1.0)]))
    return dict(fn=fn, x0=[180.0, 152.0], tol=1e-12, caps=(1000, 100000))


def g4_bfgs_example_rs():
    """examples/bfgs_example.rs:11-52 : f = x^2+2y^2+3z^2+xy+yz, x0 = (1,1,1), tol 1e-8, caps 50/20."""
    def fn(x):
        x1, x2, x3 = x
        return (x1 * x1 + 2.0 * x2 * x2 + 3.0 * x3 * x3 + x1 * x2 + x2 * x3,
                np.array([2.0 * x1 + x2, 4.0 * x2 + x1 + x3, 6.0 * x3 + x2]))
    return dict(fn=fn, x0=[1.0, 1.0, 1.0], tol=1e-8, caps=(50, 20))


def g5_ill_conditioned(gamma=90.0):
    """backtracking.rs:65-113 / morethuente.rs:303-352 : f = 1/2(x0^2 + gamma x1^2), x0 = (180,152)."""
    def fn(x):
        return (0.5 * (x[0] ** 2 + gamma * x[1] ** 2), np.array([x[0], gamma * x[1]]))
    return dict(fn=fn, x0=[180.0, 152.0])


def g7_dfp_example_rs():
    """examples/dfp_example.rs:9-46 : f = x^2 + 5y^2 + xy, x0 = (2,1), tol 1e-6, caps 100/20."""
    def fn(x):
        return (x[0] ** 2 + 5.0 * x[1] ** 2 + x[0] * x[1], np.array([2.0 * x[0] + x[1], 10.0 * x[1] + x[0]]))
    return dict(fn=fn, x0=[2.0, 1.0], tol=1e-6, caps=(100, 20))


def synth_diag(n, kappa=1e3):
    if n == 1:
        return np.ones(1)
    return kappa ** (np.arange(n, dtype=np.float64) / (n - 1))


def synth_vectors(n, seed=SEED):
    """b, x0 ~ N(0,1) from a counter-based generator (numpy Philox keyed on the seed)."""
    rng = np.random.Generator(np.random.Philox(key=seed))
    b = rng.standard_normal(n)
    x0 = rng.standard_normal(n)
    return b, x0


def synth_problem(qo, n, kappa=1e3, seed=SEED):
    """Dense SPD quadratic of SURVEY.md 8(d): Q_ij = u(seed,i,j)/n off-diagonal, Q_ii = kappa^(i/(n-1))."""
    diag = synth_diag(n, kappa)
    q = qo.synth_rows(n, 0, n, seed, diag)
    b, x0 = synth_vectors(n, seed)
    return q, b, x0, diag

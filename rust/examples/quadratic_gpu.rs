//! BFGS + More-Thuente on a small convex quadratic, three ways: the whole loop on the device with a host closure, the same
//! with a device-resident objective, and the reference's own template loop driven through the trait hooks.
//! NOT COMPILED in the build image (no Rust toolchain; see ../Cargo.toml).
use nalgebra::{DMatrix, DVector};
use optimization_solvers::{FuncEvalMultivariate, LineSearchSolver};
use optimization_solvers_hip::{DeviceObjective, GpuBFGS, GpuMoreThuente};

fn main() {
    let n = 6;
    // Q = tridiag(-1, 4, -1), b = 1: SPD, minimiser Q^-1 b
    let q = DMatrix::from_fn(n, n, |i, j| if i == j { 4.0 } else if i.abs_diff(j) == 1 { -1.0 } else { 0.0 });
    let b = DVector::from_element(n, 1.0);
    let f_and_g = |x: &DVector<f64>| -> FuncEvalMultivariate {
        let qx = &q * x;
        FuncEvalMultivariate::new(0.5 * x.dot(&qx) - b.dot(x), qx - &b)
    };
    let x0 = DVector::from_element(n, 3.0);
    let (tol, max_iter_solver, max_iter_line_search) = (1e-10, 100, 20);

    // 1. one qn_minimize call, the closure called in the reference's order
    let mut solver = GpuBFGS::new(tol, x0.clone());
    let mut ls = GpuMoreThuente::default();
    let mut seen = 0usize;
    let mut on_iteration = |s: &GpuBFGS| seen = *s.k();
    solver.minimize_on_device(&mut ls, f_and_g, max_iter_solver, max_iter_line_search, Some(&mut on_iteration)).unwrap();
    println!("closure on the host : k = {}, ||g|| = {:e}", solver.k(), f_and_g(solver.x()).g().norm());
    assert_eq!(seen, *solver.k());

    // 2. the objective on the device (row-major Q; symmetric, so column-major storage is the same bytes)
    let objective = DeviceObjective::quadratic(q.as_slice(), &b).unwrap();
    let mut solver2 = GpuBFGS::new(tol, x0.clone());
    solver2.minimize_objective(&mut ls, &objective, max_iter_solver, max_iter_line_search).unwrap();
    println!("objective on device : k = {}, f = {:e}", solver2.k(), objective.eval(solver2.x()).unwrap().f());

    // 3. the reference's template loop (ls_solver.rs:66-111) through the trait: H g and the secant update on the GPU.
    //    `solver.minimize(..)` IS the trait method (no inherent method of that name), so a reference call site compiles unchanged
    //    -- with a GPU line search, or with the reference's own `MoreThuente` running on the host:
    let mut solver3 = GpuBFGS::new(tol, x0.clone());
    solver3.minimize(&mut ls, f_and_g, max_iter_solver, max_iter_line_search, None).unwrap();
    println!("trait hooks         : k = {}", LineSearchSolver::k(&solver3));
    assert!((solver.x() - solver3.x()).norm() <= 1e-9);
    let mut solver4 = GpuBFGS::new(tol, x0);
    let mut reference_ls = optimization_solvers::MoreThuente::default();
    solver4.minimize(&mut reference_ls, f_and_g, max_iter_solver, max_iter_line_search, None).unwrap();
    assert!((solver.x() - solver4.x()).norm() <= 1e-9);
}

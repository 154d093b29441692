// qn_host_linesearch.hip.h -- host side, part 2 of 7: the line-search structs' constructors and builders (line_search/*.rs).
#pragma once
// ------------------------------------------------------------------------------------------------
// line-search parameter structs
// ------------------------------------------------------------------------------------------------
extern "C" void qn_morethuente_default(qn_linesearch* ls) { // morethuente.rs:16-28
    memset(ls, 0, sizeof(*ls));
    ls->kind = QN_LS_MORETHUENTE;
    ls->c1 = 1e-4; ls->c2 = 0.9; ls->t_min = 0.0; ls->t_max = INFINITY;
    ls->delta_min = 0.58333333; ls->delta = 0.66; ls->delta_max = 1.1;
}
extern "C" int qn_morethuente_with_deltas(qn_linesearch* ls, double dmin, double d, double dmax) {
    ls->delta_min = dmin; ls->delta = d; ls->delta_max = dmax; return QN_OK;
}
extern "C" int qn_morethuente_with_t_min(qn_linesearch* ls, double t_min) { ls->t_min = t_min; return QN_OK; }
extern "C" int qn_morethuente_with_t_max(qn_linesearch* ls, double t_max) { ls->t_max = t_max; return QN_OK; }
extern "C" int qn_morethuente_with_c1(qn_linesearch* ls, double c1) { // asserts of morethuente.rs:51-52
    if (!(c1 > 0.0)) return fail(QN_ERROR_INPUT_PARAMS, "c1 must be positive");
    if (!(c1 < ls->c2)) return fail(QN_ERROR_INPUT_PARAMS, "c1 must be less than c2");
    ls->c1 = c1; return QN_OK;
}
extern "C" int qn_morethuente_with_c2(qn_linesearch* ls, double c2) { // asserts of morethuente.rs:57-59
    if (!(c2 > 0.0)) return fail(QN_ERROR_INPUT_PARAMS, "c2 must be positive");
    if (!(c2 < 1.0)) return fail(QN_ERROR_INPUT_PARAMS, "c2 must be less than 1");
    if (!(c2 > ls->c1)) return fail(QN_ERROR_INPUT_PARAMS, "c2 must be greater than c1");
    ls->c2 = c2; return QN_OK;
}
extern "C" void qn_morethuente_b_new(qn_linesearch* ls) { // MoreThuenteB::new(n), morethuente_b.rs:18-31
    qn_morethuente_default(ls);
    ls->kind = QN_LS_MORETHUENTE_B;
}
extern "C" void qn_backtracking_b_new(qn_linesearch* ls, double c1, double beta, const double* lb, const double* ub) { // backtracking_b.rs:10-23
    memset(ls, 0, sizeof(*ls));
    ls->kind = QN_LS_BACKTRACKING_B;
    ls->bt_c1 = c1; ls->bt_beta = beta;
    ls->lower_bound_host = lb; ls->upper_bound_host = ub;
}
extern "C" void qn_linesearch_with_lower_bound(qn_linesearch* ls, const double* lb) { ls->lower_bound_host = lb; }
extern "C" void qn_linesearch_with_upper_bound(qn_linesearch* ls, const double* ub) { ls->upper_bound_host = ub; }
extern "C" void qn_backtracking_new(qn_linesearch* ls, double c1, double beta) { // backtracking.rs:8-10
    memset(ls, 0, sizeof(*ls));
    ls->kind = QN_LS_BACKTRACKING;
    ls->bt_c1 = c1; ls->bt_beta = beta;
}

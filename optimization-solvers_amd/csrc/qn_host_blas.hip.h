// qn_host_blas.hip.h -- the thin kernel-level FFI (gemv, rank-2 update, axpy, dot, nrm2 on device buffers): what a Rust host that keeps its own
// loop would bind.
#pragma once
// ------------------------------------------------------------------------------------------------
// kernel-level FFI
// ------------------------------------------------------------------------------------------------
extern "C" int qn_dev_alloc(qn_context* c, size_t bytes, void** out) { HIPCHK(hipSetDevice(c->device)); HIPCHK(hipMalloc(out, bytes)); return QN_OK; }
extern "C" int qn_dev_free(qn_context* c, void* p) { HIPCHK(hipSetDevice(c->device)); HIPCHK(hipFree(p)); return QN_OK; }
extern "C" int qn_h2d(qn_context* c, void* dst, const void* src, size_t bytes) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return QN_OK;
}
extern "C" int qn_d2h(qn_context* c, void* dst, const void* src, size_t bytes) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return QN_OK;
}
extern "C" int qn_gemv(qn_context* c, const double* a, size_t ld, size_t nrows, size_t ncols, const double* x, double* y) {
    HIPCHK(hipSetDevice(c->device));
    if (nrows == 0) return QN_OK;
    hipLaunchKernelGGL(prim_gemv_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, c->stream, a, ld, (int)nrows, (int)ncols, x, y);
    HIPCHK(hipGetLastError());
    return QN_OK;
}
extern "C" int qn_rank2_update(qn_context* c, double* h, size_t ld, size_t row0, size_t nrows, size_t n, const double* s_dev,
                               const double* u_dev, double c_ss, double c_su, double c_uu) {
    HIPCHK(hipSetDevice(c->device));
    if (nrows == 0 || n == 0) return QN_OK;
    dim3 grid((unsigned)std::min<size_t>((n + 255) / 256, 64), (unsigned)nrows);
    hipLaunchKernelGGL(prim_rank2_kernel, grid, dim3(256), 0, c->stream, h, ld, (int)row0, (int)nrows, (int)n, s_dev, u_dev, c_ss, c_su, c_uu);
    HIPCHK(hipGetLastError());
    return QN_OK;
}
extern "C" int qn_axpy(qn_context* c, size_t n, const double* x, double t, const double* d, double* out) {
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) return QN_OK;
    hipLaunchKernelGGL(prim_axpy_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, c->stream, (int)n, x, t, d, out);
    HIPCHK(hipGetLastError());
    return QN_OK;
}
extern "C" int qn_dot(qn_context* c, size_t n, const double* a, const double* b, double* out_host) {
    HIPCHK(hipSetDevice(c->device));
    double* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, sizeof(double)));
    hipLaunchKernelGGL(prim_dot_kernel, dim3(1), dim3(QN_CTL_TPB), 0, c->stream, (int)n, a, b, tmp);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_host, tmp, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipFree(tmp));
    return QN_OK;
}
extern "C" int qn_nrm2(qn_context* c, size_t n, const double* a, double* out_host) { // norm = sqrt(dot(a, a)), bfgs.rs:74,97,99
    double d = 0.0;
    QNCHK(qn_dot(c, n, a, a, &d));
    *out_host = std::sqrt(d);
    return QN_OK;
}

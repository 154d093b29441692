// qn_host_context.hip.h -- host side, part 1 of 7 (included by qn_hip.hip, in this order): the RCCL loader, qn_context and its exchanges (RCCL or
// host-staged), the partition, the communicator checks and probes.  Round 6: qn_hip.hip was one 3 400-line file; the parts are
// textual includes of ONE translation unit -- the device code object is byte for byte what it was (checked at the split).
#pragma once
// ------------------------------------------------------------------------------------------------
// RCCL, loaded lazily so the library itself has no link-time dependency on it
// ------------------------------------------------------------------------------------------------
struct RcclUniqueId { char internal[QN_UNIQUE_ID_BYTES]; };
typedef void* RcclComm;
struct RcclApi {
    void* handle = nullptr;
    int (*GetUniqueId)(RcclUniqueId*) = nullptr;
    int (*CommInitRank)(RcclComm*, int, RcclUniqueId, int) = nullptr;
    int (*CommDestroy)(RcclComm) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, RcclComm, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, RcclComm, hipStream_t) = nullptr; // optional (qn_context_set_allreduce)
    const char* (*GetErrorString)(int) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
};
static RcclApi g_rccl;
static const int kRcclDouble = 8; // ncclFloat64 / ncclDouble (rccl.h)
static const int kRcclSum = 0;    // ncclSum

static int rccl_load() {
    if (g_rccl.handle) return QN_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    const char* forced = getenv("QN_RCCL_LIB"); // the one library to load (deployments with their own build; rehearsals of a missing one)
    if (forced && *forced) h = dlopen(forced, RTLD_NOW | RTLD_GLOBAL);
    else
        for (const char* nm : names) {
            h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
    if (!h) return fail(QN_ABNORMAL_TERMINATION, std::string("cannot load librccl: ") + dlerror());
    g_rccl.GetUniqueId = (int (*)(RcclUniqueId*))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(RcclComm*, int, RcclUniqueId, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(RcclComm))dlsym(h, "ncclCommDestroy");
    g_rccl.AllGather = (int (*)(const void*, void*, size_t, int, RcclComm, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, RcclComm, hipStream_t))dlsym(h, "ncclAllReduce");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(h, "ncclGetErrorString");
    g_rccl.GroupStart = (int (*)())dlsym(h, "ncclGroupStart");
    g_rccl.GroupEnd = (int (*)())dlsym(h, "ncclGroupEnd");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather)
        return fail(QN_ABNORMAL_TERMINATION, "librccl is missing a required symbol");
    g_rccl.handle = h;
    return QN_OK;
}
#define RCCLCHK(expr)                                                                                             \
    do {                                                                                                          \
        int _r = (expr);                                                                                          \
        if (_r != 0)                                                                                              \
            return fail(QN_ABNORMAL_TERMINATION,                                                                  \
                        std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "rccl error")); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
struct qn_context {
    int device = 0;
    hipStream_t stream = nullptr;
    int lu_bulk_cus = 256;
    hipStream_t stream_lu = nullptr; // Newton's LU: the bulk of a trailing update on a stream whose CU mask leaves a quarter of the chip to the panel chain
    hipStream_t stream2 = nullptr; // Newton's Cholesky: the bulk of a trailing update, beside the next block's chain of small kernels (created on first use)
    std::vector<hipEvent_t> la_events; // ... and the events that order the two streams
    int rank = 0, world = 1;
    RcclComm comm = nullptr;
    qn_host_allgather_fn host_xchg = nullptr;
    void* host_xchg_user = nullptr;
    std::vector<double> xchg_send, xchg_recv;
    uint64_t n_comm = 0;
    uint64_t n_xchg_vector = 0, n_xchg_scalar = 0; // collectives a solver enqueued on this context: of n-vectors, of per-workgroup scalars
    // host-staged exchange in STREAM ORDER (qn_context_set_host_exchange_async): pinned staging, the callback runs as a
    // hipLaunchHostFunc node between the two copies, nothing synchronises -- the pipelined launch logic can then be rehearsed
    // with several ranks on one GPU
    int use_allreduce = 0; // symmetric-storage sharded runs: ncclAllReduce of the partial n-vectors instead of all-gather + rank-order sum
    int trial_vector = 0;  // ... the trial's partial n-vector in ONE grouped collective with its evaluation scalars (qn_context_set_trial_vector_exchange)
    int host_async = 0;
    double* pin = nullptr; // [send (cap) | recv (cap * world)]
    size_t pin_cap = 0;
    int host_async_failed = 0;
};
struct HostXchgNode { qn_context* c; size_t count; };
static void host_xchg_node(void* p) {
    HostXchgNode* nd = (HostXchgNode*)p;
    qn_context* c = nd->c;
    if (c->host_xchg(c->host_xchg_user, c->pin, c->pin + c->pin_cap, nd->count) != 0) c->host_async_failed = 1;
    delete nd;
}

extern "C" int qn_device_count(int* out) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *out = 0; return fail(QN_ABNORMAL_TERMINATION, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
    *out = n;
    return QN_OK;
}

static int context_base(int device, qn_context** out) {
    if (!out) return fail(QN_ERROR_INPUT_PARAMS, "out is null");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(QN_ABNORMAL_TERMINATION, "no HIP device visible: libqn_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(QN_ERROR_INPUT_PARAMS, "device ordinal out of range");
    HIPCHK(hipSetDevice(device));
    qn_context* c = new qn_context();
    c->device = device;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(QN_ABNORMAL_TERMINATION, std::string("hipStreamCreate: ") + hipGetErrorString(e)); }
    *out = c;
    return QN_OK;
}

extern "C" int qn_context_create(int device, qn_context** out) { return context_base(device, out); }

extern "C" int qn_comm_unique_id(void* out_128_bytes) {
    QNCHK(rccl_load());
    RcclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    memcpy(out_128_bytes, &id, sizeof(id));
    return QN_OK;
}

extern "C" int qn_context_create_sharded(int device, int rank, int world, const void* unique_id, qn_context** out) {
    if (world < 1 || rank < 0 || rank >= world) return fail(QN_ERROR_INPUT_PARAMS, "bad rank/world");
    QNCHK(context_base(device, out));
    qn_context* c = *out;
    c->rank = rank;
    c->world = world;
    if (world > 1) {
        if (!unique_id) { qn_context_destroy(c); *out = nullptr; return fail(QN_ERROR_INPUT_PARAMS, "unique_id is null"); }
        int s = rccl_load();
        if (s != QN_OK) { qn_context_destroy(c); *out = nullptr; return s; }
        RcclUniqueId id;
        memcpy(&id, unique_id, sizeof(id));
        int r = g_rccl.CommInitRank(&c->comm, world, id, rank);
        if (r != 0) {
            qn_context_destroy(c); *out = nullptr;
            return fail(QN_ABNORMAL_TERMINATION, std::string("ncclCommInitRank: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "error"));
        }
    }
    return QN_OK;
}

extern "C" int qn_context_create_sharded_host_exchange(int device, int rank, int world, qn_host_allgather_fn fn, void* user,
                                                       qn_context** out) {
    if (world < 1 || rank < 0 || rank >= world) return fail(QN_ERROR_INPUT_PARAMS, "bad rank/world");
    if (world > 1 && !fn) return fail(QN_ERROR_INPUT_PARAMS, "exchange function is null");
    QNCHK(context_base(device, out));
    (*out)->rank = rank;
    (*out)->world = world;
    (*out)->host_xchg = fn;
    (*out)->host_xchg_user = user;
    return QN_OK;
}

extern "C" void qn_context_destroy(qn_context* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    if (c->stream2) { (void)hipStreamSynchronize(c->stream2); (void)hipStreamDestroy(c->stream2); }
    if (c->stream_lu) { (void)hipStreamSynchronize(c->stream_lu); (void)hipStreamDestroy(c->stream_lu); }
    for (auto e : c->la_events) (void)hipEventDestroy(e);
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->pin) (void)hipHostFree(c->pin);
    delete c;
}
extern "C" int qn_context_synchronize(qn_context* c) {
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->host_async_failed) { c->host_async_failed = 0; return fail(QN_ABNORMAL_TERMINATION, "host exchange callback failed"); }
    return QN_OK;
}
extern "C" int qn_context_set_allreduce(qn_context* c, int on) {
    if (!c) return fail(QN_ERROR_INPUT_PARAMS, "context is null");
    if (on && c->comm && !g_rccl.AllReduce) return fail(QN_ERROR_INPUT_PARAMS, "librccl has no ncclAllReduce");
    if (on && c->trial_vector) return fail(QN_ERROR_INPUT_PARAMS, "the trial-vector exchange is an all-gather: switch it off first");
    c->use_allreduce = on ? 1 : 0;
    return QN_OK;
}
// Row-sharded second-generation runs (quadratic objective): every evaluation's collective also carries the rank's partial n-vector of the
// trial point, so an accepted evaluation needs no exchange of its own -- E + 1 collectives per iteration instead of E + 2, n doubles per rank
// more per trial (for nodes where a small collective between two launches costs much more than its bytes).  All-gather only: an
// in-place all-reduce overwrites rank 0's slice, which an unused evaluation slot of the pipelined pattern would then send again.
extern "C" int qn_context_set_trial_vector_exchange(qn_context* c, int on) {
    if (!c) return fail(QN_ERROR_INPUT_PARAMS, "context is null");
    if (on && c->use_allreduce) return fail(QN_ERROR_INPUT_PARAMS, "the trial-vector exchange is an all-gather: not with qn_context_set_allreduce");
    c->trial_vector = on ? 1 : 0;
    return QN_OK;
}
extern "C" int qn_context_set_host_exchange_async(qn_context* c, int on) {
    if (!c) return fail(QN_ERROR_INPUT_PARAMS, "context is null");
    if (on && c->world > 1 && !c->host_xchg) return fail(QN_ERROR_INPUT_PARAMS, "not a host-exchange context");
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->host_async = on ? 1 : 0;
    return QN_OK;
}
extern "C" int qn_context_rank(const qn_context* c) { return c->rank; }
extern "C" int qn_context_world(const qn_context* c) { return c->world; }
extern "C" void* qn_context_stream(qn_context* c) { return (void*)c->stream; }

// All-gather of `count` doubles per rank, in place: rank r's slice lives at buf + r*count.
static int exchange(qn_context* c, double* buf, size_t count) {
    if (c->world == 1) return QN_OK;
    c->n_comm++;
    if (c->comm) {
        RCCLCHK(g_rccl.AllGather(buf + (size_t)c->rank * count, buf, count, kRcclDouble, c->comm, c->stream));
        return QN_OK;
    }
    if (c->host_async) { // stream-ordered: D2H copy, host node, H2D copy; the caller's next synchronisation covers all three
        if (count > c->pin_cap) {
            HIPCHK(hipStreamSynchronize(c->stream)); // earlier nodes may still use the old staging area
            if (c->pin) HIPCHK(hipHostFree(c->pin));
            c->pin = nullptr;
            c->pin_cap = std::max(count, (size_t)1 << 16);
            HIPCHK(hipHostMalloc((void**)&c->pin, c->pin_cap * (size_t)(c->world + 1) * sizeof(double), hipHostMallocDefault));
        }
        HIPCHK(hipMemcpyAsync(c->pin, buf + (size_t)c->rank * count, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipLaunchHostFunc(c->stream, host_xchg_node, new HostXchgNode{c, count}));
        HIPCHK(hipMemcpyAsync(buf, c->pin + c->pin_cap, count * (size_t)c->world * sizeof(double), hipMemcpyHostToDevice, c->stream));
        return QN_OK;
    }
    // host-staged exchange (tests / bring-up)
    c->xchg_send.resize(count);
    c->xchg_recv.resize(count * (size_t)c->world);
    HIPCHK(hipMemcpyAsync(c->xchg_send.data(), buf + (size_t)c->rank * count, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (c->host_xchg(c->host_xchg_user, c->xchg_send.data(), c->xchg_recv.data(), count) != 0)
        return fail(QN_ABNORMAL_TERMINATION, "host exchange callback failed");
    HIPCHK(hipMemcpyAsync(buf, c->xchg_recv.data(), count * (size_t)c->world * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return QN_OK;
}

// Several in-place all-gathers issued as ONE RCCL group (one fused collective launch).
struct XchgItem { double* buf; size_t count; };
static int exchange_group(qn_context* c, const XchgItem* items, int nitems) {
    if (c->world == 1) return QN_OK;
    if (c->comm && g_rccl.GroupStart && g_rccl.GroupEnd) {
        c->n_comm++;
        RCCLCHK(g_rccl.GroupStart());
        for (int i = 0; i < nitems; ++i)
            RCCLCHK(g_rccl.AllGather(items[i].buf + (size_t)c->rank * items[i].count, items[i].buf, items[i].count, kRcclDouble, c->comm, c->stream));
        RCCLCHK(g_rccl.GroupEnd());
        return QN_OK;
    }
    for (int i = 0; i < nitems; ++i) QNCHK(exchange(c, items[i].buf, items[i].count));
    return QN_OK;
}

// Sum of `count` doubles per rank over the ranks: rank r's contribution lives at buf + r*count, the total lands in buf[0..count).
// RCCL: ncclAllReduce(ncclSum) -- the operation north_star names; its summation order is RCCL's (ring / tree), identical on all
// ranks but not the rank order of the default all-gather path.  Host exchange (tests): gathered and added in rank order.
__global__ void xchg_rank_sum_kernel(double* __restrict__ buf, size_t count, int world) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x) {
        double acc = buf[i];
        for (int r = 1; r < world; ++r) acc = acc + buf[(size_t)r * count + i];
        buf[i] = acc;
    }
}
static int exchange_sum(qn_context* c, double* buf, size_t count) {
    if (c->world == 1) return QN_OK;
    if (c->comm) {
        if (!g_rccl.AllReduce) return fail(QN_ABNORMAL_TERMINATION, "librccl has no ncclAllReduce");
        c->n_comm++;
        RCCLCHK(g_rccl.AllReduce(buf + (size_t)c->rank * count, buf, count, kRcclDouble, kRcclSum, c->comm, c->stream));
        return QN_OK;
    }
    QNCHK(exchange(c, buf, count));
    hipLaunchKernelGGL(xchg_rank_sum_kernel, dim3((unsigned)std::min<size_t>((count + 255) / 256, 1024)), dim3(256), 0, c->stream, buf, count, c->world);
    HIPCHK(hipGetLastError());
    return QN_OK;
}

// ------------------------------------------------------------------------------------------------
// partition: rank p owns rows [p*rpr, (p+1)*rpr); rpr is a multiple of 16 so every row tile is full
// ------------------------------------------------------------------------------------------------
static int part_rpr(size_t n, int world) {
    size_t per = (n + (size_t)world - 1) / (size_t)world;
    per = (per + 15) / 16 * 16;
    if (per == 0) per = 16;
    return (int)per;
}
static QnTile make_tile(size_t n, const qn_context* c, int cs) {
    QnTile T;
    T.n = (int)n;
    T.rpr = part_rpr(n, c->world);
    T.n_pad = T.rpr * c->world;
    T.row_off = T.rpr * c->rank;
    T.cs = cs;
    T.rank = c->rank;
    return T;
}

extern "C" int qn_partition(size_t n, int world, size_t* rows_per_rank, size_t* n_pad) {
    if (world < 1 || n == 0) return fail(QN_ERROR_INPUT_PARAMS, "bad n/world");
    const int rpr = part_rpr(n, world);
    if (rows_per_rank) *rows_per_rank = (size_t)rpr;
    if (n_pad) *n_pad = (size_t)rpr * (size_t)world;
    return QN_OK;
}

extern "C" int qn_comm_selftest(qn_context* c) {
    HIPCHK(hipSetDevice(c->device));
    QNCHK(rccl_load());
    RcclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    RcclComm comm = nullptr;
    RCCLCHK(g_rccl.CommInitRank(&comm, 1, id, 0));
    const size_t count = 1024;
    double* buf = nullptr;
    HIPCHK(hipMalloc((void**)&buf, count * sizeof(double)));
    std::vector<double> h(count), back(count, 0.0);
    for (size_t i = 0; i < count; ++i) h[i] = 0.5 * (double)i - 3.0;
    HIPCHK(hipMemcpyAsync(buf, h.data(), count * sizeof(double), hipMemcpyHostToDevice, c->stream));
    RCCLCHK(g_rccl.AllGather(buf, buf, count, kRcclDouble, comm, c->stream)); // in place, rank 0 of 1
    HIPCHK(hipMemcpyAsync(back.data(), buf, count * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    g_rccl.CommDestroy(comm);
    HIPCHK(hipFree(buf));
    if (memcmp(h.data(), back.data(), count * sizeof(double)) != 0) return fail(QN_ABNORMAL_TERMINATION, "RCCL self-test: data mismatch");
    return QN_OK;
}

// How long does ONE exchange of `count` doubles per rank take on this context, launch to completion, between other work on the stream?  (Round 6,
// DESIGN section 5 budgets ~20 us per small collective without ever having measured one between devices; bench.py --gpus N
// prints these figures in front of its timed region so that the first run on a multi-GPU node answers the question.)  `reps` exchanges, each
// bracketed by HIP events on the context's stream and each behind a small kernel-sized gap (the exchanges of a run sit between launches, not
// back to back); out_us[0] = median, out_us[1] = minimum, out_us[2] = maximum.  Collective: call on every rank with the same arguments.
extern "C" int qn_context_exchange_probe(qn_context* c, size_t count, int reps, double* out_us) {
    if (!c || !out_us || count == 0 || reps < 1 || reps > 4096) return fail(QN_ERROR_INPUT_PARAMS, "exchange probe: bad arguments");
    out_us[0] = out_us[1] = out_us[2] = 0.0;
    if (c->world == 1) return QN_OK;
    HIPCHK(hipSetDevice(c->device));
    double* buf = nullptr;
    HIPCHK(hipMalloc((void**)&buf, count * (size_t)c->world * sizeof(double)));
    HIPCHK(hipMemsetAsync(buf, 0, count * (size_t)c->world * sizeof(double), c->stream));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int st = QN_OK;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) st = fail(QN_ABNORMAL_TERMINATION, "exchange probe: event creation failed");
    std::vector<float> us;
    for (int r = 0; r < reps + 2 && st == QN_OK; ++r) { // (two untimed ones first: connection set-up, first-touch)
        if (hipEventRecord(e0, c->stream) != hipSuccess) { st = fail(QN_ABNORMAL_TERMINATION, "exchange probe: event record"); break; }
        st = exchange(c, buf, count);
        if (st != QN_OK) break;
        if (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { st = fail(QN_ABNORMAL_TERMINATION, "exchange probe: event"); break; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) { st = fail(QN_ABNORMAL_TERMINATION, "exchange probe: elapsed time"); break; }
        if (r >= 2) us.push_back(1e3f * ms);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(buf);
    if (st != QN_OK) return st;
    if (c->host_async_failed) { c->host_async_failed = 0; return fail(QN_ABNORMAL_TERMINATION, "host exchange callback failed"); }
    std::sort(us.begin(), us.end());
    out_us[0] = us[us.size() / 2]; out_us[1] = us.front(); out_us[2] = us.back();
    return QN_OK;
}

// Cross-rank check of the context's own exchange (RCCL communicator or host callback): every rank contributes a
// rank-tagged slice to one in-place all-gather and verifies all of them.  Collective: call on every rank.
extern "C" int qn_context_comm_check(qn_context* c) {
    if (!c) return fail(QN_ERROR_INPUT_PARAMS, "context is null");
    if (c->world == 1) return QN_OK;
    HIPCHK(hipSetDevice(c->device));
    const size_t count = 4096, total = count * (size_t)c->world;
    double* buf = nullptr;
    HIPCHK(hipMalloc((void**)&buf, total * sizeof(double)));
    std::vector<double> h(total, -1.0);
    for (size_t i = 0; i < count; ++i) h[(size_t)c->rank * count + i] = 1000.0 * (double)c->rank + 0.25 * (double)i;
    HIPCHK(hipMemcpyAsync(buf, h.data(), total * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int st = exchange(c, buf, count);
    if (st == QN_OK) {
        hipError_t e = hipMemcpyAsync(h.data(), buf, total * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) st = fail(QN_ABNORMAL_TERMINATION, std::string("comm check: ") + hipGetErrorString(e));
    }
    (void)hipFree(buf);
    if (st != QN_OK) return st;
    for (int r = 0; r < c->world; ++r)
        for (size_t i = 0; i < count; ++i)
            if (h[(size_t)r * count + i] != 1000.0 * (double)r + 0.25 * (double)i)
                return fail(QN_ABNORMAL_TERMINATION, "comm check: all-gather returned wrong data");
    // ... and the GROUPED path the row kernels use (three all-gathers of different sizes between ncclGroupStart / ncclGroupEnd:
    // vector slices and per-workgroup partial sums), verified the same way
    const size_t counts[3] = {1024, 1024, 9 * 32};
    double* gb[3] = {nullptr, nullptr, nullptr};
    std::vector<double> gh[3];
    for (int k = 0; k < 3 && st == QN_OK; ++k) {
        const size_t tot = counts[k] * (size_t)c->world;
        gh[k].assign(tot, -1.0);
        for (size_t i = 0; i < counts[k]; ++i) gh[k][(size_t)c->rank * counts[k] + i] = 1e6 * (k + 1) + 1000.0 * (double)c->rank + 0.5 * (double)i;
        hipError_t e = hipMalloc((void**)&gb[k], tot * sizeof(double));
        if (e == hipSuccess) e = hipMemcpyAsync(gb[k], gh[k].data(), tot * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) st = fail(QN_ABNORMAL_TERMINATION, std::string("comm check: ") + hipGetErrorString(e));
    }
    if (st == QN_OK && hipStreamSynchronize(c->stream) != hipSuccess) st = fail(QN_ABNORMAL_TERMINATION, "comm check: synchronize");
    if (st == QN_OK) {
        const XchgItem items[3] = {{gb[0], counts[0]}, {gb[1], counts[1]}, {gb[2], counts[2]}};
        st = exchange_group(c, items, 3);
    }
    for (int k = 0; k < 3 && st == QN_OK; ++k) {
        hipError_t e = hipMemcpyAsync(gh[k].data(), gb[k], gh[k].size() * sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) st = fail(QN_ABNORMAL_TERMINATION, std::string("comm check: ") + hipGetErrorString(e));
    }
    for (int k = 0; k < 3; ++k) (void)hipFree(gb[k]);
    if (st != QN_OK) return st;
    if (c->host_async_failed) { c->host_async_failed = 0; return fail(QN_ABNORMAL_TERMINATION, "host exchange callback failed"); }
    for (int k = 0; k < 3; ++k)
        for (int r = 0; r < c->world; ++r)
            for (size_t i = 0; i < counts[k]; ++i)
                if (gh[k][(size_t)r * counts[k] + i] != 1e6 * (k + 1) + 1000.0 * (double)r + 0.5 * (double)i)
                    return fail(QN_ABNORMAL_TERMINATION, "comm check: grouped all-gather returned wrong data");
    return QN_OK;
}

__global__ void qn_empty_kernel() {}
extern "C" int qn_context_event_bracket_overhead(qn_context* c, int reps, double* out_ms) {
    if (!c || !out_ms || reps < 1) return fail(QN_ERROR_INPUT_PARAMS, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    hipEvent_t a = nullptr, b = nullptr;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    // A bracket around k empty kernels reports fixed + k * (one empty dispatch).  The fixed part -- what a bracket adds to the
    // duration of the single kernel inside it -- is 2 * bracket(1) - bracket(2).
    double total[2] = {0.0, 0.0};
    for (int k = 1; k <= 2; ++k) {
        for (int i = 0; i < reps + 5; ++i) {
            HIPCHK(hipStreamSynchronize(c->stream)); // the launch meets an idle stream, as in synchronous mode
            HIPCHK(hipEventRecord(a, c->stream));
            for (int j = 0; j < k; ++j) hipLaunchKernelGGL(qn_empty_kernel, dim3(1), dim3(64), 0, c->stream);
            HIPCHK(hipEventRecord(b, c->stream));
            HIPCHK(hipEventSynchronize(b));
            float ms = 0.f;
            HIPCHK(hipEventElapsedTime(&ms, a, b));
            if (i >= 5) total[k - 1] += ms; // the first few carry one-off costs
        }
    }
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    const double b1 = total[0] / reps, b2 = total[1] / reps;
    *out_ms = std::max(0.0, 2.0 * b1 - b2);
    return QN_OK;
}

static int dev_alloc_zero(double** p, size_t count, hipStream_t st) {
    HIPCHK(hipMalloc((void**)p, count * sizeof(double)));
    HIPCHK(hipMemsetAsync(*p, 0, count * sizeof(double), st));
    return QN_OK;
}

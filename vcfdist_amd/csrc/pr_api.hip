// pr_api.hip -- host side of the C ABI (include/vcfdist_pr.h): device memory,
// work planning, kernel launches on the library's own stream, result download
// and the float finalisation of calc_prec_recall (dist.cpp:1284-1353).
//
// There is no CPU fallback: without a HIP device every entry point that needs one
// returns VPR_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/vcfdist_pr.h"
#include "pr_device.h"
#include "pr_kernels.hip"

namespace {

struct KernelClass { int nt, c, max_len; };
// thread-chunk configurations: a plane of up to nt*c cells per row
const KernelClass CLASSES[] = {
    {64, 1, 64}, {64, 4, 256}, {256, 4, 1024}, {256, 8, 2048}, {1024, 8, 8192}, {1024, 16, 16384}, {1024, 32, 32768},
};
const int N_CLASSES = sizeof(CLASSES) / sizeof(CLASSES[0]);
const size_t LDS_MAX = 160 * 1024;

struct Launch { int cls; int64_t work_off; int32_t count; };   // one k_fwd/k_bwd launch
struct Chunk { std::vector<Launch> launches; int64_t work_off; int32_t count; };

struct EvPair { hipEvent_t a, b; int kind; vpr_launch_stat st; };

}  // namespace

struct vpr_handle {
    vpr_config cfg;
    std::string err;
    hipStream_t stream = nullptr;
    hipStream_t cls_stream[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<void *> allocs;          // batch-lifetime device allocations
    DevBatch dB;
    // host mirrors needed for planning / finalisation
    int32_t n_sc = 0;
    std::vector<int64_t> var_off[4];
    std::vector<float> var_qual[4];
    int64_t n_var[4] = {0, 0, 0, 0};
    std::vector<AlnDesc> descs;
    std::vector<int32_t> work;           // alignment ids, grouped by chunk and class
    std::vector<Chunk> chunks;
    // device side
    AlnDesc *d_descs = nullptr;
    AlnOut *d_outs = nullptr;
    int32_t *d_work = nullptr;
    uint8_t *d_ws = nullptr; size_t ws_bytes = 0;
    PathEnt *d_paths = nullptr; size_t path_entries = 0;
    Section *d_secs = nullptr; int64_t n_secs_cap = 0;
    int32_t *d_fp[4] = {nullptr, nullptr, nullptr, nullptr};
    int32_t **d_fp_table = nullptr;
    EdJob *d_jobs = nullptr; int32_t jobs_cap = 0; int32_t *d_njobs = nullptr;
    uint32_t *d_err = nullptr;
    int32_t *d_ed_scratch = nullptr; size_t ed_scratch_ints = 0;
    std::vector<EvPair> events;
    vpr_timing timing;
    bool uploaded = false, executed = false;
};

namespace {

std::string g_create_err;

int fail(vpr_handle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}

#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(h, VPR_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

template <typename T>
int dev_alloc(vpr_handle *h, T **p, size_t n, bool batch_lifetime = true) {
    *p = nullptr;
    if (n == 0) n = 1;
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess) return fail(h, VPR_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", n * sizeof(T), hipGetErrorString(e));
    if (batch_lifetime) h->allocs.push_back(q);
    *p = static_cast<T *>(q);
    return VPR_OK;
}

template <typename T>
int dev_upload(vpr_handle *h, const T **dst, const T *src, size_t n) {
    T *p;
    int rc = dev_alloc(h, &p, n);
    if (rc) return rc;
    if (n) HIPCHK(h, hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, h->stream));
    *dst = p;
    return VPR_OK;
}

void free_batch(vpr_handle *h) {
    for (void *p : h->allocs) (void)hipFree(p);
    h->allocs.clear();
    for (auto &e : h->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    h->events.clear();
    h->descs.clear(); h->work.clear(); h->chunks.clear();
    h->d_ws = nullptr; h->d_paths = nullptr; h->d_secs = nullptr;
    h->d_ed_scratch = nullptr; h->ed_scratch_ints = 0;
    h->uploaded = h->executed = false;
}

int class_of(int len) {
    for (int k = 0; k < N_CLASSES; k++)
        if (len <= CLASSES[k].max_len) return k;
    return -1;
}

size_t fwd_lds_bytes(int cls, int Lq, int Lr) {
    const int C = CLASSES[cls].c, NT = CLASSES[cls].nt;
    const size_t PQ = (Lq + C - 1) / C * C, PR = (Lr + C - 1) / C * C;
    return (PQ + PR + 8 + 2 * (NT / 64)) * 4;
}
size_t bwd_lds_bytes(int cls, int Lq, int Lr) {
    const int C = CLASSES[cls].c, NT = CLASSES[cls].nt;
    const size_t PQ = (Lq + C - 1) / C * C, PR = (Lr + C - 1) / C * C;
    const size_t FQ = (PQ + 16 + 15) & ~size_t(15), FR = (PR + 16 + 15) & ~size_t(15);
    return (PQ + PR + 8) * 4 + 2 * (FQ + FR) + 4 * (NT / 64) * 4 + 16;
}

typedef void (*AlnKernel)(DevBatch, const AlnDesc *, const int32_t *, uint8_t *, AlnOut *);
AlnKernel fwd_kernel(int cls) {
    switch (cls) {
        case 0: return k_fwd<64, 1>;
        case 1: return k_fwd<64, 4>;
        case 2: return k_fwd<256, 4>;
        case 3: return k_fwd<256, 8>;
        case 4: return k_fwd<1024, 8>;
        case 5: return k_fwd<1024, 16>;
        default: return k_fwd<1024, 32>;
    }
}
AlnKernel bwd_kernel(int cls) {
    switch (cls) {
        case 0: return k_bwd<64, 1>;
        case 1: return k_bwd<64, 4>;
        case 2: return k_bwd<256, 4>;
        case 3: return k_bwd<256, 8>;
        case 4: return k_bwd<1024, 8>;
        case 5: return k_bwd<1024, 16>;
        default: return k_bwd<1024, 32>;
    }
}

int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

}  // namespace

extern "C" {

const char *vpr_last_error(const vpr_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int vpr_create(const vpr_config *cfg, vpr_handle **out) {
    if (!cfg || !out) return fail(nullptr, VPR_ERR_ARG, "vpr_create: null argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, VPR_ERR_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, VPR_ERR_ARG, "device %d out of range (have %d)", cfg->device, ndev);
    vpr_handle *h = new vpr_handle();
    h->cfg = *cfg;
    memset(&h->dB, 0, sizeof(h->dB));
    memset(&h->timing, 0, sizeof(h->timing));
    if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) {
        delete h;
        return fail(nullptr, VPR_ERR_DEVICE, "hipSetDevice/hipStreamCreate failed");
    }
    for (int k = 0; k < N_CLASSES; k++) {
        if (hipStreamCreateWithFlags(&h->cls_stream[k], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_join[k], hipEventDisableTiming) != hipSuccess) {
            return fail(nullptr, VPR_ERR_DEVICE, "hipStreamCreate failed");
        }
    }
    if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess)
        return fail(nullptr, VPR_ERR_DEVICE, "hipEventCreate failed");
    // allow the big classes to use the whole 160 KiB LDS of a CU
    for (int k = 0; k < N_CLASSES; k++) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(fwd_kernel(k)),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(LDS_MAX));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(bwd_kernel(k)),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, int(LDS_MAX));
    }
    *out = h;
    return VPR_OK;
}

void vpr_destroy(vpr_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    free_batch(h);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    for (int k = 0; k < 8; k++) {
        if (h->cls_stream[k]) (void)hipStreamDestroy(h->cls_stream[k]);
        if (h->ev_join[k]) (void)hipEventDestroy(h->ev_join[k]);
    }
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    delete h;
}

int vpr_upload(vpr_handle *h, const vpr_batch *b) {
    if (!h || !b) return VPR_ERR_ARG;
    if (b->n_sc < 0) return fail(h, VPR_ERR_ARG, "negative n_sc");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    free_batch(h);
    const int n = b->n_sc;
    h->n_sc = n;
    DevBatch &D = h->dB;
    memset(&D, 0, sizeof(D));
    D.n_sc = n;
    int rc;
    int64_t hap_len[4], ref_len = b->ref_off[n];
    for (int s = 0; s < 4; s++) {
        hap_len[s] = b->hap_off[s][n];
        h->n_var[s] = b->var_off[s][n];
        if ((rc = dev_upload(h, &D.hap_off[s], b->hap_off[s], n + 1))) return rc;
        if ((rc = dev_upload(h, &D.hap_seq[s], b->hap_seq[s], hap_len[s]))) return rc;
        if ((rc = dev_upload(h, &D.hap_ptr[s], b->hap_ptr[s], hap_len[s]))) return rc;
        if ((rc = dev_upload(h, &D.hap_flag[s], b->hap_flag[s], hap_len[s]))) return rc;
        if ((rc = dev_upload(h, &D.var_off[s], b->var_off[s], n + 1))) return rc;
        if ((rc = dev_upload(h, &D.var_pos[s], b->var_pos[s], h->n_var[s]))) return rc;
        h->var_off[s].assign(b->var_off[s], b->var_off[s] + n + 1);
        h->var_qual[s].assign(b->var_qual[s], b->var_qual[s] + h->n_var[s]);
        if ((rc = dev_alloc(h, &D.has_ins[s], ref_len))) return rc;
        HIPCHK(h, hipMemsetAsync(D.has_ins[s], 0, std::max<int64_t>(ref_len, 1), h->stream));
    }
    if ((rc = dev_upload(h, &D.ref_off, b->ref_off, n + 1))) return rc;
    if ((rc = dev_upload(h, &D.ref_seq, b->ref_seq, ref_len))) return rc;
    for (int q = 0; q < 2; q++) {
        if ((rc = dev_upload(h, &D.ref_ptr[q], b->ref_ptr[q], ref_len))) return rc;
        if ((rc = dev_upload(h, &D.ref_flag[q], b->ref_flag[q], ref_len))) return rc;
        if ((rc = dev_alloc(h, &D.cand_q[q], hap_len[q]))) return rc;
        if ((rc = dev_alloc(h, &D.cand_r[q], ref_len))) return rc;
        HIPCHK(h, hipMemsetAsync(D.cand_q[q], 0xff, std::max<int64_t>(hap_len[q], 1) * sizeof(int4), h->stream));
        HIPCHK(h, hipMemsetAsync(D.cand_r[q], 0xff, std::max<int64_t>(ref_len, 1) * sizeof(int4), h->stream));
    }
    if ((rc = dev_alloc(h, &h->d_err, 1))) return rc;
    HIPCHK(h, hipMemsetAsync(h->d_err, 0, 4, h->stream));

    // ---- K0: position attributes
    hipEvent_t e0, e1;
    HIPCHK(h, hipEventCreate(&e0));
    HIPCHK(h, hipEventCreate(&e1));
    HIPCHK(h, hipEventRecord(e0, h->stream));
    for (int q = 0; q < 2; q++) {
        if (hap_len[q] > 0)
            hipLaunchKernelGGL(k_prep_cand, dim3(unsigned((hap_len[q] + 255) / 256)), dim3(256), 0, h->stream,
                               D, q, 0, hap_len[q], h->d_err);
        if (ref_len > 0)
            hipLaunchKernelGGL(k_prep_cand, dim3(unsigned((ref_len + 255) / 256)), dim3(256), 0, h->stream,
                               D, q, 1, ref_len, h->d_err);
    }
    for (int s = 0; s < 4; s++)
        if (hap_len[s] > 0)
            hipLaunchKernelGGL(k_prep_ins, dim3(unsigned((hap_len[s] + 255) / 256)), dim3(256), 0, h->stream,
                               D, s, hap_len[s]);
    HIPCHK(h, hipEventRecord(e1, h->stream));

    // ---- plan: descriptors, classes, workspace chunks
    h->descs.resize(size_t(n) * 4);
    int64_t sec_total = 0, jobs_total = 0;
    int64_t cells = 0, bytes_alg = 0;
    for (int sc = 0; sc < n; sc++) {
        const int64_t Lr = b->ref_off[sc + 1] - b->ref_off[sc];
        int64_t Lh[4];
        for (int s = 0; s < 4; s++) Lh[s] = b->hap_off[s][sc + 1] - b->hap_off[s][sc];
        int64_t nv = 0;
        for (int s = 0; s < 4; s++) nv += b->var_off[s][sc + 1] - b->var_off[s][sc];
        bytes_alg += 6 * (Lh[0] + Lh[1] + Lh[2] + Lh[3]) + 11 * Lr + 26 * nv;
        for (int i = 0; i < 4; i++) {
            AlnDesc &d = h->descs[size_t(sc) * 4 + i];
            memset(&d, 0, sizeof(d));
            d.qs = i >> 1; d.ts = 2 + (i & 1);
            d.sc = sc; d.aln = i;
            d.q_off = b->hap_off[d.qs][sc]; d.t_off = b->hap_off[d.ts][sc]; d.r_off = b->ref_off[sc];
            d.Lq = int32_t(Lh[d.qs]); d.Lt = int32_t(Lh[d.ts]); d.Lr = int32_t(Lr);
            if (d.Lq < 1 || d.Lt < 1 || d.Lr < 1)
                return fail(h, VPR_ERR_ARG, "supercluster %d has an empty string", sc);
            d.pitch[0] = int32_t(round_up(d.Lq, 32)); d.pitch[1] = int32_t(round_up(d.Lr, 32));
            d.qv_beg = b->var_off[d.qs][sc]; d.qv_end = b->var_off[d.qs][sc + 1];
            d.tv_beg = b->var_off[d.ts][sc]; d.tv_end = b->var_off[d.ts][sc + 1];
            d.sec_cap = int32_t((d.qv_end - d.qv_beg) + (d.tv_end - d.tv_beg) + 4);
            d.sec_off = sec_total;
            sec_total += d.sec_cap;
            jobs_total += std::min(d.Lr, d.Lt) / 33;
            d.path_cap = d.Lq + d.Lr + d.Lt + 4;
            cells += int64_t(d.Lq + d.Lr) * d.Lt;
        }
    }
    bytes_alg += 2 * cells;
    h->timing.cells_dense = cells;
    h->timing.cells_touched = cells;
    h->timing.bytes_algorithmic = bytes_alg;

    // order alignments by matrix size (descending) and cut into workspace chunks
    std::vector<int32_t> order(size_t(n) * 4);
    for (size_t k = 0; k < order.size(); k++) order[k] = int32_t(k);
    auto mat_bytes = [&](int32_t a) {
        const AlnDesc &d = h->descs[a];
        return int64_t(d.pitch[0] + d.pitch[1]) * d.Lt;
    };
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return mat_bytes(x) > mat_bytes(y); });

    size_t free_b = 0, total_b = 0;
    HIPCHK(h, hipMemGetInfo(&free_b, &total_b));
    int64_t fixed = sec_total * int64_t(sizeof(Section)) + int64_t(n) * 4 * (sizeof(AlnDesc) + sizeof(AlnOut) + 4) +
                    (h->n_var[0] + h->n_var[1]) * 8 + jobs_total * int64_t(sizeof(EdJob)) + (64 << 20);
    int64_t budget = h->cfg.workspace_bytes > 0 ? h->cfg.workspace_bytes : int64_t(free_b * 0.8) - fixed;
    if (budget < (16 << 20)) budget = 16 << 20;

    h->work.clear();
    size_t k = 0;
    int64_t max_ws = 0, max_path = 0;
    while (k < order.size()) {
        Chunk ch;
        ch.work_off = int64_t(h->work.size());
        int64_t ws = 0, pe = 0;
        std::vector<std::vector<int32_t>> by_cls(N_CLASSES);
        size_t k0 = k;
        while (k < order.size()) {
            AlnDesc &d = h->descs[order[k]];
            const int64_t mb = round_up(mat_bytes(order[k]), 64) + 64;
            const int64_t pb = d.path_cap;
            if (k > k0 && ws + mb + (pe + pb) * int64_t(sizeof(PathEnt)) > budget) break;
            const int cls = class_of(std::max(d.Lq, d.Lr));
            if (cls < 0 || fwd_lds_bytes(cls, d.Lq, d.Lr) > LDS_MAX || bwd_lds_bytes(cls, d.Lq, d.Lr) > LDS_MAX)
                return fail(h, VPR_ERR_ARG, "supercluster %d alignment %d too long for this build (Lq=%d Lr=%d)",
                            d.sc, d.aln, d.Lq, d.Lr);
            d.mat_off[0] = ws;
            d.mat_off[1] = ws + round_up(int64_t(d.pitch[0]) * d.Lt, 32);
            d.path_off = pe;
            ws += mb;
            pe += pb;
            by_cls[cls].push_back(order[k]);
            k++;
        }
        if (ws + pe * int64_t(sizeof(PathEnt)) > budget && k - k0 == 1 && h->cfg.workspace_bytes > 0)
            return fail(h, VPR_ERR_NOMEM, "workspace budget %lld too small for one alignment (%lld bytes)",
                        (long long)budget, (long long)ws);
        for (int c = 0; c < N_CLASSES; c++) {
            if (by_cls[c].empty()) continue;
            Launch L{c, int64_t(h->work.size()), int32_t(by_cls[c].size())};
            h->work.insert(h->work.end(), by_cls[c].begin(), by_cls[c].end());
            ch.launches.push_back(L);
        }
        ch.count = int32_t(h->work.size() - ch.work_off);
        max_ws = std::max(max_ws, ws);
        max_path = std::max(max_path, pe);
        h->chunks.push_back(std::move(ch));
    }

    if ((rc = dev_alloc(h, &h->d_descs, h->descs.size()))) return rc;
    if (!h->descs.empty())
        HIPCHK(h, hipMemcpyAsync(h->d_descs, h->descs.data(), h->descs.size() * sizeof(AlnDesc), hipMemcpyHostToDevice, h->stream));
    if ((rc = dev_alloc(h, &h->d_outs, h->descs.size()))) return rc;
    if ((rc = dev_alloc(h, &h->d_work, h->work.size()))) return rc;
    if (!h->work.empty())
        HIPCHK(h, hipMemcpyAsync(h->d_work, h->work.data(), h->work.size() * 4, hipMemcpyHostToDevice, h->stream));
    h->ws_bytes = size_t(max_ws);
    if ((rc = dev_alloc(h, &h->d_ws, h->ws_bytes + 64))) return rc;
    h->path_entries = size_t(max_path);
    if ((rc = dev_alloc(h, &h->d_paths, h->path_entries + 1))) return rc;
    h->n_secs_cap = sec_total;
    if ((rc = dev_alloc(h, &h->d_secs, size_t(sec_total)))) return rc;
    for (int q = 0; q < 4; q++)
        if ((rc = dev_alloc(h, &h->d_fp[q], size_t(h->n_var[q >> 1])))) return rc;
    if ((rc = dev_alloc(h, &h->d_fp_table, 4))) return rc;
    HIPCHK(h, hipMemcpyAsync(h->d_fp_table, h->d_fp, sizeof(h->d_fp), hipMemcpyHostToDevice, h->stream));
    h->jobs_cap = int32_t(std::min<int64_t>(jobs_total + 1, 0x7fffffff));
    if ((rc = dev_alloc(h, &h->d_jobs, size_t(h->jobs_cap)))) return rc;
    if ((rc = dev_alloc(h, &h->d_njobs, 1))) return rc;

    HIPCHK(h, hipStreamSynchronize(h->stream));
    uint32_t err = 0;
    HIPCHK(h, hipMemcpy(&err, h->d_err, 4, hipMemcpyDeviceToHost));
    if (err) return fail(h, VPR_ERR_ARG, "more than 4 swap sources map to one position (unsupported variant layout)");
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    h->timing.ms_prep = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    h->uploaded = true;
    return VPR_OK;
}

int vpr_upload_variants(vpr_handle *h, const vpr_variants *v) {
    if (!h || !v) return VPR_ERR_ARG;
    vpr_owned_batch *ob = nullptr;
    int rc = vpr_batch_from_variants(v, &ob);
    if (rc) return fail(h, rc, "vpr_batch_from_variants failed (%d): unsorted/overlapping variants or bad coordinates", rc);
    rc = vpr_upload(h, vpr_owned_batch_view(ob));
    vpr_owned_batch_free(ob);
    return rc;
}

int vpr_execute(vpr_handle *h) {
    if (!h) return VPR_ERR_ARG;
    if (!h->uploaded) return fail(h, VPR_ERR_STATE, "vpr_execute before vpr_upload");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    for (auto &e : h->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    h->events.clear();
    hipStream_t st = h->stream;
    HIPCHK(h, hipMemsetAsync(h->d_outs, 0, std::max<size_t>(h->descs.size(), 1) * sizeof(AlnOut), st));
    for (int q = 0; q < 4; q++)
        HIPCHK(h, hipMemsetAsync(h->d_fp[q], 0xff, std::max<int64_t>(h->n_var[q >> 1], 1) * 4, st));
    HIPCHK(h, hipMemsetAsync(h->d_njobs, 0, 4, st));

    // HIP events bracket each launch on the stream the kernel is launched on
    auto timed = [&](int kind, const vpr_launch_stat &ls, hipStream_t ks, auto &&launch) -> int {
        EvPair ev; ev.kind = kind; ev.st = ls; ev.st.kind = kind;
        HIPCHK(h, hipEventCreate(&ev.a));
        HIPCHK(h, hipEventCreate(&ev.b));
        HIPCHK(h, hipEventRecord(ev.a, ks));
        launch();
        HIPCHK(h, hipEventRecord(ev.b, ks));
        h->events.push_back(ev);
        return VPR_OK;
    };
    hipEvent_t t0, t1;
    HIPCHK(h, hipEventCreate(&t0));
    HIPCHK(h, hipEventCreate(&t1));
    HIPCHK(h, hipEventRecord(t0, st));
    int64_t n_fwd = 0;
    for (const Chunk &ch : h->chunks) {
        // fork: the kernel classes of one chunk touch disjoint alignments and workspace regions, so each
        // class runs its K1 -> K2 -> K3 pipeline on its own stream and the chunk joins before the
        // workspace is reused
        HIPCHK(h, hipEventRecord(h->ev_fork, st));
        for (const Launch &L : ch.launches) {
            const KernelClass &K = CLASSES[L.cls];
            hipStream_t ks = h->cls_stream[L.cls];
            HIPCHK(h, hipStreamWaitEvent(ks, h->ev_fork, 0));
            size_t lds_f = 0, lds_b = 0;
            vpr_launch_stat ls;
            memset(&ls, 0, sizeof(ls));
            ls.threads = K.nt; ls.cells_per_thread = K.c; ls.n_units = L.count;
            int64_t in_bytes = 0;
            for (int32_t w = 0; w < L.count; w++) {   // the launch's dynamic LDS = its largest member
                const AlnDesc &d = h->descs[h->work[L.work_off + w]];
                lds_f = std::max(lds_f, fwd_lds_bytes(L.cls, d.Lq, d.Lr));
                lds_b = std::max(lds_b, bwd_lds_bytes(L.cls, d.Lq, d.Lr));
                ls.cells += int64_t(d.Lq + d.Lr) * d.Lt;
                // strings 1 B/base; pointer+flag arrays 5 B/element (q->r, r->q, t->r)
                in_bytes += 6 * int64_t(d.Lq) + 6 * int64_t(d.Lt) + 6 * int64_t(d.Lr);
            }
            ls.bytes_algorithmic = ls.cells + in_bytes;
            int rc = timed(1, ls, ks, [&] {
                hipLaunchKernelGGL(fwd_kernel(L.cls), dim3(L.count), dim3(K.nt), lds_f, ks, h->dB, h->d_descs,
                                   h->d_work + L.work_off, h->d_ws, h->d_outs);
                hipLaunchKernelGGL(k_fwd_finish, dim3((L.count + 255) / 256), dim3(256), 0, ks,
                                   h->d_work + L.work_off, L.count, h->d_outs);
            });
            if (rc) return rc;
            n_fwd++;
            ls.bytes_algorithmic = ls.cells;
            rc = timed(2, ls, ks, [&] {
                hipLaunchKernelGGL(bwd_kernel(L.cls), dim3(L.count), dim3(K.nt), lds_b, ks, h->dB, h->d_descs,
                                   h->d_work + L.work_off, h->d_ws, h->d_outs);
            });
            if (rc) return rc;
            vpr_launch_stat ws_;
            memset(&ws_, 0, sizeof(ws_));
            ws_.threads = 64; ws_.n_units = L.count;
            rc = timed(3, ws_, ks, [&] {
                hipLaunchKernelGGL(k_walk, dim3((L.count + 63) / 64), dim3(64), 0, ks, h->dB, h->d_descs,
                                   h->d_work + L.work_off, L.count, h->d_ws, h->d_outs, h->d_paths, h->d_secs,
                                   h->d_fp_table, h->d_jobs, h->d_njobs, h->jobs_cap);
            });
            if (rc) return rc;
            HIPCHK(h, hipEventRecord(h->ev_join[L.cls], ks));
            HIPCHK(h, hipStreamWaitEvent(st, h->ev_join[L.cls], 0));
        }
    }
    // K4: deferred section edit distances
    int32_t n_jobs = 0;
    HIPCHK(h, hipMemcpyAsync(&n_jobs, h->d_njobs, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    n_jobs = std::min(n_jobs, h->jobs_cap);
    if (n_jobs > 0) {
        std::vector<EdJob> jobs(n_jobs);
        HIPCHK(h, hipMemcpy(jobs.data(), h->d_jobs, size_t(n_jobs) * sizeof(EdJob), hipMemcpyDeviceToHost));
        int64_t stride = 0;
        for (const EdJob &j : jobs) stride = std::max<int64_t>(stride, std::max(j.ref_len, j.tru_len) + 1);
        stride = round_up(stride, 16);
        // run in slices so the scratch stays bounded
        const int64_t max_ints = int64_t(1) << 28;   // 1 GiB
        const int32_t per = int32_t(std::max<int64_t>(1, std::min<int64_t>(n_jobs, max_ints / stride)));
        if (h->ed_scratch_ints < size_t(per * stride)) {
            int32_t *p;
            int rc = dev_alloc(h, &p, size_t(per * stride));
            if (rc) return rc;
            h->d_ed_scratch = p;
            h->ed_scratch_ints = size_t(per * stride);
        }
        for (int32_t j0 = 0; j0 < n_jobs; j0 += per) {
            const int32_t cnt = std::min(per, n_jobs - j0);
            vpr_launch_stat es_;
            memset(&es_, 0, sizeof(es_));
            es_.threads = 64; es_.n_units = cnt;
            int rc = timed(4, es_, st, [&] {
                hipLaunchKernelGGL(k_ed, dim3(cnt), dim3(64), 0, st, h->dB, h->d_descs, h->d_jobs + j0, cnt,
                                   h->d_secs, h->d_ed_scratch, stride);
            });
            if (rc) return rc;
        }
    }
    HIPCHK(h, hipEventRecord(t1, st));
    HIPCHK(h, hipStreamSynchronize(st));
    HIPCHK(h, hipGetLastError());
    float ms = 0;
    (void)hipEventElapsedTime(&ms, t0, t1);
    h->timing.ms_total = ms;
    h->timing.ms_fwd = h->timing.ms_bwd = h->timing.ms_walk = h->timing.ms_ed = 0;
    for (auto &e : h->events) {
        float m = 0;
        (void)hipEventElapsedTime(&m, e.a, e.b);
        e.st.ms = m;
        if (e.kind == 1) h->timing.ms_fwd += m;
        else if (e.kind == 2) h->timing.ms_bwd += m;
        else if (e.kind == 3) h->timing.ms_walk += m;
        else h->timing.ms_ed += m;
    }
    h->timing.n_fwd_launches = n_fwd;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    h->executed = true;
    return VPR_OK;
}

int vpr_get_launch_stats(const vpr_handle *h, vpr_launch_stat *out, int32_t cap) {
    if (!h) return VPR_ERR_ARG;
    const int32_t n = int32_t(h->events.size());
    for (int32_t k = 0; k < n && k < cap && out; k++) out[k] = h->events[k].st;
    return n;
}

int vpr_get_timing(const vpr_handle *h, vpr_timing *t) {
    if (!h || !t) return VPR_ERR_ARG;
    *t = h->timing;
    return VPR_OK;
}

int vpr_download(vpr_handle *h, vpr_results *res) {
    if (!h || !res) return VPR_ERR_ARG;
    if (!h->executed) return fail(h, VPR_ERR_STATE, "vpr_download before vpr_execute");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const size_t na = h->descs.size();
    std::vector<AlnOut> outs(na);
    std::vector<Section> secs(size_t(h->n_secs_cap));
    std::vector<int32_t> fp[4];
    if (na) HIPCHK(h, hipMemcpy(outs.data(), h->d_outs, na * sizeof(AlnOut), hipMemcpyDeviceToHost));
    if (!secs.empty()) HIPCHK(h, hipMemcpy(secs.data(), h->d_secs, secs.size() * sizeof(Section), hipMemcpyDeviceToHost));
    for (int q = 0; q < 4; q++) {
        fp[q].resize(size_t(h->n_var[q >> 1]));
        if (!fp[q].empty()) HIPCHK(h, hipMemcpy(fp[q].data(), h->d_fp[q], fp[q].size() * 4, hipMemcpyDeviceToHost));
    }
    const float max_qual = h->cfg.max_qual;
    const double thr = h->cfg.credit_threshold;

    auto job = [&](int sc0, int sc1) {
        for (int sc = sc0; sc < sc1; sc++) {
            int32_t s[4];
            for (int i = 0; i < 4; i++) {
                const size_t a = size_t(sc) * 4 + i;
                const AlnOut &O = outs[a];
                const AlnDesc &d = h->descs[a];
                s[i] = O.s;
                res->aln_dist[a] = O.s;
                res->aln_end_plane[a] = uint8_t(O.end_plane);
                res->aln_beg_plane[a] = uint8_t(O.beg_plane);
                uint32_t status = O.status;
                const int swap = (i == 1 || i == 2);
                const int qs = d.qs, ts = d.ts;
                const float *qq = h->var_qual[qs].data();
                const int32_t *fpg = fp[qs * 2 + swap].data();
                // query variants passed on the REF plane: FP in a group of their own, dist.cpp:1157-1168
                for (int64_t v = d.qv_beg; v < d.qv_end; v++) {
                    if (fpg[v] >= 0) {
                        res->errtype[qs][swap][v] = VPR_ERRTYPE_FP;
                        res->sync_group[qs][swap][v] = fpg[v];
                        res->credit[qs][swap][v] = 0;
                        res->ref_ed[qs][swap][v] = 0;
                        res->query_ed[qs][swap][v] = 0;
                        res->callq[qs][swap][v] = qq[v];
                    }
                }
                if (!(status & (VPR_ST_ERR_NO_PTR | VPR_ST_ERR_UNFINISHED))) {
                    for (int k = 0; k < O.n_sec; k++) {
                        const Section &S = secs[size_t(d.sec_off) + k];
                        int ref_ed = S.ref_ed;
                        const int query_ed = S.query_ed;
                        const bool has_q = S.q_hi != S.q_lo, has_t = S.t_hi != S.t_lo;
                        if (!has_t && ref_ed != 0) status |= VPR_ST_WARN_REF_ED;            // dist.cpp:1203
                        if (!has_q && query_ed != ref_ed) status |= VPR_ST_WARN_QUERY_ED;   // dist.cpp:1207
                        if (query_ed > ref_ed) status |= VPR_ST_WARN_EXCEEDS;               // dist.cpp:1211
                        if (ref_ed == 0 && has_t) { status |= VPR_ST_WARN_ZERO_ED; ref_ed = 1; }  // dist.cpp:1219-1223
                        float callq = max_qual;                                             // dist.cpp:1284-1288
                        for (int64_t v = S.q_hi; v > S.q_lo; v--) callq = std::min(callq, qq[v]);
                        for (int64_t v = S.q_hi; v > S.q_lo; v--) {
                            float credit = 1 - float(query_ed) / ref_ed;
                            if (fpg[v] < 0) {   // "don't overwrite FPs", dist.cpp:1295
                                res->errtype[qs][swap][v] = (credit >= thr) ? VPR_ERRTYPE_TP : VPR_ERRTYPE_FP;
                                res->sync_group[qs][swap][v] = S.sync_group;
                                res->credit[qs][swap][v] = credit;
                                res->ref_ed[qs][swap][v] = ref_ed;
                                res->query_ed[qs][swap][v] = query_ed;
                                res->callq[qs][swap][v] = callq;
                            }
                        }
                        for (int64_t v = S.t_hi; v > S.t_lo; v--) {
                            float credit = 1 - float(query_ed) / ref_ed;
                            const bool tp = credit >= thr;
                            res->errtype[ts][swap][v] = tp ? VPR_ERRTYPE_TP : VPR_ERRTYPE_FN;
                            res->sync_group[ts][swap][v] = S.sync_group;
                            res->credit[ts][swap][v] = credit;
                            res->ref_ed[ts][swap][v] = ref_ed;
                            res->query_ed[ts][swap][v] = query_ed;
                            res->callq[ts][swap][v] = tp ? callq : max_qual;
                        }
                    }
                }
                res->aln_status[a] = status;
            }
            res->sc_phase[sc] = vpr_store_phase(s, h->cfg.phase_threshold, &res->orig_phase_dist[sc],
                                                &res->swap_phase_dist[sc]);
        }
    };
    const int n = h->n_sc;
    const int nth = int(std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32));
    if (n < 4096 || nth == 1) {
        job(0, n);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < nth; t++)
            th.emplace_back(job, int(int64_t(n) * t / nth), int(int64_t(n) * (t + 1) / nth));
        for (auto &x : th) x.join();
    }
    return VPR_OK;
}

int vpr_run(vpr_handle *h, const vpr_batch *batch, vpr_results *res) {
    int rc = vpr_upload(h, batch);
    if (rc) return rc;
    if ((rc = vpr_execute(h))) return rc;
    return vpr_download(h, res);
}

int64_t vpr_download_path(const vpr_handle *h, int32_t sc, int32_t aln, int64_t cap,
                          uint8_t *plane, int32_t *qri, int32_t *ti, uint8_t *sync, uint8_t *edit) {
    if (!h || !h->executed || sc < 0 || sc >= h->n_sc || aln < 0 || aln > 3) return VPR_ERR_ARG;
    // the path scratch is reused per workspace chunk: only alignments of the last chunk are still resident
    const size_t a = size_t(sc) * 4 + aln;
    const Chunk &last = h->chunks.back();
    bool in_last = false;
    for (int32_t w = 0; w < last.count; w++) in_last |= (h->work[last.work_off + w] == int32_t(a));
    if (!in_last) return VPR_ERR_STATE;
    AlnOut O;
    if (hipMemcpy(&O, h->d_outs + a, sizeof(O), hipMemcpyDeviceToHost) != hipSuccess) return VPR_ERR_DEVICE;
    const int64_t n = std::min<int64_t>(O.path_len, cap);
    std::vector<PathEnt> p(n);
    if (n && hipMemcpy(p.data(), h->d_paths + h->descs[a].path_off, n * sizeof(PathEnt), hipMemcpyDeviceToHost) != hipSuccess)
        return VPR_ERR_DEVICE;
    for (int64_t k = 0; k < n; k++) {
        plane[k] = uint8_t(p[k].a >> 31);
        qri[k] = int32_t(p[k].a & 0x7fffffffu);
        ti[k] = int32_t(p[k].b & 0x3fffffffu);
        sync[k] = uint8_t(p[k].b >> 31);
        edit[k] = uint8_t((p[k].b >> 30) & 1);
    }
    return O.path_len;
}

}  // extern "C"

// pr_collect.hip -- the precision/recall counters and the path's collectives: the histogram kernel of the per-variant results
// (print.cpp:328-438: counts[callset][type][TP, FP, FN][quality thresholds]), vpr_pr_counts, and -- RCCL on the library's own
// stream, on the caller's communicator -- vpr_allreduce_counts (ONE all-reduce of the histogram) and vpr_allgather_phase (the
// per-supercluster phasing of all ranks, for the per-contig Viterbi every rank then runs: phase.cpp:285-355); vpr_run.
#include <dlfcn.h>

#include "pr_host.h"

extern "C" {

// histogram of floor(callq) per (class, errtype) of the phasing each supercluster selects, for one hap slot;
// privatised per workgroup in LDS (a few thousand bins, heavily contended), flushed once
__global__ void __launch_bounds__(256) k_pr_hist(const int64_t *__restrict__ var_off, int n_sc, int64_t n_var,
                          const uint8_t *__restrict__ cls, const int32_t *__restrict__ sc_phase,
                          const int32_t *__restrict__ pb_phase, VarCols c0, VarCols c1, int callset, int min_qual,
                          int max_qual, unsigned long long *__restrict__ hist /* [2][3 classes][3][nq + 1] */) {
    extern __shared__ unsigned int blk[];      // [3][3][nq + 1]
    const int nq = max_qual - min_qual + 1, nb = 9 * (nq + 1);
    for (int k = threadIdx.x; k < nb; k += blockDim.x) blk[k] = 0;
    __syncthreads();
    const int64_t v = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (v < n_var) {
        int lo = 0, hi = n_sc;   // supercluster of the variant: largest sc with var_off[sc] <= v
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (var_off[mid] <= v) lo = mid; else hi = mid; }
        const int ph = sc_phase[lo];
        const int swap = ph == VPR_PHASE_ORIG ? 0 : (ph == VPR_PHASE_SWAP ? 1 : (pb_phase ? (pb_phase[lo] != 0) : 0));
        const VarCols &C = swap ? c1 : c0;
        const int e = C.errtype[v];
        if (e < 3) {                                         // ERRTYPE_UN etc.: skipped with a warning (print.cpp:374)
            const float q = C.callq[v];
            int b = (q < float(min_qual)) ? -1 : int(floorf(q)) - min_qual;   // last threshold index the variant counts at
            if (b >= nq) b = nq - 1;
            const int t = cls[v] > 2 ? 2 : cls[v];
            // bin nq collects the variants that count at no threshold (callq < min_qual)
            atomicAdd(&blk[(t * 3 + e) * (nq + 1) + (b < 0 ? nq : b)], 1u);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nb; k += blockDim.x)
        if (blk[k]) atomicAdd(&hist[size_t(callset) * nb + k], (unsigned long long)blk[k]);
}

int vpr_upload_var_class(vpr_handle *h, const uint8_t *const var_class[VPR_HAPS]) {
    if (!h || !var_class) return VPR_ERR_ARG;
    if (!h->uploaded) return fail(h, VPR_ERR_STATE, "vpr_upload_var_class before vpr_upload");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    for (int s = 0; s < VPR_HAPS; s++) {
        if (!h->d_cls[s]) {
            int rc = dev_alloc(h, &h->d_cls[s], size_t(h->n_var[s]));
            if (rc) return rc;
        }
        if (h->n_var[s]) HIPCHK(h, hipMemcpyAsync(h->d_cls[s], var_class[s], size_t(h->n_var[s]), hipMemcpyHostToDevice, h->stream));
    }
    HIPCHK(h, x_sync(h, h->stream, SITE));
    return VPR_OK;
}

}   // extern "C"

namespace {
// RCCL, resolved at run time: the symbols the process already has (a host that links librccl, PyTorch's copy in a Python
// process: the communicator the caller passes belongs to that one), else librccl.so.1.  The library itself has no link-time
// dependency on RCCL: a single-GPU caller never needs it.
struct Rccl {
    typedef int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
    typedef int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t);
    typedef const char *(*ErrStr)(int);
    AllReduce all_reduce = nullptr;
    AllGather all_gather = nullptr;
    ErrStr err_str = nullptr;
    std::string path;       // the library the functions come from ("" = the process's global symbols)
    // the copy of RCCL this process has ALREADY MAPPED, if any (/proc/self/maps): PyTorch loads its own librccl.so with local
    // visibility, so the global symbol table does not show it, and opening "librccl.so.1" by name beside it could bring a SECOND
    // copy into the process -- a communicator created by one copy and used through the other's functions is a crash
    static std::string mapped_rccl() {
        std::string found;
        if (FILE *f = fopen("/proc/self/maps", "r")) {
            char line[4096];
            while (fgets(line, sizeof(line), f)) {
                const char *p = strchr(line, '/');
                if (!p) continue;
                std::string pth(p);
                while (!pth.empty() && (pth.back() == '\n' || pth.back() == ' ')) pth.pop_back();
                const size_t sl = pth.rfind('/');
                if (pth.compare(sl + 1, 7, "librccl") == 0 && pth.find(".so") != std::string::npos) { found = pth; break; }
            }
            fclose(f);
        }
        return found;
    }
    static const Rccl &get() {
        static Rccl r = [] {
            Rccl x;
            void *hd = nullptr;
            const std::string mapped = mapped_rccl();
            if (!mapped.empty()) {          // the copy that is there (RTLD_NOLOAD: a handle to it, never a new mapping)
                hd = dlopen(mapped.c_str(), RTLD_NOW | RTLD_NOLOAD);
                if (hd) x.path = mapped;
            }
            if (!hd && dlsym(RTLD_DEFAULT, "ncclAllReduce")) hd = RTLD_DEFAULT;        // a host that links RCCL itself
            if (!hd) {
                for (const char *name : {"librccl.so.1", "librccl.so"}) {
                    hd = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                    if (hd) { x.path = mapped_rccl(); if (x.path.empty()) x.path = name; break; }
                }
            }
            if (hd) {
                x.all_reduce = reinterpret_cast<AllReduce>(dlsym(hd, "ncclAllReduce"));
                x.all_gather = reinterpret_cast<AllGather>(dlsym(hd, "ncclAllGather"));
                x.err_str = reinterpret_cast<ErrStr>(dlsym(hd, "ncclGetErrorString"));
            }
            return x;
        }();
        return r;
    }
};
const int RCCL_INT32 = 2, RCCL_UINT64 = 5, RCCL_SUM = 0;      // ncclDataType_t / ncclRedOp_t (rccl.h)

__global__ void k_pack_phase(const int32_t *__restrict__ idx, const int32_t *__restrict__ sc_phase, const int32_t *__restrict__ orig,
                             const int32_t *__restrict__ swap, int n, int4 *__restrict__ out) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) out[k] = make_int4(idx[k], sc_phase[k], orig[k], swap[k]);
}
}  // namespace

extern "C" {

int vpr_rccl_available(void) { return Rccl::get().all_reduce && Rccl::get().all_gather ? 1 : 0; }
const char *vpr_rccl_library(void) { return Rccl::get().path.c_str(); }

static int pr_counts_impl(vpr_handle *h, void *comm, const uint8_t *const var_class[VPR_HAPS], const int32_t *pb_phase,
                          int32_t min_qual, int32_t max_qual, int64_t *counts) {
    if (!h || !counts || max_qual < min_qual) return VPR_ERR_ARG;
    if (comm && !Rccl::get().all_reduce) return fail(h, VPR_ERR_STATE, "no RCCL in this process (librccl.so.1 not found)");
    if (!h->executed) return fail(h, VPR_ERR_STATE, "vpr_pr_counts before vpr_execute");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int nq = max_qual - min_qual + 1;
    const size_t nh = size_t(2) * 3 * 3 * size_t(nq + 1);
    int rc;
    if (nh > h->hist_cap) {
        if ((rc = dev_alloc(h, &h->d_hist, nh))) return rc;
        h->hist_cap = nh;
    }
    unsigned long long *d_hist = h->d_hist;
    int32_t *d_pb = nullptr;
    HIPCHK(h, hipMemsetAsync(d_hist, 0, nh * 8, h->stream));
    if (pb_phase && h->n_sc) {
        if (!h->d_pb && (rc = dev_alloc(h, &h->d_pb, size_t(h->n_sc)))) return rc;
        d_pb = h->d_pb;
        HIPCHK(h, hipMemcpyAsync(d_pb, pb_phase, size_t(h->n_sc) * 4, hipMemcpyHostToDevice, h->stream));
    }
    if (var_class) { int rc = vpr_upload_var_class(h, var_class); if (rc) return rc; }
    for (int s = 0; s < VPR_HAPS; s++) {
        const int64_t nv = h->n_var[s];
        if (!nv) continue;
        if (!h->d_cls[s]) return fail(h, VPR_ERR_STATE, "vpr_pr_counts: no variant classes (pass var_class or call vpr_upload_var_class)");
        hipLaunchKernelGGL(k_pr_hist, dim3(unsigned((nv + 255) / 256)), dim3(256), size_t(9) * (nq + 1) * 4, h->stream,
                           h->dB.var_off[s], h->n_sc, nv, h->d_cls[s], h->dR.sc_phase, d_pb, h->dR.v[s][0], h->dR.v[s][1],
                           s >> 1, min_qual, max_qual, d_hist);
    }
    if (comm) {     // the one collective of the path: the histogram words summed over the ranks, in place on the device
        const int e = Rccl::get().all_reduce(d_hist, d_hist, nh, RCCL_UINT64, RCCL_SUM, comm, h->stream);
        if (e) return fail(h, VPR_ERR_DEVICE, "ncclAllReduce failed: %s", Rccl::get().err_str ? Rccl::get().err_str(e) : "?");
    }
    std::vector<unsigned long long> hist(nh);
    HIPCHK(h, hipMemcpyAsync(hist.data(), d_hist, nh * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, x_sync(h, h->stream, SITE));
    // counts at threshold k: variants whose last threshold index is >= k (print.cpp:378-381, 425-428); a truth variant
    // additionally counts as FN at every threshold above its own (print.cpp:429-432)
    std::fill(counts, counts + size_t(2) * VPR_VARTYPES * 3 * size_t(nq), 0);
    auto C = [&](int cs, int t, int e, int k) -> int64_t & { return counts[((size_t(cs) * VPR_VARTYPES + t) * 3 + e) * nq + k]; };
    for (int cs = 0; cs < 2; cs++)
        for (int t = 0; t < 3; t++) {
            for (int e = 0; e < 3; e++) {
                int64_t acc = 0;
                for (int k = nq - 1; k >= 0; k--) {
                    acc += int64_t(hist[((size_t(cs) * 3 + t) * 3 + e) * (nq + 1) + k]);
                    C(cs, t, e, k) += acc;
                    C(cs, VPR_VARTYPE_ALL, e, k) += acc;
                }
            }
            if (cs == 1) {
                int64_t below = 0;   // truth variants (any errtype) whose own threshold index is < k
                for (int e = 0; e < 3; e++) below += int64_t(hist[((size_t(cs) * 3 + t) * 3 + e) * (nq + 1) + nq]);
                for (int k = 0; k < nq; k++) {
                    C(cs, t, VPR_ERRTYPE_FN, k) += below;
                    C(cs, VPR_VARTYPE_ALL, VPR_ERRTYPE_FN, k) += below;
                    for (int e = 0; e < 3; e++) below += int64_t(hist[((size_t(cs) * 3 + t) * 3 + e) * (nq + 1) + k]);
                }
            }
        }
    return VPR_OK;
}

int vpr_pr_counts(vpr_handle *h, const uint8_t *const var_class[VPR_HAPS], const int32_t *pb_phase,
                  int32_t min_qual, int32_t max_qual, int64_t *counts) {
    return pr_counts_impl(h, nullptr, var_class, pb_phase, min_qual, max_qual, counts);
}

int vpr_allreduce_counts(vpr_handle *h, void *nccl_comm, const uint8_t *const var_class[VPR_HAPS], const int32_t *pb_phase,
                         int32_t min_qual, int32_t max_qual, int64_t *counts) {
    if (!nccl_comm) return VPR_ERR_ARG;
    return pr_counts_impl(h, nccl_comm, var_class, pb_phase, min_qual, max_qual, counts);
}

int vpr_allgather_phase(vpr_handle *h, void *nccl_comm, int32_t n_ranks, const int32_t *sc_index, int32_t n_total,
                        int32_t *sc_phase, int32_t *orig_phase_dist, int32_t *swap_phase_dist) {
    if (!h || !nccl_comm || n_ranks < 1 || !sc_index || !sc_phase || !orig_phase_dist || !swap_phase_dist) return VPR_ERR_ARG;
    if (!h->executed) return fail(h, VPR_ERR_STATE, "vpr_allgather_phase before vpr_execute");
    const Rccl &R = Rccl::get();
    if (!R.all_gather) return fail(h, VPR_ERR_STATE, "no RCCL in this process (librccl.so.1 not found)");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int n = h->n_sc;
    int rc;
    // 1. how many superclusters every rank holds; 2. their {global index, sc_phase, orig, swap} records, padded to the largest share
    int32_t *d_cnt = nullptr;
    void *q = nullptr;
    if ((rc = exec_alloc(h, &q, size_t(n_ranks + 1) * 4))) return rc;
    d_cnt = static_cast<int32_t *>(q);
    HIPCHK(h, hipMemcpyAsync(d_cnt + n_ranks, &n, 4, hipMemcpyHostToDevice, h->stream));
    int e = R.all_gather(d_cnt + n_ranks, d_cnt, 1, RCCL_INT32, nccl_comm, h->stream);
    if (e) return fail(h, VPR_ERR_DEVICE, "ncclAllGather failed: %s", R.err_str ? R.err_str(e) : "?");
    std::vector<int32_t> cnt(size_t(n_ranks), 0);
    HIPCHK(h, hipMemcpyAsync(cnt.data(), d_cnt, size_t(n_ranks) * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, x_sync(h, h->stream, SITE));
    int64_t m = 1, tot = 0;
    for (int32_t c : cnt) { m = std::max<int64_t>(m, c); tot += c; }
    if (tot > n_total) return fail(h, VPR_ERR_ARG, "vpr_allgather_phase: the ranks hold %lld superclusters, n_total is %d", (long long)tot, n_total);
    int4 *d_send = nullptr, *d_recv = nullptr;
    int32_t *d_idx = nullptr;
    if ((rc = exec_alloc(h, &q, size_t(m) * 16))) return rc;
    d_send = static_cast<int4 *>(q);
    if ((rc = exec_alloc(h, &q, size_t(m) * 16 * size_t(n_ranks)))) return rc;
    d_recv = static_cast<int4 *>(q);
    if ((rc = exec_alloc(h, &q, size_t(std::max(n, 1)) * 4))) return rc;
    d_idx = static_cast<int32_t *>(q);
    HIPCHK(h, hipMemsetAsync(d_send, 0, size_t(m) * 16, h->stream));
    if (n) {
        HIPCHK(h, hipMemcpyAsync(d_idx, sc_index, size_t(n) * 4, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(k_pack_phase, dim3(unsigned((n + 255) / 256)), dim3(256), 0, h->stream, d_idx, h->dR.sc_phase,
                           h->dR.orig_phase_dist, h->dR.swap_phase_dist, n, d_send);
    }
    e = R.all_gather(d_send, d_recv, size_t(m) * 4, RCCL_INT32, nccl_comm, h->stream);
    if (e) return fail(h, VPR_ERR_DEVICE, "ncclAllGather failed: %s", R.err_str ? R.err_str(e) : "?");
    std::vector<int4> recv(size_t(m) * size_t(n_ranks));
    HIPCHK(h, hipMemcpyAsync(recv.data(), d_recv, recv.size() * 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, x_sync(h, h->stream, SITE));
    for (int r = 0; r < n_ranks; r++)
        for (int32_t k = 0; k < cnt[size_t(r)]; k++) {
            const int4 v = recv[size_t(r) * size_t(m) + size_t(k)];
            if (v.x < 0 || v.x >= n_total) return fail(h, VPR_ERR_ARG, "vpr_allgather_phase: supercluster index %d of rank %d out of range", v.x, r);
            sc_phase[v.x] = v.y; orig_phase_dist[v.x] = v.z; swap_phase_dist[v.x] = v.w;
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
    // the walk scratch lives in a workspace that is reused per chunk: only the last chunk of round 0 and the
    // retry plans are still resident (the most recent plan of an alignment holds its final walk)
    // (an alignment that is a copy of another one of its supercluster, k_hap_alias: that one's walk)
    const int src_ = h->alias.empty() ? -1 : alias_source(h->alias[size_t(sc)], aln);
    const int32_t a = sc * 4 + (src_ >= 0 ? src_ : aln);
    const uint8_t *arena = nullptr;
    bool found = false;
    for (auto it = h->resident.rbegin(); it != h->resident.rend() && !found; ++it)
        if (std::find(it->first.begin(), it->first.end(), a) != it->first.end()) { arena = it->second; found = true; }
    if (!found && h->res0_cnt > 0) {
        const int64_t pos = h->plan0_pos[size_t(a)];
        if (pos >= 0 && pos >= h->res0_off && pos < h->res0_off + h->res0_cnt) arena = h->plan0.arena;
    }
    if (!arena) return VPR_ERR_STATE;
    AlnOut O;
    AlnDesc d;
    if (hipMemcpy(&O, h->d_outs + a, sizeof(O), hipMemcpyDeviceToHost) != hipSuccess) return VPR_ERR_DEVICE;
    if (hipMemcpy(&d, h->d_descs + a, sizeof(d), hipMemcpyDeviceToHost) != hipSuccess) return VPR_ERR_DEVICE;
    const int64_t n = std::min<int64_t>(O.path_len, cap);
    std::vector<PathEnt> p(n);
    if (n && hipMemcpy(p.data(), reinterpret_cast<const PathEnt *>(arena) + d.path_off, n * sizeof(PathEnt),
                       hipMemcpyDeviceToHost) != hipSuccess)
        return VPR_ERR_DEVICE;
    for (int64_t k = 0; k < n; k++) {
        plane[k] = uint8_t(p[k].a >> 31);
        qri[k] = int32_t(p[k].a & 0x7fffffffu);
        ti[k] = int32_t(p[k].b & 0x0fffffffu);       // (bits 28, 29: credit_walk's base-equality bits)
        sync[k] = uint8_t(p[k].b >> 31);
        edit[k] = uint8_t((p[k].b >> 30) & 1);
    }
    return O.path_len;
}

}  // extern "C"

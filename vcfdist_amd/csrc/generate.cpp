// generate.cpp -- host marshalling: variants + reference -> flat Level A batch.
//
// Produces, for every supercluster and hap slot, what the reference's
// generate_ptrs_strs (src/dist.cpp:145-242) produces into std::string /
// vector<vector<int>> locals -- hap string, reference string, hap->ref and
// ref->hap index + flag arrays -- but as one pass that sizes every output and a
// second, thread-parallel pass that fills the flat CSR arrays in place.
//
// Semantics kept from the reference:
//   * region [beg,end] inclusive; INS sits before `pos` (rlen 0); alleles carry no anchor base
//   * matching run: ptr = index of the same base in the other string, flag 0
//   * SUB: both sides flag VARIANT|VAR_BEG|VAR_END
//   * INS of k: k hap entries pointing at the ref base *before* the insertion
//     (ref_len-1), all VARIANT, first |= VAR_BEG|INS_LOC, last |= VAR_END
//   * DEL of k: k ref entries pointing at the hap base before the deletion, all
//     VARIANT, first |= VAR_BEG, last |= VAR_END
// The reference string is taken from query hap 1's pass (the driver hands ref_q1
// to every alignment, dist.cpp:1856,1868); all four passes cover the same
// reference span.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

// host threads for the marshalling passes: three quarters of what the process may use -- the cgroup's CPU quota where
// there is one (a container that sees 256 cores may be allowed 16; more runnable threads than that get the group suspended)
static int host_threads() {
    static const int n = [] {
        unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        long long quota = -1, period = 100000;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(g, "%lld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *p = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(p, "%lld", &period) != 1) period = 100000; fclose(p); }
        }
        if (quota > 0 && period > 0) hw = std::min<unsigned>(hw, unsigned(std::max<long long>(1, quota / period)));
        if (const char *lw = getenv("LOCAL_WORLD_SIZE")) { const int n = atoi(lw); if (n > 1) hw = std::max(1u, hw / unsigned(n)); }   // (ranks of a node share it)
        return int(std::min<unsigned>(std::max(1u, hw - std::max(1u, hw / 4)), 32));
    }();
    return n;
}


#include "../../include/vcfdist_pr.h"

// The arrays of a marshalled batch live in two page-locked staging blocks (vpr_host_alloc: vpr_upload's copies then run
// as DMA at the link rate, beside the host's planning, instead of through the driver's bounce buffers): `small` holds the
// offsets and the variant tables, whose sizes are known up front, `big` the strings, pointers and flags sized by pass 1.
// Freed blocks are kept (up to four) and reused by the next batch: page-locking costs about as much as filling.
struct vpr_owned_batch {
    int64_t *hap_off[VPR_HAPS], *ref_off, *var_off[VPR_HAPS];
    uint8_t *hap_seq[VPR_HAPS], *hap_flag[VPR_HAPS], *ref_seq, *ref_flag[2];
    int32_t *hap_ptr[VPR_HAPS], *ref_ptr[2], *var_pos[VPR_HAPS];
    float *var_qual[VPR_HAPS];
    struct Block { void *p = nullptr; size_t bytes = 0; bool pinned = false; } small, big;
    vpr_batch view;
};

namespace {

using Block = vpr_owned_batch::Block;
std::mutex g_cache_mu;
Block g_cache[4];

Block block_get(size_t bytes) {
    // (sizes in steps of an eighth: batches of one stream differ by a few per cent, and two threads marshalling batches of
    // slightly different sizes must not take each other's block -- the smaller request grabbing the larger block left the
    // larger request to page-lock a new one, 15 ms, and to free the displaced one, 8 ms, every batch)
    size_t step = size_t(1) << 20;
    while (step * 8 < bytes) step <<= 1;
    bytes = (std::max<size_t>(bytes, 1) + step - 1) / step * step;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        Block *best = nullptr;
        for (Block &c : g_cache)
            if (c.p && c.bytes >= bytes && (!best || c.bytes < best->bytes)) best = &c;
        if (best) { Block b = *best; *best = Block(); return b; }
    }
    Block b;
    b.bytes = bytes ? bytes : 1;
    static std::atomic<bool> no_device{false};      // (asking the runtime again after a failure costs ~10 ms per call)
    if (!no_device) b.p = vpr_host_alloc(b.bytes);  // nullptr without a HIP device: plain memory (marshalling alone needs no GPU)
    b.pinned = b.p != nullptr;
    if (!b.p) no_device = true;
    if (!b.p) b.p = malloc(b.bytes);
    return b;
}

void block_release(Block &b) {
    if (b.p) { if (b.pinned) vpr_host_free(b.p); else free(b.p); }
    b = Block();
}

void block_put(Block &b) {
    if (!b.p) return;
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        Block *slot = nullptr;
        for (Block &c : g_cache) if (!c.p) { slot = &c; break; }
        if (!slot) {
            slot = &g_cache[0];
            for (Block &c : g_cache) if (c.bytes < slot->bytes) slot = &c;       // the smallest kept block makes room for a larger one
            if (slot->bytes >= b.bytes) slot = nullptr;
        }
        if (slot) { std::swap(*slot, b); }
    }
    block_release(b);       // whatever was not kept (the displaced block, or this one)
}

// carve `n` elements of T (64-byte aligned) from a block
struct Carver {
    uint8_t *base; size_t off = 0;
    template <typename T> T *take(size_t n) { T *p = base ? reinterpret_cast<T *>(base + off) : nullptr; off += (n * sizeof(T) + 63) & ~size_t(63); return p; }
};

struct Lens { int64_t hap, ref; int err; };

// Walk one (supercluster, hap slot).  With FILL=false only lengths are computed.
template <bool FILL>
Lens walk(const vpr_variants *v, int slot, int sc,
          uint8_t *hseq, int32_t *hptr, uint8_t *hflag,
          uint8_t *rseq, int32_t *rptr, uint8_t *rflag) {
    const int ctg = v->sc_ctg[sc];
    const uint8_t *fa = v->ctg_seq + v->ctg_off[ctg];
    const int64_t ctg_len = v->ctg_off[ctg + 1] - v->ctg_off[ctg];
    // Region end: get_supercluster_range (cluster.cpp:591-595) yields pos + rlen + 1 of the last variant, which lies
    // behind the contig when that variant ends on one of its last two bases.  The reference has no defined result there
    // (its substr, dist.cpp:232, silently returns fewer bases than the pointers appended beside it, and
    // calc_prec_recall_path, dist.cpp:539-546, then indexes its matrices by the pointer arrays' size); here the region is
    // cut at the contig's last base, strings and pointer arrays consistent -- see include/vcfdist_pr.h.  A region that
    // starts in front of the contig (a variant at position 0: beg = -1) is an error in the reference as well
    // (substr(-1) throws, dist.cpp:236-238).
    const int32_t beg = v->sc_beg[sc];
    const int32_t end = int32_t(std::min<int64_t>(v->sc_end[sc], ctg_len - 1));
    int64_t var = v->var_off[slot][sc];
    const int64_t var_end = v->var_off[slot][sc + 1];
    int64_t nh = 0, nr = 0;
    int32_t pos = beg;
    if (beg < 0 || beg > end) return {0, 0, VPR_ERR_ARG};
    while (pos <= end) {
        if (var < var_end && v->var_pos[slot][var] == pos) {
            const uint8_t *pool = v->allele_pool[slot];
            const int64_t r0 = v->var_ref_off[slot][var], rl = v->var_ref_len[slot][var];
            const int64_t a0 = v->var_alt_off[slot][var], al = v->var_alt_len[slot][var];
            const uint8_t type = v->var_type[slot][var];
            if (type == VPR_TYPE_INS) {
                if (FILL) {
                    for (int64_t k = 0; k < al; k++) {
                        hseq[nh + k] = pool[a0 + k];
                        hptr[nh + k] = int32_t(nr) - 1;
                        hflag[nh + k] = VPR_PTR_VARIANT;
                    }
                    if (al > 0) {
                        hflag[nh + al - 1] |= VPR_PTR_VAR_END;
                        hflag[nh] |= VPR_PTR_VAR_BEG | VPR_PTR_INS_LOC;
                    }
                }
                nh += al;
            } else if (type == VPR_TYPE_DEL) {
                if (FILL && rseq) {
                    for (int64_t k = 0; k < rl; k++) {
                        rseq[nr + k] = pool[r0 + k];
                    }
                }
                if (FILL && rptr) {
                    for (int64_t k = 0; k < rl; k++) {
                        rptr[nr + k] = int32_t(nh) - 1;
                        rflag[nr + k] = VPR_PTR_VARIANT;
                    }
                    if (rl > 0) {
                        rflag[nr + rl - 1] |= VPR_PTR_VAR_END;
                        rflag[nr] |= VPR_PTR_VAR_BEG;
                    }
                }
                nr += rl;
                pos += int32_t(rl);
            } else if (type == VPR_TYPE_SUB) {
                if (FILL) {
                    for (int64_t k = 0; k < al; k++) hseq[nh + k] = pool[a0 + k];
                    hptr[nh] = int32_t(nr);
                    hflag[nh] = VPR_PTR_VARIANT | VPR_PTR_VAR_BEG | VPR_PTR_VAR_END;
                    if (rseq) for (int64_t k = 0; k < rl; k++) rseq[nr + k] = pool[r0 + k];
                    if (rptr) {
                        rptr[nr] = int32_t(nh);
                        rflag[nr] = VPR_PTR_VARIANT | VPR_PTR_VAR_BEG | VPR_PTR_VAR_END;
                    }
                }
                if (al != 1 || rl != 1) return {0, 0, VPR_ERR_ARG};
                nh += 1;
                nr += 1;
                pos += 1;
            } else {
                return {0, 0, VPR_ERR_ARG};  // only SUB/INS/DEL reach this path (dist.cpp:199-201)
            }
            var++;
        } else {
            const int32_t stop = (var < var_end) ? v->var_pos[slot][var] : end + 1;
            if (stop <= pos) return {0, 0, VPR_ERR_ARG};  // overlapping / unsorted variants
            const int64_t n = stop - pos;
            if (FILL) {
                for (int64_t k = 0; k < n; k++) {
                    const uint8_t b = fa[pos + k];
                    hseq[nh + k] = b;
                    hptr[nh + k] = int32_t(nr + k);
                    hflag[nh + k] = 0;
                    if (rseq) rseq[nr + k] = b;
                    if (rptr) {
                        rptr[nr + k] = int32_t(nh + k);
                        rflag[nr + k] = 0;
                    }
                }
            }
            nh += n;
            nr += n;
            pos = stop;
        }
    }
    if (pos != end + 1 || var != var_end) return {0, 0, VPR_ERR_ARG};   // a variant that leaves the region / the contig
    return {nh, nr, 0};
}

}  // namespace

extern "C" {

// fill = false: offsets, lengths and the variant tables only (every check of the full pass included); the strings, pointer
// and flag arrays stay null -- vpr_upload_variants has the device write them (k_generate, pr_gen.hip)
static int batch_from_variants_impl(const vpr_variants *v, vpr_owned_batch **out, bool fill) {
    if (!v || !out || v->n_sc < 0) return VPR_ERR_ARG;
    vpr_owned_batch *B = new (std::nothrow) vpr_owned_batch();
    if (!B) return VPR_ERR_NOMEM;
    const int n = v->n_sc;
    auto carve_small = [&](Carver &c) {
        for (int h = 0; h < VPR_HAPS; h++) {
            B->hap_off[h] = c.take<int64_t>(size_t(n) + 1);
            B->var_off[h] = c.take<int64_t>(size_t(n) + 1);
            B->var_pos[h] = c.take<int32_t>(size_t(v->var_off[h][n]));
            B->var_qual[h] = c.take<float>(size_t(v->var_off[h][n]));
        }
        B->ref_off = c.take<int64_t>(size_t(n) + 1);
    };
    {
        Carver measure{nullptr};
        carve_small(measure);
        B->small = block_get(measure.off);
        if (!B->small.p) { delete B; return VPR_ERR_NOMEM; }
        Carver c{static_cast<uint8_t *>(B->small.p)};
        carve_small(c);
    }
    for (int h = 0; h < VPR_HAPS; h++) std::fill(B->hap_off[h], B->hap_off[h] + n + 1, int64_t(0));
    std::fill(B->ref_off, B->ref_off + n + 1, int64_t(0));

    const int nthreads = host_threads();
    std::atomic<int> err{0};

    // pass 1: sizes
    auto size_job = [&](int t) {
        for (int sc = int(int64_t(n) * t / nthreads), e = int(int64_t(n) * (t + 1) / nthreads); sc < e; sc++) {
            for (int h = 0; h < VPR_HAPS; h++) {
                Lens L = walk<false>(v, h, sc, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
                if (L.err) { err = L.err; return; }
                B->hap_off[h][sc + 1] = L.hap;
                if (h == 0) B->ref_off[sc + 1] = L.ref;
                else if (L.ref != B->ref_off[sc + 1]) { err = VPR_ERR_ARG; return; }
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(size_job, t);
        for (auto &x : th) x.join();
    }
    if (err) { vpr_owned_batch_free(B); return err; }
    {   // lengths -> offsets: every thread sums its slice, the slices' bases are a short serial pass, every thread then runs its
        // prefix sums from its base (five serial passes over a million superclusters were most of the sizing pass)
        std::vector<int64_t> base(size_t(nthreads + 1) * 5, 0);
        auto slice = [&](int t, int &sc0, int &sc1) { sc0 = int(int64_t(n) * t / nthreads); sc1 = int(int64_t(n) * (t + 1) / nthreads); };
        auto arr = [&](int k) -> int64_t * { return k < VPR_HAPS ? B->hap_off[k] : B->ref_off; };
        auto sum_job = [&](int t) {
            int sc0, sc1; slice(t, sc0, sc1);
            for (int k = 0; k < 5; k++) { int64_t s_ = 0; const int64_t *a = arr(k); for (int sc = sc0; sc < sc1; sc++) s_ += a[sc + 1]; base[size_t(t + 1) * 5 + k] = s_; }
        };
        auto scan_job = [&](int t) {
            int sc0, sc1; slice(t, sc0, sc1);
            for (int k = 0; k < 5; k++) { int64_t run = base[size_t(t) * 5 + k]; int64_t *a = arr(k); for (int sc = sc0; sc < sc1; sc++) { run += a[sc + 1]; a[sc + 1] = run; } }
        };
        { std::vector<std::thread> th; for (int t = 0; t < nthreads; t++) th.emplace_back(sum_job, t); for (auto &x : th) x.join(); }
        for (int t = 1; t <= nthreads; t++) for (int k = 0; k < 5; k++) base[size_t(t) * 5 + k] += base[size_t(t - 1) * 5 + k];
        { std::vector<std::thread> th; for (int t = 0; t < nthreads; t++) th.emplace_back(scan_job, t); for (auto &x : th) x.join(); }
    }

    auto carve_big = [&](Carver &c) {
        for (int h = 0; h < VPR_HAPS; h++) {
            const size_t m = size_t(B->hap_off[h][n]);
            B->hap_seq[h] = c.take<uint8_t>(m);
            B->hap_ptr[h] = c.take<int32_t>(m);
            B->hap_flag[h] = c.take<uint8_t>(m);
        }
        B->ref_seq = c.take<uint8_t>(size_t(B->ref_off[n]));
        for (int h = 0; h < 2; h++) {
            B->ref_ptr[h] = c.take<int32_t>(size_t(B->ref_off[n]));
            B->ref_flag[h] = c.take<uint8_t>(size_t(B->ref_off[n]));
        }
    };
    if (fill) {
        Carver measure{nullptr};
        carve_big(measure);
        B->big = block_get(measure.off);
        if (!B->big.p) { vpr_owned_batch_free(B); return VPR_ERR_NOMEM; }
        Carver c{static_cast<uint8_t *>(B->big.p)};
        carve_big(c);
    } else {
        Carver none{nullptr};
        carve_big(none);        // (null pointers)
    }

    // pass 2: fill
    auto fill_job = [&](int t) {
        for (int sc = int(int64_t(n) * t / nthreads), e = int(int64_t(n) * (t + 1) / nthreads); sc < e; sc++) {
            for (int h = 0; h < VPR_HAPS; h++) {
                const int64_t ho = B->hap_off[h][sc], ro = B->ref_off[sc];
                walk<true>(v, h, sc, B->hap_seq[h] + ho, B->hap_ptr[h] + ho, B->hap_flag[h] + ho,
                           h == 0 ? B->ref_seq + ro : nullptr, h < 2 ? B->ref_ptr[h] + ro : nullptr,
                           h < 2 ? B->ref_flag[h] + ro : nullptr);
            }
        }
    };
    if (fill) {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(fill_job, t);
        for (auto &x : th) x.join();
    }

    // variants: positions relative to the supercluster start (dist.cpp:1075-1080)
    auto var_job = [&](int t) {
        const int sc0 = int(int64_t(n) * t / nthreads), sc1 = int(int64_t(n) * (t + 1) / nthreads);
        for (int h = 0; h < VPR_HAPS; h++) {
            std::copy(v->var_off[h] + sc0, v->var_off[h] + sc1 + (t == nthreads - 1 ? 1 : 0), B->var_off[h] + sc0);
            const int64_t k0 = v->var_off[h][sc0], k1 = v->var_off[h][sc1];
            std::copy(v->var_qual[h] + k0, v->var_qual[h] + k1, B->var_qual[h] + k0);
            for (int sc = sc0; sc < sc1; sc++)
                for (int64_t k = v->var_off[h][sc]; k < v->var_off[h][sc + 1]; k++)
                    B->var_pos[h][k] = v->var_pos[h][k] - v->sc_beg[sc];
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; t++) th.emplace_back(var_job, t);
        for (auto &x : th) x.join();
        for (int h = 0; h < VPR_HAPS; h++) B->var_off[h][n] = v->var_off[h][n];
    }

    vpr_batch &b = B->view;
    memset(&b, 0, sizeof(b));
    b.n_sc = n;
    for (int h = 0; h < VPR_HAPS; h++) {
        b.hap_off[h] = B->hap_off[h];
        b.hap_seq[h] = B->hap_seq[h];
        b.hap_ptr[h] = B->hap_ptr[h];
        b.hap_flag[h] = B->hap_flag[h];
        b.var_off[h] = B->var_off[h];
        b.var_pos[h] = B->var_pos[h];
        b.var_qual[h] = B->var_qual[h];
    }
    b.ref_off = B->ref_off;
    b.ref_seq = B->ref_seq;
    for (int h = 0; h < 2; h++) {
        b.ref_ptr[h] = B->ref_ptr[h];
        b.ref_flag[h] = B->ref_flag[h];
    }
    *out = B;
    return VPR_OK;
}

int vpr_batch_from_variants(const vpr_variants *v, vpr_owned_batch **out) { return batch_from_variants_impl(v, out, true); }
// (library-internal: the skeleton vpr_upload_variants hands to the device generator)
int vpr_batch_skeleton_from_variants(const vpr_variants *v, vpr_owned_batch **out) { return batch_from_variants_impl(v, out, false); }

const vpr_batch *vpr_owned_batch_view(const vpr_owned_batch *b) { return b ? &b->view : nullptr; }
void vpr_owned_batch_free(vpr_owned_batch *b) {
    if (!b) return;
    block_put(b->small);
    block_put(b->big);
    delete b;
}

}  // extern "C"

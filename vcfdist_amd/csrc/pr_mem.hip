// pr_mem.hip -- the library's memory: every allocator call and blocking wait goes through the counted wrappers here (vpr_timing
// reports them per execute); the process-wide books of device memory that correct a lagging hipMemGetInfo for the memory plan
// (vpr_upload, pr_api.hip); device and page-locked blocks with the lifetime of a batch (dev_alloc, pin_alloc: kept and handed out
// again when the next batch is uploaded) or of one execute (exec_alloc, exec_pin); error reporting (fail).
#include "pr_host.h"

namespace {
std::string g_create_err;
}  // namespace
const char *vpr_create_error() { return g_create_err.c_str(); }

int fail(vpr_handle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_err = buf;
    return code;
}


inline void slow_call(vpr_handle *h, const char *what, const char *site, size_t bytes, double dt) {
    if (h && h->stall_log && dt > 5.0) fprintf(stderr, "[vpr] slow host call: %s (%zu bytes) at %s: %.1f ms\n", what, bytes, site, dt);
}
// every allocator call and blocking wait of the library goes through these: counted and timed per execute
// Switches of experiments that are closed (DESIGN.md section 6 has each one's measurement: stream priorities and padding, the zero
// level's occupancy cap, the credit walk's head, the replay's wide jobs, the shares of the memory plan ...): read only by a library
// built with -DVPR_EXPERIMENTS (make CXXFLAGS+=-DVPR_EXPERIMENTS); the shipped build runs the settings those experiments chose.
const char *exp_getenv(const char *name) {
#ifdef VPR_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
int poison_byte() {
    static const int v = [] { const char *e = getenv("VPR_POISON"); return e ? int(strtol(e, nullptr, 0)) & 0xff : -1; }();
    return v;
}
// Device memory the library leaves alone (VPR_DEV_RESERVE_MB, default 1 536): the runtime allocates on its own behind the
// library's back -- private (scratch) memory of kernels per hardware queue, hundreds of MB each for the walk and replay kernels --
// and a queue that cannot get it is ABORTED (HSA_STATUS_ERROR_OUT_OF_RESOURCES, the process dies: 125 000 stress superclusters
// with the long part from 2 048 rows ended that way, 38 MB free).  An allocation that would go below the reserve fails like an
// exhausted device instead, which every caller handles (smaller workspaces and more rounds, sub-batches of replays, VPR_ERR_NOMEM).
// (The default keeps what fitted before fitting: a GPU's share of the stress workload, 125 000 superclusters, ends with 2.4 GB free.
// Its upload takes 278 of 309 GB -- round 0's workspace 60 % of the free memory, four ladder workspaces an eighth of the rest each --
// and a single tied 16 k x 16 k alignment then wants 6.9 GB of replay stamps: the shares want planning from the batch)
int64_t dev_reserve_bytes() {
    static const int64_t v = [] { const char *e = getenv("VPR_DEV_RESERVE_MB"); return (e ? int64_t(atoll(e)) : int64_t(1536)) << 20; }();
    return v;
}
// The batch's memory plan: the share of the DEVICE the library leaves free (VPR_DEV_FREE_SHARE), and the share of the rest
// round 0's workspace may take (VPR_ARENA_SHARE; it takes what its plan asks for when that is less).  What remains is the
// ladders' and the replays' (vpr_upload).  VPR_LADDER_SHARE scales the ladders' first workspaces (diagnostic).
double free_share() { static const double v = [] { const char *e = getenv("VPR_DEV_FREE_SHARE"); return e ? atof(e) : 0.11; }(); return v; }
double arena_share() { static const double v = [] { const char *e = exp_getenv("VPR_ARENA_SHARE"); return e ? atof(e) : 0.7; }(); return v; }
double ladder_share() { static const double v = [] { const char *e = exp_getenv("VPR_LADDER_SHARE"); return e ? atof(e) : 1.0; }(); return v; }
// The process's own books of device memory (all handles): what the library holds, and what it has handed back in the last
// seconds.  hipMemGetInfo lags behind large hipFree calls -- a handle that has just released 250 GB of kept blocks was told
// "51 GB free" 15 ms later and planned its next batch for a device a third the size --, so the
// memory plan adds what the books say was freed recently, up to what the books say can be free at all (books_free).
struct DevBooks {
    std::mutex mu;
    std::unordered_map<void *, size_t> blocks;
    int64_t live = 0;                                   // bytes of hipMalloc'ed blocks of this process's handles
    int64_t foreign0 = -1;                              // what others held when the first plan was made (torch, other processes)
    std::deque<std::pair<double, int64_t>> freed;       // (time, bytes) of the frees of the last seconds
    void add(void *q, size_t b) { std::lock_guard<std::mutex> g(mu); blocks[q] = b; live += int64_t(b); }
    bool sub(void *q) {
        std::lock_guard<std::mutex> g(mu);
        auto it = blocks.find(q);
        if (it == blocks.end()) return false;
        live -= int64_t(it->second);
        freed.emplace_back(wall_ms(), int64_t(it->second));
        blocks.erase(it);
        return true;
    }
    // free bytes the plan may count on, given the driver's figure
    int64_t books_free(int64_t reported_free, int64_t total, double window_ms = 20000.0) {
        std::lock_guard<std::mutex> g(mu);
        const double now = wall_ms();
        while (!freed.empty() && now - freed.front().first > window_ms) freed.pop_front();
        int64_t recent = 0;
        for (const auto &f : freed) recent += f.second;
        if (foreign0 < 0) foreign0 = std::max<int64_t>(total - reported_free - live - recent, 0);
        const int64_t by_books = total - live - foreign0;
        return std::max(reported_free, std::min(reported_free + recent, by_books));
    }
};
static DevBooks &dev_books(int dev = -1) {      // one set of books per device (-1: the calling thread's current device)
    static DevBooks b[16];
    if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = 0;
    return b[dev & 15];
}

hipError_t x_malloc(vpr_handle *h, void **q, size_t bytes, const char *site) {
    const double t = wall_ms();
    {
        size_t fr = 0, tt = 0;
        // optional growth during an execute (a ladder's larger workspace, a replay scratch that would hold a whole launch instead of
        // sub-batches) leaves a 32nd of the device to what an execute MUST still get (the replay stamps of one large tied alignment)
        // (the reserve is about workspaces, arenas and their growth: a small array of an upload -- offsets, counters, lists -- is
        // neither what exhausts the device nor worth a hipMemGetInfo call each; on a device shared with other handles such an
        // array used to fail with less than the reserve free)
        int64_t reserve = bytes >= (size_t(64) << 20) || (h && h->soft_alloc) ? dev_reserve_bytes() : 0;
        if (reserve > 0 && hipMemGetInfo(&fr, &tt) == hipSuccess) {
            if (h && h->soft_alloc) reserve = std::max<int64_t>(reserve, std::max<int64_t>(int64_t(tt) / 32, h->mem_reserve));
        }
        if (reserve > 0 && tt > 0 && int64_t(fr) < int64_t(bytes) + reserve &&
            int64_t(bytes) + reserve < int64_t(tt)) {       // (a device smaller than the reserve: no reserve)
            if (h) h->hs.n_dev_alloc++;
            *q = nullptr;
            return hipErrorOutOfMemory;
        }
    }
    const hipError_t e = hipMalloc(q, bytes);
    const double dt = wall_ms() - t;
    if (e == hipSuccess) dev_books().add(*q, bytes);
    if (h) { h->hs.n_dev_alloc++; h->hs.ms_alloc += dt; }
    slow_call(h, "hipMalloc", site, bytes, dt);
    // debugging aid (VPR_POISON=<byte>): new device memory holds that byte instead of whatever the driver left there, so that
    // a read of something never written shows up the same way in every run
    if (e == hipSuccess && poison_byte() >= 0) { (void)hipMemset(*q, poison_byte(), bytes); (void)hipDeviceSynchronize(); }
    return e;
}
hipError_t x_free(vpr_handle *h, void *q, const char *site) {
    const double t = wall_ms();
    const hipError_t e = hipFree(q);
    const double dt = wall_ms() - t;
    if (!dev_books().sub(q))
        for (int dv = 0; dv < 16 && !dev_books(dv).sub(q); dv++) {}
    if (h) { h->hs.n_dev_free++; h->hs.ms_alloc += dt; }
    slow_call(h, "hipFree", site, 0, dt);
    return e;
}
hipError_t x_host_malloc(vpr_handle *h, void **q, size_t bytes, const char *site) {
    const double t = wall_ms();
    const hipError_t e = hipHostMalloc(q, bytes, hipHostMallocDefault);
    const double dt = wall_ms() - t;
    if (h) { h->hs.n_pin_alloc++; h->hs.ms_alloc += dt; }
    slow_call(h, "hipHostMalloc", site, bytes, dt);
    if (e == hipSuccess && poison_byte() >= 0) memset(*q, poison_byte(), bytes);
    return e;
}
hipError_t x_sync(vpr_handle *h, hipStream_t s_, const char *site) {
    const double t = wall_ms();
    const hipError_t e = hipStreamSynchronize(s_);
    const double dt = wall_ms() - t;
    if (h) h->hs.ms_sync += dt;
    slow_call(h, "hipStreamSynchronize", site, 0, dt);
    return e;
}
hipError_t x_event_sync(vpr_handle *h, hipEvent_t ev, const char *site) {
    const double t = wall_ms();
    const hipError_t e = hipEventSynchronize(ev);
    const double dt = wall_ms() - t;
    if (h) h->hs.ms_sync += dt;
    slow_call(h, "hipEventSynchronize", site, 0, dt);
    return e;
}

// a device block of at least `bytes` bytes: a kept one that is not wastefully larger, else a new allocation
void *dev_block(vpr_handle *h, size_t bytes, hipError_t *err) {
    *err = hipSuccess;
    int best = -1;
    for (size_t k = 0; k < h->dev_cache.size(); k++) {
        const size_t b = h->dev_cache[k].bytes;
        if (b >= bytes && b <= bytes + bytes / 2 + (size_t(64) << 20) && (best < 0 || b < h->dev_cache[size_t(best)].bytes)) best = int(k);
    }
    void *q = nullptr;
    size_t got = bytes;
    if (best >= 0) {
        q = h->dev_cache[size_t(best)].p; got = h->dev_cache[size_t(best)].bytes;
        h->dev_cache.erase(h->dev_cache.begin() + best);
        if (poison_byte() >= 0) { (void)hipMemset(q, poison_byte(), got); (void)hipDeviceSynchronize(); }
    } else {
        *err = x_malloc(h, &q, bytes, SITE);
        if (*err != hipSuccess) {        // out of memory with blocks kept aside: release them and try once more
            for (auto &c : h->dev_cache) (void)x_free(h, c.p, SITE);
            h->dev_cache.clear();
            (void)hipGetLastError();
            *err = x_malloc(h, &q, bytes, SITE);
            if (*err != hipSuccess) return nullptr;
        }
    }
    h->allocs.push_back(q);
    h->alloc_bytes.push_back(got);
    return q;
}

// page-locked host memory of the batch's lifetime (h->pinned_blk), from the kept blocks when one fits
int pin_alloc(vpr_handle *h, void **out, size_t bytes) {
    bytes = bytes ? bytes : 1;
    int best = -1;
    for (size_t k = 0; k < h->pin_cache.size(); k++) {
        const size_t b = h->pin_cache[k].bytes;
        if (b >= bytes && b <= 2 * bytes + (size_t(1) << 20) && (best < 0 || b < h->pin_cache[size_t(best)].bytes)) best = int(k);
    }
    if (best >= 0) {
        *out = h->pin_cache[size_t(best)].p;
        if (poison_byte() >= 0) memset(*out, poison_byte(), h->pin_cache[size_t(best)].bytes);
        h->pinned_blk.push_back(h->pin_cache[size_t(best)]);
        h->pin_cache.erase(h->pin_cache.begin() + best);
        return VPR_OK;
    }
    void *q = nullptr;
    const hipError_t e = x_host_malloc(h, &q, bytes, SITE);
    if (e != hipSuccess) return fail(h, VPR_ERR_NOMEM, "hipHostMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    h->pinned_blk.push_back(vpr_handle::Blk{q, bytes});
    *out = q;
    return VPR_OK;
}

int dev_alloc_bytes(vpr_handle *h, void **p, size_t bytes) {
    *p = nullptr;
    if (h->cfg.flags & VPR_CFG_GUARD_ALLOC) {      // debugging aid: every array its own allocation (an access far behind one faults)
        void *q = nullptr;
        hipError_t e = x_malloc(h, &q, bytes, SITE);
        if (e != hipSuccess) return fail(h, VPR_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        h->allocs.push_back(q);
        h->alloc_bytes.push_back(0);     // (0: not kept for reuse)
        *p = q;
        return VPR_OK;
    }
    if (bytes > h->pool_left) {
        // a new block: the request alone when it is large (the remainder of the old block stays usable for nothing: blocks
        // double, so at most half of what was allocated is ever lost), else the next pool size
        const size_t blk = std::max(bytes, h->pool_next);
        hipError_t e;
        void *q = dev_block(h, blk, &e);
        if (!q) return fail(h, VPR_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", blk, hipGetErrorString(e));
        if (bytes >= h->pool_next) {        // dedicated block; keep carving the previous one
            *p = q;
            return VPR_OK;
        }
        h->pool_cur = static_cast<uint8_t *>(q);
        h->pool_left = blk;
        h->pool_next = std::min(h->pool_next * 2, size_t(2) << 30);
    }
    *p = h->pool_cur;
    h->pool_cur += bytes;
    h->pool_left -= bytes;
    return VPR_OK;
}

int exec_alloc(vpr_handle *h, void **out, size_t bytes) {
    bytes = std::max<size_t>((bytes + 255) & ~size_t(255), 256);
    int best = -1;
    for (size_t k = 0; k < h->exec_blks.size(); k++) {
        const auto &b = h->exec_blks[k];
        if (!b.used && b.bytes >= bytes && (best < 0 || b.bytes < h->exec_blks[size_t(best)].bytes)) best = int(k);
    }
    if (best < 0) {
        uint8_t *q = nullptr;
        const int rc = dev_alloc(h, &q, bytes);
        if (rc) return rc;
        h->exec_blks.push_back(vpr_handle::ExecBlk{q, bytes, false});
        best = int(h->exec_blks.size()) - 1;
    }
    h->exec_blks[size_t(best)].used = true;
    *out = h->exec_blks[size_t(best)].p;
    return VPR_OK;
}

// page-locked host memory with the lifetime of one execute (strip planning tables, the selection list of the deferred edit
// distances): blocks of the batch's, handed out again by the next execute
int exec_pin(vpr_handle *h, void **out, size_t bytes) {
    bytes = std::max<size_t>((bytes + 255) & ~size_t(255), 256);
    int best = -1;
    for (size_t k = 0; k < h->exec_pins.size(); k++) {
        const auto &b = h->exec_pins[k];
        if (!b.used && b.bytes >= bytes && (best < 0 || b.bytes < h->exec_pins[size_t(best)].bytes)) best = int(k);
    }
    if (best < 0) {
        void *q = nullptr;
        const int rc = pin_alloc(h, &q, bytes + bytes / 2);
        if (rc) return rc;
        h->exec_pins.push_back(vpr_handle::ExecBlk{static_cast<uint8_t *>(q), bytes + bytes / 2, false});
        best = int(h->exec_pins.size()) - 1;
    }
    h->exec_pins[size_t(best)].used = true;
    *out = h->exec_pins[size_t(best)].p;
    return VPR_OK;
}

void free_batch(vpr_handle *h) {
    for (size_t k = 0; k < h->allocs.size(); k++) {
        if (h->alloc_bytes[k]) h->dev_cache.push_back(vpr_handle::Blk{h->allocs[k], h->alloc_bytes[k]});
        else (void)x_free(h, h->allocs[k], SITE);
    }
    h->allocs.clear(); h->alloc_bytes.clear();
    while (h->dev_cache.size() > 96) {      // (a long run over batches of very different sizes: drop the smallest blocks)
        size_t m = 0;
        for (size_t k = 1; k < h->dev_cache.size(); k++) if (h->dev_cache[k].bytes < h->dev_cache[m].bytes) m = k;
        (void)x_free(h, h->dev_cache[m].p, SITE);
        h->dev_cache.erase(h->dev_cache.begin() + long(m));
    }
    h->pool_cur = nullptr; h->pool_left = 0; h->pool_next = size_t(16) << 20;
    h->exec_blks.clear();
    h->exec_pins.clear();
    for (void *p : h->pinned) (void)hipHostFree(p);
    h->pinned.clear();
    for (auto &b : h->pinned_blk) h->pin_cache.push_back(b);
    h->pinned_blk.clear();
    while (h->pin_cache.size() > 48) { (void)hipHostFree(h->pin_cache.front().p); h->pin_cache.erase(h->pin_cache.begin()); }
    h->hp_fail = nullptr; h->hp_cnt = nullptr; h->hp_flag = nullptr;
    for (int s = 0; s < 4; s++) h->hp_dspan[s] = nullptr;
    h->res_dev = nullptr; h->res_bytes = 0; h->res_mirror = nullptr;     // (a mirror block belongs to the caller: it just stops matching)
    h->events.clear();
    h->descs.clear();
    {
        std::vector<int32_t> kw; std::vector<uint32_t> ko;
        kw.swap(h->plan0.work); ko.swap(h->plan0.off128);
        h->plan0 = Plan();
        kw.clear(); ko.clear();
        h->plan0.work.swap(kw); h->plan0.off128.swap(ko);
    }
    h->dirty.clear();
    h->d_arena = nullptr; h->d_secs = nullptr;
    for (int k = 0; k < 4; k++) h->d_cls[k] = nullptr;
    h->d_hist = nullptr; h->hist_cap = 0; h->d_pb = nullptr;
    for (int k = 0; k < 4; k++) {      // (the replay scratches are the handle's, not the batch's: they survive)
        if (h->want0 > 0 && h->lad[k].arena_bytes > 0) h->lad_hw[k] = double(h->lad[k].arena_bytes) / double(h->want0);
        LadderCtx keep;
        for (int e = 0; e < 2; e++) {
            keep.tie_scratch[e] = h->lad[k].tie_scratch[e]; keep.tie_scratch_bytes[e] = h->lad[k].tie_scratch_bytes[e];
            keep.tie_first[e] = h->lad[k].tie_first[e];
        }
        h->lad[k] = keep;
    }
    for (auto &b : h->parked) h->dev_cache.push_back(b);     // (nothing is in flight when a batch is released)
    h->parked.clear();
    h->resident.clear();
    h->res0_cnt = 0;
    h->d_ed_scratch = nullptr; h->ed_scratch_ints = 0;
    h->d_tie_list = nullptr; h->hp_tie_list = nullptr; h->tie_list_cap = 0; h->d_tie_cnt = nullptr; h->hp_tie_cnt = nullptr;
    h->hp_tie_jobs = nullptr; h->tie_jobs_cap = 0;
    h->d_tie_dec = nullptr; h->tie_dec_cap = 0; h->d_tie_ndec = nullptr; h->plan0_pos.clear();
    h->d_zl_hdr = nullptr; h->d_zl_in = nullptr; h->d_zl_log = nullptr; h->zl_wave0.clear();
    h->d_d1_hdr = nullptr; h->d_d1_in = nullptr; h->d_d1_log = nullptr; h->d_d1_fail = nullptr; h->d_d1_info = nullptr; h->d_d1_blk = nullptr;
    h->d1_in_cap = h->d1_log_cap = 0; h->d1_wave_cap = h->d1_fail_cap = 0;
    h->uploaded = h->executed = false;
}

int64_t books_free(int64_t reported_free, int64_t total) { return dev_books().books_free(reported_free, total); }

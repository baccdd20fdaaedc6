// pr_host.h -- host-side types of the library shared by its translation units: the planning structures, the handle, and the
// helpers every unit calls (error reporting, the counted allocator calls, device and page-locked blocks of the batch / the execute).
//   pr_api.hip      planner, vpr_create / vpr_upload (memory plan) / vpr_execute (Exec), all device kernels of the path
//   pr_mem.hip      allocator wrappers, the process-wide books of device memory, block caches (dev_alloc, pin_alloc, exec_alloc)
//   pr_results.hip  vpr_download, vpr_results_alloc, tallies, timing and launch statistics
//   pr_collect.hip  the counters histogram, vpr_pr_counts, and the RCCL collectives (vpr_allreduce_counts, vpr_allgather_phase)
#ifndef PR_HOST_H_
#define PR_HOST_H_
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/vcfdist_pr.h"
#include "pr_device.h"

namespace vprh {

struct KernelClass { int nt, c, max_len; };
// dense thread-chunk configurations: a plane of up to nt*c cells per row
const KernelClass CLASSES[] = {
    {64, 1, 64}, {64, 4, 256}, {256, 4, 1024}, {256, 8, 2048}, {1024, 8, 8192}, {1024, 16, 16384}, {1024, 32, 32768},
    {1024, 32, 0x7fffffff},     // no one-workgroup kernel: column strips only (pr_strip.hip)
};
const int STRIP_CLS = 4;        // kernel classes from 1024 threads x 8 cells on (more than 2048 columns): column strips
const int STRIP_ONLY_CLS = 7;
const int N_CLASSES = sizeof(CLASSES) / sizeof(CLASSES[0]);
const size_t LDS_MAX = 160 * 1024;
// window levels an alignment climbs until its exit test passes: 16 cells (four alignments per wave, only
// for alignments shorter than LONG_LT rows), 64, 256, 1024 cells (one wave per alignment), dense
// LV_Z: 16 cells, zero-distance variant (accepts only alignments with s = 0); LV_Q16: 16 cells, general.
// (enum LV_*: pr_device.h)
const int LV_WINDOW[] = {16, 16, 64, 256, 1024, 0};   // window width = flag layout of the level
const int LV_TAG[] = {8, 16, 64, 256, 1024, 0};       // value of band_ok / AlnDesc::band_pad that marks the level
// truth rows from which an alignment is a latency chain
// Alignments with LONG_LT truth rows or more are latency chains (rows are sequential): a batch holds a handful of them and
// the longest bounds the step, so they start at LONG_LV, four waves per alignment (k_fwd_wide<4>: 0.45 us per row against
// 1 us for the one-wave 64-cell kernels).  Everything shorter is throughput work for the lane / 16-cell kernels.
// (Sending long alignments through the lane kernel and the 16-cell round first was measured on the SV and stress
// workloads: no gain on the first -- what reaches the dense level there has s > 0 -- and 8x slower on the second, whose
// rejects then climb the ladder in dozens of workspace-sized rounds.)
const int LONG_LT = 1024;      // (2048 until round 4: the lane kernels' launches lasted as long as their longest waves, 2 047 rows of three dependent passes)
const int LONG_LV = 2;    // LV_C1
const int32_t CREDIT_WAVE_MAX = 16384;                // alignments of a launch up to which the credit walk takes a wavefront each
const int64_t WSEG_MAX_ROWS = int64_t(4) << 20;      // truth rows of a launch up to which its walk runs over segments (pr_walkseg.hip)

// std::vector whose resize() leaves trivially constructible elements uninitialised (the planner fills millions of
// 96-byte descriptors from several threads; zero-filling them first costs as much as the fill)
template <typename T>
struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <typename U> void construct(U *p) { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
using DescVec = std::vector<AlnDesc, NoInitAlloc<AlnDesc>>;

// fn(begin, end, thread) over [0, n) on up to PAR_MAX host threads (planning a batch of a million superclusters is a few
// passes over 4 M descriptors: memory-latency bound on one core)
const int PAR_MAX = 32;
// (a pool that lives as long as the library: starting 32 threads per pass costs more than most passes, and with several
// batches in flight on several handles the thread stacks' mmap / munmap calls serialise on the address space)
class ParPool {
  public:
    static ParPool &get() { static ParPool *p = new ParPool(); return *p; }      // (never destroyed: no join at exit)
    // run task(t) for t in [0, nt) on the workers and the calling thread
    void run(size_t nt, const std::function<void(size_t)> &task) {
        Job job;
        job.task = &task; job.nt = nt;
        {
            std::lock_guard<std::mutex> g(m_);
            jobs_.push_back(&job);
        }
        cv_.notify_all();
        work_on(job);
        std::unique_lock<std::mutex> g(m_);
        job.done_cv.wait(g, [&] { return job.done == job.nt; });
    }
  private:
    struct Job { const std::function<void(size_t)> *task; size_t nt = 0, next = 0, done = 0; std::condition_variable done_cv; };
    // CPUs the process may use: the cgroup's quota where there is one (a container that sees 256 cores may be allowed 16:
    // more runnable threads than that and the scheduler suspends the whole group for the rest of the period -- including
    // the threads that feed the GPU)
    static unsigned cpu_limit() {
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
        // one process per GPU on a node (torch.distributed.run exports the number of local ranks): they share the quota
        if (const char *lw = getenv("LOCAL_WORLD_SIZE")) { const int n = atoi(lw); if (n > 1) hw = std::max(1u, hw / unsigned(n)); }
        return hw;
    }
  public:
    // detached workers beside the calling thread for a quota of `lim` CPUs: none when the quota is 1 - 3 (the unsigned
    // `lim - max(2, lim / 4)` of round 3 wrapped around for lim = 1 and started 31)
    static unsigned pool_workers(unsigned lim) {
        const unsigned keep = std::max(2u, lim / 4);
        return (lim > keep ? std::min<unsigned>(lim - keep, PAR_MAX) : 1u) - 1;
    }
  private:
    ParPool() {
        // (a quarter of the quota stays free for the callers themselves and the runtime's threads)
        const unsigned n = pool_workers(cpu_limit());
        for (unsigned t = 0; t < n; t++) std::thread([this] { worker(); }).detach();
    }
    void work_on(Job &job) {
        for (;;) {
            size_t t;
            {
                std::lock_guard<std::mutex> g(m_);
                if (job.next >= job.nt) return;
                t = job.next++;
                if (job.next >= job.nt) jobs_.erase(std::find(jobs_.begin(), jobs_.end(), &job));
            }
            (*job.task)(t);
            {
                std::lock_guard<std::mutex> g(m_);
                if (++job.done == job.nt) job.done_cv.notify_all();
            }
        }
    }
    void worker() {
        for (;;) {
            Job *job;
            size_t t;
            {   // (a task is claimed under the lock that found the job: a job with an unfinished task cannot go away)
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return !jobs_.empty(); });
                job = jobs_.front();
                t = job->next++;
                if (job->next >= job->nt) jobs_.pop_front();
            }
            (*job->task)(t);
            {
                std::lock_guard<std::mutex> g(m_);
                if (++job->done == job->nt) job->done_cv.notify_all();
            }
        }
    }
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Job *> jobs_;
};
template <typename F>
void par_for(size_t n, F fn) {
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nt = std::min<size_t>(std::min<unsigned>(hw, PAR_MAX), n / 32768);
    if (nt <= 1) { fn(size_t(0), n, 0); return; }
    const std::function<void(size_t)> task = [&](size_t t) { fn(n * t / nt, n * (t + 1) / nt, int(t)); };
    ParPool::get().run(nt, task);
}

// base descriptors (no workspace layout) of the uploaded batch: computed on demand from host copies of its offsets
struct BaseDescs {
    std::vector<int64_t> hap_off[4], ref_off;
    BatchOffsets O;
    size_t n = 0;
    void set(const vpr_batch *b, const std::vector<int64_t> *var_off) {      // var_off: the handle's copies
        const size_t m = size_t(b->n_sc) + 1;
        for (int s = 0; s < 4; s++) {
            hap_off[s].assign(b->hap_off[s], b->hap_off[s] + m); O.hap_off[s] = hap_off[s].data();
            O.var_off[s] = var_off[s].data();
        }
        ref_off.assign(b->ref_off, b->ref_off + m); O.ref_off = ref_off.data();
        n = size_t(b->n_sc) * 4;
    }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    void clear() { n = 0; }
    AlnDesc operator[](size_t a) const { return base_desc(O, int64_t(a)); }
    // the three lengths of alignment a alone (base_desc also sums variant offsets and places the section table: a planning
    // pass over four million alignments only wants these)
    void lens(size_t a, int32_t &Lq, int32_t &Lr, int32_t &Lt) const {
        const size_t sc = a >> 2;
        const int i = int(a & 3), qs = i >> 1, ts = 2 + (i & 1);
        Lq = int32_t(O.hap_off[qs][sc + 1] - O.hap_off[qs][sc]);
        Lt = int32_t(O.hap_off[ts][sc + 1] - O.hap_off[ts][sc]);
        Lr = int32_t(O.ref_off[sc + 1] - O.ref_off[sc]);
    }
};

struct Launch { int cls; int64_t work_off; int32_t count; };   // dense: one k_fwd/k_bwd launch of a class
struct Chunk {
    int64_t work_off = 0; int32_t count = 0;   // slice of the plan's work list
    std::vector<Launch> launches;              // dense plans only
    int64_t cells = 0, in_bytes = 0;           // touched cells / input bytes of the chunk
    // windowed plans: the leading n_long long alignments get their own launch sequence (part 0)
    int32_t n_long = 0;
    int64_t part_cells[2] = {0, 0}, part_in[2] = {0, 0}, part_dense[2] = {0, 0}, part_rows[2] = {0, 0};
};
// a set of alignments with workspace offsets assigned, all at window level `lv` (a 16-cell plan holds its
// long alignments, which start at LONG_LV, in front)
struct Plan {
    int lv = LV_DENSE;
    DescVec descs;                  // compact, in work-list order (empty when `lazy`)
    // lazy plan (round 0 of a windowed batch): the descriptors are a function of the batch offsets, the alignment and its
    // workspace offset (pr_device.h: base_desc + window_layout); the host keeps 4 bytes per alignment, the device builds
    // its copy from the same 4 bytes (k_build_plan), and plan_desc() below recomputes the few the host ever looks at
    bool lazy = false;
    std::vector<uint32_t> off128;   // workspace offset of every entry in 128-byte units (lazy plans)
    int tag_or = 0, long_lt = 0;
    std::vector<int32_t> work;      // alignment ids
    std::vector<Chunk> chunks;
    AlnDesc *d_descs = nullptr;     // device copy of `descs` (plan 0 only, cached)
    int32_t *d_work = nullptr;
    uint8_t *arena = nullptr;       // workspace the offsets refer to
    int64_t arena_used = 0;         // bytes of it the largest chunk occupies
    int64_t total_need = 0;         // bytes all chunks together would occupy
};

// a retry ladder's private resources (see vpr_execute)
struct LadderCtx {
    static const int N_SLOTS = 16;
    hipStream_t ls = nullptr;
    int slot0 = 0;                          // first of its N_SLOTS fail slots
    int64_t fail_base = 0;                  // its region of the fail-list buffer
    uint8_t *arena = nullptr; int64_t arena_bytes = 0;
    int32_t *d_work = nullptr; size_t work_cap = 0;     // work lists of the plans in flight
    AlnDesc *hp_descs = nullptr; int32_t *hp_work = nullptr; size_t hp_cap = 0;   // host-pinned source of k_stage
    int slot_cur = 0; int64_t fail_cur = 0, arena_cur = 0; size_t stage_cur = 0;
    std::vector<std::pair<int, int64_t>> pending;       // (slot, fail list offset) of the launches in flight
    std::vector<Plan> plans;
    // tie ladders only: a side stream for the early replays (they run beside the repeated forward sweep), the event that
    // joins it back, and the replay scratch of each of the two streams (grown on demand, released with the batch)
    hipStream_t ls2 = nullptr; hipEvent_t ev2 = nullptr;
    uint32_t *tie_scratch[2] = {nullptr, nullptr}; int64_t tie_scratch_bytes[2] = {0, 0};
    // bytes at the front of each scratch that are preset (filled at the start of vpr_execute, beside round 0, with as much
    // as the previous execute's first launch used) and the size of that first launch
    int64_t tie_clean[2] = {0, 0}, tie_first[2] = {0, 0}; bool tie_first_seen[2] = {false, false};
};

struct EvPair { hipEvent_t a, b; int kind; vpr_launch_stat st; };
const int TIE_DEC_SLOTS = 64;     // early-replay launches per execute that can keep a decision list

}  // namespace vprh
using namespace vprh;

struct vpr_handle {
    vpr_config cfg;
    std::string err;
    bool debug = false;                  // VPR_DEBUG in the environment at vpr_create: progress lines on stderr
    // host-side cost of the current / last vpr_execute: allocator calls and blocking waits (vpr_timing reports them; with
    // VPR_STALL_LOG in the environment every such call that takes more than 5 ms is printed with its site)
    bool soft_alloc = false;            // the allocation under way is optional growth (x_malloc: larger reserve)
    struct HostStat {
        int64_t n_dev_alloc = 0, n_dev_free = 0, n_pin_alloc = 0;
        double ms_alloc = 0, ms_sync = 0, ms_idle_max = 0;
    } hs;
    bool stall_log = false;
    hipStream_t stream = nullptr;
    hipStream_t cls_stream[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // batch-lifetime device allocations: pool blocks, carved by dev_alloc.  Released blocks are KEPT (dev_cache / pin_cache)
    // and handed out again by the next upload: hipFree of a used multi-GB block takes seconds and page-locking host memory
    // runs at ~6 GB/s, which is most of what a re-upload into a used handle cost
    struct Blk { void *p; size_t bytes; };
    std::vector<Blk> dev_cache, pin_cache, pinned_blk;
    std::vector<Blk> parked;             // device blocks outgrown during an execute: to dev_cache when the batch is released
    int64_t dev_total = 0;               // the device's memory (vpr_create)
    int64_t tie_scratch_max = 0, lad_arena_max = 0;   // bounds of the replay scratches / ladder workspaces that grow on demand
    // The batch's memory plan (vpr_upload): of the device memory free at that point the library leaves mem_reserve alone,
    // round 0's workspace takes what its plan asks for (at most arena_share() of the rest), and what remains is split
    // between the ladders' workspaces and the replay scratches, which start small and grow on demand inside their halves.
    int64_t mem_reserve = 0, lad_budget = 0, tie_budget = 0, lad_bytes = 0, tie_bytes = 0;
    // what each ladder's workspace had grown to when the last batch was released, as a fraction of that batch's round-0 need:
    // the next batch's ladders start there instead of growing during its first execute
    double lad_hw[4] = {0, 0, 0, 0}; int64_t want0 = 0;
    std::vector<size_t> alloc_bytes;     // sizes of `allocs`
    std::vector<void *> allocs;
    uint8_t *pool_cur = nullptr;         // bump pointer into the newest block
    size_t pool_left = 0, pool_next = size_t(16) << 20;   // block sizes double up to 2 GiB (a batch needs ~150 arrays)
    DevBatch dB;
    // host mirrors needed for planning / finalisation
    int32_t n_sc = 0;
    uint8_t *d_alias = nullptr;          // [superclusters] alias bits (k_hap_alias), all 0 without VPR_CFG_HAP_DEDUP
    std::vector<uint8_t> alias;          // host copy
    int64_t n_aliased = 0;               // alignments of the batch that are copies of another one
    int32_t n_limit_sc = 0;              // nonzero: some supercluster of the batch is marked in DevBatch::sc_limit
    std::vector<int64_t> var_off[4];
    std::vector<float> var_qual[4];
    int64_t n_var[4] = {0, 0, 0, 0};
    BaseDescs descs;                     // base descriptors (no workspace offsets)
    std::vector<int32_t> scratch_i32[4]; // planner scratch that keeps its pages across uploads
    // device blocks with the lifetime of one execute (the strip tables and boundary columns of the wide dense sweeps,
    // pr_strip.hip): taken from the batch's allocations, handed out again by the next execute
    struct ExecBlk { uint8_t *p; size_t bytes; bool used; };
    std::vector<ExecBlk> exec_blks, exec_pins;
    bool no_strips = false;              // VPR_NO_STRIPS in the environment: wide alignments stay in one workgroup
    uint8_t *d_save = nullptr;           // second copy of the forward flags of round 0's long part (k_fwd_stripe_save), nullptr: none
    int64_t save_bytes = 0;
    bool no_flag_save = false;           // VPR_NO_FLAG_SAVE: tie rounds of the long part repeat the forward sweep
    bool seq_walk = false;               // VPR_SEQ_WALK: the sequential row-sweep walk instead of the segment-parallel one
    bool no_round_overlap = false;       // VPR_NO_ROUND_OVERLAP: a retry round is complete before the host looks at its fail lists
    std::vector<uint32_t> scratch_u32[2];
    Plan plan0;                          // first round over all alignments, cached at upload
    std::vector<uint8_t> level, level0;  // current / round-0 window level of every alignment
    int64_t last_need = 0;               // workspace bytes of the alignment make_plan could not place
    uint8_t *d_cls[4] = {nullptr, nullptr, nullptr, nullptr};   // SNP / INDEL / SV class of every variant (vpr_upload_var_class)
    int2 *hp_dspan[4] = {nullptr, nullptr, nullptr, nullptr};   // host copy of DevBatch::dspan (page-locked)
    uint8_t *res_dev = nullptr; size_t res_bytes = 0;           // the device region of the result columns (vpr_upload)
    uint8_t *res_mirror = nullptr;                              // the caller's block that mirrors it (vpr_results_alloc), if any
    unsigned long long *d_hist = nullptr;   // vpr_pr_counts: histogram words (batch lifetime, grown on demand)
    size_t hist_cap = 0;
    int32_t *d_pb = nullptr;                // vpr_pr_counts: the caller's phase-block phasing per supercluster
    std::vector<int32_t> dirty;          // alignments whose device descriptor was overwritten by a retry round
    // device side
    AlnDesc *d_descs = nullptr;
    AlnOut *d_outs = nullptr;
    uint8_t *d_arena = nullptr; int64_t arena_bytes = 0;      // workspace of the round-0 plan
    LadderCtx lad[4];                                         // two retry ladders, two tie ladders (long / short part of round 0;
                                                              // own workspaces, beside the arena)
    hipEvent_t ev_slot[2 + 4 * LadderCtx::N_SLOTS] = {};      // "fail list of slot k is complete"
    hipStream_t tie_stream[4] = {nullptr, nullptr, nullptr, nullptr};   // tie ladder k: [2k] main, [2k+1] early replays (high priority)
    hipEvent_t ev_tie2[2] = {nullptr, nullptr};
    hipEvent_t ev_side[2] = {nullptr, nullptr};               // retry ladder k: "the forward sweeps of the round are enqueued"
    hipEvent_t ev_tie[2] = {nullptr, nullptr};                // "the tie list of the long / short part of round 0 is published"
    // plans of the last execute in launch order (the last one that holds an alignment has its final walk); second = the
    // plan's workspace, nullptr once that workspace has been reused
    std::vector<std::pair<std::vector<int32_t>, uint8_t *>> resident;
    // the last chunk of the round-0 plan, whose walks are still in the round-0 workspace: a view into plan0.work (searched
    // last: a later plan of an alignment holds its final walk)
    int64_t res0_off = 0; int32_t res0_cnt = 0;
    Section *d_secs = nullptr; int64_t n_secs_cap = 0;
    int32_t *d_fp[4] = {nullptr, nullptr, nullptr, nullptr};
    int32_t **d_fp_table = nullptr;
    EdJob *d_jobs = nullptr; int32_t jobs_cap = 0; int32_t *d_njobs = nullptr;
    uint32_t *d_err = nullptr;
    int32_t *d_fail = nullptr, *d_cnt = nullptr;   // fail lists + their counters
    // tie pass (pr_tie.hip): list of marked alignments {id, level tag}, counters {marked, replay overflows}, the jobs of
    // the replay launches (host-pinned, read by the kernel directly) and the replay scratch (grown on demand)
    int4 *d_tie_list = nullptr, *hp_tie_list = nullptr; int32_t tie_list_cap = 0;   // {alignment, level tag, consulted ties, 0}
    // completion flags in host-pinned memory, written by one-thread kernels behind the work they stand for: the host
    // polls plain memory instead of HIP events (hipEventQuery in a tight loop delays the very submissions it waits for)
    int32_t *hp_flag = nullptr; int32_t flag_seq = 0;
    int32_t *d_tie_cnt = nullptr, *hp_tie_cnt = nullptr;     // [0] final pass, [1] replay overflows, [2] long part, [3] short part,
                                                             // [4] speculative candidates of the long part
    hipEvent_t ev_spec = nullptr;                            // the speculative replays of the long part are done
    TieJob *hp_tie_jobs = nullptr; size_t tie_jobs_cap = 0;
    int4 *d_tie_dec = nullptr; int64_t tie_dec_cap = 0;       // decision lists of the early replays (one region per launch)
    int32_t *d_tie_ndec = nullptr;                            // their lengths [TIE_DEC_SLOTS]
    std::vector<int32_t> plan0_pos;                           // position of every alignment in plan0's work list
    // zero-distance lane kernel (pr_zl.hip): per-wave headers, the wave-interleaved position words (batch lifetime, written
    // once by k_prep_zl) and the log blocks (shared by the chunks of plan 0, which run one after the other)
    ZlWave *d_zl_hdr = nullptr; uint32_t *d_zl_in = nullptr; uint4 *d_zl_log = nullptr;
    std::vector<int64_t> zl_wave0;                            // first wave of every chunk of plan 0
    // distance-1 lane kernel (pr_d1.hip): headers, position words and log of the waves of zero-level rejects (sized on the device
    // per execute, within these blocks), the list of what it leaves to the in-place 16-cell round and {waves used, waves dropped,
    // length of that list}
    ZlWave *d_d1_hdr = nullptr; uint32_t *d_d1_in = nullptr; uint4 *d_d1_log = nullptr; int32_t *d_d1_fail = nullptr, *d_d1_info = nullptr;
    int32_t *d_d1_blk = nullptr;                              // per-workgroup counts / offsets of the ordered fail lists (k_fails_*)
    hipEvent_t ev_offsets = nullptr;                          // upload: the batch's offsets are on the device (plan0_device waits for it)
    hipEvent_t ev_cred[2] = {nullptr, nullptr};               // the lane levels' credit walks on a side stream: fork / join
    bool side_credit = true;                                  // (VPR_NO_SIDE_CREDIT: behind the 16-cell round on the part's stream, as until round 4)
    // what the upload's one host pass over the superclusters found for round 0's plan (plan0_device): the parts' sums, the total
    // workspace need in 128-byte units, the long alignments (-matrix bytes, alignment), "some alignment cannot be placed"
    struct Plan0Pass {
        bool valid = false, bad = false;
        int64_t part_cells[2] = {0, 0}, part_in[2] = {0, 0}, part_dense[2] = {0, 0}, part_rows[2] = {0, 0}, need128 = 0, need_max = 0;
        std::vector<std::pair<int64_t, int32_t>> big;
    } p0;
    int64_t d1_in_cap = 0, d1_log_cap = 0;
    int32_t d1_wave_cap = 0, d1_fail_cap = 0;
    int32_t d1_max_rows = 256;                                // rejects of more truth rows stay with the 16-cell kernels (VPR_D1_MAX_ROWS)
    std::vector<ZlWave> zl_hdr_host;
    int32_t long_lt = LONG_LT;                                // rows from which an alignment belongs to the long part of plan 0
    // vpr_upload_variants: the variant tables of the batch being uploaded (the device generates the Level A arrays from
    // them, pr_gen.hip), and the contig sequence of the previous upload, which stays resident as long as the caller keeps
    // passing the same one (a whole-genome run uploads a contig once, not once per batch)
    const vpr_variants *gen_src = nullptr;
    uint8_t *d_ctg_seq = nullptr; const uint8_t *ctg_src = nullptr; int64_t ctg_bytes = 0; uint64_t ctg_probe = 0;
    // host-pinned, device-visible mirrors of d_fail / d_cnt: a publish kernel on the producing stream fills them, so the
    // host reads a fail list after an event wait and issues no copy that the bulk kernels of the round could starve
    int32_t *hp_fail = nullptr, *hp_cnt = nullptr;
    std::vector<void *> pinned;
    int32_t *d_ed_scratch = nullptr; size_t ed_scratch_ints = 0;
    int64_t ed_max_len = -1;                               // longest ref / truth string of the batch (-1: not looked at yet)
    std::vector<EvPair> events;
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;   // timing events, reused by every execute
    std::vector<hipStream_t> pad_streams;                  // (diagnostic, VPR_STREAM_PAD)
    int zl_lds_bytes = 0;                // (diagnostic, VPR_ZL_LDS_KB) LDS the zero level's waves ask for and never touch: caps how many of them a compute unit holds
    int lane_prio_rows = 256;            // waves of the lane levels with at least this many rows issue ahead of the others (k_zero_lane; a quarter for k_one_lane)
    DevResults dR;                       // final results, produced on the device
    vpr_timing timing;
    bool uploaded = false, executed = false;
};

// ---- helpers shared by the translation units (pr_mem.hip unless noted)
int fail(vpr_handle *h, int code, const char *fmt, ...);            // message into the handle (or the create-time slot), returns code
const char *vpr_create_error();                                      // the message of a failed vpr_create
#define HIPCHK(h, call)                                                                         \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(h, VPR_ERR_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define SITE_STR2(x) #x
#define SITE_STR(x) SITE_STR2(x)
#define SITE __FILE__ ":" SITE_STR(__LINE__)
inline double wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
const char *exp_getenv(const char *name);       // switches of closed experiments: read only by a -DVPR_EXPERIMENTS build
int poison_byte();
int64_t dev_reserve_bytes();
double free_share();
double arena_share();
double ladder_share();
int64_t books_free(int64_t reported_free, int64_t total);            // the driver's free figure corrected by the process's own books
hipError_t x_malloc(vpr_handle *h, void **q, size_t bytes, const char *site);
hipError_t x_free(vpr_handle *h, void *q, const char *site);
hipError_t x_host_malloc(vpr_handle *h, void **q, size_t bytes, const char *site);
hipError_t x_sync(vpr_handle *h, hipStream_t s_, const char *site);
hipError_t x_event_sync(vpr_handle *h, hipEvent_t ev, const char *site);
void *dev_block(vpr_handle *h, size_t bytes, hipError_t *err);
int pin_alloc(vpr_handle *h, void **out, size_t bytes);
int dev_alloc_bytes(vpr_handle *h, void **p, size_t bytes);          // 256-byte granules from the batch's pool blocks
int exec_alloc(vpr_handle *h, void **out, size_t bytes);
int exec_pin(vpr_handle *h, void **out, size_t bytes);
void free_batch(vpr_handle *h);
template <typename T>
int dev_alloc(vpr_handle *h, T **p, size_t n) {
    void *q = nullptr;
    const int rc = dev_alloc_bytes(h, &q, (std::max<size_t>(n, 1) * sizeof(T) + 255) & ~size_t(255));
    *p = static_cast<T *>(q);
    return rc;
}
template <typename T>
int dev_upload(vpr_handle *h, const T **dst, const T *src, size_t n) {
    T *p;
    int rc = dev_alloc(h, &p, n);
    if (rc) return rc;
    if (n && src) HIPCHK(h, hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, h->stream));   // (null: filled on the device)
    *dst = p;
    return VPR_OK;
}
#endif

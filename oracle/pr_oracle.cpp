// pr_oracle.cpp -- CPU restatement of vcfdist's precision/recall alignment path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the product (vcfdist_amd/, include/) may
// include, link or call this file; only tests/, __graft_entry__.smoke() and the
// cpu_baseline leg of bench.py use it, as the checker / timed CPU baseline.
//
// It restates, function by function, the algorithm of the reference
// (TimD1/vcfdist v2.6.4, all in src/dist.cpp):
//     gen_hap()        <- generate_ptrs_strs          dist.cpp:145-242
//     forward_pass()   <- calc_prec_recall_aln        dist.cpp:251-443
//     phase_of()       <- store_phase                 dist.cpp:449-475
//     backward_pass()  <- calc_prec_recall_path       dist.cpp:486-823
//     walk_path()      <- get_prec_recall_path_sync   dist.cpp:842-999
//     assign_credit()  <- calc_prec_recall            dist.cpp:1005-1401
//     edit_distance()  <- wf_ed                       dist.cpp:1406-1506
// The forward pass keeps the reference's container discipline (a FIFO queue
// seeded by iterating a std::unordered_set keyed with the same hash constants,
// dist.h:42-50) because the value kept in swap_pred for a cell with several
// optimal swap predecessors is "last writer wins" and therefore depends on that
// iteration order (dist.cpp:347,376).
//
// PARITY PIN.  The reference cannot be built in this image (every translation unit on the path includes
// htslib/vcf.h through variant.h:9, htslib is not installed, and writing a stand-in header is not allowed), and it
// ships no unit tests or golden vectors for this path.  What it does ship is a known answer, demo/output.txt, and
// the oracle chain is pinned on it (tests/test_demo_known_answer.py): the reference's demo
// VCFs + BED through the VCF/BED front end, oracle biWFA clustering, superclustering, THIS file, phasing and the
// summary reproduce the published SNP and SV rows exactly -- counts and the printed precision / recall / F1 /
// Q-score -- on a surrogate FASTA (the demo's GRCh38 slice is not distributable), and the INDEL / ALL rows one count
// lower in every column, which is what the real reference prints on a surrogate FASTA.  The real
// FASTA would close that last count.  Further pins (tests/test_oracle.py): (1) the reference-produced toy vector of
// SURVEY.md Appendix A.1, (2) textbook Levenshtein for edit_distance, (3) an independent dense dynamic programme
// (tests/dense_model.py) for distances, flags, backward scores and path pointers, (4) hand-derived credit cases.
// See DESIGN.md section 5.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <queue>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../include/vcfdist_pr.h"
#include "pr_oracle.h"

namespace {

// backtracking pointer flags, defs.h:110-120
enum : uint8_t {
    F_INS = 1, F_DEL = 2, F_MAT = 4, F_SUB = 8, F_SWP = 16, F_PATH = 32, F_MAIN = 96, F_SYNC = 128
};

struct Cell {  // dist.h:14-40 (hi = 2*aln + plane)
    int hi, qri, ti;
    Cell() : hi(0), qri(0), ti(0) {}
    Cell(int h, int q, int t) : hi(h), qri(q), ti(t) {}
    bool operator==(const Cell &o) const { return hi == o.hi && qri == o.qri && ti == o.ti; }
};

}  // namespace

namespace std {
template <> struct hash<Cell> {  // dist.h:42-50, same constants => same bucket order
    std::uint64_t operator()(const Cell &x) const noexcept {
        return (uint64_t(x.hi) * 73856093 + 0x517cc1b727220a95) ^
               (uint64_t(x.qri) * 19349669 + 0xd15f392b3d4704a2) ^
               (uint64_t(x.ti) * 83492791);
    }
};
}  // namespace std

namespace {

typedef std::unordered_set<Cell> CellSet;
typedef std::unordered_map<Cell, Cell> CellMap;

// one hap string with its pointers to the reference string
struct Hap {
    std::string seq;
    std::vector<int> ptr;
    std::vector<int> flag;
};

// Dense matrices [rows=q][cols=t], held as the reference holds them: one heap vector per row
// (std::vector<std::vector<T>>, dist.cpp:1828-1844, 523-534; the `done` matrices are std::vector<std::vector<bool>>,
// dist.cpp:293-296).  That is part of the algorithm's cost and belongs in a baseline that is timed: rows of a few KB
// come from the allocator's heap and are reused from one supercluster to the next, whereas one flat block per matrix
// is mapped and unmapped every time and spends its life in page faults (a 10 kb supercluster took 8 s that way, 5x
// the reference's own time; profiles/r03_cpu_calibration.json).
template <class T>
struct MatRows {
    int rows = 0, cols = 0;
    std::vector<std::vector<T>> m;
    void init(int r, int c, T v) { rows = r; cols = c; m.assign(size_t(r), std::vector<T>(size_t(c), v)); }
    T &at(int q, int t) { return m[size_t(q)][size_t(t)]; }
    T at(int q, int t) const { return m[size_t(q)][size_t(t)]; }
    size_t size() const { return size_t(rows) * size_t(cols); }
    T flat(size_t k) const { return m[k / size_t(cols)][k % size_t(cols)]; }
};
typedef MatRows<uint8_t> Mat8;
typedef MatRows<int16_t> Mat16;
struct MatBit {
    std::vector<std::vector<bool>> m;
    void init(int r, int c, bool v) { m.assign(size_t(r), std::vector<bool>(size_t(c), v)); }
    std::vector<bool>::reference at(int q, int t) { return m[size_t(q)][size_t(t)]; }
};

inline bool fwd_allow(int f) { return !(f & VPR_PTR_VARIANT) || (f & VPR_PTR_VAR_END); }  // dist.cpp:336-339
inline bool bwd_allow(int f) { return !(f & VPR_PTR_VARIANT) || (f & VPR_PTR_VAR_BEG); }  // dist.cpp:600-601

// inputs of one alignment (views)
struct AlnIn {
    int i;                 // alignment index 0..3
    const Hap *q;          // query hap (string + q->r)
    const Hap *t;          // truth hap (string + t->r)
    const std::string *ref;
    const std::vector<int> *r2q_ptr, *r2q_flag;  // ref -> this query hap
};

struct SwapStats {
    int64_t writes = 0, conflict_writes = 0, used = 0, used_conflict = 0, used_conflict_nonmax = 0;
};

struct AlnState {
    Mat8 aln[2];           // [plane] forward flags
    Mat8 pptr[2];          // path_ptrs
    Mat16 pscore[2];       // path_scores
    CellMap swap_pred;
    std::unordered_map<Cell, std::vector<Cell>> swap_writers;  // only cells with >1 distinct writer
    int s = 0, end_plane = 0, beg_plane = 0;
    std::vector<Cell> path;
    std::vector<uint8_t> sync, edit;
    uint32_t status = 0;
    SwapStats st;
};

// ---------------------------------------------------------------------------
// forward pass, dist.cpp:286-442
// ---------------------------------------------------------------------------
void forward_pass(const AlnIn &in, AlnState &A, bool track_writers) {
    const int qi = 2 * in.i, ri = 2 * in.i + 1;
    // (the reference works on its own copies of the strings and pointer arrays, one set per alignment: dist.cpp:270-281.  The
    // copies are part of its per-supercluster cost, so the timed restatement makes them too.)
    const std::string Q = in.q->seq, T = in.t->seq, R = *in.ref;
    const std::vector<int> q_ptr_copy = in.q->ptr, q_flag_copy = in.q->flag, t_ptr_copy = in.t->ptr, t_flag_copy = in.t->flag,
                           r_ptr_copy = *in.r2q_ptr, r_flag_copy = *in.r2q_flag;
    (void)q_ptr_copy; (void)q_flag_copy; (void)t_ptr_copy; (void)t_flag_copy; (void)r_ptr_copy; (void)r_flag_copy;
    const int Lq = Q.size(), Lt = T.size(), Lr = R.size();
    A.aln[0].init(Lq, Lt, 0);
    A.aln[1].init(Lr, Lt, 0);
    MatBit done[2];
    done[0].init(Lq, Lt, false);
    done[1].init(Lr, Lt, false);
    auto P = [&](const Cell &c) -> uint8_t & { return A.aln[c.hi == ri].at(c.qri, c.ti); };
    auto D = [&](const Cell &c) -> std::vector<bool>::reference { return done[c.hi == ri].at(c.qri, c.ti); };

    std::queue<Cell> fifo;
    fifo.push(Cell(qi, 0, 0));
    A.aln[0].at(0, 0) |= F_MAT;
    done[0].at(0, 0) = 1;
    fifo.push(Cell(ri, 0, 0));
    A.aln[1].at(0, 0) |= F_MAT;
    done[0].at(0, 0) = 1;  // dist.cpp:305 marks the QUERY cell again, not the REF one

    CellSet curr_wave, prev_wave;
    A.s = 0;
    // (*swap_pred_maps[i])[z] = x, dist.cpp:347,376 -- last writer wins.  Cells that get more than one
    // distinct writer are remembered: the kept value then depends on the container iteration order, and
    // VPR_ST_SWAP_TIE is part of the result contract.
    auto record_swap = [&](const Cell &z, const Cell &x) {
        A.st.writes++;
        auto ins = A.swap_pred.emplace(z, x);
        if (!ins.second && !(ins.first->second == x)) {
            A.st.conflict_writes++;
            auto &w = A.swap_writers[z];
            if (w.empty()) w.push_back(ins.first->second);
            if (std::find(w.begin(), w.end(), x) == w.end()) w.push_back(x);
            ins.first->second = x;
        }
    };
    (void)track_writers;

    while (true) {
        if (fifo.empty()) { A.status |= VPR_ST_ERR_UNFINISHED; return; }  // dist.cpp:314
        while (!fifo.empty()) {  // extend at the same score, dist.cpp:317-381
            Cell x = fifo.front();
            fifo.pop();
            prev_wave.insert(x);
            if (x.hi == qi) {
                Cell y(qi, x.qri + 1, x.ti + 1);
                if (y.qri < Lq && y.ti < Lt && Q[y.qri] == T[y.ti]) {
                    if (!D(y)) {
                        if (curr_wave.find(y) == curr_wave.end()) { fifo.push(y); curr_wave.insert(y); }
                        P(y) |= F_MAT;
                    }
                }
                Cell z(ri, in.q->ptr[x.qri] + 1, x.ti + 1);
                if (fwd_allow(in.q->flag[x.qri]) && fwd_allow(in.t->flag[x.ti])) {
                    if (z.qri < Lr && z.ti < Lt && R[z.qri] == T[z.ti]) {
                        if (!D(z)) {
                            if (curr_wave.find(z) == curr_wave.end()) { fifo.push(z); curr_wave.insert(z); }
                            P(z) |= F_SWP;
                            record_swap(z, x);
                        }
                    }
                }
            } else {
                Cell y(ri, x.qri + 1, x.ti + 1);
                if (y.qri < Lr && y.ti < Lt && R[y.qri] == T[y.ti]) {
                    if (!D(y)) {
                        if (curr_wave.find(y) == curr_wave.end()) { fifo.push(y); curr_wave.insert(y); }
                        P(y) |= F_MAT;
                    }
                }
                Cell z(qi, (*in.r2q_ptr)[x.qri] + 1, x.ti + 1);
                if (fwd_allow((*in.r2q_flag)[x.qri]) && fwd_allow(in.t->flag[x.ti])) {
                    if (z.qri < Lq && z.ti < Lt && Q[z.qri] == T[z.ti]) {
                        if (!D(z)) {
                            if (curr_wave.find(z) == curr_wave.end()) { fifo.push(z); curr_wave.insert(z); }
                            P(z) |= F_SWP;
                            record_swap(z, x);
                        }
                    }
                }
            }
        }
        for (const Cell &x : curr_wave) D(x) = 1;  // dist.cpp:384-387
        curr_wave.clear();
        if (done[0].at(Lq - 1, Lt - 1) || done[1].at(Lr - 1, Lt - 1)) break;  // dist.cpp:390-391

        for (const Cell &x : prev_wave) {  // next score, dist.cpp:395-424
            const int len = (x.hi == qi) ? Lq : Lr;
            if (x.qri + 1 < len) {
                Cell y(x.hi, x.qri + 1, x.ti);
                if (!D(y) && curr_wave.find(y) == curr_wave.end()) { fifo.push(y); curr_wave.insert(y); }
                if (!D(y)) P(y) |= F_INS;
            }
            if (x.ti + 1 < Lt) {
                Cell y(x.hi, x.qri, x.ti + 1);
                if (!D(y) && curr_wave.find(y) == curr_wave.end()) { fifo.push(y); curr_wave.insert(y); }
                if (!D(y)) P(y) |= F_DEL;
            }
            if (x.qri + 1 < len && x.ti + 1 < Lt) {
                Cell y(x.hi, x.qri + 1, x.ti + 1);
                if (!D(y) && curr_wave.find(y) == curr_wave.end()) { fifo.push(y); curr_wave.insert(y); }
                if (!D(y)) P(y) |= F_SUB;
            }
        }
        prev_wave.clear();
        A.s++;
    }
    // prefer the QUERY plane, dist.cpp:436-440
    if (done[0].at(Lq - 1, Lt - 1)) A.end_plane = VPR_PLANE_QUERY;
    else A.end_plane = VPR_PLANE_REF;
}

// ---------------------------------------------------------------------------
// backward max-TP pass, dist.cpp:517-823
// ---------------------------------------------------------------------------
void backward_pass(const AlnIn &in, AlnState &A) {
    const int qi = 2 * in.i, ri = 2 * in.i + 1;
    const int Lq = in.q->seq.size(), Lt = in.t->seq.size(), Lr = in.ref->size();
    A.pptr[0].init(Lq, Lt, 0);
    A.pptr[1].init(Lr, Lt, 0);
    A.pscore[0].init(Lq, Lt, -1);
    A.pscore[1].init(Lr, Lt, -1);
    MatBit done[2];
    done[0].init(Lq, Lt, false);
    done[1].init(Lr, Lt, false);
    auto AP = [&](const Cell &c) -> uint8_t & { return A.aln[c.hi == ri].at(c.qri, c.ti); };
    auto PP = [&](const Cell &c) -> uint8_t & { return A.pptr[c.hi == ri].at(c.qri, c.ti); };
    auto PS = [&](const Cell &c) -> int16_t & { return A.pscore[c.hi == ri].at(c.qri, c.ti); };
    auto D = [&](const Cell &c) -> std::vector<bool>::reference { return done[c.hi == ri].at(c.qri, c.ti); };
    // (own copies, as in the reference: dist.cpp:506-514)
    const std::vector<int> q2r = in.q->ptr, qfl = in.q->flag;
    const std::vector<int> rfl = *in.r2q_flag;
    const std::vector<int> t_ptr_copy = in.t->ptr, t_flag_copy = in.t->flag, r_ptr_copy = *in.r2q_ptr;
    const std::string q_copy = in.q->seq, t_copy = in.t->seq;
    (void)t_ptr_copy; (void)t_flag_copy; (void)r_ptr_copy; (void)q_copy; (void)t_copy;

    std::queue<Cell> fifo;
    Cell start(A.end_plane == VPR_PLANE_QUERY ? qi : ri,
               (A.end_plane == VPR_PLANE_QUERY ? Lq : Lr) - 1, Lt - 1);
    PP(start) = F_MAT;
    AP(start) |= F_MAIN;
    PS(start) = 0;
    fifo.push(start);
    CellSet curr_wave, prev_wave;
    // A one-cell alignment (end cell == start cell; only where a region was cut at the contig end, include/vcfdist_pr.h):
    // the reference's loop below never sets done(0,0) for it and spins for ever (dist.cpp:549-687).  Here the path is
    // that one cell.
    if (start.qri == 0 && start.ti == 0) { A.beg_plane = A.end_plane; if (A.end_plane == VPR_PLANE_QUERY) A.aln[0].at(0, 0) |= F_PATH; return; }

    // relax predecessor y of x with move type `mv` and bonus `tp`
    auto relax = [&](const Cell &x, const Cell &y, uint8_t mv, int tp) {
        AP(y) |= F_PATH;
        if (!D(y) && PS(x) + tp > PS(y)) {
            PP(y) = mv;
            PS(y) = PS(x) + tp;
            if (curr_wave.find(y) == curr_wave.end()) curr_wave.insert(y);
            fifo.push(y);
        } else if (!D(y) && PS(x) + tp == PS(y)) {
            PP(y) |= mv;
        }
    };
    // "enters the first base of a query variant", dist.cpp:572-574
    auto is_tp = [&](const Cell &x) -> int {
        return x.hi == qi && ((q2r[x.qri] != q2r[x.qri - 1] + 1) || (qfl[x.qri] & VPR_PTR_VAR_BEG));
    };

    while (true) {
        while (!fifo.empty()) {  // zero-cost moves, dist.cpp:550-681
            Cell x = fifo.front();
            fifo.pop();
            prev_wave.insert(x);
            if ((AP(x) & F_MAT) && x.qri > 0 && x.ti > 0)
                relax(x, Cell(x.hi, x.qri - 1, x.ti - 1), F_MAT, is_tp(x));
            if ((AP(x) & F_SWP) && x.qri > 0 && x.ti > 0 &&
                bwd_allow(x.hi == ri ? rfl[x.qri] : qfl[x.qri])) {
                auto it = A.swap_pred.find(x);
                if (it == A.swap_pred.end()) { A.status |= VPR_ST_ERR_NO_PTR; return; }  // dist.cpp:605
                A.st.used++;
                auto w = A.swap_writers.find(x);
                if (w != A.swap_writers.end()) {
                    A.st.used_conflict++;
                    int mx = -1;
                    for (const Cell &c : w->second) mx = std::max(mx, c.qri);
                    if (it->second.qri != mx) A.st.used_conflict_nonmax++;
                }
                // leaving a REF-plane cell scores 0 (dist.cpp:614); leaving a QUERY-plane
                // cell uses the same rule as MAT (dist.cpp:656-658)
                relax(x, it->second, F_SWP, x.hi == ri ? 0 : is_tp(x));
            }
        }
        for (const Cell &x : curr_wave) D(x) = 1;
        curr_wave.clear();
        if (done[0].at(0, 0) || done[1].at(0, 0)) break;  // dist.cpp:687

        for (const Cell &x : prev_wave) {  // unit-cost moves, dist.cpp:689-806
            if ((AP(x) & F_SUB) && x.qri > 0 && x.ti > 0)
                relax(x, Cell(x.hi, x.qri - 1, x.ti - 1), F_SUB, is_tp(x));
            if ((AP(x) & F_INS) && x.qri > 0)
                relax(x, Cell(x.hi, x.qri - 1, x.ti), F_INS, is_tp(x));
            if ((AP(x) & F_DEL) && x.ti > 0)
                relax(x, Cell(x.hi, x.qri, x.ti - 1), F_DEL, 0);
        }
        prev_wave.clear();
        if (fifo.empty()) { A.status |= VPR_ST_ERR_UNFINISHED; return; }  // would spin forever in the reference
    }
    A.beg_plane = (A.aln[0].at(0, 0) & F_PATH) ? VPR_PLANE_QUERY : VPR_PLANE_REF;  // dist.cpp:811-814
}

// ---------------------------------------------------------------------------
// forward walk + sync points, dist.cpp:865-998
// ---------------------------------------------------------------------------
void walk_path(const AlnIn &in, AlnState &A) {
    const int qi = 2 * in.i, ri = 2 * in.i + 1;
    const std::vector<int> &q2r = in.q->ptr, &qfl = in.q->flag;
    const std::vector<int> &t2r = in.t->ptr, &tfl = in.t->flag;
    const std::vector<int> &r2q = *in.r2q_ptr;
    const int r_size = r2q.size(), q_size = q2r.size(), t_size = t2r.size();

    std::vector<uint8_t> ref_has_ins(r_size, 0);  // dist.cpp:886-894
    for (int j = 0; j < q_size; j++)
        if (qfl[j] & VPR_PTR_INS_LOC) ref_has_ins[q2r[j]] = 1;
    for (int j = 0; j < t_size; j++)
        if (tfl[j] & VPR_PTR_INS_LOC) ref_has_ins[t2r[j]] = 1;

    int hi = (A.beg_plane == VPR_PLANE_QUERY) ? qi : ri, qri = 0, ti = 0;
    A.path.clear(); A.sync.clear(); A.edit.clear();
    A.sync.push_back(1);
    A.edit.push_back(0);
    A.path.push_back(Cell(hi, qri, ti));
    auto PP = [&](int h, int q, int t) -> uint8_t { return A.pptr[h == ri].at(q, t); };

    while ((hi == ri && qri < r_size - 1) || (hi == qi && qri < q_size - 1) || ti < t_size - 1) {
        int mv;
        const uint8_t p = PP(hi, qri, ti);
        if (hi == ri && (p & F_SWP)) {           // prefer moving onto the QUERY plane
            mv = F_SWP; hi = qi; qri = r2q[qri]; qri++; ti++; A.edit.push_back(0);
        } else if (p & F_MAT) { mv = F_MAT; qri++; ti++; A.edit.push_back(0);
        } else if (p & F_SUB) { mv = F_SUB; qri++; ti++; A.edit.push_back(1);
        } else if (p & F_INS) { mv = F_INS; qri++; A.edit.push_back(1);
        } else if (p & F_DEL) { mv = F_DEL; ti++; A.edit.push_back(1);
        } else if (hi == qi && (p & F_SWP)) {    // last choice: leave the QUERY plane
            mv = F_SWP; hi = ri; qri = q2r[qri]; qri++; ti++; A.edit.push_back(0);
        } else { A.status |= VPR_ST_ERR_NO_PTR; return; }  // dist.cpp:937

        if ((hi == qi && qri >= q_size) || (hi == ri && qri >= r_size) || ti >= t_size) break;
        A.path.push_back(Cell(hi, qri, ti));

        const int consumes_ref = mv & (F_MAT | F_SWP | F_SUB | F_DEL);  // dist.cpp:950
        bool in_truth_var = tfl[ti] & VPR_PTR_VARIANT;
        if (consumes_ref) in_truth_var = in_truth_var && !(tfl[ti] & VPR_PTR_VAR_BEG);
        bool in_query_var = (hi == ri) ? false : bool(qfl[qri] & VPR_PTR_VARIANT);
        if (hi == qi && consumes_ref) in_query_var = in_query_var && !(qfl[qri] & VPR_PTR_VAR_BEG);
        const bool is_ins_loc = ref_has_ins[t2r[ti]] || (hi == ri ? ref_has_ins[qri] : ref_has_ins[q2r[qri]]);
        const bool is_sync = !in_truth_var && !in_query_var && !is_ins_loc &&
                             t2r[ti] == (hi == ri ? qri : q2r[qri]) && (mv & (F_MAT | F_SWP | F_SUB));
        A.sync.push_back(is_sync);
    }
    A.sync.push_back(1);  // last position is a sync point, dist.cpp:995-997
    A.edit.push_back(0);
}

// ---------------------------------------------------------------------------
// wf_ed, dist.cpp:1406-1506 (score only; the caller never reads offs/ptrs)
// ---------------------------------------------------------------------------
int edit_distance(const char *query, int query_len, const char *truth, int truth_len) {
    if (!query_len) return truth_len;
    if (!truth_len) return query_len;
    int s = 0;
    const int mat_len = query_len + truth_len - 1;
    std::vector<int> cur(mat_len, -2), nxt;
    cur[query_len - 1] = -1;
    while (true) {
        bool done = false;
        for (int d = 0; d < mat_len; d++) {  // extend, dist.cpp:1429-1455
            int off = cur[d];
            const int diag = d + 1 - query_len;
            if (off == -2) continue;
            if (diag + off + 1 < 0) continue;
            if (off > query_len - 1) continue;
            if (diag + off > truth_len - 1) continue;
            while (off < query_len - 1 && diag + off < truth_len - 1) {
                if (query[off + 1] == truth[diag + off + 1]) off++;
                else break;
            }
            cur[d] = off;
            if (off == query_len - 1 && off + diag == truth_len - 1) { done = true; break; }
        }
        if (done) break;
        nxt.assign(mat_len, -2);  // next wavefront, dist.cpp:1466-1504
        s++;
        for (int d = 0; d < mat_len; d++) {
            const int diag = d + 1 - query_len;
            if (cur[d] != -2 && cur[d] + 1 < query_len && diag + cur[d] + 1 < truth_len &&
                cur[d] + 1 >= nxt[d])
                nxt[d] = cur[d] + 1;
            if (d > 0 && cur[d - 1] != -2 && diag + cur[d - 1] < truth_len && cur[d - 1] >= nxt[d])
                nxt[d] = cur[d - 1];
            if (d < mat_len - 1 && cur[d + 1] != -2 && cur[d + 1] + 1 < query_len &&
                diag + cur[d + 1] + 1 < truth_len && diag + cur[d + 1] + 1 >= -1 &&
                cur[d + 1] + 1 >= nxt[d])
                nxt[d] = cur[d + 1] + 1;
        }
        cur.swap(nxt);
    }
    return s;
}

// per-variant result slots of one (hap slot, swap)
struct VarOut {
    uint8_t *errtype; int32_t *sync_group; float *credit; int32_t *ref_ed; int32_t *query_ed; float *callq;
};

// ---------------------------------------------------------------------------
// credit assignment, dist.cpp:1035-1400
// ---------------------------------------------------------------------------
void assign_credit(const AlnIn &in, AlnState &A,
                   const int32_t *qv_pos, const float *qv_qual, int64_t q_beg, int64_t q_end, VarOut qo,
                   const int32_t *tv_pos, int64_t t_beg, int64_t t_end, VarOut to,
                   const vpr_config &cfg) {
    const int ri = 2 * in.i + 1;
    const std::vector<int> &q2r = in.q->ptr, &t2r = in.t->ptr;
    const std::string &R = *in.ref, &T = in.t->seq;

    int sync_group = 0;
    const int ti_size = t2r.size();
    int hi = 2 * in.i + A.end_plane;
    const int qri_size = (A.end_plane == VPR_PLANE_QUERY) ? q2r.size() : in.r2q_ptr->size();
    int prev_hi = hi;
    int qri = qri_size - 1;
    int sync_ref_idx = in.r2q_ptr->size();
    int prev_sync_ref_idx = sync_ref_idx;
    int prev_qri = qri;
    int ti = ti_size - 1;
    int prev_ti = ti;
    int sync_truth_idx = ti_size;
    int prev_sync_truth_idx = ti_size;
    int query_ed = 0;
    int64_t query_var_ptr = q_end - 1;
    // (the reference's `poss.size() && ptr >= 0 ? ... : 0` with contig-global indices;
    //  here indices are batch-global, an empty range gives 0 the same way when q_end==0,
    //  and otherwise the loop condition `query_var_ptr >= query_beg_idx` guards it)
    int query_var_pos = (query_var_ptr >= q_beg) ? qv_pos[query_var_ptr] : 0;
    int64_t prev_query_var_ptr = query_var_ptr;
    int64_t truth_var_ptr = t_end - 1;
    int truth_var_pos = (truth_var_ptr >= t_beg) ? tv_pos[truth_var_ptr] : 0;
    int64_t prev_truth_var_ptr = truth_var_ptr;
    int sync_idx = int(A.sync.size()) - 1;
    (void)qri; (void)ti;

    while (sync_idx >= 0) {
        const int query_ref_pos = (prev_hi == ri) ? prev_qri : q2r[prev_qri];
        while (query_ref_pos < query_var_pos && query_var_ptr >= q_beg) {  // passed a query variant
            if (hi == ri) {  // passed it on the REF plane: false positive, dist.cpp:1157-1168
                qo.errtype[query_var_ptr] = VPR_ERRTYPE_FP;
                qo.sync_group[query_var_ptr] = sync_group++;
                qo.credit[query_var_ptr] = 0;
                qo.ref_ed[query_var_ptr] = 0;
                qo.query_ed[query_var_ptr] = 0;
                qo.callq[query_var_ptr] = qv_qual[query_var_ptr];
            }
            query_var_ptr--;
            query_var_pos = (query_var_ptr < q_beg) ? -1 : qv_pos[query_var_ptr];
        }
        const int truth_ref_pos = t2r[prev_ti];
        while (truth_ref_pos < truth_var_pos && truth_var_ptr >= t_beg) {
            truth_var_ptr--;
            truth_var_pos = (truth_var_ptr < t_beg) ? -1 : tv_pos[truth_var_ptr];
        }

        if (A.sync[sync_idx]) {  // dist.cpp:1190
            sync_ref_idx = (prev_hi == ri) ? prev_qri + 1 : q2r[prev_qri] + 1;
            sync_truth_idx = prev_ti + 1;
            // std::string::substr clamps the count to the end of the string
            int rl = prev_sync_ref_idx - sync_ref_idx, tl = prev_sync_truth_idx - sync_truth_idx;
            // (a negative count converts to a huge size_t there, i.e. "rest of the string")
            if (rl < 0 || rl > int(R.size()) - sync_ref_idx) rl = int(R.size()) - sync_ref_idx;
            if (tl < 0 || tl > int(T.size()) - sync_truth_idx) tl = int(T.size()) - sync_truth_idx;
            int ref_ed = edit_distance(R.data() + sync_ref_idx, rl, T.data() + sync_truth_idx, tl);

            if (prev_truth_var_ptr == truth_var_ptr && ref_ed != 0) A.status |= VPR_ST_WARN_REF_ED;
            if (prev_query_var_ptr == query_var_ptr && query_ed != ref_ed) A.status |= VPR_ST_WARN_QUERY_ED;
            if (query_ed > ref_ed) A.status |= VPR_ST_WARN_EXCEEDS;
            if (ref_ed == 0 && truth_var_ptr != prev_truth_var_ptr) {  // dist.cpp:1219-1223
                A.status |= VPR_ST_WARN_ZERO_ED;
                ref_ed = 1;
            }

            float callq = cfg.max_qual;  // dist.cpp:1284-1288
            for (int64_t v = prev_query_var_ptr; v > query_var_ptr; v--) callq = std::min(callq, qv_qual[v]);

            for (int64_t v = prev_query_var_ptr; v > query_var_ptr; v--) {
                float credit = 1 - float(query_ed) / ref_ed;
                if (qo.errtype[v] == VPR_ERRTYPE_UN) {  // don't overwrite FPs
                    qo.errtype[v] = (credit >= cfg.credit_threshold) ? VPR_ERRTYPE_TP : VPR_ERRTYPE_FP;
                    qo.sync_group[v] = sync_group;
                    qo.credit[v] = credit;
                    qo.ref_ed[v] = ref_ed;
                    qo.query_ed[v] = query_ed;
                    qo.callq[v] = callq;
                }
            }
            for (int64_t v = prev_truth_var_ptr; v > truth_var_ptr; v--) {
                float credit = 1 - float(query_ed) / ref_ed;
                const bool tp = credit >= cfg.credit_threshold;
                to.errtype[v] = tp ? VPR_ERRTYPE_TP : VPR_ERRTYPE_FN;
                to.sync_group[v] = sync_group;
                to.credit[v] = credit;
                to.ref_ed[v] = ref_ed;
                to.query_ed[v] = query_ed;
                to.callq[v] = tp ? callq : cfg.max_qual;
            }
            if (query_var_ptr != prev_query_var_ptr || truth_var_ptr != prev_truth_var_ptr) sync_group++;
            prev_query_var_ptr = query_var_ptr;
            prev_truth_var_ptr = truth_var_ptr;
            prev_sync_ref_idx = sync_ref_idx;
            prev_sync_truth_idx = sync_truth_idx;
            query_ed = 0;
        }

        query_ed += A.edit[sync_idx];
        sync_idx--;
        if (sync_idx < 0) break;
        qri = prev_qri; ti = prev_ti; hi = prev_hi;
        prev_qri = A.path[sync_idx].qri;
        prev_ti = A.path[sync_idx].ti;
        prev_hi = A.path[sync_idx].hi;
        query_var_pos = (query_var_ptr < q_beg) ? -1 : qv_pos[query_var_ptr];
        truth_var_pos = (truth_var_ptr < t_beg) ? -1 : tv_pos[truth_var_ptr];
    }
}

// ---------------------------------------------------------------------------
// generate_ptrs_strs, dist.cpp:145-242 (min_qual = 0 on this path)
// ---------------------------------------------------------------------------
struct GenOut {
    std::string hap, ref;
    std::vector<int32_t> hap_ptr, ref_ptr;
    std::vector<uint8_t> hap_flag, ref_flag;
};

int gen_hap(const vpr_variants *v, int slot, int sc, GenOut &o) {
    const int ctg = v->sc_ctg[sc];
    const uint8_t *fa = v->ctg_seq + v->ctg_off[ctg];
    const int64_t ctg_len = v->ctg_off[ctg + 1] - v->ctg_off[ctg];
    // (end_pos cut at the contig's last base: where cluster.cpp:591-595's pos + rlen + 1 leaves the contig the reference has
    // no defined result -- dist.cpp:232 substr comes back short of the pointer arrays, dist.cpp:539 indexes by their size --
    // and product and oracle both evaluate the region that exists; include/vcfdist_pr.h)
    const int beg_pos = v->sc_beg[sc], end_pos = int(std::min<int64_t>(v->sc_end[sc], ctg_len - 1));
    int64_t var = v->var_off[slot][sc];
    const int64_t var_end = v->var_off[slot][sc + 1];
    if (beg_pos < 0 || beg_pos > end_pos) return -2;
    int ref_pos = beg_pos;
    for (; ref_pos <= end_pos;) {
        if (var < var_end && ref_pos == v->var_pos[slot][var]) {
            const uint8_t *pool = v->allele_pool[slot];
            const int64_t r0 = v->var_ref_off[slot][var], r1 = r0 + v->var_ref_len[slot][var];
            const int64_t a0 = v->var_alt_off[slot][var], a1 = a0 + v->var_alt_len[slot][var];
            switch (v->var_type[slot][var]) {
                case VPR_TYPE_INS: {
                    const int k = a1 - a0;
                    const size_t first = o.hap_flag.size();
                    o.hap_ptr.insert(o.hap_ptr.end(), k, int32_t(o.ref.size()) - 1);
                    o.hap_flag.insert(o.hap_flag.end(), k, VPR_PTR_VARIANT);
                    o.hap_flag[o.hap_flag.size() - 1] |= VPR_PTR_VAR_END;
                    o.hap_flag[first] |= VPR_PTR_VAR_BEG | VPR_PTR_INS_LOC;
                    o.hap.append((const char *)pool + a0, k);
                    break;
                }
                case VPR_TYPE_DEL: {
                    const int k = r1 - r0;
                    const size_t first = o.ref_flag.size();
                    o.ref_ptr.insert(o.ref_ptr.end(), k, int32_t(o.hap.size()) - 1);
                    o.ref_flag.insert(o.ref_flag.end(), k, VPR_PTR_VARIANT);
                    o.ref_flag[o.ref_flag.size() - 1] |= VPR_PTR_VAR_END;
                    o.ref_flag[first] |= VPR_PTR_VAR_BEG;
                    o.ref.append((const char *)pool + r0, k);
                    ref_pos += k;
                    break;
                }
                case VPR_TYPE_SUB:
                    o.ref_ptr.push_back(o.hap.size());
                    o.ref_flag.push_back(VPR_PTR_VARIANT | VPR_PTR_VAR_BEG | VPR_PTR_VAR_END);
                    o.hap_ptr.push_back(o.ref.size());
                    o.hap_flag.push_back(VPR_PTR_VARIANT | VPR_PTR_VAR_BEG | VPR_PTR_VAR_END);
                    o.ref.append((const char *)pool + r0, r1 - r0);
                    o.hap.append((const char *)pool + a0, a1 - a0);
                    ref_pos++;
                    break;
                default:
                    return -1;  // ERROR("Unexpected variant type"), dist.cpp:199-201
            }
            var++;
        } else {
            const int ref_end = (var < var_end) ? v->var_pos[slot][var] : end_pos + 1;
            if (ref_end < ref_pos || ref_end > ctg_len) return -2;
            for (int p = ref_pos; p < ref_end; p++) {
                o.hap_ptr.push_back(o.ref.size());
                o.hap_flag.push_back(0);
                o.ref_ptr.push_back(o.hap.size());
                o.ref_flag.push_back(0);
                o.hap.push_back(fa[p]);
                o.ref.push_back(fa[p]);
            }
            ref_pos = ref_end;
        }
    }
    if (ref_pos != end_pos + 1 || var != var_end) return -2;   // a variant that leaves the region / the contig
    return 0;
}

void load_hap(const vpr_batch *b, int slot, int sc, Hap &h) {
    const int64_t o0 = b->hap_off[slot][sc], o1 = b->hap_off[slot][sc + 1];
    h.seq.assign((const char *)b->hap_seq[slot] + o0, o1 - o0);
    h.ptr.assign(b->hap_ptr[slot] + o0, b->hap_ptr[slot] + o1);
    h.flag.assign(b->hap_flag[slot] + o0, b->hap_flag[slot] + o1);
}

}  // namespace

// ===========================================================================
// C entry points (ctypes)
// ===========================================================================
extern "C" {

int vpo_edit_distance(const uint8_t *a, int32_t la, const uint8_t *b, int32_t lb) {
    return edit_distance((const char *)a, la, (const char *)b, lb);
}

int32_t vpo_store_phase(const int32_t s[4], double phase_threshold, int32_t *orig, int32_t *swp) {
    // dist.cpp:456-469
    const int orig_phase_dist = s[0] + s[3];
    const int swap_phase_dist = s[2] + s[1];
    int phase = VPR_PHASE_NONE;
    if (orig_phase_dist == swap_phase_dist) phase = VPR_PHASE_NONE;
    else if (orig_phase_dist == 0) phase = VPR_PHASE_ORIG;
    else if (swap_phase_dist == 0) phase = VPR_PHASE_SWAP;
    else if (1 - float(swap_phase_dist) / orig_phase_dist > phase_threshold) phase = VPR_PHASE_SWAP;
    else if (1 - float(orig_phase_dist) / swap_phase_dist > phase_threshold) phase = VPR_PHASE_ORIG;
    *orig = orig_phase_dist;
    *swp = swap_phase_dist;
    return phase;
}

// Run the whole path over a Level A batch.  `res` arrays must be pre-initialised
// by the caller to the reference's initial values (errtype = UN, rest 0).
int vpo_run(const vpr_batch *b, const vpr_config *cfg, vpr_results *res, vpo_extra *ex) {
    const bool timing = getenv("VPO_TIMING") != nullptr;
    double tm[5] = {0, 0, 0, 0, 0};
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    for (int sc = 0; sc < b->n_sc; sc++) {
        Hap hap[VPR_HAPS];
        for (int h = 0; h < VPR_HAPS; h++) load_hap(b, h, sc, hap[h]);
        const int64_t r0 = b->ref_off[sc], r1 = b->ref_off[sc + 1];
        std::string ref((const char *)b->ref_seq + r0, r1 - r0);
        std::vector<int> r2q_ptr[2], r2q_flag[2];
        for (int h = 0; h < 2; h++) {
            r2q_ptr[h].assign(b->ref_ptr[h] + r0, b->ref_ptr[h] + r1);
            r2q_flag[h].assign(b->ref_flag[h] + r0, b->ref_flag[h] + r1);
        }
        int32_t s[4];
        std::vector<std::unique_ptr<AlnState>> st(4);
        for (int i = 0; i < 4; i++) {
            AlnIn in{i, &hap[i >> 1], &hap[2 + (i & 1)], &ref, &r2q_ptr[i >> 1], &r2q_flag[i >> 1]};
            st[i].reset(new AlnState());
            AlnState &A = *st[i];
            const double t0_ = timing ? now() : 0;
            forward_pass(in, A, ex != nullptr);
            if (timing) tm[0] += now() - t0_;
            s[i] = A.s;
            res->aln_dist[sc * 4 + i] = A.s;
            res->aln_end_plane[sc * 4 + i] = A.end_plane;
        }
        res->sc_phase[sc] = vpo_store_phase(s, cfg->phase_threshold, &res->orig_phase_dist[sc],
                                            &res->swap_phase_dist[sc]);
        for (int i = 0; i < 4; i++) {
            AlnIn in{i, &hap[i >> 1], &hap[2 + (i & 1)], &ref, &r2q_ptr[i >> 1], &r2q_flag[i >> 1]};
            AlnState &A = *st[i];
            double t1_ = timing ? now() : 0;
            if (!(A.status & VPR_ST_ERR_UNFINISHED)) backward_pass(in, A);
            if (timing) { tm[1] += now() - t1_; t1_ = now(); }
            if (!(A.status & (VPR_ST_ERR_UNFINISHED | VPR_ST_ERR_NO_PTR))) walk_path(in, A);
            if (timing) { tm[2] += now() - t1_; t1_ = now(); }
            res->aln_beg_plane[sc * 4 + i] = A.beg_plane;
            if (!(A.status & (VPR_ST_ERR_UNFINISHED | VPR_ST_ERR_NO_PTR))) {
                const int swap = (i == 1 || i == 2);
                const int qs = i >> 1, ts = 2 + (i & 1);
                VarOut qo{res->errtype[qs][swap], res->sync_group[qs][swap], res->credit[qs][swap],
                          res->ref_ed[qs][swap], res->query_ed[qs][swap], res->callq[qs][swap]};
                VarOut to{res->errtype[ts][swap], res->sync_group[ts][swap], res->credit[ts][swap],
                          res->ref_ed[ts][swap], res->query_ed[ts][swap], res->callq[ts][swap]};
                assign_credit(in, A, b->var_pos[qs], b->var_qual[qs], b->var_off[qs][sc], b->var_off[qs][sc + 1], qo,
                              b->var_pos[ts], b->var_off[ts][sc], b->var_off[ts][sc + 1], to, *cfg);
            }
            if (A.st.used_conflict) A.status |= VPR_ST_SWAP_TIE;
            res->aln_status[sc * 4 + i] = A.status;
            if (ex) {
                const int a = sc * 4 + i;
                if (ex->swap_writes) ex->swap_writes[a] = A.st.writes;
                if (ex->swap_conflict_writes) ex->swap_conflict_writes[a] = A.st.conflict_writes;
                if (ex->swap_used_conflict) ex->swap_used_conflict[a] = A.st.used_conflict;
                if (ex->swap_used_conflict_nonmax) ex->swap_used_conflict_nonmax[a] = A.st.used_conflict_nonmax;
                if (ex->path_len) ex->path_len[a] = A.path.size();
                if (ex->want_sc == sc && ex->want_aln == i && ex->path_plane) {
                    const int64_t n = std::min<int64_t>(ex->path_cap, A.path.size());
                    for (int64_t k = 0; k < n; k++) {
                        ex->path_plane[k] = A.path[k].hi & 1;
                        ex->path_qri[k] = A.path[k].qri;
                        ex->path_ti[k] = A.path[k].ti;
                    }
                    // sync/edit have one more entry than path (the final forced sync)
                    const int64_t m = std::min<int64_t>(ex->path_cap + 1, A.sync.size());
                    for (int64_t k = 0; k < m; k++) {
                        ex->path_sync[k] = A.sync[k];
                        ex->path_edit[k] = A.edit[k];
                    }
                    ex->want_len = A.path.size();
                }
                if (ex->want_sc == sc && ex->want_aln == i && ex->dump_flags[0]) {
                    // forward flags (masked to the 5 edge bits) and path_ptrs, [plane][q][t]
                    for (int p = 0; p < 2; p++) {
                        const size_t n = A.aln[p].size();
                        for (size_t k = 0; k < n; k++) {
                            ex->dump_flags[p][k] = A.aln[p].flat(k) & 31;
                            ex->dump_pptr[p][k] = A.pptr[p].flat(k) & 31;
                            ex->dump_pscore[p][k] = A.pscore[p].flat(k);
                        }
                    }
                }
            }
        }
    }
    if (timing) fprintf(stderr, "[vpo] forward %.3f s, backward %.3f s, walk %.3f s\n", tm[0], tm[1], tm[2]);
    return 0;
}

// ---- generation of Level A arrays from variants ---------------------------
struct vpo_generated {
    std::vector<int64_t> hap_off[VPR_HAPS], ref_off;
    std::vector<uint8_t> hap_seq[VPR_HAPS], hap_flag[VPR_HAPS], ref_seq, ref_flag[2];
    std::vector<int32_t> hap_ptr[VPR_HAPS], ref_ptr[2];
    int err = 0;
};

vpo_generated *vpo_gen_create(const vpr_variants *v) {
    vpo_generated *g = new vpo_generated();
    for (int h = 0; h < VPR_HAPS; h++) g->hap_off[h].push_back(0);
    g->ref_off.push_back(0);
    for (int sc = 0; sc < v->n_sc; sc++) {
        for (int h = 0; h < VPR_HAPS; h++) {
            GenOut o;
            int rc = gen_hap(v, h, sc, o);
            if (rc) g->err = rc;
            g->hap_seq[h].insert(g->hap_seq[h].end(), o.hap.begin(), o.hap.end());
            g->hap_ptr[h].insert(g->hap_ptr[h].end(), o.hap_ptr.begin(), o.hap_ptr.end());
            g->hap_flag[h].insert(g->hap_flag[h].end(), o.hap_flag.begin(), o.hap_flag.end());
            g->hap_off[h].push_back(g->hap_seq[h].size());
            if (h == 0) {  // the driver passes ref_q1 as *the* ref string, dist.cpp:1856,1868
                g->ref_seq.insert(g->ref_seq.end(), o.ref.begin(), o.ref.end());
                g->ref_off.push_back(g->ref_seq.size());
            }
            if (h < 2) {
                g->ref_ptr[h].insert(g->ref_ptr[h].end(), o.ref_ptr.begin(), o.ref_ptr.end());
                g->ref_flag[h].insert(g->ref_flag[h].end(), o.ref_flag.begin(), o.ref_flag.end());
                if ((int64_t)g->ref_ptr[h].size() != (int64_t)(h == 0 ? g->ref_seq.size() : g->ref_ptr[0].size()))
                    g->err = -3;  // ref strings of the haps differ: inconsistent input
            }
        }
    }
    return g;
}
int vpo_gen_error(const vpo_generated *g) { return g->err; }
void vpo_gen_sizes(const vpo_generated *g, int64_t hap_len[VPR_HAPS], int64_t *ref_len) {
    for (int h = 0; h < VPR_HAPS; h++) hap_len[h] = g->hap_seq[h].size();
    *ref_len = g->ref_seq.size();
}
void vpo_gen_copy(const vpo_generated *g, int64_t *hap_off[VPR_HAPS], uint8_t *hap_seq[VPR_HAPS],
                  int32_t *hap_ptr[VPR_HAPS], uint8_t *hap_flag[VPR_HAPS], int64_t *ref_off,
                  uint8_t *ref_seq, int32_t *ref_ptr[2], uint8_t *ref_flag[2]) {
    for (int h = 0; h < VPR_HAPS; h++) {
        memcpy(hap_off[h], g->hap_off[h].data(), g->hap_off[h].size() * 8);
        memcpy(hap_seq[h], g->hap_seq[h].data(), g->hap_seq[h].size());
        memcpy(hap_ptr[h], g->hap_ptr[h].data(), g->hap_ptr[h].size() * 4);
        memcpy(hap_flag[h], g->hap_flag[h].data(), g->hap_flag[h].size());
    }
    memcpy(ref_off, g->ref_off.data(), g->ref_off.size() * 8);
    memcpy(ref_seq, g->ref_seq.data(), g->ref_seq.size());
    for (int h = 0; h < 2; h++) {
        memcpy(ref_ptr[h], g->ref_ptr[h].data(), g->ref_ptr[h].size() * 4);
        memcpy(ref_flag[h], g->ref_flag[h].data(), g->ref_flag[h].size());
    }
}
void vpo_gen_free(vpo_generated *g) { delete g; }

}  // extern "C"

// TEST INFRASTRUCTURE.  The model of libstdc++'s std::unordered_set iteration order that vcfdist_amd/csrc/pr_tie.hip
// computes on the device (positions from "first insertion of the bucket" + "later insertions into the same bucket",
// one pass per rehash), checked against the real container with the reference's hash (dist.h:42-50) and clear() between
// waves (dist.cpp:425), and the bucket-count sequence the replay walks through.  Prints the sequence, then "bad = N".
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <unordered_set>
#include <vector>
struct idx1 { int hi, qri, ti; bool operator==(const idx1 &o) const { return hi == o.hi && qri == o.qri && ti == o.ti; } };
namespace std {
template <> struct hash<idx1> {
    std::uint64_t operator()(const idx1 &x) const noexcept {
        return (uint64_t(x.hi) * 73856093 + 0x517cc1b727220a95) ^ (uint64_t(x.qri) * 19349669 + 0xd15f392b3d4704a2) ^ (uint64_t(x.ti) * 83492791);
    }
};
}
static std::vector<uint64_t> seq;
static uint64_t next_b(uint64_t B) { for (uint64_t p : seq) if (p >= 2 * B) return p; return 0; }
static std::vector<int> order(const std::vector<int> &s, const std::vector<idx1> &cells, uint64_t B) {
    const int m = int(s.size());
    std::vector<uint64_t> bk(m);
    std::vector<int> first(B, -1), F(m), H(m + 1, 0), K(m, 0), G(m + 1, 0), out(m);
    for (int i = 0; i < m; i++) { bk[i] = std::hash<idx1>()(cells[s[i]]) % B; if (first[bk[i]] < 0) first[bk[i]] = i; }
    for (int i = 0; i < m; i++) { F[i] = first[bk[i]]; H[F[i]]++; }
    for (int f = m - 1; f >= 0; f--) G[f] = (f + 1 < m ? G[f + 1] + H[f + 1] : 0);
    for (int i = m - 1; i >= 0; i--) { out[G[F[i]] + K[F[i]]] = s[i]; K[F[i]]++; }
    return out;
}
int main() {
    {   // the sequence of bucket counts: the first allocation, then _M_next_bkt(2 * n)
        std::unordered_set<idx1> t; t.insert({0, 0, 0});
        uint64_t B = t.bucket_count(), last = 0;
        seq.push_back(B);
        while (B < (1ull << 22)) { std::unordered_set<idx1> u; u.rehash(2 * B); if (u.bucket_count() == last) break; B = last = u.bucket_count(); seq.push_back(B); }
        for (auto p : seq) printf("%llu ", (unsigned long long)p);
        printf("\n");
    }
    std::mt19937_64 rng(5);
    int bad = 0;
    for (int trial = 0; trial < 200; trial++) {
        std::unordered_set<idx1> real;
        uint64_t B = seq[0];
        const int nw = 1 + rng() % 10;
        for (int w = 0; w < nw; w++) {
            const int n = 1 + rng() % (trial < 80 ? 40 : (trial < 170 ? 400 : 4000));
            std::vector<idx1> cells; std::unordered_set<idx1> seen;
            while (int(cells.size()) < n) { idx1 c{int(rng() % 8), int(rng() % 300), int(rng() % 300)}; if (seen.insert(c).second) cells.push_back(c); }
            real.clear();
            for (auto &c : cells) real.insert(c);
            std::vector<idx1> it(real.begin(), real.end());
            std::vector<int> s, lam; int done = 0;
            while (true) {
                const int take = std::min(int(B) - done, n - done);
                s = lam; for (int k = 0; k < take; k++) s.push_back(done + k);
                lam = order(s, cells, B); done += take;
                if (done == n) break;
                B = next_b(B);
            }
            if (B != real.bucket_count()) bad++;
            for (int k = 0; k < n; k++) if (!(cells[lam[k]] == it[k])) { bad++; break; }
        }
    }
    printf("bad = %d\n", bad);
    return bad != 0;
}

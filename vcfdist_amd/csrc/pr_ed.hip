// pr_ed.hip -- K4b: the edit distance of a deferred sync section (wf_ed, dist.cpp:1406-1506 = the Levenshtein distance of the
// reference segment and the truth segment of the section, calc_prec_recall dist.cpp:1194-1199), BIT-PARALLEL.
//
// The sections that are deferred are the ones around SV-sized variants: two strings of thousands of bases that differ by
// blocks of hundreds to thousands.  The reference's own formulation (furthest-reaching points per diagonal, k_ed_wf) needs one
// step per unit of DISTANCE, each a barrier of a 1 024-thread workgroup over a band thousands of diagonals wide: 36 ms for the
// sections of sv_synth, on the tail of the step behind every walk.  The distance itself is plain dynamic programming, and
// Myers' bit-vector form of it (G. Myers, "A fast bit-vector algorithm for approximate string matching based on dynamic
// programming", JACM 46(3), 1999; the block form with horizontal carries between blocks, section 4) packs 64 rows of a DP
// column into two machine words of vertical differences: a column of a block costs ~25 word operations whatever the distance.
//
// Mapping: ONE WAVEFRONT per section.  One string is the pattern (the one that makes fewer steps, below): lane l owns block l of
// it (64 rows: the words Pv / Mv of vertical +1 / -1 differences and the match masks of its 64 characters), the other is the text,
// one column per step.  The blocks of a column depend on each other only through the horizontal difference at a block's last row (hout ->
// the next block's hin), so the wave runs a software pipeline down the lanes: in step s lane l advances column s - l, taking
// its hin and the column's character from lane l - 1's previous step (DPP wave shift).  n + 63 steps for a pattern of up to
// 4 096 characters; longer patterns take groups of 64 blocks one after the other, the last lane's hout of every column kept
// as two bit planes in LDS for the next group.  The distance is read off the pattern's last row: m, plus the horizontal
// differences of that row, summed by the lane that holds it.  Exact for arbitrary bytes (match masks of the four bases are
// kept; any other text byte builds its mask from the lane's pattern bytes on the spot).
#ifndef PR_ED_HIP_
#define PR_ED_HIP_

#define EDB_MAX_TEXT 65536      // columns whose carries fit the LDS bit planes (a longer text with a pattern of more than one group:
                                // the host keeps the older kernels for batches with such haplotypes)

__global__ void __launch_bounds__(64) k_ed_bits(DevBatch B, const AlnDesc *__restrict__ descs, const EdJob *__restrict__ jobs,
                                                int n_jobs, Section *__restrict__ secs) {
    __shared__ uint32_t planeP[EDB_MAX_TEXT / 32], planeM[EDB_MAX_TEXT / 32];      // hout = +1 / -1 of the group before, per column
    const int j = blockIdx.x;
    if (j >= n_jobs) return;
    const EdJob J = jobs[j];
    const AlnDesc d = descs[J.aln];
    const int lane = threadIdx.x;
    const uint8_t *X = B.ref_seq + d.r_off + J.ref_beg;
    const uint8_t *Y = B.hap_seq[d.ts] + d.t_off + J.tru_beg;
    int nx = J.ref_len, ny = J.tru_len;
    // common prefix and suffix (free, and most of a deferred section): 64 positions per step
    {
        const int mm = min(nx, ny);
        int pre = 0;
        while (pre < mm) {
            const int i = pre + lane;
            const bool diff = i < mm && X[i] != Y[i];
            const unsigned long long b = __ballot(diff);
            if (b) { pre += int(__builtin_ctzll(b)); break; }
            pre = min(pre + 64, mm);
        }
        X += pre; Y += pre; nx -= pre; ny -= pre;
        const int m2 = min(nx, ny);
        int suf = 0;
        while (suf < m2) {
            const int i = suf + lane;
            const bool diff = i < m2 && X[nx - 1 - i] != Y[ny - 1 - i];
            const unsigned long long b = __ballot(diff);
            if (b) { suf += int(__builtin_ctzll(b)); break; }
            suf = min(suf + 64, m2);
        }
        nx -= suf; ny -= suf;
    }
    // Which string is the pattern: the distance is symmetric, the pipeline is not -- a group of up to 64 pattern blocks takes
    // (text length + its blocks - 1) steps, so a pattern of a characters against a text of b costs groups(a) * (b - 1) + blocks(a)
    // steps.  Nearly always the LONGER string as the pattern wins (a 9 980-base segment against 23 bases: three groups of 86 steps
    // instead of one of 10 042; until the end of round 6 the shorter string was the pattern whatever the lengths).
    auto steps = [](int a, int b) -> long long { const int nb = (a + 63) >> 6; return (long long)((nb + 63) >> 6) * (b - 1) + nb; };
    const uint8_t *P = X, *T = Y;
    int m = nx, n = ny;
    if (nx == 0 || (ny > 0 && steps(ny, nx) < steps(nx, ny))) { P = Y; T = X; m = ny; n = nx; }
    int dist;
    if (m == 0 || n == 0) {
        dist = max(m, n);
    } else {
        const int n_blocks = (m + 63) >> 6, n_groups = (n_blocks + 63) >> 6;
        const int last_block = (m - 1) >> 6, last_bit = (m - 1) & 63;
        int score = m;                              // D[m][0]; the lane of the pattern's last row adds that row's differences
        for (int g = 0; g < n_groups; g++) {
            const int blk = g * 64 + lane;
            const bool has_blk = blk < n_blocks;
            // this lane's 64 pattern bytes, and the match masks of the four bases
            uint32_t pw[16];
            unsigned long long eqA = 0, eqC = 0, eqG = 0, eqT = 0;
#pragma unroll
            for (int w = 0; w < 16; w++) {
                uint32_t v = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int i = blk * 64 + w * 4 + k;
                    const uint32_t c = (has_blk && i < m) ? uint32_t(P[i]) : 0u;      // (0: matches no text byte -- see `other` below)
                    v |= c << (8 * k);
                    const unsigned long long bit = 1ull << (w * 4 + k);
                    eqA |= c == 'A' ? bit : 0ull; eqC |= c == 'C' ? bit : 0ull; eqG |= c == 'G' ? bit : 0ull; eqT |= c == 'T' ? bit : 0ull;
                }
                pw[w] = v;
            }
            unsigned long long Pv = ~0ull, Mv = 0ull;
            const bool tap = has_blk && blk == last_block;
            const bool keep = g + 1 < n_groups;      // the last lane's hout feeds the next group
            uint32_t accP = 0, accM = 0;            // lane 63: 32 columns of hout, then one LDS word each
            uint32_t tchunk = 0;                    // text bytes (s & ~63) + lane
            int hout_prev = 0;                      // this lane's hout of the previous step
            uint32_t c_prev = 0;                    // ... and the character it worked on
            const int n_steps = n + min(n_blocks - g * 64, 64) - 1;
            for (int s = 0; s < n_steps; s++) {
                if ((s & 63) == 0) { const int i = s + lane; tchunk = i < n ? uint32_t(T[i]) : 0u; }
                // what flows down the pipeline: lane l takes lane l - 1's previous step
                int hin = wave_shr1(hout_prev, 0);
                uint32_t c = uint32_t(wave_shr1(int(c_prev), 0));
                if (lane == 0) {
                    c = uint32_t(__builtin_amdgcn_readlane(int(tchunk), s & 63));
                    hin = 1;                        // row 0 of the matrix: D[0][j] = j
                    if (g > 0 && s < n) hin = int((planeP[s >> 5] >> (s & 31)) & 1u) - int((planeM[s >> 5] >> (s & 31)) & 1u);
                }
                const int col = s - lane;
                const bool act = has_blk && col >= 0 && col < n;
                unsigned long long Eq = c == 'A' ? eqA : c == 'C' ? eqC : c == 'G' ? eqG : c == 'T' ? eqT : 0ull;
                const bool other = act && !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
                if (__any(other)) {
                    if (other) {                    // any other byte: its mask from the lane's pattern bytes (a padding row holds 0;
                        unsigned long long e = 0;   // a text byte 0 would match it, but those rows lie below the pattern's last)
#pragma unroll
                        for (int w = 0; w < 16; w++)
#pragma unroll
                            for (int k = 0; k < 4; k++) e |= ((pw[w] >> (8 * k)) & 0xffu) == c ? 1ull << (w * 4 + k) : 0ull;
                        Eq = e;
                    }
                }
                // Myers' block step
                const unsigned long long hneg = hin < 0 ? 1ull : 0ull, hpos = hin > 0 ? 1ull : 0ull;
                const unsigned long long Xv = Eq | Mv;
                const unsigned long long Eq1 = Eq | hneg;
                const unsigned long long Xh = (((Eq1 & Pv) + Pv) ^ Pv) | Eq1;
                unsigned long long Ph = Mv | ~(Xh | Pv);
                unsigned long long Mh = Pv & Xh;
                const int hout = int(Ph >> 63) - int(Mh >> 63);
                if (tap && act) score += int((Ph >> last_bit) & 1ull) - int((Mh >> last_bit) & 1ull);
                Ph = (Ph << 1) | hpos;
                Mh = (Mh << 1) | hneg;
                const unsigned long long Pv2 = Mh | ~(Xv | Ph), Mv2 = Ph & Xv;
                Pv = act ? Pv2 : Pv;
                Mv = act ? Mv2 : Mv;
                hout_prev = act ? hout : 0;
                c_prev = c;
                if (keep && lane == 63 && act) {
                    accP |= (hout > 0 ? 1u : 0u) << (col & 31);
                    accM |= (hout < 0 ? 1u : 0u) << (col & 31);
                    if ((col & 31) == 31 || col == n - 1) { planeP[col >> 5] = accP; planeM[col >> 5] = accM; accP = 0; accM = 0; }
                }
            }
            __syncthreads();        // (one wave: orders the plane words of this group before the next group's reads)
        }
        dist = __builtin_amdgcn_readlane(score, last_block & 63);
    }
    if (lane == 0) {
        Section &S = secs[d.sec_off + J.sec];
        S.ref_ed = dist;
        S.flags &= ~SEC_DEFERRED;
    }
}

#endif

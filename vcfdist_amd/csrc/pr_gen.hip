// pr_gen.hip -- generate_ptrs_strs (dist.cpp:145-242) on the device: the haplotype strings, the reference string and the
// pointer / flag arrays between them, written straight into the resident Level A arrays from the variant tables and
// the contig sequence.  vpr_upload_variants then moves ~60 bytes per supercluster over the link (variant records and
// region bounds) instead of the ~1.5 KB of marshalled arrays; the host keeps the pass that sizes and checks every
// region (generate.cpp), because the planner needs the lengths anyway.
//
// One wavefront per (supercluster, hap slot), slots in the slow grid dimension (a thread per supercluster wrote bytes into 64
// different regions per wave store: 5.6 ms per million superclusters).  Same semantics as the host walk (generate.cpp):
// region [beg, end] inclusive and cut at the contig's last base; an INS of k bases is k hap positions pointing at the
// reference base in front of it (first: VAR_BEG | INS_LOC, last: VAR_END); a DEL of k bases is k reference positions
// pointing at the hap base in front (first VAR_BEG, last VAR_END); a SUB is one position on both sides with all three
// flags; everything else is matching bases pointing at each other.  The reference string comes from hap slot 0's pass
// (the driver hands ref_q1 to every alignment, dist.cpp:1856,1868), the ref -> hap arrays from the two query slots.
#ifndef PR_GEN_HIP_
#define PR_GEN_HIP_

struct GenTables {          // device copies of the vpr_variants arrays the generator reads
    const int64_t *ctg_off;
    const uint8_t *ctg_seq;
    const int32_t *sc_ctg, *sc_beg, *sc_end;
    const int32_t *var_pos_rel[4];      // relative to the region start (the resident DevBatch::var_pos)
    const uint8_t *var_type[4];
    const int64_t *ref_off[4], *alt_off[4];
    const int32_t *ref_len[4], *alt_len[4];
    const uint8_t *pool[4];
};

struct GenOut {
    uint8_t *hap_seq[4]; int32_t *hap_ptr[4]; uint8_t *hap_flag[4];
    uint8_t *ref_seq; int32_t *ref_ptr[2]; uint8_t *ref_flag[2];
};

#define GEN_SC_PER_WAVE 4
__global__ void __launch_bounds__(256) k_generate(DevBatch B, GenTables G, GenOut O) {
    // One WAVEFRONT per (supercluster, hap slot), GEN_SC_PER_WAVE superclusters after each other: the pieces of a region (runs of
    // matching bases between the variants, the variants) are walked by all lanes together and every piece is written by the 64
    // lanes side by side -- whole lines instead of 64 threads each writing bytes into a region of its own.
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int slot = blockIdx.y;
    const int sc0 = (blockIdx.x * 4 + wave) * GEN_SC_PER_WAVE;
    for (int q = 0; q < GEN_SC_PER_WAVE; q++) {
        const int sc = sc0 + q;
        if (sc >= B.n_sc) return;
        const int ctg = G.sc_ctg[sc];
        const uint8_t *fa = G.ctg_seq + G.ctg_off[ctg];
        const int64_t ctg_len = G.ctg_off[ctg + 1] - G.ctg_off[ctg];
        const int beg = G.sc_beg[sc];
        const int end = int(min(int64_t(G.sc_end[sc]), ctg_len - 1));
        int64_t var = B.var_off[slot][sc];
        const int64_t var_end = B.var_off[slot][sc + 1];
        const int64_t ho = B.hap_off[slot][sc], ro = B.ref_off[sc];
        uint8_t *hseq = O.hap_seq[slot] + ho, *hflag = O.hap_flag[slot] + ho;
        int32_t *hptr = O.hap_ptr[slot] + ho;
        uint8_t *rseq = slot == 0 ? O.ref_seq + ro : nullptr;
        int32_t *rptr = slot < 2 ? O.ref_ptr[slot] + ro : nullptr;
        uint8_t *rflag = slot < 2 ? O.ref_flag[slot] + ro : nullptr;
        const uint8_t *pool = G.pool[slot];
        int nh = 0, nr = 0, pos = beg;
        while (pos <= end) {          // (uniform over the wave)
            if (var < var_end && beg + G.var_pos_rel[slot][var] == pos) {
                const int type = G.var_type[slot][var];
                const int rl = G.ref_len[slot][var], al = G.alt_len[slot][var];
                const uint8_t *ra = pool + G.ref_off[slot][var], *aa = pool + G.alt_off[slot][var];
                if (type == VPR_TYPE_INS) {
                    for (int k = lane; k < al; k += 64) {
                        hseq[nh + k] = aa[k];
                        hptr[nh + k] = nr - 1;
                        hflag[nh + k] = uint8_t(PV | (k == 0 ? (PB | PI) : 0) | (k == al - 1 ? PE : 0));
                    }
                    nh += al;
                } else if (type == VPR_TYPE_DEL) {
                    for (int k = lane; k < rl; k += 64) {
                        if (rseq) rseq[nr + k] = ra[k];
                        if (rptr) {
                            rptr[nr + k] = nh - 1;
                            rflag[nr + k] = uint8_t(PV | (k == 0 ? PB : 0) | (k == rl - 1 ? PE : 0));
                        }
                    }
                    nr += rl;
                    pos += rl;
                } else {        // SUB (the host pass has refused anything else)
                    if (lane == 0) {
                        hseq[nh] = aa[0];
                        hptr[nh] = nr;
                        hflag[nh] = uint8_t(PV | PB | PE);
                        if (rseq) rseq[nr] = ra[0];
                        if (rptr) { rptr[nr] = nh; rflag[nr] = uint8_t(PV | PB | PE); }
                    }
                    nh++; nr++; pos++;
                }
                var++;
            } else {
                const int stop = var < var_end ? beg + G.var_pos_rel[slot][var] : end + 1;
                const int n = stop - pos;
                for (int k = lane; k < n; k += 64) {
                    const uint8_t b = fa[pos + k];
                    hseq[nh + k] = b;
                    hptr[nh + k] = nr + k;
                    hflag[nh + k] = 0;
                    if (rseq) rseq[nr + k] = b;
                    if (rptr) { rptr[nr + k] = nh + k; rflag[nr + k] = 0; }
                }
                nh += n; nr += n; pos = stop;
            }
        }
    }
}

#endif

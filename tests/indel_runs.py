"""Test input: superclusters around runs of directly adjacent separate deletion records -- the layout that puts many
allowed swap sources on one position (every record ends at the position behind the run; dist.cpp:335-350)."""
import numpy as np

from vcfdist_amd import _abi as A


def indel_run_superclusters(seed, n_sc=300):
    """superclusters around runs of 1..7 directly adjacent deletion records (1-3 bases each) on the query haps; the truth
    haps carry the same records, one record of the same total length, a part of them, or nothing; SNPs beside the runs on
    either side make distances above zero"""
    rng = np.random.default_rng(seed)
    W = 120
    ctg = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=W * n_sc + 10).astype(np.uint8)
    recs = [[] for _ in range(4)]          # per hap: (sc, pos, type, ref_len, alt bytes)
    for sc in range(n_sc):
        base = sc * W + 30
        k = int(rng.integers(1, 8))
        lens = rng.integers(1, 4, size=k)
        starts = base + np.concatenate([[0], np.cumsum(lens)[:-1]])
        total = int(lens.sum())
        qh = [h for h in (0, 1) if rng.random() < 0.7] or [0]
        for h in qh:
            for st, ln in zip(starts, lens):
                recs[h].append((sc, int(st), 3, int(ln), b""))
        for h in (2, 3):
            mode = int(rng.integers(0, 4))
            if mode == 0:
                for st, ln in zip(starts, lens):
                    recs[h].append((sc, int(st), 3, int(ln), b""))
            elif mode == 1:
                recs[h].append((sc, base, 3, total, b""))
            elif mode == 2:
                for st, ln in list(zip(starts, lens))[: max(1, k // 2)]:
                    recs[h].append((sc, int(st), 3, int(ln), b""))
        for h in range(4):                  # SNPs a few bases in front of / behind the run
            for pos in (base - int(rng.integers(2, 12)), base + total + int(rng.integers(1, 12))):
                if rng.random() < 0.35:
                    alt = bytes([b"ACGT"[(b"ACGT".index(int(ctg[pos])) + int(rng.integers(1, 4))) % 4]])
                    recs[h].append((sc, pos, 1, 1, alt))
    var_off, pos, typ, qual, roff, rlen, aoff, alen, pool = [], [], [], [], [], [], [], [], []
    for h in range(4):
        r = sorted(recs[h], key=lambda e: (e[0], e[1]))
        cnt = np.bincount([e[0] for e in r], minlength=n_sc) if r else np.zeros(n_sc, np.int64)
        var_off.append(np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64))
        pl, ro, ao = bytearray(), [], []
        for e in r:
            ro.append(len(pl)); pl += bytes(ctg[e[1]:e[1] + e[3]].tolist())
            ao.append(len(pl)); pl += e[4]
        pos.append(np.array([e[1] for e in r], np.int32)); typ.append(np.array([e[2] for e in r], np.uint8))
        qual.append(np.full(len(r), 30.0, np.float32))
        roff.append(np.array(ro, np.int64)); rlen.append(np.array([e[3] for e in r], np.int32))
        aoff.append(np.array(ao, np.int64)); alen.append(np.array([len(e[4]) for e in r], np.int32))
        pool.append(np.frombuffer(bytes(pl) + b"\0", np.uint8).copy())
    beg = np.arange(n_sc, dtype=np.int32) * W + 10
    end = np.arange(n_sc, dtype=np.int32) * W + 30 + 21 + 14
    return A.Variants(np.array([0, len(ctg)], np.int64), ctg, np.zeros(n_sc, np.int32), beg, end, var_off, pos, typ, qual,
                      roff, rlen, aoff, alen, pool)

"""Dense row-sweep model of the device algorithm (K1 forward, K2 backward), in
plain Python for tiny cases.  It is the executable specification the HIP kernels
follow (DESIGN.md "Kernels"), and tests compare it with the sparse-wave oracle:
same distance, same end plane, same flag bytes on every cell with D <= s, same
path_scores / path_ptrs on every cell the oracle's backward pass reached.

Independent of the oracle's code: distances come from a textbook relaxation over
rows of the truth string instead of a queue-driven wave expansion."""
import numpy as np

F_INS, F_DEL, F_MAT, F_SUB, F_SWP = 1, 2, 4, 8, 16
VARIANT, VAR_BEG, VAR_END, INS_LOC = 1, 2, 4, 8
INF = 1 << 28


def fwd_allow(f):
    return (not (f & VARIANT)) or bool(f & VAR_END)


def bwd_allow(f):
    return (not (f & VARIANT)) or bool(f & VAR_BEG)


def swap_sources(src_ptr, src_flag, n_dst):
    """For each destination index d in the other plane: the list of source indices x with
    src_ptr[x] + 1 == d and fwd_allow(src_flag[x]) (ascending)."""
    out = [[] for _ in range(n_dst)]
    for x in range(len(src_ptr)):
        d = src_ptr[x] + 1
        if 0 <= d < n_dst and fwd_allow(src_flag[x]):
            out[d].append(x)
    return out


def forward(Q, R, T, q2r, qfl, r2q, rfl, tfl):
    """Returns D[2], flags[2] (shape [len][Lt]), tie[2] (bool), choice[2] (chosen swap source or -1)."""
    Lq, Lr, Lt = len(Q), len(R), len(T)
    S = (Q, R)
    L = (Lq, Lr)
    D = [np.full((Lq, Lt), INF, np.int64), np.full((Lr, Lt), INF, np.int64)]
    FL = [np.zeros((Lq, Lt), np.uint8), np.zeros((Lr, Lt), np.uint8)]
    TIE = [np.zeros((Lq, Lt), bool), np.zeros((Lr, Lt), bool)]
    CH = [np.full((Lq, Lt), -1, np.int64), np.full((Lr, Lt), -1, np.int64)]
    # sources in the *other* plane for each destination of plane p
    src = (swap_sources(r2q, rfl, Lq), swap_sources(q2r, qfl, Lr))
    for t in range(Lt):
        for p in range(2):
            o = 1 - p
            for q in range(L[p]):
                if t == 0 and q == 0:
                    D[p][0, 0] = 0
                    FL[p][0, 0] = F_MAT
                    continue
                match = t > 0 and q > 0 and S[p][q] == T[t]
                diag = D[p][q - 1, t - 1] if (q > 0 and t > 0) else INF
                up = D[p][q, t - 1] if t > 0 else INF
                left = D[p][q - 1, t] if q > 0 else INF
                swp = INF
                cands = []
                if t > 0 and q > 0 and S[p][q] == T[t] and fwd_allow(tfl[t - 1]):
                    for x in src[p][q]:
                        cands.append((D[o][x, t - 1], x))
                    if cands:
                        swp = min(c[0] for c in cands)
                d = min(diag + (0 if match else 1), up + 1, left + 1, swp)
                D[p][q, t] = d
                f = 0
                if match and diag == d: f |= F_MAT
                if q > 0 and t > 0 and diag + 1 == d: f |= F_SUB
                if up + 1 == d: f |= F_DEL
                if left + 1 == d: f |= F_INS
                if swp == d and swp < INF:
                    f |= F_SWP
                    best = [x for (dd, x) in cands if dd == d]
                    TIE[p][q, t] = len(best) > 1
                    CH[p][q, t] = max(best)
                FL[p][q, t] = f
        # NOTE: plane p=1 at row t reads plane 0 at row t-1 only (swap consumes a truth base), so the
        # in-row order of planes does not matter.
    return D, FL, TIE, CH


def backward(Q, R, T, q2r, qfl, r2q, rfl, FL, CH, end_plane):
    """Dense reverse sweep of the max-TP longest path over the optimal DAG.
    Returns score[2] (-1 unreachable), pptr[2], tie_used (bool)."""
    Lq, Lr, Lt = len(Q), len(R), len(T)
    L = (Lq, Lr)
    SC = [np.full((Lq, Lt), -1, np.int64), np.full((Lr, Lt), -1, np.int64)]
    PP = [np.zeros((Lq, Lt), np.uint8), np.zeros((Lr, Lt), np.uint8)]
    SC[end_plane][L[end_plane] - 1, Lt - 1] = 0
    PP[end_plane][L[end_plane] - 1, Lt - 1] = F_MAT

    def tp(p, q):  # entering the first base of a query variant, only on the QUERY plane
        return int(p == 0 and ((q2r[q] != q2r[q - 1] + 1) or bool(qfl[q] & VAR_BEG)))

    fl_of = (qfl, rfl)
    ptr_of = (q2r, r2q)
    for t in range(Lt - 1, -1, -1):
        for p in range(2):
            o = 1 - p
            for q in range(L[p] - 1, -1, -1):
                cands = []  # (value, move type)
                if q + 1 < L[p] and t + 1 < Lt:
                    x = (q + 1, t + 1)
                    if SC[p][x] >= 0:
                        if FL[p][x] & F_MAT: cands.append((SC[p][x] + tp(p, q + 1), F_MAT))
                        if FL[p][x] & F_SUB: cands.append((SC[p][x] + tp(p, q + 1), F_SUB))
                if t + 1 < Lt and SC[p][q, t + 1] >= 0 and (FL[p][q, t + 1] & F_DEL):
                    cands.append((SC[p][q, t + 1], F_DEL))
                if q + 1 < L[p] and SC[p][q + 1, t] >= 0 and (FL[p][q + 1, t] & F_INS):
                    cands.append((SC[p][q + 1, t] + tp(p, q + 1), F_INS))
                # swap successor: z in the other plane whose chosen swap source is this cell
                zq = ptr_of[p][q] + 1
                if t + 1 < Lt and 0 < zq < L[o] and fwd_allow(fl_of[p][q]):
                    z = (zq, t + 1)
                    if SC[o][z] >= 0 and (FL[o][z] & F_SWP) and CH[o][z] == q and bwd_allow(fl_of[o][zq]):
                        # leaving a REF-plane cell scores 0; leaving a QUERY-plane cell scores tp(z)
                        cands.append((SC[o][z] + (tp(0, zq) if o == 0 else 0), F_SWP))
                if not cands:
                    continue
                best = max(c[0] for c in cands)
                if (p, q, t) == (end_plane, L[end_plane] - 1, Lt - 1):
                    continue
                SC[p][q, t] = best
                m = 0
                for v, mv in cands:
                    if v == best: m |= mv
                PP[p][q, t] = m
    return SC, PP

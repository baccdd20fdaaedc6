#!/usr/bin/env python
"""bench.py -- supercluster-alignments/s of the MI355X precision/recall path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`, one process per GPU
(torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE), rank 0 prints ONE JSON line.

A "step" is one pass of the hot path (K1 forward sweep, K2 backward sweep, K3 walk +
credit sections, K4 section edit distances, result download + float finalisation)
over one batch of synthetic superclusters that is already resident in HBM.

Workload (config.workload = "wgs_synth"): BASELINE.json configs[1] (HG002 WGS small
variants on one MI355X) emulated with the generator of SURVEY.md 8(d): spans
log-normal (median 20, sigma 1.2, clipped to [4, 10000]), Poisson(max(1, L/200))
sites, 80 % SNP, 70 % homozygous, truth = query kept/dropped/perturbed 0.9/0.05/0.05,
20 % tandem-repeat spans; real HG002 data is not available offline.  Each rank owns
the same number of superclusters with a rank-specific seed (weak scaling: superclusters
are independent, no data-path collective); the precision/recall counters of SURVEY 8(e),
counts[callset][SNP,INDEL,SV,ALL][TP,FP,FN][61 quality thresholds] (int64, computed on the
device), are summed with one all-reduce (RCCL) at the end of every step.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # see vcfdist_amd/api.py: before anything initialises HIP

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def make_workload(api, n_sc, seed, workload):
    if workload == "wgs_synth":
        return api.Synth(n_sc=n_sc, seed=seed, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10000)
    if workload == "stress_synth":   # configs[4]: log-uniform 32..16384
        return api.Synth(n_sc=n_sc, seed=seed, len_mode=0, len_a=32.0, len_b=16384.0, len_min=32, len_max=16384)
    raise SystemExit(f"unknown workload {workload}")


def cpu_baseline(batch, target_s=15.0):
    """Time the CPU oracle (a port of the reference's algorithm) on a bounded sample of the same workload:
    single-threaded, and with one thread per host core the way the reference's own driver spreads superclusters
    over threads (precision_recall_threads_wrapper, dist.cpp:1656).  Checker code used as a *reported baseline*
    only; `value` is the all-cores rate, `cores` the threads actually used."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from concurrent.futures import ThreadPoolExecutor
    import oracle_lib
    rng = np.random.RandomState(1234)
    n = batch.n_sc
    # probe on 2000 random superclusters (1 thread), then size the threaded sample for ~target_s of wall time
    probe = np.sort(rng.choice(n, size=min(n, 2000), replace=False))
    sub = batch.subset(probe)
    t0 = time.perf_counter()
    oracle_lib.run(sub)
    dt1 = time.perf_counter() - t0
    per = dt1 / len(probe)
    threads = max(1, min(os.cpu_count() or 1, 64))
    m = int(min(n, max(len(probe), threads * target_s / max(per, 1e-9))))
    samp = np.sort(rng.choice(n, size=m, replace=False))
    parts = [batch.subset(c) for c in np.array_split(samp, threads) if len(c)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=len(parts)) as ex:     # ctypes releases the GIL inside vpo_run
        list(ex.map(oracle_lib.run, parts))
    dt = time.perf_counter() - t0
    cells = sum(p.dense_cells() for p in parts)
    return {
        "value": round(4 * len(samp) / dt, 1), "unit": "supercluster-alignments/s", "cores": len(parts), "kind": "port",
        "sample": f"{len(samp)} superclusters drawn uniformly from the bench batch ({cells:.3e} dense cells), "
                  f"{dt:.1f} s wall on {len(parts)} threads (oracle/pr_oracle.cpp, one slice per thread)",
        "cells_per_s": round(cells / dt, 1),
        "single_thread_value": round(4 * len(probe) / dt1, 1),
        "single_thread_sample": f"{len(probe)} superclusters, {dt1:.2f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-sc", type=int, default=1000000, help="superclusters per GPU")
    ap.add_argument("--workload", default="wgs_synth")
    ap.add_argument("--seed", type=int, default=0x5eed)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # VCFDIST_BENCH_ONE_GPU=1 (plumbing check on a one-GPU box only): every rank uses cuda:0 and the collective
        # runs over gloo on host copies; the numbers of such a run mean nothing
        if os.environ.get("VCFDIST_BENCH_ONE_GPU"):
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from vcfdist_amd import api, shard, summary, _abi as A
    if rank == 0:
        api.build()                 # no-op when the in-tree library is current (the driver builds it beforehand)
    if dist is not None:
        dist.barrier()
    t_a = time.perf_counter()
    syn = make_workload(api, args.n_sc, shard.rank_seed(args.seed, rank), args.workload)
    t_b = time.perf_counter()
    batch = syn.batch(copy=False)       # host marshalling = the four generate_ptrs_strs calls per supercluster
    t_c = time.perf_counter()
    pr = api.PrecisionRecall(device=local_rank)
    pr.upload(batch)                    # inputs resident in HBM before the timed region (+ K0 prep kernels)
    vv = syn.variants()                 # SNP / INDEL / SV class of every variant (print.cpp:362-372), resident too
    summary.upload_var_class(pr, [summary.var_class(vv.var_type[s], vv.var_ref_len[s], vv.var_alt_len[s]) for s in range(4)])
    t_d = time.perf_counter()
    in_bytes = sum(a.nbytes for h in range(4) for a in (batch.hap_seq[h], batch.hap_ptr[h], batch.hap_flag[h],
                                                        batch.hap_off[h], batch.var_off[h], batch.var_pos[h],
                                                        batch.var_qual[h])) + batch.ref_seq.nbytes + \
        batch.ref_off.nbytes + sum(a.nbytes for h in range(2) for a in (batch.ref_ptr[h], batch.ref_flag[h]))
    dev = torch.device("cuda", local_rank)

    host_res = [None]

    def step():
        pr.execute()                    # K1..K5 on the device
        host_res[0] = res = pr.download(host_res[0])   # final results to (reused) host buffers
        t = torch.from_numpy(summary.pr_counts(pr, None, None)).to(dev)   # [2][4][3][61] int64, device histogram
        if dist is not None and dist.get_backend() == "gloo":
            tc = t.cpu(); dist.all_reduce(tc); t = tc.to(dev)
        elif dist is not None:
            dist.all_reduce(t)          # the one collective of the path: the precision/recall counters (int64 sum)
        return res, t

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    kern_ms = []
    stats_acc = {}
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res, t = step()
        tm = pr.timing()
        kern_ms.append(tm.ms_total)
        for s in pr.launch_stats():
            key = (s.kind, s.kernel.decode())
            a = stats_acc.setdefault(key, [0, 0.0, 0, 0, 0])
            a[0] += 1; a[1] += s.ms; a[2] += s.bytes_algorithmic; a[3] += s.cells; a[4] += s.cells_dense
    sync()
    elapsed = time.perf_counter() - t0
    if rank == 0:       # the device tally must equal the one recomputed from the downloaded results
        assert np.array_equal(shard.tally_from_results(res, batch.var_off), pr.tally()), "device tally != host tally"
        # after the timed region: per-contig phasing (host Viterbi) and the PRECISION-RECALL SUMMARY of this rank
        pb, sw, fl = summary.phase(res.sc_phase, np.ones(batch.n_sc, np.int32))
        rows = summary.pr_summary(summary.pr_counts(pr, None, pb))
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    total_aln = 4 * args.n_sc * world * args.steps
    value = total_aln / elapsed
    if rank == 0:
        tm = pr.timing()
        # dominant K1/K2 kernel (the DP sweeps the byte model of SURVEY 8(d) is about): algorithmic bytes per launch
        # / average launch duration (HIP events on the stream the kernel is launched on)
        sweeps = {k: v for k, v in stats_acc.items() if k[0] in (1, 2)}
        (kind, kname), (nl, ms, byt, cells, dense) = max(sweeps.items(), key=lambda kv: kv[1][1])
        # algorithmic bytes = 2 B per *dense* DP cell (1 B stored by the forward sweep, 1 B loaded by
        # the backward sweep) + the input arrays; `achieved` follows that formula for the launch's alignments.
        # The window kernels sweep (and move) far fewer cells than the dense matrix: `achieved_swept` counts only
        # the flag bytes of the swept cells, and `traffic` is what the PMC counters saw.
        in_b = (byt - cells) if kind == 1 else 0
        dense_bytes = (dense + in_b) / nl
        achieved = dense_bytes / (ms / nl * 1e-3) / 1e9 if ms > 0 else 0.0
        swept = (byt / nl) / (ms / nl * 1e-3) / 1e9 if ms > 0 else 0.0
        traffic = None
        try:   # HBM bytes per launch from the committed rocprofv3 PMC passes of this very command (profiles/)
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")))
            if prof["workload"] == args.workload and prof["superclusters_per_gpu"] == args.n_sc and kname in prof["kernels"]:
                k = prof["kernels"][kname]
                # FETCH_SIZE counts half of wide coalesced reads on gfx950 (MI355X_MICROARCH.md): x2; KB -> bytes
                traffic = int((2 * k["fetch_kb"] + k["write_kb"]) * 1024)
        except (OSError, KeyError, ValueError):
            pass
        roof = {
            "bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
            "launches": nl, "avg_launch_ms": round(ms / nl, 4), "algorithmic_bytes_per_launch": int(dense_bytes),
            "dense_cells_per_launch": int(dense / nl), "swept_cells_per_launch": int(cells / nl),
            "swept_bytes_per_launch": int(byt / nl), "achieved_swept": round(swept, 2),
            "frac_swept": round(swept / HBM_PEAK_GBS, 5),
            "traffic_source": "profiles/r01_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, FETCH x2)" if traffic else None,
            "note": "dense-equivalent bytes per SURVEY 8(d); the window kernels move only swept_bytes and are "
                    "VALU-issue bound, not HBM bound (DESIGN.md section 3)",
        }
        per_kernel = {k[1]: {"launches": v[0], "ms": round(v[1], 3)} for k, v in sorted(stats_acc.items())}
        out = {
            "metric": "supercluster-alignments/sec", "value": round(value, 1), "unit": "supercluster-alignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": args.workload, "superclusters_per_gpu": args.n_sc,
                       "span_dist": "lognormal(median 20, sigma 1.2) clip [4,10000]" if args.workload == "wgs_synth"
                       else "loguniform [32,16384]", "sharding": f"{world} ranks x independent superclusters"},
            "dense_cells_per_s": round(tm.cells_dense * world * args.steps / elapsed, 1),
            "kernel_ms_per_step": round(float(np.mean(kern_ms)), 3),
            "setup_not_timed": {"generate_s": round(t_b - t_a, 2), "host_marshalling_s": round(t_c - t_b, 2),
                                "upload_and_prep_s": round(t_d - t_c, 2), "input_bytes": int(in_bytes),
                                "pcie_inclusive_value": round(4 * args.n_sc / ((t_d - t_c) + elapsed / args.steps), 1)},
            "kernel_only_value": round(4 * args.n_sc / (float(np.mean(kern_ms)) * 1e-3), 1),
            "kernels": per_kernel,
            "counts_at_min_qual_TP_FP_FN": t.cpu().numpy()[:, 3, :, 0].tolist(),   # [callset][TP,FP,FN], type ALL, all ranks
            "pr_summary_rank0": [{"type": summary.NAMES[r.vartype], "threshold": "BEST" if r.best else "NONE", "qual": r.qual,
                                  "truth_tp": r.truth_tp, "query_tp": r.query_tp, "truth_fn": r.truth_fn, "query_fp": r.query_fp,
                                  "precision": round(r.precision, 6), "recall": round(r.recall, 6), "f1": round(r.f1_score, 6)}
                                 for r in rows if not r.best],
            "phasing_rank0": {"switches": int(len(sw)), "flips": int(len(fl))},
            "order_defined_swap_ties": int((res.aln_status & 1).sum()),
            "roofline": roof,
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(batch)
            out["cpu_baseline"]["host"] = f"{os.cpu_count()} logical cores visible"
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

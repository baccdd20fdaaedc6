#!/usr/bin/env python
"""bench.py -- supercluster-alignments/s of the MI355X precision/recall path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`, one process per GPU
(torch.distributed.run sets RANK/LOCAL_RANK/WORLD_SIZE), rank 0 prints ONE JSON line.

A "step" is one pass of the hot path (K1 forward sweep, K2 backward sweep incl. the replay of the
reference's container order where swap predecessors tie, K3 walk + credit sections, K4 section edit
distances, K5 finalisation; result download; device histogram + all-reduce of the counters)
over one batch of synthetic superclusters that is already resident in HBM.

Workload (config.workload = "wgs_synth"): BASELINE.json configs[1] (HG002 WGS small
variants on one MI355X) emulated with the generator of SURVEY.md 8(d): spans
log-normal (median 20, sigma 1.2, clipped to [4, 10000]), Poisson(max(1, L/200))
sites, 80 % SNP, 70 % homozygous, truth = query kept/dropped/perturbed 0.9/0.05/0.05,
20 % tandem-repeat spans; real HG002 data is not available offline.  Each rank owns
the same number of superclusters with a rank-specific seed (weak scaling: superclusters
are independent, no data-path collective); the precision/recall counters of SURVEY 8(e),
counts[callset][SNP,INDEL,SV,ALL][TP,FP,FN][61 quality thresholds] (int64, computed on the
device), are summed with one all-reduce (RCCL) at the end of every step.  `--scaling strong`
deals one synthetic genome over the ranks instead (vcfdist_amd/shard.py), `--workload
stress_synth | sv_synth` select BASELINE configs[4] (125 000 superclusters per GPU) and an
emulation of configs[2].  The line also carries `roofline` (counter-based HBM fraction and VALU
issue fraction of the dominant sweep kernel), `cpu_baseline` (the CPU port
on the host cores, with its calibration against the reference's published timings),
`step_parts_ms` and `setup_not_timed` (upload, one-shot rate).
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
N_SIMD, SCLK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMDs; shader clock of the committed SQ_BUSY_CYCLES counters
PROFILE_TAG = "r06"    # profiles/<tag>_counters_<workload>.json: tools/make_profiles.sh


def newest_profile(kind, workload=None):
    """profiles/<tag>_<kind>[_<workload>].json of this round, else the newest earlier round's (the line says which)"""
    import glob
    import re
    pat = os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{kind}" + (f"_{workload}" if workload else "") + ".json")
    cands = sorted(glob.glob(pat), key=lambda f: int(re.search(r"r(\d\d)_", os.path.basename(f)).group(1)))
    cands = [f for f in cands if int(re.search(r"r(\d\d)_", os.path.basename(f)).group(1)) <= int(PROFILE_TAG[1:])]
    return cands[-1] if cands else None


def counter_roofline(kname, workload, n_sc, avg_s, n_simd, sclk_hz):
    """HBM traffic and VALU issue fraction of kernel `kname` from the committed rocprofv3 counter passes of this workload at this
    size (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md), priced with the launch duration measured in THIS run"""
    f = newest_profile("counters", workload)
    if not f:
        return None
    try:
        prof = json.load(open(f))
        if prof["workload"] != workload or prof["superclusters_per_gpu"] != n_sc:
            return None
        if kname not in prof["kernels"]:
            # the library's launch statistics name a launch by its role (`k_tie_replay<early>`: the speculative replays), the
            # kernel trace by the instantiation (`k_tie_replay<4>` / `<8>`: waves per job): the instantiation with the most time
            base = kname.split("<")[0]
            cands = [n for n in prof["kernels"] if n.split("<")[0] == base]
            if not cands:
                return None
            kname = max(cands, key=lambda n: prof["kernels"][n].get("busy_cycles_8xcd", 0))
        k = prof["kernels"][kname]
        traffic = int((2 * k["fetch_kb"] + k["write_kb"]) * 1024)
        valu = None
        if k.get("valu_active_per_wave") and k.get("waves"):
            valu = k["valu_active_per_wave"] * 4 * k["waves"] / (n_simd * avg_s * sclk_hz)
        return {"traffic": traffic, "valu_issue_frac": valu, "source": "profiles/" + os.path.basename(f), "k": k, "prof": prof,
                "counter_kernel": kname}
    except (OSError, KeyError, ValueError):
        return None


def make_workload(api, n_sc, seed, workload):
    if workload == "wgs_synth":
        return api.Synth(n_sc=n_sc, seed=seed, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10000)
    if workload == "stress_synth":   # configs[4]: log-uniform 32..16384
        return api.Synth(n_sc=n_sc, seed=seed, len_mode=0, len_a=32.0, len_b=16384.0, len_min=32, len_max=16384)
    if workload == "sv_synth":       # configs[2] emulated (`-l 10000 -s 10002`, `-c size 100`): spans of thousands of bases with
        # SV-sized indels (geometric, mean 600, capped at a quarter of the span), small variants in between
        return api.Synth(n_sc=n_sc, seed=seed, len_mode=0, len_a=2000.0, len_b=12000.0, len_min=2000, len_max=12000,
                         var_per_base=0.002, p_snp=0.7, indel_mean=600.0)
    if workload == "joint_synth":    # configs[3]: whole-genome SNP + INDEL + SV joint evaluation (`-l 10000 -s 10002`): the
        # whole-genome length mix, 0.75 % of the superclusters carry one SV-sized indel (50 b .. 10 kb, log-uniform)
        return api.Synth(n_sc=n_sc, seed=seed, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10002,
                         p_sv=0.0075, sv_min=50, sv_max=10000)
    raise SystemExit(f"unknown workload {workload}")


def cpu_limit():
    """CPUs this process may use: the cgroup's quota where there is one (the GPU boxes show 256 logical cores and allow 16)"""
    n = os.cpu_count() or 1
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    lw = int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1)     # (ranks of one node share the quota)
    return max(1, n // max(lw, 1))


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def calibrate(strata, n_sc):
    """the port's time on the reference's four demo workloads over the reference's own (tools/calibrate_cpu.py,
    measured in the build container) misses +-15 % on some of them -- the port is faster than the reference on small
    superclusters and slower on 10 kb ones.  So every stratum's CPU time is divided by the ratio of ITS size class: the ratio
    as a function of the supercluster length, interpolated in log2(length) between the four workloads (each placed at its
    cost-weighted span, sum L^3 / sum L^2: where its time is spent), constant outside them.  `calibrated_value` is then an
    estimate of the REFERENCE's rate on this host."""
    cal_f = newest_profile("cpu_calibration")
    if not cal_f:
        return {"calibration_port_over_reference_time": None, "calibration_note": "no profiles/r*_cpu_calibration.json"}
    try:
        cal = json.load(open(cal_f))
        ratios = {k: v["oracle_over_reference"] for k, v in cal.items()}
        out = {"calibration_port_over_reference_time": ratios, "calibration_source": "profiles/" + os.path.basename(cal_f)}
        missed = {k: r for k, r in ratios.items() if not 0.85 <= r <= 1.15}
        out["calibration_within_15_percent"] = not missed
        pts = sorted((np.log2(v["cost_weighted_span"]), v["oracle_over_reference"]) for v in cal.values() if v.get("cost_weighted_span"))
        if len(pts) >= 2 and strata:
            xs, ys = np.array([p[0] for p in pts]), np.array([p[1] for p in pts])
            tot = sum(st["est_batch_seconds"] / float(np.interp(st["k"] + 0.585, xs, ys)) for st in strata)     # (octave centre 1.5 x 2^k)
            thr = max(st["threads"] for st in strata)
            out["calibrated_value"] = round(4 * n_sc / tot, 3) if tot > 0 else None
            out["calibrated_note"] = ("each stratum's measured time / the port-over-reference ratio of its size class (interpolated in "
                                      "log2 of the span between the demo workloads' cost-weighted spans "
                                      f"{[round(2 ** float(x)) for x in xs]} -> ratios {[float(y) for y in ys]}): the reference's estimated rate on this host")
        elif missed:
            out["calibration_note"] = f"outside SURVEY 8(d)'s +-15 % on {sorted(missed)}; no span figures in the calibration file to scale by"
        return out
    except (OSError, KeyError, ValueError, TypeError) as e:
        return {"calibration_port_over_reference_time": None, "calibration_note": f"calibration file unreadable ({e!r})"}


def cpu_baseline(batch, target_s=15.0, probe=True):
    """Time the CPU oracle (a port of the reference's algorithm, matrices held as the reference holds them) on a bounded
    sample of the same workload, STRATIFIED by octave of the supercluster length: the cost per supercluster grows with L^2,
    so a uniform sample of a long-tailed workload is decided by whether it happens to draw one of the few huge
    superclusters.  Every octave [2^k, 2^(k+1)) is sampled on its own (all host threads, one slice per thread, the way the
    reference's driver spreads superclusters over threads: precision_recall_threads_wrapper, dist.cpp:1656) and timed;
    the whole batch's CPU time is estimated as sum over octaves of (superclusters in the octave / measured rate), and
    `value` = alignments of the batch / that time.  Checker code used as a *reported baseline* only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from concurrent.futures import ThreadPoolExecutor
    import oracle_lib
    rng = np.random.RandomState(1234)
    n = batch.n_sc
    L = np.maximum(np.diff(batch.ref_off), 1)
    octv = np.floor(np.log2(L)).astype(np.int64)
    threads = max(1, min(cpu_limit(), 64))     # (more threads than the quota allows get the whole group suspended)
    strata = []
    octaves = [int(k) for k in np.unique(octv)]
    # per-supercluster single-thread cost model for sizing the samples only: 40 us + 15 ns x L^2
    est = lambda k: 40e-6 + 15e-9 * (1.5 * 2.0 ** k) ** 2
    budget = target_s / max(len(octaves), 1)
    t_all = time.perf_counter()
    est_total = 0.0
    n_samp = 0
    for k in octaves:
        idx = np.flatnonzero(octv == k)
        # (superclusters of 8 192+ bases: the port's dense matrices take gigabytes per alignment -- at most eight at a time)
        thr_k = threads if 2 ** k < 8192 else min(threads, 8)
        est_k = est(k)
        if not probe and 2 ** k >= 1024:
            # (the secondary workloads: the cost model above is for small variants -- with SV-sized indels a supercluster costs ten
            # times as much --, so one supercluster of the stratum is timed first and the sample is sized from it; where that one
            # alone uses up the stratum's budget it IS the sample)
            one = np.sort(rng.choice(idx, size=1))
            t0 = time.perf_counter()
            oracle_lib.run(batch.subset(one))
            est_k = time.perf_counter() - t0
            if est_k >= budget or len(idx) == 1:
                est_total += len(idx) * est_k / min(thr_k, len(idx))
                n_samp += 1
                strata.append({"k": k, "octave": f"[{2 ** k}, {2 ** (k + 1)})", "superclusters": int(len(idx)), "sampled": 1, "threads": 1,
                               "seconds": round(est_k, 3), "est_batch_seconds": round(len(idx) * est_k / min(thr_k, len(idx)), 3),
                               "note": f"one supercluster on one thread; the stratum's time on {min(thr_k, len(idx))} threads is extrapolated"})
                continue
        m = int(min(len(idx), max(thr_k if len(idx) >= thr_k else 1, budget * thr_k / est_k)))
        samp = np.sort(rng.choice(idx, size=m, replace=False))
        parts = [batch.subset(c) for c in np.array_split(samp, min(thr_k, m)) if len(c)]
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=len(parts)) as ex:     # ctypes releases the GIL inside vpo_run
            list(ex.map(oracle_lib.run, parts))
        dt = time.perf_counter() - t0
        est_total += len(idx) * dt / m
        n_samp += m
        strata.append({"k": k, "octave": f"[{2 ** k}, {2 ** (k + 1)})", "superclusters": int(len(idx)), "sampled": m, "threads": len(parts),
                       "seconds": round(dt, 3), "est_batch_seconds": round(len(idx) * dt / m, 3)})
    wall = time.perf_counter() - t_all
    if not probe:       # (the secondary workloads: the stratified estimate only)
        cal = calibrate(strata, n)
        return {"value": round(4 * n / est_total, 3), "unit": "supercluster-alignments/s", "cores": threads, "kind": "port",
                "sample": f"{n_samp} superclusters in {len(octaves)} strata by octave of length, {wall:.1f} s wall on {threads} threads "
                          f"(oracle/pr_oracle.cpp); value = batch alignments / sum over strata of (stratum size / measured rate)",
                "est_batch_seconds_all_threads": round(est_total, 2),
                "single_thread_value": round(4 * n / (est_total * threads), 3),
                "calibrated_value": cal.get("calibrated_value")}
    # single-thread probe on the most populated octave (for the per-core rate)
    k_top = max(octaves, key=lambda k: int((octv == k).sum()))
    idx = np.flatnonzero(octv == k_top)
    probe = np.sort(rng.choice(idx, size=min(len(idx), 2000), replace=False))
    t0 = time.perf_counter()
    oracle_lib.run(batch.subset(probe))
    dt1 = time.perf_counter() - t0
    return dict({
        "value": round(4 * n / est_total, 1), "unit": "supercluster-alignments/s", "cores": threads, "kind": "port",
        "sample": f"{n_samp} superclusters in {len(octaves)} strata by octave of length, {wall:.1f} s wall on {threads} threads "
                  f"(oracle/pr_oracle.cpp, one slice per thread); value = batch alignments / sum over strata of (stratum size / measured rate)",
        "est_batch_seconds_all_threads": round(est_total, 2),
        "strata": strata,
        # one thread on the WHOLE mix: the strata's thread-seconds (the slices of a stratum run side by side on `threads` threads)
        "single_thread_value": round(4 * n / (est_total * threads), 1),
        "single_thread_note": "whole mix: batch alignments / (sum over strata of estimated seconds x threads used)",
        "single_thread_value_most_populated_octave": round(4 * len(probe) / dt1, 1),
        "single_thread_sample": f"{len(probe)} superclusters of the most populated octave [{2 ** k_top}, {2 ** (k_top + 1)}), {dt1:.2f} s",
        "cpu_model": cpu_model(),
    }, **calibrate(strata, n))


def secondary_leg(api, summary, workload, n_sc, seed, steps, device, cpu_target_s=0.0):
    """one more workload after the headline (rank 0, N = 1): `steps` timed passes over one resident batch, one at a time, after
    one untimed pass; the line of the dominant sweep kernel is priced by the bytes of the cells it sweeps (no committed
    counters for these sizes)"""
    t0 = time.perf_counter()
    syn = make_workload(api, n_sc, seed, workload)
    batch = syn.batch(copy=False)
    pr = api.PrecisionRecall(device=device)
    pr.upload(batch)
    summary.upload_var_class(pr, syn.var_class())
    setup_s = time.perf_counter() - t0
    res = None
    step_ms, acc = [], {}
    for i in range(steps + 1):
        ta = time.perf_counter()
        pr.execute()
        res = pr.download(res)
        summary.pr_counts(pr, None, None)
        dt = (time.perf_counter() - ta) * 1e3
        if i == 0:
            continue
        step_ms.append(dt)
        for s_ in pr.launch_stats():
            a = acc.setdefault((s_.kind, s_.kernel.decode()), [0, 0.0, 0, 0])
            a[0] += 1; a[1] += s_.ms; a[2] += s_.bytes_algorithmic; a[3] += s_.cells
    tm = pr.timing()
    ms = float(np.mean(step_ms))
    sweeps = {k: v for k, v in acc.items() if k[0] in (1, 2) and v[1] > 0}
    (kind, kname), (nl, kms, byt, cells) = max(sweeps.items(), key=lambda kv: kv[1][1])
    avg_s = kms / nl * 1e-3
    gbs = (byt / nl) / avg_s / 1e9 if kms > 0 else 0.0
    import torch
    props = torch.cuda.get_device_properties(device)
    n_simd = 4 * int(getattr(props, "multi_processor_count", 0) or N_SIMD // 4)
    sclk_hz = float(getattr(props, "clock_rate", 0) or 0) * 1e3 or SCLK_HZ
    cr = counter_roofline(kname, workload, n_sc, avg_s, n_simd, sclk_hz)
    achieved = (cr["traffic"] / avg_s / 1e9) if cr else gbs
    roof = {"bound": "hbm", "kernel": kname, "launches_per_step": round(nl / steps, 2), "avg_launch_ms": round(kms / nl, 4),
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": cr["traffic"] if cr else None,
            "valu_issue_frac": None if not cr or cr["valu_issue_frac"] is None else round(cr["valu_issue_frac"], 4),
            "swept": {"bytes_per_launch": int(byt / nl), "GB/s": round(gbs, 2), "frac": round(gbs / HBM_PEAK_GBS, 5)},
            "counters_source": cr["source"] if cr else None,
            "note": ("traffic = FETCH_SIZE x 2 + WRITE_SIZE of the committed counter passes of this workload at this size / the launch duration "
                     "measured here (HIP events); swept = 1 B per swept cell + inputs; these launches are chains of dependent rows, not HBM-bound")
                    if cr else "no committed counters for this workload / size: bytes of the swept cells (1 B per cell + inputs) / launch duration"}
    out = {"workload": workload, "superclusters": n_sc, "steps": steps, "ms_per_step": round(ms, 3),
           "value": round(4 * n_sc / (ms * 1e-3), 1), "unit": "supercluster-alignments/s", "in_flight": 1,
           "dense_cells_per_s": round(tm.cells_dense / (ms * 1e-3), 1), "kernel_ms_per_step": round(tm.ms_total, 3),
           "setup_s": round(setup_s, 2),
           "roofline": roof,
           "top_kernels_ms_per_step": {k[1]: round(v[1] / steps, 3) for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:6]}}
    del pr
    if cpu_target_s > 0:
        try:    # the port on the host cores, stratified, a few seconds
            out["cpu_baseline"] = cpu_baseline(batch, target_s=cpu_target_s, probe=False)
            out["vs_cpu_baseline"] = round(out["value"] / out["cpu_baseline"]["value"], 1) if out["cpu_baseline"]["value"] > 0 else None
        except Exception as e:      # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)}
    return out


class stdout_to_stderr:
    """file descriptor 1 -> 2 for the duration (RCCL prints a version banner to stdout when a communicator is created; the
    driver reads ONE JSON line from this script's stdout)"""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        C.CDLL(None).fflush(None)       # (the banner sits in the C library's buffer: out with it while 1 is still 2)
        os.dup2(self.saved, 1)
        os.close(self.saved)


def launch_command(n_gpus, argv, port=None):
    """the command that runs this script as `n_gpus` ranks, one per GPU, on this node (what the driver itself runs)"""
    if port is None:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n-sc", type=int, default=1000000, help="superclusters per GPU")
    ap.add_argument("--workload", default="wgs_synth")
    ap.add_argument("--seed", type=int, default=0x5eed)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --n-sc superclusters per GPU (own seed, own contigs); strong: --n-sc-total superclusters of one "
                         "synthetic genome dealt over the ranks by estimated cells, phasing all-gathered every step")
    ap.add_argument("--n-sc-total", type=int, default=3000000)
    ap.add_argument("--in-flight", type=int, default=None,
                    help="batches resident in HBM whose steps may overlap (each behind its own library handle and host thread): "
                         "1 = strictly one step after the other.  Default: 3 for wgs_synth, 1 for the workloads of long alignments "
                         "(their replay scratches and ladder workspaces each want a large share of the free memory)")
    ap.add_argument("--one-pass-batches", type=int, default=None,
                    help="batches of the one-pass leg (upload from the variant tables + execute + download each; 0 = skip; "
                         "default 9 for wgs_synth, 0 for the other workloads)")
    ap.add_argument("--one-pass-in-flight", type=int, default=None,
                    help="batches in flight in the one-pass leg (default 3 for wgs_synth: a fresh batch keeps its host thread ~50 ms in "
                         "vpr_upload_variants and the device ~35 ms -- K0 + the step --, so a third thread fills the device: 103 M/s "
                         "against 87 with two; the headline keeps two)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the legs measured after the headline (N = 1, default workload only): two batches in flight, and the "
                         "sv_synth / stress_synth workloads (BASELINE configs[2] / configs[4]) on small batches")
    ap.add_argument("--plumbing-check", action="store_true",
                    help="launch path only (no GPU work): the ranks rendezvous over gloo, all-reduce their rank numbers and rank 0 "
                         "prints {n_gpus, rank_sum}; tests/test_distributed.py runs `bench.py --gpus 2 --plumbing-check` on CPU")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N`: become the launcher of N ranks (one process per GPU, RCCL); rank 0
        # of the new job prints the line
        os.execv(sys.executable, launch_command(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    if args.plumbing_check:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"plumbing_check": True, "n_gpus": world, "rank_sum": int(t.item())}))
        dist.barrier()
        dist.destroy_process_group()
        return
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
            with stdout_to_stderr():
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
                dist.barrier()

    from vcfdist_amd import api, shard, summary, _abi as A
    if rank == 0:
        api.build()                 # no-op when the in-tree library is current (the driver builds it beforehand)
    if dist is not None:
        dist.barrier()
    # The path's collectives through the C ABI (vpr_allreduce_counts / vpr_allgather_phase: RCCL on the library's own stream, on
    # a communicator of this job's ranks; at N = 1 a communicator of one rank, so that the default run exercises the same entry
    # point).  torch.distributed only carries the communicator's id to the ranks.  Any failure: torch.distributed's all_reduce.
    comm_ranks = None
    native_comm, collective = None, "torch.distributed all_reduce (" + (dist.get_backend() if dist is not None else "single rank: none") + ")"
    if not (dist is not None and dist.get_backend() == "gloo") and not os.environ.get("VCFDIST_BENCH_TORCH_COLLECTIVE"):
        try:
            from vcfdist_amd import rccl
            if rccl.available():
                torch.cuda.set_device(local_rank)
                uid_t = torch.zeros(rccl.ID_BYTES, dtype=torch.uint8, device=torch.device("cuda", local_rank))
                if rank == 0:
                    uid_t.copy_(torch.frombuffer(bytearray(rccl.unique_id()), dtype=torch.uint8))
                with stdout_to_stderr():
                    if dist is not None:
                        dist.broadcast(uid_t, 0)
                    uid_b = bytes(uid_t.cpu().numpy().tobytes())
                    box = []

                    def init_comm():        # (in a thread: a communicator that does not come up must not hang the bench)
                        try:
                            torch.cuda.set_device(local_rank)
                            box.append(rccl.Comm(world, rank, uid_b))
                        except Exception as e_:      # noqa: BLE001
                            box.append(e_)
                    import threading as _th
                    th_ = _th.Thread(target=init_comm, daemon=True)
                    th_.start()
                    th_.join(timeout=120.0)
                    mine_ok = bool(box) and not isinstance(box[0], Exception)
                    if dist is not None:    # every rank or none
                        okt = torch.tensor([1 if mine_ok else 0], device=torch.device("cuda", local_rank))
                        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
                        mine_ok = mine_ok and bool(int(okt.item()))
                    if mine_ok:
                        native_comm = box[0]
                        comm_ranks = native_comm.count()        # (ncclCommCount: the communicator's own idea of its size)
                        torch.cuda.synchronize()
                        collective = "vpr_allreduce_counts (RCCL through the C ABI, library stream)"
                    else:
                        sys.stderr.write(f"native RCCL communicator did not come up on every rank ({box[:1]!r}): torch.distributed collectives\n")
        except Exception as e:      # noqa: BLE001 -- the bench must not die of an optional path
            sys.stderr.write(f"native RCCL communicator not available ({e!r}): torch.distributed collectives\n")
            native_comm = None
    strong = args.scaling == "strong"
    dev = torch.device("cuda", local_rank)
    gloo = dist is not None and dist.get_backend() == "gloo"
    if args.in_flight is None:
        # Two batches resident behind two handles: one batch's latency tail -- a handful of alignments that are chains of
        # thousands of sequential rows, then their tie rounds -- runs beside the other batch's bulk (22 ms per step against 33
        # one at a time; every run of 12 on fresh boxes within 2 %).  Round 3 kept THREE: the third handle's first launches
        # then waited up to seconds behind the other handles' streams (39 streams on 8 hardware queues) and the driver's run
        # measured 133 ms per step; the first two handles never did, and the warm-up steps are watched for it below.
        # Round 5: three again -- 16.4 - 16.5 ms per step against 18.4 - 18.5 with two (100-step runs on one box; 17.0 - 18.7
        # against 18.4 - 18.7 over the driver's 20 steps, which include the pipeline's fill), six driver-command runs without a
        # stalled step (slowest step of any run 74 ms); a handle's kernels of a step now span 17 ms instead of 27, and the
        # warm-up steps are still watched.
        args.in_flight = 3 if args.workload == "wgs_synth" else 1
    if args.one_pass_batches is None:
        args.one_pass_batches = 12 if args.workload == "wgs_synth" else 0
    # (every resident batch goes through at least one untimed step: a handle's workspaces settle in its first execute)
    n_fl = max(1, min(args.in_flight, args.steps, max(args.warmup, 1)))

    class Slot:
        """one batch resident in HBM behind its own library handle"""
        pass

    def make_slot(k):
        S = Slot()
        S.t_a = time.perf_counter()
        if strong:      # every rank synthesises the same genome and keeps its share (dealt by estimated cells)
            S.syn = make_workload(api, args.n_sc_total, args.seed + 104729 * k, args.workload)
            S.t_b = time.perf_counter()
            S.whole = S.syn.batch(copy=False)
            S.my_idx = shard.deal(shard.estimate_cells(S.whole), world)[rank]
            S.batch = S.whole.subset(S.my_idx)
        else:
            S.syn = make_workload(api, args.n_sc, shard.rank_seed(args.seed + 104729 * k, rank), args.workload)
            S.t_b = time.perf_counter()
            S.batch = S.syn.batch(copy=False)   # host marshalling = the four generate_ptrs_strs calls per supercluster
        S.t_c = time.perf_counter()
        # (library defaults: all four alignments of every supercluster are computed)
        S.pr = api.PrecisionRecall(device=local_rank)
        S.t_c1 = time.perf_counter()
        S.pr.upload(S.batch)                # inputs resident in HBM before the timed region (+ K0 prep kernels)
        S.t_c2 = time.perf_counter()
        cls = S.syn.var_class()             # SNP / INDEL / SV class of every variant (print.cpp:362-372), resident too
        if strong:
            cls = [shard.subset_per_variant(cls[s], S.whole.var_off[s], S.my_idx) for s in range(4)]
        summary.upload_var_class(S.pr, cls)
        S.t_d = time.perf_counter()
        S.host_res = None
        return S

    slots = [make_slot(k) for k in range(n_fl)]
    S0 = slots[0]
    if strong:
        args.n_sc = S0.batch.n_sc
    batch, pr = S0.batch, S0.pr
    t_a, t_b, t_c, t_c1, t_c2, t_d = S0.t_a, S0.t_b, S0.t_c, S0.t_c1, S0.t_c2, S0.t_d
    in_bytes = sum(a.nbytes for h in range(4) for a in (batch.hap_seq[h], batch.hap_ptr[h], batch.hap_flag[h],
                                                        batch.hap_off[h], batch.var_off[h], batch.var_pos[h],
                                                        batch.var_qual[h])) + batch.ref_seq.nbytes + \
        batch.ref_off.nbytes + sum(a.nbytes for h in range(2) for a in (batch.ref_ptr[h], batch.ref_flag[h]))

    import threading
    parts = [0.0, 0.0, 0.0]            # host-side seconds in execute / download / counters + collective, summed over the timed steps
    lock = threading.Lock()
    turn = threading.Condition()
    # several ranks: the collectives of step i are issued behind those of step i - 1 on every rank (every rank must issue them in
    # one order); one rank: the batches in flight only take turns at the communicator
    ordered = world > 1 or bool(os.environ.get("BENCH_ORDERED"))
    coll_lock = threading.Lock()
    next_coll = [0]
    failed = []                         # exceptions of the step threads: the others stop waiting for their turn

    step_log = []                       # (step, start, end) of every timed step, seconds on this rank's clock

    def step(i, S):
        ta = time.perf_counter()
        S.pr.execute()                  # K1..K5 on the device
        tb = time.perf_counter()
        S.host_res = res = S.pr.download(S.host_res)   # final results to (reused) host buffers
        tc = time.perf_counter()
        if ordered:
            with turn:
                while next_coll[0] != i and not failed:
                    turn.wait(timeout=1.0)
        else:
            coll_lock.acquire()
        if failed:
            if not ordered:
                coll_lock.release()
            raise RuntimeError("another step failed")
        pb = None
        if strong:      # a contig's superclusters sit on all ranks: all-gather (sc_phase, orig, swap), phase redundantly
            if native_comm is not None:
                sc_phase, _, _ = rccl.allgather_phase(S.pr, native_comm, S.my_idx, S.whole.n_sc)
            else:
                sc_phase, _, _ = shard.allgather_phase(res, S.my_idx, S.whole.n_sc, device=None if gloo else dev)
            pb = summary.phase(sc_phase, np.ones(S.whole.n_sc, np.int32))[0][S.my_idx]
        if native_comm is not None:
            # the one collective of the path: the precision/recall counters [2][4][3][61] (int64 sum), all-reduced on the device
            # between the histogram kernel and the copy to the host
            t = torch.from_numpy(rccl.allreduce_counts(S.pr, native_comm, None, pb))
        else:
            t = torch.from_numpy(summary.pr_counts(S.pr, None, pb)).to(dev)   # [2][4][3][61] int64, device histogram
            if gloo:
                th = t.cpu(); dist.all_reduce(th); t = th.to(dev)
            elif dist is not None:
                dist.all_reduce(t)
        if ordered:
            with turn:
                next_coll[0] = i + 1
                turn.notify_all()
        else:
            coll_lock.release()
        te = time.perf_counter()
        with lock:
            parts[0] += tb - ta; parts[1] += tc - tb; parts[2] += te - tc
            step_log.append((i, ta, te))
        return res, t

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    kern_ms = []
    stats_acc = {}
    last = {}

    host_acc = {"n_device_allocs": 0, "n_device_frees": 0, "n_host_allocs": 0, "ms_host_alloc": 0.0, "ms_host_blocked": 0.0,
                "execute_wall_ms": []}

    acct_s = [0.0]

    units_acc = {}          # alignments in the main launches of a kernel over the timed steps, summed like stats_acc
    stats_main = stats_acc

    def account(S, stats_acc=stats_acc, kern_ms=kern_ms, host_acc=host_acc):
        t_acc = time.perf_counter()
        try:
            return account_(S, stats_acc, kern_ms, host_acc)
        finally:
            acct_s[0] += time.perf_counter() - t_acc

    def account_(S, stats_acc, kern_ms, host_acc):
        tm = S.pr.timing()
        kern_ms.append(tm.ms_total)
        if host_acc is not None:
            for k in ("n_device_allocs", "n_device_frees", "n_host_allocs", "ms_host_alloc", "ms_host_blocked"):
                host_acc[k] += getattr(tm, k)
            host_acc["execute_wall_ms"].append(tm.ms_wall)
        # a kernel runs in several roles per step (round 0 over the whole part, retry and tie rounds over a few
        # alignments): the launch of a step with the most units is the kernel's main launch, the rest is "other"
        ls = S.pr.launch_stats()
        top = {}
        for s_ in ls:
            k = (s_.kind, s_.kernel.decode())
            top[k] = max(top.get(k, 0), s_.n_units)
        for s_ in ls:
            k = (s_.kind, s_.kernel.decode())
            main = s_.n_units * 2 >= top[k]
            a = stats_acc.setdefault(k, [0, 0.0, 0, 0, 0, 0, 0.0])
            if main:
                a[0] += 1; a[1] += s_.ms; a[2] += s_.bytes_algorithmic; a[3] += s_.cells; a[4] += s_.cells_dense
                if stats_acc is stats_main:
                    units_acc[k] = units_acc.get(k, 0) + s_.n_units
            else:
                a[5] += 1; a[6] += s_.ms

    # Batches in flight start half a step apart and the order of the collectives (step i behind step i - 1) keeps them there: one
    # batch's bulk kernels then run beside the other's latency tail.  Started together they stay in lock step -- both bulks at
    # once, both tails at once: 28 ms per step instead of 22.  The delay is inside the timed region.
    stagger_s = [0.0]

    def run_steps(first, n, timed):
        """steps first .. first + n - 1; with more than one batch in flight, step i runs on slot i % n_fl from that slot's own
        host thread (vpr_execute blocks its caller), so a batch's latency tail -- its few longest alignments are chains of
        sequential rows -- overlaps the bulk of the next batch; every step is still one complete pass over one batch"""
        def worker(j):
            try:
                if dist is not None and not gloo:
                    torch.cuda.set_device(local_rank)
                if j and stagger_s[0] > 0:      # (see stagger_s)
                    time.sleep(j * stagger_s[0])
                for i in range(first + j, first + n, n_fl):
                    r = step(i, slots[j])
                    if timed:
                        with lock:
                            account(slots[j])
                            last[i] = (r, slots[j])
            except BaseException as e:      # (a failed step must not leave the other threads waiting for its turn)
                with turn:
                    failed.append(e)
                    turn.notify_all()
        if n_fl == 1:
            worker(0)
        else:
            ths = [threading.Thread(target=worker, args=(j,)) for j in range(n_fl)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
        if failed:      # (no interpreter teardown: after a device error the runtime's own cleanup can block for ever)
            sys.stderr.write(f"bench step failed: {failed[0]!r}\n")
            sys.stderr.flush()
            os._exit(1)

    import gc
    run_steps(0, max(args.warmup, 0), False)
    # the stall guard: a warm-up step (behind the handles' first ones, which allocate) that took several times the others means
    # the handles are in each other's way on this box -- the timed steps then run one batch at a time
    stall_guard = None
    if n_fl > 1 and len(step_log) > n_fl + 1:
        wl = np.array([(e - a) * 1e3 for _, a, e in sorted(step_log)][n_fl:])
        if wl.max() > max(6.0 * float(np.median(wl)), 250.0):
            # (round 3's stall was the THIRD handle's: with three in flight the timed steps fall back to two, which never showed it)
            n_fl = 2 if n_fl > 2 else 1
            stall_guard = {"warmup_step_ms": [round(float(x), 1) for x in wl],
                           "action": "timed steps with two batches in flight" if n_fl == 2 else "timed steps one batch at a time"}
    if dist is not None and world > 1:      # (all ranks alike: the collectives are issued in step order)
        tg = torch.tensor([1 if stall_guard else 0], device="cpu" if gloo else dev)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        if int(tg.item()):
            n_fl = 1
    if n_fl > 1 and len(step_log) > n_fl:
        env_st = os.environ.get("BENCH_STAGGER_MS")
        lat = np.array([(e - a) * 1e3 for _, a, e in sorted(step_log)][n_fl:])
        stagger_s[0] = (float(env_st) if env_st else 0.5 * float(np.median(lat)) / n_fl) * 1e-3
    parts[:] = [0.0, 0.0, 0.0]
    step_log.clear()
    # (the interpreter's cyclic collector otherwise runs a full collection inside one of the timed steps -- always the same one,
    # +38 ms -- over the millions of objects the setup left behind: collected now, switched off while the steps are timed)
    gc.collect()
    gc.disable()
    sync()
    t0 = time.perf_counter()
    run_steps(args.warmup, args.steps, True)
    sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    timed_log = sorted(step_log)
    if os.environ.get("BENCH_STEP_LOG") and rank == 0:
        sys.stderr.write("step ms: " + " ".join(f"{(e - a) * 1e3:.1f}" for _, a, e in timed_log) + "\n")
        sys.stderr.write("execute wall ms: " + " ".join(f"{x:.1f}" for x in host_acc["execute_wall_ms"]) + "\n")
        sys.stderr.write("kernel ms: " + " ".join(f"{x:.1f}" for x in kern_ms) + "\n")
    (res, t), S_last = last[args.warmup + args.steps - 1]
    batch, pr = S_last.batch, S_last.pr
    # two more steps of one batch ALONE (not timed, not in `value`): a kernel's launch duration without the neighbours that
    # stretch it while batches are in flight -- the figure a rocprofv3 trace of `--in-flight 1` shows
    alone_acc, alone_ms = {}, []
    one_at_a_time = None
    if n_fl > 1:
        n_al = 6
        sync()
        t_al = time.perf_counter()
        for k in range(n_al):
            step(args.warmup + args.steps + k, S_last)
            account(S_last, alone_acc, alone_ms, None)
        sync()
        dt_al = time.perf_counter() - t_al
        one_at_a_time = {"value": round(4 * args.n_sc * world / (dt_al / n_al), 1), "unit": "supercluster-alignments/s", "steps": n_al,
                         "ms_per_step": round(dt_al / n_al * 1e3, 3), "kernel_ms_per_step": round(float(np.mean(alone_ms)), 3),
                         "note": "rank 0's figure x ranks; the same steps strictly one after the other on one batch (the latency of a "
                                 "lone step: DESIGN.md section 6)"}
    if rank == 0:       # the device tally must equal the one recomputed from the downloaded results
        assert np.array_equal(shard.tally_from_results(res, batch.var_off), pr.tally()), "device tally != host tally"
        # after the timed region: per-contig phasing (host Viterbi) and the PRECISION-RECALL SUMMARY of this rank
        pb, sw, fl = summary.phase(res.sc_phase, np.ones(batch.n_sc, np.int32))
        rows = summary.pr_summary(summary.pr_counts(pr, None, pb))
    # every rank's own time per step and the part of it spent in the counters + collective (host side of the call), gathered
    per_rank = {"ms_per_step": [round(elapsed / args.steps * 1e3, 3)], "collective_ms_per_step": [round(parts[2] / args.steps * 1e3, 3)]}
    if dist is not None:
        tg_ = torch.tensor([elapsed / args.steps * 1e3, parts[2] / args.steps * 1e3], dtype=torch.float64,
                           device="cpu" if dist.get_backend() == "gloo" else dev)
        all_ = [torch.zeros_like(tg_) for _ in range(world)]
        dist.all_gather(all_, tg_)
        per_rank = {"ms_per_step": [round(float(x[0]), 3) for x in all_], "collective_ms_per_step": [round(float(x[1]), 3) for x in all_]}
        te = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    total_aln = 4 * (args.n_sc_total if strong else args.n_sc * world) * args.steps
    value = total_aln / elapsed

    # ---- one pass per batch (N = 1, weak): what a run over fresh superclusters costs.  Every step starts from the variant
    # tables in host memory: sizing + checks on the host, variant tables and contig over the link, generate_ptrs_strs on the
    # device (pr_gen.hip), position constants, planning, then the same execute + download + counters as above; batches in
    # flight overlap one batch's upload and planning with another's kernels.
    one_pass = None
    tm_main = pr.timing()
    if world == 1 and not strong and args.one_pass_batches > 0:
        nb = args.one_pass_batches
        op_parts = [0.0, 0.0, 0.0]
        n_op = args.one_pass_in_flight if args.one_pass_in_flight else (3 if args.workload == "wgs_synth" else n_fl)
        op_slots = list(slots)
        while len(op_slots) < n_op:         # (further batches, only for this leg: after the headline's timed region)
            op_slots.append(make_slot(len(op_slots)))
        op_slots = op_slots[:max(n_op, 1)]
        n_op = len(op_slots)

        for S in op_slots:                  # (the SNP / INDEL / SV class is a column of the variant tables: print.cpp:362-372)
            S.cls = S.syn.var_class()

        def op_step(S):
            ta = time.perf_counter()
            S.pr.upload_variants(S.syn.struct, S.batch)
            summary.upload_var_class(S.pr, S.cls)
            tb = time.perf_counter()
            S.pr.execute()
            tc = time.perf_counter()
            S.host_res = S.pr.download(S.host_res)  # (the page-locked block of the first batch: same shape, same column offsets)
            summary.pr_counts(S.pr, None, None)
            with lock:
                op_parts[0] += tb - ta; op_parts[1] += tc - tb; op_parts[2] += time.perf_counter() - tc

        def op_run(n):
            def worker(j):
                try:
                    for i in range(j, n, n_op):
                        op_step(op_slots[j])
                except BaseException as e:
                    failed.append(e)
            ths = [threading.Thread(target=worker, args=(j,)) for j in range(n_op)]
            for th in ths:
                th.start()
            for th in ths:
                th.join()
        op_run(2 * n_op)                    # warm-up: two batches per slot (the first execute of a handle settles its workspaces)
        if failed:
            sys.stderr.write(f"one-pass leg failed: {failed[0]!r}\n")
            sys.stderr.flush()
            os._exit(1)
        op_parts[:] = [0.0, 0.0, 0.0]
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        op_run(nb)
        if failed:
            sys.stderr.write(f"one-pass leg failed: {failed[0]!r}\n")
            sys.stderr.flush()
            os._exit(1)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t1
        one_pass = {"value": round(4 * args.n_sc * nb / dt, 1), "unit": "supercluster-alignments/s", "batches": nb, "in_flight": n_op,
                    "ms_per_batch": round(dt / nb * 1e3, 2),
                    "host_thread_ms_per_batch": {"upload_variants": round(op_parts[0] / nb * 1e3, 2), "vpr_execute": round(op_parts[1] / nb * 1e3, 2),
                                                 "download_and_counters": round(op_parts[2] / nb * 1e3, 2)},
                    "note": "every batch from its variant tables in host memory: host sizing pass, upload, generate_ptrs_strs on the "
                            "device, position constants, planning, execute, download, counters"}
        for S in op_slots:                  # (the timed batches replaced the resident ones; what follows reads the last step's results)
            S.host_res = None
        S = None
        del op_slots[len(slots):]           # (the leg's own handles: released)
        gc.collect()

    # ---- legs after the headline (rank 0 at N = 1 only; nothing here enters `value`)
    two_fl = None
    secondary = []
    if world == 1 and not strong and not args.no_secondary and args.workload == "wgs_synth":
        # (a) two batches in flight behind two handles, when the headline ran one at a time (--in-flight 1 or the stall guard)
        try:
            if n_fl > 1:
                raise StopIteration
            S2 = make_slot(1)
            pair = [slots[0], S2]
            lat = []

            def tf_worker(j, first, n, timed):
                for i in range(first + j, first + n, 2):
                    ta = time.perf_counter()
                    pair[j].pr.execute()
                    pair[j].host_res = pair[j].pr.download(pair[j].host_res)
                    summary.pr_counts(pair[j].pr, None, None)
                    if timed:
                        with lock:
                            lat.append((time.perf_counter() - ta) * 1e3)

            def tf_run(first, n, timed):
                ths = [threading.Thread(target=tf_worker, args=(j, first, n, timed)) for j in range(2)]
                for th in ths:
                    th.start()
                for th in ths:
                    th.join()
            n_tf = max(args.steps, 12)
            tf_run(0, 4, False)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            tf_run(4, n_tf, True)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t1
            two_fl = {"value": round(4 * args.n_sc * n_tf / dt, 1), "unit": "supercluster-alignments/s", "in_flight": 2, "steps": n_tf,
                      "ms_per_step": round(dt / n_tf * 1e3, 3),
                      "step_latency_ms": {"p50": round(float(np.percentile(lat, 50)), 3), "max": round(float(np.max(lat)), 3)},
                      "note": "every step still a complete pass over one batch; not the headline: `value` is measured one batch at a time"}
            del S2
        except StopIteration:
            two_fl = None
        except Exception as e:      # (a leg must not cost the headline)
            two_fl = {"error": repr(e)}
        # (b) the other single-GPU configurations of BASELINE.json on small batches.  The headline's other resident batches go
        # first: every live handle holds a dozen streams, and these legs -- latency chains on a handle of their own -- ran 1.6 x
        # slower beside three idle handles than beside two (sv_synth 0.60 - 0.67 s against 0.37)
        # (round 6: the last batch's handle too -- everything the line reports about it has been read by now --: the stress leg
        # ran 307 ms per step beside that one idle handle and 284 alone)
        S = None
        for S_ in slots:
            S_.pr.close()
        pr = None
        gc.collect()
        for wl, n_sc_, st_ in (("sv_synth", 200, 2), ("stress_synth", 20000, 2), ("joint_synth", 100000, 2)):
            try:
                secondary.append(secondary_leg(api, summary, wl, n_sc_, args.seed, st_, local_rank,
                                               cpu_target_s=0.0 if args.no_cpu_baseline else 5.0))
            except Exception as e:
                secondary.append({"workload": wl, "error": repr(e)})

    if rank == 0:
        tm = tm_main
        # dominant K1/K2 kernel (the DP sweeps the byte model of SURVEY 8(d) is about): algorithmic bytes per launch
        # / average launch duration (HIP events on the stream the kernel is launched on)
        # Dominant = the largest accumulated launch time among the sweep kernels whose launches are THROUGHPUT work -- on average
        # at least 1 % of the batch's alignments per launch.  (A launch over a handful of long alignments is a chain of dependent
        # rows: its duration is a latency, and dividing its few bytes by it says nothing about the memory system; with the long
        # part starting at 1 024 rows the accumulated time of such chains can exceed the bulk kernel's.)  No such kernel -- the
        # workloads of long alignments -- : the largest accumulated time.
        sweeps = {k: v for k, v in stats_acc.items() if k[0] in (1, 2) and v[0] > 0}
        bulk = {k: v for k, v in sweeps.items() if units_acc.get(k, 0) / v[0] >= 0.01 * 4 * args.n_sc}
        # Among the throughput kernels the dominant one is the one that moves the most ALGORITHMIC BYTES per step (SURVEY 8(d)'s
        # model: a byte per swept cell + the inputs) -- the roofline question is asked of the kernel that carries the path's
        # traffic.  (Until the end of round 6 the longest accumulated launch time decided, then the longest launch alone: the lane
        # kernel, 2.8 - 2.9 ms and 5.8 GB a step, and the 16-cell forward kernel, three launches of 2.9 - 3.0 ms and 0.5 GB together,
        # are a few per cent apart, and the line's `roofline.kernel` flipped between runs of one build.  The kernel with the largest
        # accumulated launch time is `by_time` below, whatever it sweeps.)  Ties: the longer launch alone.
        def _weight(kv):
            a_ = alone_acc.get(kv[0])
            t_ = (a_[1] / a_[0]) * (kv[1][0] / max(args.steps, 1)) if a_ and a_[0] > 0 else kv[1][1] / max(args.steps, 1)
            return (kv[1][2] / max(args.steps, 1), t_)
        (kind, kname), (nl, ms, byt, cells, dense, _, _) = max((bulk or sweeps).items(), key=_weight)
        avg_s = ms / nl * 1e-3
        # Three byte counts for that launch: (1) what the PMC counters of the committed rocprofv3
        # passes of this very command saw (FETCH_SIZE x 2 + WRITE_SIZE): `traffic`, and `achieved` = traffic / the
        # launch duration measured here -- the fraction of the HBM roofline the kernel really uses; (2) the bytes of the
        # cells it sweeps (1 B per swept cell + inputs): `swept`; (3) SURVEY 8(d)'s dense-equivalent figure, 1 B per cell
        # of the dense matrices it does not have to sweep: `dense_equivalent` -- a measure of what the windows save,
        # not of how busy HBM is.
        in_b = (byt - cells) if kind == 1 else 0
        dense_bytes = (dense + in_b) / nl
        dense_gbs = dense_bytes / avg_s / 1e9
        swept_gbs = (byt / nl) / avg_s / 1e9
        traffic = valu_frac = None
        prof_src = None
        lds_info = None
        # SIMDs and shader clock of the device this run is on (the fallbacks are MI355X's: 256 CUs x 4 SIMDs, 2.4 GHz)
        props = torch.cuda.get_device_properties(dev)
        n_simd = 4 * int(getattr(props, "multi_processor_count", 0) or N_SIMD // 4)
        sclk_hz = float(getattr(props, "clock_rate", 0) or 0) * 1e3 or SCLK_HZ
        try:
            # FETCH_SIZE counts half of wide coalesced reads on gfx950 (MI355X_MICROARCH.md): x2; KB -> bytes.  VALU issue:
            # quad-cycles a wave spends issuing VALU instructions x waves, over the SIMD-quad-cycles of the launch
            cr = counter_roofline(kname, args.workload, args.n_sc, avg_s, n_simd, sclk_hz)
            if cr:
                k, prof = cr["k"], cr["prof"]
                traffic, valu_frac, prof_src = cr["traffic"], cr["valu_issue_frac"], cr["source"]
                # LDS side: instructions per wave and the share of a wave's cycles in which it issues LDS instructions
                lds_info = {"lds_insts_per_wave": k.get("lds_insts_per_wave"),
                            "lds_active_frac_of_wave_cycles": (round(k["lds_active_per_wave"] / k["wave_cycles_per_wave"], 5)
                                                               if k.get("lds_active_per_wave") is not None and k.get("wave_cycles_per_wave") else None),
                            "other_kernels": {kn: {"lds_insts_per_wave": kv.get("lds_insts_per_wave"),
                                                   "lds_active_frac_of_wave_cycles": round(kv["lds_active_per_wave"] / kv["wave_cycles_per_wave"], 5)}
                                              for kn, kv in prof["kernels"].items()
                                              if kv.get("lds_active_per_wave") and kv.get("wave_cycles_per_wave") and kv.get("lds_insts_per_wave", 0) > 100},
                            "note": "the lane kernel keeps its state in registers and streams through HBM: LDS is not on its path; the window "
                                    "kernels stage rows and constants in LDS and exchange cells through ds_bpermute / DPP: LDS instructions "
                                    "issue in 1 - 3 % of a wave's cycles (SQ_ACTIVE_INST_LDS / SQ_WAVE_CYCLES), far from the LDS peak"}
        except (OSError, KeyError, ValueError):
            pass
        achieved = (traffic / avg_s / 1e9) if traffic else swept_gbs
        hbm_frac = achieved / HBM_PEAK_GBS
        roof = {
            "bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(hbm_frac, 5), "traffic": traffic,
            "valu_issue_frac": None if valu_frac is None else round(valu_frac, 4),
            "limiter": None if valu_frac is None else ("valu_issue" if valu_frac > hbm_frac else "hbm"),
            "launches": nl, "avg_launch_ms": round(ms / nl, 4),
            "swept": {"bytes_per_launch": int(byt / nl), "cells_per_launch": int(cells / nl), "GB/s": round(swept_gbs, 2),
                      "frac": round(swept_gbs / HBM_PEAK_GBS, 5)},
            "dense_equivalent": {"bytes_per_launch": int(dense_bytes), "cells_per_launch": int(dense / nl),
                                 "GB/s": round(dense_gbs, 2), "frac": round(dense_gbs / HBM_PEAK_GBS, 5)},
            "counters_source": prof_src,
            "lds": lds_info,
            "device": {"name": props.name, "simds": n_simd, "sclk_hz": sclk_hz},
            "alone": None,
            "note": "frac = HBM traffic of the dominant sweep kernel (PMC counters) / its launch duration (HIP events, this run) / "
                    "8 TB/s; the kernel is an integer scan that is issue- and latency-bound, not HBM-bound: see valu_issue_frac "
                    "and DESIGN.md section 6" if traffic else
                    "no committed counters for this workload / size: frac falls back to the bytes of the swept cells",
        }
        if (kind, kname) in alone_acc and alone_acc[(kind, kname)][0] > 0:
            a_ = alone_acc[(kind, kname)]
            a_s = a_[1] / a_[0] * 1e-3
            a_bytes = (traffic if traffic else a_[2] / a_[0])
            roof["alone"] = {"avg_launch_ms": round(a_[1] / a_[0], 4), "achieved": round(a_bytes / a_s / 1e9, 2),
                             "frac": round(a_bytes / a_s / 1e9 / HBM_PEAK_GBS, 5), "step_ms": round(float(np.mean(alone_ms)), 3),
                             "note": "the same kernel in two extra steps of one batch with nothing else in flight (kernel time of the step: step_ms)"}
        # the kernel with the largest accumulated device time over the timed steps, main and other launches together, whatever it
        # is (a launch over a handful of long alignments is a latency chain: `frac` of such a kernel says how little of HBM a
        # chain of dependent rows uses, not how busy the device is) -- so that the line cannot hide a slow kernel
        (bt_kind, bt_name), bt = max(stats_acc.items(), key=lambda kv: kv[1][1] + kv[1][6])
        bt_ms = bt[1] + bt[6]
        bt_nl = bt[0] + bt[5]
        bt_cr = counter_roofline(bt_name, args.workload, args.n_sc, max(bt[1], 1e-9) / max(bt[0], 1) * 1e-3, n_simd, sclk_hz)
        roof["by_time"] = {
            "kernel": bt_name, "ms_per_step": round(bt_ms / args.steps, 3), "launches_per_step": round(bt_nl / args.steps, 2),
            "share_of_kernel_ms": round(bt_ms / max(sum(v[1] + v[6] for v in stats_acc.values()), 1e-9), 4),
            "main_launch_avg_ms": round(bt[1] / max(bt[0], 1), 4),
            "traffic_main_launch": bt_cr["traffic"] if bt_cr else None,
            "frac_main_launch": round(bt_cr["traffic"] / (bt[1] / max(bt[0], 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if bt_cr and bt[1] > 0 else
                                (round((bt[2] / max(bt[0], 1)) / (bt[1] / max(bt[0], 1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if bt[1] > 0 and bt[2] > 0 else None),
            "valu_issue_frac_main_launch": None if not bt_cr or bt_cr["valu_issue_frac"] is None else round(bt_cr["valu_issue_frac"], 4),
            "counter_kernel": bt_cr["counter_kernel"] if bt_cr else None,
            "note": "largest accumulated launch time of any kernel over the timed steps (the throughput-dominant sweep kernel is `kernel` above)"}
        per_kernel = {k[1]: {"launches": v[0], "ms": round(v[1], 3), "other_launches": v[5], "other_ms": round(v[6], 3)}
                      for k, v in sorted(stats_acc.items())}
        out = {
            "metric": "supercluster-alignments/sec", "value": round(value, 1), "unit": "supercluster-alignments/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "int32", "data": "synthetic",
            "config": {"workload": args.workload, "superclusters_per_gpu": args.n_sc,
                       "span_dist": {"wgs_synth": "lognormal(median 20, sigma 1.2) clip [4,10000]", "stress_synth": "loguniform [32,16384]",
                                     "sv_synth": "loguniform [2000,12000], indels geometric mean 600",
                                     "joint_synth": "lognormal(median 20, sigma 1.2) clip [4,10002], 0.75 % of the superclusters with one SV-sized indel (50..10000)"}.get(args.workload),
                       "sharding": (f"{args.n_sc_total} superclusters dealt over {world} ranks by estimated cells, phasing all-gathered"
                                    if strong else f"{world} ranks x independent superclusters")},
            "dense_cells_per_s": round(tm.cells_dense * world * args.steps / elapsed, 1),
            "kernel_ms_per_step": round(float(np.mean(kern_ms)), 3),
            # every timed step on this rank: its own duration (start of vpr_execute -> counters reduced; with batches in flight
            # the steps overlap, so this is a step's latency, not the time per step) and the spacing of the completions
            "step_ms": (lambda d: {"p50": round(float(np.percentile(d, 50)), 3), "p95": round(float(np.percentile(d, 95)), 3),
                                   "max": round(float(d.max()), 3), "min": round(float(d.min()), 3)})(
                np.array([(e - a) * 1e3 for _, a, e in timed_log])),
            "step_completion_interval_ms": (lambda d: {"p50": round(float(np.percentile(d, 50)), 3), "p95": round(float(np.percentile(d, 95)), 3),
                                                       "max": round(float(d.max()), 3)})(
                np.diff(np.sort(np.array([t0] + [e for _, _, e in timed_log]))) * 1e3),
            "in_flight": n_fl,
            "collective": collective, "comm_ranks": comm_ranks,
            "per_rank": dict(per_rank, note="each rank's own wall clock per step over the timed region, and the host time of its counters + "
                                             "collective call per step (with batches in flight it overlaps the other batch's kernels)"),
            "bookkeeping_ms_per_step": round(acct_s[0] / max(args.steps, 1) * 1e3, 3),   # (reading the launch statistics: inside the timed region)
            "host": {"n_device_allocs": int(host_acc["n_device_allocs"]), "n_device_frees": int(host_acc["n_device_frees"]),
                     "n_host_allocs": int(host_acc["n_host_allocs"]), "ms_host_alloc": round(host_acc["ms_host_alloc"], 3),
                     "ms_host_blocked": round(host_acc["ms_host_blocked"], 3),
                     "execute_wall_ms_max": round(max(host_acc["execute_wall_ms"]), 3),
                     "note": "allocator calls and blocking waits inside the timed vpr_execute calls of rank 0 (vpr_timing): 0 once a "
                             "handle's workspaces have settled"},
            "step_parts_ms": {"vpr_execute": round(parts[0] / args.steps * 1e3, 3), "vpr_download": round(parts[1] / args.steps * 1e3, 3),
                              "counters_and_collective": round(parts[2] / args.steps * 1e3, 3)},
            "setup_not_timed": {"generate_s": round(t_b - t_a, 2), "host_marshalling_s": round(t_c - t_b, 2),
                                "upload_and_prep_s": round(t_d - t_c, 2),
                                "upload_and_prep_parts_s": {"vpr_create": round(t_c1 - t_c, 3), "vpr_upload": round(t_c2 - t_c1, 3),
                                                            "variant_classes": round(t_d - t_c2, 3)},
                                "input_bytes": int(in_bytes),
                                # one batch through upload + one step (vpr_create, a per-process cost, left out)
                                "pcie_inclusive_value": round(4 * args.n_sc / ((t_d - t_c1) + elapsed / args.steps), 1)},
            "kernel_only_value": round(4 * args.n_sc / (float(np.mean(kern_ms)) * 1e-3), 1),
            "end_to_end_value": None if not one_pass else one_pass["value"],      # = one_pass.value: every batch from its variant tables
            "lane_levels": {"zero_level_rejects_with_complete_wave_0": int(tm.n_lane1_seen), "finished_at_distance_1_lane_level": int(tm.n_lane1_finished),
                            "waves_dropped": int(tm.n_lane1_waves_dropped), "retries_incl_in_place_round": int(tm.n_band_retries),
                            "note": "pr_d1.hip: alignments the zero-distance lane kernel rejects with a complete wave 0 (at most VPR_D1_MAX_ROWS truth rows, "
                                    "default 256) run the distance-1 lane kernel; the rest re-runs in place with the 16-cell kernels"},
            "one_pass": one_pass,
            "two_in_flight": two_fl,
            "one_at_a_time": one_at_a_time,
            "stall_guard": stall_guard,
            "secondary": secondary,
            "kernels": per_kernel,
            "counts_at_min_qual_TP_FP_FN": t.cpu().numpy()[:, 3, :, 0].tolist(),   # [callset][TP,FP,FN], type ALL, all ranks
            "pr_summary_rank0": [{"type": summary.NAMES[r.vartype], "threshold": "BEST" if r.best else "NONE", "qual": r.qual,
                                  "truth_tp": r.truth_tp, "query_tp": r.query_tp, "truth_fn": r.truth_fn, "query_fp": r.query_fp,
                                  "precision": round(r.precision, 6), "recall": round(r.recall, 6), "f1": round(r.f1_score, 6)}
                                 for r in rows if not r.best],
            "phasing_rank0": {"switches": int(len(sw)), "flips": int(len(fl))},
            "tie_replays": {"alignments_with_consulted_ties": int((res.aln_status & 1).sum()), "replays_per_step": int(tm.n_tie_replays),
                            "replay_kernel_ms_per_step": round(tm.ms_tie, 3),
                            "note": "reference keeps the last writer of swap_pred (dist.cpp:347,376): replayed on the device, results exact"},
            "roofline": roof,
        }
        if not args.no_cpu_baseline and world == 1:      # (rank 0 at N = 1 only: the other ranks would wait for it)
            out["cpu_baseline"] = cpu_baseline(batch, target_s=8.0)
            out["cpu_baseline"]["host"] = f"{os.cpu_count()} logical cores visible, {cpu_limit()} allowed (cgroup quota)"
            if out["cpu_baseline"].get("calibrated_value"):
                out["cpu_baseline"]["calibrated_single_thread_value"] = round(out["cpu_baseline"]["calibrated_value"] / out["cpu_baseline"]["cores"], 1)
        # ---- the compact digest, LAST in the line (the driver keeps the line's tail): what the legs above measured
        def _sec(x):
            if "error" in x:
                return {"workload": x.get("workload"), "error": x["error"][:80]}
            r_ = x.get("roofline") or {}
            c_ = x.get("cpu_baseline") or {}
            return {"workload": x["workload"], "n_sc": x["superclusters"], "ms_per_step": x["ms_per_step"], "value": x["value"],
                    "kernel": r_.get("kernel"), "frac": r_.get("frac"), "traffic": r_.get("traffic"),
                    "cpu": c_.get("value"), "cpu_calibrated": c_.get("calibrated_value")}
        cb = out.get("cpu_baseline") or {}
        out["summary"] = {
            "value": out["value"], "ms_per_step": out["ms_per_step"], "in_flight": n_fl, "n_gpus": world,
            "comm_ranks": comm_ranks, "per_rank_ms_per_step": per_rank["ms_per_step"],
            "one_pass": None if not one_pass else {"value": one_pass["value"], "ms_per_batch": one_pass["ms_per_batch"],
                                                   "upload_variants_ms": one_pass["host_thread_ms_per_batch"]["upload_variants"]},
            "one_at_a_time_ms_per_step": None if not one_at_a_time else one_at_a_time["ms_per_step"],
            "roofline": {"kernel": roof["kernel"], "frac": roof["frac"], "traffic": roof["traffic"], "avg_launch_ms": roof["avg_launch_ms"],
                         "alone_ms": None if not roof["alone"] else roof["alone"]["avg_launch_ms"],
                         "alone_frac": None if not roof["alone"] else roof["alone"]["frac"], "valu_issue_frac": roof["valu_issue_frac"]},
            "by_time": {k_: roof["by_time"][k_] for k_ in ("kernel", "ms_per_step", "traffic_main_launch", "frac_main_launch", "valu_issue_frac_main_launch")},
            "tie_replays": {"per_step": int(tm.n_tie_replays), "kernel_ms_per_step": round(tm.ms_tie, 3)},
            "secondary": [_sec(x) for x in secondary],
            "cpu_baseline": {"value": cb.get("value"), "cores": cb.get("cores"), "single_thread": cb.get("single_thread_value"),
                             "calibrated_value": cb.get("calibrated_value"), "within_15_percent": cb.get("calibration_within_15_percent"),
                             "cpu_model": cb.get("cpu_model")},
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

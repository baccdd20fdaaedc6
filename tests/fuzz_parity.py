"""Differential fuzzing of the HIP path against the CPU oracle (test infrastructure): random workload shapes and
seeds, every result array compared bit for bit with tests/test_gpu_parity.compare.

    python tests/fuzz_parity.py [seconds] [first_seed] [shape]     # long runs, by hand on the GPU box

tests/test_gpu_parity.py::test_fuzz_smoke runs a few seconds of it inside the GPU suite."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def random_workload(seed, shape=None):
    """-> (shape, Synth keyword arguments, band_mode) of one random batch; shape 6 (huge alignments with SV-sized
    indels: the 1024-cell and dense levels, deferred edit distances beyond LDS) only on request, it is slow"""
    rng = np.random.default_rng(seed)
    drawn = int(rng.integers(0, 6))
    shape = drawn if shape is None else shape
    if shape == 0:      # WGS-like log-normal spans
        kw = dict(n_sc=int(rng.integers(2000, 12000)), len_mode=1, len_a=float(rng.uniform(8, 40)),
                  len_b=float(rng.uniform(0.8, 1.5)), len_min=4, len_max=int(rng.integers(200, 4000)))
    elif shape == 1:    # tiny, repeat-rich, indel-heavy
        hi = int(rng.integers(20, 120))
        kw = dict(n_sc=int(rng.integers(500, 4000)), len_a=5, len_b=hi, len_min=5, len_max=hi,
                  var_per_base=float(rng.uniform(0.02, 0.2)), p_snp=float(rng.uniform(0.1, 0.9)),
                  p_repeat=float(rng.uniform(0.2, 1.0)))
    elif shape == 2:    # mid sizes (64 / 256 cell windows)
        lo = int(rng.integers(60, 400))
        hi = lo + int(rng.integers(10, 1500))
        kw = dict(n_sc=int(rng.integers(40, 400)), len_a=lo, len_b=hi, len_min=lo, len_max=hi,
                  var_per_base=float(rng.uniform(0.003, 0.05)), p_repeat=float(rng.uniform(0.0, 0.6)))
    elif shape == 3:    # long alignments (latency chains, wide windows, dense fallback, deferred edit distances)
        lo = int(rng.integers(1500, 4000))
        hi = lo + int(rng.integers(10, 3000))
        kw = dict(n_sc=int(rng.integers(4, 20)), len_a=lo, len_b=hi, len_min=lo, len_max=hi,
                  var_per_base=float(rng.uniform(0.002, 0.02)), indel_mean=float(rng.uniform(2, 30)))
    elif shape == 4:    # mostly identical haplotypes (zero-distance level) with long indels
        kw = dict(n_sc=int(rng.integers(1000, 8000)), len_a=10, len_b=300, len_min=10, len_max=300,
                  p_keep=float(rng.uniform(0.9, 1.0)), p_drop=0.0, p_hom=float(rng.uniform(0.3, 1.0)),
                  indel_mean=float(rng.uniform(1, 12)), p_snp=float(rng.uniform(0.2, 0.9)))
    elif shape == 6:    # huge alignments, SV-sized indels
        lo = int(rng.integers(5000, 9000))
        hi = lo + int(rng.integers(100, 7000))
        kw = dict(n_sc=int(rng.integers(2, 6)), len_a=lo, len_b=hi, len_min=lo, len_max=hi,
                  var_per_base=float(rng.uniform(0.0005, 0.004)), indel_mean=float(rng.uniform(40, 600)),
                  p_snp=float(rng.uniform(0.2, 0.7)), p_keep=float(rng.uniform(0.6, 1.0)))
    elif shape == 7:    # the joint workload (BASELINE configs[3]): whole-genome mix, a fraction of the superclusters with one SV-sized indel
        kw = dict(n_sc=int(rng.integers(150, 500)), len_mode=1, len_a=float(rng.uniform(10, 40)), len_b=float(rng.uniform(0.8, 1.4)),
                  len_min=4, len_max=10002, p_sv=float(rng.uniform(0.02, 0.08)), sv_min=50, sv_max=int(rng.integers(300, 8000)),
                  p_keep=float(rng.uniform(0.7, 0.95)), p_drop=float(rng.uniform(0.0, 0.1)), p_hom=float(rng.uniform(0.3, 0.9)),
                  p_repeat=float(rng.uniform(0.1, 0.6)))
    else:               # everything perturbed (few zero-distance alignments)
        kw = dict(n_sc=int(rng.integers(500, 3000)), len_a=8, len_b=400, len_min=8, len_max=400,
                  p_keep=float(rng.uniform(0.3, 0.7)), p_drop=float(rng.uniform(0.1, 0.3)),
                  var_per_base=float(rng.uniform(0.01, 0.08)))
    kw["seed"] = seed
    band_mode = int(rng.choice([1, 1, 2])) if shape in (3, 6, 7) else int(rng.choice([1, 1, 1, 3, 2, 0]))
    return shape, kw, band_mode


def fuzz(budget_s, seed0, verbose=True, max_batches=None, shape=None):
    """Run random batches for budget_s seconds; raises AssertionError (with the seed) on the first mismatch."""
    import test_gpu_parity as T
    from vcfdist_amd import _abi as A, api
    t_end = time.time() + budget_s
    n_run = n_sc = n_tie = n_limit = 0
    seed = seed0
    while time.time() < t_end and (max_batches is None or n_run < max_batches):
        shape_, kw, band_mode = random_workload(seed, shape)
        syn = api.Synth(**kw)
        batch = syn.batch()
        try:
            # every other batch goes in as variant tables (the device writes the strings and pointer arrays)
            got, want, ntie, pr = T.compare(batch, A.default_config(band_mode=band_mode, flags=int(os.environ.get("VCFDIST_FUZZ_FLAGS", "0"))),
                                            variants_struct=syn.struct if seed % 2 else None)
        except AssertionError as e:
            raise AssertionError(f"fuzz mismatch: seed {seed} band_mode {band_mode} kw {kw}: {e}") from e
        except api.VprError as e:
            # the one documented refusal: an alignment that needs the dense kernels with
            # Lq + Lr beyond their LDS rows.  Anything else, or that message on a batch that fits, is a failure.
            big = max(max(batch.lens(k)[q] for q in (0, 1)) + batch.lens(k)[4] for k in range(batch.n_sc))
            if "too long for the dense kernels" in str(e) and big > 39000:
                n_limit += 1
                if verbose:
                    print(f"seed {seed} shape {shape_}: refused, dense-level size limit (largest Lq + Lr = {big})", flush=True)
                seed += 1
                continue
            raise AssertionError(f"fuzz: seed {seed} band_mode {band_mode} kw {kw}: unexpected {e}") from e
        n_run += 1
        n_sc += batch.n_sc
        n_tie += ntie
        if verbose:
            print(f"seed {seed} shape {shape_} band_mode {band_mode}: {batch.n_sc} sc, {batch.dense_cells():.2e} dense cells, "
                  f"{pr.timing().n_band_retries} retries, {ntie} ties decided other than by the largest source: ok", flush=True)
        seed += 1
    if n_limit and verbose:
        print(f"{n_limit} batches refused at the documented dense-level size limit")
    return n_run, n_sc, n_tie


if __name__ == "__main__":
    runs, scs, ties = fuzz(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 1000,
                           shape=int(sys.argv[3]) if len(sys.argv) > 3 else None)
    print(f"fuzz: {runs} batches, {scs} superclusters, {ties} ties decided other than by the largest source, no mismatch")

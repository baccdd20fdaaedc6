"""Regenerates tests/golden/regression_seed7.npz and regression_joint61.npz: inputs (generator parameters) and the ORACLE's outputs for a
small seeded batch.  These are regression vectors of the CPU restatement (oracle/pr_oracle.cpp), NOT outputs of
the reference: the reference cannot be built in this image.  Run from the repo root:
    python tests/golden/make_regression.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from vcfdist_amd import api  # noqa: E402

PARAMS = dict(n_sc=120, len_a=8, len_b=400, len_max=400, seed=7, var_per_base=0.03)
# the joint SNP + INDEL + SV shape (BASELINE configs[3]): whole-genome mix, an SV-sized indel in a tenth of the superclusters
PARAMS_JOINT = dict(n_sc=160, seed=61, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10002, p_sv=0.1, sv_min=50, sv_max=1500)
FIXTURES = {"regression_seed7.npz": PARAMS, "regression_joint61.npz": PARAMS_JOINT}


def make(params, fname):
    batch = api.Synth(**params).batch()
    ex = O.Extra(batch)
    r = O.run(batch, extra=ex)
    out = {"aln_dist": r.aln_dist, "aln_end_plane": r.aln_end_plane, "aln_beg_plane": r.aln_beg_plane,
           "aln_status": r.aln_status, "sc_phase": r.sc_phase, "orig_phase_dist": r.orig_phase_dist,
           "swap_phase_dist": r.swap_phase_dist, "nonmax_tie": ex.swap_used_conflict_nonmax}
    for h in range(4):
        for w in range(2):
            for name, _ in r.PER_VAR:
                out[f"{name}_{h}_{w}"] = getattr(r, name)[h][w]
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), fname), **out)


def main():
    for fname, params in FIXTURES.items():
        make(params, fname)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Golden vectors at the HEADLINE's regime, made by the REAL reference classes (build container only: needs /root/reference).

    python tests/golden/make_golden_dense.py

The headline bench (BASELINE.json configs[3]) runs minPts 20-50 on strips of a few hundred PETs, where cDBSCAN2's crowded
cells and Pareto edge points (cDBSCAN2.py:194-302) dominate; every other reference-made golden is sparse (minPts <= 20, chr21 /
150 k synthetic).  Input: cloops_amd.synth.synth_chrom(400000, 248956422 // 20, 3000) -- the density of chr1 of the 200 M-PET
genome on 1/20 of its length (regenerated from the seed by the tests; a checksum is stored).  Settings: the first run of every
eps of Hi-C mode 3 (pipe.py:337-340) with cuts of the size the chained sweep produces: (5000, 50, cut 0), (7500, 30, 5004),
(10000, 20, 6250).  Per setting and variant the label of every row (-1 = noise or removed by the cut filter, pipe.py:59-63; the
class sees the kept rows with their original row numbers as ids) -> dense400k_labels.npz (data only)."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refload  # noqa: E402
from cloops_amd.synth import synth_chrom  # noqa: E402

N, LENGTH, SEED = 400000, 248956422 // 20, 3000
SETTINGS = ((5000, 50, 0), (7500, 30, 5004), (10000, 20, 6250))


def main():
    assert refload.available(), "reference checkout missing"
    X, Y = synth_chrom(N, LENGTH, SEED)
    X = X.astype(np.int64); Y = Y.astype(np.int64)
    ids = np.arange(N, dtype=np.int64)
    out, meta = {}, {"n": N, "length": LENGTH, "seed": SEED, "input_sha1": hashlib.sha1(np.stack([X, Y]).astype(np.int32).tobytes()).hexdigest(), "runs": {}}
    for eps, minPts, cut in SETTINGS:
        keep = np.nonzero(Y - X >= cut)[0]
        for variant in ("v2", "v1", "block"):
            mat = np.stack([ids[keep], X[keep], Y[keep]], 1)
            lab = np.full(N, -1, np.int32)
            lab[keep] = refload.labels_dict_to_array(refload.ref_labels(variant, mat, eps, minPts), ids[keep])
            key = "%s_%d_%d_%d" % (variant, eps, minPts, cut)
            out[key] = lab
            meta["runs"][key] = {"n_in": int(len(keep)), "labelled": int((lab >= 0).sum()), "clusters": int(len(np.unique(lab[lab >= 0])))}
            print(key, meta["runs"][key], flush=True)
    np.savez_compressed(os.path.join(HERE, "dense400k_labels.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "dense400k_meta.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

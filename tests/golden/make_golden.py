#!/usr/bin/env python
"""Generates the golden vectors under tests/golden/ from the REAL reference classes.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

What is stored is DATA: inputs (X, Y[, ids]) and the label arrays / cut values the
reference produced for them -- never reference source.  The chr21 input is derived from
the reference's bundled example BEDPE with the PET rule of cLoops/io.py:49-57
(swap so the left mid-point is the smaller one; floor mid-points; id = row counter,
io.py:181-183).
"""
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import refload  # noqa: E402
import cases  # noqa: E402
from cloops_amd.synth import synth_chrom  # noqa: E402

BEDPE = os.path.join(refload.REF_ROOT, "examples", "GSM1872886_GM12878_CTCF_ChIA-PET_chr21_hg38.bedpe.gz")


def parse_bedpe_chr21():
    xs, ys = [], []
    with gzip.open(BEDPE, "rt") as fh:
        for line in fh:
            f = line.split("\n")[0].split("\t")
            if "*" in f and "-1" in f:
                continue
            if len(f) < 6:
                continue
            try:
                sa, ea, sb, eb = int(f[1]), int(f[2]), int(f[4]), int(f[5])
            except ValueError:
                continue
            if f[0] != f[3]:
                continue
            if sa + ea > sb + eb:
                sa, sb = sb, sa
                ea, eb = eb, ea
            xs.append((sa + ea) // 2)
            ys.append((sb + eb) // 2)
    return np.asarray(xs, dtype=np.int64), np.asarray(ys, dtype=np.int64)


def ref_arr(variant, ids, X, Y, eps, minPts):
    mat = np.stack([ids, X, Y], 1)
    return refload.labels_dict_to_array(refload.ref_labels(variant, mat, eps, minPts), ids)


def ref_single(variant, ids, X, Y, eps, minPts, cut):
    """pipe.py:52-110 driven by the real class -> (labels over unfiltered rows, dis, dss)."""
    d = Y - X
    keep = np.arange(len(X))
    dss = []
    if cut > 0:
        keep = np.where(d >= cut)[0]
        dss.append(d[d < cut])
    lab = np.full(len(X), -1, np.int32)
    lab[keep] = ref_arr(variant, ids[keep], X[keep], Y[keep], eps, minPts)
    dis = []
    nI = nS = 0
    for c in np.unique(lab[lab >= 0]):
        m = lab == c
        x0, x1, y0, y1 = X[m].min(), X[m].max(), Y[m].min(), Y[m].max()
        if x0 == x1 or y0 == y1:
            continue
        if x1 < y0:
            nI += 1
            dis.append(d[m])
        else:
            nS += 1
            dss.append(d[m])
    dis = np.concatenate(dis) if dis else np.zeros(0, np.int64)
    dss = np.concatenate(dss) if dss else np.zeros(0, np.int64)
    return lab, dis, dss, nI, nS


def main():
    assert refload.available(), "reference checkout missing"
    meta = {}
    X, Y = parse_bedpe_chr21()
    ids = np.arange(len(X), dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "chr21_input.npz"), X=X.astype(np.int32), Y=Y.astype(np.int32))
    meta["chr21"] = {"n": int(len(X)), "xmin": int(X.min()), "xmax": int(X.max()),
                     "ymin": int(Y.min()), "ymax": int(Y.max())}
    # --- un-chained (cut = 0) labels, all variants
    out = {}
    for variant in ("v1", "v2", "block"):
        for eps, minPts in ((500, 5), (1000, 5), (2000, 5), (5000, 20)):
            lab = ref_arr(variant, ids, X, Y, eps, minPts)
            out["%s_%d_%d" % (variant, eps, minPts)] = lab
            meta["chr21_%s_%d_%d" % (variant, eps, minPts)] = {
                "labelled": int((lab >= 0).sum()), "clusters": int(len(np.unique(lab[lab >= 0])))}
            print(variant, eps, minPts, meta["chr21_%s_%d_%d" % (variant, eps, minPts)], flush=True)
    np.savez_compressed(os.path.join(HERE, "chr21_labels.npz"), **out)
    # --- the mode-1 chain (pipe.py:247-275): eps 500,1000,2000 x minPts 5, cut carried
    ests = refload.ref_ests()
    chain = {}
    for variant in ("v2", "v1"):
        cut = 0
        steps = []
        for eps in (500, 1000, 2000):
            lab, dis, dss, nI, nS = ref_single(variant, ids, X, Y, eps, 5, cut)
            chain["%s_%d" % (variant, eps)] = lab
            step = {"eps": eps, "minPts": 5, "cut_in": int(cut), "n_in": int((Y - X >= cut).sum()),
                    "labelled": int((lab >= 0).sum()), "clusters": int(len(np.unique(lab[lab >= 0]))),
                    "inter": nI, "self": nS, "n_dis": int(len(dis)), "n_dss": int(len(dss))}
            if len(dis) and len(dss):
                cut2, frags = ests.estIntSelCutFrag(np.array(dis, dtype=float), np.array(dss, dtype=float))
                step["cut_out"] = int(cut2)
                step["frags"] = int(frags)
                cut = cut2
            steps.append(step)
            print(variant, step, flush=True)
        meta["chr21_chain_" + variant] = steps
    np.savez_compressed(os.path.join(HERE, "chr21_chain_labels.npz"), **chain)
    # --- small seeded families (inputs are regenerated from the seed by tests/cases.py)
    fam = {}
    for family, seed, ncase in (("adversarial", 0, 60), ("plain", 1, 60), ("clumpy", 2, 60)):
        rng = np.random.default_rng(seed)
        gen = getattr(cases, family + "_case")
        for k in range(ncase):
            cids, cX, cY, eps, minPts = gen(rng, k)
            for variant in ("v1", "v2", "block"):
                fam["%s_%d_%s" % (family, k, variant)] = ref_arr(variant, cids, cX, cY, eps, minPts)
            fam["%s_%d_in" % (family, k)] = np.stack([cids, cX, cY]).astype(np.int32)
            fam["%s_%d_par" % (family, k)] = np.asarray([eps, minPts], np.int32)
    np.savez_compressed(os.path.join(HERE, "families.npz"), **fam)
    # --- synthetic benchmark generator at reduced size (real reference labels)
    syn = {}
    sX, sY = synth_chrom(150000, 46709983, 424242)
    sids = np.arange(len(sX), dtype=np.int64)
    for variant in ("v1", "v2", "block"):
        for eps, minPts in ((2000, 5), (5000, 20)):
            lab = ref_arr(variant, sids, sX.astype(np.int64), sY.astype(np.int64), eps, minPts)
            syn["%s_%d_%d" % (variant, eps, minPts)] = lab
            meta["synth150k_%s_%d_%d" % (variant, eps, minPts)] = {
                "labelled": int((lab >= 0).sum()), "clusters": int(len(np.unique(lab[lab >= 0])))}
            print("synth", variant, eps, minPts, meta["synth150k_%s_%d_%d" % (variant, eps, minPts)], flush=True)
    syn["XY_sha_check"] = np.asarray([int(sX.astype(np.int64).sum() % (1 << 31)), int(sY.astype(np.int64).sum() % (1 << 31))])
    np.savez_compressed(os.path.join(HERE, "synth150k_labels.npz"), **syn)
    with open(os.path.join(HERE, "golden_meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()

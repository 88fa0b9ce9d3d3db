"""Seeded synthetic cis-PET generator -- the benchmark workload.

Per chromosome: 35 % loop PETs (40 PETs per anchor pair, sigma 300 bp around both
anchors, span log-uniform 20 kb..1 Mb), 30 % self-ligation PETs (20 per peak, distance
log-uniform 50..3000 bp), 35 % background (distance log-uniform 100 bp..L/2), random
row order, X <= Y (the swap rule of cLoops/io.py:49-57), int32 coordinates.
"""
import numpy as np

# hg38 chromosome lengths, chr1..chr22, chrX
HG38 = [
    ("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555),
    ("chr5", 181538259), ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636),
    ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
    ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345),
    ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
    ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895),
]


def _logu(rng, lo, hi, n):
    return np.exp(rng.uniform(np.log(lo), np.log(hi), n))


def synth_chrom(n, length, seed):
    """-> (X, Y) int32 arrays of n PETs on a chromosome of `length` bp."""
    rng = np.random.default_rng(seed)
    n_loop = int(round(n * 0.35))
    n_self = int(round(n * 0.30))
    n_bg = n - n_loop - n_self
    # loop PETs
    k = max(1, n_loop // 40)
    a = rng.uniform(0, max(1.0, length - 2e6), k)
    span = _logu(rng, 2e4, 1e6, k)
    which = rng.integers(0, k, n_loop)
    lx = a[which] + rng.normal(0, 300, n_loop)
    ly = a[which] + span[which] + rng.normal(0, 300, n_loop)
    # self-ligation PETs
    kp = max(1, n_self // 20)
    peaks = rng.uniform(0, length, kp)
    sx = peaks[rng.integers(0, kp, n_self)] + rng.normal(0, 300, n_self)
    sy = sx + _logu(rng, 50, 3000, n_self)
    # background
    bx = rng.uniform(0, length, n_bg)
    by = bx + _logu(rng, 100, length / 2, n_bg)
    x = np.concatenate([lx, sx, bx])
    y = np.concatenate([ly, sy, by])
    perm = rng.permutation(n)
    x = np.clip(np.rint(x[perm]), 0, length).astype(np.int64)
    y = np.clip(np.rint(y[perm]), 0, length).astype(np.int64)
    lo = np.minimum(x, y)
    hi = np.maximum(x, y)
    return lo.astype(np.int32), hi.astype(np.int32)


def chrom_sizes(n_total, chroms=None):
    """Split n_total PETs over chromosomes proportionally to hg38 length."""
    chroms = HG38 if chroms is None else chroms
    tot = float(sum(l for _, l in chroms))
    return [(name, length, int(round(n_total * length / tot))) for name, length in chroms]


def synth_genome(n_total, cfg, chroms=None):
    """Yield (name, X, Y) per chromosome; seed = 1000*cfg + chromosome index."""
    for ci, (name, length, n) in enumerate(chrom_sizes(n_total, chroms)):
        X, Y = synth_chrom(n, length, 1000 * cfg + ci)
        yield name, X, Y

"""Access to the committed golden vectors (tests/golden/, made by make_golden.py)."""
import json
import os

import numpy as np

import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_cache = {}


def _npz(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLD, name))
    return _cache[name]


def meta():
    with open(os.path.join(GOLD, "golden_meta.json")) as fh:
        return json.load(fh)


def chr21_xy():
    z = _npz("chr21_input.npz")
    return z["X"].astype(np.int64), z["Y"].astype(np.int64)


def chr21_labels(variant, eps, minPts):
    return _npz("chr21_labels.npz")["%s_%d_%d" % (variant, eps, minPts)]


def chr21_chain_labels(variant, eps):
    return _npz("chr21_chain_labels.npz")["%s_%d" % (variant, eps)]


def family_cases(family, ncase=60):
    """Yield (k, ids, X, Y, eps, minPts, {variant: golden labels})."""
    z = _npz("families.npz")
    for k in range(ncase):
        ids, X, Y = z["%s_%d_in" % (family, k)].astype(np.int64)
        eps, minPts = (int(v) for v in z["%s_%d_par" % (family, k)])
        yield k, ids, X, Y, eps, minPts, {v: z["%s_%d_%s" % (family, k, v)] for v in ("v1", "v2", "block")}


def synth150k():
    from cloops_amd.synth import synth_chrom
    X, Y = synth_chrom(150000, 46709983, 424242)
    z = _npz("synth150k_labels.npz")
    chk = z["XY_sha_check"]
    assert int(X.astype(np.int64).sum() % (1 << 31)) == int(chk[0]), "synthetic generator drifted"
    assert int(Y.astype(np.int64).sum() % (1 << 31)) == int(chk[1]), "synthetic generator drifted"
    return X.astype(np.int64), Y.astype(np.int64), z


def dense400k():
    """-> (X, Y, npz of label arrays, meta): the headline regime, labels made by the REAL classes (golden/make_golden_dense.py)"""
    import hashlib
    from cloops_amd.synth import synth_chrom
    with open(os.path.join(GOLD, "dense400k_meta.json")) as fh:
        m = json.load(fh)
    X, Y = synth_chrom(m["n"], m["length"], m["seed"])
    assert hashlib.sha1(np.stack([X, Y]).astype(np.int32).tobytes()).hexdigest() == m["input_sha1"], "synthetic generator drifted"
    return X.astype(np.int64), Y.astype(np.int64), _npz("dense400k_labels.npz"), m


DENSE_SETTINGS = ((5000, 50, 0), (7500, 30, 5004), (10000, 20, 6250))

FAMILY_SEEDS = {"adversarial": 0, "plain": 1, "clumpy": 2}


def regenerate_family(family, ncase=60):
    rng = np.random.default_rng(FAMILY_SEEDS[family])
    gen = getattr(cases, family + "_case")
    return [gen(rng, k) for k in range(ncase)]


_MIX = np.array([0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0x27D4EB2F165667C5], dtype=np.uint64)


def box_checksum(boxes):
    """order-independent checksum of int rows (minX, maxX, minY, maxY): sum over rows of a 64-bit mix, mod 2^64"""
    b = np.asarray(boxes, dtype=np.int64).reshape(-1, 4).astype(np.uint64)
    with np.errstate(over="ignore"):
        h = (b * _MIX[None, :]).sum(axis=1, dtype=np.uint64)
        h ^= h >> np.uint64(29)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(32)
        return int(h.sum(dtype=np.uint64))

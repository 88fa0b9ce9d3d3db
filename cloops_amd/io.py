"""Input side of the hot path: BEDPE -> per-chromosome PET arrays -> `.jd` / HBM.

Restates cLoops/io.py:30-59 (class PET), :62-129 (parseRawBedpe, the `eps = 0` input path), :132-189
(parseRawBedpe2), :192-203 (txt2jd) and :206-217 (parseJd) with the reference's PYTHON-2 semantics:

  * cis PETs only (chromA == chromB, io.py:168), optional chromosome filter (:171);
  * lines holding both a "*" and a "-1" field, or fewer than 6 fields, or non-integer
    coordinates are skipped (:159-166);
  * the two ends are swapped so that the left mid-point is the smaller one (:50-53);
  * mid-points are FLOOR((start + end) / 2) -- `/` on Python-2 ints (:55-56);
  * `cut` > 0 drops PETs with distance < cut (:174-175);
  * the point id is the per-chromosome row counter (:181-183).

`parse_bedpe` is the array-level loader (no temporary text files); `parseRawBedpe2` / `txt2jd`
keep the reference's file-based protocol (`<fout>/<chr>-<chr>.jd` = joblib-pickled int64 [n,3]).
"""
import gzip
import os

import numpy as np


def _open(f):
    return gzip.open(f, "rt") if f.endswith(".gz") else open(f)


def parse_bedpe(fs, cs=(), cut=0, unique=False, strand_distances=None):
    """-> (OrderedDict-like dict chrom -> int64 [n,3] rows [id, X, Y] in file order, n_lines, n_cis).

    `unique` / `strand_distances` are the two extras of the reference's `eps = 0` parser (cLoops/io.py:62-129):
    a PET whose (cA, cB) was already seen on its chromosome is dropped (:113-114), and the distances of the kept
    PETs whose two ends map to different strands are appended to the list `strand_distances` (:122-123).  That
    parser also tests only chromA against the wanted chromosomes (:98; the PETs are cis anyway)."""
    cs = set(cs) if cs else set()
    xs, ys, seen = {}, {}, {}
    i = j = 0
    for f in fs:
        with _open(f) as fh:
            for line in fh:
                i += 1
                line = line.split("\n")[0].split("\t")
                if "*" in line and "-1" in line:
                    continue
                if len(line) < 6:
                    continue
                try:
                    chromA, startA, endA = line[0], int(line[1]), int(line[2])
                    chromB, startB, endB = line[3], int(line[4]), int(line[5])
                    line[8], line[9]                      # PET.__init__ reads the strand columns (io.py:43,47)
                except (ValueError, IndexError):
                    continue
                if chromA != chromB:
                    continue
                if cs and not (chromA in cs and chromB in cs):
                    continue
                if startA + endA > startB + endB:
                    startA, startB = startB, startA
                    endA, endB = endB, endA
                cA = (startA + endA) // 2
                cB = (startB + endB) // 2
                if cut > 0 and cB - cA < cut:
                    continue
                if chromA not in xs:
                    xs[chromA], ys[chromA], seen[chromA] = [], [], set()
                if unique:
                    if (cA, cB) in seen[chromA]:
                        continue
                    seen[chromA].add((cA, cB))
                if strand_distances is not None and line[8] != line[9]:
                    strand_distances.append(cB - cA)
                xs[chromA].append(cA)
                ys[chromA].append(cB)
                j += 1
    out = {}
    for c in xs:
        n = len(xs[c])
        m = np.empty((n, 3), dtype=np.int64)
        m[:, 0] = np.arange(n)
        m[:, 1] = xs[c]
        m[:, 2] = ys[c]
        out[c] = m
    return out, i, j


def parseRawBedpe(fs, fout, cs, cut, logger=None):
    """cLoops/io.py:62-129, the parser of the `eps = 0` path (pipe.py:231-232): like parseRawBedpe2 but duplicates
    (same two mid-points on a chromosome) are removed and the distances of the PETs mapped to different strands come
    back as the second value -- `ests.estFragSize` turns them into eps (pipe.py:237-239)."""
    import joblib
    for f in fs:
        if logger is not None:
            logger.info("Parsing PETs from %s, requiring initial distance cutoff > %s" % (f, cut))
    ds = []
    mats, i, j = parse_bedpe(fs, cs, cut, unique=True, strand_distances=ds)
    cfs = []
    for c, m in mats.items():
        cf = os.path.join(fout, "%s-%s" % (c, c) + ".jd")
        joblib.dump(m, cf)
        cfs.append(cf)
    if logger is not None:
        logger.info("Totaly %s PETs from %s, in which %s cis PETs" % (i, ",".join(fs), j))
    return cfs, ds


def parseRawBedpe2(fs, fout, cs, cut, logger=None):
    """cLoops/io.py:132-189: returns the list of per-chromosome files (here `.jd` directly: the
    reference writes `.txt` and converts them with txt2jd, io.py:192-203; `txt2jd` below accepts
    both, so `cfs = [txt2jd(f) for f in parseRawBedpe2(...)]` works as in pipe.py:234-235)."""
    import joblib
    for f in fs:
        if logger is not None:
            logger.info("Parsing PETs from %s, requiring initial distance cutoff > %s" % (f, cut))
    mats, i, j = parse_bedpe(fs, cs, cut)
    cfs = []
    for c, m in mats.items():
        cf = os.path.join(fout, "%s-%s" % (c, c) + ".jd")
        joblib.dump(m, cf)
        cfs.append(cf)
    if logger is not None:
        logger.info("Totaly %s PETs from %s, in which %s cis PETs" % (i, ",".join(fs), j))
    return cfs


def txt2jd(f):
    """cLoops/io.py:192-203.  A `.jd` path is returned unchanged."""
    import joblib
    if f.endswith(".jd"):
        return f
    data = []
    for line in open(f):
        line = line.split("\n")[0].split("\t")
        data.append(list(map(int, line)))
    data = np.array(data)
    out = f.replace(".txt", ".jd")
    joblib.dump(data, out)
    os.remove(f)
    return out


def parseJd(f, cut=0):
    """cLoops/io.py:206-217."""
    from .pipe import parseJd as _p
    return _p(f, cut)

"""CPU: BEDPE input side (cLoops/io.py:30-59,132-217 restated in cloops_amd/io.py)."""
import gzip
import os

import numpy as np
import pytest

import golden_util as G
import refload
from cloops_amd import io as cio

BEDPE = os.path.join(refload.REF_ROOT, "examples", "GSM1872886_GM12878_CTCF_ChIA-PET_chr21_hg38.bedpe.gz")

LINES = [
    "chr1\t100\t201\tchr1\t5000\t5100\tid1\t1\t+\t-",        # mid 150 (floor of 150.5), 5050
    "chr1\t9000\t9100\tchr1\t300\t401\tid2\t1\t+\t-",        # ends swapped: left = (300,401) -> 350
    "chr1\t10\t20\tchr2\t30\t40\tid3\t1\t+\t-",              # trans: dropped
    "chr1\t*\t-1\tchr1\t30\t40\tid4\t1\t+\t-",               # '*' and '-1': dropped
    "chr1\t10\t20\tchr1",                                    # < 6 fields: dropped
    "chr1\tx\t20\tchr1\t30\t40\tid5\t1\t+\t-",               # non-integer: dropped
    "chr2\t7\t8\tchr2\t7\t9\tid6\t1\t-\t-",                  # 7 , 8
    "chr1\t1000\t1001\tchr1\t1100\t1101\tid7\t1\t+\t+",      # distance 100
]


def test_pet_rule_and_filters(tmp_path):
    f = tmp_path / "t.bedpe"
    f.write_text("\n".join(LINES) + "\n")
    mats, n_lines, n_cis = cio.parse_bedpe([str(f)])
    assert n_lines == 8 and n_cis == 4
    assert list(mats.keys()) == ["chr1", "chr2"]
    assert mats["chr1"].tolist() == [[0, 150, 5050], [1, 350, 9050], [2, 1000, 1100]]
    assert mats["chr2"].tolist() == [[0, 7, 8]]
    mats, _, _ = cio.parse_bedpe([str(f)], cs=["chr1"], cut=200)
    assert list(mats.keys()) == ["chr1"] and mats["chr1"].tolist() == [[0, 150, 5050], [1, 350, 9050]]


def test_jd_round_trip(tmp_path):
    f = tmp_path / "t.bedpe.gz"
    with gzip.open(str(f), "wt") as fh:
        fh.write("\n".join(LINES) + "\n")
    cfs = [cio.txt2jd(c) for c in cio.parseRawBedpe2([str(f)], str(tmp_path), [], 0)]
    assert sorted(os.path.basename(c) for c in cfs) == ["chr1-chr1.jd", "chr2-chr2.jd"]
    key, mat = cio.parseJd([c for c in cfs if "chr1" in c][0])
    assert key == ("chr1", "chr1") and mat.dtype == np.int64 and mat.shape == (3, 3)
    key, mat = cio.parseJd([c for c in cfs if "chr1" in c][0], cut=200)
    assert mat.shape == (2, 3)


@pytest.mark.skipif(not os.path.exists(BEDPE), reason="reference example data not present")
def test_example_bedpe_matches_golden_input():
    mats, n_lines, n_cis = cio.parse_bedpe([BEDPE])
    X, Y = G.chr21_xy()
    assert list(mats.keys()) == ["chr21"] and n_cis == 99674
    assert np.array_equal(mats["chr21"][:, 1], X) and np.array_equal(mats["chr21"][:, 2], Y)
    m = G.meta()["chr21"]
    assert (int(X.min()), int(X.max()), int(Y.min()), int(Y.max())) == (m["xmin"], m["xmax"], m["ymin"], m["ymax"])


@pytest.mark.skipif(not refload.available(), reason="reference checkout not present")
def test_against_reference_parser_with_py2_semantics(tmp_path):
    """The reference's own PET class + parseRawBedpe2 (sliced out of the py2-only io.py, with the
    two py2->py3 semantic patches `/ 2` -> `// 2` and "rb" -> "rt") on the example file."""
    with open(os.path.join(refload.REF_ROOT, "cLoops", "io.py")) as fh:
        lines = fh.read().split("\n")

    def block(start_pat):
        s = [i for i, l in enumerate(lines) if l.startswith(start_pat)][0]
        e = [i for i, l in enumerate(lines) if i > s and (l.startswith("def ") or l.startswith("class "))][0]
        return "\n".join(lines[s:e])
    src = block("class PET") + "\n" + block("def parseRawBedpe2")
    src = src.replace(") / 2", ") // 2").replace('"rb"', '"rt"')

    class L(object):
        def info(self, *a):
            pass
    ns = {"gzip": gzip, "os": os, "cFlush": lambda *a: None}
    exec(compile(src, "io.py:slice", "exec"), ns)
    out = tmp_path / "ref"
    out.mkdir()
    cfs = ns["parseRawBedpe2"]([BEDPE], str(out), [], 0, L())
    assert len(cfs) == 1
    ref = np.loadtxt(cfs[0], dtype=np.int64)
    mats, _, _ = cio.parse_bedpe([BEDPE])
    assert np.array_equal(ref, mats["chr21"])


@pytest.mark.skipif(not refload.available(), reason="reference checkout not present")
def test_auto_eps_path_against_the_reference(tmp_path):
    """`eps = 0` (cLoops/pipe.py:231-239): the reference's parseRawBedpe (duplicate removal, distances of the PETs mapped to
    different strands; sliced out of the py2-only io.py with the same two py2 -> py3 patches as above) and its own
    estFragSize (ests.py:23-33) on the example file, against cloops_amd.io.parseRawBedpe / cloops_amd.ests.estFragSize."""
    import joblib
    from cloops_amd import ests
    with open(os.path.join(refload.REF_ROOT, "cLoops", "io.py")) as fh:
        lines = fh.read().split("\n")

    def block(start_pat):
        s = [i for i, l in enumerate(lines) if l.startswith(start_pat)][0]
        e = [i for i, l in enumerate(lines) if i > s and (l.startswith("def ") or l.startswith("class "))][0]
        return "\n".join(lines[s:e])
    src = block("class PET") + "\n" + block("def parseRawBedpe(")
    src = src.replace(") / 2", ") // 2").replace('"rb"', '"rt"')

    class L(object):
        def info(self, *a):
            pass
    ns = {"gzip": gzip, "os": os, "cFlush": lambda *a: None}
    exec(compile(src, "io.py:slice", "exec"), ns)
    # the example file, and a file with duplicated PETs (the example holds none): the first 3000 lines followed by the
    # first 1000 again
    dup = str(tmp_path / "dup.bedpe")
    with gzip.open(BEDPE, "rt") as fh:
        head = [next(fh) for _ in range(3000)]
    with open(dup, "w") as fh:
        fh.writelines(head + head[:1000])
    for k, f in enumerate((BEDPE, dup)):
        out = tmp_path / ("ref%d" % k)
        out.mkdir()
        cfs, ds = ns["parseRawBedpe"]([f], str(out), [], 0, L())
        ref = np.loadtxt(cfs[0], dtype=np.int64)
        mine = tmp_path / ("mine%d" % k)
        mine.mkdir()
        cfs2, ds2 = cio.parseRawBedpe([f], str(mine), [], 0)
        assert len(cfs) == len(cfs2) == 1
        assert np.array_equal(ref, joblib.load(cfs2[0]))
        assert ds == ds2 and len(ds) > 100 and len(ds) < len(ref)
        want = refload.ref_ests().estFragSize(ds)
        assert ests.estFragSize(ds2) == want and want > 0
        if k == 1:
            mats, _, _ = cio.parse_bedpe([f])
            assert len(ref) < len(mats["chr21"])            # the duplicate filter really dropped PETs

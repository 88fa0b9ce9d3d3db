"""Loader for the *real* reference classes (YaqiangCao/cLoops at /root/reference).

Test infrastructure only.  The reference is Python-2 code that exists only in the
build container (never on the GPU box), so everything here degrades to "absent"
when /root/reference is missing and the tests that need it skip.

Nothing of the reference is copied into this repository: cDBSCAN.py and
blockDBSCAN.py are imported in place; cDBSCAN2.py needs the mechanical
`.iteritems()` -> `.items()` substitution to run on Python 3, which is applied
to the source text *in memory* and exec'd into a throw-away module object
(SURVEY.md Appendix A).
"""
import os
import sys
import types

REF_ROOT = os.environ.get("CLOOPS_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF_ROOT, "cLoops", "cDBSCAN2.py"))


_cache = {}


def _load_text_module(name, relpath, subst=()):
    with open(os.path.join(REF_ROOT, relpath)) as fh:
        src = fh.read()
    for a, b in subst:
        src = src.replace(a, b)
    mod = types.ModuleType(name)
    exec(compile(src, relpath, "exec"), mod.__dict__)
    return mod


def ref_classes():
    """-> dict(v1=class, v2=class, block=class) of the reference's own classes."""
    if "cls" in _cache:
        return _cache["cls"]
    if not available():
        raise RuntimeError("reference not present at %s" % REF_ROOT)
    v1 = _load_text_module("_ref_cDBSCAN", "cLoops/cDBSCAN.py")
    v2 = _load_text_module("_ref_cDBSCAN2", "cLoops/cDBSCAN2.py",
                           subst=((".iteritems()", ".items()"),))
    bl = _load_text_module("_ref_blockDBSCAN", "cLoops/blockDBSCAN.py")
    _cache["cls"] = {"v1": v1.cDBSCAN, "v2": v2.cDBSCAN, "block": bl.blockDBSCAN}
    return _cache["cls"]


def ref_ests():
    """The reference's cut estimator (cLoops/ests.py:36-61), imported in place."""
    if "ests" in _cache:
        return _cache["ests"]
    src_path = os.path.join(REF_ROOT, "cLoops", "ests.py")
    with open(src_path) as fh:
        src = fh.read()
    # ests.py does `from .utils import cFlush` (unused on this path); neutralise the
    # relative import so the file can be exec'd standalone.
    src = src.replace("from .utils import cFlush", "cFlush = None")
    mod = types.ModuleType("_ref_ests")
    exec(compile(src, src_path, "exec"), mod.__dict__)
    _cache["ests"] = mod
    return mod


def ref_labels(variant, mat, eps, minPts):
    """Run the real reference class; return its `.labels` dict (ids -> cluster id)."""
    import numpy as np
    cls = ref_classes()[variant]
    db = cls(np.asarray(mat), eps, minPts)
    return db.labels


def labels_dict_to_array(labels, ids):
    """dict {id: cid} -> int32 array aligned with `ids` (noise = -1)."""
    import numpy as np
    out = np.full(len(ids), -1, dtype=np.int32)
    pos = {int(k): i for i, k in enumerate(ids)}
    for k, v in labels.items():
        out[pos[int(k)]] = int(v)
    return out


def ref_pipe_namespace(variant="v2"):
    """The reference's OWN dispatch functions (cLoops/pipe.py: singleDBSCAN :52-110,
    runDBSCAN :113-127, filterClusterByDis :130-143, combineTwice :155-174), extracted from
    the parsed module (pipe.py cannot be imported: it pulls seaborn and the py2-only io.py)
    and exec'd in a namespace wired to the real clustering class, the real parseJd
    (cLoops/io.py:206-217, sliced out of the py2-only file) and the real estIntSelCutFrag."""
    import ast
    import sys as _sys
    import numpy as np
    import pandas as pd
    import joblib
    from joblib import Parallel, delayed
    key = "pipe_" + variant
    if key in _cache:
        return _cache[key]
    ns = {"np": np, "pd": pd, "sys": _sys, "os": os, "joblib": joblib, "Parallel": Parallel, "delayed": delayed,
          "DBSCAN": ref_classes()[variant], "estIntSelCutFrag": ref_ests().estIntSelCutFrag}
    # parseJd: slice `def parseJd` ... up to the next top-level def out of io.py
    with open(os.path.join(REF_ROOT, "cLoops", "io.py")) as fh:
        lines = fh.read().split("\n")
    start = [i for i, l in enumerate(lines) if l.startswith("def parseJd")][0]
    end = [i for i, l in enumerate(lines) if i > start and l.startswith("def ")][0]
    exec(compile("\n".join(lines[start:end]), "io.py:parseJd", "exec"), ns)
    with open(os.path.join(REF_ROOT, "cLoops", "pipe.py")) as fh:
        tree = ast.parse(fh.read())
    want = {"singleDBSCAN", "runDBSCAN", "filterClusterByDis", "checkSameLoop", "combineTwice"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    mod = ast.Module(body=body, type_ignores=[])
    exec(compile(mod, "pipe.py:dispatch", "exec"), ns)
    _cache[key] = ns
    return ns


def ref_cmodel_namespace():
    """The reference's significance module (cLoops/cModel.py) exec'd in memory with the mechanical
    py2 -> py3 patches the survey lists (section 8f-3): xrange -> range, dict.keys() indexing ->
    list, floor `/` on ints in getNearbyPairRegions (cModel.py:89-93); its io imports (parseJd,
    parseIv) are sliced out of the py2-only cLoops/io.py.  Used to MAKE golden vectors only."""
    import gc
    import joblib
    import numpy as np
    import pandas as pd
    if "cmodel" in _cache:
        return _cache["cmodel"]
    with open(os.path.join(REF_ROOT, "cLoops", "io.py")) as fh:
        lines = fh.read().split("\n")

    def block(start_pat):
        s = [i for i, l in enumerate(lines) if l.startswith(start_pat)][0]
        e = [i for i, l in enumerate(lines) if i > s and (l.startswith("def ") or l.startswith("class "))]
        return "\n".join(lines[s:(e[0] if e else len(lines))])
    ns = {"np": np, "pd": pd, "os": os, "joblib": joblib, "gc": gc, "cFlush": lambda *a: None}
    exec(compile(block("def parseJd") + "\n" + block("def parseIv"), "io.py:slice", "exec"), ns)
    with open(os.path.join(REF_ROOT, "cLoops", "cModel.py")) as fh:
        src = fh.read()
    src = src.replace("from cLoops.io import parseJd, parseIv", "").replace("from cLoops.utils import cFlush", "")
    src = src.replace("xrange", "range")
    src = src.replace("keys = ds.keys()", "keys = list(ds.keys())")
    # Python-2 integer division in getNearbyPairRegions (cModel.py:89-93)
    src = src.replace("ca = sum(iva) / 2", "ca = sum(iva) // 2").replace("cb = sum(ivb) / 2", "cb = sum(ivb) // 2")
    src = src.replace("sa = (iva[1] - iva[0]) / 2", "sa = (iva[1] - iva[0]) // 2").replace("sb = (ivb[1] - ivb[0]) / 2", "sb = (ivb[1] - ivb[0]) // 2")
    src = src.replace("step = (sa + sb) / 2", "step = (sa + sb) // 2")
    exec(compile(src, "cModel.py:py3", "exec"), ns)
    _cache["cmodel"] = ns
    return ns


def ref_stripes_namespace():
    """The stripe functions of scripts/callStripes (singleStripDBSCAN :37-72, filterCandidateStripes :75-86,
    getNearbyStripes :89-120, getStripePsFdr :123-185, estStripeSig :188-233, markStripeSig :236-269),
    extracted from the parsed script (it cannot be imported: py2-only cLoops.io / cLoops.utils) and exec'd in a
    namespace wired to the real variant-1 class, the real parseJd and the converted reference cModel
    (ref_cmodel_namespace).  Mechanical py2 -> py3 patches: xrange -> range, integer `/` -> `//` in
    getNearbyStripes and filterCandidateStripes.  Used to MAKE golden vectors only."""
    import ast
    import numpy as np
    import pandas as pd
    import gc
    from scipy.stats import hypergeom, binom, poisson, combine_pvalues
    if "stripes" in _cache:
        return _cache["stripes"]
    cm = ref_cmodel_namespace()
    ns = {"np": np, "pd": pd, "gc": gc, "os": os, "hypergeom": hypergeom, "binom": binom, "poisson": poisson,
          "combine_pvalues": combine_pvalues, "DBSCAN": ref_classes()["v1"], "parseJd": cm["parseJd"],
          "getGenomeCoverage": cm["getGenomeCoverage"], "getPETsforRegions": cm["getPETsforRegions"],
          "getCounts": cm["getCounts"], "cFlush": lambda *a: None}
    with open(os.path.join(REF_ROOT, "scripts", "callStripes")) as fh:
        src = fh.read()
    src = src.replace("xrange", "range")
    for a, b in (("ca = sum(iva) / 2", "ca = sum(iva) // 2"), ("cb = sum(ivb) / 2", "cb = sum(ivb) // 2"),
                 ("sa = (iva[1] - iva[0]) / 2", "sa = (iva[1] - iva[0]) // 2"), ("sb = (ivb[1] - ivb[0]) / 2", "sb = (ivb[1] - ivb[0]) // 2"),
                 ("if (xlen / ylen > lengthFoldDiff) or (ylen / xlen >", "if (xlen // ylen > lengthFoldDiff) or (ylen // xlen >")):
        assert a in src, a
        src = src.replace(a, b)
    tree = ast.parse(src)
    want = {"singleStripDBSCAN", "filterCandidateStripes", "getNearbyStripes", "getStripePsFdr", "estStripeSig", "markStripeSig"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert len(body) == len(want)
    exec(compile(ast.Module(body=body, type_ignores=[]), "callStripes:functions", "exec"), ns)
    _cache["stripes"] = ns
    return ns


def ref_saturation_namespace():
    """The functions of scripts/jd2saturation that can be pinned (generateSamplingData :32-55, singleDBSCAN :56-108,
    runDBSCAN :111-127, getLoops :154-178), extracted from the parsed script -- it cannot be imported: `from cLoops.pipe import
    checkOverlap` (:25) names a function cLoops/pipe.py does not have -- and exec'd in a namespace wired to the REAL
    variant-1 class, the real parseJd / filterClusterByDis / combineTwice / estIntSelCutFrag; `runStat` is replaced by a
    recorder (the significance step has its own goldens).  Mechanical py2 -> py3 patch: xrange -> range.  Used to MAKE golden
    vectors only."""
    import ast
    import numpy as np
    import pandas as pd
    import joblib
    if "saturation" in _cache:
        return _cache["saturation"]
    pn = ref_pipe_namespace("v1")

    class _Log(object):
        def info(self, *a):
            pass
        warning = error = info
    ns = {"np": np, "pd": pd, "os": os, "joblib": joblib, "logger": _Log(), "DBSCAN": ref_classes()["v1"], "parseJd": pn["parseJd"],
          "estIntSelCutFrag": ref_ests().estIntSelCutFrag, "filterClusterByDis": pn["filterClusterByDis"], "combineTwice": pn["combineTwice"],
          "recorded": []}
    ns["runStat"] = lambda dataI, minPts, cut, fout, hic=0: ns["recorded"].append((dataI, minPts, cut, fout)) or 0
    with open(os.path.join(REF_ROOT, "scripts", "jd2saturation")) as fh:
        src = fh.read().replace("xrange", "range")
    tree = ast.parse(src)
    want = {"generateSamplingData", "singleDBSCAN", "runDBSCAN", "getLoops"}
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want]
    assert len(body) == len(want)
    exec(compile(ast.Module(body=body, type_ignores=[]), "jd2saturation:functions", "exec"), ns)
    _cache["saturation"] = ns
    return ns

"""CPU stand-in for cloops_amd.api.Chromosome used ONLY by the `not gpu` tests of the host
logic (pipe.py, wrappers): same interface, results from the CPU oracle.  Installed by
monkeypatching in the tests; the product never sees it."""
import numpy as np

import oracle
from cloops_amd import api


class FakeChromosome(object):
    def __init__(self, X, Y, device=0, stream=None):
        self.X = np.asarray(X, np.int64)
        self.Y = np.asarray(Y, np.int64)
        self.n = len(self.X)
        self.device = device

    def close(self):
        pass

    def subsample(self, rows):
        rows = np.asarray(rows, np.int64)
        return FakeChromosome(self.X[rows], self.Y[rows], self.device)

    def set_profiling(self, on=True):
        pass

    def set_layout_reuse(self, on=True):
        pass

    def set_device_labels(self, on=True):
        pass

    def set_count_reuse(self, on=True):
        pass

    def set_count_floor(self, min_pts):
        pass

    def set_count_thresholds(self, min_pts_list):
        pass

    def set_eps_list(self, eps_list):
        pass

    def sweep_plan(self, eps_list, min_pts_list):
        pass

    def set_stream(self, stream):
        pass

    def set_sort_index(self, mode=1):
        pass

    def cluster(self, variant, eps, minPts, cut=0, want_labels=True, want_boxes=True, pinned=False):
        vname = {1: "v1", 2: "v2", 3: "block"}[api.VARIANTS[variant]]
        lab = oracle.single_dbscan(vname, self.X, self.Y, eps, minPts, cut)["labels"]
        ml = int(lab.max()) if len(lab) else -1
        boxes = np.zeros(ml + 1, api.BOX_DTYPE)
        for c in np.unique(lab[lab >= 0]):
            m = lab == c
            boxes[c] = (self.X[m].min(), self.X[m].max(), self.Y[m].min(), self.Y[m].max(), m.sum())
        return api.ClusterResult(lab, int(len(np.unique(lab[lab >= 0]))), ml, boxes, None)

    # ---- the asynchronous / statistics surface used by pipe.runSweepFast -------------------------
    def cluster_async(self, variant, eps, minPts, cut=0, want_labels=True, want_boxes=True):
        self._pending = getattr(self, "_pending", [])
        self._pending.append((variant, eps, minPts, cut))

    def wait(self, copy=False):
        variant, eps, minPts, cut = self._pending.pop(0)
        res = self.cluster(variant, eps, minPts, cut)
        self._last = (res, cut)
        return res

    # ---- candidate loops of a sweep (kernels K10) -------------------------------------------------
    def cand_reset(self):
        self._cand = []

    def cand_append(self, step):
        res, _ = self._last
        b = res.boxes
        K = len(b)
        if K == 0:
            return 0, 0
        t = np.stack([b["min_x"], b["max_x"], b["min_y"], b["max_y"]], 1).astype(np.int64)
        ok = (b["count"] > 0) & (t[:, 0] != t[:, 1]) & (t[:, 2] != t[:, 3])
        inter = ok & (t[:, 1] < t[:, 2])
        if inter.any():
            self._cand.append((step, t[inter]))
        return int(inter.sum()), int((ok & ~inter).sum())

    def step_async(self, variant, eps, minPts, cut, step, fine_lo=-1):
        self.cluster_async(variant, eps, minPts, cut)
        self._step = (step, cut, fine_lo)

    def step_result(self):
        step, cut, fine_lo = self._step
        ni, ns = self.cand_append(step)
        s = self.dist_summary(cut)
        s["fine_lo"], s["fine"] = fine_lo, None
        if fine_lo >= 0:
            g, ad = self._groups(cut)
            a = ad[(g == 1) & (ad >= fine_lo) & (ad < fine_lo + 2048)].astype(np.int64)
            s["fine"] = np.bincount(a - fine_lo, minlength=2048).astype(np.int64)
        return ni, ns, s

    def cand_finish(self, final_cut, capacity):
        from cloops_amd import pipe
        by_step = {}
        for step, rows in self._cand:
            by_step.setdefault(step, []).append(rows)
        b = pipe._combine_steps([np.concatenate(v) for _, v in sorted(by_step.items())]).astype(np.int64)
        dmid = (b[:, 2] + b[:, 3]) // 2 - (b[:, 0] + b[:, 1]) // 2
        return b[dmid >= final_cut].astype(np.int32)

    def cand_finish_device(self, final_cut):
        """the device form on the stand-in: (token, rows) -- the 'pointer' is only valid while the handle is alive"""
        self._dev_table = self.cand_finish(final_cut, 0)
        return id(self._dev_table), len(self._dev_table)

    def set_table_export(self, on=True):
        pass

    def last_n_in(self):
        res, cut = self._last
        d = self.Y - self.X
        return int((d >= cut).sum()) if cut > 0 else self.n

    def _groups(self, cut):
        res, _ = self._last
        lab = res.labels
        b = res.boxes
        K = len(b)
        cls = np.full(K + 1, -1, np.int64)
        if K:
            ok = (b["count"] > 0) & (b["min_x"] != b["max_x"]) & (b["min_y"] != b["max_y"])
            inter = ok & (b["max_x"] < b["min_y"])
            cls[:K][inter] = 0
            cls[:K][ok & ~inter] = 1
        d = self.Y - self.X
        g = cls[np.where(lab >= 0, lab, K)]
        if cut > 0:
            g = np.where(d < cut, 1, g)
        return g, np.abs(d)

    def dist_summary(self, cut=0):
        from cloops_amd import ests
        g, ad = self._groups(cut)
        xshift = 11.0
        out = {"n_all": [], "n_pos": [], "sumx": [], "sumxx": [], "xshift": xshift, "loghist": np.zeros(3840, np.int64)}
        for k in (0, 1):
            a = ad[g == k]
            out["n_all"].append(int(len(a)))
            a = a[a > 0]
            out["n_pos"].append(int(len(a)))
            x = np.log2(a.astype(np.float64)) - xshift
            out["sumx"].append(float(x.sum()))
            out["sumxx"].append(float((x * x).sum()))
            if k == 1 and len(a):
                a = a.astype(np.int64)
                e = np.floor(np.log2(a.astype(np.float64))).astype(np.int64)
                e = np.where((np.int64(1) << e) > a, e - 1, e)                  # guard against a rounded-up log2
                e = np.where((np.int64(1) << (e + 1)) <= a, e + 1, e)
                m = np.where(e >= 7, a >> np.maximum(e - 7, 0), a << np.maximum(7 - e, 0)) & 127
                out["loghist"] = np.bincount(e * 128 + m, minlength=3840).astype(np.int64)
                assert ests.logbin(int(a[0])) == int(e[0] * 128 + m[0])
        return out

    def dist_bin_hist(self, cut, lo, hi, shift):
        g, ad = self._groups(cut)
        a = ad[(g == 1) & (ad >= lo) & (ad < hi)].astype(np.int64)
        return np.bincount((a - lo) >> shift, minlength=2048).astype(np.int64)

    def sig_counts(self, windows, cut=0):
        """numpy stand-in of kernel K8 on explicit index sets (sorted arrays of row positions)"""
        X, Y = self.X, self.Y
        if cut > 0:
            keep = (Y - X) >= cut
            X, Y = X[keep], Y[keep]
        m = IndexSets(X, Y)
        w = np.asarray(windows).reshape(-1, 44)
        out = np.zeros((len(w), 144), np.int32)
        for n in range(len(w)):
            A = [m.region(w[n, k], w[n, 22 + k]) for k in range(11)]
            B = [m.region(w[n, 11 + k], w[n, 33 + k]) for k in range(11)]
            out[n, 0:11] = [len(a) for a in A]
            out[n, 11:22] = [len(b) for b in B]
            out[n, 22] = len(np.intersect1d(m.side(w[n, 0], w[n, 22], 0), m.side(w[n, 11], w[n, 33], 1), assume_unique=True))
            for k in range(11):
                for l in range(11):
                    out[n, 23 + 11 * k + l] = len(np.intersect1d(A[k], B[l], assume_unique=True))
        return out, len(X)


class IndexSets(object):
    """Row-position sets of the PETs with one end inside an interval (both ends inclusive) -- what kernel K8
    counts without materialising; test-side model of the reference's coverage dictionaries."""

    def __init__(self, X, Y):
        self.order = [np.argsort(X, kind="stable"), np.argsort(Y, kind="stable")]
        self.keys = [np.asarray(X)[self.order[0]], np.asarray(Y)[self.order[1]]]

    def side(self, lo, hi, axis):
        l = np.searchsorted(self.keys[axis], lo, side="left")
        r = np.searchsorted(self.keys[axis], hi, side="right")
        return np.sort(self.order[axis][l:r])

    def region(self, lo, hi):
        return np.union1d(self.side(lo, hi, 0), self.side(lo, hi, 1))

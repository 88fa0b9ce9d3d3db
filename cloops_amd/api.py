"""Array-level host API over the C ABI: a chromosome resident in HBM and clustering runs on it.

This is what the reference-shaped wrappers (cDBSCAN.py, cDBSCAN2.py, blockDBSCAN.py,
pipe.py) and bench.py are built on.
"""
import ctypes
import os

import numpy as np

from . import _lib

VARIANTS = {"v1": _lib.VARIANT_CDBSCAN1, "cDBSCAN": _lib.VARIANT_CDBSCAN1, 1: 1,
            "v2": _lib.VARIANT_CDBSCAN2, "cDBSCAN2": _lib.VARIANT_CDBSCAN2, 2: 2,
            "block": _lib.VARIANT_BLOCK, "blockDBSCAN": _lib.VARIANT_BLOCK, 3: 3}

BOX_DTYPE = np.dtype([("min_x", "<i4"), ("max_x", "<i4"), ("min_y", "<i4"), ("max_y", "<i4"), ("count", "<i4")])


# developer / test knob: every new handle is switched to this traversal level (cl_set_traversal; None = the library's default).
# The library itself reads no environment variable.
TRAVERSAL_OVERRIDE = int(os.environ["CLOOPS_TRAVERSAL"]) if os.environ.get("CLOOPS_TRAVERSAL", "") != "" else None


def device_count():
    return _lib.load().cl_device_count()


def _as_i32(a, name):
    a = np.asarray(a)
    if a.ndim != 1:
        raise ValueError("%s must be one-dimensional" % name)
    if a.dtype.kind == "f":
        if a.size and not np.all(a == np.floor(a)):
            raise TypeError("%s: integer coordinates required (the reference's PET mid-points are ints)" % name)
    elif a.dtype.kind not in "iu":
        raise TypeError("%s: integer coordinates required" % name)
    if a.size and (a.min() <= -(1 << 29) or a.max() >= (1 << 29)):
        raise _lib.CloopsHipError(_lib.CL_ERR_DOMAIN, "coordinates must satisfy |X|,|Y| < 2^29")
    return np.ascontiguousarray(a, dtype=np.int32)


class ClusterResult(object):
    """labels: int32[n] aligned to the input rows (-1 = not in the reference's `.labels`);
    boxes: structured array indexed by cluster id (count == 0 marks an id gap of variant 1)."""
    __slots__ = ("labels", "n_clusters", "max_label", "boxes", "timing")

    def __init__(self, labels, n_clusters, max_label, boxes, timing):
        self.labels = labels
        self.n_clusters = n_clusters
        self.max_label = max_label
        self.boxes = boxes
        self.timing = timing


class Chromosome(object):
    """One chromosome's PET coordinates resident in HBM (cl_chrom of include/cloops_hip.h)."""

    def __init__(self, X, Y, device=0, stream=None):
        lib = _lib.load()
        self._lib = lib
        self._h = ctypes.c_void_p()
        X = _as_i32(X, "X")
        Y = _as_i32(Y, "Y")
        if X.shape != Y.shape:
            raise ValueError("X and Y differ in length")
        self.n = int(X.shape[0])
        self.device = device
        self._init_state()
        _lib.check(lib.cl_chrom_create(int(device), ctypes.c_void_p(stream), X.ctypes.data_as(ctypes.c_void_p),
                                       Y.ctypes.data_as(ctypes.c_void_p), self.n, 0, ctypes.byref(self._h)))
        self._after_create()

    def _after_create(self):
        if TRAVERSAL_OVERRIDE is not None:
            self._lib.cl_set_traversal(self._h, int(TRAVERSAL_OVERRIDE))

    def _init_state(self):
        self._profiling = False
        self._pinned = [None, None]
        self._pinned_arr = [None, None]
        self._inflight = []
        self._enq = 0

    @classmethod
    def from_device_pointers(cls, x_ptr, y_ptr, n, device=0, stream=None, keepalive=None):
        """Wrap int32 device arrays (e.g. torch tensors' data_ptr()) without copying."""
        self = cls.__new__(cls)
        lib = _lib.load()
        self._lib = lib
        self._h = ctypes.c_void_p()
        self.n = int(n)
        self.device = device
        self._keepalive = keepalive
        self._init_state()
        _lib.check(lib.cl_chrom_create(int(device), ctypes.c_void_p(stream), ctypes.c_void_p(x_ptr),
                                       ctypes.c_void_p(y_ptr), self.n, 1, ctypes.byref(self._h)))
        self._after_create()
        return self

    def subsample(self, rows):
        """a new resident chromosome made of `rows` of this one, in the order given (`mat[rows, :]` of
        scripts/jd2saturation:46-47), gathered on the device (cl_chrom_subsample)"""
        rows = np.ascontiguousarray(rows, dtype=np.int64)
        new = Chromosome.__new__(Chromosome)
        new._lib = self._lib
        new._h = ctypes.c_void_p()
        new.n = int(rows.shape[0])
        new.device = self.device
        new._init_state()
        _lib.check(self._lib.cl_chrom_subsample(self._h, rows.ctypes.data_as(ctypes.c_void_p), new.n, ctypes.byref(new._h)))
        new._after_create()
        return new

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.cl_chrom_destroy(self._h)
            self._h = ctypes.c_void_p()
        for k, p in enumerate(getattr(self, "_pinned", [])):
            if p:
                self._lib.cl_host_free(ctypes.c_void_p(p))
                self._pinned[k] = None
        for k, p in enumerate(getattr(self, "_pin_pairs", None) or []):
            if p:
                self._lib.cl_host_free(ctypes.c_void_p(p))
                self._pin_pairs[k] = None
        for k, p in enumerate(getattr(self, "_pin_mask", None) or []):
            if p:
                self._lib.cl_host_free(ctypes.c_void_p(p))
                self._pin_mask[k] = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _pinned_labels(self, which=0):
        """Page-locked int32[n] result buffers owned by this object (two: one per result slot)."""
        if self._pinned[which] is None:
            p = self._lib.cl_host_alloc(max(4, self.n * 4))
            if not p:
                raise MemoryError("cl_host_alloc failed")
            self._pinned[which] = p
            self._pinned_arr[which] = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_int32)), shape=(self.n,))
        return self._pinned_arr[which]

    def set_profiling(self, on=True):
        self._profiling = bool(on)
        self._lib.cl_set_profiling(self._h, 1 if on else 0)

    def set_device_labels(self, on=True):
        """row-aligned device labels for runs without a host destination (default on); the sweep driver switches
        them off: it only needs tables and distance statistics (cl_set_device_labels)"""
        self._lib.cl_set_device_labels(self._h, 1 if on else 0)

    def set_table_export(self, on=True):
        """copy of the cluster table to pinned host memory at the end of a run (default on; cl_set_table_export)"""
        if getattr(self, "_export", True) != bool(on):
            self._lib.cl_set_table_export(self._h, 1 if on else 0)
            self._export = bool(on)

    # ---- candidate loops of a sweep, kept on the device (K10) ----
    def cand_reset(self):
        _lib.check(self._lib.cl_cand_reset(self._h))

    def cand_append(self, step):
        """classify the last run's cluster table (pipe.py:83-97) and append its inter-ligation boxes under `step`
        -> (n_inter, n_self)"""
        ni, ns = ctypes.c_int64(0), ctypes.c_int64(0)
        _lib.check(self._lib.cl_cand_append(self._h, int(step), ctypes.byref(ni), ctypes.byref(ns)))
        return int(ni.value), int(ns.value)

    def step_async(self, variant, eps, minPts, cut, step, fine_lo=-1):
        """one sweep step in one asynchronous call (cl_cluster_step_async): the run, then -- in its own stream -- the
        classification of its table, the append of its inter-ligation boxes under `step` and the distance summary;
        pair with wait(), then read step_result()"""
        _lib.check(self._lib.cl_cluster_step_async(self._h, VARIANTS[variant], int(eps), int(minPts), int(cut), int(step), int(fine_lo)))
        self._export = False
        self._inflight.append((None, False))
        self._enq += 1

    def step_result(self):
        """-> (n_inter, n_self, summary dict as dist_summary) of the last completed sweep step (no GPU work)"""
        ni, ns = ctypes.c_int64(0), ctypes.c_int64(0)
        st = _lib.ClDsummary()
        _lib.check(self._lib.cl_step_result(self._h, ctypes.byref(ni), ctypes.byref(ns), ctypes.byref(st)))
        return int(ni.value), int(ns.value), self._summary_dict(st)

    def cand_finish(self, final_cut, capacity):
        """combineTwice + filterClusterByDis over everything appended since cand_reset -> int32 [k, 4] boxes
        (minX, maxX, minY, maxY) in append order"""
        # the rows land in a host buffer the handle keeps (grown on demand) and the kept ones are copied out: a fresh
        # capacity-sized array per call (tens of MB of untouched pages under a pageable device-to-host copy, 23 at once) made
        # one sweep in five 20-30 ms slower (200 M genome: 0.165-0.184 s outliers among 0.155 s sweeps)
        buf = getattr(self, "_cand_buf", None)
        if buf is None or len(buf) < capacity:
            buf = self._cand_buf = np.zeros((max(int(capacity) + int(capacity) // 4, 1), 4), dtype=np.int32)
        k = ctypes.c_int64(0)
        _lib.check(self._lib.cl_cand_finish(self._h, int(final_cut), buf.ctypes.data_as(ctypes.c_void_p), int(capacity), ctypes.byref(k)))
        return buf[: int(k.value)].copy()

    def cand_finish_device(self, final_cut):
        """the same, leaving the boxes on the device -> (device pointer, rows); valid until the handle's next sweep finishes
        (cl_cand_finish_device: what comm.Comm.gather_device sends to the merging rank)"""
        ptr, k = ctypes.c_void_p(), ctypes.c_int64(0)
        _lib.check(self._lib.cl_cand_finish_device(self._h, int(final_cut), ctypes.byref(ptr), ctypes.byref(k)))
        return (ptr.value or 0), int(k.value)

    def set_sort_index(self, mode=1):
        """rows kept sorted by the in-strip coordinate, every eps' layout from a 2-pass strip sort of that order:
        0 = built at the handle's second sort (default), 1 = at the first, -1 = never (cl_set_sort_index)"""
        self._lib.cl_set_sort_index(self._h, int(mode))

    def set_count_reuse(self, on=True):
        """region-query words of the first run at an eps re-used by the later runs at that eps (default on; results
        identical either way -- cl_set_count_reuse of include/cloops_hip.h)"""
        self._lib.cl_set_count_reuse(self._h, 1 if on else 0)

    def sweep_plan(self, eps_list, min_pts_list):
        """announce a sweep `for ep in eps: for m in minPts:` (cLoops/pipe.py:241-281) in one call (cl_sweep_plan): one sort for all
        layouts, one region query per eps, the cut band only for the later runs.  Two empty lists end the plan."""
        ev = sorted(set(int(e) for e in eps_list))
        mv = sorted(set(int(m) for m in min_pts_list))
        ea = (ctypes.c_int32 * max(len(ev), 1))(*ev)
        ma = (ctypes.c_int32 * max(len(mv), 1))(*mv)
        _lib.check(self._lib.cl_sweep_plan(self._h, ea, len(ev), ma, len(mv)))

    def drop_indexes(self):
        """forget the q index, the fine layout, the last eps' layout and the cached counts (cl_chrom_drop_indexes): the next run
        sorts like the first run on a fresh dataset, allocations stay"""
        _lib.check(self._lib.cl_chrom_drop_indexes(self._h))

    def set_traversal(self, level=4):
        """how far a run works on its core / walker lists instead of LDS tiles over every PET: 0 .. 3 (default 3; results
        identical at every level -- cl_set_traversal of include/cloops_hip.h)"""
        self._lib.cl_set_traversal(self._h, int(level))

    def set_count_floor(self, min_pts):
        """the smallest minPts that will follow at the current eps (cl_set_count_floor); 0 = unknown"""
        self._lib.cl_set_count_floor(self._h, int(min_pts))

    def set_count_thresholds(self, min_pts_list):
        """the minPts values that will be asked for at the current eps (cl_set_count_thresholds); empty = unknown"""
        vals = [int(m) for m in min_pts_list]
        arr = (ctypes.c_int32 * max(1, len(vals)))(*vals)
        self._lib.cl_set_count_thresholds(self._h, arr, len(vals))

    def set_stream(self, stream):
        """move the idle handle to another caller-made / library-made stream of its device (cl_chrom_set_stream)"""
        _lib.check(self._lib.cl_chrom_set_stream(self._h, ctypes.c_void_p(stream)))

    def set_eps_list(self, eps_list):
        """the eps values that will be asked for (cl_set_eps_list): with a common divisor the layouts come from one fine sort; empty = unknown"""
        vals = [int(e) for e in eps_list]
        arr = (ctypes.c_int32 * max(1, len(vals)))(*vals)
        self._lib.cl_set_eps_list(self._h, arr, len(vals))

    def last_region_mode(self):
        """0 = the last enqueued run did a full region query, 1 = re-used the kept words, 2 = re-used them outside the cut band"""
        return int(self._lib.cl_last_region_mode(self._h))

    def set_layout_reuse(self, on=True):
        """keep the sorted arrays of the last eps and start further runs at that eps from a compaction by the cut
        (default on; results identical either way -- cl_set_layout_reuse of include/cloops_hip.h)"""
        self._lib.cl_set_layout_reuse(self._h, 1 if on else 0)

    def timing(self):
        t = _lib.ClTiming()
        _lib.check(self._lib.cl_get_timing(self._h, ctypes.byref(t)))
        return {k: getattr(t, k) for k, _ in _lib.ClTiming._fields_}

    def _boxes(self, max_label, copy):
        k = max_label + 1
        if k <= 0:
            return np.zeros(0, dtype=BOX_DTYPE)
        ptr = self._lib.cl_boxes_host(self._h)
        view = np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_int32)), shape=(k * 5,)).view(BOX_DTYPE)
        return view.copy() if copy else view

    def cluster_async(self, variant, eps, minPts, cut=0, want_labels=True, want_boxes=True):
        """Enqueue one run without blocking (at most two in flight); pair with wait().
        Labels land in one of two reusable pinned buffers (or stay on the device); `want_boxes=False`
        leaves the cluster table on the device too (cand_append / the distance statistics read it there)."""
        v = VARIANTS[variant]
        self.set_table_export(want_boxes)
        labels = self._pinned_labels(self._enq & 1) if want_labels else None
        _lib.check(self._lib.cl_cluster_async(self._h, v, int(eps), int(minPts), int(cut),
                                              labels.ctypes.data_as(ctypes.c_void_p) if want_labels else None))
        self._inflight.append((labels, bool(want_boxes)))
        self._enq += 1

    def cluster_pairs_async(self, variant, eps, minPts, cut=0, want_boxes=True):
        """Enqueue a run whose labels come back the way the reference holds them -- clustered PETs only (cDBSCAN2.py:186-191): one
        (row, label) pair per labelled PET in a page-locked buffer (cl_cluster_pairs_async); pair with wait_pairs()."""
        v = VARIANTS[variant]
        self.set_table_export(want_boxes)
        which = self._enq & 1
        if getattr(self, "_pin_pairs", None) is None:
            self._pin_pairs, self._pin_pairs_arr = [None, None], [None, None]
        if self._pin_pairs[which] is None:
            p = self._lib.cl_host_alloc(max(8, self.n * 8))
            if not p:
                raise MemoryError("cl_host_alloc(%d) failed" % (self.n * 8))
            self._pin_pairs[which] = p
            self._pin_pairs_arr[which] = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_int32)), shape=(self.n, 2))
        buf = self._pin_pairs_arr[which]
        _lib.check(self._lib.cl_cluster_pairs_async(self._h, v, int(eps), int(minPts), int(cut), buf.ctypes.data_as(ctypes.c_void_p), self.n))
        self._inflight.append((buf, bool(want_boxes)))
        self._enq += 1

    def wait_pairs(self, copy=False, defer=False):
        """completes the oldest run enqueued by cluster_pairs_async -> (ClusterResult with labels None, pairs int32 [K, 2]: row, label).
        defer=True: the pairs are still crossing PCIe when this returns -- call pairs_sync() before reading them (a loop over many
        chromosomes waits for all of them first: their copies then run side by side)"""
        buf, exported = self._inflight.pop(0)
        nc, ml = ctypes.c_int32(0), ctypes.c_int32(-1)
        self._lib.cl_set_pairs_defer(self._h, 1 if defer else 0)
        _lib.check(self._lib.cl_wait(self._h, ctypes.byref(nc), ctypes.byref(ml)))
        k = int(self._lib.cl_last_n_labelled(self._h))
        pairs = buf[:k].copy() if copy else buf[:k]
        boxes = self._boxes(ml.value, copy) if exported else None
        return ClusterResult(None, nc.value, ml.value, boxes, self.timing() if self._profiling else None), pairs

    def pairs_sync(self):
        _lib.check(self._lib.cl_pairs_sync(self._h))

    def cluster_rowmask_async(self, variant, eps, minPts, cut=0, want_boxes=True):
        """Enqueue a run whose labels come back in their smallest form: one bit per input row (set = clustered) and the labels of the
        set rows in ascending row order (cl_cluster_rowmask_async) -- 4 bytes per clustered PET + n / 8 bytes over PCIe instead of
        the 8 bytes per clustered PET of cluster_pairs_async; pair with wait_rowmask()."""
        v = VARIANTS[variant]
        self.set_table_export(want_boxes)
        which = self._enq & 1
        if getattr(self, "_pin_mask", None) is None:
            self._pin_mask = [None, None]
        nw = (self.n + 63) // 64
        nbytes = 8 * nw + 4 * max(self.n, 1)
        if self._pin_mask[which] is None:
            p = self._lib.cl_host_alloc(nbytes)
            if not p:
                raise MemoryError("cl_host_alloc(%d) failed" % nbytes)
            self._pin_mask[which] = p
        p = self._pin_mask[which]
        _lib.check(self._lib.cl_cluster_rowmask_async(self._h, v, int(eps), int(minPts), int(cut), ctypes.c_void_p(p), max(self.n, 1)))
        self._inflight.append((p, bool(want_boxes)))
        self._enq += 1

    def wait_rowmask(self, copy=False, defer=False):
        """completes the oldest run enqueued by cluster_rowmask_async -> (ClusterResult with labels None, mask uint64 [ceil(n / 64)],
        labels int32 [K] of the set rows in ascending row order); rows_of_mask(mask) gives the rows.  defer as in wait_pairs()."""
        p, exported = self._inflight.pop(0)
        nc, ml = ctypes.c_int32(0), ctypes.c_int32(-1)
        self._lib.cl_set_pairs_defer(self._h, 1 if defer else 0)
        _lib.check(self._lib.cl_wait(self._h, ctypes.byref(nc), ctypes.byref(ml)))
        k = int(self._lib.cl_last_n_labelled(self._h))
        nw = (self.n + 63) // 64
        mask = np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint64)), shape=(nw,))
        labels = np.ctypeslib.as_array(ctypes.cast(p + 8 * nw, ctypes.POINTER(ctypes.c_int32)), shape=(max(k, 1),))[:k]
        if copy:
            mask, labels = mask.copy(), labels.copy()
        boxes = self._boxes(ml.value, copy) if exported else None
        return ClusterResult(None, nc.value, ml.value, boxes, self.timing() if self._profiling else None), mask, labels

    def rows_of_mask(self, mask):
        """the set rows of a wait_rowmask() mask, ascending (int64)"""
        bits = np.unpackbits(np.ascontiguousarray(mask).view(np.uint8), bitorder="little")[: self.n]
        return np.flatnonzero(bits)

    def wait(self, copy=False):
        """Complete the oldest in-flight run -> ClusterResult (labels / boxes are VIEWS of pinned
        buffers that stay valid until two more runs have been enqueued, unless copy=True)."""
        labels, exported = self._inflight.pop(0)
        nc = ctypes.c_int32(0)
        ml = ctypes.c_int32(-1)
        _lib.check(self._lib.cl_wait(self._h, ctypes.byref(nc), ctypes.byref(ml)))
        boxes = self._boxes(ml.value, copy) if exported else None
        if labels is not None and copy:
            labels = labels.copy()
        return ClusterResult(labels, nc.value, ml.value, boxes, self.timing() if self._profiling else None)

    def cluster(self, variant, eps, minPts, cut=0, want_labels=True, want_boxes=True, pinned=False):
        """One synchronous run.  `pinned=True` returns labels and boxes as VIEWS of reusable
        page-locked buffers (valid until the next run on this chromosome) -- the fast path used
        by the sweep driver and bench."""
        v = VARIANTS[variant]
        if self._inflight:
            raise RuntimeError("asynchronous runs in flight: call wait() first")
        self.set_table_export(True)
        if want_labels:
            labels = self._pinned_labels(0) if pinned else np.empty(self.n, dtype=np.int32)
        else:
            labels = None
        nc = ctypes.c_int32(0)
        ml = ctypes.c_int32(-1)
        lp = labels.ctypes.data_as(ctypes.c_void_p) if want_labels else None
        _lib.check(self._lib.cl_cluster(self._h, v, int(eps), int(minPts), int(cut), lp,
                                        ctypes.byref(nc), ctypes.byref(ml)))
        boxes = self._boxes(ml.value, copy=not pinned) if want_boxes else None
        return ClusterResult(labels, nc.value, ml.value, boxes, self.timing() if self._profiling else None)

    def cluster_weighted(self, eps, minPts, wx=1, wy=1):
        """Variant 1 under the metric wx*|dX| + wy*|dY| <= eps: the labels of
        `cDBSCAN(mat * [1, wx, wy], eps, minPts)` (scripts/callStripes:37-52); boxes in unscaled
        coordinates (cl_cluster_weighted of include/cloops_hip.h)."""
        if self._inflight:
            raise RuntimeError("asynchronous runs in flight: call wait() first")
        labels = np.empty(self.n, dtype=np.int32)
        nc = ctypes.c_int32(0)
        ml = ctypes.c_int32(-1)
        _lib.check(self._lib.cl_cluster_weighted(self._h, int(eps), int(minPts), int(wx), int(wy),
                                                 labels.ctypes.data_as(ctypes.c_void_p), ctypes.byref(nc), ctypes.byref(ml)))
        return ClusterResult(labels, nc.value, ml.value, self._boxes(ml.value, copy=True), self.timing() if self._profiling else None)

    def last_n_in(self):
        """PETs of the last completed run that entered DBSCAN (after the cut filter)."""
        return int(self._lib.cl_last_n_in(self._h))

    # ---- distance statistics of the last completed run (K7; inputs of ests.estIntSelCutFrag) ----
    def dist_summary(self, cut=0):
        """-> dict(n_all=[inter, self], n_pos=[...], sumx=[...], sumxx=[...], xshift, loghist=uint64[3840])
        (cl_dist_summary: one pass; x = log2|d| - xshift)."""
        st = _lib.ClDsummary()
        _lib.check(self._lib.cl_dist_summary(self._h, int(cut), ctypes.byref(st)))
        return self._summary_dict(st)

    @staticmethod
    def _summary_dict(st):
        return {"n_all": [int(st.n_all[0]), int(st.n_all[1])], "n_pos": [int(st.n_pos[0]), int(st.n_pos[1])],
                "sumx": [float(st.sumx[0]), float(st.sumx[1])], "sumxx": [float(st.sumxx[0]), float(st.sumxx[1])],
                "xshift": float(st.xshift), "loghist": np.ctypeslib.as_array(st.loghist).astype(np.int64),
                "fine_lo": int(st.fine_lo), "fine": np.ctypeslib.as_array(st.fine).astype(np.int64) if st.fine_lo >= 0 else None}

    def dist_bin_hist(self, cut, lo, hi, shift):
        """histogram (int64[2048]) of (|d| - lo) >> shift over the self group's lo <= |d| < hi (cl_dist_bin_hist)"""
        out = np.zeros(2048, dtype=np.uint64)
        _lib.check(self._lib.cl_dist_bin_hist(self._h, int(cut), int(lo), int(hi), int(shift),
                                              out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))))
        return out.astype(np.int64)

    def sig_counts(self, windows, cut=0):
        """K8: interval counts for the significance test.  windows: int32 [R, 44] (lo[22], hi[22];
        A0..A10 then B0..B10) -> (int32 [R, 144] counts, N)  (cl_sig_counts of include/cloops_hip.h)."""
        w = np.ascontiguousarray(windows, dtype=np.int32).reshape(-1, 44)
        out = np.zeros((len(w), 144), dtype=np.int32)
        npets = ctypes.c_int64(0)
        _lib.check(self._lib.cl_sig_counts(self._h, int(cut), len(w), w.ctypes.data_as(ctypes.c_void_p),
                                           out.ctypes.data_as(ctypes.c_void_p), ctypes.byref(npets)))
        return out, int(npets.value)

    def neighbor_counts(self, eps, cut=0):
        out = np.full(self.n, -1, dtype=np.int32)
        _lib.check(self._lib.cl_neighbor_counts(self._h, int(eps), int(cut), out.ctypes.data_as(ctypes.c_void_p)))
        return out

/*
 * cloops_hip.h -- C ABI of libcloops_hip.so, the MI355X (gfx950) implementation of the
 * cDBSCAN / cDBSCAN2 / blockDBSCAN hot path of YaqiangCao/cLoops.
 *
 * Plain C: opaque handles, plain pointers and sizes, integer error codes; no exceptions
 * and no torch / numpy types cross this boundary.  Every entry point cites the reference
 * interface it replaces (paths relative to the reference checkout).
 *
 * The reference classes take `mat` = int64[N,3] rows [pointId, X, Y] (cLoops/io.py:192-217)
 * and produce `.labels` = {pointId: clusterId} for clustered points only.  Here a
 * chromosome's X and Y live in HBM as two int32 arrays (|X|,|Y| < 2^29, n < 2^31 - 1024 rows) and labels come
 * back as an int32 array ALIGNED TO THE INPUT ROWS (-1 = absent from `.labels`); the host
 * wrapper (cloops_amd/) turns that into the dict lazily.
 */
#ifndef CLOOPS_HIP_H
#define CLOOPS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes (return values; 0 = success) ------------------------------------- */
#define CL_OK             0
#define CL_ERR_ARG       -1   /* bad argument (null pointer, eps <= 0, unknown variant ...)   */
#define CL_ERR_HIP       -2   /* a HIP runtime call failed; see cl_last_error()               */
#define CL_ERR_EMPTY     -3   /* empty input where the reference raises IndexError
                                 (cLoops/cDBSCAN.py:77, cLoops/blockDBSCAN.py:74)             */
#define CL_ERR_DOMAIN    -4   /* coordinates outside the supported domain: |X|,|Y| >= 2^29, or
                                 (variant 2) X > Y / negative coordinate, where cDBSCAN2's
                                 trunc-toward-zero cell rule (cLoops/cDBSCAN2.py:69-70) stops
                                 being an exact grid                                          */
#define CL_ERR_GRID      -5   /* eps so small that the strip/row table would exceed 2^28 rows, or coordinate
                                 extent so large that (X+Y range / eps + 2) * 2^ceil(log2 eps) leaves int32
                                 (never for |X|,|Y| < 2^28).  (The candidate buffer of a sweep grows on demand.)      */
#define CL_ERR_NODEVICE  -6   /* no usable HIP device                                         */
#define CL_ERR_HASH      -7   /* cl_cand_finish: two different boxes shared a 64-bit hash under each of four independent
                                 salts (the pass is redone under another salt three times before this is reported) */

/* ---- clustering variants ---------------------------------------------------------- */
#define CL_VARIANT_CDBSCAN1  1   /* cLoops/cDBSCAN.py:6      class cDBSCAN      (scripts/callStripes:29,
                                    scripts/jd2saturation:23)                                  */
#define CL_VARIANT_CDBSCAN2  2   /* cLoops/cDBSCAN2.py:7     class cDBSCAN      (production, pipe.py:42) */
#define CL_VARIANT_BLOCK     3   /* cLoops/blockDBSCAN.py:6  class blockDBSCAN  (pipe.py:43)   */

/* A chromosome's PETs resident in HBM (+ the reusable device workspace and the result of
 * the last clustering run on it).  Replaces the per-step `joblib.load` of the .jd file
 * (cLoops/io.py:206-217, called at cLoops/pipe.py:58): X,Y cross PCIe once per chromosome,
 * not once per (eps, minPts) step. */
typedef struct cl_chrom cl_chrom;

/* One candidate-loop row: what cLoops/pipe.py:78-102 derives per cluster label. */
typedef struct {
    int32_t min_x, max_x, min_y, max_y;   /* bounding box of the member PETs            */
    int32_t count;                        /* number of member PETs                       */
} cl_box;

/* Per-kernel device timings of the last cl_cluster() call, HIP-event measured on the
 * stream the kernels ran on (only filled when profiling is enabled, see cl_set_profiling). */
typedef struct {
    float ms_keys;        /* K0  filter + key build                                      */
    float ms_sort;        /* K1  radix sort + gather + strip table                       */
    float ms_region;      /* K2  region query (neighbour count)  <- the roofline kernel  */
    float ms_union;       /* K3  core union-find + flatten + component keys              */
    float ms_border;      /* K4  border assignment (+ release fix-up / small-cluster drop) */
    float ms_table;       /* K5  ranks, labels scatter, cluster table                    */
    float ms_d2h;         /* labels + table to host                                      */
    float ms_total;       /* first kernel -> results on host                             */
    int64_t n_in;         /* PETs that entered DBSCAN (after the cut filter)             */
    int64_t n_strips;     /* rows of the strip table (C+1 term of the algorithmic bytes) */
    float ms_bracket;     /* calibration: the same event bracket around an EMPTY kernel (event packets +
                             dispatch gap + ~1 us of empty kernel); a phase that is one kernel (ms_region)
                             reads kernel time + about this much                                    */
    float ms_band;        /* K2 on the cut band of a run that re-uses the counts of its eps (traversal level 4: a kernel
                             of its own, inside the ms_sort phase; 0 = none)                        */
    int64_t n_queried;    /* PETs the region query of ms_region covered (0 = none ran; level 4 queries the whole base
                             layout once per eps: all rows)                                         */
} cl_timing;

/* Human-readable description of the last error on the calling thread. */
const char* cl_last_error(void);

/* Number of visible HIP devices (0 if none / no driver). */
int cl_device_count(void);

/*
 * Upload one chromosome.  `x`, `y`: n int32 coordinates (anchor mid-points, the X and Y
 * columns of the reference's `mat`, cLoops/io.py:49-57,192-203), host pointers if
 * `on_device` == 0, else device pointers on `device` that must stay valid for the life of
 * the handle (no copy is made).  `stream`: a hipStream_t to run on, or NULL for a private
 * stream.  n may be 0.  The handle reserves its whole per-PET workspace here (about 225 B/PET in
 * one allocation: sorted layouts, component / border / label arrays, both result slots, the sweep's
 * q index and a first candidate buffer), so that no run pays for device or pinned-host allocations.
 */
int cl_chrom_create(int device, void* stream, const int32_t* x, const int32_t* y, int64_t n,
                    int on_device, cl_chrom** out);
void cl_chrom_destroy(cl_chrom* c);
int64_t cl_chrom_size(const cl_chrom* c);

/*
 * One clustering run = `DBSCAN(mat, eps, minPts)` of cLoops/pipe.py:70 preceded by the
 * distance pre-filter of cLoops/pipe.py:59-63 (`cut` > 0 keeps rows with Y-X >= cut; pass 0
 * for the bare class constructors cLoops/cDBSCAN.py:12, cLoops/cDBSCAN2.py:13,
 * cLoops/blockDBSCAN.py:13).
 *
 *   labels_out   n int32, host memory (or NULL to leave labels on the device):
 *                cluster id of every input row, -1 for rows that are filtered, noise, or
 *                otherwise absent from the reference's `.labels`.  Ids are exactly the
 *                reference's ids (variant 1 keeps its gaps).
 *   n_clusters   number of distinct cluster ids;   max_label: largest id (or -1).
 *
 * Errors: CL_ERR_EMPTY for variants 1/3 when no row survives the filter and cut == 0
 * (the reference's IndexError); with cut > 0 an empty survivor set is not an error
 * (cLoops/pipe.py:64-65 returns early) and yields zero clusters.
 */
int cl_cluster(cl_chrom* c, int variant, int32_t eps, int32_t min_pts, int32_t cut,
               int32_t* labels_out, int32_t* n_clusters, int32_t* max_label);

/*
 * Variant 1 under an axis-weighted city-block metric: the result of
 *     cDBSCAN(mat * [1, wx, wy], eps, minPts)            (cLoops/cDBSCAN.py:12)
 * i.e. of scripts/callStripes:37-52 (singleStripDBSCAN), which multiplies the X or the Y column by
 * `ext` (50) before clustering to find stripes: two PETs are neighbours iff wx*|dX| + wy*|dY| <= eps.
 * The scaled coordinates (up to 1.25e10) never exist as int32: the kernels work on 64-bit rotated
 * coordinates.  labels_out / n_clusters / max_label as for cl_cluster (ids of variant 1, gaps kept);
 * cl_get_boxes() afterwards returns the boxes in UNSCALED coordinates (callStripes:59-66 divides the
 * scaled extrema by ext again).  1 <= wx, wy <= 4096.  Synchronous; no cut (callStripes uses none).
 */
int cl_cluster_weighted(cl_chrom* c, int32_t eps, int32_t min_pts, int32_t wx, int32_t wy,
                        int32_t* labels_out, int32_t* n_clusters, int32_t* max_label);

/*
 * Asynchronous form for sweeps with a fixed cut (every (eps, minPts) step of the same
 * chromosome is independent there): cl_cluster_async() enqueues one run and returns without
 * blocking; cl_wait() completes the OLDEST outstanding run and reports its cluster count.
 * At most two runs may be in flight; the D2H copy of run k then overlaps the kernels of run
 * k+1.  `labels_out` must stay valid (and should be pinned, cl_host_alloc) until the matching
 * cl_wait() returns; use a different buffer for the second in-flight run.
 * cl_cluster() == cl_cluster_async() + cl_wait().
 */
int cl_cluster_async(cl_chrom* c, int variant, int32_t eps, int32_t min_pts, int32_t cut,
                     int32_t* labels_out);
int cl_wait(cl_chrom* c, int32_t* n_clusters, int32_t* max_label);

/*
 * Cluster table of the last run, indexed by cluster id 0..max_label (count == 0 for the
 * id gaps of variant 1): bounding box + size per label, the inputs of cLoops/pipe.py:83-102.
 * `boxes_out`: (max_label+1) rows of host memory.
 */
int cl_get_boxes(cl_chrom* c, cl_box* boxes_out);
/* Zero-copy view of the same table: (max_label+1) rows in pinned host memory owned by the
 * handle, valid until the next cl_wait()/cl_cluster() that lands in the same result slot
 * (i.e. for at least one more run).  NULL if there is no result or no cluster. */
const cl_box* cl_boxes_host(const cl_chrom* c);
/* Number of PETs of the last completed run that passed the cut filter and entered DBSCAN
 * (`len(mat)` after cLoops/pipe.py:59-62). */
int64_t cl_last_n_in(const cl_chrom* c);

/*
 * Region query alone (kernel K2): neighbour counts |{q : |Xp-Xq|+|Yp-Yq| <= eps}|, self
 * included, per input row (-1 for rows removed by `cut`) -- the quantity the reference
 * compares with minPts (cLoops/cDBSCAN.py:168,177 `len(regionQuery)`,
 * cLoops/cDBSCAN2.py:333-334 `n + cell_pt_num`).  Exposed for parity tests and for the
 * roofline measurement of bench.py.
 */
int cl_neighbor_counts(cl_chrom* c, int32_t eps, int32_t cut, int32_t* counts_out);

/*
 * Distance statistics of the last completed run -- the inputs of cLoops/ests.py:36-61
 * (estIntSelCutFrag), reduced on the GPU instead of materialising the reference's `dis` / `dss`
 * lists (cLoops/pipe.py:63,106-109).  Group 0 = PETs of inter-ligation clusters, group 1 = PETs of
 * self-ligation clusters plus the PETs removed by `cut` (pass the SAME cut as to cl_cluster).
 *   cl_dist_summary  : ONE pass: n_all = len(dis) / len(dss); n_pos, sumx, sumxx = count, sum and sum of squares of
 *                      x = log2|d| - xshift over d != 0 (mean and standard deviation follow from them); loghist =
 *                      histogram of the self group's |d| over bins that are monotone in |d|:
 *                      bin = floor(log2 d) * 128 + (the 7 bits below the leading one)  -- the first level of the EXACT
 *                      median (ests.py:52,58), fine enough to end the search in one more pass for any realistic data
 *   cl_dist_bin_hist : refinement: histogram of (|d| - lo) >> shift over the self group's lo <= |d| < hi (2048 bins)
 * Both are additive over chromosomes (and over GPUs), which is what the sweep driver uses.  The kernels read the
 * run's sorted arrays (distance = the in-strip coordinate) and its labels in sorted order -- no row-aligned labels
 * are needed.
 */
#define CL_DIST_LOGBINS 3840
typedef struct {
    int64_t n_all[2];
    int64_t n_pos[2];
    double sumx[2];
    double sumxx[2];
    double xshift;
    uint64_t loghist[CL_DIST_LOGBINS];
    int64_t fine_lo;                 /* >= 0: `fine` holds the exact histogram of the self group's fine_lo <= |d| < fine_lo + 2048 */
    uint64_t fine[2048];
} cl_dsummary;
int cl_dist_summary(cl_chrom* c, int32_t cut, cl_dsummary* out);
int cl_dist_bin_hist(cl_chrom* c, int32_t cut, uint32_t lo, uint32_t hi, int shift, uint64_t* hist2048);

/*
 * Interval counting for the significance test of candidate loops (cLoops/cModel.py:60-80,108-143):
 * for every record, 11 A windows (the anchor iva + the 10 shifted windows of getNearbyPairRegions,
 * cModel.py:83-105) and 11 B windows, each [lo, hi] inclusive.  `windows`: n_records x 44 int32 laid
 * out as lo[22] (A0..A10, B0..B10) then hi[22].  `out`: n_records x 144 int32:
 *   [0..10]   |S(A_k)|   with S(W) = {PETs with X in W} | {PETs with Y in W}     (ra = [0])
 *   [11..21]  |S(B_l)|                                                            (rb = [11])
 *   [22]      rab = |{X in A_0} & {Y in B_0}|                                     (cModel.py:79)
 *   [23 + 11*k + l]  |S(A_k) & S(B_l)|
 * `cut` > 0 restricts the PETs to Y-X >= cut like parseJd(f, cut) (cLoops/io.py:213-216);
 * *n_pets = number of PETs in the model (N of cModel.py:270).
 */
int cl_sig_counts(cl_chrom* c, int32_t cut, int32_t n_records, const int32_t* windows, int32_t* out, int64_t* n_pets);

/* Device pointer to the labels of the last run (n int32, row aligned) -- lets the caller
 * keep results on the GPU (e.g. to hand them to RCCL) without a host round trip.  NULL if the run did not
 * produce row-aligned labels (see cl_set_device_labels). */
const int32_t* cl_labels_device(const cl_chrom* c);

/*
 * Candidate loops of a sweep, kept on the device (kernels K10).  Replaces, for one chromosome, the record lists that
 * cLoops/pipe.py:241-281 carries from step to step: `cl_cand_append` classifies the cluster table of the last
 * completed run like pipe.py:83-97 (skip degenerate boxes; inter-ligation iff maxX < minY) and appends the
 * inter-ligation boxes, in ascending cluster id, with the given step number; it returns how many inter- and
 * self-ligation boxes the run had.  `cl_cand_finish` applies combineTwice (pipe.py:155-174: a box survives in the
 * step of its first appearance, duplicates inside one step all stay) and filterClusterByDis (pipe.py:130-143:
 * floor mid-point distance >= final_cut) and copies the surviving boxes {minX, maxX, minY, maxY} to boxes_out in
 * append order -- the order of the reference's record list.  capacity / *n_out in rows of 4 int32.
 * cl_cand_reset starts a new sweep.
 */
int cl_cand_reset(cl_chrom* c);
/* The labels of a run as the reference holds them: cDBSCAN(mat, eps, minPts).labels is a dict of the CLUSTERED points only
 * (cDBSCAN2.py:186-191, cDBSCAN.py:143-152).  Like cl_cluster_async, but instead of n row-aligned labels the run leaves one
 * (row, label) int32 pair per labelled PET, in no particular order: its last kernel stages them in device memory and
 * cl_wait copies exactly cl_last_n_labelled(c) of them into `pinned_pairs_out` (page-locked host memory: cl_host_alloc;
 * capacity_pairs pairs -- n always suffices; a run that labels MORE than capacity_pairs PETs makes cl_wait return
 * CL_ERR_ARG and copies nothing).  A sweep that wants labels on the host every run moves 8 bytes per clustered PET
 * over PCIe instead of 4 bytes per PET.  Rotated variants at traversal level >= 3 and minPts 2 .. 128 only (CL_ERR_ARG
 * otherwise). */
int cl_cluster_pairs_async(cl_chrom* c, int variant, int32_t eps, int32_t min_pts, int32_t cut, int32_t* pinned_pairs_out,
                           int64_t capacity_pairs);
int64_t cl_last_n_labelled(const cl_chrom* c);
/* The same set -- the clustered points and their cluster ids, what cDBSCAN(mat, eps, minPts).labels holds (cDBSCAN2.py:186-191,
 * cDBSCAN.py:143-152) -- in its smallest form: ceil(n / 64) 64-bit mask words (bit r % 64 of word r / 64 set <=> input row r is
 * clustered), followed by one int32 label per set bit in ASCENDING ROW ORDER.  4 bytes per clustered PET + n / 8 bytes cross PCIe
 * instead of 8 bytes per clustered PET: the label-inclusive sweep is bound by that copy.  `pinned_out` (page-locked host memory:
 * cl_host_alloc) must hold 8 * ceil(n / 64) + 4 * capacity_labels bytes; cl_wait copies the mask and exactly
 * cl_last_n_labelled(c) labels (a run that labels MORE than capacity_labels PETs makes cl_wait return CL_ERR_ARG and copies
 * nothing; n always suffices); cl_set_pairs_defer / cl_pairs_sync apply to this copy as well.  Same restrictions as
 * cl_cluster_pairs_async. */
int cl_cluster_rowmask_async(cl_chrom* c, int variant, int32_t eps, int32_t min_pts, int32_t cut, void* pinned_out,
                             int64_t capacity_labels);
/* cl_set_pairs_defer(c, 1): cl_wait of a pairs run returns with the copy of the pairs to the host still in flight (their number
 * is known: cl_last_n_labelled); cl_pairs_sync(c) completes it.  A host that collects many chromosomes waits for all of them
 * first and syncs afterwards, so that their copies cross PCIe side by side. */
void cl_set_pairs_defer(cl_chrom* c, int enabled);
int cl_pairs_sync(cl_chrom* c);

/* One step of a sweep in ONE asynchronous call: cl_cluster_async(labels_out = NULL) followed, in the run's own stream,
 * by what cl_cand_append and cl_dist_summary do for that run (same `cut`).  After cl_wait the results are on the host:
 * cl_step_result copies them out without touching the GPU.  One step in flight per chromosome.  `fine_lo` >= 0 (a guess
 * of where the self group's median will fall, e.g. the previous step's) makes the summary also histogram the distances
 * fine_lo .. fine_lo + 2047 exactly: when the median lands there no refinement pass (cl_dist_bin_hist) is needed. */
int cl_cluster_step_async(cl_chrom* c, int variant, int32_t eps, int32_t min_pts, int32_t cut, int32_t step, int64_t fine_lo);
int cl_step_result(cl_chrom* c, int64_t* n_inter, int64_t* n_self, cl_dsummary* out);
int cl_cand_append(cl_chrom* c, int32_t step, int64_t* n_inter, int64_t* n_self);
int cl_cand_finish(cl_chrom* c, int32_t final_cut, int32_t* boxes_out, int64_t capacity, int64_t* n_out);
/* The same, leaving the surviving boxes ON THE DEVICE: *dev_rows_out = n_out rows of {minX, maxX, minY, maxY} int32 in device
 * memory of the handle (valid until its next sweep finishes or it is destroyed) -- what cl_comm_gather_device
 * (include/cloops_comm.h) sends to the rank that merges the ranks' tables (cLoops/pipe.py:119-127 merges in the parent),
 * without a detour through the host on the sending ranks. */
int cl_cand_finish_device(cl_chrom* c, int32_t final_cut, const int32_t** dev_rows_out, int64_t* n_out);

/* The cluster table of a run goes to pinned host memory at its end (cl_boxes_host / cl_get_boxes); enabled = 0
 * skips that copy for the following runs (callers that only use cl_cand_append / the distance statistics). */
void cl_set_table_export(cl_chrom* c, int enabled);

/* Row-aligned device labels for runs WITHOUT a host destination (labels_out == NULL): enabled = 1 (default) keeps
 * producing them (for cl_labels_device); enabled = 0 skips the scatter to input-row order -- the sweep driver only
 * needs the cluster table and the distance statistics, which work on the sorted order. */
void cl_set_device_labels(cl_chrom* c, int enabled);

/* Enable (1) / disable (0) HIP-event timing of the kernels of subsequent runs. */
void cl_set_profiling(cl_chrom* c, int enabled);
int cl_get_timing(const cl_chrom* c, cl_timing* out);

/* ---- A sweep in one call -------------------------------------------------------------------------------------------------
 * The reference's driver runs `for ep in eps: for m in minPts:` over every chromosome (cLoops/pipe.py:241-281, the lists of a mode:
 * pipe.py:310-344).  A caller that is about to do the same tells the handle ONCE:
 *     cl_sweep_plan(c, eps, n_eps, min_pts, n_min_pts);
 * and then calls cl_cluster / cl_cluster_async / cl_cluster_step_async per (eps, minPts, cut) as before.  With the plan the handle
 * sorts its rows once for all announced eps (when they share a divisor), keeps the layout of an eps for its runs, makes the
 * neighbour counts of an eps once -- exact enough for every announced minPts -- and re-queries only the cut band of the later runs.
 * Results are identical with and without a plan; without one every run pays for itself.  n_eps = n_min_pts = 0 ends the plan.
 * (The plan is the announcement of the two lists: cl_set_sort_index(1) + cl_set_eps_list + cl_set_count_thresholds.  Layout reuse,
 * count reuse and the traversal level keep their values -- on / on / 4 unless the caller changed them.)
 * The cl_set_* entries below are the plan's parts, kept for tests and measurements (each documents what it switches); a caller
 * needs none of them. */
int cl_sweep_plan(cl_chrom* c, const int32_t* eps, int32_t n_eps, const int32_t* min_pts, int32_t n_min_pts);

/* Sorted-layout reuse (default: enabled).  The sorted order of a chromosome's PETs depends on eps only (not on
 * minPts; a cut only removes rows), and the sweep of cLoops/pipe.py:241-281 walks eps in its OUTER loop: with reuse
 * enabled the handle keeps the sorted arrays of the last eps and every further run at that eps starts from one
 * stable stream compaction by the cut (pipe.py:59-62) instead of a sort.  Results are identical either way; nothing
 * of a result is kept between runs.  enabled = 0: every run sorts for itself (used by benchmarks that repeat one
 * (eps, minPts) and must pay the whole run every time). */
void cl_set_layout_reuse(cl_chrom* c, int enabled);

/* The q index.  A run's sorted order is (strip, q) with ties in input-row order, and only the strip depends on eps.
 * A handle that sorts more than one layout (the eps loop of cLoops/pipe.py:241-281) keeps its rows sorted by q once
 * (12 B/PET, 4 radix passes) and gets every layout from a stable sort of that sequence by the strip bits alone
 * (2 passes instead of 5) -- the same permutation.  mode 0 (default): the index is built at the handle's second
 * sort, so a one-shot run never pays for it; 1: at the first sort (the sweep driver, which knows more eps are
 * coming); -1: never.  Results are identical in every mode. */
void cl_set_sort_index(cl_chrom* c, int mode);

/* Region-query reuse inside one eps (default: enabled; needs layout reuse).  The neighbour count of a PET
 * (cDBSCAN.py:186-205 regionQuery, cDBSCAN2.py:333-334) does not depend on minPts, and a cut (pipe.py:59-62) removes
 * only PETs whose distance is below it -- in the sorted layout a prefix of every strip -- so the count of a PET changes
 * between two runs of one eps only if its distance lies within eps above the larger of their cuts.  The sweep of
 * cLoops/pipe.py:247-250 walks minPts (descending) INSIDE eps: the handle keeps the per-PET words of the first run at
 * an eps (counts saturated at its minPts) and every later run at that eps with an announced minPts runs the region
 * query on the cut band alone; the other PETs' words are read in place.  Results are identical with enabled = 0
 * (every run does its own full region query).
 * cl_set_count_thresholds: the minPts values the caller will ask for at the current eps (the inner loop of
 * cLoops/pipe.py:247-250; the sweep driver knows its list; values outside 2..128 are ignored).  The first run then
 * keeps every count as exact as the tests `count >= minPts` of those values need it (a PET whose count is known to lie
 * between two neighbouring values of the list is not counted further).
 * cl_set_count_floor: the older, coarser form -- every minPts from min_pts up to the first run's is served (counts
 * exact from there up).  Either call replaces what the other announced; nothing announced (default) = only runs that
 * repeat the first run's minPts re-use its words.
 * cl_last_region_mode: what the last enqueued run did -- 0 full region query, 1 words re-used as they were (same
 * cut), 2 words carried through the compaction + region query on the band. */
/* cl_set_eps_list: the eps values the caller will ask for (the outer loop of cLoops/pipe.py:241-281; the sweep driver knows its
 * list).  When they share a divisor w >= 16 with max(eps) / w <= 8 (Hi-C mode 3: 5000 / 7500 / 10000 -> 2500) the handle sorts its
 * rows once by strips of width w; the layout of every announced eps is then a per-strip merge of that order (a strip of width
 * k w = k consecutive strips of width w) instead of a sort -- the same (strip, distance) order; PETs of one strip at EQUAL distance
 * follow each other by run instead of by input row, an order no result depends on.  n = 0 forgets the list. */
void cl_set_eps_list(cl_chrom* c, const int32_t* eps, int32_t n);
/* cl_chrom_drop_indexes: forget every order and count the handle has derived from its rows (q index, fine layout, the layout of the
 * last eps, cached neighbour counts); allocations stay.  The next run pays what the first run on a fresh dataset pays for its
 * sorts -- measurements of "one dataset, one sweep" (cLoops/pipe.py:247-275 sweeps a dataset once) without re-uploading. */
int cl_chrom_drop_indexes(cl_chrom* c);
void cl_set_count_reuse(cl_chrom* c, int enabled);
void cl_set_count_floor(cl_chrom* c, int32_t min_pts);
void cl_set_count_thresholds(cl_chrom* c, const int32_t* min_pts, int32_t n);
int cl_last_region_mode(const cl_chrom* c);

/* How the part of a rotated run behind the region query (components, border rule, labels) walks the data.  The reference
 * expands clusters from core points only (cDBSCAN2.py:114-192 queryGrid, cDBSCAN.py:155-184 expandCluster); levels 3 and 4
 * do the same: the run's cores and its non-core PETs that have a neighbour are compacted into two lists right behind the
 * region query and nothing else is touched again -- level 3 from a copy of the layout compacted by the run's cut, level 4
 * (DEFAULT) straight from the eps' base layout (a cut removes a prefix of every strip: nothing is copied, the count cache
 * lives in base positions).  Levels 0..2 keep the LDS-tile kernels over every PET of the run for the components (0), the
 * border rule (<= 1) and the labels (<= 2).  Results are identical at every level (the tests compare all five); values
 * outside 0..4 are clamped. */
void cl_set_traversal(cl_chrom* c, int level);

/* A HIP stream for cl_chrom_create(..., stream, ...) made by the library (for callers without a HIP binding of their
 * own, like the ctypes host side).  Several handles may share one stream: their runs then execute in enqueue order in
 * that stream; their D2H copies go through ONE copy stream that belongs to the stream (made when the first handle has
 * labels to copy), behind an event of the run, so that a run's labels cross PCIe while the next handle's kernels
 * execute.  (A handle on a stream the caller made otherwise issues its copies in that stream; a handle with a stream of
 * its own -- stream == NULL at creation -- overlaps them with its next run through a copy stream of its own.)  Make the
 * streams before anything else that makes streams: the runtime deals its hardware queues in creation order.  The sweep driver keeps a few shared streams per
 * device instead of one per chromosome: how many kernels run side by side is then the application's choice, not a
 * property of how the runtime maps dozens of streams onto its hardware queues.  Destroy a stream after its handles. */
void* cl_stream_create(int device);
/* cl_chrom_set_stream: move an idle handle (no run in flight) that was created on a caller's / library-made stream to another
 * stream of the same device -- the sweep driver balances its chromosomes over its shared streams when a sweep starts (a handle is
 * bound to a stream when it is uploaded, long before the set of chromosomes of a sweep is known).  Waits for the handle's old
 * stream.  CL_ERR_ARG for a handle with a stream of its own (stream == NULL at creation) or a NULL stream. */
int cl_chrom_set_stream(cl_chrom* c, void* stream);
void cl_stream_destroy(void* stream);

/* Page-locked host memory for result buffers (labels_out / boxes_out / counts_out): D2H
 * copies into pinned memory run at PCIe rate instead of through a staging buffer.  Plain
 * malloc'ed memory works everywhere too, only slower. */
void* cl_host_alloc(int64_t bytes);
void cl_host_free(void* p);

/* A new chromosome handle made of `m` rows of a resident one, in the order given (rows may repeat): what
 * scripts/jd2saturation:32-55 does with `mat[ns, :]` + joblib.dump for every re-sampling depth -- here the rows are
 * gathered on the device, only the row list crosses PCIe.  The new handle is independent of `src` (own device memory,
 * own stream); destroy it with cl_chrom_destroy. */
int cl_chrom_subsample(cl_chrom* src, const int64_t* rows, int64_t m, cl_chrom** out);

/* Testing hook: the per-PET workspace of a handle is reserved as ONE allocation when the chromosome is uploaded (best
 * effort: if that allocation fails the buffers are allocated one by one at the first run).  extra_bytes > 0 is added to
 * the size of that allocation for the handles created afterwards, so that a test can make it fail; 0 restores it. */
void cl_debug_arena_overcommit(int64_t extra_bytes);

/* Library version: major*10000 + minor*100 + patch. */
int cl_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CLOOPS_HIP_H */

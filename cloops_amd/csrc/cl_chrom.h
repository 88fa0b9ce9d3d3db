// cl_chrom.h -- host side shared by the translation units: the chromosome handle (device buffers, result slots, sweep
// state), the structures the step tail exchanges with the K7 / K10 kernels, and the functions / kernels one unit uses
// from another.  A __global__ function may be launched from a unit that only sees its declaration: the launch goes
// through the defining unit's host stub.
#pragma once
#include "cl_common.h"
#include "cl_table.h"

#define BIGTPB 1024
#define AGG_H 512
// slot of `key` in a workgroup's LDS hash table of AGG_H keys (-1 = table crowded: the caller goes to global memory)
__device__ __forceinline__ int agg_slot(int* keys, int key)
{
    unsigned h = ((unsigned)key * 2654435761u) >> 23;            // 9 bits
    for (int probe = 0; probe < 24; ++probe) {
        const int old = atomicCAS(&keys[h], -1, key);
        if (old == -1 || old == key) return (int)h;
        h = (h + 1) & (AGG_H - 1);
    }
    return -1;                                                    // table crowded: caller goes to global memory
}
#define FLAT_PER 4          // PETs per thread
struct Stats { int amin, amax, vmin, vmax, xmin, xmax, ymin, ymax; };
__device__ __forceinline__ int je_minus(const int* __restrict__ cstart, int j) { return cstart[j + 1] - cstart[j]; }
static inline int bits_for(unsigned v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }


#define CL_FINE_KMAX 8           // most runs per strip a layout is merged from (cl_set_eps_list)
#define K7_BLOCKS 2048           // workgroups (= fixed partial sums) of the K7 reductions: 8 waves per SIMD (512 left the loads
#define K7_STEP_BLOCKS 256       // ... of the sweep step's k7_summary: 1024 threads each (every workgroup ends with up to 5 888 atomics into the two
#define K7_STEP_THREADS 1024      //     histograms -- 2048 x 256 threads: 67 us per chr1 step, 256 x 1024: 47 us)
#define K7_LOGBINS 3840          // 30 octaves x 128: bin = floor(log2 d) * 128 + the next 7 bits of d (monotone in d)
#define K7_FINE 2048
#define K7_XSHIFT 11.0           // sums are taken over x = log2|d| - K7_XSHIFT (less cancellation in sum x^2 - (sum x)^2 / n)
#define CAND_BLOCK 2048

struct K7Src {
    int sorted;                                         // 0 = input rows, 1 = the run's sorted arrays, 2 = the run's lists (sv / slab hold one entry per core and
                                                        // walker, dM their number)
    int n; int M; int v0;
    const int* dM;                                      // if set: M is read from the device (the run has not been waited for yet)
    const int* X; const int* Y; const int* labels;      // input rows (+ row-order labels)
    const int* sv; const int* slab;                     // sorted q and sorted-order labels of [0, M)
    const int* dh;                                      // if set: dh[d] = number of input rows with Y - X == d for 0 <= d < cut, and no row has
                                                        // Y - X < 0: the PETs dropped by the cut come from it, not from a pass over the rows
};

struct K7Part { double sx[2]; double sxx[2]; long long n_all[2]; long long n_pos[2]; };

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    bool fresh = false;               // (re)allocated since the flag was last cleared
    bool owned = true;                // false: a slice of the handle's arena (never freed on its own)
    int ensure(size_t need)
    {
        if (need <= bytes) return CL_OK;
        fresh = true;
        if (p && owned) (void)hipFree(p);
        p = nullptr; bytes = 0; owned = true;
        size_t want = need + need / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "hipMalloc", hipGetErrorString(e));
        bytes = want;
        return CL_OK;
    }
    void adopt(void* slice, size_t n) { if (p && owned) (void)hipFree(p); p = slice; bytes = n; owned = false; fresh = true; }
    void release() { if (p && owned) (void)hipFree(p); p = nullptr; bytes = 0; owned = true; }
    template <typename T> T* as() { return (T*)p; }
};

struct cl_chrom {
    long long tmp_n = -1;             // n the rocPRIM temporary storage was last sized for
    DevBuf arena;                     // ONE allocation behind the per-PET workspace of the handle (ensure_workspace): the first
                                      // run of a handle pays one hipMalloc instead of forty
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t n = 0;
    int *d_x = nullptr, *d_y = nullptr;
    bool own_xy = false;
    Stats st{};
    // workspace
    DevBuf keys_in, keys_out, vals_in, vals_out, sort_tmp, scan_tmp;
    DevBuf qb_key, qb_val;            // the q index (k_make_qkeys): rows sorted by q, persistent
    int qindex_layout = -1;           // layout the q index was built for (-1: none)
    // the fine layout (cl_set_eps_list): rows sorted by (strip of width fine_w, q) once; the layout of an eps = k fine_w is a per-strip merge of it
    DevBuf fq, fsp, frow, fstrip;
    int fine_w = 0;                   // announced common divisor of the sweep's eps values (0: none)
    int fine_valid_w = 0, fine_layout = -1;       // what the fine buffers hold
    GridParams fine_g;
    int sort_index_mode = 0;          // cl_set_sort_index: 0 = build at the second sort, 1 = at the first, -1 = never
    long long n_sorts = 0;            // layouts sorted on this handle so far
    DevBuf sv, sa, strip, cnt, parent, root, head, headidx, cellfirst, compkey, ncore, bsize, owner, state;
    DevBuf flag, rankscan, ulist, lo, hi, recs, counters, chainflag, chainhead, usize, b_cstart, b_ckey, b_nb, b_cx, b_cy, tile_s0;
    int* h_pinned = nullptr;          // small pinned staging (stats, block scalars)
    struct StripPlan { int layout, eps, maxlen; };
    std::vector<StripPlan> plans;     // longest strip over all rows per (layout, eps): picks the sort path
    u32* srow = nullptr;              // sorted position -> input row of the run being enqueued
    // working set of the run being enqueued: sorted (q, sp), strip table, tile table.  They alias either the
    // workspace buffers (sv, sa, strip, tile_s0) or, for a run without cut filter, the base layout itself.
    int *w_sv = nullptr, *w_sa = nullptr, *w_strip = nullptr, *w_tile = nullptr;
    // Base layout: the sorted arrays of ALL rows (cut = 0) for one (variant layout, eps), kept until eps changes.
    // The sort order does not depend on minPts and a cut only REMOVES rows, so every further run of a sweep at this
    // eps is one stable stream compaction of the base layout instead of five radix passes (cLoops/pipe.py:241-281
    // walks eps in the outer loop).  Nothing of a result is kept: neighbour counts, components, labels are redone.
    DevBuf bq, bsp, brow, bstrip, btile, sel_tmp;
    struct BaseLayout { bool valid = false; int layout = -1, eps = 0; } base;
    // Count cache: the K2 words (cl_common.h "K2W") of the FIRST run on a base layout, kept while the layout lives.  A word
    // holds the PET's neighbour count (saturated at `cap`, exact as far as the minPts values of `tmask` need it: GridParams::tmask) and does not depend on minPts otherwise;
    // a cut removes a prefix of every strip (q IS the distance), so between two runs of one eps the count of a PET changes
    // only if q - eps < max(the two cuts): every later run with a minPts of the set takes the words of the PETs beyond that
    // band as they are (k_cut_copy<true> carries them through its compaction) and runs K2 on the band alone.  Results are
    // identical with the cache switched off (cl_set_count_reuse); cLoops/pipe.py:247-250 walks minPts inside eps, descending.
    struct CountCache { bool valid = false; bool cut_on_base = false /* made on the base layout by a run under a cut: the PETs within eps above thr counted removed neighbours -- every run under a cut re-queries its band */; bool base_space = false /* level 4: words by base position, hints in base positions (k_lists.hip) */; int layout = -1, eps = 0, thr = 0 /* q threshold of the run's cut, 0 = none */, cap = 0; u32 tmask[4] = {0, 0, 0, 0} /* the minPts values the words serve, bit t - 1 */; } rc;
    DevBuf rc_cnt, rc_pre, rc_poff, rc_dpre, rc_D, rc_blen;   // the words in the sorted order of the run that made them; per strip: PETs its cut removed
                                      // from the strip / from all strips up to and including it
    bool reuse_counts = true;         // cl_set_count_reuse
    int count_floor = 0;              // cl_set_count_floor: smallest minPts later runs of this eps will ask for (0 = unknown)
    u32 count_tmask[4] = {0, 0, 0, 0}; // cl_set_count_thresholds: the minPts values themselves (bit t - 1; all zero = not announced)
    int* w_cnt = nullptr;             // where K2 writes the words of the run being enqueued (cnt or rc_cnt)
    WordSrc ws{};                     // where its consumers read them
    int init_nclr = 0;                // > 0: the run being enqueued has not cleared its key bitmap / counters yet (words to clear)
    DevBuf blk_tmp;                   // block sums / offsets of the in-kernel scan over the key bitmap (k_rank_scan)
    DevBuf rootlist, cflag8;          // K3: the components' roots (k_flatten); per PET: core / opens a chain / ends one (k_chain_flags)
    // List form of a run (k_lists.hip): the run's CORES and its WALKERS (non-core PETs with a neighbour: the only ones a border
    // rule can label) as compact arrays in sorted order, plus a rank index over the run's positions (one bit per PET and the
    // exclusive core / walker count in front of every 64-PET group), so that any position range of the layout maps to a
    // contiguous range of the core array in O(1).  K3 / K4 / K5 then touch nothing else.
    DevBuf l_mask, l_rank, l_blk, l_cstrip, l_wpos, l_wenc, l_dist, l_aux;
    // ... built straight from the BASE layout (traversal level 4): a cut removes a prefix of every strip, so the list kernels
    // apply it by index (per strip: first kept base index, end of the cut band, PETs removed: l_tab) and a re-using run
    // copies nothing at all; variant 2's cell minima are a property of the base layout (bkey, once per eps) except in the one
    // cell per strip the cut goes through (l_fix, per run)
    DevBuf l_tab, l_fix, bkey;
    bool bkey_valid = false;          // bkey belongs to the current base layout
    int run_level = 0;                // traversal level of the run being enqueued (run_sort_and_count decides: level 4 needs the count cache's tables)
    const int* w_dM = nullptr;        // device: PETs that entered DBSCAN in the run being enqueued
    bool run_rows = true;             // the run being enqueued produces row-aligned labels
    int2* pairs_out = nullptr;        // cl_cluster_pairs_async: (row, label) of every labelled PET, staged by the label kernel in the slot's
    long long pairs_cap = 0;          // device buffer and copied by cl_wait into the caller's page-locked buffer (their count: header word 6)
    bool pairs_defer = false, pairs_copy_pending = false;      // cl_set_pairs_defer / cl_pairs_sync
    void* mask_out = nullptr;         // cl_cluster_rowmask_async: one bit per row + the labels of the set rows in row order, made on the device from the
    long long mask_cap = 0;           // row-aligned labels (k_rowmask_count / k_rowmask_write) in the slot's `pairs` buffer, copied out by cl_wait like the pairs
    bool l4_make_base = false;        // level 4: the run makes the words of its eps -- K2 on the base layout, launched behind the band query
    bool l4_cut = false;              // level 4: the run has a cut (the per-strip tables of k_cut_strips apply)
    bool l4_band = false;             // ... and re-uses counts under another cut (the band's words are fresh: c->cnt, by run position)
    bool fuse_chains = true;          // the chains of a run are made inside k_union_c (false: k_chain_c, a kernel of its own -- developer A/B)
    bool l_sup_dirty = false;         // k_classify's superblock sums may be non-zero (a run that failed between k_classify and k_chain_c)
    GridParams dbg_g{}; int dbg_nm = 0;   // the last rotated run's grid and size (developer build: kernels timed on their own)
    int traversal = 4;                // cl_set_traversal: 0 = tile kernels over every PET (rounds 1-4), 1 = K3 on the core list,
                                      // 2 = + border rule on the walker list, 3 = + labels / table / statistics from the lists,
                                      // 4 = + the lists built from the base layout (no copy of the layout for a run that re-uses counts)
    bool last_k2_mode_make = false;   // the run being enqueued made the words itself (level 4 under a cut: on the base layout, then its band)
    int last_k2_mode = 0;             // 0 = full K2, 1 = words re-used as they are (same cut), 2 = remapped + K2 on the band
    std::vector<long long> dcum;      // dcum[k] = number of PETs with Y - X < k, k = 0 .. 65536 (empty: unknown)
    const int* k_total = nullptr;     // device: where the run left the number of ids handed out (null: rankscan[n])
    bool hdr_packed = false;          // the run's own kernels have written the slot header (no k_pack_header)
    DevBuf dhist;                     // device: number of PETs with Y - X == d, d = 0 .. 65535 (+ one slot for d < 0)
    long long n_neg = 0;              // PETs with Y - X < 0
    int run_m = 0;                    // PETs that enter DBSCAN in the run being enqueued (exact when run_m_exact, else n)
    bool run_m_exact = false;
    bool reuse_layout = true;
    // Result slots: two runs may be in flight (cl_cluster_async) -- the labels / table / header of
    // run k live in slot k & 1, so the D2H copy of run k (copy stream) overlaps the kernels of run k+1.
    struct Slot {
        DevBuf labels, table;
        bool pending = false;
        int n_strips = 0;
        hipEvent_t ev_done = nullptr, ev_copied = nullptr;
        hipEvent_t ev[10]{};          // profiling marks of the run that used this slot (8, 9: around the band query of a level-4 run)
        bool band_timed = false;      // ... which ran
        long long n_queried = 0;      // PETs the run's region query covered (0: none)
        int* h_hdr = nullptr;         // pinned: {K, overflow, M}
        cl_box* h_boxes = nullptr;    // pinned host copy of the cluster table
        size_t h_boxes_cap = 0;
        int32_t* labels_out = nullptr;
        DevBuf slab;                  // labels in sorted order (rotated variants)
        DevBuf pairs;                 // cl_cluster_pairs_async: (row, label) of the labelled PETs on the device (copied out by cl_wait: their number is
        int2* pairs_host = nullptr;   //   only known when the run has completed)
        long long pairs_host_cap = 0; //   the caller's capacity (pairs): cl_wait refuses a run that labelled more
        void* mask_host = nullptr;    // cl_cluster_rowmask_async: the caller's buffer (mask words, then labels) and its capacity in labels
        long long mask_host_cap = 0;
        bool exported = true;         // the table rows were stored to h_boxes
        bool step_valid = false;      // the run carried the sweep-step tail (classification, candidate append, distance summary)
        bool host_written = false;    // ... and its last kernel stored header + step output in pinned host memory itself
        bool wait_done = false;       // cl_wait waits for ev_done (nothing went through the copy stream)
        long long fine_lo = -1;       // fine window of that tail's summary (-1 = none)
        int kmax = 0;                 // upper bound of the number of cluster ids of the run (host-known; the count itself is on the device)
        DevBuf d_step;                // device: {n_inter, n_self} + K7 partials + log histogram of that tail
        char* h_step = nullptr;       // pinned host copy
        bool rows_valid = false;      // `labels` (row order) was produced by the run
        bool sorted_src = false;      // the run left sorted (q, label) arrays for the distance statistics
        const int* k7_sv = nullptr;   // sorted q of the run (list form: q of every core and walker)
        const int* k7_lcnt = nullptr; // list form (k_lists.hip): device {cores, walkers} -- slab / k7_sv hold one entry per core and walker
        int k7_v0 = 0;                // d = q + k7_v0
    } slot[2];
    bool device_labels = true;        // produce row-order device labels even without a host destination (cl_set_device_labels)
    bool export_table = true;         // copy the cluster table to pinned host memory at the end of a run (cl_set_table_export)
    int pending_step = -1;            // >= 0: the run being enqueued is step `pending_step` of a sweep (cl_cluster_step_async)
    int pending_cut = 0;
    long long pending_fine_lo = -1;   // >= 0: the step's summary also histograms the self group's [fine_lo, fine_lo + 2048) exactly
    DevBuf cand_box, cand_step, cand_keep, cand_out;   // K10: candidate loops of the running sweep
    long long cand_n = 0, cand_cap = 0;
    DevBuf hdr;                       // device result headers, 16 ints per slot
    DevBuf k7_cls, k7_parts;          // K7: class per cluster id, per-workgroup partials
    DevBuf sig_tx, sig_ty, sig_tmp, sig_sorttmp, sig_m, sig_win, sig_out;   // K8: sorted PET tables, windows, counts
    bool sig_ready = false; int sig_cut = 0;
    bool k7_classified = false;       // k7_cls matches the last completed run
    hipStream_t copy_stream = nullptr, aux_stream = nullptr;
    int copy_mode = 0;                // 0: the D2H copies of a run go through copy_stream, 1: they are issued in `stream` (caller's stream),
                                      // 2: through the copy stream that cl_stream_create made for `stream` (shared by its handles)
    hipStream_t shared_copy = nullptr;
    int enq = 0, deq = 0;             // runs enqueued / completed
    int cur = 0;                      // slot of the run being enqueued
    // last completed result
    int last_slot = -1;
    int last_K = 0;                   // max_label + 1
    bool have_result = false;
    // profiling
    bool profiling = false;
    cl_timing timing{};
    bool ev_ready = false;
    float ev_bracket_ms = 0.f;        // event bracket around an empty kernel (calibration, see cl_timing)
};


#define LAUNCH(kernel, nthreads, ...) \
    hipLaunchKernelGGL(kernel, dim3(nblocks(nthreads)), dim3(TPB), 0, c->stream, __VA_ARGS__)

// cloops_hip.hip
int ensure_workspace(cl_chrom* c, int S);
int ensure_events(cl_chrom* c);
void ev_record(cl_chrom* c, int k);
int ensure_cand_capacity(cl_chrom* c, long long need);
Table make_table_slot(cl_chrom* c, int slot);
Table make_table(cl_chrom* c);
const int* k7_hist_for(cl_chrom* c, int cut);
int finish_enqueue(cl_chrom* c, int n_strips, const int* d_M, int32_t* labels_out);
// k_block.hip, k_weighted.hip
int run_block(cl_chrom* c, int eps, int minPts, int cut, int32_t* labels_out);
int run_weighted(cl_chrom* c, int eps, int minPts, int wx, int wy, int32_t* labels_out);

// kernels of cloops_hip.hip that k_block.hip / k_weighted.hip launch
__global__ void k_init_table(Table t, const int* __restrict__ rankscan, int n);
__global__ void k_init_arrays(int n, int* __restrict__ parent, int* __restrict__ compkey, int* __restrict__ ncore,
                              int* __restrict__ bsize, int* __restrict__ usize, int* __restrict__ cellfirst,
                              int* __restrict__ flag, int* __restrict__ state, int* __restrict__ counters);
__global__ void k_strip_table(const u64* __restrict__ skeys, int n, int S, int shift, int* __restrict__ strip_start);
__global__ void k_root_labels(GridParams g, const int* __restrict__ strip_start, const int* __restrict__ root,
                              const int* __restrict__ compkey, const int* __restrict__ ncore, const int* __restrict__ bsize,
                              const int* __restrict__ state, const int* __restrict__ rankscan, int* __restrict__ rlabel);
__global__ void k_rank_flags(GridParams g, const int* __restrict__ strip_start, const int* __restrict__ root,
                             const int* __restrict__ compkey, const int* __restrict__ state, int* __restrict__ flag);
__global__ void __launch_bounds__(BIGTPB)
k_flatten(GridParams g, const int* __restrict__ strip_start, const int* __restrict__ cnt,
          const int* __restrict__ chainid, int* parent, const u32* __restrict__ srow,
          const int* __restrict__ head, const int* __restrict__ cellfirst,
          int* __restrict__ root, int* __restrict__ compkey, int* __restrict__ ncore,
          int* __restrict__ rootlist , int* __restrict__ counters);

// ---- variant 2 release rule: a border point's adjacent components in ascending key order (cloops_hip.hip) ----
struct Rec { int pt; int r[4]; };
#define OWNER_CONTESTED 0x40000000
__device__ __forceinline__ int owner_root(int o) { return o < 0 ? -1 : (o & (OWNER_CONTESTED - 1)); }

// k_lists.hip: the list form of K3 / K4 / K5 (host side; every function enqueues on c->stream)
struct ListRun {                                        // device views of the run's lists (valid after lists_build)
    int npos;                                           // positions of the layout the lists index (the run's layout: PETs of the run; the base layout: all rows)
    const int* pstrip;                                  // ... and its strip table
    const int* lcnt;                                    // {C, W}: cores, walkers
    const unsigned long long* cmask; const unsigned long long* wmask;
    const int* cgrank; const int* wgrank;
    int2* cpair; int* cpos; int* ckey; int2* wpair; int* wpos; int* wenc; int* cstrip;
};
int lists_build(cl_chrom* c, const GridParams& g, int nm, ListRun* out);
int lists_build_base(cl_chrom* c, const GridParams& g, int nm, ListRun* out);      // level 4: from the base layout + the cut's per-strip tables
int lists_base_keys(cl_chrom* c, const GridParams& g);
int lists_words_to_base(cl_chrom* c, const GridParams& g);                         // level 4, first run of an eps under a cut: its words to base positions                             // variant 2: cell minima of the base layout (once per eps)
int lists_union_flatten(cl_chrom* c, const GridParams& g, int nm, const ListRun& L);
int lists_scatter_root(cl_chrom* c, int nm, const ListRun& L);
int lists_border(cl_chrom* c, const GridParams& g, int nm, const ListRun& L);
int lists_emit_records(cl_chrom* c, const GridParams& g, int nm, const ListRun& L);
int lists_scatter_owner(cl_chrom* c, int nm, const ListRun& L);
int lists_final(cl_chrom* c, const GridParams& g, int nm, const ListRun& L, bool rows, int* pair_count);
int lists_rowmask(cl_chrom* c, int* total);

// kernels of k_sweep.hip that the step tail (finish_enqueue) launches
__global__ void __launch_bounds__(256)
k_step_classify_count(const int* __restrict__ dK, Table t, signed char* __restrict__ cls, int* __restrict__ bcount, int nb,
                      unsigned long long* __restrict__ zero, int nzero);
__global__ void __launch_bounds__(256)
k_cand_append(const int* __restrict__ dK, const signed char* __restrict__ cls, Table t, const int* __restrict__ boff ,
              const int* __restrict__ bcount, int base, int step, int cap, int4* __restrict__ cbox, int* __restrict__ cstep);
__global__ void __launch_bounds__(TPB)
k7_summary(K7Src s, int cut, const signed char* __restrict__ cls, K7Part* __restrict__ parts, unsigned long long* __restrict__ loghist,
           unsigned fine_lo, unsigned long long* __restrict__ fine );
__global__ void __launch_bounds__(256)
k7_reduce_parts(const K7Part* __restrict__ parts, int nparts, K7Part* out, const int* __restrict__ bcount , int nb,
                long long* totals, const volatile unsigned long long* dev_step , int step_words,
                unsigned long long* __restrict__ host_step, const int* __restrict__ dev_hdr, int* __restrict__ host_hdr);

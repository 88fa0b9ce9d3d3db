// k_weighted.hip -- K9 of libcloops_hip.so: variant 1 under an axis-weighted metric (scripts/callStripes:37-72), kernels and driver.
#include "cl_chrom.h"

// ==========================================================================================
// K9: variant 1 under an axis-weighted city-block metric  wx*|dX| + wy*|dY| <= eps
// ==========================================================================================
// scripts/callStripes:37-72 (singleStripDBSCAN) multiplies the X or the Y column by `ext` (50) and runs
// cDBSCAN (variant 1) on the scaled matrix.  Scaled coordinates reach 1.25e10, so the rotated pair
// U = wx*X + wy*Y (strip coordinate), W = wy*Y - wx*X (in-strip coordinate) is 64-bit here and the sort key
// is  strip << qbits | (W - W0)  (<= 64 bits; U rides in a separate array, gathered after the sort).
// This second caller is not a throughput path: the kernels are the plain global-memory form of rule R1
// (searches on the sorted keys, one thread per PET), sharing sort, strip table, union-find, flatten,
// ranks and the cluster table with the main path.  Results: the ids of cDBSCAN(mat * [1, wx, wy], eps, minPts).
struct G64 { int eps, minPts, S, qbits, wx, wy; long long U0, W0; };

__global__ void k64_keys(const int* __restrict__ X, const int* __restrict__ Y, int n, G64 g, u64* __restrict__ keys, u32* __restrict__ vals)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const long long x = X[r], y = Y[r];
    const long long U = g.wx * x + g.wy * y - g.U0, W = g.wy * y - g.wx * x - g.W0;
    keys[r] = ((u64)((unsigned long long)U / (unsigned)g.eps) << g.qbits) | (u64)W;
    vals[r] = (u32)r;
}
__global__ void k64_p(int n, G64 g, const int* __restrict__ X, const int* __restrict__ Y, const u32* __restrict__ srow,
                      long long* __restrict__ p64)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 r = srow[i];
    p64[i] = (long long)g.wx * X[r] + (long long)g.wy * Y[r] - g.U0;
}
__device__ __forceinline__ int lb_keys(const u64* __restrict__ k, int lo, int hi, u64 target)
{
    while (lo < hi) { const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1); if (k[mid] < target) lo = mid + 1; else hi = mid; }
    return lo;
}
// window of strip t = [sb, se) with q in [qi - eps, qi + eps]  ->  [*w0, *w1)
__device__ __forceinline__ void win64(const G64& g, const u64* __restrict__ k, int t, int sb, int se, long long qi, int* w0, int* w1)
{
    const u64 base = (u64)(u32)t << g.qbits, qmask = (1ull << g.qbits) - 1ull;
    const long long lo = qi - g.eps, hi = qi + g.eps;
    *w0 = lb_keys(k, sb, se, base + (u64)(lo < 0 ? 0 : lo));
    *w1 = lb_keys(k, *w0, se, base + ((u64)hi > qmask ? qmask : (u64)hi) + 1ull);
}
__global__ void k64_count(int n, G64 g, const u64* __restrict__ k, const long long* __restrict__ p64,
                          const int* __restrict__ strip_start, int* __restrict__ cnt)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 qmask = (1ull << g.qbits) - 1ull;
    const int s = (int)(k[i] >> g.qbits);
    const long long qi = (long long)(k[i] & qmask), pi = p64[i];
    int w0, w1;
    win64(g, k, s, strip_start[s], strip_start[s + 1], qi, &w0, &w1);
    int c = w1 - w0;                                     // own strip: |dU| < eps is implied
    for (int d = -1; d <= 1 && c < g.minPts; d += 2) {
        const int t = s + d;
        if (t < 0 || t >= g.S) continue;
        win64(g, k, t, strip_start[t], strip_start[t + 1], qi, &w0, &w1);
        for (int j = w0; j < w1 && c < g.minPts; ++j) {
            const long long dp = p64[j] - pi;
            c += ((dp < 0 ? -dp : dp) <= g.eps) ? 1 : 0;
        }
    }
    cnt[i] = c;                                          // saturated at minPts
}
__global__ void k64_union(int n, G64 g, const u64* __restrict__ k, const long long* __restrict__ p64,
                          const int* __restrict__ strip_start, const int* __restrict__ cnt, int* parent)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || cnt[i] < g.minPts) return;
    const u64 qmask = (1ull << g.qbits) - 1ull;
    const int s = (int)(k[i] >> g.qbits);
    const long long qi = (long long)(k[i] & qmask), pi = p64[i];
    // own strip: the next core within eps (consecutive cores within eps chain the whole window together)
    const int e = strip_start[s + 1];
    for (int j = i + 1; j < e && (long long)(k[j] & qmask) - qi <= g.eps; ++j)
        if (cnt[j] >= g.minPts) { uf_unite(parent, i, j); break; }
    if (s > 0) {
        int w0, w1;
        win64(g, k, s - 1, strip_start[s - 1], strip_start[s], qi, &w0, &w1);
        int last = -1;
        for (int j = w0; j < w1; ++j) {
            if (cnt[j] < g.minPts) continue;
            const long long dp = p64[j] - pi;
            if ((dp < 0 ? -dp : dp) > g.eps) continue;
            const int rj = parent[j];                     // any ancestor: only used to skip repeated work
            if (rj != last) { uf_unite(parent, i, j); last = rj; }
        }
    }
}
// border points by rule R1 (cDBSCAN.py:172-173, 179-182): see k_border
__global__ void k64_border(int n, G64 g, const u64* __restrict__ k, const long long* __restrict__ p64,
                           const int* __restrict__ strip_start, const int* __restrict__ root, const int* __restrict__ compkey,
                           const int* __restrict__ ncore, const u32* __restrict__ srow, int* __restrict__ owner, int* __restrict__ bsize)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ri = root[i];
    if (ri >= 0) { owner[i] = ri; return; }
    const u64 qmask = (1ull << g.qbits) - 1ull;
    const int s = (int)(k[i] >> g.qbits);
    const long long qi = (long long)(k[i] & qmask), pi = p64[i];
    int bestk = INT_MAX, best = -1, tk = -1, tbest = -1, lastr = -1, lastk = 0;
    for (int d = -1; d <= 1; ++d) {
        const int t = s + d;
        if (t < 0 || t >= g.S) continue;
        int w0, w1;
        win64(g, k, t, strip_start[t], strip_start[t + 1], qi, &w0, &w1);
        for (int j = w0; j < w1; ++j) {
            const int r = root[j];
            if (r < 0) continue;
            if (d != 0) { const long long dp = p64[j] - pi; if ((dp < 0 ? -dp : dp) > g.eps) continue; }
            int kk;
            if (r == lastr) kk = lastk; else { kk = compkey[r]; lastr = r; lastk = kk; }
            if (kk < bestk) { bestk = kk; best = r; }
            if ((int)srow[j] == kk && kk > tk) { tk = kk; tbest = r; }     // j is its component's start point
        }
    }
    const int o = tbest >= 0 ? tbest : best;
    owner[i] = o;
    if (o >= 0 && ncore[o] < g.minPts) atomicAdd(&bsize[o], 1);
}
__global__ void __launch_bounds__(BIGTPB)
k64_final(int n, const int* __restrict__ X, const int* __restrict__ Y, const u32* __restrict__ srow, const int* __restrict__ owner,
          const int* __restrict__ rlabel, int* __restrict__ labels, Table t)
{
    __shared__ TableLds h;
    table_lds_init(h);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int lab = -1, x = 0, y = 0;
    if (i < n) {
        const int o = owner[i];
        if (o >= 0) lab = rlabel[o];
        const u32 r = srow[i];
        labels[r] = lab;
        x = X[r]; y = Y[r];
    }
    table_accumulate(t, h, lab, x, y);
    table_flush(t, h);
}


// ---- variant 1 under the weighted metric (K9) ------------------------------------------------
int run_weighted(cl_chrom* c, int eps, int minPts, int wx, int wy, int32_t* labels_out)
{
    int rc;
    const int n = (int)c->n;
    G64 g; g.eps = eps; g.minPts = minPts; g.wx = wx; g.wy = wy;
    // bounds of U = wx*X + wy*Y and W = wy*Y - wx*X from the upload statistics
    const long long umin = (long long)wx * c->st.xmin + (long long)wy * c->st.ymin, umax = (long long)wx * c->st.xmax + (long long)wy * c->st.ymax;
    const long long wmin = (long long)wy * c->st.ymin - (long long)wx * c->st.xmax, wmax = (long long)wy * c->st.ymax - (long long)wx * c->st.xmin;
    g.U0 = umin; g.W0 = wmin;
    const long long S = (umax - umin) / eps + 1;
    if (S > (1LL << 28)) return fail(CL_ERR_GRID, "eps too small for the scaled coordinate extent (strip table > 2^28 rows)");
    g.S = (int)S;
    int qbits = 1; while (qbits < 63 && ((wmax - wmin) >> qbits) != 0) ++qbits;
    g.qbits = qbits;
    const int strip_bits = std::max(1, bits_for((unsigned)g.S));
    if (qbits + strip_bits > 64) return fail(CL_ERR_GRID, "scaled coordinates need more than 64 key bits");
    if ((rc = ensure_workspace(c, g.S))) return rc;
    if ((rc = ensure_events(c))) return rc;
    int* strip = c->strip.as<int>();
    int* cnt = c->cnt.as<int>();
    int* counters = c->counters.as<int>();
    GridParams gi{};                                     // what the shared kernels read: S, minPts, variant
    gi.eps = eps; gi.minPts = minPts; gi.variant = CL_VARIANT_CDBSCAN1; gi.S = g.S;
    LAUNCH(k_init_arrays, n + 1, n, c->parent.as<int>(), c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(),
           c->usize.as<int>(), c->cellfirst.as<int>(), c->flag.as<int>(), c->state.as<int>(), counters);
    HIP_TRY(hipMemsetAsync(c->slot[c->cur].labels.p, 0xFF, (size_t)n * 4, c->stream));
    ev_record(c, 0);
    LAUNCH(k64_keys, n, c->d_x, c->d_y, n, g, c->keys_in.as<u64>(), c->vals_in.as<u32>());
    ev_record(c, 1);
    {
        size_t tmp_bytes = c->sort_tmp.bytes;
        hipError_t e = rocprim::radix_sort_pairs<SortConfig>(c->sort_tmp.p, tmp_bytes, c->keys_in.as<u64>(), c->keys_out.as<u64>(),
                                                 c->vals_in.as<u32>(), c->vals_out.as<u32>(), (size_t)n, 0, qbits + strip_bits, c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs", hipGetErrorString(e));
    }
    const u64* sk = c->keys_out.as<u64>();
    const u32* srow = c->vals_out.as<u32>();
    c->srow = c->vals_out.as<u32>();
    long long* p64 = c->keys_in.as<long long>();         // the unsorted keys are dead after the sort
    LAUNCH(k_strip_table, g.S + 2, sk, n, g.S, qbits, strip);
    LAUNCH(k64_p, n, n, g, c->d_x, c->d_y, srow, p64);
    ev_record(c, 2);
    LAUNCH(k64_count, n, n, g, sk, p64, strip, cnt);
    ev_record(c, 3);
    LAUNCH(k64_union, n, n, g, sk, p64, strip, cnt, c->parent.as<int>());
    hipLaunchKernelGGL(k_flatten, dim3(nblocks(n, BIGTPB * FLAT_PER)), dim3(BIGTPB), 0, c->stream, gi, strip, cnt, (const int*)nullptr, c->parent.as<int>(), srow,
                       c->head.as<int>(), c->cellfirst.as<int>(), c->root.as<int>(), c->compkey.as<int>(), c->ncore.as<int>(), (int*)nullptr, counters);
    ev_record(c, 4);
    LAUNCH(k64_border, n, n, g, sk, p64, strip, c->root.as<int>(), c->compkey.as<int>(), c->ncore.as<int>(), srow,
           c->owner.as<int>(), c->bsize.as<int>());
    ev_record(c, 5);
    LAUNCH(k_rank_flags, n, gi, strip, c->root.as<int>(), c->compkey.as<int>(), c->state.as<int>(), c->flag.as<int>());
    {
        size_t tb = c->scan_tmp.bytes;
        hipError_t e = rocprim::exclusive_scan(c->scan_tmp.p, tb, c->flag.as<int>(), c->rankscan.as<int>(), 0, (size_t)n + 1,
                                               rocprim::plus<int>(), c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "exclusive_scan", hipGetErrorString(e));
    }
    Table t = make_table(c);
    LAUNCH(k_init_table, n + 1, t, c->rankscan.as<int>(), n);
    LAUNCH(k_root_labels, n, gi, strip, c->root.as<int>(), c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(),
           c->state.as<int>(), c->rankscan.as<int>(), c->chainhead.as<int>());
    hipLaunchKernelGGL(k64_final, dim3(nblocks(n, BIGTPB)), dim3(BIGTPB), 0, c->stream, n, c->d_x, c->d_y, srow, c->owner.as<int>(),
                       c->chainhead.as<int>(), c->slot[c->cur].labels.as<int>(), t);
    HIP_TRY(hipGetLastError());
    { cl_chrom::Slot& sl = c->slot[c->cur]; sl.rows_valid = true; sl.sorted_src = false; }
    return finish_enqueue(c, g.S + 2, strip + g.S, labels_out);
}

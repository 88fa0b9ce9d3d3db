// cloops_hip.hip -- MI355X (gfx950 / CDNA4) implementation of the cDBSCAN hot path of
// YaqiangCao/cLoops behind the C ABI of include/cloops_hip.h.
//
// Not a port: the reference (cLoops/cDBSCAN.py, cLoops/cDBSCAN2.py, cLoops/blockDBSCAN.py)
// is sequential, visit-order dependent Python over dicts.  This file implements the
// ORDER-FREE closed forms of those three algorithms
// as data-parallel integer kernels, and reproduces the reference's cluster ids bit-exactly.
//
// Data layout (variants 1 and 2).  The city-block ball |dX|+|dY| <= eps is the square
// max(|da|,|dv|) <= eps in the rotated coordinates a = Y-X, v = X+Y (cDBSCAN2.py:67-68).
// PETs are radix-sorted by the 64-bit key (strip(a) << 32 | v) where strip(a) = a / eps:
// a *strip* is an eps-wide band of `a`, internally ordered by v.  For a query point in
// strip s, every neighbour lies in strips s-1, s, s+1, and inside each strip in one
// CONTIGUOUS v-window [v-eps, v+eps] -- three coalesced candidate ranges per point instead
// of nine cells; inside the own strip the `a` test is implied, so its contribution to the
// neighbour count is a pure index difference.  A dense table strip_start[] (one int per
// strip) replaces every hash/dict lookup of the reference.
//
// Kernels (names as in DESIGN.md):
//   K0 k_make_keys      cut filter (pipe.py:59-63) + sort keys
//   K1 rocPRIM radix sort, k_gather_sorted, k_strip_table
//   K2 k_region_count   neighbour counts -> core flags           (the roofline kernel)
//   K3 k_union_cores    lock-free union-find over core-core edges; k_flatten
//   K4 k_border         border ownership per variant rule; v2 release fix-up
//   K5 k_rank_flags / scan / k_final_labels   reference cluster ids + cluster table
//   K6 block variant    cell table, links, cell-level union (blockDBSCAN.py)
#include <map>
#include <mutex>
#include "cl_chrom.h"
#include "cl_band.h"

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
thread_local std::string g_err;



extern "C" const char* cl_last_error(void) { return g_err.c_str(); }
extern "C" int cl_version(void) { return CL_VERSION_NUM; }

extern "C" void* cl_host_alloc(int64_t bytes)
{
    void* p = nullptr;
    if (bytes <= 0) return nullptr;
    if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) { fail(CL_ERR_HIP, "hipHostMalloc"); return nullptr; }
    return p;
}
extern "C" void cl_host_free(void* p) { if (p) (void)hipHostFree(p); }

// A stream made here gets a COPY stream of its own the first time one of its handles has labels to send to the host: the handles
// that share the stream send their device-to-host copies through it (behind an event of the run), so that a run's labels cross
// PCIe while the next handle's kernels execute.  One copy stream per shared compute stream -- not one per handle: dozens of
// copy streams waiting on events of other hardware queues is what blocked those queues head of line --
// and made late: the HIP runtime deals hardware queues to streams in the order they are created, a sweep (which copies
// nothing) keeps the mapping of its three compute streams.
static std::mutex g_pair_mu;
static std::map<hipStream_t, hipStream_t> g_copy_of;
extern "C" void* cl_stream_create(int device)
{
    hipStream_t s = nullptr;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { fail(CL_ERR_HIP, "hipStreamCreate"); return nullptr; }
    std::lock_guard<std::mutex> lk(g_pair_mu);
    g_copy_of[s] = nullptr;
    return (void*)s;
}
extern "C" void cl_stream_destroy(void* stream)
{
    if (!stream) return;
    hipStream_t cs = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pair_mu);
        auto it = g_copy_of.find((hipStream_t)stream);
        if (it != g_copy_of.end()) { cs = it->second; g_copy_of.erase(it); }
    }
    if (cs) (void)hipStreamDestroy(cs);
    (void)hipStreamDestroy((hipStream_t)stream);
}
static bool library_made_stream(hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_pair_mu);
    return g_copy_of.find(s) != g_copy_of.end();
}
// the copy stream of a library-made stream (made on first use), or null
static hipStream_t paired_copy_stream(hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_pair_mu);
    auto it = g_copy_of.find(s);
    if (it == g_copy_of.end()) return nullptr;
    if (!it->second) {
        int prio_lo = 0, prio_hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
        hipStream_t cs = nullptr;
        if (hipStreamCreateWithPriority(&cs, hipStreamNonBlocking, prio_hi) != hipSuccess) return nullptr;
        it->second = cs;
    }
    return it->second;
}

extern "C" int cl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------
// chromosome statistics (once per upload)
// ------------------------------------------------------------------------------------------

__global__ void k_stats(const int* __restrict__ X, const int* __restrict__ Y, long long n, Stats* out)
{
    int amin = INT_MAX, amax = INT_MIN, vmin = INT_MAX, vmax = INT_MIN;
    int xmin = INT_MAX, xmax = INT_MIN, ymin = INT_MAX, ymax = INT_MIN;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        int x = X[i], y = Y[i];
        xmin = min(xmin, x); xmax = max(xmax, x); ymin = min(ymin, y); ymax = max(ymax, y);
        // |x|,|y| < 2^29 is validated from xmin..ymax afterwards; clamp here to stay defined
        long long a = (long long)y - x, v = (long long)y + x;
        int ai = (int)max(min(a, (long long)INT_MAX), (long long)INT_MIN);
        int vi = (int)max(min(v, (long long)INT_MAX), (long long)INT_MIN);
        amin = min(amin, ai); amax = max(amax, ai); vmin = min(vmin, vi); vmax = max(vmax, vi);
    }
    // wave reduction, workgroup reduction in LDS, then one set of atomics per workgroup
    for (int off = 32; off > 0; off >>= 1) {
        amin = min(amin, __shfl_down(amin, off)); amax = max(amax, __shfl_down(amax, off));
        vmin = min(vmin, __shfl_down(vmin, off)); vmax = max(vmax, __shfl_down(vmax, off));
        xmin = min(xmin, __shfl_down(xmin, off)); xmax = max(xmax, __shfl_down(xmax, off));
        ymin = min(ymin, __shfl_down(ymin, off)); ymax = max(ymax, __shfl_down(ymax, off));
    }
    __shared__ int red[TPB / 64][8];
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[wv][0] = amin; red[wv][1] = amax; red[wv][2] = vmin; red[wv][3] = vmax;
        red[wv][4] = xmin; red[wv][5] = xmax; red[wv][6] = ymin; red[wv][7] = ymax;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const bool is_min = (threadIdx.x & 1) == 0;
        int v = red[0][threadIdx.x];
        for (int w = 1; w < TPB / 64; ++w) v = is_min ? min(v, red[w][threadIdx.x]) : max(v, red[w][threadIdx.x]);
        int* dst = &out->amin + threadIdx.x;          // Stats is 8 consecutive ints in this order
        if (is_min) atomicMin(dst, v); else atomicMax(dst, v);
    }
}

// histogram of the distances d = Y - X below 65536 (one-off per upload): the host keeps its running sum, so that the number of
// PETs that pass a cut (pipe.py:59-62) is known when a run is ENQUEUED -- grids and scans are then sized by M, not by n
#define DCUM_BINS 65537
#define DH_LOCAL 8192            // distances below it are counted in LDS first: self-ligation PETs (a third of a library, d of a few hundred
                                 // to a few thousand bp) put millions of atomics on a few thousand global addresses (3.1 ms per 16 M PETs; now 0.3)
__global__ void __launch_bounds__(TPB)
k_dhist(const int* __restrict__ X, const int* __restrict__ Y, long long n, int* __restrict__ hist)
{
    __shared__ int lh[DH_LOCAL];
    for (int k = threadIdx.x; k < DH_LOCAL; k += blockDim.x) lh[k] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long d = (long long)Y[i] - X[i];
        // distances >= 65536 need no bin (the running sum stops there); counting them would put half of the PETs on one address
        if (d >= 0 && d < DH_LOCAL) atomicAdd(&lh[(int)d], 1);
        else if (d < DCUM_BINS - 1) atomicAdd(&hist[d < 0 ? DCUM_BINS : (int)d], 1);      // slot DCUM_BINS: d < 0 (X > Y rows)
    }
    __syncthreads();
    for (int k = threadIdx.x; k < DH_LOCAL; k += blockDim.x) { const int v = lh[k]; if (v) atomicAdd(&hist[k], v); }
}

// ------------------------------------------------------------------------------------------
// K0: keys
// ------------------------------------------------------------------------------------------
__global__ void k_make_keys(const int* __restrict__ X, const int* __restrict__ Y, int n, GridParams g,
                            u64* __restrict__ keys, u32* __restrict__ vals)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    int x = X[r], y = Y[r];
    int a = y - x;
    bool valid = (g.cut <= 0) || (a >= g.cut);            // pipe.py:59-62  d >= cut
    int v = x + y;
    int prel = (g.swap ? v : a) - g.A0;                   // strip coordinate
    u32 qrel = (u32)((g.swap ? a : v) - g.V0);            // in-strip coordinate
    const int sabs = div_eps(g, prel);
    const u32 rem = (u32)(prel - sabs * g.eps);           // p mod eps
    const int sh = g.qbits + g.rbits;
    u64 key = valid ? (((u64)(u32)(sabs - g.s0) << sh) | ((u64)qrel << g.rbits) | rem) : ((u64)(u32)g.S << sh);
    keys[r] = key;
    vals[r] = (u32)r;
}

// K1b: sorted coordinates, decoded from the sorted keys (coalesced; no gather through row ids)
__global__ void k_decode_sorted(int n, GridParams g, const u64* __restrict__ skeys,
                                int* __restrict__ sv, int* __restrict__ sa, int* __restrict__ tile_s0,
                                const u32* __restrict__ rows_in, u32* __restrict__ rows_out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 k = skeys[i];
    if (rows_out) rows_out[i] = rows_in[i];
    const int sh = g.qbits + g.rbits;
    const int strip = (int)(k >> sh);
    if ((i & 255) == 0) tile_s0[i >> 8] = min(strip, g.S);       // strip of every 256th sorted PET (K2 stages its strip-table slice from it)
    if (strip >= g.S) { sv[i] = INT_MAX; sa[i] = g.S << g.rbits; return; }
    sv[i] = (int)((k >> g.rbits) & ((1ull << g.qbits) - 1ull));
    sa[i] = (strip << g.rbits) | (int)(k & ((1ull << g.rbits) - 1ull));
}

// ------------------------------------------------------------------------------------------
// K1 through the q index.  The sorted order of a run is (strip, q) with ties in input-row order, and only the strip
// depends on eps.  A handle that sorts more than once (a sweep: one layout per eps) therefore keeps its rows sorted by q
// ONCE -- qb_key[i] = q, qb_val[i] = strip coordinate << 32 | row, stable, i.e. ties in row order -- and a layout for
// some eps is a STABLE sort of that sequence by the strip bits alone: 2 radix passes (9-bit digits, <= 2^18 strips)
// instead of 5 over (strip, q), the same permutation bit for bit.  The index costs 4 passes once and 12 B/PET.
// ------------------------------------------------------------------------------------------
__global__ void k_make_qkeys(const int* __restrict__ X, const int* __restrict__ Y, int n, GridParams g,
                             u32* __restrict__ keyq, u64* __restrict__ val)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int x = X[r], y = Y[r];
    const int a = y - x, v = x + y;
    const u32 prel = (u32)((g.swap ? v : a) - g.A0);      // strip coordinate, >= 0
    keyq[r] = (u32)((g.swap ? a : v) - g.V0);             // in-strip coordinate, >= 0
    val[r] = ((u64)prel << 32) | (u32)r;
}
// keys of one layout from the q index: key = sp (strip << rbits | p mod eps; rows removed by the cut: strip S),
// value = q << 32 | row
__global__ void k_make_spkeys(int n, GridParams g, const u32* __restrict__ keyq, const u64* __restrict__ valq,
                              u32* __restrict__ key, u64* __restrict__ val)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 q = keyq[i];
    const u64 pv = valq[i];
    const int prel = (int)(pv >> 32);
    const int a = (g.swap ? (int)q + g.V0 : prel + g.A0);  // Y - X
    const bool valid = (g.cut <= 0) || (a >= g.cut);      // pipe.py:59-62  d >= cut
    const int sabs = div_eps(g, prel);
    const u32 rem = (u32)(prel - sabs * g.eps);
    key[i] = valid ? (((u32)(sabs - g.s0) << g.rbits) | rem) : ((u32)g.S << g.rbits);
    val[i] = ((u64)q << 32) | (u32)pv;
}
__global__ void k_decode_sp(int n, GridParams g, const u32* __restrict__ skey, const u64* __restrict__ sval,
                            int* __restrict__ sv, int* __restrict__ sa, int* __restrict__ tile_s0, u32* __restrict__ rows_out)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u32 k = skey[i];
    const u64 v = sval[i];
    rows_out[i] = (u32)v;
    const int strip = (int)(k >> g.rbits);
    if ((i & 255) == 0) tile_s0[i >> 8] = min(strip, g.S);
    if (strip >= g.S) { sv[i] = INT_MAX; sa[i] = g.S << g.rbits; return; }
    sv[i] = (int)(v >> 32);
    sa[i] = (int)k;
}
// The same two kernels as ITERATORS of the sort (a layout of all rows: no cut in the keys): the first radix pass makes its
// keys and values from the q index while it reads it, the last pass writes the sorted sp straight into the layout's sp array and
// splits its 64-bit values into the q and row arrays -- 24 B/PET of k_make_spkeys and 24 B/PET of k_decode_sp less per layout.
struct SpKeyOfIndex {
    const u32* keyq; const u64* valq; GridParams g;
    __device__ __forceinline__ u32 operator()(int i) const
    {
        const int prel = (int)(valq[i] >> 32);
        const int sabs = div_eps(g, prel);
        return ((u32)(sabs - g.s0) << g.rbits) | (u32)(prel - sabs * g.eps);
    }
};
struct SpValOfIndex {
    const u32* keyq; const u64* valq;
    __device__ __forceinline__ u64 operator()(int i) const { return ((u64)keyq[i] << 32) | (u32)valq[i]; }
};
struct SplitRef {
    int* q; u32* row;
    __device__ __forceinline__ const SplitRef& operator=(u64 v) const { *q = (int)(v >> 32); *row = (u32)v; return *this; }
    __device__ __forceinline__ operator u64() const { return ((u64)(u32)*q << 32) | *row; }      // (rocPRIM's small-input merge sort reads its output back)
    __device__ __forceinline__ const SplitRef& operator=(const SplitRef& o) const { *q = *o.q; *row = *o.row; return *this; }   // values, not pointers
};
struct SplitOut {                                       // output iterator: value -> (q array, row array)
    using iterator_category = std::random_access_iterator_tag;
    using value_type = u64;
    using difference_type = std::ptrdiff_t;
    using pointer = void;
    using reference = SplitRef;
    int* q; u32* row;
    __host__ __device__ __forceinline__ SplitRef operator*() const { return SplitRef{q, row}; }
    __host__ __device__ __forceinline__ SplitRef operator[](difference_type i) const { return SplitRef{q + i, row + i}; }
    __host__ __device__ __forceinline__ SplitOut operator+(difference_type d) const { return SplitOut{q + d, row + d}; }
    __host__ __device__ __forceinline__ SplitOut operator-(difference_type d) const { return SplitOut{q - d, row - d}; }
    __host__ __device__ __forceinline__ difference_type operator-(const SplitOut& o) const { return q - o.q; }
    __host__ __device__ __forceinline__ SplitOut& operator+=(difference_type d) { q += d; row += d; return *this; }
    __host__ __device__ __forceinline__ SplitOut& operator-=(difference_type d) { q -= d; row -= d; return *this; }
    __host__ __device__ __forceinline__ SplitOut& operator++() { ++q; ++row; return *this; }
    __host__ __device__ __forceinline__ SplitOut operator++(int) { SplitOut t = *this; ++q; ++row; return t; }
    __host__ __device__ __forceinline__ bool operator==(const SplitOut& o) const { return q == o.q; }
    __host__ __device__ __forceinline__ bool operator!=(const SplitOut& o) const { return q != o.q; }
    __host__ __device__ __forceinline__ bool operator<(const SplitOut& o) const { return q < o.q; }
};

// ------------------------------------------------------------------------------------------
// A layout from the FINE layout (cl_set_eps_list).  The eps values of a sweep usually share a large divisor w (Hi-C mode 3:
// 5000 / 7500 / 10000 -> 2500), and the strips of every eps start at the same origin: a strip of width eps = k w is k
// consecutive strips of width w.  With the rows sorted ONCE by (fine strip, q) -- the same 2-pass strip sort of the q index --
// the layout of an eps is no sort at all: a strip's PETs are one contiguous segment of the fine layout, made of k runs that are
// each in (q, row) order, and every PET finds its place inside the segment by k - 1 bisections of the other runs (its
// neighbours in the wave search the same few hundred entries, staged in LDS).  The same order as the sort's except among equal
// distances of one strip (by run instead of by input row: no kernel reads that order).  One pass of 12 B/PET in, 12 B/PET out instead of two
// radix passes + histogram + the sort's own resets.
// ------------------------------------------------------------------------------------------
__global__ void k_strips_from_fine(int S, int s0, int k, int F, int f0, const int* __restrict__ fstrip, int n, int* __restrict__ strip_start)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > S + 1) return;
    const long long f = (long long)k * ((long long)s + s0) - f0;
    strip_start[s] = s >= S ? n : fstrip[(int)std::min<long long>(std::max<long long>(f, 0), F)];
}
#define LFF_T 1024               // PETs of the fine layout per workgroup
#define LFF_CAP 4096             // staged entries: the strips a tile's PETs belong to, whole (a tile + one strip on either side)
#define LFF_NST 127              // rows of the fine strip table staged per tile
#define LFF_PAD 512              // keys of all ones behind them: unclamped probes of searches up to 9 steps deep
__global__ void __launch_bounds__(256)
k_layout_from_fine(int n, GridParams g, GridParams gf, int k, unsigned kmagic, int qb, const int* __restrict__ fq, const int* __restrict__ fsp, const u32* __restrict__ frow,
                   const int* __restrict__ fstrip, int* __restrict__ dq, int* __restrict__ dsp, u32* __restrict__ drow, int* __restrict__ dtile)
{
    __shared__ int lq[LFF_CAP + LFF_PAD];
    const int t0 = blockIdx.x * LFF_T, t1 = min(n, t0 + LFF_T);
    if (t0 >= n) return;
    // the tile's own PETs first: their loads depend on nothing but t0 and are in flight under the chain of dependent loads below
    // (first / last strip -> table rows -> the staged runs)
    constexpr int E = LFF_T / 256;
    int q[E], spf[E], fbase[E], r[E], dst[E];
    u32 row[E];
    bool in[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = t0 + e * 256 + (int)threadIdx.x;
        in[e] = i < t1;
        const int ic = in[e] ? i : t0;
        q[e] = fq[ic]; row[e] = frow[ic]; spf[e] = fsp[ic];
    }
    // the strips (of width eps) the tile's first and last PET belong to are contiguous in the fine layout: [a, b) holds every run
    // any PET of the tile has to be ranked in
    const int sb_first = (((fsp[t0] >> gf.rbits) + gf.s0) / k) * k - gf.s0, sb_last = (((fsp[t1 - 1] >> gf.rbits) + gf.s0) / k) * k - gf.s0;
    const int f_first = max(sb_first, 0), f_end = min(sb_last + k, gf.S);
    const int a = fstrip[f_first], b = fstrip[f_end];
    const bool staged = b - a <= LFF_CAP;                 // (a pile-up strip longer than the staging area: searches in global memory)
    // Sorted-key form (round 6): the staged runs as ONE sorted array of 32-bit keys  run << qb | q  (run = fine strip - first staged
    // one; q < 2^qb) -- "entries of run t below x" is then a plain lower bound of the key (t, x) from the run's start: a probe is one
    // LDS read at an immediate offset, ONE compare and a select, no index clamps (behind the staged entries: LFF_PAD keys of all ones).
    const bool keyed = staged && qb > 0 && (f_end - f_first) < (1 << (32 - qb)) - 1;
    // the rows of the fine strip table the tile needs, staged as well (a PET reads 2 + 2 (k - 1) of them: dependent global loads otherwise)
    __shared__ int lst[LFF_NST + 1];
    const bool slice = f_end - f_first <= LFF_NST;
    if (slice) for (int j = threadIdx.x; j <= f_end - f_first; j += 256) lst[j] = fstrip[f_first + j];
    if (keyed) {
        // (eight loads of a thread in flight before the first LDS store: a rolled loop is a chain of round trips per tile)
        for (int j0 = 0; j0 < b - a; j0 += 8 * 256) {
            int vq[8], vs[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = j0 + u * 256 + (int)threadIdx.x; const bool ok = j < b - a; vq[u] = ok ? fq[a + j] : 0; vs[u] = ok ? fsp[a + j] : 0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = j0 + u * 256 + (int)threadIdx.x; if (j < b - a) lq[j] = (int)((unsigned)vq[u] | ((unsigned)((vs[u] >> gf.rbits) - f_first) << qb)); }
        }
        for (int j = threadIdx.x; j < LFF_PAD; j += 256) lq[b - a + j] = -1;
    } else if (staged) for (int j = threadIdx.x; j < b - a; j += 256) lq[j] = fq[a + j];
    __syncthreads();
    auto fst = [&](int f) { return slice ? lst[f - f_first] : fstrip[f]; };      // f in [f_first, f_end]
    // four PETs per thread, their searches side by side (every probe is a dependent round trip: four chains in flight)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int i = t0 + e * 256 + (int)threadIdx.x;
        const int f = spf[e] >> gf.rbits;                // fine strip: row of the fine table
        const int fabs = f + gf.s0;                      // (>= 0: strips count from the common origin A0)
        const int sabs = k == 1 ? fabs : (int)__umulhi((unsigned)fabs, kmagic);      // the strip of width eps: fabs / k (kmagic = ceil(2^32 / k): exact below 2^29; k = 1 has no 32-bit magic) ...
        r[e] = fabs - sabs * k;                          // ... and the PET's run inside it
        fbase[e] = sabs * k - gf.s0;                     // table row of the strip's first run (< 0: runs in front of the chromosome's first strip)
        dst[e] = in[e] ? fst(max(fbase[e], 0)) + (i - fst(f)) : 0;
    }
#ifdef CLOOPS_DEVEL
    const int kk_abl = (g.dbg2 & (1 << 12)) ? 0 : k;      // (ablation: no searches)
#else
    const int kk_abl = k;
#endif
    // Order inside a strip: q, ties by run, then by the place inside the run -- a PET's place = its place in its own run + the
    // entries with q' <= q of every run in front of its own + those with q' < q of every run behind (one compare per probe:
    // q' < q + 1 resp. q' < q).  Equal distances are not in input-row order as behind the sort; nothing reads that order.
    for (int rr = 0; rr < kk_abl; ++rr) {
        int lo[E], len[E], key[E], pos[E];
        int nmax = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int ff = fbase[e] + rr;
            const bool on = in[e] && rr != r[e] && ff >= 0 && ff < gf.S;
            lo[e] = on ? fst(ff) : 0; len[e] = on ? fst(ff + 1) - lo[e] : 0;
            key[e] = q[e] + (rr < r[e] ? 1 : 0);
            pos[e] = 0;
            nmax = max(nmax, len[e]);
        }
        nmax = wave_max_i(nmax);
        int nsteps = 0;
        while ((1 << nsteps) <= nmax) ++nsteps;          // (wave-uniform)
        if (keyed && (1 << nsteps) <= LFF_PAD) {
            // (positions in BYTES: ds_read_b32 at an immediate offset, v_cmp_lt_u32, v_cndmask, v_add per probe)
            unsigned tk[E];
            int p4[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const bool on = len[e] > 0;
                tk[e] = on ? ((unsigned)key[e] | ((unsigned)(fbase[e] + rr - f_first) << qb)) : 0u;      // (nothing is below key 0: a run that is not searched)
                p4[e] = on ? (lo[e] - a) * 4 : 0;
            }
            const char* lb = reinterpret_cast<const char*>(lq);
#define LFF_STEP(ST) _Pragma("unroll") for (int e = 0; e < E; ++e) { const unsigned v = *reinterpret_cast<const unsigned*>(lb + p4[e] + ((ST) - 1) * 4); p4[e] = v < tk[e] ? p4[e] + (ST) * 4 : p4[e]; }
            switch (nsteps) {
            case 9: LFF_STEP(256)
            case 8: LFF_STEP(128)
            case 7: LFF_STEP(64)
            case 6: LFF_STEP(32)
            case 5: LFF_STEP(16)
            case 4: LFF_STEP(8)
            case 3: LFF_STEP(4)
            case 2: LFF_STEP(2)
            case 1: LFF_STEP(1)
            default: break;
            }
#undef LFF_STEP
#pragma unroll
            for (int e = 0; e < E; ++e) pos[e] = len[e] > 0 ? (p4[e] >> 2) - (lo[e] - a) : 0;
        } else if (staged && !keyed) {
            for (int step = nsteps > 0 ? 1 << (nsteps - 1) : 0; step >= 1; step >>= 1) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int p = pos[e] + step - 1;
                    const int v = lq[lo[e] - a + min(p, max(len[e] - 1, 0))];
                    pos[e] = (p < len[e] && v < key[e]) ? pos[e] + step : pos[e];
                }
            }
        } else {
            for (int step = nsteps > 0 ? 1 << (nsteps - 1) : 0; step >= 1; step >>= 1) {
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int p = pos[e] + step - 1;
                    const int v = fq[lo[e] + min(p, max(len[e] - 1, 0))];
                    pos[e] = (p < len[e] && v < key[e]) ? pos[e] + step : pos[e];
                }
            }
        }
#pragma unroll
        for (int e = 0; e < E; ++e) dst[e] += pos[e];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (!in[e]) continue;
#ifdef CLOOPS_DEVEL
        if (g.dbg2 & (1 << 13)) { if (dst[e] == 0x7fffffff) dq[0] = 1; continue; }      // (ablation: no stores)
#endif
        const int sabs = k == 1 ? fbase[e] + gf.s0 : (int)__umulhi((unsigned)(fbase[e] + gf.s0), kmagic);
        dq[dst[e]] = q[e];
        dsp[dst[e]] = ((sabs - g.s0) << g.rbits) | (r[e] * gf.eps + (spf[e] & (gf.peps - 1)));
        drow[dst[e]] = row[e];
        if ((dst[e] & 255) == 0) dtile[dst[e] >> 8] = min(sabs - g.s0, g.S);
    }
}

// tile_s0 (optional): strip of every 256th sorted PET (what k_decode_sp leaves, for a layout that was not decoded)
__global__ void k_strip_table32(const u32* __restrict__ skeys, int n, int S, int shift, int* __restrict__ strip_start, int* __restrict__ tile_s0 = nullptr)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (tile_s0) for (int u = t; u < (n + 255) / 256; u += gridDim.x * blockDim.x) tile_s0[u] = min((int)(skeys[u * 256] >> shift), S);
    if (t > S + 1) return;
    if (t == S + 1) { strip_start[t] = n; return; }
    const u32 target = (u32)t << shift;
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (skeys[mid] < target) lo = mid + 1; else hi = mid;
    }
    strip_start[t] = lo;
}

// K1c: strip_start[t] = first sorted index whose strip >= t, t = 0..S+1
// (strip_start[S] = M = number of rows that entered DBSCAN, strip_start[S+1] = n)
__global__ void k_strip_table(const u64* __restrict__ skeys, int n, int S, int shift, int* __restrict__ strip_start)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > S + 1) return;
    if (t == S + 1) { strip_start[t] = n; return; }
    u64 target = (u64)(u32)t << shift;
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (skeys[mid] < target) lo = mid + 1; else hi = mid;
    }
    strip_start[t] = lo;
}

// ------------------------------------------------------------------------------------------
// K1 (hybrid): when every strip is short, only the STRIP bits are radix-sorted (2 passes instead of 5 on
// the bench workload) and the order inside a strip is finished here: a PET's place in its strip is the
// number of PETs of the strip with a smaller q (ties: the one that comes first, i.e. the smaller input
// row -- the passes are stable), counted on an LDS window.  The kernel also decodes (q, p) and moves
// the row ids, so it replaces k_decode_sorted as well.  "Every strip is short" is a property of the
// chromosome and eps, measured once per (layout, eps) over ALL rows (a cut only removes rows) and kept in
// the handle (strip_maxlen below) -- results are never cached, only this choice of algorithm.
// ------------------------------------------------------------------------------------------
#define HS_TPB 256
#define HS_HALO 256
#define HS_WIN (HS_TPB + 2 * HS_HALO)
#define HS_LMAX 256          // longest strip the in-strip ranking accepts (must be <= HS_HALO)
enum { CTR_MAXLEN = 48, CTR_M = 49 };

__global__ void k_strip_hist(const int* __restrict__ X, const int* __restrict__ Y, int n, GridParams g, int* __restrict__ hist)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int x = X[r], y = Y[r];
    const int prel = (g.swap ? x + y : y - x) - g.A0;
    atomicAdd(&hist[div_eps(g, prel) - g.s0], 1);
}
__global__ void k_max_int(const int* __restrict__ v, int n, int* __restrict__ out)
{
    __shared__ int red[TPB / 64];
    int m = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = max(m, v[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; ++w) m = max(m, red[w]);
        atomicMax(out, m);
    }
}

__global__ void __launch_bounds__(HS_TPB)
k_strip_sort(int n, GridParams g, const u64* __restrict__ keys, const u32* __restrict__ rows,
             const int* __restrict__ strip_start, int* __restrict__ sv, int* __restrict__ sa, u32* __restrict__ srow,
             int* __restrict__ tile_s0, int* __restrict__ counters)
{
    __shared__ u32 lq[HS_WIN];
    const int t0 = blockIdx.x * HS_TPB, base = t0 - HS_HALO;
    const u64 qmask = (1ull << g.qbits) - 1ull;
    u64 kk[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int gi = base + (int)threadIdx.x + u * HS_TPB;
        kk[u] = (gi >= 0 && gi < n) ? keys[gi] : 0ull;
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) lq[threadIdx.x + u * HS_TPB] = (u32)((kk[u] >> g.rbits) & qmask);
    const int i = t0 + threadIdx.x;
    const u32 row = i < n ? rows[i] : 0u;
    __syncthreads();
    if (i >= n) return;
    const u64 k = kk[1];                                   // slot 1 of the thread is its own PET (HS_HALO == HS_TPB)
    static_assert(HS_HALO == HS_TPB, "own PET = staging slot 1");
    const int strip = (int)(k >> (g.qbits + g.rbits));
    if (strip >= g.S) {                                                            // filtered rows keep their places
        sv[i] = INT_MAX; sa[i] = g.S << g.rbits; srow[i] = row;
        if ((i & 255) == 0) tile_s0[i >> 8] = g.S;
        return;
    }
    const int b = strip_start[strip], e = strip_start[strip + 1];
    if (e - b > HS_LMAX) { counters[CTR_OVERFLOW] = 4; return; }                  // cannot happen (strip_maxlen)
    const u32 qi = lq[i - base];
    int rank = 0;
    for (int j = b; j < e; ++j) {
        const u32 qj = lq[j - base];
        rank += (qj < qi || (qj == qi && j < i)) ? 1 : 0;
    }
    const int dst = b + rank;
    sv[dst] = (int)qi;
    sa[dst] = (strip << g.rbits) | (int)(k & ((1ull << g.rbits) - 1ull));
    srow[dst] = row;
    if ((dst & 255) == 0) tile_s0[dst >> 8] = strip;
}


// ------------------------------------------------------------------------------------------
// LDS tile framework shared by the traversal kernels K3 (chains, union) and K4 (border, records)
// ------------------------------------------------------------------------------------------
// Same tiling as K2: a workgroup owns 256 consecutive sorted PETs and stages them plus a halo
// as (q,p) pairs and one int of per-PET payload (neighbour count or component root).  Strip
// segments inside the staged range are walked in LDS; anything else falls back to global memory.
// Two shapes: NT = 256 PETs + 128 halo where strips are short (sparse data), NT = 1024 + 512 where strips hold hundreds
// of PETs (dense data at large eps: the three strips of a query must fit the window, or the walk falls back to global memory).
#define T_STEPS_LONG 11          // search depth for staged segments of 256 .. 2047 PETs
enum { CF_CORE = 1, CF_OPEN = 2, CF_LAST = 4 };      // k_chain_flags -> k_chain_parent

struct Tile {
    LdsPairs w;         // (q, p), indexed by global sorted index
    LdsInts x;          // payload, indexed by global sorted index
    int wbeg, wend;     // staged index range [wbeg, wend)
    int t0;             // first PET of the tile
    const unsigned long long* m;   // optional: bit k of word w = "the staged PET with window index 64 w + k has a payload >= 0" (a core)
    // first staged PET of [j, end) with a payload >= 0, or end ([j, end) inside the staged range)
    __device__ __forceinline__ int next_set(int j, int end) const
    {
        int k = j - w.base;
        const int kend = end - w.base;
        while (k < kend) {
            const unsigned long long bits = m[k >> 6] >> (k & 63);
            if (bits) return min(k + __ffsll((long long)bits) - 1, kend) + w.base;
            k = (k | 63) + 1;
        }
        return end;
    }
    // last staged PET of [beg, j] with a payload >= 0, or beg - 1
    __device__ __forceinline__ int prev_set(int j, int beg) const
    {
        int k = j - w.base;
        const int kbeg = beg - w.base;
        while (k >= kbeg) {
            const unsigned long long bits = m[k >> 6] << (63 - (k & 63));
            if (bits) { k -= __clzll((long long)bits); return (k >= kbeg ? k : kbeg - 1) + w.base; }
            k = (k & ~63) - 1;
        }
        return beg - 1;
    }
};

__device__ __forceinline__ int tile_of_block(int bid)
{
    const int xcd = bid & 7, kseq = bid >> 3;
    return ((kseq / K2_RUN) * 8 + xcd) * K2_RUN + (kseq % K2_RUN);
}
static inline int tile_grid(int ntiles) { return ((ntiles + 8 * K2_RUN - 1) / (8 * K2_RUN)) * (8 * K2_RUN); }

// all threads of the workgroup; returns false (for the whole workgroup) if the tile is empty
// The (q, p) arrays are the padded sorted arrays (SORT_PAD sentinels on both sides): unpredicated 16-byte loads, pairs
// interleaved on the way into LDS (lw must be 16-byte aligned).  The payload array is not padded: predicated dwords.
// lmask (optional, (NT + 2 * HALO) / 64 words): Tile::m, built from the payload loads with one ballot per wave and pass.
// NT = PETs of the tile, NTH = threads of the workgroup (NT / NTH PETs per thread).
template <int NT, int HALO, int NTH = NT>
__device__ __forceinline__ bool tile_stage(Tile& t, int2* lw, int* lx, int ntiles, int M,
                                           const int* __restrict__ gq, const int* __restrict__ gp,
                                           const int* __restrict__ gx, unsigned long long* lmask = nullptr, int fill = -1,
                                           int tile_in = -1 /* the tile, if it is not the workgroup's own */)
{
    constexpr int T_WIN = NT + 2 * HALO, NV = T_WIN / 4;
    static_assert(NT % 64 == 0 && HALO % 64 == 0 && NT + HALO + 64 <= SORT_PAD, "window shape");
    const int tile = tile_in >= 0 ? tile_in : tile_of_block(blockIdx.x);
    t.t0 = tile * NT;
    if (tile >= ntiles || t.t0 >= M) return false;
    const int base = t.t0 - HALO;
    {
        const int4* __restrict__ gq4 = reinterpret_cast<const int4*>(gq + base);
        const int4* __restrict__ gp4 = reinterpret_cast<const int4*>(gp + base);
        int4* l4 = reinterpret_cast<int4*>(lw);
        for (int c = threadIdx.x; c < NV; c += NTH) {
            const int4 q = gq4[c], p = gp4[c];
            l4[2 * c] = make_int4(q.x, p.x, q.y, p.y);
            l4[2 * c + 1] = make_int4(q.z, p.z, q.w, p.w);
        }
    }
    for (int k = threadIdx.x; k < T_WIN; k += NTH) {
        const int gi = base + k;
        const bool in = gi >= 0 && gi < M;
        const int x = in ? gx[gi] : fill;                // every payload test is `>= (something >= 0)` (K2 words: fill = 0, "no neighbour")
        lx[k] = x;
        if (lmask) {
            const unsigned long long bal = __ballot(in && x >= 0);
            if ((threadIdx.x & 63) == 0) lmask[k >> 6] = bal;
        }
    }
    __syncthreads();
    t.w.a = lw; t.w.base = base; t.x.a = lx; t.x.base = base; t.m = lmask;
    t.wbeg = max(base, 0); t.wend = min(base + T_WIN, M);
    return true;
}

// The same staging with the K2 WORDS as the payload, read through a WordSrc (a run that re-uses the words of an earlier run
// of its eps finds them at another place, strip by strip): the thread that loads four consecutive (q, sp) pairs also loads
// their four words -- the strip of a pair is in its sp.  Outside [0, M): word 0 ("no neighbour").
template <int NT, int HALO, int NTH = NT>
__device__ __forceinline__ bool tile_stage_words(Tile& t, int2* lw, int* lx, int ntiles, int M,
                                                 const int* __restrict__ gq, const int* __restrict__ gp, const WordSrc& ws)
{
    constexpr int T_WIN = NT + 2 * HALO, NV = T_WIN / 4;
    static_assert(NT % 64 == 0 && HALO % 64 == 0 && NT + HALO + 64 <= SORT_PAD, "window shape");
    const int tile = tile_of_block(blockIdx.x);
    t.t0 = tile * NT;
    if (tile >= ntiles || t.t0 >= M) return false;
    const int base = t.t0 - HALO;
    {
        const int4* __restrict__ gq4 = reinterpret_cast<const int4*>(gq + base);
        const int4* __restrict__ gp4 = reinterpret_cast<const int4*>(gp + base);
        int4* l4 = reinterpret_cast<int4*>(lw);
        int4* x4 = reinterpret_cast<int4*>(lx);
        for (int c = threadIdx.x; c < NV; c += NTH) {
            const int4 q = gq4[c], p = gp4[c];
            const int gi = base + 4 * c;
            int4 w;
            w.x = (gi >= 0 && gi < M) ? ws.raw(gi, q.x, p.x) : 0;
            w.y = (gi + 1 >= 0 && gi + 1 < M) ? ws.raw(gi + 1, q.y, p.y) : 0;
            w.z = (gi + 2 >= 0 && gi + 2 < M) ? ws.raw(gi + 2, q.z, p.z) : 0;
            w.w = (gi + 3 >= 0 && gi + 3 < M) ? ws.raw(gi + 3, q.w, p.w) : 0;
            l4[2 * c] = make_int4(q.x, p.x, q.y, p.y);
            l4[2 * c + 1] = make_int4(q.z, p.z, q.w, p.w);
            x4[c] = w;
        }
    }
    __syncthreads();
    t.w.a = lw; t.w.base = base; t.x.a = lx; t.x.base = base; t.m = nullptr;
    t.wbeg = max(base, 0); t.wend = min(base + T_WIN, M);
    return true;
}

// visit, in ascending order, every j of the strip segment [sb,se) with q_j in [qlo,qhi]:
// f(j, q_j, p_j, x_j).  SET: only the PETs with a payload >= 0 need a visit (the tile carries the mask Tile::m): the walk
// goes from set bit to set bit -- a window of background noise costs one or two mask words instead of its candidates.
template <bool SET = false, typename F>
__device__ __forceinline__ void tile_visit_segment(const Tile& t, const int* __restrict__ gq, const int* __restrict__ gp,
                                                   const int* __restrict__ gx, int sb, int se, int qlo, int qhi, F&& f)
{
    if (sb >= se) return;
    // Only the part of the segment that can hold the q window has to be staged: the segment is sorted by q, so if the
    // first staged PET of it lies below qlo everything in front of the window does, and if the last staged one lies
    // above qhi everything behind it does (the windows of the neighbour strips lie about one strip population away
    // from the PET: whole strips are rarely inside the staged range).
    bool staged = true;
    if (sb < t.wbeg) { if (t.wbeg < se && t.w[t.wbeg].x < qlo) sb = t.wbeg; else staged = false; }
    if (staged && se > t.wend) { if (t.wend > sb && t.w[t.wend - 1].x > qhi) se = t.wend; else staged = false; }
    if (staged && se - sb <= 2047) {
        int j = (se - sb <= 255) ? lds_lower_bound8(t.w, sb, se, qlo) : lds_lower_bound8<T_STEPS_LONG>(t.w, sb, se, qlo);
        if (SET) {
            const int2* __restrict__ lw = t.w.a; const int* __restrict__ lx = t.x.a;
            const int base = t.w.base, kend = se - base;
            int k = j - base;
            while (k < kend) {
                unsigned long long bits = t.m[k >> 6] >> (k & 63);
                bool out = false;
                while (bits) {                                  // two set bits per round, their LDS reads in flight together
                    const int i0 = k + __ffsll((long long)bits) - 1;
                    bits &= bits - 1;
                    const bool two = bits != 0;
                    const int i1 = two ? k + __ffsll((long long)bits) - 1 : i0;
                    bits &= bits - 1;                           // (0 stays 0)
                    if (i0 >= kend) { out = true; break; }
                    const int i1c = min(i1, kend - 1);
                    const int2 c0 = lw[i0], c1 = lw[i1c];
                    const int x0 = lx[i0], x1 = lx[i1c];
                    if (c0.x > qhi) { out = true; break; }
                    f(i0 + base, c0.x, c0.y, x0);
                    if (two) {
                        if (i1 >= kend || c1.x > qhi) { out = true; break; }
                        f(i1 + base, c1.x, c1.y, x1);
                    }
                }
                if (out) break;
                k = (k | 63) + 1;
            }
            return;
        }
        // four candidates per round, all LDS reads in flight before the first of them is looked at
        while (j < se) {
            int2 c[4]; int x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int idx = min(j + k, se - 1); c[k] = t.w[idx]; x[k] = t.x[idx]; }
            bool out = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (out || j + k >= se || c[k].x > qhi) { out = true; continue; }
                f(j + k, c[k].x, c[k].y, x[k]);
            }
            if (out) break;
            j += 4;
        }
    } else {
        // the window is not staged (a strip population beyond the halo): global memory, with the loads of 4 candidates
        // in flight before the first of them is looked at
        int j = lower_bound_4(gq, sb, se, qlo);
        while (j < se) {
            int q[4], p[4], x[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int idx = min(j + k, se - 1); q[k] = gq[idx]; p[k] = gp[idx]; x[k] = gx[idx]; }
            bool out = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (out || j + k >= se || q[k] > qhi) { out = true; continue; }
                f(j + k, q[k], p[k], x[k]);
            }
            if (out) break;
            j += 4;
        }
    }
}
// Walk the sorted order from j on while more(pair) holds (a monotone predicate: the end of a q window inside one strip);
// see(j, x_j) for every PET with a payload >= 0 whose pair passes acc().  The first four candidates are read at once
// (most windows of non-core PETs end there); longer windows go from set bit to set bit of the tile's mask, and the pair at
// the start of the next mask word tells whether the window reaches it.  What lies outside the staged range (the
// sentinel pads included in it) is read from global memory.
// skip(x): the walk may pass over a staged PET with payload x without looking at its pair (k_border, variant 2: a core of a
// component the point is already known to be adjacent to cannot change anything -- a window next to a cluster is mostly that).
template <int T_WIN, typename MORE, typename ACC, typename SEE, typename SKIPF>
__device__ __forceinline__ void tile_walk_from(const Tile& t, const int* __restrict__ gq, const int* __restrict__ gp,
                                               const int* __restrict__ gx, int M, int j, MORE&& more, ACC&& acc, SEE&& see, SKIPF&& skip, int dbg = 0,
                                               const int* uw = nullptr /* optional, per mask word: the payload ALL its set PETs share (-1: none set, -2: several) */)
{
    const int base = t.w.base;
    int k = j - base;
#ifdef CLOOPS_DEVEL
    if ((dbg & 4194304) && !(k >= 0 && k + 4 <= T_WIN)) return;       // developer ablation: no walk that starts outside the staged range
#endif
    if (k >= 0 && k + 4 <= T_WIN) {
        const int2* __restrict__ lw = t.w.a; const int* __restrict__ lx = t.x.a;
        {
            int2 c[4]; int x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { c[u] = lw[k + u]; x[u] = lx[k + u]; }
            bool out = false;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (out || !more(c[u])) { out = true; continue; }
                if (x[u] >= 0 && acc(c[u])) see(j + u, x[u]);
            }
            if (out) return;
        }
#ifdef CLOOPS_DEVEL
        if (dbg & 2097152) return;                                       // developer ablation: the first four candidates only
#endif
        k += 4;
        {
            // The window goes on: where does it end?  more() is monotone along the sorted order, so its end among the next 63
            // staged PETs is a 6-step search on the pairs; the cores of exactly that stretch are the set bits of a 64-bit slice of
            // the mask (it may straddle two words) -- they are visited without asking more() again, and none beyond the end.
            int len = 0;
#pragma unroll
            for (int step = 32; step >= 1; step >>= 1) {
                const int idx = k + len + step - 1;
                const int2 v = lw[min(idx, T_WIN - 1)];
                len = (idx < T_WIN && more(v)) ? len + step : len;
            }
            const int w0 = k >> 6, sh = k & 63;
            constexpr int NW = T_WIN / 64;
            // (a mask word whose cores all carry a payload the walk passes over -- the inside of the cluster a border point sits
            //  next to -- counts as empty: one LDS read instead of a round per two cores)
            unsigned long long win = (uw && skip(uw[w0])) ? 0ull : t.m[w0] >> sh;
            if (sh && w0 + 1 < NW && !(uw && skip(uw[w0 + 1]))) win |= t.m[w0 + 1] << (64 - sh);
            win &= (1ull << len) - 1ull;                          // len <= 63
            while (win) {                                        // two cores per round
                const int i0 = k + __ffsll((long long)win) - 1;
                win &= win - 1;
                const bool two = win != 0;
                const int i1 = two ? k + __ffsll((long long)win) - 1 : i0;
                win &= win - 1;
                const int x0 = lx[i0], x1 = lx[i1];
                const bool s0 = skip(x0), s1 = !two || skip(x1);
                if (s0 & s1) continue;
                const int2 c0 = lw[i0], c1 = lw[i1];
                if (!s0 && acc(c0)) see(i0 + base, x0);
                if (!s1 && acc(c1)) see(i1 + base, x1);
            }
            if (k + len >= T_WIN) { j = T_WIN + base; goto global_part; }       // the window reaches the end of the staged range: global memory
            if (len < 63) return;                                // the window ended inside the slice
            k += 63;
        }
        while (k < T_WIN) {
            const int knext = (k | 63) + 1;
            unsigned long long bits = (uw && skip(uw[k >> 6])) ? 0ull : t.m[k >> 6] >> (k & 63);
            const int2 cn = lw[min(knext, T_WIN - 1)];
            const bool goes_on = knext >= T_WIN || more(cn);
            while (bits) {                                      // two set bits per round
                const int i0 = k + __ffsll((long long)bits) - 1;
                bits &= bits - 1;
                const bool two = bits != 0;
                const int i1 = two ? k + __ffsll((long long)bits) - 1 : i0;
                bits &= bits - 1;
                const int x0 = lx[i0], x1 = lx[i1];
                const bool s0 = skip(x0), s1 = !two || skip(x1);
                if (s0 & s1) continue;                          // (passing over them may carry the walk beyond the window's end, inside
                                                                //  this mask word: the next PET that is looked at ends it)
                const int2 c0 = lw[i0], c1 = lw[i1];
                if (!s0) {
                    if (!more(c0)) return;
                    if (acc(c0)) see(i0 + base, x0);
                }
                if (!s1) {
                    if (!more(c1)) return;
                    if (acc(c1)) see(i1 + base, x1);
                }
            }
            if (!goes_on) return;
            k = knext;
        }
        j = k + base;
    }
global_part:
#ifdef CLOOPS_DEVEL
    if (dbg & 8388608) return;                                           // developer ablation: nothing in global memory
#endif
    for (;;) {
        int q[4], p[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { q[u] = gq[j + u]; p[u] = gp[j + u]; x[u] = j + u < M ? gx[j + u] : -1; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!more(make_int2(q[u], p[u]))) return;
            if (x[u] >= 0 && acc(make_int2(q[u], p[u]))) see(j + u, x[u]);
        }
        j += 4;
    }
}

// own strip: walk left from i-1 down to b while q >= qlo, then right from i+1 up to e while
// q <= qhi; f(j, x_j) returns true to stop that direction early.  `dirs`: bit0 left, bit1 right.
// gxf(j, q_j): the payload of a PET outside the staged range.
template <typename GX, typename F>
__device__ __forceinline__ void tile_visit_own(const Tile& t, const int* __restrict__ gq, GX&& gxf,
                                               int i, int b, int e, int qlo, int qhi, int dirs, F&& f)
{
    if (dirs & 1)
        for (int j = i - 1; j >= b; --j) {
            const bool in = j >= t.wbeg;
            const int q = in ? t.w[j].x : gq[j];
            if (q < qlo) break;
            if (f(j, in ? t.x[j] : gxf(j, q))) break;
        }
    if (dirs & 2)
        for (int j = i + 1; j < e; ++j) {
            const bool in = j < t.wend;
            const int q = in ? t.w[j].x : gq[j];
            if (q > qhi) break;
            if (f(j, in ? t.x[j] : gxf(j, q))) break;
        }
}

// own strip, every PET of the q window (no early exit): four candidates per LDS round trip while the walk stays inside
// the staged range, the one-by-one walk of tile_visit_own for what is left
template <typename F>
__device__ __forceinline__ void tile_visit_own_all(const Tile& t, const int* __restrict__ gq, const int* __restrict__ gx,
                                                   int i, int b, int e, int qlo, int qhi, F&& f)
{
    int j = i - 1;
    while (j >= b && j - 3 >= t.wbeg) {
        int q[4], x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int idx = max(j - k, b); q[k] = t.w[idx].x; x[k] = t.x[idx]; }
        bool out = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (out || j - k < b || q[k] < qlo) { out = true; continue; }
            f(j - k, x[k]);
        }
        if (out) { j = b - 1; break; }
        j -= 4;
    }
    for (; j >= b; --j) {
        const bool in = j >= t.wbeg;
        if ((in ? t.w[j].x : gq[j]) < qlo) break;
        f(j, in ? t.x[j] : gx[j]);
    }
    j = i + 1;
    while (j < e && j + 3 < t.wend) {
        int q[4], x[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int idx = min(j + k, e - 1); q[k] = t.w[idx].x; x[k] = t.x[idx]; }
        bool out = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (out || j + k >= e || q[k] > qhi) { out = true; continue; }
            f(j + k, x[k]);
        }
        if (out) { j = e; break; }
        j += 4;
    }
    for (; j < e; ++j) {
        const bool in = j < t.wend;
        if ((in ? t.w[j].x : gq[j]) > qhi) break;
        f(j, in ? t.x[j] : gx[j]);
    }
}

// scatter counts back to input-row order (cl_neighbor_counts)
__global__ void k_scatter_counts(const int* __restrict__ strip_start, int S, const u32* __restrict__ srow,
                                 const int* __restrict__ cnt, int* __restrict__ out)
{
    const int M = strip_start[S];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    out[srow[i]] = cnt[i];
}

// ------------------------------------------------------------------------------------------
// per-run initialisation of the per-point / per-root arrays
// ------------------------------------------------------------------------------------------
__global__ void k_init_flags(int n, int* __restrict__ flag, int* __restrict__ counters)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 16) counters[i] = 0;
    if (i <= n) flag[i] = 0;           // n+1 entries
}
__global__ void k_init_arrays(int n, int* __restrict__ parent, int* __restrict__ compkey, int* __restrict__ ncore,
                              int* __restrict__ bsize, int* __restrict__ usize, int* __restrict__ cellfirst,
                              int* __restrict__ flag, int* __restrict__ state, int* __restrict__ counters)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 16 && counters) counters[i] = 0;
    if (i > n) return;
    flag[i] = 0;                       // n+1 entries
    if (i == n) return;
    parent[i] = i;
    compkey[i] = INT_MAX;
    ncore[i] = 0;
    bsize[i] = 0;
    usize[i] = 0;
    cellfirst[i] = INT_MAX;
    state[i] = 0;
}

// ------------------------------------------------------------------------------------------
// K3: union of core points
// ------------------------------------------------------------------------------------------
// Inside a strip every pair is within eps in `a`, so core points whose v-gaps are <= eps
// form a CHAIN.  Chains are resolved without union-find: chainflag[i] = i+1 for a core that
// opens a chain (no earlier core of its strip within eps), 0 otherwise; an inclusive
// max-scan then gives every core its chain head.  The union-find forest starts flat
// (parent = chain head), so no million-long pointer chains ever exist -- dense diagonals
// (self-ligation PETs) become one chain per strip.
// NT PETs per tile, NTH threads: NT / NTH PETs per thread, 64 consecutive PETs per wave and pass (the staging and its halo are amortised)
template <int NT, int HALO, int NTH = NT>
__global__ void __launch_bounds__(NTH)
k_chain_flags(GridParams g, int ntiles, int n, const int* __restrict__ sv, const int* __restrict__ sa,
              const int* __restrict__ strip_start, WordSrc ws, unsigned char* __restrict__ chainflag,
              int* __restrict__ head, int* __restrict__ wavelast, const u32* __restrict__ srow, int* __restrict__ cellfirst)
{
    constexpr int T_WIN = NT + 2 * HALO, NG = NT / 64;
    __shared__ __attribute__((aligned(16))) int2 lw[T_WIN];
    __shared__ __attribute__((aligned(16))) int lx[T_WIN];
    // variant 2: cellfirst[cell head] = smallest input row of the cell's PETs (the dict insertion order of cDBSCAN2.py:117), by LDS
    // atomics on the tile's own cells; the tile that holds a cell's head also walks the part of the cell behind its last PET
    __shared__ int lmin[NT];
    __shared__ int l_hlast;
    __shared__ unsigned long long l_core[T_WIN / 64];   // bit k of word w: the staged PET with window index 64 w + k is a core
    __shared__ unsigned long long l_chead[NG];          // variant 2, bit k of word w: PET t0 + 64 w + k is the first of its rotated cell
    const int M = strip_start[g.S];
    if (head) for (int k = threadIdx.x; k < NT; k += NTH) lmin[k] = INT_MAX;
    Tile t;
    if (!tile_stage_words<NT, HALO, NTH>(t, lw, lx, ntiles, M, sv, sa, ws)) return;
#ifdef CLOOPS_DEVEL
    if (g.dbg & (1 << 27)) return;                      // developer ablation (results invalid): staging only
#endif
    const int lane = threadIdx.x & 63;
    // Two bit masks instead of per-PET searches.  (1) The cores of the staged window (word 0 = "no neighbour" outside [0, M)): a
    // core's nearest earlier / later core in sorted order is a find-first-set away, and since a strip is sorted by q that PET
    // alone decides whether ANY core of the strip lies within eps on that side.  (2) Variant 2: a PET starts a rotated cell
    // (strip, q / eps) iff its predecessor lies in an earlier strip or below the cell's lower q edge; the head of a PET's cell
    // is then the latest such PET at or before it.
    for (int k = threadIdx.x; k < T_WIN; k += NTH) {
        const unsigned long long bal = __ballot(cw_core(lx[k], g.minPts));
        if (lane == 0) l_core[k >> 6] = bal;
    }
    if (head)
        for (int u = 0; u < NT / NTH; ++u) {
            const int i = t.t0 + (int)threadIdx.x + u * NTH;
            const int2 me = t.w[i], pv = t.w[i - 1];     // (HALO >= 1: the predecessor is staged; i = 0 has none)
            const int p0 = me.y & ~(g.peps - 1), q0 = div_eps(g, me.x) * g.eps;      // lower edges of the rotated cell (sp space / q space)
            const unsigned long long bal = __ballot(i < M && (i == 0 || (pv.y & ~(g.peps - 1)) != p0 || pv.x < q0));
            if (lane == 0) l_chead[u * (NTH / 64) + (threadIdx.x >> 6)] = bal;
        }
    __syncthreads();
    t.m = l_core;
    unsigned headmask = 0u;                             // bit u: the thread's u-th PET is the head of its cell
    for (int u = 0; u < NT / NTH; ++u) {
    const int i = t.t0 + (int)threadIdx.x + u * NTH;
    if (i >= M) continue;
    const int2 me = t.w[i];
    const bool core = cw_core(t.x[i], g.minPts);
    int pos = -1;
#ifdef CLOOPS_DEVEL
    if (head && !(g.dbg & (1 << 28))) {
#else
    if (head) {
#endif
        // variant 2: head of the PET's rotated cell = the latest cell-opening PET at or before it: inside the 64-PET group from
        // its ballot, else the nearest earlier group of the tile that has one, else the cell began in front of the tile
        // (variant 2 runs with A0 = V0 = 0)
        const int grp = u * (NTH / 64) + (int)(threadIdx.x >> 6);
        const unsigned long long upto = l_chead[grp] & ((2ull << lane) - 1ull);
        if (upto) pos = (i - lane) + 63 - __clzll((long long)upto);
        else
            for (int g2 = grp - 1; g2 >= 0; --g2) {
                const unsigned long long o2 = l_chead[g2];
                if (o2) { pos = t.t0 + 64 * g2 + 63 - __clzll((long long)o2); break; }
            }
        if (pos < 0 && core) {
            // the cell began in front of the tile (that tile takes care of its minimum) and only a core needs to know where:
            // a bisection on the left halo, then on the strip in global memory
            const int p0 = me.y & ~(g.peps - 1), q0 = div_eps(g, me.x) * g.eps;
            int lo = t.wbeg, hi = t.t0;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const int2 c = t.w[mid];
                if (c.y < p0 || c.x < q0) lo = mid + 1; else hi = mid;
            }
            if (lo == t.wbeg && lo > 0) {
                lo = strip_start[strip_of(g, me.y)]; hi = t.wbeg;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (sv[mid] < q0) lo = mid + 1; else hi = mid;          // (inside the PET's own strip)
                }
            }
            pos = lo;
        }
#ifdef CLOOPS_DEVEL
        if (!(g.dbg & (1 << 30)))
#endif
        if (pos >= t.t0) atomicMin(&lmin[pos - t.t0], (int)srow[i]);        // (a cell that began in an earlier tile: that tile walks it)
        headmask |= (pos == i ? 1u : 0u) << u;
        if (i == min(t.t0 + NT, M) - 1) l_hlast = pos;
    }
    int f = 0, last = 0;
#ifdef CLOOPS_DEVEL
    if (core && !(g.dbg & (1 << 29))) {
#else
    if (core) {
#endif
        const int p0 = me.y & ~(g.peps - 1);             // the strip's block of sp values
        const int qlo = sat_add(me.x, -g.eps), qhi = sat_add(me.x, g.eps);
        f = i + 1;
        last = 1;
        // the nearest earlier core of the sorted order: if it lies in this strip and within eps, the PET continues its chain;
        // if it does not, no earlier core of the strip does (sorted by q).  The same on the other side.  (The strip table is
        // looked up only when a strip goes on outside the staged range.)
        {
            const int j = t.prev_set(i - 1, t.wbeg);
            // variant 2: the component key is a minimum over the CELLS of its cores (cDBSCAN2.py:117-140), so one core per cell
            // is enough to carry the cell to k_flatten: the first one (no core between the cell's head and the PET; a core
            // that cannot tell, because the cell began in front of the staged range, carries it as well) -- the others get -1
            // and cost k_flatten neither the look-up of the cell nor an atomic
            if (head) head[i] = (j < pos || j < t.wbeg) ? pos : -1;
            if (j >= t.wbeg) { const int2 c = t.w[j]; f = ((c.y & ~(g.peps - 1)) == p0 && c.x >= qlo) ? 0 : f; }
            else if (t.wbeg > 0 && (t.w[t.wbeg].y & ~(g.peps - 1)) == p0) {
                const int b = strip_start[strip_of(g, me.y)];
                for (int k = t.wbeg - 1; k >= b; --k) {           // the strip began in front of the staged range: global memory
                    const int qk = sv[k];
                    if (qk < qlo) break;
                    if (cw_core(ws.raw(k, qk, me.y), g.minPts)) { f = 0; break; }
                }
            }
        }
        {
            const int j = t.next_set(i + 1, t.wend);
            if (j < t.wend) { const int2 c = t.w[j]; last = ((c.y & ~(g.peps - 1)) == p0 && c.x <= qhi) ? 0 : last; }
            else if (t.wend < M && (t.w[t.wend - 1].y & ~(g.peps - 1)) == p0) {
                const int e = strip_start[strip_of(g, me.y) + 1];
                for (int k = t.wend; k < e; ++k) {
                    const int qk = sv[k];
                    if (qk > qhi) break;
                    if (cw_core(ws.raw(k, qk, me.y), g.minPts)) { last = 0; break; }
                }
            }
        }
    }
    // one byte per PET for k_chain_parent: CF_CORE, CF_OPEN (no earlier core of its strip within eps), CF_LAST (no later one:
    // its q is the chain's upper end)
    chainflag[i] = (unsigned char)((core ? CF_CORE : 0) | (f ? CF_OPEN : 0) | (last ? CF_LAST : 0));
    // wavelast[w] = the last chain-opening PET (+1) among the 64 PETs [64 w, 64 w + 64), 0 = none: what k_chain_parent needs to
    // find a core's chain head without a scan over all PETs (the lanes still here are the wave's PETs below M; lane 0 is one)
    const unsigned long long ob = __ballot(f != 0);
    if ((threadIdx.x & 63) == 0) wavelast[i >> 6] = ob ? i + (64 - __clzll((long long)ob)) : 0;
    }
    if (!head) return;
    __syncthreads();
    const int tend = t.t0 + NT;
    if (threadIdx.x < 64 && tend < M && l_hlast >= t.t0) {
        // the cell of the tile's last PET may go on behind the tile: wave 0 walks it, 64 PETs per round (right halo, then global memory)
        const int2 lp = t.w[tend - 1];
        const int p0 = lp.y & ~(g.peps - 1), qend = div_eps(g, lp.x) * g.eps + g.eps;
        int m = INT_MAX;
        for (int j0 = tend; j0 < M; j0 += 64) {
            const int j = j0 + (int)threadIdx.x;
            bool in = j < M;
            if (in) {
                const int2 c = j < t.wend ? t.w[j] : make_int2(sv[j], sa[j]);
                in = (c.y & ~(g.peps - 1)) == p0 && c.x < qend;
            }
            if (in) m = min(m, (int)srow[j]);
            if (__ballot(in) != ~0ull) break;            // the cell ends inside this round (its PETs are contiguous)
        }
        m = dpp_reduce_wave(m, OpMin());
        if (threadIdx.x == 0 && m != INT_MAX) atomicMin(&lmin[l_hlast - t.t0], m);
    }
    __syncthreads();
    for (int u = 0; u < NT / NTH; ++u)
        if (headmask & (1u << u)) { const int i = t.t0 + (int)threadIdx.x + u * NTH; cellfirst[i] = lmin[i - t.t0]; }
}
// parent[] = chain head for core points (flat forest to start from); chainid[] = the same for
// core points and -1 for everything else (what the union kernel stages as its payload)
// Every component root is a chain head, so the per-root accumulators are reset here, by the chain
// heads only, instead of memset-ing five N-sized arrays per run.
// pmax32 (optional): the 32-PET block summaries of the union scan (max strip coordinate over the block's CORE PETs, see
// k_union_cores) come out of the same pass -- every thread already knows whether its PET is a core
#define CP_PER 4
__global__ void k_chain_parent(const int* __restrict__ strip_start, int S, const unsigned char* __restrict__ cflag,
                               const int* __restrict__ wavelast, int* __restrict__ parent, int* __restrict__ chainid,
                               int* __restrict__ compkey, int* __restrict__ ncore, int* __restrict__ bsize,
                               int* __restrict__ usize, int* __restrict__ state,
                               const int* __restrict__ sv, int* __restrict__ chain_qend,
                               const int* __restrict__ sa, int* __restrict__ pmax32 /* or null */)
{
    const int M = strip_start[S];
    // A core's chain head = the latest chain-opening PET at or before it in sorted order (what an inclusive max-scan of the
    // flags i + 1 / 0 gives): inside the wave from a ballot, else the nearest earlier 64-PET group that has one
    // (k_chain_flags left wavelast[]; normally the group right in front -- 64 groups are looked at per round trip).
    // CP_PER PETs per thread (a wave handles CP_PER runs of 64 consecutive PETs): all their loads are in flight together.
    const int lane = threadIdx.x & 63;
    int ii[CP_PER], fl[CP_PER], spv[CP_PER], qv[CP_PER];
#pragma unroll
    for (int e = 0; e < CP_PER; ++e) {
        ii[e] = (blockIdx.x * CP_PER + e) * (int)blockDim.x + (int)threadIdx.x;
        const bool in = ii[e] < M;
        fl[e] = in ? (int)cflag[ii[e]] : 0;              // CF_CORE | CF_OPEN | CF_LAST
        spv[e] = (in && pmax32 && (fl[e] & CF_CORE)) ? sa[ii[e]] : INT_MIN;
        qv[e] = (in && (fl[e] & CF_LAST)) ? sv[ii[e]] : 0;
    }
#pragma unroll
    for (int e = 0; e < CP_PER; ++e) {
        const int i = ii[e];
        const bool core = (fl[e] & CF_CORE) != 0;
        const unsigned long long open = __ballot((fl[e] & CF_OPEN) != 0);
        const unsigned long long upto = open & ((2ull << lane) - 1ull);
        int head1 = upto ? (i - lane) + (64 - __clzll((long long)upto)) : 0;
        if (__any(core && !upto)) {
            int carry = 0;
            for (int base = ((i - lane) >> 6) - 1; base >= 0; base -= 64) {
                const int idx = base - lane;
                const int v = idx >= 0 ? wavelast[idx] : 0;
                const unsigned long long bal = __ballot(v != 0);
                if (bal) { carry = __builtin_amdgcn_readlane(v, __ffsll((long long)bal) - 1); break; }
            }
            if (!upto) head1 = carry;
        }
        if (i < M) {
            const int h = core ? head1 - 1 : -1;
            chainid[i] = h;
            // (only chain heads are ever looked up in the forest: k_union_cores unites heads, k_flatten starts from chainid)
            if (h == i) { parent[i] = i; compkey[i] = INT_MAX; ncore[i] = 0; bsize[i] = 0; usize[i] = 0; state[i] = ST_LIVE; }
            if (core && (fl[e] & CF_LAST)) chain_qend[h] = qv[e];       // indexed by chain head
        }
        if (pmax32) {                                          // uniform: every lane of the wave takes part in the reduction
            int v = spv[e];
            v = dpp_reduce_halves(v, OpMax());                 // lanes 31 and 63 hold the maxima of their 32-PET blocks
            if ((threadIdx.x & 31) == 31 && (i - 31) < M) pmax32[i >> 5] = v;
        }
    }
}

// Cross-strip edges: a core i of strip s against the cores of strip s-1 in its window (the
// pairs with strip s+1 are handled from the other endpoint).  Chains, not points, are what
// has to be united: every lane collects the distinct chains B of strip s-1 it touches, the
// wave then keeps ONE lane per distinct (own chain A, chain B) pair, and only those lanes run
// the (latency-bound, global-memory) union-find step -- directly on the chain heads.
#define UNION_MAXB 4

// NT PETs per tile, NTH threads (as k_border): the cores of the whole tile in one list, walked in rounds of NTH
template <int NT, int HALO, int NTH = NT>
__global__ void __launch_bounds__(NTH)
k_union_cores(GridParams g, int ntiles, const int* __restrict__ sv, const int* __restrict__ sa,
              const int* __restrict__ strip_start, const int* __restrict__ chainid, const int* __restrict__ chain_qend,
              const int* __restrict__ pmax32, int* parent)
{
    __shared__ __attribute__((aligned(16))) int2 lw[NT + 2 * HALO];
    __shared__ int lx[NT + 2 * HALO];
    __shared__ short l_list[NT];
    __shared__ int l_total;
    const int M = strip_start[g.S];
    Tile t;
    if (threadIdx.x == 0) l_total = 0;
    if (!tile_stage<NT, HALO, NTH>(t, lw, lx, ntiles, M, sv, sa, chainid)) return;
#pragma unroll
    for (int u = 0; u < NT / NTH; ++u) {
        const int tix = (int)threadIdx.x + u * NTH, i0 = t.t0 + tix;
        const bool core = i0 < M && t.x[i0 < M ? i0 : t.t0] >= 0;
        const unsigned long long bal = __ballot(core);
        int base = 0;
        if (bal) {
            const int first = __ffsll((long long)bal) - 1;
            if ((int)(threadIdx.x & 63) == first) base = atomicAdd(&l_total, __popcll(bal));
            base = __builtin_amdgcn_readlane(base, first);
        }
        if (core) l_list[base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = (short)tix;
    }
    __syncthreads();
    const int total = l_total;
#ifdef CLOOPS_DEVEL
    if (g.dbg & (1 << 24)) return;                      // developer ablation (results invalid): staging and the core list only
#endif
    for (int h = (int)threadIdx.x; h < total; h += NTH) {
    const int i = t.t0 + l_list[h];
    const int2 me = t.w[i];
    const int A = t.x[i];
    const int s = strip_of(g, me.y);
    int Bs[UNION_MAXB];
#pragma unroll
    for (int k = 0; k < UNION_MAXB; ++k) Bs[k] = -1;
    int nb = 0;
    if (s > 0) {
        int tb = strip_start[s - 1];
        const int b = strip_start[s];
        const int qlo = sat_add(me.x, -g.eps), qhi = sat_add(me.x, g.eps);
        // strip s-1 ends where strip s begins, i.e. inside the staged range; if its first staged PET lies below qlo the
        // part in front of the window cannot hold a candidate (sorted by q) and the staged part is the whole search range
        if (tb < t.wbeg && t.wbeg < b && t.w[t.wbeg].x < qlo) tb = t.wbeg;
        auto touch = [&](int B) {
            bool seen = false;
#pragma unroll
            for (int k = 0; k < UNION_MAXB; ++k) seen |= (Bs[k] == B);
            if (seen) return;
            if (nb < UNION_MAXB) {
#pragma unroll
                for (int k = 0; k < UNION_MAXB; ++k) if (k == nb) Bs[k] = B;
                ++nb;
            } else {
                uf_unite(parent, A, B);                    // more chains than slots: unite right away
            }
        };
        if (tb >= t.wbeg && b - tb <= 2047) {
            // short strips: the whole neighbour strip is staged.  Every candidate lies one strip below, so
            // "within eps in p" is p_j >= p_i - eps.  In a well-filled strip the walk jumps past a chain once
            // it has been touched (its last core has q = chain_qend[chain]): a window covered by one chain
            // costs one candidate instead of a hundred.
            const int T = me.y - g.peps;
            const bool dense = b - tb > 48;
            // search depth chosen per wave (a per-lane choice would make most waves run every variant)
            int j;
            const bool longA = __any(b - tb > 255);
            if (!__any(b - tb > 31)) j = lds_lower_bound8<5>(t.w, tb, b, qlo);
            else if (!__any(b - tb > 63)) j = lds_lower_bound8<6>(t.w, tb, b, qlo);
            else if (!longA) j = lds_lower_bound8<8>(t.w, tb, b, qlo);
            else j = lds_lower_bound8<T_STEPS_LONG>(t.w, tb, b, qlo);
#ifdef CLOOPS_DEVEL
            if (g.dbg & (1 << 25)) j = b;                // developer ablation: the searches, no candidate
#endif
            while (j < b) {
                // four candidates per round, all LDS reads in flight before the first compare
                int2 cv[4]; int bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const int idx = min(j + u, b - 1); cv[u] = t.w[idx]; bv[u] = t.x[idx]; }
                int next = j + 4;
                bool stop = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (stop || j + u >= b) continue;
                    if (cv[u].x > qhi) { stop = true; next = b; continue; }
                    if (bv[u] >= 0 && cv[u].y >= T) {
                        touch(bv[u]);
                        if (dense) {
                            const int qe = chain_qend[bv[u]];
                            stop = true;
                            next = (qe >= qhi) ? b : (longA ? lds_upper_bound8<T_STEPS_LONG>(t.w, j + u + 1, b, qe) : lds_upper_bound8(t.w, j + u + 1, b, qe));
                        }
                    }
                }
                j = next;
            }
        } else {
            // long strips (dense data at large eps: hundreds of candidates per window, nearly all of them in
            // ONE chain): once a chain has been touched the scan jumps past its last core, whose q is
            // chain_qend[chain] -- a window that is one chain costs two searches instead of a full scan
            // Every candidate j lies one strip below, so p_j < p_i and "within eps in p" is p_j >= p_i - eps.
            // A PET near the top of its strip is within eps in p of few PETs of the strip below and used to
            // walk through most of its window, one round trip per candidate: the tail of the kernel.  Now
            //  * the window end k1 is found up front (no q loads in the walk),
            //  * whole 32-PET blocks are skipped on their summary pmax32 (8 summaries per round trip),
            //  * inside a block the candidates are fetched UNION_CH at a time.
            const int T = me.y - g.peps;
            int k = lower_bound_4(sv, tb, b, qlo);
            const int k1 = lower_bound_4(sv, k, b, sat_add(qhi, 1));
            while (k < k1) {
                if (pmax32 && (k & 31) == 0 && k + 32 <= k1) {
                    const int blk = k >> 5, nblk = min(8, (k1 - k) >> 5);
                    int m[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) m[u] = pmax32[blk + min(u, nblk - 1)];
                    int hit = nblk;
#pragma unroll
                    for (int u = 0; u < 8; ++u) if (hit == nblk && u < nblk && m[u] >= T) hit = u;
                    k += hit * 32;
                    if (hit == nblk) continue;
                }
                const int lim = pmax32 ? min(k1, (k | 31) + 1) : k1;
                int cv[UNION_CH], pv[UNION_CH];
#pragma unroll
                for (int u = 0; u < UNION_CH; ++u) {
                    const int idx = min(k + u, lim - 1);
                    cv[u] = chainid[idx]; pv[u] = sa[idx];
                }
                int next = min(k + UNION_CH, lim);
                bool found = false;
#pragma unroll
                for (int u = 0; u < UNION_CH; ++u) {
                    if (found || k + u >= lim) continue;
                    if (cv[u] >= 0 && pv[u] >= T) {
                        touch(cv[u]);
                        const int qe = chain_qend[cv[u]];
                        found = true;
                        next = (qe >= qhi) ? k1 : lower_bound_4(sv, k + u + 1, k1, qe + 1);
                    }
                }
                k = next;
            }
        }
    }
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < UNION_MAXB; ++k) {
        const int B = Bs[k];
        unsigned long long pending = __ballot(B >= 0);
        bool rep = false;
        while (pending) {
            const int leader = __ffsll((long long)pending) - 1;
            const int LA = __builtin_amdgcn_readlane(A, leader), LB = __builtin_amdgcn_readlane(B, leader);
            const unsigned long long m = __ballot(B >= 0 && A == LA && B == LB);
            if (lane == leader) rep = true;
            pending &= ~m;
        }
#ifdef CLOOPS_DEVEL
        if (g.dbg & (1 << 26)) rep = false;              // developer ablation: no union-find step
#endif
        if (rep) uf_unite(parent, A, B);
    }
    }
}

// ------------------------------------------------------------------------------------------
// Two-level reduce-by-key in front of global atomics.  A giant component (the self-ligation
// diagonal at large eps: millions of PETs in ONE cluster) would otherwise put one atomic per
// wave on the same few addresses, which the L2 serialises at ~2 ns each (2.5 ms per kernel on a
// 16 M-PET chromosome).  Level 1: lanes of a wave sharing a key are reduced with ballots and
// shuffles.  Level 2: the per-wave results of a 1024-thread workgroup meet in a small LDS hash
// table; one global atomic per key per WORKGROUP remains.
// ------------------------------------------------------------------------------------------

// K3b: root per core point, component keys and core counts.
//   variant 1: key = smallest input row of a core point = the component's start point
//              (cDBSCAN.py:134-137)
//   variant 2: key = smallest cellfirst over the cells holding its core points
//              (cDBSCAN2.py:117-140)
__global__ void __launch_bounds__(BIGTPB)
k_flatten(GridParams g, const int* __restrict__ strip_start, const int* __restrict__ cnt /* plain counts, or null with chainid */,
          const int* __restrict__ chainid /* or null: chain head of a core PET, -1 otherwise */, int* parent, const u32* __restrict__ srow,
          const int* __restrict__ head, const int* __restrict__ cellfirst,
          int* __restrict__ root, int* __restrict__ compkey, int* __restrict__ ncore,
          int* __restrict__ rootlist /* or null */, int* __restrict__ counters)
{
    __shared__ int hkey[AGG_H], hmin[AGG_H], hcnt[AGG_H];
    __shared__ int l_nroot, l_rootbase;
    if (threadIdx.x < AGG_H) { hkey[threadIdx.x] = -1; hmin[threadIdx.x] = INT_MAX; hcnt[threadIdx.x] = 0; }
    if (threadIdx.x == 0) l_nroot = 0;
    __syncthreads();
    const int M = strip_start[g.S];
    // FLAT_PER PETs per thread: the kernel waits on dependent gathers (forest walk, cell -> first row) 88 % of its time at full
    // occupancy, so the walks of a thread's PETs advance together -- two independent chains per thread in flight
    int ii[FLAT_PER], r[FLAT_PER], key[FLAT_PER], x[FLAT_PER], hd[FLAT_PER];
    bool core[FLAT_PER];
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e) {
        ii[e] = (blockIdx.x * FLAT_PER + e) * BIGTPB + (int)threadIdx.x;
        if (!chainid) { core[e] = ii[e] < M && cw_core(cnt[ii[e]], g.minPts); x[e] = -1; }
        else { x[e] = ii[e] < M ? chainid[ii[e]] : -1; core[e] = x[e] >= 0; }      // the walk starts at the PET's chain head
    }
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e) {
        if (!chainid) x[e] = core[e] ? parent[ii[e]] : -1;
        hd[e] = (core[e] && g.variant == CL_VARIANT_CDBSCAN2) ? head[ii[e]] : 0;
    }
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e)
        key[e] = !core[e] ? INT_MAX : ((g.variant == CL_VARIANT_CDBSCAN2) ? (hd[e] >= 0 ? cellfirst[hd[e]] : INT_MAX) : (int)srow[ii[e]]);      // (-1: an earlier core of the cell carries it)
#ifdef CLOOPS_DEVEL
    if (g.dbg2 & 1) { if (key[0] == 12345 && x[0] == 54321) root[0] = 0; return; }      // developer ablation: the loads only
#endif
    {
        // the union kernel has completed (kernel boundary = coherent): plain loads, all chains of the thread step together
        bool todo = false;
#pragma unroll
        for (int e = 0; e < FLAT_PER; ++e) todo |= core[e];
#ifdef CLOOPS_DEVEL
        if (g.dbg2 & 2) todo = false;                    // developer ablation: no forest walk
#endif
        while (todo) {
            int p[FLAT_PER];
#pragma unroll
            for (int e = 0; e < FLAT_PER; ++e) p[e] = core[e] ? parent[x[e]] : -1;
            todo = false;
#pragma unroll
            for (int e = 0; e < FLAT_PER; ++e) { todo |= core[e] && p[e] != x[e]; x[e] = core[e] ? p[e] : x[e]; }
        }
    }
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e) {
        r[e] = core[e] ? x[e] : -1;
        if (ii[e] < M) root[ii[e]] = r[e];
    }
#ifdef CLOOPS_DEVEL
    if (g.dbg2 & 4) return;                              // developer ablation: no root list, no aggregation
#endif
    const int lane = threadIdx.x & 63;
    // the components' roots as a compact list (the per-component kernels that follow walk K entries instead of
    // testing every PET): ranks inside the workgroup through LDS, one global atomic per workgroup
    int myslot[FLAT_PER];
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e) {
        myslot[e] = -1;
        if (rootlist) {
            const bool isroot = r[e] == ii[e] && r[e] >= 0;
            const unsigned long long rb = __ballot(isroot);
            if (rb) {
                int wbase = 0;
                if (lane == 0) wbase = atomicAdd(&l_nroot, __popcll(rb));
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                if (isroot) myslot[e] = wbase + __builtin_amdgcn_mbcnt_hi((unsigned)(rb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)rb, 0u));
            }
        }
    }
    // a wave of one component (the inside of a large cluster): one reduction, one insertion by its first lane;
    // anything else: every lane goes to the workgroup's LDS table itself -- lanes that share a root meet on one LDS
    // address, which the LDS serialises far cheaper than a loop over the wave's distinct roots (or global atomics) would
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e) {
        const unsigned long long pending = __ballot(r[e] >= 0);
        if (pending) {
            const int leader = __ffsll((long long)pending) - 1;
            const int R = __builtin_amdgcn_readlane(r[e], leader);
            const unsigned long long m = __ballot(r[e] == R);
            if (m == pending && __popcll(m) >= 16) {
                int mk = r[e] == R ? key[e] : INT_MAX;
                mk = dpp_reduce_wave(mk, OpMin());
                if (lane == leader) {
                    const int sl = agg_slot(hkey, R);
                    if (sl >= 0) { atomicMin(&hmin[sl], mk); atomicAdd(&hcnt[sl], __popcll(m)); }
                    else { atomicMin(&compkey[R], mk); atomicAdd(&ncore[R], __popcll(m)); }
                }
            } else if (r[e] >= 0) {
                // (a segmented reduction over runs of one root -- as table_accumulate does for labels -- measured +29 % here: the
                //  runs are short and broken by non-core PETs)
                const int sl = agg_slot(hkey, r[e]);
                if (sl >= 0) { if (key[e] != INT_MAX) atomicMin(&hmin[sl], key[e]); atomicAdd(&hcnt[sl], 1); }
                else { if (key[e] != INT_MAX) atomicMin(&compkey[r[e]], key[e]); atomicAdd(&ncore[r[e]], 1); }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < AGG_H && hkey[threadIdx.x] >= 0) {
        atomicMin(&compkey[hkey[threadIdx.x]], hmin[threadIdx.x]);
        atomicAdd(&ncore[hkey[threadIdx.x]], hcnt[threadIdx.x]);
    }
    if (rootlist) {
        if (threadIdx.x == 0) l_rootbase = l_nroot ? atomicAdd(&counters[CTR_NROOT], l_nroot) : 0;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < FLAT_PER; ++e) if (myslot[e] >= 0) rootlist[l_rootbase + myslot[e]] = ii[e];
    }
}

// ------------------------------------------------------------------------------------------
// K4: border points
//   variant 2 (R2): the lowest-key adjacent component (first come, cDBSCAN2.py:130,212,352)
//   variant 1 (R1): max over adjacent components whose START POINT is a neighbour
//                   (unconditional overwrite, cDBSCAN.py:172-173), else the lowest-key
//                   adjacent component (first come, cDBSCAN.py:179-182)
// owner[i] = root of the owning component (cores: their own root), -1 = noise
// ------------------------------------------------------------------------------------------

// NT PETs per tile, NTH threads: NT / NTH PETs per thread in the first pass (staging and its barrier amortised: a halo of 128 on
// both sides is 1.25 x the tile at NT = 1024 instead of 2 x at 256), then the border points that have to walk are compacted over the
// WHOLE tile and walked in rounds of NTH -- a 256-PET tile left its one walking wave (or its second) half empty.
template <int NT, int HALO, int NTH = NT>
__global__ void __launch_bounds__(NTH)
k_border(GridParams g, int ntiles, const int* __restrict__ sv, const int* __restrict__ sa,
         const int* __restrict__ strip_start, const int* __restrict__ root, const int* __restrict__ compkey,
         const int* __restrict__ ncore, const u32* __restrict__ srow, int* __restrict__ owner, int* __restrict__ bsize,
         int* __restrict__ usize, WordSrc ws, int* __restrict__ clist, int* __restrict__ counters)
{
    __shared__ __attribute__((aligned(16))) int2 lw[NT + 2 * HALO];
    __shared__ int lx[NT + 2 * HALO];
    __shared__ short l_list[NT];
    __shared__ int l_total;
    __shared__ unsigned long long l_mask[(NT + 2 * HALO) / 64];
    __shared__ int l_uroot[(NT + 2 * HALO) / 64];       // per mask word: the root all its cores share (-1: no core, -2: several roots)
    __shared__ int l_enc[NT];
    constexpr int T_WIN = NT + 2 * HALO;
    const int M = strip_start[g.S];
    Tile t;
    if (threadIdx.x == 0) l_total = 0;
    if (!tile_stage<NT, HALO, NTH>(t, lw, lx, ntiles, M, sv, sa, root, l_mask)) return;
    for (int k = threadIdx.x; k < T_WIN; k += NTH) {
        const int x = lx[k];
        const unsigned long long cb = __ballot(x >= 0);
        const int R = cb ? __builtin_amdgcn_readlane(x, __ffsll((long long)cb) - 1) : -1;
        const unsigned long long other = __ballot(x >= 0 && x != R);
        if ((threadIdx.x & 63) == 0) l_uroot[k >> 6] = other ? -2 : R;
    }
#ifdef CLOOPS_DEVEL
#define KB_ABL(bit) (g.dbg & (bit))
    if (KB_ABL(65536)) return;                           // developer ablation (results invalid): staging only
#else
#define KB_ABL(bit) 0
#endif
    // first pass, NT / NTH PETs per thread: cores keep their root, isolated non-cores are noise, the rest has to walk
    // (the K2 words of the thread's non-core PETs: table entries first, then the words -- all loads of a stage in flight together)
    constexpr int PER = NT / NTH;
    int ri_[PER], enc_[PER];
    WordSrc::Where at_[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int tix = (int)threadIdx.x + u * NTH, i0 = t.t0 + tix;
        ri_[u] = i0 < M ? t.x[i0] : 0;
        const int2 pe = t.w[i0 < M ? i0 : t.t0];
        at_[u] = ws.where(i0, pe.x, pe.y);
        if (ri_[u] >= 0) at_[u].src = nullptr;
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) enc_[u] = at_[u].src ? at_[u].src[at_[u].idx] : 0;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int tix = (int)threadIdx.x + u * NTH, i0 = t.t0 + tix;
        bool border = false;
        if (i0 < M) {
            const int ri = ri_[u];
            if (ri >= 0) owner[i0] = ri;
            else {
                // K2 left either the neighbour count of a non-core PET (itself included) or its word with count and hints
                // (k_region_core): a count <= 1 = nothing within eps -- most of the background noise ends here
                const int enc = ws.shifted(enc_[u], at_[u]);
                l_enc[tix] = enc;
                if (cw_count(enc) <= 1) owner[i0] = -1; else border = true;
            }
        }
        // workgroup-wide list of the walkers: one LDS atomic per wave and pass
        const unsigned long long bal = __ballot(border);
        int base = 0;
        if (bal) {
            const int first = __ffsll((long long)bal) - 1;
            if ((int)(threadIdx.x & 63) == first) base = atomicAdd(&l_total, __popcll(bal));
            base = __builtin_amdgcn_readlane(base, first);
        }
        if (border) l_list[base + __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] = (short)tix;
    }
    __syncthreads();
    const int total = KB_ABL(131072) ? 0 : l_total;      // (ablation: no walkers)
    for (int h = (int)threadIdx.x; h < total; h += NTH) {
    const int i = t.t0 + l_list[h];
    const int enc = l_enc[l_list[h]];
    const int2 me = t.w[i];
    const int qlo = sat_add(me.x, -g.eps), qhi = sat_add(me.x, g.eps);
    const bool v1 = g.variant == CL_VARIANT_CDBSCAN1;
    int bestk = INT_MAX, best = -1, tk = -1, tbest = -1, lastr = -1, lastk = 0, first = -1;
    int adj = -1;                                       // variant 2: the component seen last (cores of it are passed over from then on)
    bool contested = false;
    auto see = [&](int j, int r) {
        if (r < 0) return;
        if (!v1) adj = r;
        if (first < 0) first = r; else if (r != first) contested = true;
        int k;
        if (r == lastr) k = lastk; else { k = compkey[r]; lastr = r; lastk = k; }
        if (k < bestk) { bestk = k; best = r; }
        if (v1 && (int)srow[j] == k && k > tk) { tk = k; tbest = r; }     // j is its component's start point
    };
    // Only core PETs count (see() ignores the rest).  Own strip: a non-core PET has fewer than minPts PETs of its strip
    // in its q window, so for minPts <= 128 the window lies inside the staged range and at most minPts - 1 positions
    // away.  All cores on ONE side of it are within eps of each other (same strip, q inside one eps) -- one component:
    // for variant 2 the nearest core on either side, found in the mask, stands for all of them.  Variant 1 needs every
    // neighbour (its start-point rule looks at single PETs), as do windows that may leave the staged range.
    const bool hinted = !v1 && enc < 0 && ((unsigned)enc & K2H_NONE) != K2H_NONE;
    if (hinted) {
        // variant 2 with K2's hints: no strip table, no searches.  Strips are aligned blocks of the strip coordinate:
        // "same strip" and "still inside the neighbour strip" are predicates of the staged pairs.
        const int pbeg = me.y & ~(g.peps - 1), pend = pbeg + g.peps, pend2 = pend + g.peps;
        const int plo = me.y - g.peps, phi = me.y + g.peps;
        if (!KB_ABL(262144)) {
            const int lb = max(i - (g.minPts - 1), t.w.base), re = min(i + g.minPts, t.w.base + T_WIN);
            const int jl = t.prev_set(i - 1, lb), jr = t.next_set(i + 1, re);
            const bool hl = jl >= lb, hr = jr < re;
            const int2 cl = t.w[hl ? jl : i], cr = t.w[hr ? jr : i];
            const int rl = t.x[hl ? jl : i], rr = t.x[hr ? jr : i];
            if (hl && cl.y >= pbeg && cl.x >= qlo) see(jl, rl);
            if (hr && cr.y < pend && cr.x <= qhi) see(jr, rr);
        }
        const int ja = i - (int)((unsigned)enc & K2H_MASK), jb = i + (int)(((unsigned)enc >> K2H_BITS) & K2H_MASK);
        if (!KB_ABL(524288))
        tile_walk_from<T_WIN>(t, sv, sa, root, M, ja, [&](int2 c) { return (c.y < pbeg) & (c.x <= qhi); },
                              [&](int2 c) { return c.y >= plo; }, see, [&](int x) { return x == adj; }, g.dbg, l_uroot);       // one strip below: sp can only be too low
        if (!KB_ABL(1048576))
        tile_walk_from<T_WIN>(t, sv, sa, root, M, jb, [&](int2 c) { return (c.y < pend2) & (c.x <= qhi); },
                              [&](int2 c) { return c.y <= phi; }, see, [&](int x) { return x == adj; }, g.dbg, l_uroot);       // one strip above: only too high
    } else {
        const int s = strip_of(g, me.y);
        const int b = strip_start[s], e = strip_start[s + 1];
        const int tb = s > 0 ? strip_start[s - 1] : b;
        const int te = s + 1 < g.S ? strip_start[s + 2] : e;
        if (!v1 && g.minPts <= 128) {
            const int lb = max(b, i - (g.minPts - 1)), re = min(e, i + g.minPts);
            const int jl = t.prev_set(i - 1, lb), jr = t.next_set(i + 1, re);
            const bool hl = jl >= lb, hr = jr < re;
            const int ql = t.w[hl ? jl : i].x, qr = t.w[hr ? jr : i].x;
            const int rl = t.x[hl ? jl : i], rr = t.x[hr ? jr : i];
            if (hl && ql >= qlo) see(jl, rl);
            if (hr && qr <= qhi) see(jr, rr);
        } else {
            tile_visit_own_all(t, sv, root, i, b, e, qlo, qhi, [&](int j, int r) { see(j, r); });
        }
        tile_visit_segment<true>(t, sv, sa, root, tb, b, qlo, qhi, [&](int j, int, int pj, int r) {
            const int da = pj - me.y; if ((da < 0 ? -da : da) <= g.peps) see(j, r); });
        tile_visit_segment<true>(t, sv, sa, root, e, te, qlo, qhi, [&](int j, int, int pj, int r) {
            const int da = pj - me.y; if ((da < 0 ? -da : da) <= g.peps) see(j, r); });
    }
    const int o = (v1 && tbest >= 0) ? tbest : best;
    owner[i] = o < 0 ? -1 : (contested ? (o | OWNER_CONTESTED) : o);

    // counts per owning component, reduced over the lanes of the wave that share the owner.  Only
    // components that are not already >= minPts on their cores need them (release rule of variant 2,
    // drop rule of variant 1) -- a giant component never sees one of these atomics.
    {
        const int lane = threadIdx.x & 63;
        const bool cnt_me = o >= 0 && ncore[o] < g.minPts;
        // only a component that is not live on its cores alone can end up uncertain (k_mark_uncertain_l): its CONTESTED border
        // points are all k_emit_records has to look at -- they are listed here (one atomic per wave)
        {
            const bool want = cnt_me && contested;
            const unsigned long long wb = __ballot(want);
            if (wb) {
                const int first = __ffsll((long long)wb) - 1;
                int base = 0;
                if (lane == first) base = atomicAdd(&counters[CTR_NFLAG], __popcll(wb));
                base = __builtin_amdgcn_readlane(base, first);
                if (want) clist[base + __builtin_amdgcn_mbcnt_hi((unsigned)(wb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wb, 0u))] = i;
            }
        }
        unsigned long long pending = __ballot(cnt_me);
        while (pending) {
            const int leader = __ffsll((long long)pending) - 1;
            const int O = __builtin_amdgcn_readlane(o, leader);
            const unsigned long long m = __ballot(cnt_me && o == O);
            const unsigned long long mu = __ballot(cnt_me && o == O && !contested);
            if (lane == leader) {
                atomicAdd(&bsize[O], __popcll(m));
                if (mu) atomicAdd(&usize[O], __popcll(mu));
            }
            pending &= ~m;
        }
    }
    }
}

// ---- variant 2 release rule (cDBSCAN2.py:180-183) ------------------------------------------
// A component is surely live if cores + first-come borders >= minPts (its share can only grow
// when lower components die).  The rest form the small uncertain set U, resolved in key order.


// records: for every border point adjacent to an uncertain component, its (<= 4, geometric
// bound) distinct adjacent components in ascending key order

// The contested border points that k_border listed (clist: points of components that are not live on their cores alone; a few
// hundred to a few ten thousand per run), one WAVE each.  Only one whose first-come owner (its lowest-key adjacent component)
// is UNCERTAIN can change hands: the fix-up walks a record's components in key order and a component that is surely live ends
// the walk (k_resolve_release), so a record that starts with a live component contributes nothing -- most listed points leave
// after that test.  The others collect their adjacent components.  The kernel is a handful of waves deep in dependent loads,
// so what counts is the number of round trips per point: the four walks (own strip downwards / upwards, one strip below, one
// above) start where K2's hints say -- no strip table, no search -- each is "from a start index while a monotone predicate of
// (q, sp) holds", the wave reads 64 candidates of every walk at once, all four first rounds in flight together, and looks at
// every DISTINCT root once.  (A thread per point that walked its windows one load after the other took 40 us per run.)
__global__ void __launch_bounds__(TPB)
k_emit_records(GridParams g, const int* __restrict__ sv, const int* __restrict__ sa,
               const int* __restrict__ strip_start, const int* __restrict__ root, const int* __restrict__ compkey,
               const int* __restrict__ state, const int* __restrict__ owner, Rec* __restrict__ recs, int rec_cap,
               int* __restrict__ counters, const int* __restrict__ clist, WordSrc ws)
{
    if (counters[CTR_NU] == 0) return;
    const int nlist = counters[CTR_NFLAG];
    const int M = strip_start[g.S];
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x * (TPB / 64);
    const int wave = blockIdx.x * (TPB / 64) + (int)(threadIdx.x >> 6);
    for (int k0 = 0; k0 < nlist; k0 += 64 * nwaves) {
    // the test, 64 listed points per wave at once; a wave's 64 are spread over the whole list (neighbours in the list are
    // neighbours on the chromosome and tend to pass or fail together)
    const int kl = k0 + lane * nwaves + wave;
    const int ci = kl < nlist ? clist[kl] : -1;
    const int co = ci >= 0 ? owner[ci] : -1;
    unsigned long long todo = __ballot(ci >= 0 && state[owner_root(co)] == ST_UNKNOWN);
    while (todo) {
        const int src = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int i = __builtin_amdgcn_readlane(ci, src);                // (wave-uniform from here on)
        const int qi = sv[i], pi = sa[i];
        const int enc = ws.word(i, qi, pi);
        const int qlo = sat_add(qi, -g.eps), qhi = sat_add(qi, g.eps);
        const int pbeg = pi & ~(g.peps - 1), pend = pbeg + g.peps, pend2 = pend + g.peps;      // strips are aligned blocks of sp
        const int plo = pi - g.peps, phi = pi + g.peps;
        int ja, jb;
        if (enc < 0 && ((unsigned)enc & K2H_NONE) != K2H_NONE) {
            ja = i - (int)((unsigned)enc & K2H_MASK); jb = i + (int)(((unsigned)enc >> K2H_BITS) & K2H_MASK);
        } else {
            const int s = strip_of(g, pi);
            const int b = strip_start[s], e = strip_start[s + 1];
            ja = s > 0 ? lower_bound_4(sv, strip_start[s - 1], b, qlo) : b;
            jb = lower_bound_4(sv, e, s + 1 < g.S ? strip_start[s + 2] : e, qlo);
        }
        int rr[4] = {-1, -1, -1, -1};
        int kk[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX};
        int nr = 0;
        bool any_u = false, overflow = false;
        auto see = [&](int r) {                                          // (every lane keeps the same list)
            for (int q = 0; q < 4; ++q) if (rr[q] == r) return;
            if (nr == 4) { overflow = true; return; }
            const int key = compkey[r];
            int q = nr++;
            while (q > 0 && kk[q - 1] > key) { kk[q] = kk[q - 1]; rr[q] = rr[q - 1]; --q; }
            kk[q] = key; rr[q] = r;
            if (state[r] == ST_UNKNOWN) any_u = true;
        };
        // walk w: 0 = own strip downwards from i - 1, 1 = own strip upwards from i + 1, 2 = one strip below from ja, 3 = one above from jb
        const int start[4] = {i - 1, i + 1, ja, jb};
        auto more = [&](int w, int q, int p) {
            return w == 0 ? ((p >= pbeg) & (q >= qlo)) : w == 1 ? ((p < pend) & (q <= qhi)) : w == 2 ? ((p < pbeg) & (q <= qhi)) : ((p < pend2) & (q <= qhi));
        };
        auto acc = [&](int w, int p) { return w == 2 ? p >= plo : (w == 3 ? p <= phi : true); };
        int q0[4], p0[4], r0[4];
        bool in0[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const int j = w == 0 ? start[w] - lane : start[w] + lane;
            in0[w] = j >= 0 && j < M;
            q0[w] = in0[w] ? sv[j] : 0; p0[w] = in0[w] ? sa[j] : 0; r0[w] = in0[w] ? root[j] : -1;
        }
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            // one round: 64 candidates -> true if the walk goes on behind them
            auto round = [&](bool in, int q, int p, int r) {
                const bool ok = in && more(w, q, p);
                const int rv = (ok && acc(w, p)) ? r : -1;
                unsigned long long pending = __ballot(rv >= 0);
                while (pending) {
                    const int R = __builtin_amdgcn_readlane(rv, __ffsll((long long)pending) - 1);
                    pending &= ~__ballot(rv == R);
                    see(R);
                }
                return __ballot(ok) == ~0ull;
            };
            bool on = round(in0[w], q0[w], p0[w], r0[w]);
            // a window that is longer than the first 64 (dense data: up to thousands): four rounds' loads in flight together
            for (int rnd = 1; on; rnd += 4) {
                int q[4], p[4], r[4];
                bool in[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = w == 0 ? start[w] - lane - 64 * (rnd + u) : start[w] + lane + 64 * (rnd + u);
                    in[u] = j >= 0 && j < M;
                    q[u] = in[u] ? sv[j] : 0; p[u] = in[u] ? sa[j] : 0; r[u] = in[u] ? root[j] : -1;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) if (on) on = round(in[u], q[u], p[u], r[u]);
            }
        }
        if (lane == 0) {
            if (overflow) atomicExch(&counters[CTR_OVERFLOW], 1);
            if (any_u) {
                const int idx = atomicAdd(&counters[CTR_NREC], 1);
                if (idx >= rec_cap) atomicExch(&counters[CTR_OVERFLOW], 2);
                else {
                    Rec rec; rec.pt = i;
                    for (int q = 0; q < 4; ++q) rec.r[q] = rr[q];
                    recs[idx] = rec;
                }
            }
        }
    }
    }
}

// one workgroup; rounds until every uncertain component is decided.  lo = borders surely
// available (every lower-key adjacent component dead), hi = possibly available (none live).
__global__ void __launch_bounds__(1024)
k_resolve_release(int minPts, const int* __restrict__ ncore, const int* __restrict__ usize, int* state,
                  const int* __restrict__ ulist, const Rec* __restrict__ recs, int* lo, int* hi,
                  const int* __restrict__ counters)
{
    const int nU = counters[CTR_NU];
    if (nU == 0) return;
    const int nrec = counters[CTR_NREC];
    __shared__ int remaining;
    for (;;) {
        // uncontested borders are always available to their only adjacent component
        for (int u = threadIdx.x; u < nU; u += blockDim.x) { int c = ulist[u]; lo[c] = usize[c]; hi[c] = usize[c]; }
        if (threadIdx.x == 0) remaining = 0;
        __syncthreads();
        for (int q = threadIdx.x; q < nrec; q += blockDim.x) {
            const Rec rec = recs[q];
            bool allDead = true, noneLive = true;
            for (int k = 0; k < 4; ++k) {
                int c = rec.r[k];
                if (c < 0) break;
                int st = state[c];
                if (st == ST_UNKNOWN) {
                    if (allDead) atomicAdd(&lo[c], 1);
                    if (noneLive) atomicAdd(&hi[c], 1);
                }
                if (st != ST_DEAD) allDead = false;
                if (st == ST_LIVE) noneLive = false;
            }
        }
        __syncthreads();
        for (int u = threadIdx.x; u < nU; u += blockDim.x) {
            int c = ulist[u];
            if (state[c] != ST_UNKNOWN) continue;
            if (ncore[c] + lo[c] >= minPts) state[c] = ST_LIVE;
            else if (ncore[c] + hi[c] < minPts) state[c] = ST_DEAD;
            else atomicAdd(&remaining, 1);
        }
        __syncthreads();
        if (remaining == 0) break;
        __syncthreads();
    }
}

__global__ void k_apply_records(const Rec* __restrict__ recs, const int* __restrict__ state,
                                int* __restrict__ owner, const int* __restrict__ counters)
{
    const int nrec = counters[CTR_NREC];
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nrec) return;
    const Rec rec = recs[q];
    int o = -1;
    for (int k = 0; k < 4; ++k) {
        int c = rec.r[k];
        if (c < 0) break;
        if (state[c] != ST_DEAD) { o = c; break; }
    }
    owner[rec.pt] = o;
}

// ------------------------------------------------------------------------------------------
// K5: cluster ids = rank of the component key among the kept components; labels; table
// ------------------------------------------------------------------------------------------
// the same three per-component steps on the root list of k_flatten (K entries instead of a test per PET)
__global__ void k_mark_uncertain_l(GridParams g, const int* __restrict__ rootlist, const int* __restrict__ ncore, const int* __restrict__ bsize,
                                   int* __restrict__ state, int* __restrict__ ulist, int* __restrict__ counters)
{
    const int K = counters[CTR_NROOT];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
        const int i = rootlist[k];
        if (ncore[i] + bsize[i] < g.minPts) {
            state[i] = ST_UNKNOWN;
            ulist[atomicAdd(&counters[CTR_NU], 1)] = i;
        }
    }
}
// The same two steps on a BITMAP of the keys (keys are input rows, 0 .. n-1): one bit per key instead of one int, the
// ranks come from an exclusive scan over the words' popcounts (n / 32 elements instead of n + 1) plus a popcount
// inside the key's word -- the per-run clearing and the scan shrink 32-fold.
struct PopcWord { __host__ __device__ int operator()(unsigned w) const { return __popc(w); } };
__global__ void k_rank_bits_l(GridParams g, const int* __restrict__ rootlist, const int* __restrict__ counters,
                              const int* __restrict__ compkey, const int* __restrict__ state, unsigned* __restrict__ bits,
                              const Rec* __restrict__ recs /* or null */, int* __restrict__ owner)
{
    if (recs) {                                         // variant 2: k_apply_records rides along (both only need the final states)
        const int nrec = counters[CTR_NREC];
        for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < nrec; q += gridDim.x * blockDim.x) {
            const Rec rec = recs[q];
            int o = -1;
            for (int k = 0; k < 4; ++k) {
                const int c = rec.r[k];
                if (c < 0) break;
                if (state[c] != ST_DEAD) { o = c; break; }
            }
            owner[rec.pt] = o;
        }
    }
    const int K = counters[CTR_NROOT];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
        const int i = rootlist[k];
        if (g.variant == CL_VARIANT_CDBSCAN2 && state[i] == ST_DEAD) continue;      // cDBSCAN2.py:183-185 / cDBSCAN.py:136-152
        const int key = compkey[i];
        atomicOr(&bits[key >> 5], 1u << (key & 31));
    }
}
__global__ void k_rank_flags(GridParams g, const int* __restrict__ strip_start, const int* __restrict__ root,
                             const int* __restrict__ compkey, const int* __restrict__ state, int* __restrict__ flag)
{
    const int M = strip_start[g.S];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    if (root[i] != i) return;
    // variant 2: released components do not consume an id (cDBSCAN2.py:183-185);
    // variant 1: every component consumes one, dropped clusters leave gaps (cDBSCAN.py:136-152)
    if (g.variant == CL_VARIANT_CDBSCAN2 && state[i] == ST_DEAD) return;
    flag[compkey[i]] = 1;
}


// label of every component root (-1 = not kept), so that k_final_labels needs ONE gather per PET
// instead of the chain owner -> compkey -> rank (+ state / sizes)
__global__ void k_root_labels(GridParams g, const int* __restrict__ strip_start, const int* __restrict__ root,
                              const int* __restrict__ compkey, const int* __restrict__ ncore, const int* __restrict__ bsize,
                              const int* __restrict__ state, const int* __restrict__ rankscan, int* __restrict__ rlabel)
{
    const int M = strip_start[g.S];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    if (root[i] != i) return;
    const bool keep = (g.variant == CL_VARIANT_CDBSCAN2) ? (state[i] != ST_DEAD)
                                                         : (ncore[i] + bsize[i] >= g.minPts);   // cDBSCAN.py:149-152
    rlabel[i] = keep ? rankscan[compkey[i]] : -1;
}

__global__ void k_init_table(Table t, const int* __restrict__ rankscan, int n)
{
    const int K = rankscan[n];      // total number of ids handed out
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    t.count[k] = 0; t.minx[k] = INT_MAX; t.maxx[k] = INT_MIN; t.miny[k] = INT_MAX; t.maxy[k] = INT_MIN;
}

// Exclusive scan of the popcounts of the key bitmap's words, two-level: wloc[w] = set bits in front of word w inside its
// workgroup's 1024 words, wboff[block] by the last workgroup to finish (scan_tail_last_block; wboff[gridDim.x] = the number of
// ids).  One launch where rocPRIM's scan was three (a fill, the look-back state, the scan).
__global__ void __launch_bounds__(256)
k_rank_scan(int nw, const unsigned* __restrict__ bits, int* __restrict__ wloc, int* __restrict__ bsum, int* __restrict__ wboff, int* ticket)
{
    __shared__ int red[4];
    const int w0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    int c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = w0 + k < nw ? __popc(bits[w0 + k]) : 0;
    const int mine = c[0] + c[1] + c[2] + c[3];
    int tot;
    int run = wg256_inclusive_scan(mine, red, tot) - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k) { if (w0 + k < nw) wloc[w0 + k] = run; run += c[k]; }
    scan_tail_last_block(tot, bsum, wboff, ticket, red);
}

__global__ void k_root_labels_bits_l(GridParams g, const int* __restrict__ rootlist, const int* __restrict__ counters,
                                     const int* __restrict__ compkey, const int* __restrict__ ncore, const int* __restrict__ bsize,
                                     const int* __restrict__ state, const unsigned* __restrict__ bits,
                                     const int* __restrict__ wloc, const int* __restrict__ wboff /* ranks of the bitmap's words, two-level (k_rank_scan) */,
                                     int* __restrict__ rlabel, Table t, int nblkw, int* __restrict__ hdr, const int* __restrict__ d_M)
{
    const int ids = wboff[nblkw];                        // total number of ids handed out
    if (blockIdx.x == 0 && threadIdx.x == 0) {          // k_pack_header rides along (nothing behind this kernel raises a flag)
        hdr[0] = ids; hdr[1] = counters[CTR_OVERFLOW]; hdr[2] = d_M[0]; hdr[3] = 0; hdr[4] = -1; hdr[5] = 0; hdr[6] = 0;      // (6: labelled PETs handed out as pairs)
    }
    const int K = counters[CTR_NROOT];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
        const int i = rootlist[k];
        const bool keep = (g.variant == CL_VARIANT_CDBSCAN2) ? (state[i] != ST_DEAD)
                                                             : (ncore[i] + bsize[i] >= g.minPts);   // cDBSCAN.py:149-152
        const int key = compkey[i];
        const int w = key >> 5;
        rlabel[i] = keep ? wloc[w] + wboff[w >> 10] + __popc(bits[w] & ((1u << (key & 31)) - 1u)) : -1;
    }
    // k_init_table rides along: the rows of the ids handed out
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < ids; k += gridDim.x * blockDim.x) {
        t.count[k] = 0; t.minx[k] = INT_MAX; t.maxx[k] = INT_MIN; t.miny[k] = INT_MAX; t.maxy[k] = INT_MIN;
    }
}

// Labels are scattered to input-row order; the cluster table (pipe.py:78-102) is reduced
// per wave first: sorted order keeps a cluster's PETs in neighbouring lanes, so a wave
// usually carries a handful of labels and a giant cluster costs 5 atomics per wave, not
// 5 per PET.
#define FINAL_CHUNKS 4          // PETs per workgroup = FINAL_CHUNKS * BIGTPB: one LDS table, one flush
// slab[i] = label of sorted position i (coalesced; what the distance statistics K7 read); labels[row] (the scatter to
// input-row order, 4-byte stores all over the array) only when the caller wants row-aligned labels
__global__ void __launch_bounds__(BIGTPB)
k_final_labels(GridParams g, const int* __restrict__ strip_start, const int* __restrict__ sv,
               const int* __restrict__ sa, const u32* __restrict__ srow, const int* __restrict__ owner,
               const int* __restrict__ rlabel, int* __restrict__ labels, int* __restrict__ slab, Table t)
{
    __shared__ TableLds h;
    table_lds_init(h);
    const int M = strip_start[g.S];
    // the owner -> label round trips of a thread's FINAL_CHUNKS PETs are all in flight before the first of them is used
    // (the kernel is bound by those dependent gathers, not by bytes)
    int idx[FINAL_CHUNKS], own[FINAL_CHUNKS], lab[FINAL_CHUNKS], x[FINAL_CHUNKS], y[FINAL_CHUNKS];
#pragma unroll
    for (int ch = 0; ch < FINAL_CHUNKS; ++ch) {
        idx[ch] = (blockIdx.x * FINAL_CHUNKS + ch) * BIGTPB + threadIdx.x;
        own[ch] = idx[ch] < M ? owner_root(owner[idx[ch]]) : -1;
    }
#pragma unroll
    for (int ch = 0; ch < FINAL_CHUNKS; ++ch) lab[ch] = own[ch] >= 0 ? rlabel[own[ch]] : -1;
#ifdef CLOOPS_DEVEL
    if (g.dbg2 & 16) { if (lab[0] == 123456789) slab[0] = 0; return; }      // developer ablation: owner -> label only
#endif
#pragma unroll
    for (int ch = 0; ch < FINAL_CHUNKS; ++ch) {
        const int i = idx[ch];
        x[ch] = 0; y[ch] = 0;
        if (i < M) {
            slab[i] = lab[ch];
            if (labels) labels[srow[i]] = lab[ch];
            if (lab[ch] >= 0) {                          // noise has no box: its coordinates are never loaded
                // X = (v - a) / 2, Y = (v + a) / 2 exactly (v and a have equal parity)
                const int spv = sa[i];
                int pp = ((spv >> g.rbits) + g.s0) * g.eps + (spv & (g.peps - 1)) + g.A0, qq = sv[i] + g.V0;
                int a = g.swap ? qq : pp, v = g.swap ? pp : qq;
                x[ch] = (v - a) / 2; y[ch] = (v + a) / 2;
            }
        }
    }
#ifdef CLOOPS_DEVEL
    if (g.dbg2 & 32) return;                             // developer ablation: no cluster table
#endif
#pragma unroll
    for (int ch = 0; ch < FINAL_CHUNKS; ++ch) table_accumulate(t, h, lab[ch], x[ch], y[ch]);
    table_flush(t, h);
}


// The candidate buffer of a sweep (K10) holds every inter-ligation box of every step: it grows on demand (contents kept) --
// before a step is enqueued it has room for all the boxes the step can produce (one per cluster id, at most n / minPts).
int ensure_cand_capacity(cl_chrom* c, long long need)
{
    if (need <= c->cand_cap) return CL_OK;
    const long long cap = std::max<long long>(std::max<long long>(need, 2 * c->cand_cap), 1 << 20);
    DevBuf nb, ns;
    int rc;
    if ((rc = nb.ensure((size_t)cap * 16)) || (rc = ns.ensure((size_t)cap * 4))) { nb.release(); ns.release(); return rc; }
    if (c->cand_n > 0) {
        HIP_TRY(hipMemcpyAsync(nb.p, c->cand_box.p, (size_t)c->cand_n * 16, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipMemcpyAsync(ns.p, c->cand_step.p, (size_t)c->cand_n * 4, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    c->cand_box.release(); c->cand_step.release();
    c->cand_box = nb; c->cand_step = ns;
    c->cand_cap = cap;
    return CL_OK;
}

static void free_chrom(cl_chrom* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    DevBuf* bufs[] = {&c->keys_in, &c->keys_out, &c->vals_in, &c->vals_out, &c->sort_tmp, &c->scan_tmp, &c->qb_key, &c->qb_val, &c->sv, &c->sa,
                      &c->strip, &c->cnt, &c->parent, &c->root, &c->head, &c->headidx, &c->cellfirst, &c->compkey,
                      &c->ncore, &c->bsize, &c->owner, &c->state, &c->flag, &c->rankscan, &c->slot[0].labels, &c->slot[0].table, &c->slot[1].labels, &c->slot[1].table, &c->slot[0].slab, &c->slot[1].slab, &c->slot[0].d_step, &c->slot[1].d_step, &c->slot[0].pairs, &c->slot[1].pairs, &c->hdr, &c->k7_cls, &c->k7_parts, &c->sig_tx, &c->sig_ty, &c->sig_tmp, &c->sig_sorttmp, &c->sig_m, &c->sig_win, &c->sig_out,
                      &c->ulist, &c->lo, &c->hi, &c->recs, &c->counters, &c->chainflag, &c->chainhead, &c->usize, &c->b_cstart, &c->b_ckey, &c->b_nb, &c->b_cx, &c->b_cy, &c->tile_s0, &c->bq, &c->bsp, &c->brow, &c->bstrip, &c->btile, &c->sel_tmp, &c->cand_box, &c->cand_step, &c->cand_keep, &c->cand_out, &c->dhist,
                      &c->rc_cnt, &c->rc_pre, &c->rc_poff, &c->rc_dpre, &c->rc_D, &c->rc_blen, &c->rootlist, &c->cflag8, &c->blk_tmp,
                      &c->l_mask, &c->l_rank, &c->l_blk, &c->l_cstrip, &c->l_wpos, &c->l_wenc, &c->l_dist, &c->l_aux, &c->l_tab, &c->l_fix, &c->bkey,
                      &c->fq, &c->fsp, &c->frow, &c->fstrip};
    for (DevBuf* b : bufs) b->release();
    c->arena.release();                                  // (after its slices have been dropped)
    if (c->own_xy) { if (c->d_x) (void)hipFree(c->d_x); if (c->d_y) (void)hipFree(c->d_y); }
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    for (auto& sl : c->slot) {
        if (sl.h_boxes) (void)hipHostFree(sl.h_boxes);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
        if (sl.ev_copied) (void)hipEventDestroy(sl.ev_copied);
        if (c->ev_ready) for (auto& e : sl.ev) (void)hipEventDestroy(e);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

extern "C" void cl_chrom_destroy(cl_chrom* c) { free_chrom(c); }
extern "C" int64_t cl_chrom_size(const cl_chrom* c) { return c ? c->n : -1; }
extern "C" void cl_set_profiling(cl_chrom* c, int enabled) { if (c) c->profiling = enabled != 0; }
extern "C" void cl_set_layout_reuse(cl_chrom* c, int enabled) { if (c) { c->reuse_layout = enabled != 0; c->base.valid = false; c->rc.valid = false; } }
extern "C" void cl_set_count_reuse(cl_chrom* c, int enabled) { if (c) { c->reuse_counts = enabled != 0; c->rc.valid = false; } }
extern "C" void cl_set_traversal(cl_chrom* c, int level) { if (c) { c->traversal = level < 0 ? 0 : (level > 4 ? 4 : level); c->rc.valid = false; } }
extern "C" int cl_last_region_mode(const cl_chrom* c) { return c ? c->last_k2_mode : 0; }
// Forget everything the handle has DERIVED from its rows -- the q index, the fine layout, the base layout of the last eps, the
// cached neighbour counts -- and keep every allocation: the next run pays what the first run on a freshly uploaded dataset pays
// for its orders, without the allocations of a new handle (bench.py `cold_sweep_s`: a dataset is swept once, pipe.py:247-275).
extern "C" int cl_chrom_drop_indexes(cl_chrom* c)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    if (c->slot[0].pending || c->slot[1].pending) return fail(CL_ERR_ARG, "cl_chrom_drop_indexes: a run is in flight");
    c->qindex_layout = -1;
    c->fine_valid_w = 0; c->fine_layout = -1;
    c->base.valid = false;
    c->rc.valid = false;
    return CL_OK;
}
extern "C" void cl_set_count_floor(cl_chrom* c, int32_t min_pts)
{
    if (!c) return;
    c->count_floor = min_pts > 0 ? min_pts : 0;
    for (auto& w : c->count_tmask) w = 0;
}
extern "C" int cl_chrom_set_stream(cl_chrom* c, void* stream)
{
    if (!c || !stream || c->own_stream) return fail(CL_ERR_ARG, "cl_chrom_set_stream: needs a handle on a caller's stream and a stream");
    if (c->slot[0].pending || c->slot[1].pending) return fail(CL_ERR_ARG, "cl_chrom_set_stream: a run is in flight");
    if ((hipStream_t)stream == c->stream) return CL_OK;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));             // (everything the handle enqueued there is done: the new stream starts clean)
    c->stream = (hipStream_t)stream;
    c->shared_copy = nullptr;                             // the copy stream belongs to the compute stream: looked up again when needed
    c->copy_mode = library_made_stream(c->stream) ? 2 : 1;
    return CL_OK;
}

extern "C" void cl_set_eps_list(cl_chrom* c, const int32_t* eps, int32_t n)
{
    if (!c) return;
    long long gcd = 0, mx = 0;
    int distinct = 0;
    for (int k = 0; eps && k < n; ++k) {
        if (eps[k] <= 0) { gcd = 0; distinct = 0; break; }
        bool seen = false;
        for (int j = 0; j < k; ++j) seen |= eps[j] == eps[k];
        distinct += seen ? 0 : 1;
        long long a = eps[k], b = gcd;
        while (b) { const long long t = a % b; a = b; b = t; }
        gcd = a; mx = std::max<long long>(mx, eps[k]);
    }
    // worth a layout of its own: several eps values, strips of the common width not absurdly narrow, at most CL_FINE_KMAX runs per strip
    c->fine_w = (distinct >= 2 && gcd >= 16 && mx / gcd <= CL_FINE_KMAX) ? (int)gcd : 0;
}

extern "C" void cl_set_count_thresholds(cl_chrom* c, const int32_t* min_pts, int32_t n)
{
    if (!c) return;
    c->count_floor = 0;
    for (auto& w : c->count_tmask) w = 0;
    for (int k = 0; min_pts && k < n; ++k)
        if (min_pts[k] >= 2 && min_pts[k] <= 128) c->count_tmask[(min_pts[k] - 1) >> 5] |= 1u << ((min_pts[k] - 1) & 31);
}
extern "C" void cl_set_sort_index(cl_chrom* c, int mode) { if (c) c->sort_index_mode = mode > 0 ? 1 : (mode < 0 ? -1 : 0); }
// One call for a sweep (cLoops/pipe.py:241-281: `for ep in eps: for m in minPts:`): everything the handle can prepare from knowing
// both lists -- the q index from the first sort on and one fine sort for all layouts (several eps), counts bracketed for the minPts
// list.  n_eps = n_min_pts = 0 ends the plan (a later one-off run serves itself).
extern "C" int cl_sweep_plan(cl_chrom* c, const int32_t* eps, int32_t n_eps, const int32_t* min_pts, int32_t n_min_pts)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    if (n_eps < 0 || n_min_pts < 0 || (n_eps > 0 && !eps) || (n_min_pts > 0 && !min_pts)) return fail(CL_ERR_ARG, "cl_sweep_plan: bad lists");
    for (int k = 0; k < n_eps; ++k) if (eps[k] <= 0) return fail(CL_ERR_ARG, "cl_sweep_plan: eps must be positive");
    for (int k = 0; k < n_min_pts; ++k) if (min_pts[k] <= 0) return fail(CL_ERR_ARG, "cl_sweep_plan: minPts must be positive");
    if (c->slot[0].pending || c->slot[1].pending) return fail(CL_ERR_ARG, "cl_sweep_plan: a run is in flight");
    // (layout reuse, count reuse and the traversal level are left as they are: all three are on / 4 by default, and a caller that
    //  switched one off -- the tests of the older levels, the measurements without re-use -- meant it)
    cl_set_count_thresholds(c, min_pts, n_min_pts);
    int distinct = 0;
    for (int k = 0; k < n_eps; ++k) { bool seen = false; for (int j = 0; j < k; ++j) seen |= eps[j] == eps[k]; distinct += seen ? 0 : 1; }
    if (distinct > 1) { cl_set_sort_index(c, 1); cl_set_eps_list(c, eps, n_eps); }
    else cl_set_eps_list(c, nullptr, 0);
    return CL_OK;
}
extern "C" int cl_get_timing(const cl_chrom* c, cl_timing* out)
{
    if (!c || !out) return fail(CL_ERR_ARG, "cl_get_timing: null argument");
    *out = c->timing;
    return CL_OK;
}
extern "C" const int32_t* cl_labels_device(const cl_chrom* c)
{
    return (c && c->last_slot >= 0 && c->slot[c->last_slot].rows_valid) ? (const int32_t*)c->slot[c->last_slot].labels.p : nullptr;
}
extern "C" void cl_set_device_labels(cl_chrom* c, int enabled) { if (c) c->device_labels = enabled != 0; }
extern "C" void cl_set_table_export(cl_chrom* c, int enabled) { if (c) c->export_table = enabled != 0; }


__global__ void k_init_pads(int* __restrict__ svbuf, int* __restrict__ sabuf, long long n)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 2 * SORT_PAD) return;
    const bool left = k < SORT_PAD;
    const long long idx = left ? k : (long long)SORT_PAD + n + (k - SORT_PAD);
    svbuf[idx] = left ? 0 : INT_MAX;
    sabuf[idx] = left ? INT_MIN : INT_MAX;
}

static int reserve_workspace(cl_chrom* c);
#ifdef CLOOPS_DEVEL
// developer build: CLOOPS_SKIP=<mask> leaves kernels of a run out (results invalid; what a kernel costs the SWEEP, not its own time)
static int skip_mask() { static const int m = getenv("CLOOPS_SKIP") ? atoi(getenv("CLOOPS_SKIP")) : 0; return m; }
#define SKIP(bit) (skip_mask() & (bit))
#else
#define SKIP(bit) 0
#endif

extern "C" int cl_chrom_create(int device, void* stream, const int32_t* x, const int32_t* y, int64_t n,
                               int on_device, cl_chrom** out)
{
    if (!out) return fail(CL_ERR_ARG, "cl_chrom_create: out is null");
    *out = nullptr;
    if (n < 0 || n >= (1LL << 31) - 1024) return fail(CL_ERR_ARG, "cl_chrom_create: n out of range (0 .. 2^31 - 1024)");
    if (n > 0 && (!x || !y)) return fail(CL_ERR_ARG, "cl_chrom_create: null coordinates");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(CL_ERR_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(CL_ERR_ARG, "cl_chrom_create: bad device index");
    HIP_TRY(hipSetDevice(device));
    cl_chrom* c = new cl_chrom();
    c->device = device;
    c->n = n;
    int rc = CL_OK;
    do {
        if (stream) c->stream = (hipStream_t)stream;
        else {
            if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { rc = fail(CL_ERR_HIP, "hipStreamCreate"); break; }
            c->own_stream = true;
        }
        // ONE pinned block per handle: 4 KB of scratch, then per result slot the 64-byte header and the step output of a sweep
        // step (hipHostMalloc costs about a millisecond a call: two per slot and handle were half of the first sweep's overhead)
        const size_t step_bytes = ((16 + sizeof(K7Part) + K7_LOGBINS * 8 + K7_FINE * 8 + 255) / 256) * 256;
        if (hipHostMalloc((void**)&c->h_pinned, 4096 + 2 * (256 + step_bytes), hipHostMallocDefault) != hipSuccess) { rc = fail(CL_ERR_HIP, "hipHostMalloc"); break; }
        {
            // copies must not queue up behind the next run's kernels: give their streams the highest priority
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            // A handle with a stream of its own sends the D2H copies of a run through a copy stream (behind an event: they overlap
            // the kernels of its next run).  A handle on a stream the CALLER made -- typically shared by several handles, whose
            // kernels overlap each other anyway -- issues them in that stream: a copy stream per handle waiting on events of
            // other hardware queues is what made the label-copying forms fall off a cliff (23 handles, 8 hardware queues: 3 s
            // per sweep instead of 0.05 s -- head-of-line blocking of the queues the copy streams share).
            c->copy_mode = c->own_stream ? 0 : 1;
            if (!c->own_stream) {
                // ... unless the library made the stream (cl_stream_create): its handles share ONE copy stream
                if (library_made_stream(c->stream)) c->copy_mode = 2;
#ifdef CLOOPS_DEVEL
                if (getenv("CLOOPS_COPY_IN_STREAM")) c->copy_mode = 1;     // developer A/B: the copies in the compute stream itself
#endif
            }
            if (hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
                hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, prio_hi) != hipSuccess) { rc = fail(CL_ERR_HIP, "hipStreamCreate(copy)"); break; }
        }
        bool okslots = true;
        int kslot = 0;
        for (auto& sl : c->slot) {
            okslots = okslots && hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming) == hipSuccess;
            okslots = okslots && hipEventCreateWithFlags(&sl.ev_copied, hipEventDisableTiming) == hipSuccess;
            char* base = (char*)c->h_pinned + 4096 + (size_t)kslot * (256 + step_bytes);
            sl.h_hdr = (int*)base;
            sl.h_step = base + 256;
            ++kslot;
        }
        if (!okslots) { rc = fail(CL_ERR_HIP, "result slot setup"); break; }
        if (on_device) { c->d_x = (int*)x; c->d_y = (int*)y; }
        else if (n > 0) {
            c->own_xy = true;
            if (hipMalloc((void**)&c->d_x, n * 4) != hipSuccess || hipMalloc((void**)&c->d_y, n * 4) != hipSuccess) { rc = fail(CL_ERR_HIP, "hipMalloc X/Y"); break; }
            // (a copy staged through page-locked buffers by the calling thread measured no faster than this plain copy from pageable
            //  memory, 4-5 GB/s either way on the pool's hosts: one thread's memcpy is the limit; a caller that wants PCIe rate passes
            //  page-locked arrays -- cl_host_alloc -- or device pointers)
            if (hipMemcpyAsync(c->d_x, x, n * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                hipMemcpyAsync(c->d_y, y, n * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(CL_ERR_HIP, "hipMemcpy X/Y"); break; }
        }
        if (n > 0) {
            if ((rc = c->counters.ensure(256))) break;
            // (the tickets of the in-kernel scans live behind the counters and must start at zero)
            if (hipMemsetAsync(c->counters.p, 0, 256, c->stream) != hipSuccess) { rc = fail(CL_ERR_HIP, "counters memset"); break; }
            Stats init = {INT_MAX, INT_MIN, INT_MAX, INT_MIN, INT_MAX, INT_MIN, INT_MAX, INT_MIN};
            Stats* hs = (Stats*)c->h_pinned;
            *hs = init;
            if (hipMemcpyAsync(c->counters.p, hs, sizeof(Stats), hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(CL_ERR_HIP, "stats init"); break; }
            int grid = std::min(nblocks(n), 512);
            hipLaunchKernelGGL(k_stats, dim3(grid), dim3(TPB), 0, c->stream, c->d_x, c->d_y, (long long)n, (Stats*)c->counters.p);
            if (hipMemcpyAsync(hs, c->counters.p, sizeof(Stats), hipMemcpyDeviceToHost, c->stream) != hipSuccess) { rc = fail(CL_ERR_HIP, "stats readback"); break; }
            hipError_t e = hipStreamSynchronize(c->stream);
            if (e != hipSuccess) { rc = fail(CL_ERR_HIP, "stats sync", hipGetErrorString(e)); break; }
            c->st = *hs;
            {
                // distance histogram below 65536 -> running sum on the host (see k_dhist)
                // the histogram itself stays on the device: K7 takes the PETs below a cut from it instead of from the rows
                if ((rc = c->dhist.ensure((size_t)(DCUM_BINS + 1) * 4))) break;
                std::vector<int> hh(DCUM_BINS + 1);
                if (hipMemsetAsync(c->dhist.p, 0, (size_t)(DCUM_BINS + 1) * 4, c->stream) != hipSuccess) { rc = fail(CL_ERR_HIP, "dhist memset"); break; }
                hipLaunchKernelGGL(k_dhist, dim3(std::min(nblocks(n), 2048)), dim3(TPB), 0, c->stream, c->d_x, c->d_y, (long long)n, c->dhist.as<int>());
                if (hipMemcpyAsync(hh.data(), c->dhist.p, (size_t)(DCUM_BINS + 1) * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                    hipStreamSynchronize(c->stream) != hipSuccess) { rc = fail(CL_ERR_HIP, "dhist readback"); break; }
                c->dcum.assign(DCUM_BINS, 0);
                c->n_neg = hh[DCUM_BINS];
                long long run = c->n_neg;
                for (int k = 0; k < DCUM_BINS; ++k) { c->dcum[k] = run; run += hh[k]; }      // dcum[k] = #(d < k)
            }
            const int LIM = 1 << 29;
            if (c->st.xmin <= -LIM || c->st.xmax >= LIM || c->st.ymin <= -LIM || c->st.ymax >= LIM) {
                rc = fail(CL_ERR_DOMAIN, "coordinates must satisfy |X|,|Y| < 2^29");
                break;
            }
        }
    } while (0);
    if (rc != CL_OK) { std::string keep = g_err; free_chrom(c); g_err = keep; return rc; }
    (void)reserve_workspace(c);                          // best effort: without it the buffers are allocated one by one at the first run
    *out = c;
    return CL_OK;
}

// ---- a chromosome made of rows of a resident one (scripts/jd2saturation:32-55 re-samples a .jd) ------------------------
__global__ void k_gather_rows(const int* __restrict__ X, const int* __restrict__ Y, const int* __restrict__ rows, int m,
                              int* __restrict__ xo, int* __restrict__ yo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const int r = rows[i];
    xo[i] = X[r]; yo[i] = Y[r];
}
extern "C" int cl_chrom_subsample(cl_chrom* src, const int64_t* rows, int64_t m, cl_chrom** out)
{
    if (!src || !out) return fail(CL_ERR_ARG, "cl_chrom_subsample: null argument");
    *out = nullptr;
    if (m < 0 || m >= (1LL << 31) - 1024 || (m > 0 && !rows)) return fail(CL_ERR_ARG, "cl_chrom_subsample: bad row list");
    if (src->enq != src->deq) return fail(CL_ERR_ARG, "cl_chrom_subsample: asynchronous runs still in flight on the source, call cl_wait first");
    HIP_TRY(hipSetDevice(src->device));
    std::vector<int> r32((size_t)m);
    for (int64_t i = 0; i < m; ++i) {
        if (rows[i] < 0 || rows[i] >= src->n) return fail(CL_ERR_ARG, "cl_chrom_subsample: row index out of range");
        r32[(size_t)i] = (int)rows[i];
    }
    int *dx = nullptr, *dy = nullptr, *dr = nullptr;
    int rc = CL_OK;
    if (m > 0) {
        if (hipMalloc((void**)&dx, (size_t)m * 4) != hipSuccess || hipMalloc((void**)&dy, (size_t)m * 4) != hipSuccess ||
            hipMalloc((void**)&dr, (size_t)m * 4) != hipSuccess) rc = fail(CL_ERR_HIP, "hipMalloc (subsample)");
        if (rc == CL_OK && hipMemcpyAsync(dr, r32.data(), (size_t)m * 4, hipMemcpyHostToDevice, src->stream) != hipSuccess) rc = fail(CL_ERR_HIP, "hipMemcpy (row list)");
        if (rc == CL_OK) {
            hipLaunchKernelGGL(k_gather_rows, dim3(nblocks(m)), dim3(TPB), 0, src->stream, (const int*)src->d_x, (const int*)src->d_y, (const int*)dr, (int)m, dx, dy);
            if (hipStreamSynchronize(src->stream) != hipSuccess) rc = fail(CL_ERR_HIP, "gather (subsample)");
        }
        if (dr) (void)hipFree(dr);
    }
    // the coordinates are on the device already: the new handle takes them over (statistics, distance histogram, workspace as usual)
    if (rc == CL_OK) rc = cl_chrom_create(src->device, nullptr, dx, dy, m, 1, out);
    if (rc != CL_OK) { std::string keep = g_err; if (dx) (void)hipFree(dx); if (dy) (void)hipFree(dy); g_err = keep; return rc; }
    (*out)->own_xy = true;
    return CL_OK;
}

// ---- cut filter as a stream compaction of the base layout ------------------------------------------------
// pipe.py:59-62 keeps d = Y - X >= cut; in the strip layout d = q + V0, so the test reads the sorted q alone.
#define CMP_TPB 256
#define CMP_PER 8                       // elements per thread
#define CMP_BLOCK (CMP_TPB * CMP_PER)
// Stable compaction of the base layout by STRIPS (its in-strip coordinate is q = Y - X - V0, so the rows a cut removes are
// a PREFIX of every strip): the kept length of every strip by one bisection per strip, an exclusive scan over the S strips
// -- which IS the new strip table -- and one copy pass (12 B/PET read, 12 B per kept PET written) that also leaves the tile
// table, the sentinels behind the last kept PET and M.  (Round 2 first did it by flags: per-block counts over all PETs, a
// scan over the blocks, a ballot-ranked scatter, then bisections of the compacted array for the strip table -- 5 launches.)
// Count cache (see run_sort_and_count): pre_out (a run whose K2 words are kept) = the number of PETs the cut removes from
// every strip; pre_ref + dpre_out (a run that re-uses kept words) = how many MORE PETs this cut removes from the strip than
// the cut of the run that made the words (negative: fewer) -- what the hint fields of a word shift by.
// ... and for such a run blen_out[s] = how many of the strip's kept PETs lie in the cut band (q < bandq) / within eps of it
// (q < bandq + eps): the work list and the staging ranges of k_region_band.
__global__ void __launch_bounds__(256)
k_cut_strips(int S, int thr, const int* __restrict__ bstrip, const int* __restrict__ bq,
             int* __restrict__ sloc /* [S+1]: the new strip table, exclusive inside the workgroup's 256 strips */,
             int* __restrict__ bsum, int* __restrict__ boff /* [gridDim.x + 1]: ... + the workgroups' offsets */, int* ticket,
             int* __restrict__ src0 /* [S] first kept source index */,
             int* __restrict__ pre_out /* or null */, const int* __restrict__ pre_ref /* or null */,
             int* __restrict__ dpre_out /* with pre_ref */, int2* __restrict__ blen_out /* with pre_ref */, int bandq, int eps,
             int* __restrict__ clr /* or null */, int nclr, int* __restrict__ counters,
             int4* __restrict__ tab_out /* or null: per strip {first kept base index, end of its cut band, PETs the cut removes from it, 0}: what the
                                           list kernels need to apply the cut to the BASE layout without a copy (k_lists.hip) */,
             const u32* __restrict__ brow /* with fix_out */, GridParams g, int2* __restrict__ fix_out /* or null (variant 2): the strip's BOUNDARY cell */)
{
    __shared__ int red[4];
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (clr) {
        // the run's first kernel also clears its key bitmap and counters (what k_init_flags does for a run without a cut)
        for (int u = s; u < nclr; u += gridDim.x * blockDim.x) clr[u] = 0;
        if (s < 16) counters[s] = 0;
    }
    int kv = 0;                                          // kept length of the strip (0 for s >= S)
    if (s == S) { if (pre_out) pre_out[S] = 0; if (pre_ref) dpre_out[S] = 0; }
    if (s < S) {
        const int b = bstrip[s];
        int lo = b;
        const int e = bstrip[s + 1];
        int hi = e;
        if (!pre_ref) {
            while (lo < hi) { const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1); if (bq[mid] < thr) lo = mid + 1; else hi = mid; }
        } else {
            // a run that re-uses kept words also needs the ends of its cut band (q < bandq) and of the band's neighbourhood
            // (q < bandq + eps) in the strip: thr <= bandq, so all three are lower bounds over the whole strip -- three independent
            // bisections advance together, their loads in flight at the same time (one after the other they were three dependent
            // chains of ~8 round trips: 20 us against 13 us for the kernel on chr1)
            int l1 = b, h1 = e, l2 = b, h2 = e;
            const int q2 = bandq + eps;
            while ((lo < hi) | (l1 < h1) | (l2 < h2)) {
                const int m0 = (int)(((unsigned)lo + (unsigned)hi) >> 1), m1 = (int)(((unsigned)l1 + (unsigned)h1) >> 1), m2 = (int)(((unsigned)l2 + (unsigned)h2) >> 1);
                const int v0 = bq[m0], v1 = bq[m1], v2 = bq[m2];      // (a finished search reads bq[its answer]: at most bq[e], inside the padded array)
                if (lo < hi) { if (v0 < thr) lo = m0 + 1; else hi = m0; }
                if (l1 < h1) { if (v1 < bandq) l1 = m1 + 1; else h1 = m1; }
                if (l2 < h2) { if (v2 < q2) l2 = m2 + 1; else h2 = m2; }
            }
            dpre_out[s] = (lo - b) - pre_ref[s];
            blen_out[s] = make_int2(l1 - lo, l2 - lo);
        }
        kv = e - lo;
        src0[s] = lo;
        if (pre_out) pre_out[s] = lo - b;
        if (tab_out) tab_out[s] = make_int4(lo, pre_ref ? lo + blen_out[s].x : lo, lo - b, 0);
        if (fix_out) {
            // variant 2: the rotated cell (strip, q / eps) that the cut goes THROUGH keeps only its PETs with q >= the threshold; its
            // smallest input row (cDBSCAN2.py:117) is taken over those -- every other cell keeps the minimum of the base layout
            // (bkey, once per eps).  {end of that cell (base index), its minimum}; no such cell: {lo, -}.
            int2 fx = make_int2(lo, INT_MAX);
            if (lo < e && lo > b) {
                const int q0 = div_eps(g, bq[lo]) * g.eps;
                if (bq[lo - 1] >= q0) {
                    const int qend = q0 + g.eps;
                    int k = lo, m = INT_MAX;
                    bool on = true;
                    while (on) {                          // four PETs per round trip
                        int qv[4]; u32 rv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { const int kk = min(k + u, e - 1); qv[u] = bq[kk]; rv[u] = brow[kk]; }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            if (on && k + u < e && qv[u] < qend) m = min(m, (int)rv[u]); else if (on) { on = false; k += u; }
                        }
                        if (on) k += 4;
                    }
                    fx = make_int2(k, m);
                }
            }
            fix_out[s] = fx;
        }
    }
    if (s == S && tab_out) tab_out[S] = make_int4(bstrip[S], bstrip[S], 0, 0);
    // the exclusive scan of the kept lengths IS the new strip table: inside the workgroup here, the workgroups' offsets by the
    // last workgroup to finish (rocPRIM's scan was three launches: a fill, the look-back state, the scan)
    int tot;
    const int incl = wg256_inclusive_scan(kv, red, tot);
    if (s <= S) sloc[s] = incl - kv;
    scan_tail_last_block(tot, bsum, boff, ticket, red);
}
// Count cache: poff_out (a run whose words are kept, with a cut): poff_out[s] = PETs the cut removes in strips <= s;
// poff_ref + D_out (a run that re-uses kept words): D_out[s] = that number for this cut minus poff_ref[s] -- sorted position
// i of strip s in this layout is position i + D_out[s] of the layout the words were made on (WordSrc).
// BAND (such a run): the first nbb workgroups do K2 on the cut band (cl_band.h: latency-bound work of a few per cent of the
// PETs, hidden behind the copy instead of a launch of its own) -- each of their four waves takes KB_SB strips.
template <bool BAND>
__global__ void __launch_bounds__(CMP_TPB)
k_cut_copy(int n, int S, int rbits, int thr, const int* __restrict__ bq, const int* __restrict__ bsp, const u32* __restrict__ brow,
           const int* __restrict__ src0, const int* __restrict__ sloc, const int* __restrict__ sboffs /* the new strip table, two-level (k_cut_strips) */,
           int* __restrict__ strip_start /* [0..S+1]: the table in one piece, written here for everything that follows */,
           int* __restrict__ sv, int* __restrict__ sa, u32* __restrict__ srow, int* __restrict__ tile_s0, int* __restrict__ d_M,
           int expect_m, int* __restrict__ counters, int* __restrict__ poff_out /* or null */,
           const int* __restrict__ poff_ref /* or null */, int* __restrict__ D_out /* with poff_ref */,
           int nbb, const int2* __restrict__ blen, int* __restrict__ band_words, int eps, int minPts, int dbg)
{
    if (BAND) {
        __shared__ int2 l_band[CMP_TPB / 64][KB_CAP];
        if ((int)blockIdx.x < nbb) {
            const int wv = threadIdx.x >> 6;
            band_wave(blockIdx.x * (CMP_TPB / 64) + wv, threadIdx.x & 63, l_band[wv], S, eps, 1 << rbits, minPts, bq, bsp, src0,
                      sloc, sboffs, blen, band_words, dbg);
            return;
        }
    }
    const int cb = (int)blockIdx.x - (BAND ? nbb : 0), ncb = (int)gridDim.x - (BAND ? nbb : 0);      // copy workgroups
    const int M = sloc[S] + sboffs[S >> 8];
    const int base = cb * CMP_BLOCK;
    // stage by stage over the thread's CMP_PER PETs, so that the loads of a stage are all in flight together (q -> sp / row ->
    // the two per-strip table entries are dependent round trips)
    int q[CMP_PER], sp[CMP_PER], d0[CMP_PER], s0[CMP_PER]; u32 row[CMP_PER]; bool keep[CMP_PER];
#pragma unroll
    for (int k = 0; k < CMP_PER; ++k) {
        const int i = base + k * CMP_TPB + (int)threadIdx.x;
        q[k] = i < n ? bq[i] : INT_MIN;
        keep[k] = i < n && q[k] >= thr;
    }
#pragma unroll
    for (int k = 0; k < CMP_PER; ++k) {
        const int i = base + k * CMP_TPB + (int)threadIdx.x;
        sp[k] = keep[k] ? bsp[i] : 0;
        row[k] = keep[k] ? brow[i] : 0u;
    }
#pragma unroll
    for (int k = 0; k < CMP_PER; ++k) {
        const int st = sp[k] >> rbits;
        d0[k] = keep[k] ? sloc[st] + sboffs[st >> 8] : 0;
        s0[k] = keep[k] ? src0[st] : 0;
    }
#pragma unroll
    for (int k = 0; k < CMP_PER; ++k) {
        if (keep[k]) {
            const int i = base + k * CMP_TPB + (int)threadIdx.x;
            const int dst = d0[k] + (i - s0[k]);
            sv[dst] = q[k]; sa[dst] = sp[k]; srow[dst] = row[k];
            if ((dst & 255) == 0) tile_s0[dst >> 8] = sp[k] >> rbits;
        }
    }
    for (int u = cb * CMP_TPB + (int)threadIdx.x; u <= S; u += ncb * CMP_TPB) {
        const int st = sloc[u] + sboffs[u >> 8];
        strip_start[u] = st;
        if (poff_out || poff_ref) {
            const int po = (u < S ? src0[u] : n) - st;
            if (poff_out) poff_out[u] = po; else D_out[u] = po - poff_ref[u];
        }
    }
    // what k_after_compact did besides the strip table: tiles behind M, sentinels, M itself
    const int t = cb * CMP_TPB + (int)threadIdx.x;
    for (int u = t; u <= n / 256; u += ncb * CMP_TPB) if (u * 256 >= M) tile_s0[u] = S;
    for (int u = t; u < SORT_PAD; u += ncb * CMP_TPB) if (M + u < n) { sv[M + u] = INT_MAX; sa[M + u] = S << rbits; }
    if (t == 0) {
        d_M[0] = M;
        strip_start[S + 1] = n;
        if (expect_m >= 0 && expect_m != M) counters[CTR_OVERFLOW] = 8;      // the host sized the run by a wrong M: fail loudly
    }
}

// K2 on the cut band as a launch of its own (traversal level 4: a run that re-uses counts makes no copy of the layout, so the
// band query has no compaction kernel to ride in); also leaves M, the PETs that enter DBSCAN (the total of the strip scan)
__global__ void __launch_bounds__(CMP_TPB)
k_band(int S, int rbits, int eps, int minPts, const int* __restrict__ bq, const int* __restrict__ bsp, const int* __restrict__ src0,
       const int* __restrict__ sloc, const int* __restrict__ sboffs, const int2* __restrict__ blen, int* __restrict__ band_words,
       int* __restrict__ d_M, int expect_m, int* __restrict__ counters, int dbg)
{
    __shared__ int2 l_band[CMP_TPB / 64][KB_CAP];
    const int wv = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int M = sloc[S] + sboffs[S >> 8];
        d_M[0] = M;
        if (expect_m >= 0 && expect_m != M) counters[CTR_OVERFLOW] = 8;      // the host sized the run by a wrong M: fail loudly
    }
    band_wave<true>(blockIdx.x * (CMP_TPB / 64) + wv, threadIdx.x & 63, l_band[wv], S, eps, 1 << rbits, minPts, bq, bsp, src0, sloc, sboffs, blen, band_words, dbg);
}
__global__ void k_store_m(int S, const int* __restrict__ sloc, const int* __restrict__ sboffs, int* __restrict__ d_M, int expect_m, int* __restrict__ counters)
{
    const int M = sloc[S] + sboffs[S >> 8];
    d_M[0] = M;
    if (expect_m >= 0 && expect_m != M) counters[CTR_OVERFLOW] = 8;
}

// workspace for a run over n rows
// (buffer, bytes) of the per-PET workspace of a handle; the last two rows are the base layout of the current eps, the q index
// and the class array of the step tail (sweeps)
#define WORKSPACE_WANTS(c, n) \
    struct Want { DevBuf* b; size_t bytes; }; \
    const Want wants[] = { \
        {&c->keys_in, n * 8}, {&c->keys_out, n * 8}, {&c->vals_in, n * 4}, {&c->vals_out, n * 4}, \
        {&c->sv, (n + 2 * SORT_PAD) * 4}, {&c->sa, (n + 2 * SORT_PAD) * 4}, {&c->cnt, n * 4}, \
        {&c->parent, n * 4}, {&c->root, n * 4}, {&c->head, n * 4}, {&c->cellfirst, n * 4}, \
        {&c->compkey, n * 4}, {&c->ncore, n * 4}, {&c->bsize, n * 4}, {&c->owner, n * 4}, {&c->state, n * 4}, \
        {&c->flag, (n + 1) * 4}, {&c->rankscan, (n + 1) * 4}, {&c->hdr, 256}, \
        {&c->slot[0].labels, n * 4}, {&c->slot[0].table, (n + 1) * sizeof(cl_box)}, {&c->slot[0].slab, n * 4}, \
        {&c->slot[1].labels, n * 4}, {&c->slot[1].table, (n + 1) * sizeof(cl_box)}, {&c->slot[1].slab, n * 4}, \
        {&c->ulist, n * 4}, {&c->lo, n * 4}, {&c->hi, n * 4}, {&c->recs, n * sizeof(Rec)}, \
        {&c->chainflag, n * 4}, {&c->chainhead, n * 4}, {&c->usize, n * 4}, {&c->tile_s0, (n / 256 + 2) * 4}, \
        {&c->bq, (n + 2 * SORT_PAD) * 4}, {&c->bsp, (n + 2 * SORT_PAD) * 4}, {&c->brow, n * 4}, {&c->btile, (n / 256 + 2) * 4}, \
        {&c->fq, n * 4}, {&c->fsp, n * 4}, {&c->frow, n * 4}, \
        {&c->qb_key, n * 4}, {&c->qb_val, n * 8}, {&c->k7_cls, n + 16}, {&c->rc_cnt, n * 4}, {&c->rootlist, n * 4}, {&c->cflag8, n + 16}, \
        {&c->slot[0].d_step, 16 + sizeof(K7Part) + K7_LOGBINS * 8 + K7_FINE * 8 + K7_BLOCKS * sizeof(K7Part)}, \
        {&c->slot[1].d_step, 16 + sizeof(K7Part) + K7_LOGBINS * 8 + K7_FINE * 8 + K7_BLOCKS * sizeof(K7Part)}, \
    };

// The per-PET workspace of a handle comes out of ONE allocation, reserved when the chromosome is uploaded: mapping the
// ~225 B/PET of device memory is what the first run of a handle used to pay for (a third of the first sweep of a process:
// 0.28 s against 0.20 s), and forty hipMalloc calls per chromosome on top.  A buffer that has to grow later (it does not for a
// fixed chromosome) falls back to an allocation of its own.
// rocPRIM temporary storage for the sorts / scans of a handle with n rows
static int workspace_tmp_sizes(cl_chrom* c, size_t* sort_out, size_t* scan_out)
{
    const size_t n = (size_t)c->n;
    size_t sort_bytes = 0, scan_bytes = 0, scan2 = 0, scan3 = 0, sb2 = 0;
    hipError_t e = rocprim::radix_sort_pairs<SortConfig>(nullptr, sort_bytes, (u64*)nullptr, (u64*)nullptr, (u32*)nullptr, (u32*)nullptr,
                                                         n, 0, 64, c->stream);
    if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs size query", hipGetErrorString(e));
    e = rocprim::radix_sort_pairs<SortConfig>(nullptr, sb2, (u32*)nullptr, (u32*)nullptr, (u64*)nullptr, (u64*)nullptr, n, 0, 32, c->stream);
    if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs size query (q index)", hipGetErrorString(e));
    e = rocprim::inclusive_scan(nullptr, scan_bytes, (int*)nullptr, (int*)nullptr, n, rocprim::maximum<int>(), c->stream);
    if (e != hipSuccess) return fail(CL_ERR_HIP, "inclusive_scan size query", hipGetErrorString(e));
    e = rocprim::exclusive_scan(nullptr, scan2, (int*)nullptr, (int*)nullptr, 0, n + 1, rocprim::plus<int>(), c->stream);
    if (e != hipSuccess) return fail(CL_ERR_HIP, "exclusive_scan size query", hipGetErrorString(e));
    e = rocprim::inclusive_scan_by_key(nullptr, scan3, rocprim::make_reverse_iterator((int*)nullptr), rocprim::make_reverse_iterator((int*)nullptr),
                                       rocprim::make_reverse_iterator((int*)nullptr), n, rocprim::minimum<int>(), rocprim::equal_to<int>(), c->stream);
    if (e != hipSuccess) return fail(CL_ERR_HIP, "inclusive_scan_by_key size query", hipGetErrorString(e));
    size_t sb3 = 0;
    {
        // the layout sort with its keys and outputs as iterators (sort_layout): sized here so that the arena covers it
        GridParams g0{};
        const auto kin = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), SpKeyOfIndex{nullptr, nullptr, g0});
        const auto vin = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), SpValOfIndex{nullptr, nullptr});
        e = rocprim::radix_sort_pairs<SortConfig>(nullptr, sb3, kin, (u32*)nullptr, vin, SplitOut{nullptr, nullptr}, n, 0, 32, c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs size query (layout)", hipGetErrorString(e));
    }
    *sort_out = std::max<size_t>(std::max(std::max(sort_bytes, sb2), sb3), 16);
    *scan_out = std::max<size_t>(std::max(std::max(scan_bytes, scan2), scan3), 16);
    return CL_OK;
}

static size_t g_arena_overcommit = 0;
extern "C" void cl_debug_arena_overcommit(int64_t extra_bytes) { g_arena_overcommit = extra_bytes > 0 ? (size_t)extra_bytes : 0; }

static int reserve_workspace(cl_chrom* c)
{
    const size_t n = (size_t)c->n;
    WORKSPACE_WANTS(c, n);
    if (!c->arena.p && n > 0) {
        // + what a first run would otherwise allocate piece by piece on the host's critical path: rocPRIM's temporary
        // storage, the strip tables and the compaction scratch for eps >= 1000 (a smaller eps grows them), a first
        // candidate buffer of max(n / 8, min(2^20, n)) boxes (it grows on demand)
        // best effort: whatever fails here (size queries, the one big hipMalloc under memory pressure) must leave neither a sticky
        // HIP error nor an error text behind -- the buffers are then allocated one by one at the first run
        const std::string keep_err = g_err;
        size_t sort_bytes = 16, scan_bytes = 16;
        if (workspace_tmp_sizes(c, &sort_bytes, &scan_bytes) != CL_OK) { (void)hipGetLastError(); g_err = keep_err; return CL_OK; }
        const size_t s_guess = (size_t)(((long long)c->st.vmax - c->st.vmin) / 1000 + 64);
        const long long cand0 = std::max<long long>((long long)n / 8, std::min<long long>(1 << 20, (long long)n + 1024));   // (small handles stay small)
        std::vector<Want> all(std::begin(wants), std::end(wants));
        all.push_back({&c->sort_tmp, sort_bytes}); all.push_back({&c->scan_tmp, scan_bytes});
        all.push_back({&c->strip, (s_guess + 2) * 4}); all.push_back({&c->bstrip, (s_guess + 2) * 4}); all.push_back({&c->sel_tmp, (s_guess + 2) * 8 + (s_guess / 256 + 4) * 8 + 64});
        all.push_back({&c->blk_tmp, (n / 32 / 1024 + 4) * 8});     // (block sums / offsets of the in-kernel scans ride in sel_tmp / blk_tmp)
        all.push_back({&c->cand_box, (size_t)cand0 * 16}); all.push_back({&c->cand_step, (size_t)cand0 * 4});
        bool untouched = true;
        size_t total = 0;
        for (const Want& w : all) { untouched = untouched && w.b->p == nullptr; total += ((w.bytes + 255) / 256) * 256 + 256; }
        total += g_arena_overcommit;                     // (cl_debug_arena_overcommit: tests make this allocation fail)
        if (untouched && c->arena.ensure(total) != CL_OK) { (void)hipGetLastError(); g_err = keep_err; }
        else if (untouched) {
            char* at = (char*)c->arena.p;
            for (const Want& w : all) { const size_t sz = ((w.bytes + 255) / 256) * 256 + 256; w.b->adopt(at, sz); at += sz; }
            c->cand_cap = cand0;
        }
    }
    return CL_OK;
}

int ensure_workspace(cl_chrom* c, int S)
{
    const size_t n = (size_t)c->n;
    int rc;
    WORKSPACE_WANTS(c, n);
    if (!c->arena.p) (void)reserve_workspace(c);
    const cl_chrom::Slot* other = &c->slot[1 - c->cur];
    for (const Want& w : wants) if (w.b != &other->labels && w.b != &other->table && w.b != &other->slab && w.b != &c->slot[0].d_step && w.b != &c->slot[1].d_step && w.b != &c->bq && w.b != &c->bsp && w.b != &c->brow && w.b != &c->btile && w.b != &c->qb_key &&
                                    w.b != &c->qb_val && w.b != &c->fq && w.b != &c->fsp && w.b != &c->frow && w.b != &c->k7_cls && w.b != &c->rc_cnt && w.b != &c->rootlist && w.b != &c->cflag8 && (rc = w.b->ensure(w.bytes))) return rc;
    if ((rc = c->strip.ensure(((size_t)S + 2) * 4)) || (rc = c->counters.ensure(256))) return rc;      // (counters: allocated at upload)
    if (c->sv.fresh || c->sa.fresh) {
        // sentinel pads around the sorted arrays (k_region_core stages its windows without bounds checks)
        hipLaunchKernelGGL(k_init_pads, dim3(nblocks(2 * SORT_PAD)), dim3(TPB), 0, c->stream, c->sv.as<int>(), c->sa.as<int>(), (long long)n);
        c->sv.fresh = c->sa.fresh = false;
    }
    // rocPRIM temporary storage (part of the arena when the handle has one)
    size_t sort_bytes = 16, scan_bytes = 16;
    if (c->sort_tmp.bytes < 16 || c->scan_tmp.bytes < 16 || c->tmp_n != c->n) {
        if ((rc = workspace_tmp_sizes(c, &sort_bytes, &scan_bytes))) return rc;
        if ((rc = c->sort_tmp.ensure(sort_bytes)) || (rc = c->scan_tmp.ensure(scan_bytes))) return rc;
        c->tmp_n = c->n;
    }
    return CL_OK;
}


// Build GridParams for the rotated-strip layout (variants 1 and 2)
static int make_grid(cl_chrom* c, int variant, int eps, int minPts, int cut, GridParams* g)
{
    g->eps = eps; g->minPts = minPts; g->cut = cut; g->variant = variant;
    g->dbg = 0; g->dbg2 = 0;
    for (auto& w : g->tmask) w = 0;
    if (minPts >= 1 && minPts <= 128) g->tmask[(minPts - 1) >> 5] = 1u << ((minPts - 1) & 31);      // a one-off run serves its own minPts
    g->tgap = minPts - 1;
    g->qmin = INT_MIN;
#ifdef CLOOPS_DEVEL
    // developer build only (-DCLOOPS_DEVEL): ablation knobs that can change results; never in the shipped library
    { const char* e = getenv("CLOOPS_DBG"); g->dbg = e ? atoi(e) : 0; }
    { const char* e = getenv("CLOOPS_DBG2"); g->dbg2 = e ? atoi(e) : 0; }
#endif
    g->swap = (g->dbg & 16) ? 0 : 1;
    {
        const unsigned d = (unsigned)eps;
        int l = 0; while ((1ull << l) < d) ++l;                          // ceil(log2 d)
        g->magic = (u32)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
        g->sh1 = l < 1 ? l : 1; g->sh2 = l > 1 ? l - 1 : 0;
    }
    if (variant == CL_VARIANT_CDBSCAN2) {
        // absolute rotated cells (cDBSCAN2.py:67-70); exact only for 0 <= X <= Y
        if (c->st.amin < 0 || c->st.xmin < 0) return fail(CL_ERR_DOMAIN, "variant 2 (cDBSCAN2) needs 0 <= X <= Y for every PET");
        g->A0 = 0; g->V0 = 0;
    } else {
        g->A0 = g->swap ? c->st.vmin : c->st.amin; g->V0 = g->swap ? c->st.amin : c->st.vmin;
    }
    const int pmin = g->swap ? c->st.vmin : c->st.amin, pmax = g->swap ? c->st.vmax : c->st.amax;
    long long lo = ((long long)pmin - g->A0) / eps;
    long long hi = ((long long)pmax - g->A0) / eps;
    long long S = hi - lo + 1;
    if (S > (1LL << 28)) return fail(CL_ERR_GRID, "eps too small for the coordinate extent (strip table > 2^28 rows)");
    g->s0 = (int)lo; g->S = (int)S;
    const int qmin = g->swap ? c->st.amin : c->st.vmin, qmax = g->swap ? c->st.amax : c->st.vmax;
    (void)qmin;
    g->qbits = std::max(1, bits_for((unsigned)((long long)qmax - g->V0)));
    g->qtop = (int)std::min<long long>((long long)qmax - g->V0, INT_MAX);
    g->rbits = bits_for((unsigned)(eps - 1));
    g->peps = 1 << g->rbits;
    // the strip coordinate lives in the kernels as sp = strip << rbits | remainder (GridParams): sp + peps must stay an int
    if ((S + 2) << g->rbits > (long long)INT_MAX)
        return fail(CL_ERR_GRID, "coordinate extent too large for this eps (X+Y range + 2*eps must stay below 2^30)");
    return CL_OK;
}


void ev_record(cl_chrom* c, int k)
{
    if (c->profiling) (void)hipEventRecord(c->slot[c->cur].ev[k], c->stream);
}

// Longest strip over ALL rows of the chromosome for this layout and eps (an upper bound for every cut):
// measured on first use (one histogram pass + a blocking read-back, once per (layout, eps) and handle).
static int strip_maxlen(cl_chrom* c, const GridParams& g, int* out)
{
    const int layout = (g.variant == CL_VARIANT_CDBSCAN2 ? 2 : 0) | (g.swap ? 1 : 0);
    for (const auto& pl : c->plans)
        if (pl.layout == layout && pl.eps == g.eps) { *out = pl.maxlen; return CL_OK; }
    const int n = (int)c->n;
    int* hist = c->strip.as<int>();
    int* dmax = c->counters.as<int>() + CTR_MAXLEN;
    HIP_TRY(hipMemsetAsync(hist, 0, ((size_t)g.S + 2) * 4, c->stream));
    HIP_TRY(hipMemsetAsync(dmax, 0, 4, c->stream));
    LAUNCH(k_strip_hist, n, c->d_x, c->d_y, n, g, hist);
    hipLaunchKernelGGL(k_max_int, dim3(std::min(nblocks(g.S), 1024)), dim3(TPB), 0, c->stream, hist, g.S, dmax);
    int* h = c->h_pinned + 128;
    HIP_TRY(hipMemcpyAsync(h, dmax, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->plans.size() >= 64) c->plans.erase(c->plans.begin());
    c->plans.push_back({layout, g.eps, *h});
    *out = *h;
    return CL_OK;
}

// K0 + K1: keys, sort, strip table, decoded sorted arrays for the rows that pass `g.cut`, written to the given
// destination buffers (sorted arrays with their sentinel pads already in place)
static int sort_layout(cl_chrom* c, const GridParams& g, int* dsv, int* dsa, u32* drow /* or null: c->srow aliases a sort buffer */,
                       int* dstrip, int* dtile)
{
    const int n = (int)c->n;
    const int sh = g.qbits + g.rbits, strip_bits = std::max(1, bits_for((unsigned)g.S));
    // the q index: built at the handle's second sort (or its first, if the caller announced several: cl_set_sort_index)
    const int layout = (g.variant == CL_VARIANT_CDBSCAN2 ? 2 : 0) | (g.swap ? 1 : 0);
    const bool use_index = c->sort_index_mode > 0 || (c->sort_index_mode == 0 && (c->n_sorts >= 1 || c->qindex_layout == layout));
    ++c->n_sorts;
    // hybrid sort: worth it when it saves at least two radix passes (9-bit digits) and every strip is short
    bool hybrid = false;
    // (a chromosome with more than 64 PETs per strip on average is not measured at all: its longest strip is
    // practically never <= HS_LMAX, and the histogram pass on such data is slow -- many atomics per strip)
    if (!use_index && !(g.dbg & 256) && (g.qbits + strip_bits + 8) / 9 - (strip_bits + 8) / 9 >= 2 && (long long)n <= 64LL * g.S) {
        int maxlen = 0, rc;
        if ((rc = strip_maxlen(c, g, &maxlen))) return rc;
        hybrid = maxlen <= HS_LMAX;
    }
    if (use_index) {
        int rc;
        u32* k32_in = c->vals_in.as<u32>(); u32* k32_out = c->vals_out.as<u32>();       // the 4 n / 8 n byte sort buffers swap roles
        u64* v64_in = c->keys_in.as<u64>(); u64* v64_out = c->keys_out.as<u64>();
        if (c->qindex_layout != layout) {
            if ((rc = c->qb_key.ensure((size_t)n * 4)) || (rc = c->qb_val.ensure((size_t)n * 8))) return rc;
            LAUNCH(k_make_qkeys, n, c->d_x, c->d_y, n, g, k32_in, v64_in);
            size_t tb = c->sort_tmp.bytes;
            hipError_t e = rocprim::radix_sort_pairs<SortConfig>(c->sort_tmp.p, tb, k32_in, c->qb_key.as<u32>(), v64_in, c->qb_val.as<u64>(),
                                                                 (size_t)n, 0, g.qbits, c->stream);
            if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs(q index)", hipGetErrorString(e));
            c->qindex_layout = layout;
        }
        ev_record(c, 0);
        if (g.cut <= 0) {
            // every row is in the layout: keys and values are made inside the first radix pass, the last one writes the layout
            ev_record(c, 1);
            u32* rows = drow ? drow : k32_in;
            auto strip_sort = [&](const GridParams& gx, int* osa, int* osv, u32* orow, int* ostrip, int* otile) -> int {
                const int sbits = std::max(1, bits_for((unsigned)gx.S));
                const auto kin = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), SpKeyOfIndex{c->qb_key.as<u32>(), c->qb_val.as<u64>(), gx});
                const auto vin = rocprim::make_transform_iterator(rocprim::counting_iterator<int>(0), SpValOfIndex{c->qb_key.as<u32>(), c->qb_val.as<u64>()});
                size_t need = 0;
                hipError_t e = rocprim::radix_sort_pairs<SortConfig>(nullptr, need, kin, (u32*)osa, vin, SplitOut{osv, orow}, (size_t)n, gx.rbits, gx.rbits + sbits, c->stream);
                if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs(strips) size query", hipGetErrorString(e));
                int rc2;
                if (need > c->sort_tmp.bytes && (rc2 = c->sort_tmp.ensure(need))) return rc2;
                size_t tb = c->sort_tmp.bytes;
                e = rocprim::radix_sort_pairs<SortConfig>(c->sort_tmp.p, tb, kin, (u32*)osa, vin, SplitOut{osv, orow}, (size_t)n, gx.rbits, gx.rbits + sbits, c->stream);
                if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs(strips)", hipGetErrorString(e));
                LAUNCH(k_strip_table32, gx.S + 2, (const u32*)osa, n, gx.S, gx.rbits, ostrip, otile);
                return CL_OK;
            };
            const int fw = c->fine_w;
            if (fw > 0 && g.eps % fw == 0 && g.eps / fw <= CL_FINE_KMAX) {
                // the eps values of the sweep share the divisor fw (cl_set_eps_list): the layout comes from the fine one
                if (c->fine_valid_w != fw || c->fine_layout != layout) {
                    GridParams gf;
                    if (make_grid(c, g.variant, fw, 1, 0, &gf) != CL_OK) c->fine_w = 0;      // (too many strips of that width: sort as before)
                    else {
                        gf.dbg = g.dbg; gf.dbg2 = g.dbg2;
                        if ((rc = c->fq.ensure((size_t)n * 4)) || (rc = c->fsp.ensure((size_t)n * 4)) || (rc = c->frow.ensure((size_t)n * 4)) ||
                            (rc = c->fstrip.ensure(((size_t)gf.S + 2) * 4))) return rc;
                        if ((rc = strip_sort(gf, c->fsp.as<int>(), c->fq.as<int>(), c->frow.as<u32>(), c->fstrip.as<int>(), (int*)nullptr))) return rc;
                        c->fine_g = gf; c->fine_valid_w = fw; c->fine_layout = layout;
                    }
                }
                if (c->fine_w > 0) {
                    const GridParams& gf = c->fine_g;
                    const int k = g.eps / fw;
                    LAUNCH(k_strips_from_fine, g.S + 2, g.S, g.s0, k, gf.S, gf.s0, (const int*)c->fstrip.as<int>(), n, dstrip);
                    // (q + 1 -- the tie-break key -- must fit the q field of the sorted keys: qb bits; 0 = the field would leave no room for runs)
                    int qb = bits_for((unsigned)std::min<long long>((long long)gf.qtop + 1, 0x7fffffffLL));
                    if (qb > 28) qb = 0;
                    hipLaunchKernelGGL(k_layout_from_fine, dim3(nblocks(n, LFF_T)), dim3(256), 0, c->stream, n, g, gf, k, (unsigned)(((1ull << 32) + (unsigned)k - 1ull) / (unsigned)k), qb, (const int*)c->fq.as<int>(),
                                       (const int*)c->fsp.as<int>(), (const u32*)c->frow.as<u32>(), (const int*)c->fstrip.as<int>(), dsv, dsa, rows, dtile);
                    HIP_TRY(hipGetLastError());
                    c->srow = rows;
                    return CL_OK;
                }
            }
            if ((rc = strip_sort(g, dsa, dsv, rows, dstrip, dtile))) return rc;
            c->srow = rows;
            return CL_OK;
        }
        LAUNCH(k_make_spkeys, n, n, g, (const u32*)c->qb_key.as<u32>(), (const u64*)c->qb_val.as<u64>(), k32_in, v64_in);
        ev_record(c, 1);
        size_t tb = c->sort_tmp.bytes;
        hipError_t e = rocprim::radix_sort_pairs<SortConfig>(c->sort_tmp.p, tb, k32_in, k32_out, v64_in, v64_out, (size_t)n, g.rbits, g.rbits + strip_bits, c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs(strips)", hipGetErrorString(e));
        LAUNCH(k_strip_table32, g.S + 2, (const u32*)k32_out, n, g.S, g.rbits, dstrip, (int*)nullptr);
        u32* rows = drow ? drow : k32_in;                 // the unsorted keys are dead after the sort
        LAUNCH(k_decode_sp, n, n, g, (const u32*)k32_out, (const u64*)v64_out, dsv, dsa, dtile, rows);
        c->srow = rows;
        return CL_OK;
    }
    ev_record(c, 0);
    LAUNCH(k_make_keys, n, c->d_x, c->d_y, n, g, c->keys_in.as<u64>(), c->vals_in.as<u32>());
    ev_record(c, 1);
    size_t tmp_bytes = c->sort_tmp.bytes;
    const int end_bit = sh + strip_bits;
    hipError_t e = rocprim::radix_sort_pairs<SortConfig>(c->sort_tmp.p, tmp_bytes, c->keys_in.as<u64>(), c->keys_out.as<u64>(),
                                             c->vals_in.as<u32>(), c->vals_out.as<u32>(), (size_t)n, hybrid ? sh : g.rbits, end_bit, c->stream);
    if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs", hipGetErrorString(e));
    LAUNCH(k_strip_table, g.S + 2, c->keys_out.as<u64>(), n, g.S, sh, dstrip);
    if (hybrid) {
        u32* rows = drow ? drow : c->vals_in.as<u32>();  // the unsorted row ids are dead after the sort
        hipLaunchKernelGGL(k_strip_sort, dim3(nblocks(n, HS_TPB)), dim3(HS_TPB), 0, c->stream, n, g, c->keys_out.as<u64>(),
                           c->vals_out.as<u32>(), dstrip, dsv, dsa, rows, dtile, c->counters.as<int>());
        c->srow = rows;
    } else {
        LAUNCH(k_decode_sorted, n, n, g, c->keys_out.as<u64>(), dsv, dsa, dtile, c->vals_out.as<u32>(), drow);
        c->srow = drow ? drow : c->vals_out.as<u32>();
    }
    return CL_OK;
}

// K0 + K1 + K2: sorted working set of the run (keys / sort, or a compaction of the base layout), then the neighbour
// counts.  Leaves c->w_sv, w_sa, srow, w_strip, w_tile pointing at the sorted arrays of this run.
static int run_sort_and_count(cl_chrom* c, const GridParams& g, bool exact)
{
    const int n = (int)c->n;
    int rc;
    // M = PETs that pass the cut, from the distance histogram of the upload (exact for cut <= 65536)
    c->run_m = n; c->run_m_exact = g.cut <= 0;
    if (g.cut > 0 && g.cut < DCUM_BINS && !c->dcum.empty()) { c->run_m = (int)(n - c->dcum[g.cut]); c->run_m_exact = true; }
    int* wsv = c->sv.as<int>() + SORT_PAD;
    int* wsa = c->sa.as<int>() + SORT_PAD;
    c->w_cnt = c->cnt.as<int>();
    c->last_k2_mode = 0; c->last_k2_mode_make = false;
    GridParams gk = g;                                    // what K2 sees: the minPts values its words serve, filled in below
    bool k2_band = false, k2_skip = false;
    c->ws = WordSrc{c->cnt.as<int>(), nullptr, nullptr, nullptr, 0, g.rbits};
    c->run_level = exact ? 0 : std::min(c->traversal, 3);
    c->w_dM = nullptr; c->l4_cut = false; c->l4_band = false; c->l4_make_base = false;
    c->slot[c->cur].band_timed = false; c->slot[c->cur].n_queried = 0;
    if (!c->reuse_layout) {
        // every run sorts for itself (the cut filter rides in the keys: filtered rows go behind the last strip)
        if ((rc = sort_layout(c, g, wsv, wsa, nullptr, c->strip.as<int>(), c->tile_s0.as<int>()))) return rc;
        c->w_sv = wsv; c->w_sa = wsa; c->w_strip = c->strip.as<int>(); c->w_tile = c->tile_s0.as<int>();
    } else {
        const int layout = (g.variant == CL_VARIANT_CDBSCAN2 ? 2 : 0) | (g.swap ? 1 : 0);
        if (!c->base.valid || c->base.layout != layout || c->base.eps != g.eps) {
            c->base.valid = false;
            if ((rc = c->bq.ensure(((size_t)n + 2 * SORT_PAD) * 4)) || (rc = c->bsp.ensure(((size_t)n + 2 * SORT_PAD) * 4)) ||
                (rc = c->brow.ensure((size_t)n * 4)) || (rc = c->bstrip.ensure(((size_t)g.S + 2) * 4)) ||
                (rc = c->btile.ensure(((size_t)n / 256 + 2) * 4))) return rc;
            if (c->bq.fresh || c->bsp.fresh) {
                hipLaunchKernelGGL(k_init_pads, dim3(nblocks(2 * SORT_PAD)), dim3(TPB), 0, c->stream, c->bq.as<int>(), c->bsp.as<int>(), (long long)n);
                c->bq.fresh = c->bsp.fresh = false;
            }
            GridParams g0 = g;
            g0.cut = 0;
            c->rc.valid = false;                          // the cached words belong to the layout that is being replaced
            c->bkey_valid = false;                        // ... and so do variant 2's cell minima
            if ((rc = sort_layout(c, g0, c->bq.as<int>() + SORT_PAD, c->bsp.as<int>() + SORT_PAD, c->brow.as<u32>(),
                                  c->bstrip.as<int>(), c->btile.as<int>()))) return rc;
            c->base.valid = true; c->base.layout = layout; c->base.eps = g.eps;
        } else {
            ev_record(c, 0);
            ev_record(c, 1);
        }
        const long long dmin = g.swap ? c->st.amin : 0;   // swap == 0 is a developer layout: always compacts
        const bool on_base = g.cut <= 0 || (g.swap && (long long)g.cut <= dmin);
        // ---- count cache (cl_chrom::rc): what this run does about its K2 words -------------------------------------
        //   make : the run's K2 writes the cache (first clustering run on this layout, or one the cached words cannot serve)
        //   same : same cut as the run that made the words -- nothing to do, the consumers read the cache
        //   remap: another cut -- K2 runs on the cut band only (k_region_band), the consumers read every other PET's word at
        //          its place in the layout the words were made on (WordSrc)
        const int m1 = g.minPts - 1;
        const bool cacheable = !exact && c->reuse_counts && g.swap && m1 >= 1 && m1 <= 127;
        const int thr_new = on_base ? 0 : g.cut - g.V0;   // q >= 0 everywhere: threshold 0 removes nothing
        enum { RC_NONE, RC_MAKE, RC_SAME, RC_REMAP } rcmode = RC_NONE;
        // traversal level 4 (lists from the base layout, count cache in base-position space) needs the cache: a run it does not take
        // (exact counts, minPts outside 2 .. 128, cache switched off) works on a copy of the layout as before
        const bool want4 = c->traversal >= 4 && cacheable;
        if (cacheable) {
            const bool serves = c->rc.valid && c->rc.layout == layout && c->rc.eps == g.eps && g.minPts <= c->rc.cap &&
                                ((c->rc.tmask[m1 >> 5] >> (m1 & 31)) & 1u) && c->rc.base_space == want4;
            if (serves && thr_new == c->rc.thr && !(c->rc.cut_on_base && thr_new != 0)) rcmode = RC_SAME;
            else if (serves && !on_base && (long long)g.S <= 8LL * n) rcmode = RC_REMAP;      // (the band kernel works strip by strip)
            else rcmode = RC_MAKE;
            if ((rc = c->rc_cnt.ensure((size_t)n * 4)) || (rc = c->rc_pre.ensure(((size_t)g.S + 2) * 4)) ||
                (rc = c->rc_poff.ensure(((size_t)g.S + 2) * 4)) || (rc = c->rc_dpre.ensure(((size_t)g.S + 2) * 4)) ||
                (rc = c->rc_D.ensure(((size_t)g.S + 2) * 4)) || (rc = c->rc_blen.ensure(((size_t)g.S + 2) * 8))) return rc;
        }
        c->run_level = want4 ? 4 : std::min(c->traversal, 3);
        if (exact) c->run_level = 0;
        if (c->run_level == 4 && g.variant == CL_VARIANT_CDBSCAN2 && !c->bkey_valid) {
            if ((rc = lists_base_keys(c, g))) return rc;
            c->bkey_valid = true;
        }
        if (rcmode == RC_MAKE) {
            c->rc.valid = true; c->rc.layout = layout; c->rc.eps = g.eps; c->rc.thr = thr_new; c->rc.cap = g.minPts; c->rc.base_space = want4;
            c->rc.cut_on_base = false;
            // the minPts values these words will be asked about: the announced ones up to this run's (cl_set_count_thresholds), or
            // everything from the announced floor up (cl_set_count_floor), and this run's own
            for (int k = 0; k < 4; ++k) c->rc.tmask[k] = 0;
            for (int t = 2; t < g.minPts; ++t) {
                const bool in = ((c->count_tmask[(t - 1) >> 5] >> ((t - 1) & 31)) & 1u) || (c->count_floor > 0 && t >= c->count_floor);
                if (in) c->rc.tmask[(t - 1) >> 5] |= 1u << ((t - 1) & 31);
            }
            c->rc.tmask[m1 >> 5] |= 1u << (m1 & 31);
            for (int k = 0; k < 4; ++k) gk.tmask[k] = c->rc.tmask[k];
            gk.tgap = 0;
            for (int t = 2, prev = 1; t <= g.minPts; ++t)
                if ((c->rc.tmask[(t - 1) >> 5] >> ((t - 1) & 31)) & 1u) { gk.tgap = std::max(gk.tgap, t - prev); prev = t; }
            c->w_cnt = c->rc_cnt.as<int>();
            // (level 4 under a cut: K2 queries the compact copy into the work buffer, lists_words_to_base moves the words to their
            //  base positions in the cache)
            if (want4 && !on_base) c->w_cnt = c->cnt.as<int>();
            c->ws.rc = c->w_cnt;
            if (on_base) {
                HIP_TRY(hipMemsetAsync(c->rc_pre.p, 0, ((size_t)g.S + 2) * 4, c->stream));
                HIP_TRY(hipMemsetAsync(c->rc_poff.p, 0, ((size_t)g.S + 2) * 4, c->stream));
            }
            if (want4 && !on_base && (long long)g.S <= 8LL * n) {
                // Level 4: the words of an eps are ALWAYS made on the base layout itself (threshold 0: every row), also when the first
                // run of the eps has a cut -- that run then re-uses them like every later one (fresh words for its cut band, the
                // cached ones beyond it).  One region query over all rows (213 us on chr1) instead of a copy of the layout, the
                // query on the copy and the move of its words to base positions (54 + 146 + 77 us) -- and no copy of the layout is
                // ever made (cLoops/pipe.py:59-63: the cut only removes a prefix of every strip).
                HIP_TRY(hipMemsetAsync(c->rc_pre.p, 0, ((size_t)g.S + 2) * 4, c->stream));
                HIP_TRY(hipMemsetAsync(c->rc_poff.p, 0, ((size_t)g.S + 2) * 4, c->stream));
                c->l4_make_base = true;                      // (launched in the region-query bracket, below)
                // (the PETs this run's cut removes get no word: a later run under a smaller cut finds them in its cut band, which
                //  reaches up to the larger of the two thresholds + eps)
                c->rc.thr = thr_new;
                c->rc.cut_on_base = true;
                c->w_cnt = c->rc_cnt.as<int>();
                rcmode = RC_REMAP;
                c->last_k2_mode_make = true;
            }
        }
        if (rcmode == RC_SAME) {
            c->w_cnt = c->rc_cnt.as<int>();
            c->ws.rc = c->w_cnt;
            k2_skip = true;
            c->last_k2_mode = 1;
        } else if (rcmode == RC_REMAP) {
            // the two cuts differ in the PETs with q in [min, max) of the thresholds: a PET keeps its count iff q - eps >= max
            c->ws = WordSrc{c->rc_cnt.as<int>(), c->cnt.as<int>(), c->rc_D.as<int>(), c->rc_dpre.as<int>(), std::max(thr_new, c->rc.thr) + g.eps, g.rbits};
            k2_band = true;
            c->last_k2_mode = c->last_k2_mode_make ? 0 : 2;
        }
        if (on_base) {
            // no row is filtered: the run works on the base layout itself
            c->w_sv = c->bq.as<int>() + SORT_PAD; c->w_sa = c->bsp.as<int>() + SORT_PAD; c->srow = c->brow.as<u32>();
            c->w_strip = c->bstrip.as<int>(); c->w_tile = c->btile.as<int>();
        } else {
            if (!g.swap) return fail(CL_ERR_ARG, "internal: layout reuse needs the v-band layout");
            // stable compaction of the base layout by d = q + V0 >= cut (pipe.py:59-62): same order as sorting the
            // filtered rows, one pass over 12 B/PET
            const int nb = nblocks(n, CMP_BLOCK);
            const int nblk = nblocks(g.S + 1, 256);       // workgroups of k_cut_strips = blocks of the two-level strip table
            if ((rc = c->sel_tmp.ensure(((size_t)g.S + 2) * 8 + ((size_t)nblk + 2) * 8 + 64))) return rc;
            int* sloc = c->sel_tmp.as<int>();
            int* src0 = sloc + g.S + 2;
            int* bsum = src0 + g.S + 2;
            int* sboffs = bsum + nblk + 2;
            int* d_M = c->counters.as<int>() + CTR_M;
            const int thr = g.cut - g.V0;
            const int* bq = c->bq.as<int>() + SORT_PAD;
            const bool l4 = c->run_level == 4;
            c->l4_cut = true; c->l4_band = rcmode == RC_REMAP;
            if (l4 && ((rc = c->l_tab.ensure(((size_t)g.S + 2) * 16)) || (rc = c->l_fix.ensure(((size_t)g.S + 2) * 8)))) return rc;
            LAUNCH(k_cut_strips, g.S + 1, g.S, thr, (const int*)c->bstrip.as<int>(), bq, sloc, bsum, sboffs, c->counters.as<int>() + CTR_TICKET_A, src0,
                   rcmode == RC_MAKE ? c->rc_pre.as<int>() : (int*)nullptr,
                   rcmode == RC_REMAP ? (const int*)c->rc_pre.as<int>() : (const int*)nullptr,
                   rcmode == RC_REMAP ? c->rc_dpre.as<int>() : (int*)nullptr, rcmode == RC_REMAP ? c->rc_blen.as<int2>() : (int2*)nullptr,
                   c->ws.bandq, g.eps, c->init_nclr > 0 ? c->flag.as<int>() : (int*)nullptr, c->init_nclr, c->counters.as<int>(),
                   l4 ? c->l_tab.as<int4>() : (int4*)nullptr, (const u32*)c->brow.as<u32>(), g,
                   (l4 && g.variant == CL_VARIANT_CDBSCAN2) ? c->l_fix.as<int2>() : (int2*)nullptr);
            c->init_nclr = 0;
            // (the kernel that opens a clustering run also clears its key bitmap and counters: init_nclr > 0)
            if (l4 && rcmode == RC_REMAP) {
                // no copy of the layout: the band query alone (it reads the base layout and writes its words at the PETs' places in the
                // run's -- virtual -- layout); the list kernels apply the cut by index
                const int nbb = nblocks(nblocks(g.S, KB_SB), CMP_TPB / 64);
                if (c->profiling) { (void)hipEventRecord(c->slot[c->cur].ev[8], c->stream); c->slot[c->cur].band_timed = true; }
                hipLaunchKernelGGL(k_band, dim3(nbb), dim3(CMP_TPB), 0, c->stream, g.S, g.rbits, g.eps, g.minPts, bq, (const int*)(c->bsp.as<int>() + SORT_PAD),
                                   (const int*)src0, (const int*)sloc, (const int*)sboffs, (const int2*)c->rc_blen.as<int2>(), c->cnt.as<int>(), d_M,
                                   c->run_m_exact ? c->run_m : -1, c->counters.as<int>(), g.dbg);
                if (c->profiling) (void)hipEventRecord(c->slot[c->cur].ev[9], c->stream);
                c->w_dM = d_M;
            } else if (l4 && rcmode == RC_SAME) {
                hipLaunchKernelGGL(k_store_m, dim3(1), dim3(1), 0, c->stream, g.S, (const int*)sloc, (const int*)sboffs, d_M, c->run_m_exact ? c->run_m : -1, c->counters.as<int>());
                c->w_dM = d_M;
            } else
            if (rcmode == RC_REMAP) {
                const int nbb = nblocks(nblocks(g.S, KB_SB), CMP_TPB / 64);      // workgroups that do K2 on the cut band (four waves of KB_SB strips each)
                hipLaunchKernelGGL(k_cut_copy<true>, dim3(nbb + nb), dim3(CMP_TPB), 0, c->stream, n, g.S, g.rbits, thr, bq, (const int*)(c->bsp.as<int>() + SORT_PAD),
                                   (const u32*)c->brow.as<u32>(), (const int*)src0, (const int*)sloc, (const int*)sboffs, c->strip.as<int>(), wsv, wsa, c->vals_out.as<u32>(), c->tile_s0.as<int>(),
                                   d_M, c->run_m_exact ? c->run_m : -1, c->counters.as<int>(), (int*)nullptr, (const int*)c->rc_poff.as<int>(), c->rc_D.as<int>(),
                                   nbb, (const int2*)c->rc_blen.as<int2>(), c->cnt.as<int>(), g.eps, g.minPts, g.dbg);
            } else
            hipLaunchKernelGGL(k_cut_copy<false>, dim3(nb), dim3(CMP_TPB), 0, c->stream, n, g.S, g.rbits, thr, bq, (const int*)(c->bsp.as<int>() + SORT_PAD),
                               (const u32*)c->brow.as<u32>(), (const int*)src0, (const int*)sloc, (const int*)sboffs, c->strip.as<int>(), wsv, wsa, c->vals_out.as<u32>(), c->tile_s0.as<int>(),
                               d_M, c->run_m_exact ? c->run_m : -1, c->counters.as<int>(), rcmode == RC_MAKE ? c->rc_poff.as<int>() : (int*)nullptr,
                               (const int*)nullptr, (int*)nullptr, 0, (const int2*)nullptr, (int*)nullptr, 0, 0, 0);
            c->w_sv = wsv; c->w_sa = wsa; c->srow = c->vals_out.as<u32>();
            c->w_strip = c->strip.as<int>(); c->w_tile = c->tile_s0.as<int>();
            if (!c->w_dM) c->w_dM = d_M;
        }
    }
    ev_record(c, 2);
    rc = CL_OK;
    if (c->l4_make_base) {
        // level 4: the words of the eps, on the base layout itself (every row), whatever this run's cut is
        GridParams g0 = gk;
        g0.cut = 0;
        g0.qmin = c->rc.thr;
        rc = cl_launch_region(c->stream, g0, n, n, false, c->bq.as<int>() + SORT_PAD, c->bsp.as<int>() + SORT_PAD, c->bstrip.as<int>(), c->btile.as<int>(),
                              c->rc_cnt.as<int>());
        c->slot[c->cur].n_queried = c->run_m;             // (the PETs that got a word: those the cut keeps)
    } else if (!k2_skip && !k2_band && !SKIP(256)) {
        rc = cl_launch_region(c->stream, gk, n, c->run_m, exact, c->w_sv, c->w_sa, c->w_strip, c->w_tile, c->w_cnt);
        c->slot[c->cur].n_queried = c->run_m;
    }
    if (rc) return rc;
    if (c->run_level == 4 && c->l4_cut && !k2_skip && !k2_band && (rc = lists_words_to_base(c, g))) return rc;
    ev_record(c, 3);
    HIP_TRY(hipGetLastError());
    return CL_OK;
}

__global__ void k_nop() {}

int ensure_events(cl_chrom* c)
{
    if (c->profiling && !c->ev_ready) {
        for (auto& sl : c->slot) for (auto& e : sl.ev) HIP_TRY(hipEventCreate(&e));
        c->ev_ready = true;
        // calibration of the event bracket itself: an EMPTY kernel between two event records (median of 9).
        // A bracket around one kernel reads kernel time + this (event packets, dispatch gap).
        float v[9];
        hipEvent_t a = c->slot[0].ev[0], b = c->slot[0].ev[1];
        for (int r = 0; r < 9; ++r) {
            HIP_TRY(hipEventRecord(a, c->stream));
            hipLaunchKernelGGL(k_nop, dim3(1), dim3(64), 0, c->stream);
            HIP_TRY(hipEventRecord(b, c->stream));
            HIP_TRY(hipEventSynchronize(b));
            (void)hipEventElapsedTime(&v[r], a, b);
        }
        std::sort(v, v + 9);
        c->ev_bracket_ms = v[4];
    }
    return CL_OK;
}

static int check_args(cl_chrom* c, int eps, int minPts, int cut)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    if (eps <= 0) return fail(CL_ERR_ARG, "eps must be > 0 (the reference divides by it: ZeroDivisionError)");
    if (eps >= (1 << 30)) return fail(CL_ERR_ARG, "eps must be < 2^30");
    if (minPts < 0) return fail(CL_ERR_ARG, "minPts must be >= 0");
    (void)cut;
    return CL_OK;
}

extern "C" int cl_neighbor_counts(cl_chrom* c, int32_t eps, int32_t cut, int32_t* counts_out)
{
    int rc = check_args(c, eps, 1, cut);
    if (rc) return rc;
    if (c->n == 0) return CL_OK;
    if (!counts_out) return fail(CL_ERR_ARG, "counts_out is null");
    HIP_TRY(hipSetDevice(c->device));
    GridParams g;
    if ((rc = make_grid(c, CL_VARIANT_CDBSCAN1, eps, 1, cut, &g))) return rc;
    if ((rc = ensure_workspace(c, g.S))) return rc;
    if ((rc = ensure_events(c))) return rc;
    if ((rc = run_sort_and_count(c, g, true))) return rc;
    const int n = (int)c->n;
    HIP_TRY(hipMemsetAsync(c->slot[c->cur].labels.p, 0xFF, (size_t)n * 4, c->stream));
    LAUNCH(k_scatter_counts, n, c->w_strip, g.S, c->srow, c->cnt.as<int>(), c->slot[c->cur].labels.as<int>());
    HIP_TRY(hipMemcpyAsync(counts_out, c->slot[c->cur].labels.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (c->profiling) {
        memset(&c->timing, 0, sizeof(c->timing));
        hipEvent_t* ev = c->slot[c->cur].ev;
        (void)hipEventElapsedTime(&c->timing.ms_keys, ev[0], ev[1]);
        (void)hipEventElapsedTime(&c->timing.ms_sort, ev[1], ev[2]);
        (void)hipEventElapsedTime(&c->timing.ms_region, ev[2], ev[3]);
        c->timing.n_strips = g.S + 2;
    }
    c->have_result = false;
    return CL_OK;
}

// Shared tail of every variant.  finish_enqueue(): pack {K, overflow, M} into the slot's device
// header, then (copy stream, behind an event) header + labels to the host.  Nothing blocks the
// host; the compute stream is free for the next run.  finish_wait(): complete the oldest run.
__global__ void k_pack_header(int* __restrict__ hdr, const int* __restrict__ rankscan_total, const int* __restrict__ counters,
                              const int* __restrict__ d_M)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        hdr[0] = rankscan_total[0];          // K  = ids handed out
        hdr[1] = counters[CTR_OVERFLOW];
        hdr[2] = d_M[0];                     // M  = PETs that entered DBSCAN
        hdr[3] = 0;                          // n_clusters (filled by k_export_table)
        hdr[4] = -1;                         // max_label
        hdr[5] = 0;                          // 1 = table truncated (host buffer too small)
    }
}

// The cluster table goes to the host from INSIDE the compute stream: the kernel knows K (the host
// does not, without a round trip) and stores the K rows straight into pinned host memory, counting
// the non-empty ids on the way.  cl_wait() then needs no GPU work at all -- a copy issued there
// would queue behind the kernels of the next run that is already executing.
__global__ void k_export_table(int* __restrict__ hdr, Table t, cl_box* __restrict__ host_rows, int cap)
{
    const int K = hdr[0];
    const bool store = cap >= 0;                        // cap < 0: count the non-empty ids only (the caller does not read the rows)
    const int lim = store ? min(K, cap) : K;
    int nc = 0, ml = -1;
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < lim; k += gridDim.x * blockDim.x) {
        if (store) {
            cl_box b = t.get(k);
            if (b.count > 0) { ++nc; ml = k; } else { b.min_x = b.max_x = b.min_y = b.max_y = 0; }
            host_rows[k] = b;
        } else if (t.count[k] > 0) { ++nc; ml = k; }
    }
    for (int o = 32; o > 0; o >>= 1) { nc += __shfl_down(nc, o); ml = max(ml, __shfl_down(ml, o)); }
    if ((threadIdx.x & 63) == 0) {
        if (nc) atomicAdd(&hdr[3], nc);
        if (ml >= 0) atomicMax(&hdr[4], ml);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && store && K > cap) hdr[5] = 1;
}

Table make_table_slot(cl_chrom* c, int slot)
{
    Table t;
    int* base = c->slot[slot].table.as<int>();
    const size_t stride = (size_t)c->n + 1;
    t.count = base; t.minx = base + stride; t.maxx = base + 2 * stride; t.miny = base + 3 * stride; t.maxy = base + 4 * stride;
    return t;
}
Table make_table(cl_chrom* c) { return make_table_slot(c, c->cur); }

// the upload's distance histogram, when it covers every PET a cut of `cut` drops (K7Src::dh)
const int* k7_hist_for(cl_chrom* c, int cut)
{
    return (cut > 0 && cut < DCUM_BINS && c->n_neg == 0 && !c->dcum.empty() && c->dhist.p) ? c->dhist.as<int>() : (const int*)nullptr;
}

int finish_enqueue(cl_chrom* c, int n_strips, const int* d_M, int32_t* labels_out)
{
    const int n = (int)c->n;
    cl_chrom::Slot& sl = c->slot[c->cur];
    int* dh = c->hdr.as<int>() + 16 * c->cur;
    // the pinned host rows of the cluster table: only a run that exports its table needs them (a sweep step does not, and
    // page-locking (n / 16 + 65 536) rows per result slot was most of the first sweep's overhead: 1 .. 10 ms per handle and slot)
    if (sl.h_boxes_cap == 0 && c->export_table && c->pending_step < 0) {
        const size_t cap = (size_t)n / 16 + 65536;
        HIP_TRY(hipHostMalloc((void**)&sl.h_boxes, cap * sizeof(cl_box), hipHostMallocDefault));
        sl.h_boxes_cap = cap;
    }
    if (!c->hdr_packed)
        hipLaunchKernelGGL(k_pack_header, dim3(1), dim3(64), 0, c->stream, dh, c->k_total ? c->k_total : c->rankscan.as<int>() + n, c->counters.as<int>(), d_M);
    c->k_total = nullptr; c->hdr_packed = false;
    // (without export the kernel still counts the non-empty ids for the header; cap 0 = no row is stored)
    // (a sweep step reads neither the rows nor n_clusters / max_label: its header keeps the values of k_pack_header)
    const bool exported = c->export_table && c->pending_step < 0;
    if (c->pending_step < 0)
        hipLaunchKernelGGL(k_export_table, dim3(256), dim3(TPB), 0, c->stream, dh, make_table(c), sl.h_boxes,
                           exported ? (int)std::min<size_t>(sl.h_boxes_cap, 0x7fffffff) : -1);
    sl.exported = exported;
    sl.step_valid = false;
    sl.host_written = false;
    if (c->pending_step >= 0) {
        // sweep-step tail, still inside the run's stream: classify the table (pipe.py:83-97), append the inter-ligation
        // boxes to the chromosome's candidate buffer, reduce the distance statistics -- the host gets everything with
        // the run's own completion (one wait per chromosome and step)
        int rc;
        if ((rc = c->k7_cls.ensure((size_t)n + 16))) return rc;
        const int kmax = std::max(1, std::min(sl.kmax, n));     // the number of ids K is only known on the device: K <= kmax
        if ((rc = ensure_cand_capacity(c, c->cand_n + kmax))) return rc;
        // step output (device and pinned host): 16 B box totals | the reduced statistics (one K7Part) | log histogram | fine window;
        // the workgroup partials live behind it on the device only
        const size_t out_bytes = 16 + sizeof(K7Part) + K7_LOGBINS * 8 + K7_FINE * 8;
        if ((rc = sl.d_step.ensure(out_bytes + K7_BLOCKS * sizeof(K7Part)))) return rc;
        const int nb = nblocks(kmax, CAND_BLOCK);
        if ((rc = c->sel_tmp.ensure((size_t)nb * 12 + 64))) return rc;
        int* bcount = c->sel_tmp.as<int>();
        Table t = make_table(c);
        signed char* cls = c->k7_cls.as<signed char>();
        char* ds = (char*)sl.d_step.p;
        unsigned long long* lh = (unsigned long long*)(ds + 16 + sizeof(K7Part));
        hipLaunchKernelGGL(k_step_classify_count, dim3(nb), dim3(256), 0, c->stream, (const int*)dh, t, cls, bcount, nb,
                           lh, K7_LOGBINS + K7_FINE);                                       // + log histogram and the fine window behind it cleared
        // (no scan over the nb block counts: every append block sums the few counts in front of it itself)
        hipLaunchKernelGGL(k_cand_append, dim3(nb), dim3(256), 0, c->stream, (const int*)dh, cls, t, (const int*)nullptr, (const int*)bcount,
                           (int)c->cand_n, c->pending_step, (int)std::min<long long>(c->cand_cap, INT_MAX), c->cand_box.as<int4>(), c->cand_step.as<int>());
        K7Part* parts = (K7Part*)(ds + out_bytes);
        K7Src src{};
        src.sorted = sl.sorted_src ? (sl.k7_lcnt ? 2 : 1) : 0; src.n = n; src.M = 0; src.v0 = sl.k7_v0; src.dM = sl.k7_lcnt ? sl.k7_lcnt : d_M;
        src.X = c->d_x; src.Y = c->d_y; src.labels = sl.labels.as<int>(); src.sv = sl.k7_sv; src.slab = sl.slab.as<int>();
        src.dh = k7_hist_for(c, c->pending_cut);
        int k7b = K7_STEP_BLOCKS;
#ifdef CLOOPS_DEVEL
        int k7t = K7_STEP_THREADS;
        { const char* e = getenv("CLOOPS_K7B"); if (e) k7b = std::max(1, std::min(atoi(e), K7_BLOCKS)); e = getenv("CLOOPS_K7T"); if (e) k7t = atoi(e); }
#else
        const int k7t = K7_STEP_THREADS;
#endif
        if (!SKIP(16)) hipLaunchKernelGGL(k7_summary, dim3(k7b), dim3(k7t), 0, c->stream, src, c->pending_cut, cls, parts, lh,
                           (unsigned)c->pending_fine_lo, c->pending_fine_lo >= 0 ? lh + K7_LOGBINS : (unsigned long long*)nullptr);
        static_assert((16 + sizeof(K7Part)) % 8 == 0, "step output in 8-byte words");
        hipLaunchKernelGGL(k7_reduce_parts, dim3(1), dim3(256), 0, c->stream, (const K7Part*)parts, k7b, (K7Part*)(ds + 16),
                           (const int*)bcount, nb, (long long*)ds, (const unsigned long long*)ds, (int)(out_bytes / 8),
                           (unsigned long long*)sl.h_step, (const int*)dh, sl.h_hdr);
#ifdef CLOOPS_DEVEL
        {
            // developer probe: the step's last (tiny, idempotent) kernel launched CLOOPS_DUP more times -- what a small launch costs the SWEEP
            static const int dup = getenv("CLOOPS_DUP") ? atoi(getenv("CLOOPS_DUP")) : 0;
            for (int k = 0; k < dup; ++k)
                hipLaunchKernelGGL(k7_reduce_parts, dim3(1), dim3(256), 0, c->stream, (const K7Part*)parts, k7b, (K7Part*)(ds + 16),
                                   (const int*)bcount, nb, (long long*)ds, (const unsigned long long*)ds, (int)(out_bytes / 8),
                                   (unsigned long long*)sl.h_step, (const int*)dh, sl.h_hdr);
        }
#endif
        sl.host_written = labels_out == nullptr;        // nothing left for the copy stream
        sl.fine_lo = c->pending_fine_lo;
        sl.step_valid = true;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(sl.ev_done, c->stream));
    ev_record(c, 6);
    sl.wait_done = sl.host_written && !c->profiling;
    if (sl.wait_done) {
        // sweep step: k7_reduce_parts has stored header and step output in pinned host memory; completion = the run's own event
    } else {
        if (c->copy_mode == 2 && !c->shared_copy && !(c->shared_copy = paired_copy_stream(c->stream))) return fail(CL_ERR_HIP, "hipStreamCreate(copy)");
        hipStream_t cs = c->copy_mode == 1 ? c->stream : (c->copy_mode == 2 ? c->shared_copy : c->copy_stream);
        if (cs != c->stream) HIP_TRY(hipStreamWaitEvent(cs, sl.ev_done, 0));
        HIP_TRY(hipMemcpyAsync(sl.h_hdr, dh, 32, hipMemcpyDeviceToHost, cs));
        if (sl.step_valid) HIP_TRY(hipMemcpyAsync(sl.h_step, sl.d_step.p, 16 + sizeof(K7Part) + K7_LOGBINS * 8 + K7_FINE * 8, hipMemcpyDeviceToHost, cs));
        if (labels_out) HIP_TRY(hipMemcpyAsync(labels_out, sl.labels.p, (size_t)n * 4, hipMemcpyDeviceToHost, cs));
        if (c->profiling) (void)hipEventRecord(sl.ev[7], cs);
        HIP_TRY(hipEventRecord(sl.ev_copied, cs));
    }
    sl.pending = true;
    sl.n_strips = n_strips;
    sl.labels_out = labels_out;
    c->enq++;
    c->cur ^= 1;
    return CL_OK;
}

static int finish_wait(cl_chrom* c, int32_t* n_clusters, int32_t* max_label)
{
    if (c->deq == c->enq) return fail(CL_ERR_ARG, "cl_wait: no run in flight");
    const int w = c->deq & 1;
    cl_chrom::Slot& sl = c->slot[w];
#ifdef CLOOPS_DEVEL
    static const bool dbg_wait = getenv("CLOOPS_DBG_WAIT") != nullptr;
#else
    const bool dbg_wait = false;
#endif
    timespec ts0{}, ts1{};
    if (dbg_wait) {
        clock_gettime(CLOCK_MONOTONIC, &ts0);
        hipError_t q1 = hipEventQuery(sl.ev_done), q2 = hipEventQuery(sl.ev_copied);
        fprintf(stderr, "[wait] run %d slot %d: compute_done=%d copied=%d next_done=%d\n", c->deq, w, q1 == hipSuccess, q2 == hipSuccess,
                (int)(hipEventQuery(c->slot[w ^ 1].ev_done) == hipSuccess));
    }
    HIP_TRY(hipEventSynchronize(sl.wait_done ? sl.ev_done : sl.ev_copied));
    if (dbg_wait) {
        clock_gettime(CLOCK_MONOTONIC, &ts1);
        fprintf(stderr, "[wait]   ev_copied after %.0f us; next_done=%d\n", (ts1.tv_sec - ts0.tv_sec) * 1e6 + (ts1.tv_nsec - ts0.tv_nsec) / 1e3,
                (int)(hipEventQuery(c->slot[w ^ 1].ev_done) == hipSuccess));
    }
    sl.pending = false;
    c->deq++;
#ifdef CLOOPS_DEVEL
    if (getenv("CLOOPS_DBG_COUNTERS")) {                 // developer build: the device counters of the run that has just completed
        int h[8] = {0};
        (void)hipMemcpy(h, c->counters.p, sizeof(h), hipMemcpyDeviceToHost);
        fprintf(stderr, "[counters] n=%lld uncertain=%d records=%d overflow=%d roots=%d listed=%d\n", (long long)c->n, h[CTR_NU], h[CTR_NREC], h[CTR_OVERFLOW], h[CTR_NROOT], h[CTR_NFLAG]);
    }
#endif
    const int K = sl.h_hdr[0];
    if (sl.pairs_host) {
        // the (row, label) pairs of a cl_cluster_pairs_async run: their number is in the header that has just arrived
        int2* dst = sl.pairs_host;
        sl.pairs_host = nullptr;
        const long long kp = sl.h_hdr[6];
        // the kernel clamps its stores at the capacity but keeps counting: a run that labelled more PETs than the caller's
        // buffer holds is an argument error, nothing is copied (the staging buffer holds `capacity` pairs, not kp)
        if (kp > sl.pairs_host_cap)
            return fail(CL_ERR_ARG, "cl_cluster_pairs_async: the run labelled more PETs than capacity_pairs (n always suffices)");
        if (kp > 0 && sl.h_hdr[1] == 0) {
            HIP_TRY(hipMemcpyAsync(dst, sl.pairs.p, (size_t)kp * 8, hipMemcpyDeviceToHost, c->aux_stream));
            // (cl_set_pairs_defer: the caller completes the copy with cl_pairs_sync -- a loop over many handles then has all their
            //  copies in flight together instead of one engine's worth at a time)
            if (c->pairs_defer) c->pairs_copy_pending = true;
            else HIP_TRY(hipStreamSynchronize(c->aux_stream));
        }
    }
    if (sl.mask_host) {
        // cl_cluster_rowmask_async: ceil(n / 64) mask words, then one label per set bit (their number: header word 6), one copy
        void* dst = sl.mask_host;
        sl.mask_host = nullptr;
        const long long kp = sl.h_hdr[6];
        if (kp > sl.mask_host_cap)
            return fail(CL_ERR_ARG, "cl_cluster_rowmask_async: the run labelled more PETs than capacity_labels (n always suffices)");
        if (sl.h_hdr[1] == 0) {
            const size_t nw = (size_t)((c->n + 63) / 64);
            HIP_TRY(hipMemcpyAsync(dst, sl.pairs.p, nw * 8 + (size_t)kp * 4, hipMemcpyDeviceToHost, c->aux_stream));
            if (c->pairs_defer) c->pairs_copy_pending = true;
            else HIP_TRY(hipStreamSynchronize(c->aux_stream));
        }
    }
    if (sl.h_hdr[1] != 0)
        return fail(CL_ERR_HIP, sl.h_hdr[1] == 8 ? "internal: the number of PETs that passed the cut differs from the host's count"
                                : sl.h_hdr[1] == 4 ? "internal: strip longer than the hybrid sort accepts"
                                                 : "internal: release-record overflow (border point with > 4 adjacent components)");
    // the table rows are already in the slot's pinned cache (k_export_table); only if that cache was
    // too small (K > capacity, reported in the header) grow it and fetch the rows with a copy
    int nc = sl.h_hdr[3], ml = sl.h_hdr[4];
    if (sl.h_hdr[5] != 0) {
        if (sl.h_boxes) (void)hipHostFree(sl.h_boxes);
        sl.h_boxes = nullptr; sl.h_boxes_cap = 0;
        size_t cap = (size_t)K + (size_t)K / 4 + 1024;
        HIP_TRY(hipHostMalloc((void**)&sl.h_boxes, cap * sizeof(cl_box), hipHostMallocDefault));
        sl.h_boxes_cap = cap;
        // slow path (first run with very many clusters): export again into the larger buffer
        int* dh = c->hdr.as<int>() + 16 * w;
        int reset[3] = {0, -1, 0};
        HIP_TRY(hipMemcpyAsync(dh + 3, reset, 12, hipMemcpyHostToDevice, c->aux_stream));
        hipLaunchKernelGGL(k_export_table, dim3(256), dim3(TPB), 0, c->aux_stream, dh, make_table_slot(c, w), sl.h_boxes, (int)std::min<size_t>(cap, 0x7fffffff));
        HIP_TRY(hipMemcpyAsync(sl.h_hdr, dh, 32, hipMemcpyDeviceToHost, c->aux_stream));
        HIP_TRY(hipStreamSynchronize(c->aux_stream));
        nc = sl.h_hdr[3]; ml = sl.h_hdr[4];
    }
    if (n_clusters) *n_clusters = nc;
    if (max_label) *max_label = ml;
    c->last_K = ml + 1;
    c->last_slot = w;
    c->have_result = true;
    c->k7_classified = sl.step_valid;                    // the step tail has classified this run's table already
    if (sl.step_valid) {
        const long long ni = ((const long long*)sl.h_step)[0];
        if (c->cand_n + ni > c->cand_cap) return fail(CL_ERR_GRID, "internal: candidate buffer overrun");
        c->cand_n += ni;
    }
    if (c->profiling) {
        cl_timing& tm = c->timing;
        memset(&tm, 0, sizeof(tm));
        hipEvent_t* ev = sl.ev;
        (void)hipEventElapsedTime(&tm.ms_keys, ev[0], ev[1]);
        (void)hipEventElapsedTime(&tm.ms_sort, ev[1], ev[2]);
        (void)hipEventElapsedTime(&tm.ms_region, ev[2], ev[3]);
        (void)hipEventElapsedTime(&tm.ms_union, ev[3], ev[4]);
        (void)hipEventElapsedTime(&tm.ms_border, ev[4], ev[5]);
        (void)hipEventElapsedTime(&tm.ms_table, ev[5], ev[6]);
        (void)hipEventElapsedTime(&tm.ms_d2h, ev[6], ev[7]);
        (void)hipEventElapsedTime(&tm.ms_total, ev[0], ev[7]);
        tm.n_in = sl.h_hdr[2];
        tm.n_strips = sl.n_strips;
        tm.ms_bracket = c->ev_bracket_ms;
        tm.ms_band = 0.f;
        if (sl.band_timed) (void)hipEventElapsedTime(&tm.ms_band, ev[8], ev[9]);
        tm.n_queried = sl.n_queried;
    }
    return CL_OK;
}


static int run_rotated(cl_chrom* c, int variant, int eps, int minPts, int cut, int32_t* labels_out);




extern "C" int cl_cluster_weighted(cl_chrom* c, int32_t eps, int32_t min_pts, int32_t wx, int32_t wy, int32_t* labels_out,
                                   int32_t* n_clusters, int32_t* max_label)
{
    int rc = check_args(c, eps, min_pts, 0);
    if (rc) return rc;
    if (wx < 1 || wy < 1 || wx > 4096 || wy > 4096) return fail(CL_ERR_ARG, "cl_cluster_weighted: weights must be in 1..4096");
    if (c->enq != c->deq) return fail(CL_ERR_ARG, "cl_cluster_weighted: asynchronous runs still in flight, call cl_wait first");
    if (c->n == 0) return fail(CL_ERR_EMPTY, "empty input (reference raises IndexError)");
    HIP_TRY(hipSetDevice(c->device));
    c->have_result = false;
    c->last_K = 0;
    if ((rc = run_weighted(c, eps, min_pts, wx, wy, labels_out))) return rc;
    return finish_wait(c, n_clusters, max_label);
}

extern "C" int cl_cluster_async(cl_chrom* c, int variant, int32_t eps, int32_t min_pts, int32_t cut, int32_t* labels_out)
{
    int rc = check_args(c, eps, min_pts, cut);
    if (rc) return rc;
    if (variant != CL_VARIANT_CDBSCAN1 && variant != CL_VARIANT_CDBSCAN2 && variant != CL_VARIANT_BLOCK)
        return fail(CL_ERR_ARG, "unknown variant");
    if (c->enq - c->deq >= 2) return fail(CL_ERR_ARG, "cl_cluster_async: two runs already in flight, call cl_wait first");
    if (c->n == 0) {
        // cDBSCAN.py:77 / blockDBSCAN.py:74: mat[0] on an empty mat raises; cDBSCAN2 returns {}
        if (variant != CL_VARIANT_CDBSCAN2 && cut <= 0) return fail(CL_ERR_EMPTY, "empty input (reference raises IndexError)");
        return fail(CL_ERR_ARG, "cl_cluster_async: empty chromosome (use cl_cluster)");
    }
    HIP_TRY(hipSetDevice(c->device));
    if (variant == CL_VARIANT_BLOCK) return run_block(c, eps, min_pts, cut, labels_out);
    return run_rotated(c, variant, eps, min_pts, cut, labels_out);
}

extern "C" int cl_cluster_pairs_async(cl_chrom* c, int variant, int32_t eps, int32_t min_pts, int32_t cut, int32_t* pinned_pairs_out, int64_t capacity_pairs)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    if (!pinned_pairs_out || capacity_pairs <= 0) return fail(CL_ERR_ARG, "cl_cluster_pairs_async: no pair buffer");
    if (variant != CL_VARIANT_CDBSCAN1 && variant != CL_VARIANT_CDBSCAN2) return fail(CL_ERR_ARG, "cl_cluster_pairs_async: rotated variants only");
    if (c->traversal < 3 || min_pts < 2 || min_pts > 128) return fail(CL_ERR_ARG, "cl_cluster_pairs_async: needs the list form of the run (traversal level >= 3, minPts 2 .. 128)");
    c->pairs_out = (int2*)pinned_pairs_out;
    c->pairs_cap = capacity_pairs;
    const int rc = cl_cluster_async(c, variant, eps, min_pts, cut, nullptr);
    c->pairs_out = nullptr; c->pairs_cap = 0;
    return rc;
}
extern "C" int cl_cluster_rowmask_async(cl_chrom* c, int variant, int32_t eps, int32_t min_pts, int32_t cut, void* pinned_out, int64_t capacity_labels)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    if (!pinned_out || capacity_labels <= 0) return fail(CL_ERR_ARG, "cl_cluster_rowmask_async: no output buffer");
    if (variant != CL_VARIANT_CDBSCAN1 && variant != CL_VARIANT_CDBSCAN2) return fail(CL_ERR_ARG, "cl_cluster_rowmask_async: rotated variants only");
    if (c->traversal < 3 || min_pts < 2 || min_pts > 128) return fail(CL_ERR_ARG, "cl_cluster_rowmask_async: needs the list form of the run (traversal level >= 3, minPts 2 .. 128)");
    c->mask_out = pinned_out;
    c->mask_cap = capacity_labels;
    const int rc = cl_cluster_async(c, variant, eps, min_pts, cut, nullptr);
    c->mask_out = nullptr; c->mask_cap = 0;
    return rc;
}
extern "C" void cl_set_pairs_defer(cl_chrom* c, int enabled) { if (c) c->pairs_defer = enabled != 0; }
extern "C" int cl_pairs_sync(cl_chrom* c)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    if (c->pairs_copy_pending) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipStreamSynchronize(c->aux_stream));
        c->pairs_copy_pending = false;
    }
    return CL_OK;
}
extern "C" int64_t cl_last_n_labelled(const cl_chrom* c)
{
    if (!c || !c->have_result || c->last_slot < 0) return -1;
    return c->slot[c->last_slot].h_hdr[6];               // (may exceed the capacity of a run cl_wait refused with CL_ERR_ARG)
}

extern "C" int cl_cluster_step_async(cl_chrom* c, int variant, int32_t eps, int32_t min_pts, int32_t cut, int32_t step, int64_t fine_lo)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    if (step < 0) return fail(CL_ERR_ARG, "cl_cluster_step_async: step must be >= 0");
    if (c->enq != c->deq) return fail(CL_ERR_ARG, "cl_cluster_step_async: one sweep step in flight per chromosome");
    c->pending_step = step;
    c->pending_cut = cut;
    c->pending_fine_lo = (fine_lo >= 0 && fine_lo < (1LL << 31)) ? fine_lo : -1;
    const int rc = cl_cluster_async(c, variant, eps, min_pts, cut, nullptr);
    c->pending_step = -1;
    return rc;
}

extern "C" int cl_step_result(cl_chrom* c, int64_t* n_inter, int64_t* n_self, cl_dsummary* out)
{
    if (!c || !out) return fail(CL_ERR_ARG, "cl_step_result: null argument");
    if (!c->have_result || c->last_slot < 0 || !c->slot[c->last_slot].step_valid)
        return fail(CL_ERR_ARG, "cl_step_result: the last completed run was not a sweep step");
    const char* h = c->slot[c->last_slot].h_step;
    if (n_inter) *n_inter = ((const long long*)h)[0];
    if (n_self) *n_self = ((const long long*)h)[1];
    memset(out, 0, sizeof(*out));
    out->xshift = K7_XSHIFT;
    const K7Part* part = (const K7Part*)(h + 16);           // reduced on the device in a fixed order (k7_reduce_block)
    for (int g = 0; g < 2; ++g) { out->sumx[g] = part->sx[g]; out->sumxx[g] = part->sxx[g]; out->n_all[g] = part->n_all[g]; out->n_pos[g] = part->n_pos[g]; }
    memcpy(out->loghist, h + 16 + sizeof(K7Part), K7_LOGBINS * 8);
    out->fine_lo = c->slot[c->last_slot].fine_lo;
    if (out->fine_lo >= 0) memcpy(out->fine, h + 16 + sizeof(K7Part) + K7_LOGBINS * 8, K7_FINE * 8);
    return CL_OK;
}

extern "C" int cl_wait(cl_chrom* c, int32_t* n_clusters, int32_t* max_label)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    if (n_clusters) *n_clusters = 0;
    if (max_label) *max_label = -1;
    HIP_TRY(hipSetDevice(c->device));
    return finish_wait(c, n_clusters, max_label);
}

extern "C" int cl_cluster(cl_chrom* c, int variant, int32_t eps, int32_t min_pts, int32_t cut,
                          int32_t* labels_out, int32_t* n_clusters, int32_t* max_label)
{
    int rc = check_args(c, eps, min_pts, cut);
    if (rc) return rc;
    if (n_clusters) *n_clusters = 0;
    if (max_label) *max_label = -1;
    if (c->enq != c->deq) return fail(CL_ERR_ARG, "cl_cluster: asynchronous runs still in flight, call cl_wait first");
    if (c->n == 0) {
        if (variant != CL_VARIANT_CDBSCAN1 && variant != CL_VARIANT_CDBSCAN2 && variant != CL_VARIANT_BLOCK)
            return fail(CL_ERR_ARG, "unknown variant");
        // cDBSCAN.py:77 / blockDBSCAN.py:74: mat[0] on an empty mat raises; cDBSCAN2 returns {}
        if (variant != CL_VARIANT_CDBSCAN2 && cut <= 0) return fail(CL_ERR_EMPTY, "empty input (reference raises IndexError)");
        c->have_result = true; c->last_K = 0; c->last_slot = -1;
        return CL_OK;
    }
    c->have_result = false;
    c->last_K = 0;
    rc = cl_cluster_async(c, variant, eps, min_pts, cut, labels_out);
    if (rc) return rc;
    return finish_wait(c, n_clusters, max_label);
}

// A run that fails half-way (allocation, a HIP error between two launches) must not leave state behind that a later run would
// trust: the count cache it may have marked valid before its words were written, the tickets of the in-kernel scans and the
// superblock sums of the list kernels (all "zero between kernels").
struct RunGuard {
    cl_chrom* c; bool armed = true;
    explicit RunGuard(cl_chrom* cc) : c(cc) {}
    ~RunGuard()
    {
        if (!armed) return;
        c->rc.valid = false;
        c->l_sup_dirty = true;
        if (c->counters.p) (void)hipMemsetAsync(c->counters.as<int>() + CTR_TICKET_A, 0, 8, c->stream);
        (void)hipGetLastError();
    }
};

static int run_rotated(cl_chrom* c, int variant, int eps, int minPts, int cut, int32_t* labels_out)
{
    int rc;
    GridParams g;
    RunGuard guard(c);
#ifdef CLOOPS_DEVEL
    // developer build: host time of the enqueue, by section (CLOOPS_TRACE_ENQ=<ms> prints the calls above that)
    static const double trace_ms = getenv("CLOOPS_TRACE_ENQ") ? atof(getenv("CLOOPS_TRACE_ENQ")) : -1.0;
    struct EnqTrace {
        double t[4]; int k = 0; double lim; long long n;
        static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
        void mark() { if (k < 4) t[k++] = now(); }
        ~EnqTrace() { mark(); if (lim >= 0 && k == 4 && t[3] - t[0] > lim) fprintf(stderr, "[enq n=%lld] grid+workspace %.2f ms | sort+K2 %.2f ms | rest %.2f ms\n", n, t[1] - t[0], t[2] - t[1], t[3] - t[2]); }
    } tr; tr.lim = trace_ms; tr.n = c->n; tr.mark();
#define ENQ_MARK() tr.mark()
#else
#define ENQ_MARK() do { } while (0)
#endif
    if ((rc = make_grid(c, variant, eps, minPts, cut, &g))) return rc;
    if ((rc = ensure_workspace(c, g.S))) return rc;
    if ((rc = ensure_events(c))) return rc;
    ENQ_MARK();
    const int n = (int)c->n;
    int* counters = c->counters.as<int>();
    // tile shape of the traversal kernels.  The wide shape (1024 PETs + 512 halo: long strips stay in LDS) measured SLOWER
    // than the narrow one on the dense workloads (chr1 of the 200 M genome, eps 5000-10000: K3 +7 %, K4 +12..22 %): the
    // walks are bound by the candidates they touch, not by where those live -- it stays a developer knob.
    int wide = 0;
#ifdef CLOOPS_DEVEL
    if (const char* e = getenv("CLOOPS_TILE_WIDE")) wide = atoi(e);
#endif
    const int tile_nt = wide == 1 ? 1024 : TPB;
    int ntiles = 0, tgrid = 0;                          // set once the number of PETs that pass the cut is known
#define TILE_LAUNCH_H(halo, kernel, ...)                                                                             \
    do {                                                                                                             \
        if (wide == 1) hipLaunchKernelGGL((kernel<1024, 512>), dim3(tgrid), dim3(1024), 0, c->stream, __VA_ARGS__);  \
        else if ((halo) == 512) hipLaunchKernelGGL((kernel<TPB, 512>), dim3(tgrid), dim3(TPB), 0, c->stream, __VA_ARGS__); \
        else if ((halo) == 256) hipLaunchKernelGGL((kernel<TPB, 256>), dim3(tgrid), dim3(TPB), 0, c->stream, __VA_ARGS__); \
        else hipLaunchKernelGGL((kernel<TPB, 128>), dim3(tgrid), dim3(TPB), 0, c->stream, __VA_ARGS__);              \
    } while (0)
#define TILE_LAUNCH(kernel, ...) TILE_LAUNCH_H(wide == 2 ? 512 : (wide == 6 ? 256 : 128), kernel, __VA_ARGS__)

    const int nw = (n + 31) / 32;                       // words of the key bitmap (K5); one more word takes the scan's total
    // the bitmap and the counters are cleared by the first kernel of the cut compaction (k_cut_strips); a run without one clears
    // them in a launch of its own
    c->init_nclr = nw + 1;
    // row-aligned labels only when somebody reads them: k_final_labels then writes the label (or -1) of every PET that
    // entered DBSCAN and only the rows removed by the cut filter need the -1 fill
    const bool rows = labels_out != nullptr || c->device_labels || c->pairs_out != nullptr || c->mask_out != nullptr;
    // how far the run works on lists (k_lists.hip; cl_set_traversal): 0 = tile kernels over every PET, 1 = K3 on the core list,
    // 2 = + the border rule on the walker list, 3 = + labels / table / distance list from the lists (only labelled PETs are
    // written: the row-aligned array is filled with -1 first)
    if (rows && !c->pairs_out && (cut > 0 || (wide == 0 && c->traversal >= 3))) HIP_TRY(hipMemsetAsync(c->slot[c->cur].labels.p, 0xFF, (size_t)n * 4, c->stream));
    c->run_rows = rows;
    const int trav_saved = c->traversal;
    if (wide != 0) c->traversal = 0;                    // (developer tile shapes: the tile kernels)
    rc = run_sort_and_count(c, g, false);
    c->traversal = trav_saved;
    if (rc) return rc;
    const int level = c->run_level;
    // (a level-3 default that fell to the tile kernels for this run -- exact counts are not a clustering run -- cannot happen here)
    if (c->init_nclr > 0) { LAUNCH(k_init_flags, nw + 1, nw, c->flag.as<int>(), counters); c->init_nclr = 0; }
    const WordSrc ws = c->ws;                           // where the K2 words of this run live (the handle's count cache / the work buffer)
    if ((rc = c->rootlist.ensure((size_t)n * 4)) || (rc = c->cflag8.ensure((size_t)n + 16))) return rc;
    unsigned char* cflag = c->cflag8.as<unsigned char>();
    ENQ_MARK();
    {
        cl_chrom::Slot& sl = c->slot[c->cur];
        sl.rows_valid = rows && !c->pairs_out; sl.sorted_src = g.swap != 0; sl.k7_sv = c->w_sv; sl.k7_v0 = g.V0; sl.k7_lcnt = nullptr;
        // variant 2 hands an id only to a live cluster, which has >= minPts members (cDBSCAN2.py:180-185): K <= n / minPts;
        // variant 1 numbers every component, dropped ones included (cDBSCAN.py:136-152): K <= n
        sl.kmax = (variant == CL_VARIANT_CDBSCAN2 && minPts >= 1) ? n / minPts + 1 : n;
    }
    int* strip = c->w_strip;
    int* sv = c->w_sv;
    int* sa = c->w_sa;
    const u32* srow = c->srow;
    // everything behind the sort works on the nm = M PETs that passed the cut (known on the host from the upload's distance
    // histogram; nm = n when it is not): grids, tiles and scans are sized by it
    const int nm = std::max(1, c->run_m);
    ntiles = nblocks(nm, tile_nt);
    tgrid = tile_grid(ntiles);

    // K3
    int* pmax32 = nullptr;
    ListRun L{};
    c->dbg_g = g; c->dbg_nm = nm;
    if (level >= 1) {
        if ((rc = (level >= 4 ? lists_build_base(c, g, nm, &L) : lists_build(c, g, nm, &L))) || (rc = lists_union_flatten(c, g, nm, L))) return rc;
        if (level == 1 && (rc = lists_scatter_root(c, nm, L))) return rc;
    } else {
        // own-strip chains; variant 2: the same tile kernel also finds every PET's cell head
        int* head = variant == CL_VARIANT_CDBSCAN2 ? c->head.as<int>() : nullptr;
        if (wide == 0) {
            const int nt_c = nblocks(nm, 1024);
            if (!SKIP(32)) hipLaunchKernelGGL((k_chain_flags<1024, 128, TPB>), dim3(tile_grid(nt_c)), dim3(TPB), 0, c->stream, g, nt_c, nm, sv, sa, strip, ws,
                               cflag, head, c->chainhead.as<int>() /* wavelast: the buffer is free until the labels */, srow, c->cellfirst.as<int>());
        } else
        TILE_LAUNCH(k_chain_flags, g, ntiles, nm, sv, sa, strip, ws, cflag,
                           head, c->chainhead.as<int>() /* wavelast: the buffer is free until the labels */, srow, c->cellfirst.as<int>());
        // long strips (dense data at large eps): 32-PET block summaries for the union scan (`hi` is free until K4)
        pmax32 = ((long long)n > 64LL * g.S) ? c->hi.as<int>() : nullptr;
        if (!SKIP(64)) LAUNCH(k_chain_parent, (nm + CP_PER - 1) / CP_PER, strip, g.S, (const unsigned char*)cflag, c->chainhead.as<int>(), c->parent.as<int>(), c->chainflag.as<int>(),
               c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(), c->usize.as<int>(), c->state.as<int>(),
               sv, c->lo.as<int>(), sa, pmax32);   // chain ends live in `lo` until the release fix-up reuses it
    }
    // the union walk looks one strip back, i.e. about one strip population in front of the PET: a 256-PET halo keeps most of
    // those windows in LDS on dense data (chr1 of the 200 M genome, eps 5000-10000: -12..-16 %); short strips stay with 128
    const int union_halo = (long long)n > 40LL * g.S ? 256 : 128;
    if (level >= 1) { }
    else if (wide == 0) {
        const int nt_u = nblocks(nm, 1024);
        if (SKIP(2)) { } else if (union_halo == 256) hipLaunchKernelGGL((k_union_cores<1024, 256, TPB>), dim3(tile_grid(nt_u)), dim3(TPB), 0, c->stream, g, nt_u, sv, sa, strip,
                                                  c->chainflag.as<int>(), c->lo.as<int>(), pmax32, c->parent.as<int>());
        else hipLaunchKernelGGL((k_union_cores<1024, 128, TPB>), dim3(tile_grid(nt_u)), dim3(TPB), 0, c->stream, g, nt_u, sv, sa, strip,
                                c->chainflag.as<int>(), c->lo.as<int>(), pmax32, c->parent.as<int>());
    } else
    TILE_LAUNCH_H((wide == 2 || wide == 4) ? 512 : union_halo, k_union_cores, g, ntiles, sv, sa, strip, c->chainflag.as<int>(), c->lo.as<int>(),
                       pmax32, c->parent.as<int>());
    if (level == 0 && !SKIP(4)) hipLaunchKernelGGL(k_flatten, dim3(nblocks(nm, BIGTPB * FLAT_PER)), dim3(BIGTPB), 0, c->stream, g, strip, (const int*)nullptr, (const int*)c->chainflag.as<int>(),
           c->parent.as<int>(), srow, c->head.as<int>(), c->cellfirst.as<int>(),
           c->root.as<int>(), c->compkey.as<int>(), c->ncore.as<int>(), c->rootlist.as<int>(), counters);
    int* rootlist = c->rootlist.as<int>();
    ev_record(c, 4);
    // K4
    int* clist = c->chainflag.as<int>();                // the contested border points k_emit_records looks at (the chain ids of K3 are dead)
    if (level >= 2) { if ((rc = lists_border(c, g, nm, L))) return rc; }
    else if (wide == 0) {
        // 1024 PETs per workgroup of 256 threads (4 per thread in the first pass, the walkers of the whole tile in one list)
        const int nt_b = nblocks(std::max(1, c->run_m), 1024);
        if (!SKIP(1)) hipLaunchKernelGGL((k_border<1024, 128, TPB>), dim3(tile_grid(nt_b)), dim3(TPB), 0, c->stream, g, nt_b, sv, sa, strip, c->root.as<int>(),
                           c->compkey.as<int>(), c->ncore.as<int>(), srow, c->owner.as<int>(), c->bsize.as<int>(), c->usize.as<int>(), ws, clist, counters);
    } else
    TILE_LAUNCH_H((wide >= 2 && wide <= 4) ? 512 : (wide >= 5 ? 256 : 128), k_border, g, ntiles, sv, sa, strip, c->root.as<int>(), c->compkey.as<int>(),
                       c->ncore.as<int>(), srow, c->owner.as<int>(), c->bsize.as<int>(), c->usize.as<int>(), ws, clist, counters);
    if (variant == CL_VARIANT_CDBSCAN2) {
        const int rec_cap = n;
        hipLaunchKernelGGL(k_mark_uncertain_l, dim3(512), dim3(TPB), 0, c->stream, g, rootlist, c->ncore.as<int>(), c->bsize.as<int>(), c->state.as<int>(),
                           c->ulist.as<int>(), counters);
        if (level >= 2) { if ((rc = lists_emit_records(c, g, nm, L))) return rc; }
        else if (!SKIP(128)) hipLaunchKernelGGL(k_emit_records, dim3(2048), dim3(TPB), 0, c->stream, g, sv, sa, strip, c->root.as<int>(),
                           c->compkey.as<int>(), c->state.as<int>(), c->owner.as<int>(), c->recs.as<Rec>(), rec_cap, counters,
                           (const int*)clist, ws);
        hipLaunchKernelGGL(k_resolve_release, dim3(1), dim3(1024), 0, c->stream, minPts, c->ncore.as<int>(), c->usize.as<int>(), c->state.as<int>(),
                           c->ulist.as<int>(), c->recs.as<Rec>(), c->lo.as<int>(), c->hi.as<int>(), counters);
    }
    ev_record(c, 5);
    // K5
    hipLaunchKernelGGL(k_rank_bits_l, dim3(512), dim3(TPB), 0, c->stream, g, rootlist, counters, c->compkey.as<int>(), c->state.as<int>(), c->flag.as<unsigned>(),
                       variant == CL_VARIANT_CDBSCAN2 ? (const Rec*)c->recs.as<Rec>() : (const Rec*)nullptr, c->owner.as<int>());
    // ranks of the keys: exclusive scan over the popcounts of the bitmap's words, inside one kernel
    const int nblkw = nblocks(nw, 1024);
    if ((rc = c->blk_tmp.ensure(((size_t)nblkw + 2) * 8))) return rc;
    int* wbsum = c->blk_tmp.as<int>();
    int* wboff = wbsum + nblkw + 1;
    hipLaunchKernelGGL(k_rank_scan, dim3(nblkw), dim3(256), 0, c->stream, nw, (const unsigned*)c->flag.as<unsigned>(), c->rankscan.as<int>(), wbsum, wboff,
                       counters + CTR_TICKET_B);
    c->k_total = wboff + nblkw;
    Table t = make_table(c);
    // rlabel reuses the chainhead buffer (free after k_chain_parent); the kernel also resets the table rows of the ids handed out
    hipLaunchKernelGGL(k_root_labels_bits_l, dim3(512), dim3(TPB), 0, c->stream, g, rootlist, counters, c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(),
                       c->state.as<int>(), c->flag.as<unsigned>(), (const int*)c->rankscan.as<int>(), (const int*)wboff, c->chainhead.as<int>(), t, nblkw,
                       c->hdr.as<int>() + 16 * c->cur, c->w_dM ? c->w_dM : (const int*)(strip + g.S));
    c->hdr_packed = true;
    if (level >= 3) {
        c->slot[c->cur].pairs_host = c->pairs_out;
        c->slot[c->cur].pairs_host_cap = c->pairs_cap;
        if (c->pairs_out && (rc = c->slot[c->cur].pairs.ensure((size_t)std::max<long long>(c->pairs_cap, 1) * 8))) return rc;
        if ((rc = lists_final(c, g, nm, L, rows, c->hdr.as<int>() + 16 * c->cur + 6))) return rc;
        cl_chrom::Slot& sl = c->slot[c->cur];
        sl.mask_host = c->mask_out; sl.mask_host_cap = c->mask_cap;
        // (cl_cluster_rowmask_async: the row-aligned labels the kernel above has just written -> mask words + labels in row order,
        //  their number in header word 6 like the pairs')
        if (c->mask_out && (rc = lists_rowmask(c, c->hdr.as<int>() + 16 * c->cur + 6))) return rc;
        sl.k7_lcnt = L.lcnt; sl.k7_sv = c->l_dist.as<int>();      // the distance statistics read the run's lists (K7Src::sorted == 2)
    } else {
        if (c->mask_out) return fail(CL_ERR_ARG, "cl_cluster_rowmask_async: this run did not take the list form");
        if (level == 2 && (rc = lists_scatter_owner(c, nm, L))) return rc;
        if (!SKIP(8)) hipLaunchKernelGGL(k_final_labels, dim3(nblocks(nm, BIGTPB * FINAL_CHUNKS)), dim3(BIGTPB), 0, c->stream, g, strip, sv, sa, srow,
                       level == 2 ? c->l_aux.as<int>() : c->owner.as<int>(),
                       c->chainhead.as<int>(), rows ? c->slot[c->cur].labels.as<int>() : (int*)nullptr, c->slot[c->cur].slab.as<int>(), t);
    }
    HIP_TRY(hipGetLastError());
    rc = finish_enqueue(c, g.S + 2, c->w_dM ? c->w_dM : strip + g.S, labels_out);
    guard.armed = rc != CL_OK;
    return rc;
}


extern "C" int cl_get_boxes(cl_chrom* c, cl_box* boxes_out)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    if (!c->have_result) return fail(CL_ERR_ARG, "cl_get_boxes: no clustering result available");
    const int K = c->last_K;
    if (K <= 0) return CL_OK;
    if (!boxes_out) return fail(CL_ERR_ARG, "boxes_out is null");
    if (c->last_slot >= 0 && !c->slot[c->last_slot].exported) return fail(CL_ERR_ARG, "cl_get_boxes: the run was made with the table export switched off");
    memcpy(boxes_out, c->slot[c->last_slot].h_boxes, (size_t)K * sizeof(cl_box));
    return CL_OK;
}

extern "C" int64_t cl_last_n_in(const cl_chrom* c)
{
    if (!c || !c->have_result || c->last_slot < 0) return 0;
    return c->slot[c->last_slot].h_hdr[2];
}

extern "C" const cl_box* cl_boxes_host(const cl_chrom* c)
{
    if (!c || !c->have_result || c->last_slot < 0 || c->last_K <= 0 || !c->slot[c->last_slot].exported) return nullptr;
    return c->slot[c->last_slot].h_boxes;
}

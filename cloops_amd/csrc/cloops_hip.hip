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
#include "cl_common.h"

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

extern "C" int cl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------------------
// chromosome statistics (once per upload)
// ------------------------------------------------------------------------------------------
struct Stats { int amin, amax, vmin, vmax, xmin, xmax, ymin, ymax; };

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
__global__ void k_dhist(const int* __restrict__ X, const int* __restrict__ Y, long long n, int* __restrict__ hist)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long d = (long long)Y[i] - X[i];
        // distances >= 65536 need no bin (the running sum stops there); counting them would put half of the PETs on one address
        if (d < DCUM_BINS - 1) atomicAdd(&hist[d < 0 ? DCUM_BINS : (int)d], 1);      // slot DCUM_BINS: d < 0 (X > Y rows)
    }
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
__global__ void k_strip_table32(const u32* __restrict__ skeys, int n, int S, int shift, int* __restrict__ strip_start)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
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
template <int NT, int HALO>
__device__ __forceinline__ bool tile_stage(Tile& t, int2* lw, int* lx, int ntiles, int M,
                                           const int* __restrict__ gq, const int* __restrict__ gp,
                                           const int* __restrict__ gx, unsigned long long* lmask = nullptr)
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
        for (int c = threadIdx.x; c < NV; c += NT) {
            const int4 q = gq4[c], p = gp4[c];
            l4[2 * c] = make_int4(q.x, p.x, q.y, p.y);
            l4[2 * c + 1] = make_int4(q.z, p.z, q.w, p.w);
        }
    }
    for (int k = threadIdx.x; k < T_WIN; k += NT) {
        const int gi = base + k;
        const bool in = gi >= 0 && gi < M;
        const int x = in ? gx[gi] : -1;                  // every payload test is `>= (something >= 0)`
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
template <int T_WIN, typename MORE, typename ACC, typename SEE>
__device__ __forceinline__ void tile_walk_from(const Tile& t, const int* __restrict__ gq, const int* __restrict__ gp,
                                               const int* __restrict__ gx, int M, int j, MORE&& more, ACC&& acc, SEE&& see)
{
    const int base = t.w.base;
    int k = j - base;
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
        k += 4;
        while (k < T_WIN) {
            const int knext = (k | 63) + 1;
            unsigned long long bits = t.m[k >> 6] >> (k & 63);
            const int2 cn = lw[min(knext, T_WIN - 1)];
            const bool goes_on = knext >= T_WIN || more(cn);
            while (bits) {                                      // two set bits per round
                const int i0 = k + __ffsll((long long)bits) - 1;
                bits &= bits - 1;
                const bool two = bits != 0;
                const int i1 = two ? k + __ffsll((long long)bits) - 1 : i0;
                bits &= bits - 1;
                const int2 c0 = lw[i0], c1 = lw[i1];
                const int x0 = lx[i0], x1 = lx[i1];
                if (!more(c0)) return;
                if (acc(c0)) see(i0 + base, x0);
                if (two) {
                    if (!more(c1)) return;
                    if (acc(c1)) see(i1 + base, x1);
                }
            }
            if (!goes_on) return;
            k = knext;
        }
        j = k + base;
    }
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
template <typename F>
__device__ __forceinline__ void tile_visit_own(const Tile& t, const int* __restrict__ gq, const int* __restrict__ gx,
                                               int i, int b, int e, int qlo, int qhi, int dirs, F&& f)
{
    if (dirs & 1)
        for (int j = i - 1; j >= b; --j) {
            const bool in = j >= t.wbeg;
            const int q = in ? t.w[j].x : gq[j];
            if (q < qlo) break;
            if (f(j, in ? t.x[j] : gx[j])) break;
        }
    if (dirs & 2)
        for (int j = i + 1; j < e; ++j) {
            const bool in = j < t.wend;
            const int q = in ? t.w[j].x : gq[j];
            if (q > qhi) break;
            if (f(j, in ? t.x[j] : gx[j])) break;
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
template <int NT, int HALO>
__global__ void __launch_bounds__(NT)
k_chain_flags(GridParams g, int ntiles, int n, const int* __restrict__ sv, const int* __restrict__ sa,
              const int* __restrict__ strip_start, const int* __restrict__ cnt, int* __restrict__ chainflag,
              int* __restrict__ head, int* __restrict__ wavelast)
{
    __shared__ __attribute__((aligned(16))) int2 lw[NT + 2 * HALO];
    __shared__ int lx[NT + 2 * HALO];
    const int M = strip_start[g.S];
    if (head) {                                          // filtered tail: singleton cells (keys of the cellfirst scan)
        const int ig = tile_of_block(blockIdx.x) * NT + threadIdx.x;
        if (ig >= M && ig < n) head[ig] = ig;
    }
    Tile t;
    if (!tile_stage<NT, HALO>(t, lw, lx, ntiles, M, sv, sa, cnt)) return;
    const int i = t.t0 + threadIdx.x;
    if (i >= M) return;
    const int2 me = t.w[i];
    if (head) {
        // variant 2: head of the PET's rotated cell (strip, q / eps) = first PET of the sorted order that is
        // neither in an earlier strip nor below the cell's lower q edge -- a bisection on the staged tile
        // instead of head flags + a max-scan over all PETs (variant 2 runs with A0 = V0 = 0)
        const int p0 = me.y & ~(g.peps - 1), q0 = div_eps(g, me.x) * g.eps;      // lower edges of the rotated cell (sp space / q space)
        int pos = t.wbeg;
        constexpr int TOP = (NT + HALO < 512) ? 256 : ((NT + HALO < 1024) ? 512 : 1024);      // 2 * TOP - 1 >= the staged range up to i
        static_assert(2 * TOP > NT + HALO, "cell-head bisection covers the window");
#pragma unroll
        for (int step = TOP; step >= 1; step >>= 1) {
            const int idx = pos + step - 1;
            const int2 c = t.w[min(idx, i)];
            pos = (idx <= i && (c.y < p0 || c.x < q0)) ? pos + step : pos;
        }
        if (pos == t.wbeg && t.wbeg > 0) {               // the cell starts before the staged window (pile-up)
            int lo = 0, hi = t.wbeg;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sa[mid] < p0 || sv[mid] < q0) lo = mid + 1; else hi = mid;
            }
            pos = lo;
        }
        head[i] = pos;
    }
    int f = 0, last = 0;
    if (t.x[i] >= g.minPts) {
        const int s = strip_of(g, me.y);
        const int b = strip_start[s], e = strip_start[s + 1];
        f = i + 1;
        last = 1;
        tile_visit_own(t, sv, cnt, i, b, e, sat_add(me.x, -g.eps), sat_add(me.x, g.eps), 1,
                       [&](int, int cj) { if (cj >= g.minPts) { f = 0; return true; } return false; });
        tile_visit_own(t, sv, cnt, i, b, e, sat_add(me.x, -g.eps), sat_add(me.x, g.eps), 2,
                       [&](int, int cj) { if (cj >= g.minPts) { last = 0; return true; } return false; });
    }
    chainflag[i] = f | (last ? (int)0x80000000u : 0);   // sign bit: last core of its chain (its q is the chain's upper end)
    // wavelast[w] = the last chain-opening PET (+1) among the 64 PETs [64 w, 64 w + 64), 0 = none: what k_chain_parent needs to
    // find a core's chain head without a scan over all PETs (the lanes still here are the wave's PETs below M; lane 0 is one)
    const unsigned long long ob = __ballot(f != 0);
    if ((threadIdx.x & 63) == 0) wavelast[i >> 6] = ob ? i + (64 - __clzll((long long)ob)) : 0;
}
// parent[] = chain head for core points (flat forest to start from); chainid[] = the same for
// core points and -1 for everything else (what the union kernel stages as its payload)
// Every component root is a chain head, so the per-root accumulators are reset here, by the chain
// heads only, instead of memset-ing five N-sized arrays per run.
// pmax32 (optional): the 32-PET block summaries of the union scan (max strip coordinate over the block's CORE PETs, see
// k_union_cores) come out of the same pass -- every thread already knows whether its PET is a core
#define CP_PER 4
__global__ void k_chain_parent(const int* __restrict__ strip_start, int S, const int* __restrict__ cnt, int minPts,
                               const int* __restrict__ wavelast, int* __restrict__ parent, int* chainid /* in: chain flags */,
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
    int ii[CP_PER], fl[CP_PER], cn[CP_PER], spv[CP_PER], qv[CP_PER];
#pragma unroll
    for (int e = 0; e < CP_PER; ++e) {
        ii[e] = (blockIdx.x * CP_PER + e) * (int)blockDim.x + (int)threadIdx.x;
        const bool in = ii[e] < M;
        fl[e] = in ? chainid[ii[e]] : 0;                 // i + 1 if the PET opens a chain, sign bit: last core of its chain
        cn[e] = in ? cnt[ii[e]] : INT_MIN;
        spv[e] = (in && pmax32) ? sa[ii[e]] : INT_MIN;
        qv[e] = (in && fl[e] < 0) ? sv[ii[e]] : 0;
    }
#pragma unroll
    for (int e = 0; e < CP_PER; ++e) {
        const int i = ii[e];
        const bool core = cn[e] >= minPts;
        const unsigned long long open = __ballot((fl[e] & 0x7fffffff) != 0);
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
            if (core) parent[i] = h;                            // (only cores are ever looked up in the forest)
            chainid[i] = h;
            if (h == i) { compkey[i] = INT_MAX; ncore[i] = 0; bsize[i] = 0; usize[i] = 0; state[i] = ST_LIVE; }
            if (core && fl[e] < 0) chain_qend[h] = qv[e];       // indexed by chain head
        }
        if (pmax32) {                                          // uniform: every lane of the wave takes part in the reduction
            int v = core ? spv[e] : INT_MIN;
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

template <int NT, int HALO>
__global__ void __launch_bounds__(NT)
k_union_cores(GridParams g, int ntiles, const int* __restrict__ sv, const int* __restrict__ sa,
              const int* __restrict__ strip_start, const int* __restrict__ chainid, const int* __restrict__ chain_qend,
              const int* __restrict__ pmax32, int* parent)
{
    __shared__ __attribute__((aligned(16))) int2 lw[NT + 2 * HALO];
    __shared__ int lx[NT + 2 * HALO];
    __shared__ short l_list[NT];
    __shared__ int l_wcount[NT / 64];
    const int M = strip_start[g.S];
    Tile t;
    if (!tile_stage<NT, HALO>(t, lw, lx, ntiles, M, sv, sa, chainid)) return;
    const int i0 = t.t0 + threadIdx.x;
    const int total = block_compact<NT>(i0 < M && t.x[i0 < M ? i0 : t.t0] >= 0, l_list, l_wcount);
    if ((int)threadIdx.x >= total) return;
    const int i = t.t0 + l_list[threadIdx.x];
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
        if (rep) uf_unite(parent, A, B);
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
#define BIGTPB 1024
#define AGG_H 512
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

// K3b: root per core point, component keys and core counts.
//   variant 1: key = smallest input row of a core point = the component's start point
//              (cDBSCAN.py:134-137)
//   variant 2: key = smallest cellfirst over the cells holding its core points
//              (cDBSCAN2.py:117-140)
#define FLAT_PER 4          // PETs per thread
__global__ void __launch_bounds__(BIGTPB)
k_flatten(GridParams g, const int* __restrict__ strip_start, const int* __restrict__ cnt,
          int* parent, const u32* __restrict__ srow,
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
        core[e] = ii[e] < M && cnt[ii[e]] >= g.minPts;
    }
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e) {
        x[e] = core[e] ? parent[ii[e]] : -1;            // a core's parent is its chain head to start with
        hd[e] = (core[e] && g.variant == CL_VARIANT_CDBSCAN2) ? head[ii[e]] : 0;
    }
#pragma unroll
    for (int e = 0; e < FLAT_PER; ++e)
        key[e] = !core[e] ? INT_MAX : ((g.variant == CL_VARIANT_CDBSCAN2) ? cellfirst[hd[e]] : (int)srow[ii[e]]);
    {
        // the union kernel has completed (kernel boundary = coherent): plain loads, all chains of the thread step together
        bool todo = false;
#pragma unroll
        for (int e = 0; e < FLAT_PER; ++e) todo |= core[e];
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
                const int sl = agg_slot(hkey, r[e]);
                if (sl >= 0) { atomicMin(&hmin[sl], key[e]); atomicAdd(&hcnt[sl], 1); }
                else { atomicMin(&compkey[r[e]], key[e]); atomicAdd(&ncore[r[e]], 1); }
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
#define OWNER_CONTESTED 0x40000000
__device__ __forceinline__ int owner_root(int o) { return o < 0 ? -1 : (o & (OWNER_CONTESTED - 1)); }

template <int NT, int HALO>
__global__ void __launch_bounds__(NT)
k_border(GridParams g, int ntiles, const int* __restrict__ sv, const int* __restrict__ sa,
         const int* __restrict__ strip_start, const int* __restrict__ root, const int* __restrict__ compkey,
         const int* __restrict__ ncore, const u32* __restrict__ srow, int* __restrict__ owner, int* __restrict__ bsize,
         int* __restrict__ usize, const int* __restrict__ cnt, int* __restrict__ tileflag)
{
    // tileflag[k] = 1 if the 256 PETs [256 k, 256 k + 256) hold a CONTESTED border point (one adjacent to more than one component)
    // whose owner is not live on its cores alone: only such tiles can have anything for k_emit_records, which then leaves after
    // one load instead of testing 256 PETs
    if (threadIdx.x < NT / 256) { const int tl = tile_of_block(blockIdx.x); if (tl < ntiles) tileflag[tl * (NT / 256) + threadIdx.x] = 0; }
    __shared__ __attribute__((aligned(16))) int2 lw[NT + 2 * HALO];
    __shared__ int lx[NT + 2 * HALO];
    __shared__ short l_list[NT];
    __shared__ int l_wcount[NT / 64];
    __shared__ unsigned long long l_mask[(NT + 2 * HALO) / 64];
    __shared__ int l_enc[NT];
    const int M = strip_start[g.S];
    Tile t;
    if (!tile_stage<NT, HALO>(t, lw, lx, ntiles, M, sv, sa, root, l_mask)) return;
    const int i0 = t.t0 + threadIdx.x;
    bool border = false;
    if (i0 < M) {
        const int ri = t.x[i0];
        if (ri >= 0) owner[i0] = ri;
        else {
            // K2 left either the neighbour count of a non-core PET (itself included) or its hint word (k_region_core):
            // nothing within eps -- most of the background noise ends here
            const int enc = cnt[i0];
            l_enc[threadIdx.x] = enc;
            if (enc < 0 ? (((unsigned)enc & K2H_ISOLATED) != 0u) : (enc <= 1)) owner[i0] = -1; else border = true;
        }
    }
    const int total = block_compact<NT>(border, l_list, l_wcount);
    if ((int)threadIdx.x >= total) return;
    const int i = t.t0 + l_list[threadIdx.x];
    const int enc = l_enc[l_list[threadIdx.x]];
    const int2 me = t.w[i];
    const int qlo = sat_add(me.x, -g.eps), qhi = sat_add(me.x, g.eps);
    const bool v1 = g.variant == CL_VARIANT_CDBSCAN1;
    int bestk = INT_MAX, best = -1, tk = -1, tbest = -1, lastr = -1, lastk = 0, first = -1;
    bool contested = false;
    auto see = [&](int j, int r) {
        if (r < 0) return;
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
        constexpr int T_WIN = NT + 2 * HALO;
        const int pbeg = me.y & ~(g.peps - 1), pend = pbeg + g.peps, pend2 = pend + g.peps;
        const int plo = me.y - g.peps, phi = me.y + g.peps;
        {
            const int lb = max(i - (g.minPts - 1), t.w.base), re = min(i + g.minPts, t.w.base + T_WIN);
            const int jl = t.prev_set(i - 1, lb), jr = t.next_set(i + 1, re);
            const bool hl = jl >= lb, hr = jr < re;
            const int2 cl = t.w[hl ? jl : i], cr = t.w[hr ? jr : i];
            const int rl = t.x[hl ? jl : i], rr = t.x[hr ? jr : i];
            if (hl && cl.y >= pbeg && cl.x >= qlo) see(jl, rl);
            if (hr && cr.y < pend && cr.x <= qhi) see(jr, rr);
        }
        const int ja = i - (int)((unsigned)enc & K2H_MASK), jb = i + (int)(((unsigned)enc >> K2H_BITS) & K2H_MASK);
        tile_walk_from<T_WIN>(t, sv, sa, root, M, ja, [&](int2 c) { return (c.y < pbeg) & (c.x <= qhi); },
                              [&](int2 c) { return c.y >= plo; }, see);       // one strip below: sp can only be too low
        tile_walk_from<T_WIN>(t, sv, sa, root, M, jb, [&](int2 c) { return (c.y < pend2) & (c.x <= qhi); },
                              [&](int2 c) { return c.y <= phi; }, see);       // one strip above: only too high
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
        // only a component that is not live on its cores alone can end up uncertain (k_mark_uncertain_l): the tiles without a
        // contested border point of such a component have nothing for k_emit_records
        if (cnt_me && contested) tileflag[i >> 8] = 1;          // (behind the staging barrier: ordered after the reset above)
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

// ---- variant 2 release rule (cDBSCAN2.py:180-183) ------------------------------------------
// A component is surely live if cores + first-come borders >= minPts (its share can only grow
// when lower components die).  The rest form the small uncertain set U, resolved in key order.

__global__ void k_mark_uncertain(GridParams g, const int* __restrict__ strip_start, const int* __restrict__ root,
                                 const int* __restrict__ ncore, const int* __restrict__ bsize,
                                 int* __restrict__ state, int* __restrict__ ulist, int* __restrict__ counters)
{
    const int M = strip_start[g.S];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    if (root[i] != i) return;
    if (ncore[i] + bsize[i] < g.minPts) {
        state[i] = ST_UNKNOWN;
        int k = atomicAdd(&counters[CTR_NU], 1);
        ulist[k] = i;
    }
}

// records: for every border point adjacent to an uncertain component, its (<= 4, geometric
// bound) distinct adjacent components in ascending key order
struct Rec { int pt; int r[4]; };

template <int NT, int HALO>
__global__ void __launch_bounds__(NT)
k_emit_records(GridParams g, int ntiles, const int* __restrict__ sv, const int* __restrict__ sa,
               const int* __restrict__ strip_start, const int* __restrict__ root, const int* __restrict__ compkey,
               const int* __restrict__ state, const int* __restrict__ owner, Rec* __restrict__ recs, int rec_cap,
               int* __restrict__ counters, const int* __restrict__ tileflag, int nflags)
{
    __shared__ __attribute__((aligned(16))) int2 lw[NT + 2 * HALO];
    __shared__ int lx[NT + 2 * HALO];
    __shared__ short l_list[NT];
    __shared__ int l_wcount[NT / 64];
    if (counters[CTR_NU] == 0) return;
    {
        // (k_border's flags are per 256-PET tile; this kernel may run on larger tiles)
        const int tl = tile_of_block(blockIdx.x);
        if (tl >= ntiles) return;
        constexpr int F = NT / 256;
        int any = 0;
#pragma unroll
        for (int k = 0; k < F; ++k) any |= (tl * F + k < nflags) ? tileflag[tl * F + k] : 0;
        if (any == 0) return;
    }
    const int M = strip_start[g.S];
    {
        // Only a CONTESTED border point whose first-come owner (its lowest-key adjacent component) is
        // UNCERTAIN can change hands: the fix-up walks a record's components in key order and a component
        // that is surely live ends the walk (k_resolve_release), so a record that starts with a live component
        // contributes nothing.  Nearly all tiles have no such point and leave before staging anything.
        const int ip = tile_of_block(blockIdx.x) * NT + threadIdx.x;
        const int op = ip < M ? owner[ip] : -1;
        const bool cand = op >= 0 && (op & OWNER_CONTESTED) && state[owner_root(op)] == ST_UNKNOWN;
        if (!__syncthreads_or(cand)) return;
    }
    Tile t;
    if (!tile_stage<NT, HALO>(t, lw, lx, ntiles, M, sv, sa, root)) return;
    const int i0 = t.t0 + threadIdx.x;
    const bool act = i0 < M && t.x[i0] < 0 && owner[i0] >= 0 && (owner[i0] & OWNER_CONTESTED) &&
                     state[owner_root(owner[i0])] == ST_UNKNOWN;
    const int total = block_compact<NT>(act, l_list, l_wcount);
    if ((int)threadIdx.x >= total) return;
    const int i = t.t0 + l_list[threadIdx.x];
    const int2 me = t.w[i];
    const int s = strip_of(g, me.y);
    const int qlo = sat_add(me.x, -g.eps), qhi = sat_add(me.x, g.eps);
    const int b = strip_start[s], e = strip_start[s + 1];
    const int tb = s > 0 ? strip_start[s - 1] : b;
    const int te = s + 1 < g.S ? strip_start[s + 2] : e;
    int rr[4] = {-1, -1, -1, -1};
    int kk[4] = {INT_MAX, INT_MAX, INT_MAX, INT_MAX};
    int nr = 0;
    bool any_u = false, overflow = false;
    auto see = [&](int r) {
        if (r < 0) return;
        for (int q = 0; q < 4; ++q) if (rr[q] == r) return;
        if (nr == 4) { overflow = true; return; }
        const int k = compkey[r];
        int q = nr++;
        while (q > 0 && kk[q - 1] > k) { kk[q] = kk[q - 1]; rr[q] = rr[q - 1]; --q; }
        kk[q] = k; rr[q] = r;
        if (state[r] == ST_UNKNOWN) any_u = true;
    };
    tile_visit_own_all(t, sv, root, i, b, e, qlo, qhi, [&](int, int r) { see(r); });
    tile_visit_segment(t, sv, sa, root, tb, b, qlo, qhi, [&](int, int, int pj, int r) {
        const int da = pj - me.y; if ((da < 0 ? -da : da) <= g.peps) see(r); });
    tile_visit_segment(t, sv, sa, root, e, te, qlo, qhi, [&](int, int, int pj, int r) {
        const int da = pj - me.y; if ((da < 0 ? -da : da) <= g.peps) see(r); });
    if (overflow) atomicExch(&counters[CTR_OVERFLOW], 1);
    if (!any_u) return;
    const int idx = atomicAdd(&counters[CTR_NREC], 1);
    if (idx >= rec_cap) { atomicExch(&counters[CTR_OVERFLOW], 2); return; }
    Rec rec; rec.pt = i;
    for (int q = 0; q < 4; ++q) rec.r[q] = rr[q];
    recs[idx] = rec;
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

// Device cluster table, struct-of-arrays: the five accumulators of one id live in five different cache
// lines, so the atomics of a hot id (a giant component) spread over five L2 channels instead of
// queueing on one line (AoS measured 3x slower on the 16 M-PET giant-component case).
struct Table {
    int* count; int* minx; int* maxx; int* miny; int* maxy;
    __device__ __forceinline__ cl_box get(int k) const
    {
        cl_box b; b.min_x = minx[k]; b.max_x = maxx[k]; b.min_y = miny[k]; b.max_y = maxy[k]; b.count = count[k];
        return b;
    }
};

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

__global__ void k_root_labels_bits_l(GridParams g, const int* __restrict__ rootlist, const int* __restrict__ counters,
                                     const int* __restrict__ compkey, const int* __restrict__ ncore, const int* __restrict__ bsize,
                                     const int* __restrict__ state, const unsigned* __restrict__ bits, const int* __restrict__ wordrank,
                                     int* __restrict__ rlabel, Table t, int nw, int* __restrict__ hdr, const int* __restrict__ d_M)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) {          // k_pack_header rides along (nothing behind this kernel raises a flag)
        hdr[0] = wordrank[nw]; hdr[1] = counters[CTR_OVERFLOW]; hdr[2] = d_M[0]; hdr[3] = 0; hdr[4] = -1; hdr[5] = 0;
    }
    const int K = counters[CTR_NROOT];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < K; k += gridDim.x * blockDim.x) {
        const int i = rootlist[k];
        const bool keep = (g.variant == CL_VARIANT_CDBSCAN2) ? (state[i] != ST_DEAD)
                                                             : (ncore[i] + bsize[i] >= g.minPts);   // cDBSCAN.py:149-152
        const int key = compkey[i];
        rlabel[i] = keep ? wordrank[key >> 5] + __popc(bits[key >> 5] & ((1u << (key & 31)) - 1u)) : -1;
    }
    // k_init_table rides along: the rows of the ids handed out
    const int ids = wordrank[nw];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < ids; k += gridDim.x * blockDim.x) {
        t.count[k] = 0; t.minx[k] = INT_MAX; t.maxx[k] = INT_MIN; t.miny[k] = INT_MAX; t.maxy[k] = INT_MIN;
    }
}

__device__ __forceinline__ int je_minus(const int* __restrict__ cstart, int j) { return cstart[j + 1] - cstart[j]; }
__device__ __forceinline__ int wave_min_i(int v) { return dpp_reduce_wave(v, OpMin()); }
__device__ __forceinline__ int wave_max_i(int v) { return dpp_reduce_wave(v, OpMax()); }

// Cluster table (pipe.py:78-102) by the two-level reduce-by-key above; called by all threads
// of a BIGTPB workgroup (sorted order keeps a cluster's PETs in neighbouring lanes, so a wave
// usually carries a handful of labels).
#define TAB_H 512
struct TableLds { int key[TAB_H], cnt[TAB_H], mnx[TAB_H], mxx[TAB_H], mny[TAB_H], mxy[TAB_H]; };

__device__ __forceinline__ void table_lds_init(TableLds& h)
{
    for (int k = threadIdx.x; k < TAB_H; k += blockDim.x) {
        h.key[k] = -1; h.cnt[k] = 0; h.mnx[k] = INT_MAX; h.mxx[k] = INT_MIN; h.mny[k] = INT_MAX; h.mxy[k] = INT_MIN;
    }
    __syncthreads();
}
__device__ __forceinline__ int tab_slot(int* keys, int key)
{
    unsigned h = ((unsigned)key * 2654435761u) >> 23;            // 9 bits = log2(TAB_H)
    for (int probe = 0; probe < 16; ++probe) {
        const int old = atomicCAS(&keys[h], -1, key);
        if (old == -1 || old == key) return (int)h;
        h = (h + 1) & (TAB_H - 1);
    }
    return -1;
}
// level 1 + insertion into the workgroup's LDS table (no barrier inside: may be called repeatedly)
__device__ __forceinline__ void table_accumulate(const Table& t, TableLds& h, int lab, int x, int y)
{
    const int lane = threadIdx.x & 63;
    const unsigned long long pending = __ballot(lab >= 0);
    if (!pending) return;
    const int leader = __ffsll((long long)pending) - 1;
    const int L = __builtin_amdgcn_readlane(lab, leader);
    const unsigned long long m = __ballot(lab == L);
    if (m == pending && __popcll(m) >= 16) {
        // the wave lies inside one cluster: four reductions, one insertion
        const bool mine = lab == L;
        const int cm = __popcll(m);
        int mnx = wave_min_i(mine ? x : INT_MAX), mxx = wave_max_i(mine ? x : INT_MIN);
        int mny = wave_min_i(mine ? y : INT_MAX), mxy = wave_max_i(mine ? y : INT_MIN);
        if (lane == leader) {
            const int sl = tab_slot(h.key, L);
            if (sl >= 0) {
                atomicAdd(&h.cnt[sl], cm);
                atomicMin(&h.mnx[sl], mnx); atomicMax(&h.mxx[sl], mxx);
                atomicMin(&h.mny[sl], mny); atomicMax(&h.mxy[sl], mxy);
            } else {
                atomicAdd(&t.count[L], cm);
                atomicMin(&t.minx[L], mnx); atomicMax(&t.maxx[L], mxx);
                atomicMin(&t.miny[L], mny); atomicMax(&t.maxy[L], mxy);
            }
        }
    } else if (lab >= 0) {
        // several clusters (and noise) in the wave: every lane straight into the LDS table -- lanes of one cluster meet on
        // one LDS address, which the LDS unit serialises at a fraction of what a loop over the distinct labels costs
        const int sl = tab_slot(h.key, lab);
        if (sl >= 0) {
            atomicAdd(&h.cnt[sl], 1);
            atomicMin(&h.mnx[sl], x); atomicMax(&h.mxx[sl], x);
            atomicMin(&h.mny[sl], y); atomicMax(&h.mxy[sl], y);
        } else {
            atomicAdd(&t.count[lab], 1);
            atomicMin(&t.minx[lab], x); atomicMax(&t.maxx[lab], x);
            atomicMin(&t.miny[lab], y); atomicMax(&t.maxy[lab], y);
        }
    }
}
// level 2 -> global: one set of atomics per key of the workgroup
__device__ __forceinline__ void table_flush(const Table& t, TableLds& h)
{
    __syncthreads();
    for (int k = threadIdx.x; k < TAB_H; k += blockDim.x) {
        if (h.key[k] < 0) continue;
        const int L = h.key[k];
        atomicAdd(&t.count[L], h.cnt[k]);
        atomicMin(&t.minx[L], h.mnx[k]); atomicMax(&t.maxx[L], h.mxx[k]);
        atomicMin(&t.miny[L], h.mny[k]); atomicMax(&t.maxy[L], h.mxy[k]);
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
#pragma unroll
    for (int ch = 0; ch < FINAL_CHUNKS; ++ch) table_accumulate(t, h, lab[ch], x[ch], y[ch]);
    table_flush(t, h);
}


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

// ==========================================================================================
// K6: blockDBSCAN (cLoops/blockDBSCAN.py) -- DBSCAN over grid CELLS
// ==========================================================================================
// Closed form: unrotated eps-grid anchored at the filtered set's (minX,minY)
// (:74-82); cells whose 9-cell population is < minPts and whose existing neighbours are all
// like that are deleted (:101-122); two surviving 8-adjacent cells are LINKED iff their
// float64 centroids are within eps (city block) or some point pair is (:204-239); a cell is
// CORE iff own + linked population >= minPts (:181,191); components of core cells are ranked
// by their first cell in insertion order (= smallest input row of the cell's first point,
// :148-152); a non-core cell linked to core cells takes the LARGEST adjacent rank
// (unconditional overwrite, :195-198); points inherit their cell's label (:154-168).
struct BlkParams {
    int eps, minPts, cut;
    int R;            // rows of the cell-row table; key row R marks filtered PETs
    int n;
    int nyb, rb;      // sort key = ((nx << nyb | ny) << 2*rb) | rx << rb | ry ; only the cell bits are sorted
    u32 magic; int sh1, sh2;      // v / eps by multiply-shift (same constants as GridParams)
};
__device__ __forceinline__ u32 blk_div(const BlkParams& p, u32 n)
{
    const u32 t1 = __umulhi(p.magic, n);
    return (t1 + ((n - t1) >> p.sh1)) >> p.sh2;
}
struct BlkScalars { int minx, miny, M, C; };

__global__ void k_blk_init_scalars(BlkScalars* sc, int minx, int miny)
{
    sc->minx = minx; sc->miny = miny; sc->M = 0; sc->C = 0;
}

__global__ void k_blk_minmax(const int* __restrict__ X, const int* __restrict__ Y, int n, int cut, BlkScalars* sc)
{
    __shared__ int red[2][TPB / 64];
    int mx = INT_MAX, my = INT_MAX;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int x = X[i], y = Y[i];
        if (y - x < cut) continue;
        mx = min(mx, x); my = min(my, y);
    }
    for (int o = 32; o > 0; o >>= 1) { mx = min(mx, __shfl_down(mx, o)); my = min(my, __shfl_down(my, o)); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = mx; red[1][threadIdx.x >> 6] = my; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < TPB / 64; ++w) { mx = min(mx, red[0][w]); my = min(my, red[1][w]); }
        if (mx != INT_MAX) { atomicMin(&sc->minx, mx); atomicMin(&sc->miny, my); }
    }
}

__global__ void k_blk_keys(const int* __restrict__ X, const int* __restrict__ Y, BlkParams p, const BlkScalars* __restrict__ sc,
                           u64* __restrict__ keys, u32* __restrict__ vals)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.n) return;
    int x = X[r], y = Y[r];
    bool valid = (p.cut <= 0) || (y - x >= p.cut);
    // nx = int((X - minX) / cw) + 1 (:81-82); the +1 is dropped (cells are only compared)
    u64 key = (u64)(u32)p.R << (p.nyb + 2 * p.rb);
    if (valid) {
        const u32 ux = (u32)(x - sc->minx), uy = (u32)(y - sc->miny);
        const u32 nx = blk_div(p, ux), ny = blk_div(p, uy);
        const u32 rx = ux - nx * (u32)p.eps, ry = uy - ny * (u32)p.eps;
        key = ((((u64)nx << p.nyb) | ny) << (2 * p.rb)) | ((u64)rx << p.rb) | ry;
    }
    keys[r] = key;
    vals[r] = (u32)r;
}

// decode the sorted keys back into coordinates (no gather), mark cell heads
__global__ void k_blk_gather(BlkParams p, const u64* __restrict__ skeys,
                             int* __restrict__ sx, int* __restrict__ sy, int* __restrict__ headflag, BlkScalars* sc)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.n) return;
    const int cb = 2 * p.rb;
    const u64 k = skeys[i];
    const u64 cell = k >> cb;
    const u32 nx = (u32)(cell >> p.nyb), ny = (u32)(cell & ((1ull << p.nyb) - 1));
    const bool valid = nx < (u32)p.R;
    const u32 rmask = (1u << p.rb) - 1;
    sx[i] = sc->minx + (int)(nx * (u32)p.eps + ((u32)(k >> p.rb) & rmask));
    sy[i] = sc->miny + (int)(ny * (u32)p.eps + ((u32)k & rmask));
    const u64 prev = i ? (skeys[i - 1] >> cb) : ~0ull;
    headflag[i] = (valid && prev != cell) ? 1 : 0;
    if (!valid && (i == 0 || (u32)(prev >> p.nyb) < (u32)p.R)) sc->M = i;   // first filtered row
    if (valid && i == p.n - 1) sc->M = p.n;
}

// cid[i] = (inclusive prefix sum of headflag)[i] - 1 ; per-cell arrays
__global__ void k_blk_cells(BlkParams p, const u64* __restrict__ skeys, const int* __restrict__ headflag,
                            const int* __restrict__ cidp1, const u32* __restrict__ srow, BlkScalars* sc,
                            int* __restrict__ cstart, u64* __restrict__ ckey, int* __restrict__ cfirst)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int M = sc->M;
    if (i >= M) return;
    if (headflag[i]) {
        int c = cidp1[i] - 1;
        cstart[c] = i;
        ckey[c] = skeys[i] >> (2 * p.rb);
        cfirst[c] = (int)srow[i];       // stable sort: first of the run = smallest input row
    }
    if (i == M - 1) { int C = cidp1[i]; sc->C = C; cstart[C] = M; }
}

// rowcell[r] = first cell index whose row >= r, r = 0..R
__global__ void k_blk_rowtable(BlkParams p, const BlkScalars* __restrict__ sc, const u64* __restrict__ ckey,
                               int* __restrict__ rowcell)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > p.R) return;
    const int C = sc->C;
    u64 target = (u64)(u32)r << p.nyb;
    int lo = 0, hi = C;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (ckey[mid] < target) lo = mid + 1; else hi = mid; }
    rowcell[r] = lo;
}

// neighbour indices (8 per cell, order (dx,dy) = (-1,-1),(-1,0),(-1,1),(0,-1),(0,1),(1,-1),(1,0),(1,1); the
// reverse of direction q is 7-q), 9-cell population test, centroids.  Cells of one row are consecutive in
// the cell table, so each neighbouring row costs ONE binary search (for ny-1) plus a walk over <= 3 cells.
__global__ void k_blk_neighbors(BlkParams p, const BlkScalars* __restrict__ sc, const u64* __restrict__ ckey,
                                const int* __restrict__ rowcell, const int* __restrict__ cstart,
                                const int* __restrict__ sx, const int* __restrict__ sy,
                                int* __restrict__ nb, int* __restrict__ low, double* __restrict__ cx, double* __restrict__ cy)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int C = sc->C;
    if (c >= C) return;
    const u64 k = ckey[c];
    const long long nx = (long long)(k >> p.nyb), ny = (long long)(k & ((1ull << p.nyb) - 1));
    const int cb = cstart[c], ce = cstart[c + 1];
    int tot = ce - cb;
    int res[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) res[q] = -1;
    // same row: the neighbours are c-1 / c+1 when their keys are k-1 / k+1 (ny-1 >= 0 checked: k-1 would borrow)
    if (ny > 0 && c > 0 && ckey[c - 1] == k - 1) res[3] = c - 1;
    if (c + 1 < C && ckey[c + 1] == k + 1 && ny + 1 < (1ll << p.nyb)) res[4] = c + 1;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const long long rx = nx + (side ? 1 : -1);
        if (rx < 0 || rx >= p.R) continue;
        const int rlo = rowcell[rx], rhi = rowcell[rx + 1];
        const long long y0 = ny > 0 ? ny - 1 : 0;
        const u64 target = ((u64)rx << p.nyb) | (u64)y0;
        int lo = rlo, hi = rhi;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (ckey[mid] < target) lo = mid + 1; else hi = mid; }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (lo >= rhi) break;
            const long long yy = (long long)(ckey[lo] & ((1ull << p.nyb) - 1));
            const long long d = yy - ny;
            if (d > 1) break;
            if (d == -1) res[side * 5 + 0] = lo;       // side 0 -> q 0..2, side 1 -> q 5..7
            if (d == 0) res[side * 5 + 1] = lo;
            if (d == 1) res[side * 5 + 2] = lo;
            ++lo;
        }
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) if (res[q] >= 0) tot += cstart[res[q] + 1] - cstart[res[q]];
    int4* o = reinterpret_cast<int4*>(nb + (size_t)c * 8);
    o[0] = make_int4(res[0], res[1], res[2], res[3]);
    o[1] = make_int4(res[4], res[5], res[6], res[7]);
    low[c] = tot < p.minPts ? 1 : 0;
    long long sumx = 0, sumy = 0;
    for (int t = cb; t < ce; ++t) { sumx += sx[t]; sumy += sy[t]; }
    double m = (double)(ce - cb);
    cx[c] = (double)sumx / m;          // true division of Python ints (:136-137)
    cy[c] = (double)sumy / m;
}

// alive[c] = population of the cell if it survives the 9-cell test (:215-228), else 0
__global__ void k_blk_alive(const BlkScalars* __restrict__ sc, const int* __restrict__ nb, const int* __restrict__ low,
                            const int* __restrict__ cstart, int* __restrict__ alive, int* __restrict__ linkbits)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= sc->C) return;
    int a = 1;
    if (low[c]) {
        a = 0;
        for (int q = 0; q < 8; ++q) { int j = nb[(size_t)c * 8 + q]; if (j >= 0 && !low[j]) { a = 1; break; } }
    }
    alive[c] = a ? cstart[c + 1] - cstart[c] : 0;
    linkbits[c] = 0;
}

// link bits: one thread per (cell, forward direction q = 4..7); the link test is symmetric (same centroid
// distance, same point pairs), so the thread sets bit q of its cell and bit 7-q of the neighbour.
// (A work list + 16 lanes per undecided cell pair was measured slower: the list append costs more
// than the pair loops it spreads.)
__global__ void k_blk_links(BlkParams p, const BlkScalars* __restrict__ sc, const int* __restrict__ nb,
                            const int* __restrict__ alive, const int* __restrict__ cstart,
                            const int* __restrict__ sx, const int* __restrict__ sy,
                            const double* __restrict__ cx, const double* __restrict__ cy,
                            int* __restrict__ linkbits)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = t >> 2, q = 4 + (t & 3);
    if (c >= sc->C) return;
    const int j = nb[(size_t)c * 8 + q];
    if (j < 0) return;
    const int na = alive[c], nbj = alive[j];
    if (!na || !nbj) return;
    bool linked = (fabs(cx[c] - cx[j]) + fabs(cy[c] - cy[j])) <= (double)p.eps;        // :232
    // two single-PET cells: the centroids ARE the PETs, the pair test below cannot differ
    if (!linked && (na > 1 || nbj > 1)) {                                              // getGridDist :204-213
        int ab = cstart[c], ae = ab + na, bb = cstart[j], be = bb + nbj;
        if (na > nbj) { int x0 = ab, x1 = ae; ab = bb; ae = be; bb = x0; be = x1; }    // walk the larger cell inside
        for (int s = ab; s < ae && !linked; ++s) {
            const int x = sx[s], y = sy[s];
            int u = bb;
            for (; u + 4 <= be && !linked; u += 4) {
                int d[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    int dx = x - sx[u + k], dy = y - sy[u + k];
                    d[k] = (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy);
                }
                linked = min(min(d[0], d[1]), min(d[2], d[3])) <= p.eps;
            }
            for (; u < be && !linked; ++u) {
                int dx = x - sx[u], dy = y - sy[u];
                linked = (dx < 0 ? -dx : dx) + (dy < 0 ? -dy : dy) <= p.eps;
            }
        }
    }
    if (linked) { atomicOr(&linkbits[c], 1 << q); atomicOr(&linkbits[j], 1 << (7 - q)); }
}

// population over the linked cells + core flag (:236-240)
__global__ void k_blk_core(BlkParams p, const BlkScalars* __restrict__ sc, const int* __restrict__ nb,
                           const int* __restrict__ alive, const int* __restrict__ cstart,
                           const int* __restrict__ linkbits, int* __restrict__ corec)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= sc->C) return;
    int core = 0;
    if (alive[c]) {
        const int bits = linkbits[c];
        int psum = cstart[c + 1] - cstart[c];
        for (int q = 0; q < 8; ++q)
            if (bits & (1 << q)) psum += je_minus(cstart, nb[(size_t)c * 8 + q]);
        core = psum >= p.minPts ? 1 : 0;
    }
    corec[c] = core;
}

__global__ void k_blk_union(const BlkScalars* __restrict__ sc, const int* __restrict__ nb, const int* __restrict__ linkbits,
                            const int* __restrict__ corec, int* parent)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= sc->C) return;
    if (!corec[c]) return;
    int bits = linkbits[c];
    for (int q = 0; q < 8; ++q) {
        if (!(bits & (1 << q))) continue;
        int j = nb[(size_t)c * 8 + q];
        if (j < c && corec[j]) uf_unite(parent, c, j);
    }
}

// root per core cell + component key = smallest cfirst (two-level reduce-by-key like k_flatten: a giant
// component would otherwise serialise millions of atomicMin on one address)
__global__ void __launch_bounds__(BIGTPB)
k_blk_flatten(const BlkScalars* __restrict__ sc, const int* __restrict__ corec, int* parent,
              const int* __restrict__ cfirst, int* __restrict__ root, int* __restrict__ compkey)
{
    __shared__ int hkey[AGG_H], hmin[AGG_H];
    if (threadIdx.x < AGG_H) { hkey[threadIdx.x] = -1; hmin[threadIdx.x] = INT_MAX; }
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    int r = -1, key = INT_MAX;
    if (c < sc->C) {
        if (corec[c]) { r = uf_find(parent, c); key = cfirst[c]; }
        root[c] = r;
    }
    const int lane = threadIdx.x & 63;
    unsigned long long pending = __ballot(r >= 0);
    while (pending) {
        const int leader = __ffsll((long long)pending) - 1;
        const int R = __builtin_amdgcn_readlane(r, leader);
        const unsigned long long m = __ballot(r == R);
        const bool mine = r == R;
        if (__popcll(m) >= 4) {
            int mk = mine ? key : INT_MAX;
            mk = dpp_reduce_wave(mk, OpMin());
            if (lane == leader) {
                const int sl = agg_slot(hkey, R);
                if (sl >= 0) atomicMin(&hmin[sl], mk); else atomicMin(&compkey[R], mk);
            }
        } else if (mine) {
            atomicMin(&compkey[R], key);
        }
        pending &= ~m;
    }
    __syncthreads();
    if (threadIdx.x < AGG_H && hkey[threadIdx.x] >= 0) atomicMin(&compkey[hkey[threadIdx.x]], hmin[threadIdx.x]);
}

__global__ void k_blk_rank_flags(const BlkScalars* __restrict__ sc, const int* __restrict__ root,
                                 const int* __restrict__ compkey, int* __restrict__ flag)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= sc->C) return;
    if (root[c] == c) flag[compkey[c]] = 1;
}

__global__ void k_blk_cell_labels(const BlkScalars* __restrict__ sc, const int* __restrict__ nb,
                                  const int* __restrict__ linkbits, const int* __restrict__ alive,
                                  const int* __restrict__ root, const int* __restrict__ compkey,
                                  const int* __restrict__ rankscan, int* __restrict__ clab)
{
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= sc->C) return;
    int lab = -1;
    if (root[c] >= 0) lab = rankscan[compkey[root[c]]];
    else if (alive[c]) {
        int bits = linkbits[c];
        for (int q = 0; q < 8; ++q) {
            if (!(bits & (1 << q))) continue;
            int j = nb[(size_t)c * 8 + q];
            if (root[j] >= 0) lab = max(lab, rankscan[compkey[root[j]]]);     // :195-198 last writer = highest rank
        }
    }
    clab[c] = lab;
}

__global__ void __launch_bounds__(BIGTPB)
k_blk_point_labels(BlkParams p, const BlkScalars* __restrict__ sc, const int* __restrict__ cidp1,
                   const int* __restrict__ clab, const u32* __restrict__ srow, const int* __restrict__ sx,
                   const int* __restrict__ sy, int* __restrict__ labels, Table t)
{
    __shared__ TableLds h;
    table_lds_init(h);
    const int M = sc->M;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int lab = -1, x = 0, y = 0;
    if (i < M) {
        lab = clab[cidp1[i] - 1];
        labels[srow[i]] = lab;
        x = sx[i]; y = sy[i];
    }
    table_accumulate(t, h, lab, x, y);
    table_flush(t, h);
}

// ==========================================================================================
// K7: distance statistics of one step (the inputs of cLoops/ests.py:36-61, estIntSelCutFrag)
// ==========================================================================================
// pipe.py:106-109 collects `dis` = Y-X of the PETs in inter-ligation clusters and `dss` = Y-X of
// the PETs in self-ligation clusters plus the PETs dropped by the cut (pipe.py:63); ests.py then
// needs counts, mean / std of log2(|d|) over d > 0 for both groups and the median of the self
// group.  At tens of millions of PETs per step the host-side masks, log2 and np.median cost
// ~20x the clustering itself, so the sums are reduced here (fixed order: deterministic) and the
// median comes from an exact 4-pass radix select on the integer distances.
// group 0 = inter, group 1 = self (+ short), -1 = in no group
__global__ void k7_classify(const int* __restrict__ hdr, Table t, signed char* __restrict__ cls)
{
    const int K = hdr[0];
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const cl_box b = t.get(k);
    signed char c = -1;
    if (b.count > 0 && b.min_x != b.max_x && b.min_y != b.max_y)           // pipe.py:83-85
        c = (b.max_x < b.min_y) ? 0 : 1;                                    // pipe.py:97
    cls[k] = c;
}

// Source of the per-PET (distance, label) pairs of the last completed run:
//   sorted  the run's sorted arrays: d = q + V0 and the label of sorted position i (rotated variants; the PETs
//           removed by the cut are not in them and come from the input rows: d = Y - X < cut)
//   rows    input-row order (variants that only produce row-order labels)
struct K7Src {
    int sorted; int n; int M; int v0;
    const int* dM;                                      // if set: M is read from the device (the run has not been waited for yet)
    const int* X; const int* Y; const int* labels;      // input rows (+ row-order labels)
    const int* sv; const int* slab;                     // sorted q and sorted-order labels of [0, M)
    const int* dh;                                      // if set: dh[d] = number of input rows with Y - X == d for 0 <= d < cut, and no row has
                                                        // Y - X < 0: the PETs dropped by the cut come from it, not from a pass over the rows
};
#define K7_BLOCKS 2048           // workgroups (= fixed partial sums) of the K7 reductions: 8 waves per SIMD (512 left the loads
                                 // of a 66-element sequential loop per thread uncovered: 177 us -> see DESIGN.md)
#define K7_LOGBINS 3840          // 30 octaves x 128: bin = floor(log2 d) * 128 + the next 7 bits of d (monotone in d)
#define K7_FINE 2048
#define K7_XSHIFT 11.0           // sums are taken over x = log2|d| - K7_XSHIFT (less cancellation in sum x^2 - (sum x)^2 / n)

__device__ __forceinline__ int k7_logbin(unsigned d)      // d >= 1
{
    const int e = 31 - __clz((int)d);
    const unsigned m = e >= 7 ? ((d >> (e - 7)) & 127u) : ((d << (7 - e)) & 127u);
    return e * 128 + (int)m;
}
// f(group, |d|) for every PET of a group, in a fixed order per thread (deterministic partial sums): the block works
// on fixed contiguous ranges of the sources
template <typename F>
__device__ __forceinline__ void k7_for_each(const K7Src& s, int cut, const signed char* __restrict__ cls, F&& f)
{
    if (s.sorted) {
        const int M = s.dM ? s.dM[0] : s.M;
        const int per = (M + gridDim.x - 1) / gridDim.x;
        const int i0 = blockIdx.x * per, i1 = min(M, i0 + per);
        // four PETs per round: the label loads, then the class gathers, of all four are in flight together (the walk is bound
        // by the dependent label -> class round trips, not by bytes); f() still sees the PETs in ascending order
        int i = i0 + threadIdx.x;
        const int bd = blockDim.x;
        for (; i + 3 * bd < i1; i += 4 * bd) {
            int lab[4], d[4], g[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { lab[k] = s.slab[i + k * bd]; d[k] = s.sv[i + k * bd] + s.v0; }
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = lab[k] >= 0 ? (int)cls[lab[k]] : -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) if (g[k] >= 0) f(g[k], d[k] < 0 ? -d[k] : d[k], 1);     // ests.py:42-43 np.abs
        }
        for (; i < i1; i += bd) {
            const int lab = s.slab[i];
            const int g = lab >= 0 ? (int)cls[lab] : -1;
            const int d = s.sv[i] + s.v0;
            if (g >= 0) f(g, d < 0 ? -d : d, 1);
        }
        if (cut > 0 && s.dh) {                                     // pipe.py:63: short PETs go to dss -- all PETs of one distance at once
            for (int d = blockIdx.x * blockDim.x + threadIdx.x; d < cut; d += gridDim.x * blockDim.x) {
                const int w = s.dh[d];
                if (w) f(1, d, w);
            }
        } else if (cut > 0) {
            const int perr = (s.n + gridDim.x - 1) / gridDim.x;
            const int r0 = blockIdx.x * perr, r1 = min(s.n, r0 + perr);
            for (int r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
                const int d = s.Y[r] - s.X[r];
                if (d < cut) f(1, d < 0 ? -d : d, 1);
            }
        }
    } else {
        const int perr = (s.n + gridDim.x - 1) / gridDim.x;
        const int r0 = blockIdx.x * perr, r1 = min(s.n, r0 + perr);
        for (int r = r0 + threadIdx.x; r < r1; r += blockDim.x) {
            const int d = s.Y[r] - s.X[r];
            int g = 1;
            if (!(cut > 0 && d < cut)) { const int lab = s.labels[r]; g = lab >= 0 ? (int)cls[lab] : -1; }
            if (g >= 0) f(g, d < 0 ? -d : d, 1);
        }
    }
}

struct K7Part { double sx[2]; double sxx[2]; long long n_all[2]; long long n_pos[2]; };

// fixed-order reduction of the workgroup partials (deterministic: thread t sums blocks t, t+256, ... in order, then a fixed
// tree) -- the host reads 64 bytes instead of K7_BLOCKS partials; in a sweep step the candidate totals ride along.  Called by
// the first 256 threads of ONE workgroup; sd / sn: 4 x 256 doubles / long longs of LDS.
__device__ __forceinline__ void k7_reduce_block(const K7Part* __restrict__ parts, int nparts, K7Part* __restrict__ out,
                                                const int* __restrict__ bcount /* or null */, int nb, long long* __restrict__ totals,
                                                double (*sd)[256], long long (*sn)[256])
{
    const int tid = threadIdx.x;
    double a[4] = {0, 0, 0, 0}; long long c[6] = {0, 0, 0, 0, 0, 0};
    for (int k = tid; k < nparts; k += 256) {
        const K7Part p = parts[k];
        a[0] += p.sx[0]; a[1] += p.sx[1]; a[2] += p.sxx[0]; a[3] += p.sxx[1];
        c[0] += p.n_all[0]; c[1] += p.n_all[1]; c[2] += p.n_pos[0]; c[3] += p.n_pos[1];
    }
    if (bcount) for (int k = tid; k < nb; k += 256) { c[4] += bcount[k]; c[5] += bcount[nb + k]; }      // inter / self boxes of the run
    for (int q = 0; q < 4; ++q) { sd[q][tid] = a[q]; sn[q][tid] = c[q]; }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) for (int q = 0; q < 4; ++q) { sd[q][tid] += sd[q][tid + o]; sn[q][tid] += sn[q][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        K7Part p;
        p.sx[0] = sd[0][0]; p.sx[1] = sd[1][0]; p.sxx[0] = sd[2][0]; p.sxx[1] = sd[3][0];
        p.n_all[0] = sn[0][0]; p.n_all[1] = sn[1][0]; p.n_pos[0] = sn[2][0]; p.n_pos[1] = sn[3][0];
        *out = p;
    }
    if (bcount) {
        __syncthreads();
        sn[0][tid] = c[4]; sn[1][tid] = c[5];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if (tid < o) { sn[0][tid] += sn[0][tid + o]; sn[1][tid] += sn[1][tid + o]; }
            __syncthreads();
        }
        if (tid == 0) { totals[0] = sn[0][0]; totals[1] = sn[1][0]; }
    }
}

// (a "last workgroup reduces" step inside k7_summary instead of this launch measured 4 % SLOWER on the whole sweep)
// host_step / host_hdr (sweep steps): the step's whole output -- totals, statistics, the two histograms -- and the run's header go
// to pinned host memory from HERE, the last kernel of the step (the host reads them after the stream's completion event): no
// copy-stream hand-over and no copy packets per run
__global__ void __launch_bounds__(256)
k7_reduce_parts(const K7Part* __restrict__ parts, int nparts, K7Part* out, const int* __restrict__ bcount /* or null */, int nb,
                long long* totals, const volatile unsigned long long* dev_step /* or null */, int step_words,
                unsigned long long* __restrict__ host_step, const int* __restrict__ dev_hdr, int* __restrict__ host_hdr)
{
    // `out`, `totals` and `dev_step` are views of ONE device buffer (the step output: totals | reduced part | histograms):
    // no restrict on them, and the copy to the host re-reads what thread 0 has just stored (volatile loads behind the fence)
    __shared__ double sd[4][256];
    __shared__ long long sn[4][256];
    k7_reduce_block(parts, nparts, out, bcount, nb, totals, sd, sn);
    if (dev_step) {
        __threadfence();
        __syncthreads();                                // totals and the reduced part are written (same workgroup: visible)
        for (int k = threadIdx.x; k < step_words; k += 256) host_step[k] = dev_step[k];
        if (threadIdx.x < 8) host_hdr[threadIdx.x] = dev_hdr[threadIdx.x];
    }
}

// one pass: counts, sum x and sum x^2 (x = log2|d| - K7_XSHIFT over d != 0) for both groups, and the log-binned
// histogram of the self group's |d| (first level of the exact median)
__global__ void __launch_bounds__(TPB)
k7_summary(K7Src s, int cut, const signed char* __restrict__ cls, K7Part* __restrict__ parts, unsigned long long* __restrict__ loghist,
           unsigned fine_lo, unsigned long long* __restrict__ fine /* or null: exact histogram of the self group's fine_lo <= |d| < fine_lo + 2048 */)
{
    __shared__ unsigned int h[K7_LOGBINS];
    __shared__ unsigned int hf[K7_FINE];
    for (int k = threadIdx.x; k < K7_LOGBINS; k += blockDim.x) h[k] = 0u;
    for (int k = threadIdx.x; k < K7_FINE; k += blockDim.x) hf[k] = 0u;
    __syncthreads();
    double sx[2] = {0.0, 0.0}, sxx[2] = {0.0, 0.0};
    long long na[2] = {0, 0}, np_[2] = {0, 0};
    const bool want_fine = fine != nullptr;
    k7_for_each(s, cut, cls, [&](int g, int ad, int w) {                  // w PETs of this group and distance
        na[g] += w;
        if (ad > 0) {
            const double x = log2((double)ad) - K7_XSHIFT, wx = (double)w * x;
            np_[g] += w; sx[g] += wx; sxx[g] += wx * x;
            if (g == 1) {
                atomicAdd(&h[k7_logbin((unsigned)ad)], (unsigned)w);
                const unsigned off = (unsigned)ad - fine_lo;                  // wraps for ad < fine_lo: out of range
                if (want_fine && off < (unsigned)K7_FINE) atomicAdd(&hf[off], (unsigned)w);
            }
        }
    });
    __shared__ double s_d[4][TPB / 64];
    __shared__ long long s_n[4][TPB / 64];
    for (int g = 0; g < 2; ++g)
        for (int o = 32; o > 0; o >>= 1) {
            sx[g] += __shfl_down(sx[g], o); sxx[g] += __shfl_down(sxx[g], o);
            na[g] += __shfl_down(na[g], o); np_[g] += __shfl_down(np_[g], o);
        }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) {
        s_d[0][wv] = sx[0]; s_d[1][wv] = sx[1]; s_d[2][wv] = sxx[0]; s_d[3][wv] = sxx[1];
        s_n[0][wv] = na[0]; s_n[1][wv] = na[1]; s_n[2][wv] = np_[0]; s_n[3][wv] = np_[1];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        K7Part p;
        for (int g = 0; g < 2; ++g) {
            double a = 0, b = 0; long long c = 0, d = 0;
            for (int w = 0; w < TPB / 64; ++w) { a += s_d[g][w]; b += s_d[2 + g][w]; c += s_n[g][w]; d += s_n[2 + g][w]; }
            p.sx[g] = a; p.sxx[g] = b; p.n_all[g] = c; p.n_pos[g] = d;
        }
        parts[blockIdx.x] = p;
    }
    for (int k = threadIdx.x; k < K7_LOGBINS; k += blockDim.x)
        if (h[k]) atomicAdd(&loghist[k], (unsigned long long)h[k]);
    if (want_fine)
        for (int k = threadIdx.x; k < K7_FINE; k += blockDim.x)
            if (hf[k]) atomicAdd(&fine[k], (unsigned long long)hf[k]);
}

// refinement pass of the exact median: histogram of (|d| - lo) >> shift over the self group's lo <= |d| < hi
__global__ void __launch_bounds__(TPB)
k7_bin_hist(K7Src s, int cut, const signed char* __restrict__ cls, unsigned lo, unsigned hi, int shift, unsigned long long* __restrict__ hist)
{
    __shared__ unsigned int h[K7_FINE];
    for (int k = threadIdx.x; k < K7_FINE; k += blockDim.x) h[k] = 0u;
    __syncthreads();
    k7_for_each(s, cut, cls, [&](int g, int ad, int w) {
        const unsigned u = (unsigned)ad;
        if (g == 1 && u >= lo && u < hi) atomicAdd(&h[min((u - lo) >> shift, (unsigned)(K7_FINE - 1))], (unsigned)w);
    });
    __syncthreads();
    for (int k = threadIdx.x; k < K7_FINE; k += blockDim.x)
        if (h[k]) atomicAdd(&hist[k], (unsigned long long)h[k]);
}

// ==========================================================================================
// K10: the candidate loops of a sweep, kept on the device
// ==========================================================================================
// The sweep driver used to pull every run's cluster table over PCIe, classify it with numpy and, at the end, dedup
// the concatenation of all steps on the host (combineTwice, cLoops/pipe.py:155-174: a box is kept in the step where
// it FIRST appears; duplicates inside one step all stay) and filter it by the final cut (filterClusterByDis,
// pipe.py:130-143, Python-2 floor mid-points).  Here a run's inter-ligation boxes (pipe.py:83-97) are appended, in
// ascending cluster id, to a per-chromosome device buffer together with their step number; at the end of the sweep
// one 64-bit-hash radix sort groups equal boxes (stable: the first of a group is its first appearance), the exact
// boxes are compared inside a group, and the survivors are compacted in append order -- the order the reference's
// record lists have.  Only the final table crosses PCIe.
#define CAND_BLOCK 2048
__global__ void __launch_bounds__(256)
k_cand_count(const int* __restrict__ dK, const signed char* __restrict__ cls, int* __restrict__ bcount /* [nb] inter, [nb] self */, int nb)
{
    __shared__ int red[2][4];
    const int K = dK[0];
    const int base = blockIdx.x * CAND_BLOCK;
    int ci = 0, cs = 0;
    for (int k = threadIdx.x; k < CAND_BLOCK; k += 256) {
        const int i = base + k;
        const int c = i < K ? (int)cls[i] : -1;
        ci += c == 0; cs += c == 1;
    }
    for (int o = 32; o > 0; o >>= 1) { ci += __shfl_down(ci, o); cs += __shfl_down(cs, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ci; red[1][threadIdx.x >> 6] = cs; }
    __syncthreads();
    if (threadIdx.x == 0) { bcount[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3]; bcount[nb + blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3]; }
}
// sweep step: k7_classify + k_cand_count in one launch, which also clears the step's histograms (`zero`, nzero 8-byte words)
__global__ void __launch_bounds__(256)
k_step_classify_count(const int* __restrict__ dK, Table t, signed char* __restrict__ cls, int* __restrict__ bcount, int nb,
                      unsigned long long* __restrict__ zero, int nzero)
{
    __shared__ int red[2][4];
    for (int k = blockIdx.x * 256 + threadIdx.x; k < nzero; k += gridDim.x * 256) zero[k] = 0ull;
    const int K = dK[0];
    const int base = blockIdx.x * CAND_BLOCK;
    int ci = 0, cs = 0;
    for (int k = threadIdx.x; k < CAND_BLOCK; k += 256) {
        const int i = base + k;
        if (i < K) {
            const cl_box b = t.get(i);
            signed char c = -1;
            if (b.count > 0 && b.min_x != b.max_x && b.min_y != b.max_y)       // pipe.py:83-85
                c = (b.max_x < b.min_y) ? 0 : 1;                                // pipe.py:97
            cls[i] = c;
            ci += c == 0; cs += c == 1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { ci += __shfl_down(ci, o); cs += __shfl_down(cs, o); }
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = ci; red[1][threadIdx.x >> 6] = cs; }
    __syncthreads();
    if (threadIdx.x == 0) { bcount[blockIdx.x] = red[0][0] + red[0][1] + red[0][2] + red[0][3]; bcount[nb + blockIdx.x] = red[1][0] + red[1][1] + red[1][2] + red[1][3]; }
}
// ordered scatter of the flagged elements of [0, N): dst = base + boff[block] + rank inside the block (element order)
template <typename F, typename W>
__device__ __forceinline__ void ordered_scatter_block(int N, const int* __restrict__ boff /* or null: */, const int* __restrict__ bcount,
                                                      F&& flagged, W&& write)
{
    __shared__ int l_cnt[(CAND_BLOCK / 256) * 4];
    const int base = blockIdx.x * CAND_BLOCK;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bool keep[CAND_BLOCK / 256]; int before[CAND_BLOCK / 256];
#pragma unroll
    for (int k = 0; k < CAND_BLOCK / 256; ++k) {
        const int i = base + k * 256 + (int)threadIdx.x;
        keep[k] = i < N && flagged(i);
        const unsigned long long bal = __ballot(keep[k]);
        before[k] = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
        if (lane == 0) l_cnt[k * 4 + wv] = __popcll(bal);
    }
    __syncthreads();
    int pre;
    if (boff) pre = boff[blockIdx.x];
    else {
        // no scan over the block counts: a block sums the counts in front of it itself (a few hundred at most: sweep steps)
        __shared__ int l_pre[4];
        int sum = 0;
        for (int k = threadIdx.x; k < (int)blockIdx.x; k += 256) sum += bcount[k];
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o);
        if (lane == 0) l_pre[wv] = sum;
        __syncthreads();
        pre = l_pre[0] + l_pre[1] + l_pre[2] + l_pre[3];
    }
#pragma unroll
    for (int k = 0; k < CAND_BLOCK / 256; ++k) {
        int mine = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w == wv) mine = pre; pre += l_cnt[k * 4 + w]; }
        if (keep[k]) write(base + k * 256 + (int)threadIdx.x, mine + before[k]);
    }
}
__global__ void __launch_bounds__(256)
k_cand_append(const int* __restrict__ dK, const signed char* __restrict__ cls, Table t, const int* __restrict__ boff /* or null: */,
              const int* __restrict__ bcount, int base, int step, int cap, int4* __restrict__ cbox, int* __restrict__ cstep)
{
    ordered_scatter_block(dK[0], boff, bcount, [&](int i) { return cls[i] == 0; },
                          [&](int i, int r) { const int d = base + r; if (d < cap) { cbox[d] = make_int4(t.minx[i], t.maxx[i], t.miny[i], t.maxy[i]); cstep[d] = step; } });
}
__device__ __forceinline__ u64 box_hash(int4 b, u64 salt)
{
    u64 h = salt ^ ((u64)(u32)b.x * 0x9E3779B97F4A7C15ull) ^ ((u64)(u32)b.y * 0xC2B2AE3D27D4EB4Full) ^ ((u64)(u32)b.z * 0x165667B19E3779F9ull) ^ ((u64)(u32)b.w * 0xD6E8FEB86659FD93ull);
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return h;
}
__global__ void k_cand_hash(int N, const int4* __restrict__ cbox, u64 salt, u64* __restrict__ keys, u32* __restrict__ vals)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) { keys[i] = box_hash(cbox[i], salt); vals[i] = (u32)i; }
}
__device__ __forceinline__ long long floordiv2(long long a) { return a >> 1; }      // floor(a / 2) for any sign (pipe.py:138 on Python-2 ints)
__global__ void k_cand_mark(int N, const u64* __restrict__ skeys, const u32* __restrict__ svals, const int4* __restrict__ cbox,
                            const int* __restrict__ cstep, int final_cut, unsigned char* __restrict__ keep, int* __restrict__ flags)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const u64 key = skeys[j];
    int j0 = j, guard = 0;
    while (j0 > 0 && skeys[j0 - 1] == key && guard < 65536) { --j0; ++guard; }
    if (guard >= 65536) atomicExch(&flags[0], 2);
    const u32 p = svals[j], hp = svals[j0];              // stable sort: the head of a group is its first appearance
    const int4 b = cbox[p], hb = cbox[hp];
    const bool same = b.x == hb.x && b.y == hb.y && b.z == hb.z && b.w == hb.w;
    if (!same) atomicExch(&flags[0], 1);                 // two different boxes share a 64-bit hash: the caller redoes this chromosome exactly
    const long long d = floordiv2((long long)b.z + b.w) - floordiv2((long long)b.x + b.y);
    keep[p] = (same && cstep[p] == cstep[hp] && d >= (long long)final_cut) ? 1 : 0;
}
__global__ void __launch_bounds__(256)
k_flag_count(int N, const unsigned char* __restrict__ keep, int* __restrict__ bcount)
{
    __shared__ int red[4];
    const int base = blockIdx.x * CAND_BLOCK;
    int c = 0;
    for (int k = threadIdx.x; k < CAND_BLOCK; k += 256) { const int i = base + k; c += (i < N && keep[i]) ? 1 : 0; }
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) bcount[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void __launch_bounds__(256)
k_cand_emit(int N, const unsigned char* __restrict__ keep, const int4* __restrict__ cbox, const int* __restrict__ boff, int4* __restrict__ out)
{
    ordered_scatter_block(N, boff, (const int*)nullptr, [&](int i) { return keep[i] != 0; }, [&](int i, int r) { out[r] = cbox[i]; });
}

// ==========================================================================================
// K8: interval counting for the significance test (cLoops/cModel.py:60-80, 108-143)
// ==========================================================================================
// For a candidate loop with anchors iva, ivb the reference builds Python sets of the PETs that have
// an end inside a window, S(W) = {i : X_i in W} | {i : Y_i in W}, for the two anchors and for 10 + 10
// shifted windows, and needs |S(A_k)|, |S(B_l)|, |S(A_k) & S(B_l)| and rab = |{X in iva} & {Y in ivb}|.
// One workgroup per candidate: the PETs with an end inside the span of the A windows (resp. B windows)
// are two contiguous slices of the X-sorted and Y-sorted PET tables; every PET gets an 11-bit
// membership mask per side and bumps the counters in LDS.  Pure integer work; the p-values stay on
// the host (scipy), fed with exactly the reference's counts.
#define SIG_W 11                       // window 0 = the anchor itself, 1..10 = cModel.getNearbyPairRegions
#define SIG_OUT (2 * SIG_W + 1 + SIG_W * SIG_W)

__global__ void k8_split(const int* __restrict__ X, const int* __restrict__ Y, int n, int cut,
                         u64* __restrict__ kx, u64* __restrict__ ky)
{
    int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const int x = X[r], y = Y[r];
    const bool valid = cut <= 0 || (y - x) >= cut;            // parseJd(f, cut), io.py:213-216
    // sort key = coordinate (biased to be non-negative), payload = the other coordinate
    kx[r] = valid ? (((u64)(u32)(x + (1 << 30)) << 32) | (u32)(y + (1 << 30))) : ~0ull;
    ky[r] = valid ? (((u64)(u32)(y + (1 << 30)) << 32) | (u32)(x + (1 << 30))) : ~0ull;
}

// first index with (key >> 32) >= v   /   > v   in a sorted u64 table of m valid entries
__device__ __forceinline__ int k8_lb(const u64* __restrict__ t, int m, long long v)
{
    const u64 target = v <= -(1ll << 30) ? 0ull : ((u64)(u32)(v + (1 << 30)) << 32);
    int lo = 0, hi = m;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (t[mid] < target) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ __forceinline__ int k8_ub(const u64* __restrict__ t, int m, long long v)
{
    return k8_lb(t, m, v + 1);
}

struct SigWin { int lo[2 * SIG_W]; int hi[2 * SIG_W]; };      // [0..10] = A windows, [11..21] = B windows

__global__ void __launch_bounds__(TPB)
k8_counts(const u64* __restrict__ tx, const u64* __restrict__ ty, const int* __restrict__ d_m, int nrec,
          const SigWin* __restrict__ wins, int* __restrict__ out)
{
    __shared__ int wlo[2 * SIG_W], whi[2 * SIG_W];
    __shared__ int c_a[SIG_W], c_b[SIG_W], c_ab[SIG_W * SIG_W], c_rab;
    __shared__ int rng[8];
    const int rec = blockIdx.x;
    if (rec >= nrec) return;
    const int m = d_m[0];
    if (threadIdx.x < 2 * SIG_W) { wlo[threadIdx.x] = wins[rec].lo[threadIdx.x]; whi[threadIdx.x] = wins[rec].hi[threadIdx.x]; }
    if (threadIdx.x < SIG_W) { c_a[threadIdx.x] = 0; c_b[threadIdx.x] = 0; }
    for (int k = threadIdx.x; k < SIG_W * SIG_W; k += blockDim.x) c_ab[k] = 0;
    if (threadIdx.x == 0) c_rab = 0;
    __syncthreads();
    if (threadIdx.x < 4) {
        // spans of the A and of the B windows; slices of the X-sorted (0,2) and Y-sorted (1,3) tables
        const int side = threadIdx.x >> 1, off = side * SIG_W;
        int lo = wlo[off], hi = whi[off];
        for (int k = 1; k < SIG_W; ++k) { lo = min(lo, wlo[off + k]); hi = max(hi, whi[off + k]); }
        const u64* t = (threadIdx.x & 1) ? ty : tx;
        rng[threadIdx.x * 2] = k8_lb(t, m, lo);
        rng[threadIdx.x * 2 + 1] = k8_ub(t, m, hi);
        if ((threadIdx.x & 1) == 0) { /* keep spans for the dedupe test */ }
    }
    __syncthreads();
    int spanlo[2], spanhi[2];
    for (int side = 0; side < 2; ++side) {
        int lo = wlo[side * SIG_W], hi = whi[side * SIG_W];
        for (int k = 1; k < SIG_W; ++k) { lo = min(lo, wlo[side * SIG_W + k]); hi = max(hi, whi[side * SIG_W + k]); }
        spanlo[side] = lo; spanhi[side] = hi;
    }
    for (int side = 0; side < 2; ++side) {
        for (int tab = 0; tab < 2; ++tab) {
            const u64* t = tab ? ty : tx;
            const int b = rng[(side * 2 + tab) * 2], e = rng[(side * 2 + tab) * 2 + 1];
            for (int j = b + (int)threadIdx.x; j < e; j += blockDim.x) {
                const u64 kv = t[j];
                const int first = (int)(u32)(kv >> 32) - (1 << 30), second = (int)(u32)(kv & 0xffffffffu) - (1 << 30);
                const int x = tab ? second : first, y = tab ? first : second;
                // a PET with both ends inside the span is in both slices: count it from the X table only
                if (tab == 1 && x >= spanlo[side] && x <= spanhi[side]) continue;
                unsigned ma = 0, mb = 0;
#pragma unroll
                for (int k = 0; k < SIG_W; ++k) {
                    ma |= (unsigned)(((x >= wlo[k]) & (x <= whi[k])) | ((y >= wlo[k]) & (y <= whi[k]))) << k;
                    mb |= (unsigned)(((x >= wlo[SIG_W + k]) & (x <= whi[SIG_W + k])) | ((y >= wlo[SIG_W + k]) & (y <= whi[SIG_W + k]))) << k;
                }
                if (side == 0) {
                    for (unsigned a = ma; a; a &= a - 1) {
                        const int k = __ffs(a) - 1;
                        atomicAdd(&c_a[k], 1);
                        for (unsigned bb = mb; bb; bb &= bb - 1) atomicAdd(&c_ab[k * SIG_W + (__ffs(bb) - 1)], 1);
                    }
                    // rab = |{X in iva} & {Y in ivb}|  (cModel.py:79): needs x in A_0, found in the X table
                    if (tab == 0 && x >= wlo[0] && x <= whi[0] && y >= wlo[SIG_W] && y <= whi[SIG_W]) atomicAdd(&c_rab, 1);
                } else {
                    for (unsigned bb = mb; bb; bb &= bb - 1) atomicAdd(&c_b[__ffs(bb) - 1], 1);
                }
            }
        }
    }
    __syncthreads();
    int* o = out + (size_t)rec * SIG_OUT;
    if (threadIdx.x < SIG_W) { o[threadIdx.x] = c_a[threadIdx.x]; o[SIG_W + threadIdx.x] = c_b[threadIdx.x]; }
    if (threadIdx.x == 0) o[2 * SIG_W] = c_rab;
    for (int k = threadIdx.x; k < SIG_W * SIG_W; k += blockDim.x) o[2 * SIG_W + 1 + k] = c_ab[k];
}

__global__ void k8_count_valid(const u64* __restrict__ t, int n, int* __restrict__ d_m)
{
    // number of valid (non-sentinel) entries of the sorted table = lower bound of the sentinel
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int lo = 0, hi = n;
        while (lo < hi) { int mid = (lo + hi) >> 1; if (t[mid] != ~0ull) lo = mid + 1; else hi = mid; }
        d_m[0] = lo;
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    bool fresh = false;               // (re)allocated since the flag was last cleared
    int ensure(size_t need)
    {
        if (need <= bytes) return CL_OK;
        fresh = true;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        size_t want = need + need / 8 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "hipMalloc", hipGetErrorString(e));
        bytes = want;
        return CL_OK;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    template <typename T> T* as() { return (T*)p; }
};

struct cl_chrom {
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
    int sort_index_mode = 0;          // cl_set_sort_index: 0 = build at the second sort, 1 = at the first, -1 = never
    long long n_sorts = 0;            // layouts sorted on this handle so far
    DevBuf sv, sa, strip, cnt, parent, root, head, headidx, cellfirst, compkey, ncore, bsize, owner, state;
    DevBuf tileflag;                  // per 256-PET tile: holds a contested border point (k_border -> k_emit_records)
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
        hipEvent_t ev[8]{};           // profiling marks of the run that used this slot
        int* h_hdr = nullptr;         // pinned: {K, overflow, M}
        cl_box* h_boxes = nullptr;    // pinned host copy of the cluster table
        size_t h_boxes_cap = 0;
        int32_t* labels_out = nullptr;
        DevBuf slab;                  // labels in sorted order (rotated variants)
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
        const int* k7_sv = nullptr;   // sorted q of the run
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

// The candidate buffer of a sweep (K10) holds every inter-ligation box of every step: it grows on demand (contents kept) --
// before a step is enqueued it has room for all the boxes the step can produce (one per cluster id, at most n / minPts).
static int ensure_cand_capacity(cl_chrom* c, long long need)
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
                      &c->ncore, &c->bsize, &c->owner, &c->state, &c->flag, &c->rankscan, &c->slot[0].labels, &c->slot[0].table, &c->slot[1].labels, &c->slot[1].table, &c->slot[0].slab, &c->slot[1].slab, &c->slot[0].d_step, &c->slot[1].d_step, &c->hdr, &c->k7_cls, &c->k7_parts, &c->sig_tx, &c->sig_ty, &c->sig_tmp, &c->sig_sorttmp, &c->sig_m, &c->sig_win, &c->sig_out,
                      &c->ulist, &c->lo, &c->hi, &c->recs, &c->counters, &c->tileflag, &c->chainflag, &c->chainhead, &c->usize, &c->b_cstart, &c->b_ckey, &c->b_nb, &c->b_cx, &c->b_cy, &c->tile_s0, &c->bq, &c->bsp, &c->brow, &c->bstrip, &c->btile, &c->sel_tmp, &c->cand_box, &c->cand_step, &c->cand_keep, &c->cand_out, &c->dhist};
    for (DevBuf* b : bufs) b->release();
    if (c->own_xy) { if (c->d_x) (void)hipFree(c->d_x); if (c->d_y) (void)hipFree(c->d_y); }
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    for (auto& sl : c->slot) {
        if (sl.h_boxes) (void)hipHostFree(sl.h_boxes);
        if (sl.h_hdr) (void)hipHostFree(sl.h_hdr);
        if (sl.h_step) (void)hipHostFree(sl.h_step);
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
extern "C" void cl_set_layout_reuse(cl_chrom* c, int enabled) { if (c) { c->reuse_layout = enabled != 0; c->base.valid = false; } }
extern "C" void cl_set_sort_index(cl_chrom* c, int mode) { if (c) c->sort_index_mode = mode > 0 ? 1 : (mode < 0 ? -1 : 0); }
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
        if (hipHostMalloc((void**)&c->h_pinned, 4096, hipHostMallocDefault) != hipSuccess) { rc = fail(CL_ERR_HIP, "hipHostMalloc"); break; }
        {
            // copies must not queue up behind the next run's kernels: give their streams the highest priority
            int prio_lo = 0, prio_hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
            if (hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, prio_hi) != hipSuccess ||
                hipStreamCreateWithPriority(&c->aux_stream, hipStreamNonBlocking, prio_hi) != hipSuccess) { rc = fail(CL_ERR_HIP, "hipStreamCreate(copy)"); break; }
        }
        bool okslots = true;
        for (auto& sl : c->slot) {
            okslots = okslots && hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming) == hipSuccess;
            okslots = okslots && hipEventCreateWithFlags(&sl.ev_copied, hipEventDisableTiming) == hipSuccess;
            okslots = okslots && hipHostMalloc((void**)&sl.h_hdr, 64, hipHostMallocDefault) == hipSuccess;
        }
        if (!okslots) { rc = fail(CL_ERR_HIP, "result slot setup"); break; }
        if (on_device) { c->d_x = (int*)x; c->d_y = (int*)y; }
        else if (n > 0) {
            c->own_xy = true;
            if (hipMalloc((void**)&c->d_x, n * 4) != hipSuccess || hipMalloc((void**)&c->d_y, n * 4) != hipSuccess) { rc = fail(CL_ERR_HIP, "hipMalloc X/Y"); break; }
            if (hipMemcpyAsync(c->d_x, x, n * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess ||
                hipMemcpyAsync(c->d_y, y, n * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) { rc = fail(CL_ERR_HIP, "hipMemcpy X/Y"); break; }
        }
        if (n > 0) {
            if ((rc = c->counters.ensure(256))) break;
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
    *out = c;
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
__global__ void k_cut_strips(int S, int thr, const int* __restrict__ bstrip, const int* __restrict__ bq,
                             int* __restrict__ kept /* [S+1] */, int* __restrict__ src0 /* [S] first kept source index */)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s > S) return;
    if (s == S) { kept[S] = 0; return; }
    int lo = bstrip[s];
    const int e = bstrip[s + 1];
    int hi = e;
    while (lo < hi) { const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1); if (bq[mid] < thr) lo = mid + 1; else hi = mid; }
    kept[s] = e - lo;
    src0[s] = lo;
}
__global__ void __launch_bounds__(CMP_TPB)
k_cut_copy(int n, int S, int rbits, int thr, const int* __restrict__ bq, const int* __restrict__ bsp, const u32* __restrict__ brow,
           const int* __restrict__ src0, int* __restrict__ strip_start /* [0..S] = the scan; [S+1] written here */,
           int* __restrict__ sv, int* __restrict__ sa, u32* __restrict__ srow, int* __restrict__ tile_s0, int* __restrict__ d_M,
           int expect_m, int* __restrict__ counters)
{
    const int M = strip_start[S];
    const int base = blockIdx.x * CMP_BLOCK;
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
        d0[k] = keep[k] ? strip_start[st] : 0;
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
    // what k_after_compact did besides the strip table: tiles behind M, sentinels, M itself
    const int t = blockIdx.x * CMP_TPB + (int)threadIdx.x;
    for (int u = t; u <= n / 256; u += gridDim.x * CMP_TPB) if (u * 256 >= M) tile_s0[u] = S;
    for (int u = t; u < SORT_PAD; u += gridDim.x * CMP_TPB) if (M + u < n) { sv[M + u] = INT_MAX; sa[M + u] = S << rbits; }
    if (t == 0) {
        d_M[0] = M;
        strip_start[S + 1] = n;
        if (expect_m >= 0 && expect_m != M) counters[CTR_OVERFLOW] = 8;      // the host sized the run by a wrong M: fail loudly
    }
}

// workspace for a run over n rows
static int ensure_workspace(cl_chrom* c, int S)
{
    const size_t n = (size_t)c->n;
    int rc;
#define ENS(buf, bytes) if ((rc = c->buf.ensure(bytes))) return rc
    ENS(keys_in, n * 8); ENS(keys_out, n * 8); ENS(vals_in, n * 4); ENS(vals_out, n * 4);
    ENS(sv, (n + 2 * SORT_PAD) * 4); ENS(sa, (n + 2 * SORT_PAD) * 4); ENS(strip, ((size_t)S + 2) * 4); ENS(cnt, n * 4);
    ENS(parent, n * 4); ENS(root, n * 4); ENS(head, n * 4); ENS(cellfirst, n * 4);
    ENS(compkey, n * 4); ENS(ncore, n * 4); ENS(bsize, n * 4); ENS(owner, n * 4); ENS(state, n * 4);
    ENS(flag, (n + 1) * 4); ENS(rankscan, (n + 1) * 4); ENS(hdr, 256);
    ENS(slot[c->cur].labels, n * 4); ENS(slot[c->cur].table, (n + 1) * sizeof(cl_box)); ENS(slot[c->cur].slab, n * 4);
    ENS(ulist, n * 4); ENS(lo, n * 4); ENS(hi, n * 4); ENS(recs, n * sizeof(Rec)); ENS(counters, 256);
    ENS(chainflag, n * 4); ENS(chainhead, n * 4); ENS(usize, n * 4); ENS(tile_s0, (n / 256 + 2) * 4); ENS(tileflag, (n / 256 + 2) * 4);
#undef ENS
    if (c->sv.fresh || c->sa.fresh) {
        // sentinel pads around the sorted arrays (k_region_core stages its windows without bounds checks)
        hipLaunchKernelGGL(k_init_pads, dim3(nblocks(2 * SORT_PAD)), dim3(TPB), 0, c->stream, c->sv.as<int>(), c->sa.as<int>(), (long long)n);
        c->sv.fresh = c->sa.fresh = false;
    }
    // rocPRIM temporary storage
    size_t sort_bytes = 0, scan_bytes = 0, scan2 = 0;
    hipError_t e = rocprim::radix_sort_pairs<SortConfig>(nullptr, sort_bytes, (u64*)nullptr, (u64*)nullptr, (u32*)nullptr, (u32*)nullptr,
                                             n, 0, 64, c->stream);
    if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs size query", hipGetErrorString(e));
    {
        size_t sb2 = 0;
        e = rocprim::radix_sort_pairs<SortConfig>(nullptr, sb2, (u32*)nullptr, (u32*)nullptr, (u64*)nullptr, (u64*)nullptr, n, 0, 32, c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs size query (q index)", hipGetErrorString(e));
        sort_bytes = std::max(sort_bytes, sb2);
    }
    e = rocprim::inclusive_scan(nullptr, scan_bytes, (int*)nullptr, (int*)nullptr, n, rocprim::maximum<int>(), c->stream);
    if (e != hipSuccess) return fail(CL_ERR_HIP, "inclusive_scan size query", hipGetErrorString(e));
    e = rocprim::exclusive_scan(nullptr, scan2, (int*)nullptr, (int*)nullptr, 0, n + 1, rocprim::plus<int>(), c->stream);
    if (e != hipSuccess) return fail(CL_ERR_HIP, "exclusive_scan size query", hipGetErrorString(e));
    if ((rc = c->sort_tmp.ensure(std::max<size_t>(sort_bytes, 16)))) return rc;
    size_t scan3 = 0;
    e = rocprim::inclusive_scan_by_key(nullptr, scan3, rocprim::make_reverse_iterator((int*)nullptr), rocprim::make_reverse_iterator((int*)nullptr),
                                       rocprim::make_reverse_iterator((int*)nullptr), n, rocprim::minimum<int>(), rocprim::equal_to<int>(), c->stream);
    if (e != hipSuccess) return fail(CL_ERR_HIP, "inclusive_scan_by_key size query", hipGetErrorString(e));
    if ((rc = c->scan_tmp.ensure(std::max<size_t>(std::max(std::max(scan_bytes, scan2), scan3), 16)))) return rc;
    return CL_OK;
}

static int bits_for(unsigned v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }

// Build GridParams for the rotated-strip layout (variants 1 and 2)
static int make_grid(cl_chrom* c, int variant, int eps, int minPts, int cut, GridParams* g)
{
    g->eps = eps; g->minPts = minPts; g->cut = cut; g->variant = variant;
    g->dbg = 0;
#ifdef CLOOPS_DEVEL
    // developer build only (-DCLOOPS_DEVEL): ablation knobs that can change results; never in the shipped library
    { const char* e = getenv("CLOOPS_DBG"); g->dbg = e ? atoi(e) : 0; }
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
    g->rbits = bits_for((unsigned)(eps - 1));
    g->peps = 1 << g->rbits;
    // the strip coordinate lives in the kernels as sp = strip << rbits | remainder (GridParams): sp + peps must stay an int
    if ((S + 2) << g->rbits > (long long)INT_MAX)
        return fail(CL_ERR_GRID, "coordinate extent too large for this eps (X+Y range + 2*eps must stay below 2^30)");
    return CL_OK;
}

#define LAUNCH(kernel, nthreads, ...) \
    hipLaunchKernelGGL(kernel, dim3(nblocks(nthreads)), dim3(TPB), 0, c->stream, __VA_ARGS__)

static void ev_record(cl_chrom* c, int k)
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
        LAUNCH(k_make_spkeys, n, n, g, (const u32*)c->qb_key.as<u32>(), (const u64*)c->qb_val.as<u64>(), k32_in, v64_in);
        ev_record(c, 1);
        size_t tb = c->sort_tmp.bytes;
        hipError_t e = rocprim::radix_sort_pairs<SortConfig>(c->sort_tmp.p, tb, k32_in, k32_out, v64_in, v64_out, (size_t)n, g.rbits, g.rbits + strip_bits, c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs(strips)", hipGetErrorString(e));
        LAUNCH(k_strip_table32, g.S + 2, (const u32*)k32_out, n, g.S, g.rbits, dstrip);
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
            if ((rc = sort_layout(c, g0, c->bq.as<int>() + SORT_PAD, c->bsp.as<int>() + SORT_PAD, c->brow.as<u32>(),
                                  c->bstrip.as<int>(), c->btile.as<int>()))) return rc;
            c->base.valid = true; c->base.layout = layout; c->base.eps = g.eps;
        } else {
            ev_record(c, 0);
            ev_record(c, 1);
        }
        const long long dmin = g.swap ? c->st.amin : 0;   // swap == 0 is a developer layout: always compacts
        if (g.cut <= 0 || (g.swap && (long long)g.cut <= dmin)) {
            // no row is filtered: the run works on the base layout itself
            c->w_sv = c->bq.as<int>() + SORT_PAD; c->w_sa = c->bsp.as<int>() + SORT_PAD; c->srow = c->brow.as<u32>();
            c->w_strip = c->bstrip.as<int>(); c->w_tile = c->btile.as<int>();
        } else {
            if (!g.swap) return fail(CL_ERR_ARG, "internal: layout reuse needs the v-band layout");
            // stable compaction of the base layout by d = q + V0 >= cut (pipe.py:59-62): same order as sorting the
            // filtered rows, one pass over 12 B/PET
            const int nb = nblocks(n, CMP_BLOCK);
            if ((rc = c->sel_tmp.ensure(((size_t)g.S + 2) * 8 + 64))) return rc;
            int* kept = c->sel_tmp.as<int>();
            int* src0 = kept + g.S + 2;
            int* d_M = c->counters.as<int>() + CTR_M;
            const int thr = g.cut - g.V0;
            const int* bq = c->bq.as<int>() + SORT_PAD;
            LAUNCH(k_cut_strips, g.S + 1, g.S, thr, (const int*)c->bstrip.as<int>(), bq, kept, src0);
            size_t tb = c->scan_tmp.bytes;
            hipError_t e = rocprim::exclusive_scan(c->scan_tmp.p, tb, kept, c->strip.as<int>(), 0, (size_t)g.S + 1, rocprim::plus<int>(), c->stream);
            if (e != hipSuccess) return fail(CL_ERR_HIP, "exclusive_scan(cut)", hipGetErrorString(e));
            hipLaunchKernelGGL(k_cut_copy, dim3(nb), dim3(CMP_TPB), 0, c->stream, n, g.S, g.rbits, thr, bq, (const int*)(c->bsp.as<int>() + SORT_PAD),
                               (const u32*)c->brow.as<u32>(), (const int*)src0, c->strip.as<int>(), wsv, wsa, c->vals_out.as<u32>(), c->tile_s0.as<int>(),
                               d_M, c->run_m_exact ? c->run_m : -1, c->counters.as<int>());
            c->w_sv = wsv; c->w_sa = wsa; c->srow = c->vals_out.as<u32>();
            c->w_strip = c->strip.as<int>(); c->w_tile = c->tile_s0.as<int>();
        }
    }
    ev_record(c, 2);
    if ((rc = cl_launch_region(c->stream, g, n, c->run_m, exact, c->w_sv, c->w_sa, c->w_strip, c->w_tile, c->cnt.as<int>()))) return rc;
    ev_record(c, 3);
    HIP_TRY(hipGetLastError());
    return CL_OK;
}

__global__ void k_nop() {}

static int ensure_events(cl_chrom* c)
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

static Table make_table_slot(cl_chrom* c, int slot)
{
    Table t;
    int* base = c->slot[slot].table.as<int>();
    const size_t stride = (size_t)c->n + 1;
    t.count = base; t.minx = base + stride; t.maxx = base + 2 * stride; t.miny = base + 3 * stride; t.maxy = base + 4 * stride;
    return t;
}
static Table make_table(cl_chrom* c) { return make_table_slot(c, c->cur); }

// the upload's distance histogram, when it covers every PET a cut of `cut` drops (K7Src::dh)
static const int* k7_hist_for(cl_chrom* c, int cut)
{
    return (cut > 0 && cut < DCUM_BINS && c->n_neg == 0 && !c->dcum.empty() && c->dhist.p) ? c->dhist.as<int>() : (const int*)nullptr;
}

static int finish_enqueue(cl_chrom* c, int n_strips, const int* d_M, int32_t* labels_out)
{
    const int n = (int)c->n;
    cl_chrom::Slot& sl = c->slot[c->cur];
    int* dh = c->hdr.as<int>() + 16 * c->cur;
    if (sl.h_boxes_cap == 0) {
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
        if (!sl.h_step) HIP_TRY(hipHostMalloc((void**)&sl.h_step, out_bytes, hipHostMallocDefault));
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
        src.sorted = sl.sorted_src ? 1 : 0; src.n = n; src.M = 0; src.v0 = sl.k7_v0; src.dM = d_M;
        src.X = c->d_x; src.Y = c->d_y; src.labels = sl.labels.as<int>(); src.sv = sl.k7_sv; src.slab = sl.slab.as<int>();
        src.dh = k7_hist_for(c, c->pending_cut);
        hipLaunchKernelGGL(k7_summary, dim3(K7_BLOCKS), dim3(TPB), 0, c->stream, src, c->pending_cut, cls, parts, lh,
                           (unsigned)c->pending_fine_lo, c->pending_fine_lo >= 0 ? lh + K7_LOGBINS : (unsigned long long*)nullptr);
        static_assert((16 + sizeof(K7Part)) % 8 == 0, "step output in 8-byte words");
        hipLaunchKernelGGL(k7_reduce_parts, dim3(1), dim3(256), 0, c->stream, (const K7Part*)parts, K7_BLOCKS, (K7Part*)(ds + 16),
                           (const int*)bcount, nb, (long long*)ds, (const unsigned long long*)ds, (int)(out_bytes / 8),
                           (unsigned long long*)sl.h_step, (const int*)dh, sl.h_hdr);
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
        HIP_TRY(hipStreamWaitEvent(c->copy_stream, sl.ev_done, 0));
        HIP_TRY(hipMemcpyAsync(sl.h_hdr, dh, 32, hipMemcpyDeviceToHost, c->copy_stream));
        if (sl.step_valid) HIP_TRY(hipMemcpyAsync(sl.h_step, sl.d_step.p, 16 + sizeof(K7Part) + K7_LOGBINS * 8 + K7_FINE * 8, hipMemcpyDeviceToHost, c->copy_stream));
        if (labels_out) HIP_TRY(hipMemcpyAsync(labels_out, sl.labels.p, (size_t)n * 4, hipMemcpyDeviceToHost, c->copy_stream));
        if (c->profiling) (void)hipEventRecord(sl.ev[7], c->copy_stream);
        HIP_TRY(hipEventRecord(sl.ev_copied, c->copy_stream));
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
    const int K = sl.h_hdr[0];
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
    }
    return CL_OK;
}


static int run_rotated(cl_chrom* c, int variant, int eps, int minPts, int cut, int32_t* labels_out);

// ---- variant 3 host driver -----------------------------------------------------------------
static int run_block(cl_chrom* c, int eps, int minPts, int cut, int32_t* labels_out)
{
    int rc;
    const int n = (int)c->n;
    long long R = ((long long)c->st.xmax - c->st.xmin) / eps + 1;
    if (R > (1LL << 28)) return fail(CL_ERR_GRID, "eps too small for the coordinate extent (cell-row table > 2^28 rows)");
    if ((rc = ensure_workspace(c, (int)R))) return rc;
    if ((rc = ensure_events(c))) return rc;
#define ENSB(buf, bytes) if ((rc = c->buf.ensure(bytes))) return rc
    ENSB(b_cstart, ((size_t)n + 1) * 4); ENSB(b_ckey, (size_t)n * 8); ENSB(b_nb, (size_t)n * 32);
    ENSB(b_cx, (size_t)n * 8); ENSB(b_cy, (size_t)n * 8);
#undef ENSB
    BlkParams p; p.eps = eps; p.minPts = minPts; p.cut = cut; p.R = (int)R; p.n = n;
    p.nyb = std::max(1, bits_for((unsigned)(((long long)c->st.ymax - c->st.ymin) / eps)));
    p.rb = bits_for((unsigned)(eps - 1));
    {
        const unsigned d = (unsigned)eps;
        int l = 0; while ((1ull << l) < d) ++l;
        p.magic = (u32)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
        p.sh1 = l < 1 ? l : 1; p.sh2 = l > 1 ? l - 1 : 0;
    }
    int* counters = c->counters.as<int>();
    BlkScalars* sc = (BlkScalars*)(counters + 32);
    LAUNCH(k_init_arrays, n + 1, n, c->parent.as<int>(), c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(),
           c->usize.as<int>(), c->cellfirst.as<int>(), c->flag.as<int>(), c->state.as<int>(), counters);
    HIP_TRY(hipMemsetAsync(c->slot[c->cur].labels.p, 0xFF, (size_t)n * 4, c->stream));
    // minX / minY of the (filtered) mat (blockDBSCAN.py:74-80): known from the upload statistics when
    // nothing is filtered, one reduction pass otherwise
    hipLaunchKernelGGL(k_blk_init_scalars, dim3(1), dim3(1), 0, c->stream, sc, cut > 0 ? INT_MAX : c->st.xmin,
                       cut > 0 ? INT_MAX : c->st.ymin);
    ev_record(c, 0);
    if (cut > 0)
        hipLaunchKernelGGL(k_blk_minmax, dim3(std::min(nblocks(n), 2048)), dim3(TPB), 0, c->stream, c->d_x, c->d_y, n, cut, sc);
    LAUNCH(k_blk_keys, n, c->d_x, c->d_y, p, sc, c->keys_in.as<u64>(), c->vals_in.as<u32>());
    ev_record(c, 1);
    {
        size_t tmp_bytes = c->sort_tmp.bytes;
        const int begin_bit = 2 * p.rb;
        const int end_bit = begin_bit + p.nyb + std::max(1, bits_for((unsigned)p.R));
        hipError_t e = rocprim::radix_sort_pairs<SortConfig>(c->sort_tmp.p, tmp_bytes, c->keys_in.as<u64>(), c->keys_out.as<u64>(),
                                                 c->vals_in.as<u32>(), c->vals_out.as<u32>(), (size_t)n, begin_bit, end_bit, c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs", hipGetErrorString(e));
    }
    u64* skeys = c->keys_out.as<u64>();
    u32* srow = c->vals_out.as<u32>();
    int* sx = (c->sv.as<int>() + SORT_PAD);
    int* sy = (c->sa.as<int>() + SORT_PAD);
    int* headflag = c->chainflag.as<int>();
    int* cidp1 = c->chainhead.as<int>();
    LAUNCH(k_blk_gather, n, p, skeys, sx, sy, headflag, sc);
    {
        size_t tb = c->scan_tmp.bytes;
        hipError_t e = rocprim::inclusive_scan(c->scan_tmp.p, tb, headflag, cidp1, (size_t)n, rocprim::plus<int>(), c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "inclusive_scan(cells)", hipGetErrorString(e));
    }
    int* cstart = c->b_cstart.as<int>();
    u64* ckey = c->b_ckey.as<u64>();
    int* cfirst = c->cellfirst.as<int>();
    int* rowcell = c->strip.as<int>();
    LAUNCH(k_blk_cells, n, p, skeys, headflag, cidp1, srow, sc, cstart, ckey, cfirst);
    LAUNCH(k_blk_rowtable, p.R + 1, p, sc, ckey, rowcell);
    ev_record(c, 2);
    int* nb = c->b_nb.as<int>();
    int* low = c->ncore.as<int>();
    int* alive = c->bsize.as<int>();
    double* cx = c->b_cx.as<double>();
    double* cy = c->b_cy.as<double>();
    int* linkbits = c->owner.as<int>();
    int* corec = c->state.as<int>();
    LAUNCH(k_blk_neighbors, n, p, sc, ckey, rowcell, cstart, sx, sy, nb, low, cx, cy);
    LAUNCH(k_blk_alive, n, sc, nb, low, cstart, alive, linkbits);
    hipLaunchKernelGGL(k_blk_links, dim3((unsigned)(((size_t)n * 4 + TPB - 1) / TPB)), dim3(TPB), 0, c->stream,
                       p, sc, nb, alive, cstart, sx, sy, cx, cy, linkbits);
    LAUNCH(k_blk_core, n, p, sc, nb, alive, cstart, linkbits, corec);
    ev_record(c, 3);
    LAUNCH(k_blk_union, n, sc, nb, linkbits, corec, c->parent.as<int>());
    hipLaunchKernelGGL(k_blk_flatten, dim3(nblocks(n, BIGTPB)), dim3(BIGTPB), 0, c->stream, sc, corec, c->parent.as<int>(), cfirst,
                       c->root.as<int>(), c->compkey.as<int>());
    ev_record(c, 4);
    LAUNCH(k_blk_rank_flags, n, sc, c->root.as<int>(), c->compkey.as<int>(), c->flag.as<int>());
    {
        size_t tb = c->scan_tmp.bytes;
        hipError_t e = rocprim::exclusive_scan(c->scan_tmp.p, tb, c->flag.as<int>(), c->rankscan.as<int>(), 0, (size_t)n + 1,
                                               rocprim::plus<int>(), c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "exclusive_scan", hipGetErrorString(e));
    }
    ev_record(c, 5);
    Table t = make_table(c);
    LAUNCH(k_init_table, n + 1, t, c->rankscan.as<int>(), n);
    int* clab = c->cnt.as<int>();
    LAUNCH(k_blk_cell_labels, n, sc, nb, linkbits, alive, c->root.as<int>(), c->compkey.as<int>(), c->rankscan.as<int>(), clab);
    hipLaunchKernelGGL(k_blk_point_labels, dim3(nblocks(n, BIGTPB)), dim3(BIGTPB), 0, c->stream, p, sc, cidp1, clab, srow, sx, sy, c->slot[c->cur].labels.as<int>(), t);
    HIP_TRY(hipGetLastError());
    // blockDBSCAN.py:74: an empty (fully filtered) mat raises only when the class is called
    // on it; pipe.py:64-65 returns before that, so cut > 0 with no survivors is just empty.
    { cl_chrom::Slot& sl = c->slot[c->cur]; sl.rows_valid = true; sl.sorted_src = false; }
    return finish_enqueue(c, p.R + 1, &sc->M, labels_out);
}


// ---- variant 1 under the weighted metric (K9) ------------------------------------------------
static int run_weighted(cl_chrom* c, int eps, int minPts, int wx, int wy, int32_t* labels_out)
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
    hipLaunchKernelGGL(k_flatten, dim3(nblocks(n, BIGTPB * FLAT_PER)), dim3(BIGTPB), 0, c->stream, gi, strip, cnt, c->parent.as<int>(), srow,
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

extern "C" int cl_cluster_step_async(cl_chrom* c, int variant, int32_t eps, int32_t min_pts, int32_t cut, int32_t step, int64_t fine_lo)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    if (step < 0) return fail(CL_ERR_ARG, "cl_cluster_step_async: step must be >= 0");
    if (variant == CL_VARIANT_BLOCK) return fail(CL_ERR_ARG, "cl_cluster_step_async: rotated variants only");
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

static int run_rotated(cl_chrom* c, int variant, int eps, int minPts, int cut, int32_t* labels_out)
{
    int rc;
    GridParams g;
    if ((rc = make_grid(c, variant, eps, minPts, cut, &g))) return rc;
    if ((rc = ensure_workspace(c, g.S))) return rc;
    if ((rc = ensure_events(c))) return rc;
    const int n = (int)c->n;
    int* cnt = c->cnt.as<int>();
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
    LAUNCH(k_init_flags, nw + 1, nw, c->flag.as<int>(), counters);
    // row-aligned labels only when somebody reads them: k_final_labels then writes the label (or -1) of every PET that
    // entered DBSCAN and only the rows removed by the cut filter need the -1 fill
    const bool rows = labels_out != nullptr || c->device_labels;
    if (rows && cut > 0) HIP_TRY(hipMemsetAsync(c->slot[c->cur].labels.p, 0xFF, (size_t)n * 4, c->stream));
    if ((rc = run_sort_and_count(c, g, false))) return rc;
    {
        cl_chrom::Slot& sl = c->slot[c->cur];
        sl.rows_valid = rows; sl.sorted_src = g.swap != 0; sl.k7_sv = c->w_sv; sl.k7_v0 = g.V0;
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
    {
        // own-strip chains; variant 2: the same tile kernel also finds every PET's cell head
        int* head = variant == CL_VARIANT_CDBSCAN2 ? c->head.as<int>() : nullptr;
        TILE_LAUNCH(k_chain_flags, g, ntiles, nm, sv, sa, strip, cnt, c->chainflag.as<int>(),
                           head, c->chainhead.as<int>() /* wavelast: the buffer is free until the labels */);
        if (head) {
            // cellfirst: segmented suffix-min of the input rows, keyed by the cell's head index, so that
            // cellfirst[head] = smallest row of the whole cell (replaces one atomicMin per PET)
            size_t tb = c->scan_tmp.bytes;
            hipError_t e = rocprim::inclusive_scan_by_key(c->scan_tmp.p, tb, rocprim::make_reverse_iterator(head + nm),
                                               rocprim::make_reverse_iterator((int*)srow + nm),
                                               rocprim::make_reverse_iterator(c->cellfirst.as<int>() + nm), (size_t)nm,
                                               rocprim::minimum<int>(), rocprim::equal_to<int>(), c->stream);
            if (e != hipSuccess) return fail(CL_ERR_HIP, "inclusive_scan_by_key", hipGetErrorString(e));
        }
        // long strips (dense data at large eps): 32-PET block summaries for the union scan (`hi` is free until K4)
        pmax32 = ((long long)n > 64LL * g.S) ? c->hi.as<int>() : nullptr;
        LAUNCH(k_chain_parent, (nm + CP_PER - 1) / CP_PER, strip, g.S, cnt, g.minPts, c->chainhead.as<int>(), c->parent.as<int>(), c->chainflag.as<int>(),
               c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(), c->usize.as<int>(), c->state.as<int>(),
               sv, c->lo.as<int>(), sa, pmax32);   // chain ends live in `lo` until the release fix-up reuses it
    }
    // the union walk looks one strip back, i.e. about one strip population in front of the PET: a 256-PET halo keeps most of
    // those windows in LDS on dense data (chr1 of the 200 M genome, eps 5000-10000: -12..-16 %); short strips stay with 128
    const int union_halo = (long long)n > 40LL * g.S ? 256 : 128;
    TILE_LAUNCH_H((wide == 2 || wide == 4) ? 512 : union_halo, k_union_cores, g, ntiles, sv, sa, strip, c->chainflag.as<int>(), c->lo.as<int>(),
                       pmax32, c->parent.as<int>());
    hipLaunchKernelGGL(k_flatten, dim3(nblocks(nm, BIGTPB * FLAT_PER)), dim3(BIGTPB), 0, c->stream, g, strip, cnt, c->parent.as<int>(), srow, c->head.as<int>(), c->cellfirst.as<int>(),
           c->root.as<int>(), c->compkey.as<int>(), c->ncore.as<int>(), c->chainflag.as<int>() /* root list: the chain ids are dead */, counters);
    int* rootlist = c->chainflag.as<int>();
    ev_record(c, 4);
    // K4
    TILE_LAUNCH_H((wide >= 2 && wide <= 4) ? 512 : (wide >= 5 ? 256 : 128), k_border, g, ntiles, sv, sa, strip, c->root.as<int>(), c->compkey.as<int>(),
                       c->ncore.as<int>(), srow, c->owner.as<int>(), c->bsize.as<int>(), c->usize.as<int>(), cnt, c->tileflag.as<int>());
    if (variant == CL_VARIANT_CDBSCAN2) {
        const int rec_cap = n;
        hipLaunchKernelGGL(k_mark_uncertain_l, dim3(512), dim3(TPB), 0, c->stream, g, rootlist, c->ncore.as<int>(), c->bsize.as<int>(), c->state.as<int>(),
                           c->ulist.as<int>(), counters);
        TILE_LAUNCH(k_emit_records, g, ntiles, sv, sa, strip, c->root.as<int>(),
                           c->compkey.as<int>(), c->state.as<int>(), c->owner.as<int>(), c->recs.as<Rec>(), rec_cap, counters,
                           (const int*)c->tileflag.as<int>(), ntiles);
        hipLaunchKernelGGL(k_resolve_release, dim3(1), dim3(1024), 0, c->stream, minPts, c->ncore.as<int>(), c->usize.as<int>(), c->state.as<int>(),
                           c->ulist.as<int>(), c->recs.as<Rec>(), c->lo.as<int>(), c->hi.as<int>(), counters);
    }
    ev_record(c, 5);
    // K5
    hipLaunchKernelGGL(k_rank_bits_l, dim3(512), dim3(TPB), 0, c->stream, g, rootlist, counters, c->compkey.as<int>(), c->state.as<int>(), c->flag.as<unsigned>(),
                       variant == CL_VARIANT_CDBSCAN2 ? (const Rec*)c->recs.as<Rec>() : (const Rec*)nullptr, c->owner.as<int>());
    {
        size_t tb = c->scan_tmp.bytes;
        hipError_t e = rocprim::exclusive_scan(c->scan_tmp.p, tb, rocprim::make_transform_iterator(c->flag.as<unsigned>(), PopcWord()),
                                               c->rankscan.as<int>(), 0, (size_t)nw + 1, rocprim::plus<int>(), c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "exclusive_scan", hipGetErrorString(e));
    }
    c->k_total = c->rankscan.as<int>() + nw;
    Table t = make_table(c);
    // rlabel reuses the chainhead buffer (free after k_chain_parent); the kernel also resets the table rows of the ids handed out
    hipLaunchKernelGGL(k_root_labels_bits_l, dim3(512), dim3(TPB), 0, c->stream, g, rootlist, counters, c->compkey.as<int>(), c->ncore.as<int>(), c->bsize.as<int>(),
                       c->state.as<int>(), c->flag.as<unsigned>(), c->rankscan.as<int>(), c->chainhead.as<int>(), t, nw,
                       c->hdr.as<int>() + 16 * c->cur, (const int*)(strip + g.S));
    c->hdr_packed = true;
    hipLaunchKernelGGL(k_final_labels, dim3(nblocks(nm, BIGTPB * FINAL_CHUNKS)), dim3(BIGTPB), 0, c->stream, g, strip, sv, sa, srow, c->owner.as<int>(),
                       c->chainhead.as<int>(), rows ? c->slot[c->cur].labels.as<int>() : (int*)nullptr, c->slot[c->cur].slab.as<int>(), t);
    HIP_TRY(hipGetLastError());
    return finish_enqueue(c, g.S + 2, strip + g.S, labels_out);
}


// ---- K7 host entry points --------------------------------------------------------------------
static int k7_prepare(cl_chrom* c)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    if (!c->have_result || c->last_slot < 0) return fail(CL_ERR_ARG, "distance statistics need a completed clustering run");
    if (c->enq != c->deq) return fail(CL_ERR_ARG, "distance statistics: asynchronous runs still in flight");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = c->k7_cls.ensure((size_t)c->n + 16))) return rc;
    if ((rc = c->k7_parts.ensure(K7_BLOCKS * sizeof(K7Part) + K7_LOGBINS * 8 + 4096))) return rc;
    if (!c->k7_classified) {
        int* dh = c->hdr.as<int>() + 16 * c->last_slot;
        LAUNCH(k7_classify, c->n + 1, dh, make_table_slot(c, c->last_slot), c->k7_cls.as<signed char>());
        c->k7_classified = true;
    }
    return CL_OK;
}

static K7Src k7_source(cl_chrom* c, int cut)
{
    cl_chrom::Slot& sl = c->slot[c->last_slot];
    K7Src s{};
    s.dh = k7_hist_for(c, cut);
    s.sorted = sl.sorted_src ? 1 : 0; s.n = (int)c->n; s.M = sl.h_hdr[2]; s.v0 = sl.k7_v0;
    s.X = c->d_x; s.Y = c->d_y; s.labels = sl.labels.as<int>(); s.sv = sl.k7_sv; s.slab = sl.slab.as<int>();
    return s;
}

extern "C" int cl_dist_summary(cl_chrom* c, int32_t cut, cl_dsummary* out)
{
    if (!out) return fail(CL_ERR_ARG, "cl_dist_summary: out is null");
    memset(out, 0, sizeof(*out));
    out->xshift = K7_XSHIFT;
    if (c && c->n == 0) return CL_OK;
    int rc = k7_prepare(c);
    if (rc) return rc;
    if (!c->slot[c->last_slot].sorted_src && !c->slot[c->last_slot].rows_valid) return fail(CL_ERR_ARG, "cl_dist_summary: the last run left no labels");
    unsigned long long* dh = (unsigned long long*)((char*)c->k7_parts.p + K7_BLOCKS * sizeof(K7Part));
    HIP_TRY(hipMemsetAsync(dh, 0, K7_LOGBINS * 8, c->stream));
    K7Part* dpart = (K7Part*)((char*)dh + K7_LOGBINS * 8);                  // behind the histogram (the buffer's spare 4 KB)
    hipLaunchKernelGGL(k7_summary, dim3(K7_BLOCKS), dim3(TPB), 0, c->stream, k7_source(c, cut), cut, c->k7_cls.as<signed char>(), c->k7_parts.as<K7Part>(), dh,
                       0u, (unsigned long long*)nullptr);
    hipLaunchKernelGGL(k7_reduce_parts, dim3(1), dim3(256), 0, c->stream, (const K7Part*)c->k7_parts.as<K7Part>(), K7_BLOCKS, dpart,
                       (const int*)nullptr, 0, (long long*)nullptr, (const unsigned long long*)nullptr, 0, (unsigned long long*)nullptr,
                       (const int*)nullptr, (int*)nullptr);
    K7Part part;
    HIP_TRY(hipMemcpyAsync(&part, dpart, sizeof(K7Part), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(out->loghist, dh, K7_LOGBINS * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int g = 0; g < 2; ++g) { out->sumx[g] = part.sx[g]; out->sumxx[g] = part.sxx[g]; out->n_all[g] = part.n_all[g]; out->n_pos[g] = part.n_pos[g]; }
    out->fine_lo = -1;
    return CL_OK;
}

extern "C" int cl_dist_bin_hist(cl_chrom* c, int32_t cut, uint32_t lo, uint32_t hi, int shift, uint64_t* hist2048)
{
    if (!hist2048) return fail(CL_ERR_ARG, "cl_dist_bin_hist: out is null");
    memset(hist2048, 0, K7_FINE * sizeof(uint64_t));
    if (shift < 0 || shift > 31 || hi < lo || (((uint64_t)hi - lo + ((1ull << shift) - 1)) >> shift) > K7_FINE)
        return fail(CL_ERR_ARG, "cl_dist_bin_hist: (hi - lo) >> shift must fit 2048 bins");
    if (c && c->n == 0) return CL_OK;
    int rc = k7_prepare(c);
    if (rc) return rc;
    unsigned long long* dh = (unsigned long long*)c->k7_parts.p;
    HIP_TRY(hipMemsetAsync(dh, 0, K7_FINE * 8, c->stream));
    const int n = (int)c->n;
    hipLaunchKernelGGL(k7_bin_hist, dim3(std::min(nblocks(n), K7_BLOCKS)), dim3(TPB), 0, c->stream, k7_source(c, cut), cut, c->k7_cls.as<signed char>(),
                       (unsigned)lo, (unsigned)hi, shift, dh);
    HIP_TRY(hipMemcpyAsync(hist2048, dh, K7_FINE * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return CL_OK;
}

// ---- K10 host entry points -----------------------------------------------------------------------
extern "C" int cl_cand_reset(cl_chrom* c)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    c->cand_n = 0;
    return CL_OK;
}

extern "C" int cl_cand_append(cl_chrom* c, int32_t step, int64_t* n_inter, int64_t* n_self)
{
    if (n_inter) *n_inter = 0;
    if (n_self) *n_self = 0;
    if (c && c->n == 0) return CL_OK;
    int rc = k7_prepare(c);                              // classifies the table of the last completed run (pipe.py:83-97)
    if (rc) return rc;
    cl_chrom::Slot& sl = c->slot[c->last_slot];
    const int K = sl.h_hdr[0];
    if (K <= 0) return CL_OK;
    if ((rc = ensure_cand_capacity(c, c->cand_n + K))) return rc;
    const int nb = nblocks(K, CAND_BLOCK);
    if ((rc = c->sel_tmp.ensure((size_t)nb * 12 + 64))) return rc;
    int* bcount = c->sel_tmp.as<int>();
    int* boff = bcount + 2 * nb;
    const int* dK = c->hdr.as<int>() + 16 * c->last_slot;
    hipLaunchKernelGGL(k_cand_count, dim3(nb), dim3(256), 0, c->stream, dK, c->k7_cls.as<signed char>(), bcount, nb);
    size_t tb = c->scan_tmp.bytes;
    hipError_t e = rocprim::exclusive_scan(c->scan_tmp.p, tb, bcount, boff, 0, (size_t)nb, rocprim::plus<int>(), c->stream);
    if (e != hipSuccess) return fail(CL_ERR_HIP, "exclusive_scan(cand)", hipGetErrorString(e));
    hipLaunchKernelGGL(k_cand_append, dim3(nb), dim3(256), 0, c->stream, dK, c->k7_cls.as<signed char>(), make_table_slot(c, c->last_slot),
                       (const int*)boff, (const int*)bcount, (int)c->cand_n, (int)step, (int)std::min<long long>(c->cand_cap, INT_MAX), c->cand_box.as<int4>(), c->cand_step.as<int>());
    std::vector<int> h(2 * nb);
    HIP_TRY(hipMemcpyAsync(h.data(), bcount, (size_t)2 * nb * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    long long ni = 0, ns = 0;
    for (int k = 0; k < nb; ++k) { ni += h[k]; ns += h[nb + k]; }
    if (c->cand_n + ni > c->cand_cap) return fail(CL_ERR_GRID, "internal: candidate buffer overrun");
    c->cand_n += ni;
    if (n_inter) *n_inter = ni;
    if (n_self) *n_self = ns;
    return CL_OK;
}

extern "C" int cl_cand_finish(cl_chrom* c, int32_t final_cut, int32_t* boxes_out, int64_t capacity, int64_t* n_out)
{
    if (!c || !n_out) return fail(CL_ERR_ARG, "cl_cand_finish: null argument");
    *n_out = 0;
    const long long N = c->cand_n;
    if (N == 0) return CL_OK;
    if (c->enq != c->deq) return fail(CL_ERR_ARG, "cl_cand_finish: asynchronous runs still in flight");
    if (N > INT_MAX - 1024) return fail(CL_ERR_GRID, "cl_cand_finish: more than 2^31 candidates");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
    if ((rc = ensure_workspace(c, 1))) return rc;
    if ((rc = c->cand_keep.ensure((size_t)N + 64)) || (rc = c->cand_out.ensure((size_t)N * 16))) return rc;
    const int n = (int)N;
    // the sort buffers of the handle are sized for its PETs; a sweep of many steps on a strongly clustered chromosome can
    // leave more candidates than that: then the dedup sorts in buffers of its own (released at the end), and rocPRIM's
    // temporary storage is sized from N with the configuration the sort below uses
    struct Tmp { DevBuf kin, kout, vin, vout; ~Tmp() { kin.release(); kout.release(); vin.release(); vout.release(); } } tmp;
    u64 *kin = c->keys_in.as<u64>(), *kout = c->keys_out.as<u64>();
    u32 *vin = c->vals_in.as<u32>(), *vout = c->vals_out.as<u32>();
    if (N > c->n) {
        if ((rc = tmp.kin.ensure((size_t)N * 8)) || (rc = tmp.kout.ensure((size_t)N * 8)) || (rc = tmp.vin.ensure((size_t)N * 4)) ||
            (rc = tmp.vout.ensure((size_t)N * 4))) return rc;
        kin = tmp.kin.as<u64>(); kout = tmp.kout.as<u64>(); vin = tmp.vin.as<u32>(); vout = tmp.vout.as<u32>();
    }
    {
        size_t need = 0;
        hipError_t e0 = rocprim::radix_sort_pairs(nullptr, need, (u64*)nullptr, (u64*)nullptr, (u32*)nullptr, (u32*)nullptr, (size_t)n, 0, 64, c->stream);
        if (e0 != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs size query (cand)", hipGetErrorString(e0));
        if ((rc = c->sort_tmp.ensure(std::max<size_t>(need, 16)))) return rc;
    }
    int* flags = c->counters.as<int>() + 60;
    // two different boxes sharing a 64-bit hash would be merged: the exact compare inside k_cand_mark notices, and the
    // pass is redone under another salt (a collision under four independent hashes does not happen)
    int hflag = 0;
    for (int attempt = 0; attempt < 4; ++attempt) {
        HIP_TRY(hipMemsetAsync(flags, 0, 4, c->stream));
        LAUNCH(k_cand_hash, n, n, c->cand_box.as<int4>(), (u64)attempt * 0x9FB21C651E98DF25ull, kin, vin);
        size_t tmp_bytes = c->sort_tmp.bytes;
        hipError_t e = rocprim::radix_sort_pairs(c->sort_tmp.p, tmp_bytes, kin, kout, vin, vout, (size_t)n, 0, 64, c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_pairs(cand)", hipGetErrorString(e));
        LAUNCH(k_cand_mark, n, n, (const u64*)kout, (const u32*)vout, c->cand_box.as<int4>(), c->cand_step.as<int>(), (int)final_cut,
               c->cand_keep.as<unsigned char>(), flags);
        HIP_TRY(hipMemcpyAsync(&hflag, flags, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (hflag == 0) break;
    }
    if (hflag != 0) return fail(CL_ERR_HASH, "candidate dedup: hash collisions under four salts");
    hipError_t e;
    const int nb = nblocks(n, CAND_BLOCK);
    if ((rc = c->sel_tmp.ensure((size_t)nb * 8 + 64))) return rc;
    int* bcount = c->sel_tmp.as<int>();
    int* boff = bcount + nb;
    hipLaunchKernelGGL(k_flag_count, dim3(nb), dim3(256), 0, c->stream, n, c->cand_keep.as<unsigned char>(), bcount);
    size_t tb = c->scan_tmp.bytes;
    e = rocprim::exclusive_scan(c->scan_tmp.p, tb, bcount, boff, 0, (size_t)nb, rocprim::plus<int>(), c->stream);
    if (e != hipSuccess) return fail(CL_ERR_HIP, "exclusive_scan(cand out)", hipGetErrorString(e));
    hipLaunchKernelGGL(k_cand_emit, dim3(nb), dim3(256), 0, c->stream, n, c->cand_keep.as<unsigned char>(), c->cand_box.as<int4>(), (const int*)boff, c->cand_out.as<int4>());
    int tail[2];
    HIP_TRY(hipMemcpyAsync(&tail[0], boff + nb - 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&tail[1], bcount + nb - 1, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    const long long kept = (long long)tail[0] + tail[1];
    *n_out = kept;
    if (kept > capacity) return fail(CL_ERR_ARG, "cl_cand_finish: boxes_out too small");
    if (kept > 0) {
        if (!boxes_out) return fail(CL_ERR_ARG, "cl_cand_finish: boxes_out is null");
        HIP_TRY(hipMemcpyAsync(boxes_out, c->cand_out.p, (size_t)kept * 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return CL_OK;
}

// ---- K8 host entry point ------------------------------------------------------------------------
extern "C" int cl_sig_counts(cl_chrom* c, int32_t cut, int32_t n_records, const int32_t* windows, int32_t* out,
                             int64_t* n_pets)
{
    if (!c) return fail(CL_ERR_ARG, "null chromosome handle");
    if (n_pets) *n_pets = 0;
    if (n_records < 0 || (n_records > 0 && (!windows || !out))) return fail(CL_ERR_ARG, "cl_sig_counts: bad arguments");
    if (c->enq != c->deq) return fail(CL_ERR_ARG, "cl_sig_counts: asynchronous runs still in flight");
    if (c->n == 0) { if (n_records) memset(out, 0, (size_t)n_records * SIG_OUT * 4); return CL_OK; }
    HIP_TRY(hipSetDevice(c->device));
    const int n = (int)c->n;
    int rc;
    if (!c->sig_ready || c->sig_cut != cut) {
        // X-sorted and Y-sorted tables of the PETs that pass parseJd(f, cut); built once per (chromosome, cut)
        if ((rc = c->sig_tx.ensure((size_t)n * 8)) || (rc = c->sig_ty.ensure((size_t)n * 8)) ||
            (rc = c->sig_tmp.ensure((size_t)n * 8)) || (rc = c->sig_m.ensure(64))) return rc;
        LAUNCH(k8_split, n, c->d_x, c->d_y, n, cut, c->sig_tmp.as<u64>(), c->sig_ty.as<u64>());
        size_t bytes = 0;
        hipError_t e = rocprim::radix_sort_keys(nullptr, bytes, (u64*)nullptr, (u64*)nullptr, (size_t)n, 0, 64, c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_keys size query", hipGetErrorString(e));
        if ((rc = c->sig_sorttmp.ensure(std::max<size_t>(bytes, 16)))) return rc;
        bytes = c->sig_sorttmp.bytes;
        e = rocprim::radix_sort_keys(c->sig_sorttmp.p, bytes, c->sig_tmp.as<u64>(), c->sig_tx.as<u64>(), (size_t)n, 0, 64, c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_keys(X)", hipGetErrorString(e));
        HIP_TRY(hipMemcpyAsync(c->sig_tmp.p, c->sig_ty.p, (size_t)n * 8, hipMemcpyDeviceToDevice, c->stream));
        bytes = c->sig_sorttmp.bytes;
        e = rocprim::radix_sort_keys(c->sig_sorttmp.p, bytes, c->sig_tmp.as<u64>(), c->sig_ty.as<u64>(), (size_t)n, 0, 64, c->stream);
        if (e != hipSuccess) return fail(CL_ERR_HIP, "radix_sort_keys(Y)", hipGetErrorString(e));
        hipLaunchKernelGGL(k8_count_valid, dim3(1), dim3(64), 0, c->stream, c->sig_tx.as<u64>(), n, c->sig_m.as<int>());
        c->sig_ready = true; c->sig_cut = cut;
    }
    int hm = 0;
    HIP_TRY(hipMemcpyAsync(&hm, c->sig_m.p, 4, hipMemcpyDeviceToHost, c->stream));
    if (n_records > 0) {
        if ((rc = c->sig_win.ensure((size_t)n_records * sizeof(SigWin))) || (rc = c->sig_out.ensure((size_t)n_records * SIG_OUT * 4))) return rc;
        HIP_TRY(hipMemcpyAsync(c->sig_win.p, windows, (size_t)n_records * sizeof(SigWin), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k8_counts, dim3(n_records), dim3(TPB), 0, c->stream, c->sig_tx.as<u64>(), c->sig_ty.as<u64>(), c->sig_m.as<int>(),
                           n_records, c->sig_win.as<SigWin>(), c->sig_out.as<int>());
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(out, c->sig_out.p, (size_t)n_records * SIG_OUT * 4, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (n_pets) *n_pets = hm;
    return CL_OK;
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

// k_sweep.hip -- the sweep side of libcloops_hip.so: K7 (distance statistics of a step, cLoops/ests.py:36-61), K10 (candidate
// loops of a sweep: combineTwice / filterClusterByDis, cLoops/pipe.py:130-174), K8 (interval counts of the significance
// step, cLoops/cModel.py) -- kernels and their C entry points.
#include "cl_chrom.h"
#include "cl_log2_tab.h"

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
                                 // of a 66-element sequential loop per thread uncovered: 177 us -> see DESIGN.md)

// log2 of an integer distance 1 <= d < 2^31, to the last bit or two of a double: d = 2^e m, m in [1, 2) exact; the top 7 bits of m
// pick an interval with centre c; r = m / c - 1 (one FMA with the tabulated reciprocal, |r| < 2^-8); log2 d = e + log2 c + log2(1 + r)
// with six terms of the series -- a third of the instructions of the library's general-purpose log2 (k7_summary takes one per
// clustered PET).  rcp / lg: the tables of cl_log2_tab.h, in LDS.
#ifndef K7_LIBM_LOG2
#define K7_LIBM_LOG2 0             // 1: the library's log2 (A/B)
#endif
__device__ __forceinline__ double k7_log2(unsigned d, const double* __restrict__ rcp, const double* __restrict__ lg)
{
    const int e = 31 - __clz((int)d);
    const unsigned k = ((d << (31 - e)) >> 24) & 127u;
    const double m = __hiloint2double((int)(0x3ff00000u | (((d << (31 - e)) & 0x7fffffffu) >> 11)), (int)(d << (31 - e) << 21));
    const double r = fma(m, rcp[k], -1.0);
    double p = -0.24044917348149393;                                  // -1 / (6 ln 2)
    p = fma(p, r, 0.28853900817779268);                               //  1 / (5 ln 2)
    p = fma(p, r, -0.36067376022224085);                              // -1 / (4 ln 2)
    p = fma(p, r, 0.48089834696298783);                               //  1 / (3 ln 2)
    p = fma(p, r, -0.72134752044448170);                              // -1 / (2 ln 2)
    p = fma(p, r, 1.4426950408889634);                                //  1 / ln 2
    return ((double)e + lg[k]) + p * r;
}
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
        // (list form: the labelled candidates of the run -- its cores and walkers -- instead of every PET of the layout)
        const int M = s.sorted == 2 ? s.dM[0] + s.dM[1] : (s.dM ? s.dM[0] : s.M);
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
__global__ void __launch_bounds__(1024)
k7_summary(K7Src s, int cut, const signed char* __restrict__ cls, K7Part* __restrict__ parts, unsigned long long* __restrict__ loghist,
           unsigned fine_lo, unsigned long long* __restrict__ fine /* or null: exact histogram of the self group's fine_lo <= |d| < fine_lo + 2048 */)
{
    __shared__ unsigned int h[K7_LOGBINS];
    __shared__ unsigned int hf[K7_FINE];
    __shared__ double l_rcp[128], l_lg[128];
    if (threadIdx.x < 128) { l_rcp[threadIdx.x] = K7_RCP[threadIdx.x]; l_lg[threadIdx.x] = K7_LOG[threadIdx.x]; }
    for (int k = threadIdx.x; k < K7_LOGBINS; k += blockDim.x) h[k] = 0u;
    for (int k = threadIdx.x; k < K7_FINE; k += blockDim.x) hf[k] = 0u;
    __syncthreads();
    double sx[2] = {0.0, 0.0}, sxx[2] = {0.0, 0.0};
    long long na[2] = {0, 0}, np_[2] = {0, 0};
    const bool want_fine = fine != nullptr;
    k7_for_each(s, cut, cls, [&](int g, int ad, int w) {                  // w PETs of this group and distance
        na[g] += w;
        if (ad > 0) {
            const double x = (K7_LIBM_LOG2 ? log2((double)ad) : k7_log2((unsigned)ad, l_rcp, l_lg)) - K7_XSHIFT, wx = (double)w * x;
            np_[g] += w; sx[g] += wx; sxx[g] += wx * x;
            if (g == 1) {
                atomicAdd(&h[k7_logbin((unsigned)ad)], (unsigned)w);
                const unsigned off = (unsigned)ad - fine_lo;                  // wraps for ad < fine_lo: out of range
                if (want_fine && off < (unsigned)K7_FINE) atomicAdd(&hf[off], (unsigned)w);
            }
        }
    });
    __shared__ double s_d[4][16];
    __shared__ long long s_n[4][16];
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
            for (int w = 0; w < (int)blockDim.x / 64; ++w) { a += s_d[g][w]; b += s_d[2 + g][w]; c += s_n[g][w]; d += s_n[2 + g][w]; }
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
    s.sorted = sl.sorted_src ? (sl.k7_lcnt ? 2 : 1) : 0; s.n = (int)c->n; s.M = sl.h_hdr[2]; s.v0 = sl.k7_v0;
    s.dM = sl.k7_lcnt;
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

// combineTwice + filterClusterByDis over the chromosome's candidate buffer -> the surviving boxes in c->cand_out (device), *kept of them
static int cand_finish_core(cl_chrom* c, int32_t final_cut, long long* kept_out);

extern "C" int cl_cand_finish(cl_chrom* c, int32_t final_cut, int32_t* boxes_out, int64_t capacity, int64_t* n_out)
{
    if (!c || !n_out) return fail(CL_ERR_ARG, "cl_cand_finish: null argument");
    *n_out = 0;
    long long kept = 0;
    int rc = cand_finish_core(c, final_cut, &kept);
    if (rc) return rc;
    *n_out = kept;
    if (kept > capacity) return fail(CL_ERR_ARG, "cl_cand_finish: boxes_out too small");
    if (kept > 0) {
        if (!boxes_out) return fail(CL_ERR_ARG, "cl_cand_finish: boxes_out is null");
        HIP_TRY(hipMemcpyAsync(boxes_out, c->cand_out.p, (size_t)kept * 16, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return CL_OK;
}

extern "C" int cl_cand_finish_device(cl_chrom* c, int32_t final_cut, const int32_t** dev_rows_out, int64_t* n_out)
{
    if (!c || !n_out || !dev_rows_out) return fail(CL_ERR_ARG, "cl_cand_finish_device: null argument");
    *n_out = 0; *dev_rows_out = nullptr;
    long long kept = 0;
    int rc = cand_finish_core(c, final_cut, &kept);
    if (rc) return rc;
    *n_out = kept;
    *dev_rows_out = kept > 0 ? (const int32_t*)c->cand_out.p : nullptr;      // valid until the handle's next sweep finishes (or it is destroyed)
    return CL_OK;
}

static int cand_finish_core(cl_chrom* c, int32_t final_cut, long long* kept_out)
{
    *kept_out = 0;
    const long long N = c->cand_n;
    if (N == 0) return CL_OK;
    if (c->enq != c->deq) return fail(CL_ERR_ARG, "cl_cand_finish: asynchronous runs still in flight");
    if (N > INT_MAX - 1024) return fail(CL_ERR_GRID, "cl_cand_finish: more than 2^31 candidates");
    HIP_TRY(hipSetDevice(c->device));
    int rc;
#ifdef CLOOPS_DEVEL
    struct CfTrace {
        double t[8]; int k = 0; long long n;
        static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
        void mark() { if (k < 8) t[k++] = now(); }
        ~CfTrace() { mark(); if (getenv("CLOOPS_TRACE_CAND")) { fprintf(stderr, "[cand n=%lld]", n); for (int i = 1; i < k; ++i) fprintf(stderr, " %.2f", t[i] - t[i - 1]); fprintf(stderr, " ms\n"); } }
    } cft; cft.n = c->n; cft.mark();
#define CF_MARK() cft.mark()
#else
#define CF_MARK() do { } while (0)
#endif
    if ((rc = ensure_workspace(c, 1))) return rc;
    if ((rc = c->cand_keep.ensure((size_t)N + 64)) || (rc = c->cand_out.ensure((size_t)N * 16))) return rc;
    CF_MARK();
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
    CF_MARK();
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
    CF_MARK();
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
    CF_MARK();
    *kept_out = (long long)tail[0] + tail[1];
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


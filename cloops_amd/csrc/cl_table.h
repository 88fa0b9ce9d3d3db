// cl_table.h -- the device cluster table {minX, maxX, minY, maxY, count} per id (cLoops/pipe.py:78-102) and its two-level
// reduce-by-key (wave reductions on the DPP network, an LDS hash table per workgroup, one global atomic per key and
// workgroup): shared by the label kernels of all variants.
#pragma once
#include "cl_common.h"

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

__device__ __forceinline__ int wave_min_i(int v) { return dpp_reduce_wave(v, OpMin()); }
__device__ __forceinline__ int wave_max_i(int v) { return dpp_reduce_wave(v, OpMax()); }

// Cluster table (pipe.py:78-102) by the two-level reduce-by-key above; called by all threads
// of a BIGTPB workgroup (sorted order keeps a cluster's PETs in neighbouring lanes, so a wave
// usually carries a handful of labels).
#ifndef TAB_BITS
#define TAB_BITS 9
#endif
#define TAB_H (1 << TAB_BITS)
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
    unsigned h = ((unsigned)key * 2654435761u) >> (32 - TAB_BITS);
    for (int probe = 0; probe < 16; ++probe) {
        const int old = atomicCAS(&keys[h], -1, key);
        if (old == -1 || old == key) return (int)h;
        h = (h + 1) & (TAB_H - 1);
    }
    return -1;
}
// Runs of equal values in the lanes of a wave, for a segmented reduction toward every run's LAST lane on the DPP network:
// dist = lanes of the run in front of this lane, last = the run ends here.  (All 64 lanes must be active.)
__device__ __forceinline__ int wave_shr1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }     // lane L <- lane L - 1 (lane 0 keeps its own)
struct WaveRuns { int dist; bool last; };
__device__ __forceinline__ WaveRuns wave_runs(int v)
{
    const int lane = threadIdx.x & 63;
    const int prev = wave_shr1(v);
    const unsigned long long B = __ballot(lane == 0 || prev != v);
    const int start = 63 - __clzll((long long)(B & (~0ull >> (63 - lane))));
    WaveRuns r;
    r.dist = lane - start;
    r.last = lane == 63 || ((B >> (lane + 1)) & 1ull);
    return r;
}
// one step of the segmented inclusive scan: lane L combines with the DPP source only if that lane belongs to its run.  A lane
// the network gives nothing (row edge, masked row) reads its own value back, and min / max with oneself change nothing.
#define SEG_STEP(ctrl, rowmask, cond, ...) seg_step_values<ctrl, rowmask>((cond), __VA_ARGS__)
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void seg_step_values(bool take, int& mn0, int& mx0, int& mn1, int& mx1)
{
    const int a = min(mn0, __builtin_amdgcn_update_dpp(mn0, mn0, CTRL, ROWMASK, 0xf, false));
    const int b = max(mx0, __builtin_amdgcn_update_dpp(mx0, mx0, CTRL, ROWMASK, 0xf, false));
    const int c = min(mn1, __builtin_amdgcn_update_dpp(mn1, mn1, CTRL, ROWMASK, 0xf, false));
    const int d = max(mx1, __builtin_amdgcn_update_dpp(mx1, mx1, CTRL, ROWMASK, 0xf, false));
    mn0 = take ? a : mn0; mx0 = take ? b : mx0; mn1 = take ? c : mn1; mx1 = take ? d : mx1;
}
template <int CTRL, int ROWMASK>
__device__ __forceinline__ void seg_step_min(bool take, int& m)
{
    const int a = min(m, __builtin_amdgcn_update_dpp(m, m, CTRL, ROWMASK, 0xf, false));
    m = take ? a : m;
}
// level 1 + insertion into the workgroup's LDS table (no barrier inside: may be called repeatedly; by whole waves)
__device__ __forceinline__ void table_accumulate(const Table& t, TableLds& h, int lab, int x, int y)
{
    if (!__any(lab >= 0)) return;
    // Sorted order keeps a cluster's PETs of one strip in NEIGHBOURING lanes: the wave is cut into runs of equal labels (the
    // inside of a large cluster: one run), every run is reduced toward its last lane -- row shifts, then the two row broadcasts,
    // a lane taking a partial only from a lane of its own run -- and that lane alone goes to the LDS table: one set of atomics
    // per run instead of one per PET, no two lanes of a run on one LDS address, and nothing through the LDS crossbar.
    const int lane = threadIdx.x & 63;
    const WaveRuns r = wave_runs(lab);
    int mnx = x, mxx = x, mny = y, mxy = y;
    SEG_STEP(0x111, 0xf, r.dist >= 1, mnx, mxx, mny, mxy);
    SEG_STEP(0x112, 0xf, r.dist >= 2, mnx, mxx, mny, mxy);
    SEG_STEP(0x114, 0xf, r.dist >= 4, mnx, mxx, mny, mxy);
    SEG_STEP(0x118, 0xf, r.dist >= 8, mnx, mxx, mny, mxy);
    SEG_STEP(0x142, 0xa, r.dist > (lane & 15), mnx, mxx, mny, mxy);            // row_bcast:15: the run began in front of this row
    SEG_STEP(0x143, 0xc, r.dist > (lane & 31), mnx, mxx, mny, mxy);            // row_bcast:31: ... in front of this half
    if (r.last && lab >= 0) {
        const int cm = r.dist + 1;
        const int sl = tab_slot(h.key, lab);
        if (sl >= 0) {
            atomicAdd(&h.cnt[sl], cm);
            atomicMin(&h.mnx[sl], mnx); atomicMax(&h.mxx[sl], mxx);
            atomicMin(&h.mny[sl], mny); atomicMax(&h.mxy[sl], mxy);
        } else {
            atomicAdd(&t.count[lab], cm);
            atomicMin(&t.minx[lab], mnx); atomicMax(&t.maxx[lab], mxx);
            atomicMin(&t.miny[lab], mny); atomicMax(&t.maxy[lab], mxy);
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



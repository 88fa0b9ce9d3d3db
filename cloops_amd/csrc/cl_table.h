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
    } else {
        // several clusters (and noise) in the wave.  Sorted order keeps a cluster's PETs of one strip in NEIGHBOURING lanes: the
        // wave is cut into runs of equal labels, every run is reduced over its lanes (segmented shuffles: a lane takes the
        // partial of the lane d further on while that lane still belongs to its run) and only the first lane of a run goes to
        // the LDS table -- one set of atomics per run instead of one per PET, and no two lanes of a run on one LDS address.
        const int prev = __shfl_up(lab, 1);
        const bool brk = lane == 0 || prev != lab;
        const unsigned long long B = __ballot(brk);
        const unsigned long long rest = lane == 63 ? 0ull : (B >> (lane + 1));
        const int nb = rest ? lane + __ffsll((long long)rest) : 64;        // first lane behind this lane's run
        int mnx = x, mxx = x, mny = y, mxy = y;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int a = __shfl_down(mnx, d), b = __shfl_down(mxx, d), c = __shfl_down(mny, d), e = __shfl_down(mxy, d);
            if (lane + d < nb) { mnx = min(mnx, a); mxx = max(mxx, b); mny = min(mny, c); mxy = max(mxy, e); }
        }
        if (brk && lab >= 0) {
            const int cm = nb - lane;
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


